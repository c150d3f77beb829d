"""BUILD CONTAINER (no GPU needed): compile the BAKED run-time instance of a workload the way rtpbr_sample() would and
print the register / scratch / LDS use of its kernels; with --asm also dump the disassembly.
    python tools/jit_offline.py src [W H] [--waves N] [--fast] [--asm out.s]
"""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import workloads
from raytracingpbr_amd._capi import HIP_LIB_PATH
from raytracingpbr_amd.config import Config
from raytracingpbr_amd.dataclass import SDFObject

args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if args else "src"
W, H = (int(args[1]), int(args[2])) if len(args) >= 3 else (0, 0)
waves = int(sys.argv[sys.argv.index("--waves") + 1]) if "--waves" in sys.argv else 0
fast = 1 if "--fast" in sys.argv else 0
wl = workloads.get(name, W, H)
lib = C.CDLL(HIP_LIB_PATH)
lib.rtpbr_last_error.restype = C.c_char_p
n = len(wl.scene.objects)
arr = (SDFObject * n)(*wl.scene.objects)
buf = C.create_string_buffer(1024)
cache = os.environ.setdefault("RTPBR_JIT_CACHE", tempfile.mkdtemp(prefix="rtpbr_jit_"))
lib.rtpbr_test_jit_build_baked.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Config), C.c_int, C.c_int, C.c_char_p, C.c_size_t]
rc = lib.rtpbr_test_jit_build_baked(arr, n, 1 if wl.scene.scale10 else 0, C.byref(wl.cfg), waves, fast, buf, 1024)
if rc:
    raise SystemExit("build failed: %s" % lib.rtpbr_last_error().decode())
path = buf.value.decode()
print(path)
t = path + ".o"
subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                "--input=" + path, "--output=" + t], check=True)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", t], capture_output=True, text=True, check=True).stdout
keys = (".name:", ".vgpr_count", ".sgpr_count", "spill_count", "private_segment_fixed", "group_segment_fixed")
rows = [l.strip() for l in notes.splitlines() if any(k in l for k in keys)]
for i in range(0, len(rows), 7):
    print("\t".join(rows[i:i + 7]))
if "--asm" in sys.argv:
    out = sys.argv[sys.argv.index("--asm") + 1]
    with open(out, "w") as f:
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", t], stdout=f, check=True)
    print("disassembly ->", out)
