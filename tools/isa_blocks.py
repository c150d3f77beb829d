"""Dump basic-block sizes / opcode histogram of one kernel from the -save-temps .s file."""
import re, sys, subprocess, os, collections
src = sys.argv[1] if len(sys.argv) > 1 else "/root/repo/raytracingpbr_amd/csrc/rt_kernels.hip"
kern = sys.argv[2] if len(sys.argv) > 2 else "_ZN2rt11trace_pathsILi1ELi8EEEvNS_6ParamsE"
os.makedirs("/tmp/probe", exist_ok=True)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-use-amdgpu-trackers=1", "-save-temps", "-c", src, "-o", "k.o"],
               cwd="/tmp/probe", stderr=subprocess.DEVNULL)
s = open("/tmp/probe/" + os.path.basename(src).replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s").read()
m = re.search(r"^" + re.escape(kern) + r":(.*?)\.Lfunc_end", s, re.S | re.M)
body = m.group(1).split("\n")
blocks, cur = [], ["entry", [], []]
for l in body:
    t = l.strip()
    mm = re.match(r"^(\.LBB\d+_\d+):", t)
    if mm:
        blocks.append(cur); cur = [mm.group(1), [], []]
    elif t and not t.startswith((";", ".")):
        cur[1].append(t)
        if t.startswith(("s_cbranch", "s_branch")): cur[2].append(t.split()[0].replace("s_cbranch_", "") + "->" + t.split()[-1])
blocks.append(cur)
tot = sum(len(b[1]) for b in blocks)
print("total", tot)
for name, ins, br in blocks:
    if len(ins) >= int(os.environ.get("MIN", 40)):
        c = collections.Counter(i.split()[0] for i in ins)
        print(name, len(ins), br, c.most_common(10))
