"""RCCL communicator set-up time on this box (1-rank communicator — the box has one GPU), three ways:
  c_abi_rocm   rtpbr_rccl_init through the C ABI with ROCm's /opt/rocm/lib/librccl.so.1   (bench.py's default transport)
  c_abi_wheel  the same entry points with RTPBR_RCCL_LIB = the librccl bundled in the PyTorch wheel
  torch_nccl   torch.distributed.init_process_group("nccl") + first gather               (bench.py --transport torch)
Each in its own process with a time limit; writes gpurun_out/r03_rccl_init.json."""
import glob, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIMIT = int(os.environ.get("LIMIT", "420"))
c_abi = r'''
import sys, time, json
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from cases import case_by_name
from raytracingpbr_amd import Renderer
case = case_by_name("cornell_v3_8b_wide")
r = Renderer(case.scene, case.cfg); r.set_tiles(16, 16, 0, 1)
t = time.time(); uid = r.rccl_unique_id(); t1 = time.time() - t
t = time.time(); r.rccl_init(uid, 0, 1); r.sync(); t2 = time.time() - t
r.sample(2)
t = time.time(); r.gather_tiles(); r.sync(); t3 = time.time() - t
t = time.time(); r.gather_tiles(); r.sync(); t4 = time.time() - t
n, rk, ver = r.rccl_info()
print(json.dumps({"unique_id_s": round(t1, 3), "comm_init_s": round(t2, 3), "first_gather_s": round(t3, 4), "second_gather_s": round(t4, 5),
                  "rccl_nranks": n, "rccl_version": ver}), flush=True)
''' % (ROOT, ROOT)
torch_nccl = r'''
import os, time, json, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
t0 = time.time()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t1 = time.time()
x = torch.ones(1 << 20, device="cuda"); out = [torch.empty_like(x)]
dist.gather(x, out, dst=0); torch.cuda.synchronize(); t2 = time.time()
print(json.dumps({"init_process_group_s": round(t1 - t0, 3), "first_gather_s": round(t2 - t1, 3)}), flush=True)
dist.destroy_process_group()
'''
import torch
wheel = sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")))
res = {"limit_s": LIMIT, "wheel_librccl": wheel[0] if wheel else None, "torch": torch.__version__}
legs = [("c_abi_rocm", c_abi, {}), ("c_abi_wheel", c_abi, {"RTPBR_RCCL_LIB": wheel[0]} if wheel else None), ("torch_nccl", torch_nccl, {})]
for name, code, extra in legs:
    if extra is None:
        res[name] = "no bundled librccl found"
        continue
    env = dict(os.environ); env.update(extra)
    t = time.time()
    try:
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=LIMIT)
        line = [l for l in out.stdout.split("\n") if l.startswith("{")]
        res[name] = json.loads(line[-1]) if line else {"error": out.stderr[-400:]}
    except subprocess.TimeoutExpired:
        res[name] = {"timeout_after_s": LIMIT}
    res[name + "_process_s"] = round(time.time() - t, 1)
    print(name, res[name], flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03_rccl_init.json"), "w"), indent=1)
