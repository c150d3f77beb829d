"""How long does a 1-rank RCCL communicator take to initialise on this box, per environment setting?"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from cases import case_by_name
from raytracingpbr_amd import Renderer
case = case_by_name("cornell_v3_8b_wide")
r = Renderer(case.scene, case.cfg); r.set_tiles(16, 16, 0, 1)
t = time.time(); uid = r.rccl_unique_id(); t1 = time.time() - t
t = time.time(); r.rccl_init(uid, 0, 1); t2 = time.time() - t
r.sample(2)
t = time.time(); r.gather_tiles(); r.sync(); t3 = time.time() - t
print("unique_id %%.2f s, init %%.2f s, first gather %%.2f s" %% (t1, t2, t3), flush=True)
''' % (ROOT, ROOT)
envs = [{"NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1", "RCCL_MSCCL_ENABLE": "0", "RCCL_MSCCLPP_ENABLE": "0", "HSA_NO_SCRATCH_RECLAIM": "1"},
        {"RCCL_MSCCL_ENABLE": "0", "RCCL_MSCCLPP_ENABLE": "0"},
        {"NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1"},
        {}]
for e in envs:
    env = dict(os.environ); env.update(e)
    t = time.time()
    try:
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=400)
        print(e, "->", out.stdout.strip().split("\n")[-1] if out.stdout.strip() else out.stderr[-300:], "total %.1f s" % (time.time() - t), flush=True)
    except subprocess.TimeoutExpired:
        print(e, "-> timeout after 400 s", flush=True)
