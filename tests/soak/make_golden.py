"""Generates tests/golden/<case>.npz with the CPU oracle (K2 fixtures).

These are outputs of this repo's own oracle (the reference cannot run here: Taichi is not
installed), committed so that (a) the oracle is guarded against drift across machines and
compilers and (b) the GPU tests have a second, oracle-independent comparison target.
  python tests/soak/make_golden.py [case-name ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import all_cases, fingerprint          # noqa: E402
from oracle_backend import OracleRenderer         # noqa: E402

want = set(sys.argv[1:])
for case in all_cases():
    if want and case.name not in want:
        continue
    t = time.time()
    r = OracleRenderer(case.scene, case.cfg)
    case.run(r)
    fp = fingerprint(r)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", case.name + ".npz"), **fp)
    print(f"{case.name}: {time.time() - t:.1f}s counters={fp['counters'].tolist()}")
