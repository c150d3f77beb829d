"""Per-kernel device time of rank 0's share of the headline frame under the tile partition (G virtual ranks), against 1/G of the
whole frame: where the strong-scaling loss of small launches sits.  python tools/gpu_share.py [KEY=VALUE ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box
from raytracingpbr_amd.tiles import default_tile
W, H, SPP = 1920, 1080, 256
opts = dict(kv.split("=") for kv in sys.argv[1:])
base = None
for G in (1, 2, 4, 8, 16):
    r = Renderer(cornell_box("v3", aspect=W / H), Config.cornell_v3(W, H, 0, 8))
    r.set_option("jit", 1); r.set_option("jit_bake", 1)
    for k, v in opts.items(): r.set_option(k, int(v))
    if G > 1:
        tw, th = default_tile(W, H, G)
        r.set_tiles(tw, th, 0, G)
    r.set_option("reserve_spp", SPP)
    r.sample(SPP); r.sync()
    best = None
    for _ in range(3):
        r.refresh(); r.sample(SPP)
        tr, tot, n = r.last_sample_ms(); pr, pn = r.last_primary_ms()
        if best is None or tot < best[1]: best = (tr, tot, pr)
    tr, tot, pr = best
    if base is None: base = best
    print(f"G={G:2d}: total {tot:7.2f} ms (ideal {base[1] / G:6.2f}, x{tot / (base[1] / G):.3f})  trace {tr:6.2f} (x{tr / (base[0] / G):.3f})  primary {pr:6.2f} (x{pr / (base[2] / G):.3f})  rest {tot - tr - pr:5.2f}", flush=True)
    r.close()
