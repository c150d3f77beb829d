"""Sweep scheduler knobs at 1080p (timing only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracingpbr_amd import Config, Renderer, cornell_box
cfg = Config.cornell_v3(1920, 1080, seed=0, max_raytrace=8)
sc = cornell_box("v3", aspect=1920 / 1080)
for wpc in (0, 16, 20, 24, 28, 32):
    for wl in (16, 24, 32):
        g = Renderer(sc, cfg); g.set_option("wait_lanes", wl); g.set_option("waves_per_cu", wpc)
        g.sample(4); g.sync(); g.sample(64); tr, tot, n = g.last_sample_ms(); c = g.counters()
        print(f"waves_per_cu={wpc:2d} wait_lanes={wl:2d} trace_ms={tr:8.2f} Msamples/s={c.samples / tr / 1e3:8.1f}", flush=True)
        g.close()
