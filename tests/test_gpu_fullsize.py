"""Every BASELINE.json config at its FULL size on the GPU (run with -m gpu).

The oracle cannot render these frames in seconds, so each config is covered three ways:
  (a) the whole frame at 1 spp, bit-exact against the oracle (every pixel's first sample);
  (b) a sparse subset of tiles at the config's full spp, bit-exact against the oracle (sample indices 0..spp-1 of
      those pixels: the same records the full-size run produces, by tile / launch-split invariance);
  (c) size-independent properties of the full-size run: count == spp everywhere, finiteness, additivity of
      progressive accumulation (a + b samples == a+b samples straight, bitwise), subset == full on the subset's
      pixels, gathered == untiled.
Configs: C3 glass bunny 1920x1080 / 1024 spp / 16 bounces / MAX_RAYMARCH 2048; C4 Tokyo-style IBL 3840x2160 /
512 spp / 3072x1536 env / 4 ranks; C5 Cornell 7680x4320 / 8 ranks, progressive; plus the largest untiled launch
(8K, the 32-bit work-item split).  C1 / C2 live in test_gpu_parity.py.
"""
import numpy as np
import pytest

from oracle_backend import OracleRenderer
from raytracingpbr_amd import SHAPE, Config, Renderer, bunny, cornell_box, src_scene
from raytracingpbr_amd.ibl import load_bunny_weights, synthetic_env
from raytracingpbr_amd.tiles import TileLayout

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


VARIANTS = ["aot", "jit_baked"]      # the ahead-of-time kernels, and what bench.py times: run-time compiled for the scene, baked


def hip(scene, cfg, setup, variant):
    g = Renderer(scene, cfg)
    setup(g)
    if variant == "jit_baked":
        g.set_option("jit", 2)            # strict: an error if the run-time instance cannot be used
        g.set_option("jit_bake", 2)
    else:
        g.set_option("jit", 0)            # the ahead-of-time instances, also where the library would compile one by itself
    return g


_ORACLE = {}


def oracle_once(key, make):
    """oracle results shared by the kernel variants of one test: (image_buffer, counters)"""
    if key not in _ORACLE:
        o = make()
        _ORACLE[key] = (o.image_buffer, counters(o))
    return _ORACLE[key]


def both(scene, cfg, setup, variant="aot"):
    g, o = hip(scene, cfg, setup, variant), OracleRenderer(scene, cfg)
    setup(o)
    return g, o


@pytest.fixture(autouse=True)
def _private_jit_cache(tmp_path, monkeypatch):
    monkeypatch.setenv("RTPBR_JIT_CACHE", str(tmp_path))


def counters(r):
    c = r.counters()
    return (c.samples, c.raycasts, c.march_steps, c.hits, c.sky_lookups, c.deposits)


# ---------------------------------------------------------------------------------------------- C3
@pytest.mark.parametrize("variant", VARIANTS)
def test_c3_glass_bunny_1080p_1024spp(variant):
    W, H, SPP = 1920, 1080, 1024
    sc = bunny(aspect=W / H)
    cfg = Config.bunny_glass(W, H, seed=0, max_raytrace=16, frame=0)
    assert cfg.max_raymarch == 2048
    env = synthetic_env(3072, 1536, seed=0)            # the environment bench.py's `c3` times (raytracingpbr_amd/workloads.py)

    def setup(r):
        r.set_env(env, 1.8, 2.2)
        r.set_shape_data(SHAPE.BUNNY, load_bunny_weights())
    def o_full():
        o = OracleRenderer(sc, cfg); setup(o); o.sample(1)
        return o

    def o_sub():
        o = OracleRenderer(sc, cfg); setup(o); o.set_tiles(16, 16, 437, 1020); o.sample(SPP)
        return o
    # (a) full frame, first sample of every pixel
    g = hip(sc, cfg, setup, variant)
    g.sample(1)
    assert g.counter("jit_active") == (1 if variant == "jit_baked" else 0)
    want, want_ctr = oracle_once("c3_full", o_full)
    assert counters(g) == want_ctr
    assert np.array_equal(bits(g.image_buffer), bits(want))
    # (b) 8 tiles of 16x16 (one of 1020 ranks), all 1024 samples
    world = 1020
    gs = hip(sc, cfg, setup, variant)
    gs.set_tiles(16, 16, 437, world)
    gs.sample(SPP)
    sub = gs.image_buffer
    want, want_ctr = oracle_once("c3_sub", o_sub)
    assert counters(gs) == want_ctr
    assert np.array_equal(bits(sub), bits(want))
    own = TileLayout(W, H, 16, 16, world).owner_map() == 437
    assert own.sum() == 8 * 256 and np.all(sub[own][:, 3] == SPP)
    # (c) the full config: 2.1 G samples
    g.sample(SPP - 1)                                       # 1 + 1023: progressive accumulation
    full = g.image_buffer
    assert np.all(full[..., 3] == SPP) and np.all(np.isfinite(full))
    assert np.array_equal(bits(full[own]), bits(sub[own]))
    c = g.counters()
    assert c.samples == W * H * (SPP - 1)
    g.post_process()
    px = g.image_pixels
    assert np.all(np.isfinite(px)) and px.min() >= 0.0 and px.max() <= 1.0


# ---------------------------------------------------------------------------------------------- C4
@pytest.mark.parametrize("variant", VARIANTS)
def test_c4_tokyo_ibl_4k_four_ranks(variant):
    W, H, SPP, G = 3840, 2160, 512, 4
    sc = src_scene(aspect=W / H, tokyo=True)
    cfg = Config.tokyo_ibl(W, H, seed=0, max_raytrace=512)
    env = synthetic_env(3072, 1536, seed=0)                 # the "3k" map of the config

    def setup(r):
        r.set_env(env, 1.8, 2.2)
    # (b) sparse subset at full spp vs the oracle (exercises the 57 MB env gather on the device)
    world = 4050

    def o_sub():
        o = OracleRenderer(sc, cfg); setup(o); o.set_tiles(16, 16, 1234, world); o.sample(SPP)
        return o
    gs = hip(sc, cfg, setup, variant)
    gs.set_tiles(16, 16, 1234, world)
    gs.sample(SPP)
    assert gs.counter("jit_active") == (1 if variant == "jit_baked" else 0)
    sub = gs.image_buffer
    want, want_ctr = oracle_once("c4_sub", o_sub)
    assert counters(gs) == want_ctr
    assert np.array_equal(bits(sub), bits(want))
    own_sub = TileLayout(W, H, 16, 16, world).owner_map() == 1234
    # (c) gathered frame == untiled frame at 2 spp: G virtual ranks render, pack, rank 0 unpacks
    import torch
    lay = TileLayout(W, H, 32, 32, G)
    ref = hip(sc, cfg, setup, variant)
    ref.sample(2)
    want = ref.image_buffer
    ref.close()
    root = hip(sc, cfg, setup, variant)
    root.set_tiles(32, 32, 0, G)
    root.sample(2)
    for rank in range(1, G):
        r = hip(sc, cfg, setup, variant)
        r.set_tiles(32, 32, rank, G)
        r.sample(2)
        buf = torch.empty(lay.packed_pixels * 4, dtype=torch.float32, device="cuda")
        r.pack_tiles(buf.data_ptr())
        r.sync()
        root.unpack_tiles(buf.data_ptr(), rank)
        root.sync()
        r.close()
    assert np.array_equal(bits(root.image_buffer), bits(want))
    # (c) rank 0's full share of the config: 1.06 G samples
    root.refresh()
    root.set_option("sample_base", 0)
    root.sample(SPP)
    share = root.image_buffer
    mine = lay.owner_map() == 0
    assert np.all(share[mine][:, 3] == SPP) and np.all(share[~mine] == 0) and np.all(np.isfinite(share))
    both_ = mine & own_sub
    assert both_.any() and np.array_equal(bits(share[both_]), bits(sub[both_]))
    assert root.counters().sky_lookups > 0.3 * mine.sum() * SPP


# ---------------------------------------------------------------------------------------------- C5
@pytest.mark.parametrize("variant", VARIANTS)
def test_c5_cornell_8k_eight_ranks_progressive(variant):
    W, H, G = 7680, 4320, 8
    sc = cornell_box("v3", aspect=W / H)
    cfg = Config.cornell_v3(W, H, seed=0, max_raytrace=8)
    lay = TileLayout(W, H, 32, 32, G)
    mine = lay.owner_map() == 0
    # rank 0 of 8: two progressive calls of 256 spp (the config accumulates 16 of them)
    nothing = lambda r: None
    g = hip(sc, cfg, nothing, variant)
    g.set_tiles(32, 32, 0, G)
    g.sample(256)
    g.sample(256)
    assert g.counter("jit_active") == (1 if variant == "jit_baked" else 0)
    two = g.image_buffer
    assert np.all(two[mine][:, 3] == 512) and np.all(two[~mine] == 0) and np.all(np.isfinite(two))
    assert g.packed_bytes() == lay.packed_pixels * 16
    # additivity: 512 straight == 256 + 256 (sample indices continue)
    s = hip(sc, cfg, nothing, variant)
    s.set_tiles(32, 32, 0, G)
    s.sample(512)
    assert np.array_equal(bits(s.image_buffer), bits(two))
    s.close()
    # (b) sparse subset (one of 16200 ranks of 16x16 tiles = 8 tiles) at 512 spp vs the oracle
    world = 16200

    def o_sub():
        o = OracleRenderer(sc, cfg); o.set_tiles(16, 16, 7777, world); o.sample(512)
        return o
    gs = hip(sc, cfg, nothing, variant)
    gs.set_tiles(16, 16, 7777, world)
    gs.sample(512)
    sub = gs.image_buffer
    want, want_ctr = oracle_once("c5_sub", o_sub)
    assert counters(gs) == want_ctr
    assert np.array_equal(bits(sub), bits(want))
    own_sub = TileLayout(W, H, 16, 16, world).owner_map() == 7777
    both_ = mine & own_sub
    assert both_.any() and np.array_equal(bits(two[both_]), bits(sub[both_]))
    g.close()
    gs.close()


def test_untiled_8k_launch_and_work_item_split():
    """One untiled 7680x4320 frame: 33 M pixels per launch; 160 spp are 5.3 G work items (> 2^32), so the call must
    split itself into launches and still equal progressive accumulation bit for bit (checked on the full frame)."""
    W, H = 7680, 4320
    sc = cornell_box("v3", aspect=W / H)
    cfg = Config.cornell_v3(W, H, seed=0, max_raytrace=8)
    a = Renderer(sc, cfg)
    a.sample(1)
    o = OracleRenderer(sc, cfg)
    o.set_tiles(16, 16, 3, 16200)                     # the oracle checks a sparse subset of the untiled frame
    o.sample(1)
    own = TileLayout(W, H, 16, 16, 16200).owner_map() == 3
    one = a.image_buffer
    assert np.array_equal(bits(one[own]), bits(o.image_buffer[own]))
    assert np.all(one[..., 3] == 1.0)
    a.sample(159)                                      # 33.2 M x 159 = 5.3 G items > 2^32: internal split
    b = Renderer(sc, cfg)
    b.sample(40)
    b.sample(120)
    fa, fb = a.image_buffer, b.image_buffer
    assert np.all(fa[..., 3] == 160.0) and np.array_equal(bits(fa), bits(fb))
    a.close()
    b.close()
