#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

Default workload (BASELINE.json configs[1]): Cornell Box 1920x1080, 256 spp, 8 bounces, the cornell_box_v3 variant of
the reference (examples/cornell_box/cornell_box_v3).  One "step" = one complete render of that frame: refresh, 256
samples per pixel through the HIP kernels (+ ordered accumulation), and for N > 1 the single RCCL gather of the per-tile
radiance to rank 0.  Inputs (scene constants, camera, environment) are resident on the device before the timed region.
N ranks share the FIXED frame (tiles dealt round-robin) -> strong scaling.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.

  --workload c2 (default)  Cornell Box 1920x1080, 256 spp, 8 bounces          (configs[1], the metric)
  --workload c1            Cornell Box 256x256, 16 spp, 4 bounces             (configs[0])
  --workload c3            SDF glass bunny 1920x1080, 1024 spp, 16 bounces    (configs[2])
  --workload c4            Tokyo IBL 3840x2160, 512 spp, 3k env               (configs[3])
  --workload c5            Cornell Box 7680x4320, 256 spp per step            (configs[4]'s frame; its 4096 spp = 16 steps)
  --workload src           src/ persistent-ray pipeline 1920x1080, 256 bounce-steps per pixel and step

`roofline` is the bound that applies — FP32 vector (VALU) issue: algorithmic FLOPs of SURVEY.md 8(d) per launch / the
kernels' HIP-event time vs the 157.3 TFLOP/s FP32 vector peak — and carries `traffic`, the HBM bytes per launch of the
dominant kernel from the committed PMC passes.  `hbm` is the view the metric's name asks for.  At N = 1 the default
line also carries `configs`: every other BASELINE config (multi-GPU ones as rank 0's share) and the src/ form, each at
a bounded sample count with its own FLOP model and roofline fraction; `jit`: what the run-time compiled kernels cost
at first use and what the same step reaches without them; `cpu_baseline`: the CPU restatement timed on this host.

N > 1: the data path's one collective runs through the product's own C ABI (rtpbr_rccl_init / rtpbr_gather_tiles =
ncclCommInitRank / ncclGather of ROCm's librccl); torch.distributed (gloo) is only the control plane that ships the
128-byte id, the barriers and the max-over-ranks time.  --transport torch uses torch.distributed's nccl backend for the
gather instead; --transport host (functional tests on one device or CPU-side) stages through host tensors.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector (v_fma_f32 wave64 = 2 cycles on SIMD-32)
VALU_UBENCH_TFLOPS = 103.0     # tools/ubench/valu_rate.hip on this chip: a pure v_fma_f32 stream issues every 2.45 cycles at the ~1.95 GHz it sustains


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5", "src"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (src: bounce-steps per pixel) of one step; 0 = the config's")
    ap.add_argument("--bounces", type=int, default=0)
    ap.add_argument("--scheduler", type=int, default=-1, help="0 = in-register refill, 1 = LDS ray pool (library default)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="rtpbr_set_option knob (A/B runs), repeatable")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config block of the default line")
    ap.add_argument("--no-jit", action="store_true", help="use the ahead-of-time kernels instead of the run-time compiled, scene-specialised ones")
    ap.add_argument("--keep-jit-cache", action="store_true", help="use the user's code-object cache (default: a fresh one, so that first-use compile time is measured)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch", "host"],
                    help="N > 1 gather: rccl = the library's own ncclGather (default), torch = torch.distributed nccl, host = gloo through host memory")
    ap.add_argument("--backend", default=None, help="(compatibility) nccl = --transport torch, gloo = --transport host")
    ap.add_argument("--collective-at-1", action="store_true", help="with --gpus 1: still set up a (1-rank) RCCL communicator and run the gather every step")
    ap.add_argument("--same-device", action="store_true", help="functional test: all ranks share GPU 0 (--transport host, or --transport rccl with RTPBR_RCCL_LIB "
                                                                "pointing at tests/stubs/libfake_rccl.so: RCCL itself refuses two ranks on one device)")
    ap.add_argument("--check-gather", action="store_true", help="(the default for N > 1) rank 0 also renders the frame untiled, OUTSIDE the timed region, and reports whether the gathered frame equals it bit for bit")
    ap.add_argument("--no-check-gather", action="store_true", help="N > 1: skip the untiled reference render")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    a = ap.parse_args()
    if a.backend == "gloo":
        a.transport = "host"
    elif a.backend == "nccl":
        a.transport = "torch"
    return a


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs but the container is limited by cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_baseline(wl, budget_s):
    """CPU restatement (kind 'port') on the host cores, bounded sample of the same workload.  Timed with the
    -O3 -march=native build of the same source (oracle/Makefile target `fast`, compiled HERE for this host's CPU;
    the exactly-rounded -O2 build stays the parity checker); falls back to the checker build if gcc is missing."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_backend
    from oracle_backend import OracleRenderer
    cores = usable_cores()
    build = "-O2 -march=x86-64-v3 exact build"
    try:
        import subprocess
        subprocess.run(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "fast"], check=True, capture_output=True, timeout=120)   # -B: -march=native must be built on THIS host
        oracle_backend.use_library(os.path.join(ROOT, "oracle", "librt_oracle_fast.so"))
        build = "-O3 -march=native build"
    except Exception:
        pass
    cfg = wl.cfg
    o = OracleRenderer(wl.scene, cfg, threads=cores)
    wl.setup(o)
    t0 = time.perf_counter()
    o.sample(1)
    t1 = time.perf_counter() - t0
    n = int(max(1, min(64, (budget_s - t1) / max(t1, 1e-3))))
    t0 = time.perf_counter()
    o.sample(n)
    dt = time.perf_counter() - t0
    samples = cfg.width * cfg.height * n
    what = "bounce-steps" if wl.family == "src" else "spp"
    return {"value": round(samples / dt / 1e6, 4), "unit": wl.unit, "cores": cores, "kind": "port",
            "sample": f"{cfg.width}x{cfg.height} x {n} {what}, MAX_RAYTRACE {cfg.max_raytrace} ({samples / 1e6:.1f} M units, {dt:.1f} s), "
                      f"C restatement of the reference path ({build}) with OpenMP over {cores} threads (Taichi unavailable)"}


def make_renderer(wl, device, a, jit=True, bake=True):
    from raytracingpbr_amd import Renderer
    r = Renderer(wl.scene, wl.cfg, device=device)
    wl.setup(r)
    if a.scheduler >= 0:
        r.set_option("scheduler", a.scheduler)
    if jit:
        # production setting for an offline render of a fixed scene: the kernels compiled at run time for THIS scene and
        # configuration (rt_jit.hip: ~1-2 s once, then a disk cache), as Taichi JIT-compiles the reference's kernels; falls
        # back to the ahead-of-time instance if hipcc is not available on the box
        r.set_option("jit", 1)
        # bake: True / 2 = scene, configuration AND the camera frame baked (a fixed-camera offline render), 1 = scene and
        # configuration only (the camera stays a launch argument: what an interactive host uses), False / 0 = nothing baked
        r.set_option("jit_bake", 2 if bake is True else int(bake))
    else:
        r.set_option("jit", 0)
    for kv in a.opt:
        k, v = kv.split("=")
        r.set_option(k, int(v))
    return r


def measure(wl, r, steps, warmup, fence=None, gather=None, one_step_launches=False, sync_every_step=True):
    """W warm-up + K timed steps of one workload on an already configured renderer; returns the wall time and the per-launch
    figures of the kernels (HIP events recorded on the context's own stream by the library).  sync_every_step = False: the K
    steps are enqueued back to back and the host waits once, at the end (what a host that does not look at every step does; for
    sub-millisecond steps — C1 — reading the events after every step costs a fifth of the step); the kernel figures are then
    those of the LAST step, scaled."""
    form_src = wl.family == "src"

    def step():
        r.refresh()
        if one_step_launches:
            for _ in range(wl.spp):
                r.sample(1)
        else:
            r.sample(wl.spp)
        if gather is not None:
            gather()

    fence = fence or r.sync
    # Sub-millisecond launches (C1's steps, the one-step src/ launches) are timed WITHOUT the library's per-kernel events (option
    # timing = 0: what a host that does not ask for kernel times sets; an event is ~4 us of idle queue between two small kernels);
    # their kernel figures come from one further step with the events on, outside the timed region, scaled.
    events_off = one_step_launches or not sync_every_step
    if events_off:
        r.set_option("timing", 0)
    for _ in range(warmup):
        step()
    fence()
    trace_ms = primary_ms = 0.0
    launches = primary_launches = 0
    t0 = time.perf_counter()
    if events_off:
        for _ in range(steps):
            step()
            if sync_every_step:
                fence()
        fence()
        dt = time.perf_counter() - t0
        r.set_option("timing", 1)
        step()
        tr, _tot, n = r.last_sample_ms()
        if one_step_launches:        # (the library keeps the events of the last rtpbr_sample() call only: scale the last launch)
            tr, n = tr * wl.spp, n * wl.spp
        pr, pn = (0.0, 0) if form_src else r.last_primary_ms()
        return {"dt": dt, "trace_ms": tr * steps, "launches": n * steps, "primary_ms": pr * steps, "primary_launches": pn * steps,
                "events": "off in the timed region (option timing = 0); kernel figures from one further step, scaled"}
    for _ in range(steps):
        step()
        # HIP-event timing of the kernels; reading it waits for the step, which the timed region must wait for anyway
        tr, _tot, n = r.last_sample_ms()
        trace_ms += tr
        launches += n
        if not form_src:
            pr, pn = r.last_primary_ms()
            primary_ms += pr
            primary_launches += pn
    fence()
    dt = time.perf_counter() - t0
    return {"dt": dt, "trace_ms": trace_ms, "launches": launches, "primary_ms": primary_ms, "primary_launches": primary_launches}


def roofline_of(wl, r, m, steps, pixels):
    """the VALU roofline of one measured workload on this rank: algorithmic FLOPs of the last step / kernel time"""
    c = r.counters()
    mlp = r.counter("mlp_lane_evals") if wl.family == "bunny" else 0
    fpu = wl.flop_per_unit(c, mlp)
    units_per_step = c.samples
    kernel_s = (m["trace_ms"] + m["primary_ms"]) / 1e3 / steps
    tflops = fpu * units_per_step / max(kernel_s, 1e-9) / 1e12
    return c, fpu, kernel_s, tflops


def side_config(name, a, device, rank0_of=1):
    """one entry of the default line's `configs` block: the named BASELINE config (rank 0's share of a multi-GPU one) at a
    bounded sample count, with its own FLOP model"""
    from raytracingpbr_amd import workloads
    from raytracingpbr_amd.tiles import default_tile
    fast = name.endswith("_fast")
    name_ = name[:-5] if fast else name
    bounded = {"c1": 16, "c2": 256, "c3": 256, "c3_valu": 256, "c4": 256, "c5": 128, "src": 256, "src_4k": 256, "src_768": 256, "src_1step": 256}[name_]
    base = {"src_4k": "src", "src_768": "src", "src_1step": "src", "c3_valu": "c3"}.get(name_, name_)
    dims = {"src_4k": (3840, 2160), "src_768": (768, 432)}.get(name_, (0, 0))
    wl = workloads.get(base, dims[0], dims[1], bounded)
    r = make_renderer(wl, device, a, jit=not a.no_jit)
    if fast:
        # the tolerance flavour of the same kernels (option precision = 1: hardware sqrt / rcp / sin / exp, contraction — the
        # regime the reference itself runs in, Taichi's default fast_math); NOT bit-exact, held to the north star's per-pixel
        # L2 < 1e-3 against the oracle by tests/test_gpu_fast.py; the headline stays the exact kernels
        r.set_option("precision", 1)
    if name == "c3_valu":
        # north_star: "no MFMA".  The same network on the vector ALU only — SAME kernel (LDS ray pool, half-pass policy, baked
        # configuration), only the 8 MFMAs per layer replaced by fma chains — beside the default instance whose hidden layers
        # run as f32 MFMA (bit-identical results; same FP32 peak rate): the A/B that backs the choice
        r.set_option("mlp_mfma", 0)
    one_step_launches = name_ == "src_1step"      # the way the reference calls it: ONE bounce-step per pathtrace() launch (src/renderer.py:29-30)
    W, H = wl.cfg.width, wl.cfg.height
    share = ""
    if wl.virtual_world > 1:
        tw, th = default_tile(W, H, wl.virtual_world)
        r.set_tiles(tw, th, 0, wl.virtual_world)
        share = f", rank 0 of {wl.virtual_world} ({tw}x{th} tiles dealt round-robin)"
    if wl.family != "src":
        r.set_option("reserve_spp", wl.spp)
    r.sample(wl.spp)
    r.sync()
    steps = 20 if name_ == "c1" else 2
    # (C1 is a 0.5 ms launch: its 20 steps are enqueued back to back, one wait at the end)
    # (the fused src/ launches settle over three launches: cost plan, then room for the chain kernel once a plan has produced a
    # chain set, then the age-weighted shares of the new grid)
    m = measure(wl, r, steps, 3 if (wl.family == "src" and not one_step_launches) else 1, one_step_launches=one_step_launches, sync_every_step=name_ != "c1")
    c, fpu, kernel_s, tflops = roofline_of(wl, r, m, steps, W * H)
    if one_step_launches:       # the counters are those of the LAST launch (one bounce-step per pixel)
        kernel_s = m["trace_ms"] / 1e3 / steps
        tflops = fpu * c.samples * wl.spp / max(kernel_s, 1e-9) / 1e12
        c.samples *= wl.spp
    units = c.samples * steps
    title = wl.title.replace(f"{workloads.get(wl.name).spp} spp", f"{wl.spp} spp") + share
    if one_step_launches:
        title = title.replace(f"{wl.spp} launches fused", f"{wl.spp} separate launches of ONE bounce-step each (rtpbr_sample(ctx, 1), as src/renderer.py:29-30 calls pathtrace())")
    if name == "c3_valu":
        title += "; option mlp_mfma = 0: the network on the vector ALU only"
    if fast:
        title += "; option precision = 1: tolerance flavour (per-pixel L2 < 1e-3 vs the oracle, tests/test_gpu_fast.py), not bit-exact"
    out = {"workload": title,
           "value": round(units / m["dt"] / 1e6, 1), "unit": wl.unit, "units_per_step": c.samples, "steps": steps,
           "ms_per_step": round(m["dt"] / steps * 1e3, 3), "kernel_ms_per_step": round(kernel_s * 1e3, 3),
           "algorithmic_flop_per_unit": round(fpu), "achieved_tflops": round(tflops, 2), "frac": round(tflops / VALU_PEAK_TFLOPS, 4),
           "raycasts_per_unit": round(c.raycasts / max(c.samples, 1), 3), "march_steps_per_raycast": round(c.march_steps / max(c.raycasts, 1), 2),
           "run_time_kernels": bool(r.counter("jit_active"))}
    if wl.family == "bunny":
        out["mlp_evaluations_per_unit"] = round(r.counter("mlp_lane_evals") / max(c.samples, 1), 2)
    if name == "c2_fast":
        # HBM traffic of this flavour's step from the committed PMC passes (rocprofv3 cannot run inside this process): FETCH x 2 + WRITE
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_fast.json")))
            out["hbm_fast"] = {"hbm_bytes_per_step": tj["fast"]["hbm_bytes_per_step"], "times_algorithmic": tj["fast"]["times_algorithmic"],
                               "exact_kernels_hbm_bytes_per_step": tj["exact"]["hbm_bytes_per_step"], "exact_times_algorithmic": tj["exact"]["times_algorithmic"],
                               "kind": "a COMMITTED constant from profiles/hbm_traffic_fast.json (separate FETCH_SIZE / WRITE_SIZE passes), not a measurement of this run",
                               "note": "no staging in this flavour (LDS accumulators + f32 atomics); what remains is the primary records, 5 B per sample written and read"}
        except Exception:
            pass
    if "events" in m:
        out["events"] = m["events"]
    if name == "c3_valu":
        out["differs_from_c3_in"] = "mlp_mfma only (same pool kernel, same pass policy, same run-time instance)"
    r.close()
    return out


def tolerance_block(wl, a, device, torch):
    """The tolerance flavour (option precision = 1) of the headline as a first-class, SELF-CHECKING result: measured with the same
    --steps / --warmup as `value`, with its own roofline, and — outside the timed region, in this very run — held against the
    exact kernels: both flavours render the same frame (same sample indices), the display-space per-pixel L2 between the two
    image_pixels is computed ON THE DEVICE (rtpbr_buffer_device_ptr -> torch tensors on the renderers' own memory), and the flip
    rate (samples whose colour differs beyond rounding) is counted on 16 single-sample frames.  The exact kernels are the
    oracle's bits (tests/test_gpu_parity.py, test_gpu_fullsize.py), so this is the distance to the oracle's frame."""
    from raytracingpbr_amd.renderer import BUF_IMAGE_BUFFER, BUF_IMAGE_PIXELS
    W, H, SPP = wl.cfg.width, wl.cfg.height, wl.spp
    rf = make_renderer(wl, device, a, jit=True)
    rf.set_option("precision", 1)
    rf.set_option("reserve_spp", SPP)
    rf.sample(SPP)
    rf.sync()
    m = measure(wl, rf, a.steps, a.warmup)
    c, fpu, kernel_s, tflops = roofline_of(wl, rf, m, a.steps, W * H)
    out = {"what": "option precision = 1: hardware sqrt / rsq / rcp / exp, contraction, no exact decision bands (rt_math.hpp RT_FAST_MATH) — the regime the "
                   "reference itself runs in (Taichi's default fast_math, src/config.py:5); held to the north star's per-pixel L2 < 1e-3; `value` above stays the exact kernels",
           "value": round(W * H * SPP * a.steps / m["dt"] / 1e6, 2), "unit": wl.unit, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(m["dt"] / a.steps * 1e3, 3),
           "roofline": {"bound": "valu", "achieved": round(tflops, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / VALU_PEAK_TFLOPS, 4),
                        "kernels": "rt_jit_primary + rt_jit_trace (tolerance flavour: LDS accumulators, no staging, no accumulate kernel)",
                        "avg_launch_ms": round(m["trace_ms"] / max(m["launches"], 1), 3),
                        "primary_rays_avg_launch_ms": round(m["primary_ms"] / max(m["primary_launches"], 1), 3),
                        "algorithmic_flop_per_unit": round(fpu)},
           "run_time_kernels": bool(rf.counter("jit_active"))}
    # ---- the self-check (not timed): one frame of the same sample indices in both flavours, compared on the device
    re = make_renderer(wl, device, a, jit=True)
    re.set_option("reserve_spp", SPP)
    for r in (re, rf):
        r.refresh()
        r.set_option("sample_base", 0)
        r.sample(SPP)
        r.post_process()
        r.sync()
    te = torch.as_tensor(re.device_array(BUF_IMAGE_PIXELS), device="cuda")
    tf = torch.as_tensor(rf.device_array(BUF_IMAGE_PIXELS), device="cuda")
    d = te.double() - tf.double()
    out["l2_display_vs_exact"] = float(torch.sqrt(torch.mean(d * d)).item())
    out["max_abs_pixel_difference"] = float(d.abs().max().item())
    out["pixels_that_differ"] = float((d.abs().amax(dim=-1) > 0).double().mean().item())
    out["bar"] = 1e-3
    out["within_bar"] = bool(out["l2_display_vs_exact"] < 1e-3)
    flips = total = 0
    for k in range(16):
        for r in (re, rf):
            r.refresh()
            r.set_option("sample_base", k)
            r.sample(1)
            r.sync()
        ae = torch.as_tensor(re.device_array(BUF_IMAGE_BUFFER), device="cuda")[..., :3].double()
        af = torch.as_tensor(rf.device_array(BUF_IMAGE_BUFFER), device="cuda")[..., :3].double()
        dd = (ae - af).abs().amax(dim=-1)
        tol = 1e-3 * torch.clamp(ae.abs().amax(dim=-1), min=1.0)
        flips += int((dd > tol).sum().item())
        total += dd.numel()
    out["flip_rate"] = flips / total
    out["flip_rate_note"] = f"{flips} of {total} samples (16 single-sample frames) whose colour differs from the exact kernels' by more than 1e-3 relative"
    out["self_check"] = ("l2_display_vs_exact: both flavours rendered this frame (sample indices 0..spp-1) in this run, outside the timed region; the two image_pixels "
                         "were compared on the device through rtpbr_buffer_device_ptr; the exact kernels are bit-identical with the CPU oracle (pytest -m gpu)")
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_fast.json")))
        out["hbm"] = {"hbm_bytes_per_step": tj["fast"]["hbm_bytes_per_step"], "times_algorithmic": tj["fast"]["times_algorithmic"],
                      "kind": "a COMMITTED constant from profiles/hbm_traffic_fast.json (separate FETCH_SIZE / WRITE_SIZE passes), not a measurement of this run"}
    except Exception:
        pass
    re.close()
    rf.close()
    return out


def frame_latency(a, device, W=768, H=432, frames=200):
    """One displayed frame the way the reference produces it (/root/reference src/renderer.py:25-32 + src/main.py:62-64):
    Renderer.render() = SAMPLES_PER_FRAME (1) x pathtrace() of ONE bounce-step + post_process(), then the host reads
    image_pixels — at the reference's own window size (src/config.py:7).  Wall-clock per frame, every frame synchronised by
    its read-back; `device_ms_per_frame` is the same loop without the read-back (one sync at the end)."""
    import numpy as np
    from raytracingpbr_amd import workloads
    wl = workloads.get("src", W, H, 1)
    r = make_renderer(wl, device, a, jit=not a.no_jit)
    r.set_option("timing", 0)                 # a viewer does not ask for kernel times: no events between the kernels (they are on for sample_kernels_ms below)
    for _ in range(96):                       # the cost plan (64 steps on record) and the run-time instance exist before the timed frames
        r.render()
    from raytracingpbr_amd.renderer import BUF_IMAGE_PIXELS
    px = r.image_pixels
    r.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        r.render()
        px = r.image_pixels                   # (W, H, 3) float32 into a FRESH host array, as field.to_numpy() returns one
    dt_fresh = time.perf_counter() - t0
    pinned = r.host_array(BUF_IMAGE_PIXELS)   # the same page-locked host buffer every frame (rtpbr_host_alloc): what a viewer does
    r.read_into(BUF_IMAGE_PIXELS, pinned)
    t0 = time.perf_counter()
    for _ in range(frames):
        r.render()
        r.read_into(BUF_IMAGE_PIXELS, pinned)
    dt = time.perf_counter() - t0
    # ... and handed over WITHOUT stalling the device (round 6: rtpbr_read_buffer_async): frame k's copy into one of two
    # page-locked buffers is in flight while frame k+1 is sampled; the host takes delivery of frame k-1 each iteration
    pinned2 = [pinned, r.host_array(BUF_IMAGE_PIXELS)]
    prev = r.read_async(BUF_IMAGE_PIXELS, pinned2[1])
    r.read_wait(prev)
    t0 = time.perf_counter()
    prev = None
    for k in range(frames):
        r.sample(1)                           # render() = pathtrace() ...
        if prev is not None:
            r.read_wait(prev)                 # (frame k-1 is in host memory now: delivered while frame k is being sampled)
        r.post_process()                      # ... + post_process(): image_pixels is free again, no device-side wait is enqueued
        prev = r.read_async(BUF_IMAGE_PIXELS, pinned2[k & 1])
    r.read_wait(prev)
    dt_pipe = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(frames):
        r.render()
    r.sync()
    dt_dev = time.perf_counter() - t0
    # device time of the sample kernels (gen + march + shade of one bounce-step), HIP events, mean over 64 frames (a single launch
    # varies by +-10 %: the launch is as long as its slowest wave)
    r.set_option("timing", 1)
    tr = 0.0
    for _ in range(64):
        r.render()
        tr += r.last_sample_ms()[0]
    tr /= 64.0
    split = bool(r.counter("jit_active"))
    r.close()
    return {"workload": f"src/ pipeline {W}x{H}: Renderer.render() = sample(1) + post_process(), then image_pixels read to the host — "
                        f"one displayed frame of the reference (src/renderer.py:25-32, src/main.py:62-64), {frames} frames",
            "ms_per_frame": round(dt / frames * 1e3, 4), "frames_per_s": round(frames / dt, 1),
            "ms_per_frame_fresh_host_array": round(dt_fresh / frames * 1e3, 4),
            "ms_per_frame_pipelined": round(dt_pipe / frames * 1e3, 4), "frames_per_s_pipelined": round(frames / dt_pipe, 1),
            "pipelined": "rtpbr_read_buffer_async into two page-locked buffers: frame k's copy overlaps frame k+1's sample kernels, every frame is delivered to the host (one frame later); "
                         "a consumer on the GPU takes rtpbr_buffer_device_ptr and pays device_ms_per_frame",
            "events": "option timing = 0 in the frame loops (no HIP events between the kernels), 1 for sample_kernels_ms",
            "host_buffer": "ms_per_frame reads image_pixels into one page-locked host buffer (rtpbr_host_alloc) every frame; ms_per_frame_fresh_host_array allocates a numpy array per frame (round 4's figure)",
            "device_ms_per_frame": round(dt_dev / frames * 1e3, 4), "sample_kernels_ms": round(tr, 4),
            "readback_bytes_per_frame": int(np.asarray(px).nbytes), "run_time_kernels": split,
            "note": "the bounce-step of a one-step launch runs as the wavefront split (gen / march / shade kernels, rt_split.hpp)"}


def main():
    a = parse()
    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    dist = None
    if a.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.transport == "torch":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)       # control plane only
    # The catalog of code objects compiled at BUILD time for the BASELINE scenes (raytracingpbr_amd/prebuild.py) would serve this
    # run without any compilation.  The main flow keeps measuring the real first use (hipcc --genco into a fresh cache), so it
    # looks past the catalog — unless there is no compiler on this machine, in which case the catalog is exactly what a target
    # without hipcc runs on; `jit.no_compiler_value` below is that configuration, measured in every default run.
    import shutil as _sh
    # (the compiler rt_jit.hip would fork: $HIPCC if set — taken literally —, else /opt/rocm/bin/hipcc, else hipcc on PATH)
    _cc = os.environ.get("HIPCC") or ("/opt/rocm/bin/hipcc" if os.access("/opt/rocm/bin/hipcc", os.X_OK) else "hipcc")
    have_compiler = bool(_sh.which(_cc))
    if have_compiler and "RTPBR_JIT_CATALOG" not in os.environ and not a.keep_jit_cache:
        os.environ["RTPBR_JIT_CATALOG"] = "/nonexistent-rtpbr-catalog"
        catalog_hidden = True
    else:
        catalog_hidden = False
    if not a.keep_jit_cache and "RTPBR_JIT_CACHE" not in os.environ:
        # a fresh private code-object cache: the first use below then measures the real compile; ranks share it
        d = [tempfile.mkdtemp(prefix="rtpbr-bench-jit-")] if rank == 0 else [None]
        if dist is not None:
            dist.broadcast_object_list(d, src=0)
        os.environ["RTPBR_JIT_CACHE"] = d[0]
        if rank == 0:
            import atexit
            import shutil
            atexit.register(shutil.rmtree, d[0], True)      # the code objects of this run only

    from raytracingpbr_amd import workloads
    from raytracingpbr_amd.distributed import TileGather
    from raytracingpbr_amd.tiles import default_tile

    wl = workloads.get(a.workload, a.width, a.height, a.spp, a.bounces)
    W, H, SPP = wl.cfg.width, wl.cfg.height, wl.spp
    dev = torch.device("cuda", local_rank)
    r = make_renderer(wl, local_rank, a, jit=not a.no_jit)

    # ---- N > 1: tiles + the one gather
    gather, multi = None, {}
    if world > 1 or a.collective_at_1:
        tw, th = default_tile(W, H, max(world, 2))
        if a.transport == "rccl":
            r.set_tiles(tw, th, rank, world)
            uid = [r.rccl_unique_id() if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(uid, src=0)             # 128 bytes over the gloo control plane
            t0 = time.perf_counter()
            r.rccl_init(uid[0], rank, world)                       # ncclCommInitRank, ROCm's librccl behind the C ABI
            r.sync()
            multi["comm_init_s"] = round(time.perf_counter() - t0, 3)
            n, rr, ver = r.rccl_info()
            multi.update({"rccl_nranks": n, "rccl_rank": rr, "rccl_version": ver})
            gather = r.gather_tiles
        else:
            tg = TileGather(r, rank, world, tile=(tw, th), device=dev if a.transport == "torch" else None)
            gather = tg.gather
        multi["transport"] = {"rccl": "rtpbr_gather_tiles (ncclGather of ROCm's librccl through the C ABI)",
                              "torch": "torch.distributed nccl gather", "host": "gloo gather through host memory (functional test)"}[a.transport]
        multi["tile"] = [tw, th]
        multi["bytes_gathered_per_rank"] = r.packed_bytes()

    # ---- device buffers and kernels exist before the timed region, whatever --warmup is; the first use is timed
    if wl.family != "src":
        r.set_option("reserve_spp", SPP)
    # (a whole step, so that every launch of the dominant kernel in this process is the same size and the rocprofv3 --stats
    # average of the same command agrees with the HIP-event average below)
    t0 = time.perf_counter()
    r.sample(SPP)
    r.sync()
    first_use_s = time.perf_counter() - t0

    def fence():
        r.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gather_ms = []

    def timed_gather():
        r.sync()                                   # (render done: lets the gather be timed by itself; microseconds)
        t = time.perf_counter()
        gather()
        r.sync()
        gather_ms.append((time.perf_counter() - t) * 1e3)

    m = measure(wl, r, a.steps, a.warmup, fence, timed_gather if gather is not None else None)
    dt = m["dt"]
    render_ms = m["trace_ms"] + m["primary_ms"]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if a.transport == "torch" else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "kernel_ms_per_step": round(render_ms / a.steps, 3),
                                          "gather_ms_per_step": round(sum(gather_ms[-a.steps:]) / a.steps, 3), "wall_s": round(m["dt"], 4)})
        multi["per_rank"] = per_rank
        if not a.no_check_gather and rank == 0:
            import numpy as np
            # the same sequence of steps, untiled (sample indices advance from step to step; the src/ form also keeps ray state)
            full = make_renderer(wl, local_rank, a, jit=not a.no_jit)
            full.sample(SPP)
            for _ in range(a.warmup + a.steps):
                full.refresh()
                full.sample(SPP)
            multi["gathered_equals_untiled"] = bool(np.array_equal(np.ascontiguousarray(r.image_buffer).view(np.uint32),
                                                                   np.ascontiguousarray(full.image_buffer).view(np.uint32)))
            full.close()

    if rank == 0:
        c, fpu, kernel_s, tflops = roofline_of(wl, r, m, a.steps, W * H)
        launches = max(m["launches"], 1)
        total_units = W * H * SPP * a.steps
        value = total_units / dt / 1e6
        # ---- per-launch figures of the dominant kernel on this rank
        avg_launch_s = m["trace_ms"] / 1e3 / launches
        k_per_launch = SPP * a.steps / launches
        units_per_launch = c.samples * a.steps / launches
        sky_frac = c.sky_lookups / max(c.samples, 1)
        if wl.family == "src":      # SURVEY 8(d), src form, per bounce-step: T6 R+W once per launch, 32 B per deposit, 12 B per sky lookup
            bytes_per_unit = 80.0 / k_per_launch + 32.0 * c.deposits / max(c.samples, 1) + 12.0 * sky_frac
        else:                       # examples form, K spp per launch: T7 R+W per pixel and launch + one texel per escaping path
            bytes_per_unit = 32.0 / k_per_launch + 12.0 * sky_frac
        alg_bytes = bytes_per_unit * units_per_launch
        achieved_gbs = alg_bytes / max(avg_launch_s, 1e-9) / 1e9
        # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside
        # this process); only reported when it was measured on this very workload
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            w = tj["workload"]
            if a.workload == "c2" and (w["width"], w["height"], w["spp"], w["bounces"]) == (W, H, SPP, wl.cfg.max_raytrace) and world == 1 \
                    and abs(w["spp_per_launch"] - k_per_launch) < 1e-6:
                traffic, traffic_src = tj["hbm_bytes_per_launch"], tj["source"]
        except Exception:
            pass
        if wl.family == "src":
            res = 32        # option "residency" (library default 32): bounce-steps between two T6 round trips of a pixel (multi-pass walk)
            for kv in a.opt:
                if kv.startswith("residency="):
                    res = int(kv.split("=")[1])
            impl_bytes = int(80 * W * H * max(1, SPP // res) + 32 * c.deposits + 16 * c.sky_lookups)
        else:
            split_on = m["primary_launches"] > 0
            impl_bytes = int((24 + (10 if split_on else 0)) * c.samples + 32 * W * H * (launches / a.steps) + 16 * c.sky_lookups)
        jit_on = bool(r.counter("jit_active"))
        if wl.family == "src":
            kname, pname = ("rt_jit_persistent_pool" if jit_on else "persistent_pool"), None
        else:
            kname, pname = ("rt_jit_trace", "rt_jit_primary") if jit_on else ("trace_paths_pool", "primary_rays")
            if m["primary_launches"] == 0:
                pname = None
        what = "bounce-steps per pixel" if wl.family == "src" else "spp"
        out = {
            "metric": f"{wl.unit.replace('/s', '')}/sec (pixels x {what} / s), {wl.short} {W}x{H}",
            "value": round(value, 2), "unit": wl.unit, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl.title}; one step = refresh + {SPP} {what} through the sample kernels"
                                   + ("" if wl.family == "src" else " + ordered accumulation")
                                   + (f" + 1 RCCL gather of {world} tile sets" if world > 1 else ""),
                       "parallelism": f"tiles{world}" if world > 1 else "single",
                       "kernels": "run-time compiled for this scene (object table, render configuration and camera frame baked)" if jit_on
                                  else "ahead-of-time instances",
                       "raycasts_per_unit": round(c.raycasts / max(c.samples, 1), 3),
                       "march_steps_per_raycast": round(c.march_steps / max(c.raycasts, 1), 3)},
            # the binding roofline: FP32 vector issue.  achieved = algorithmic FLOPs per launch / HIP-event time of the
            # kernels that do them; traffic = HBM bytes per launch of the dominant kernel
            "roofline": {"bound": "valu", "achieved": round(tflops, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tflops / VALU_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "avg_launch_ms": round(avg_launch_s * 1e3, 3), "launches_timed": m["launches"],
                         "kernels": f"{pname} + {kname}" if pname else kname,
                         "primary_rays_avg_launch_ms": round(m["primary_ms"] / max(m["primary_launches"], 1), 3),
                         "primary_rays_launches_timed": m["primary_launches"],
                         "algorithmic_flop_per_unit": round(fpu),
                         "ubench_ceiling": VALU_UBENCH_TFLOPS, "frac_of_ubench_ceiling": round(tflops / VALU_UBENCH_TFLOPS, 4),
                         "note": "branchy scalar FP32 on the vector ALU; peak = 157.3 TFLOP/s FP32 vector (spec); ubench_ceiling = what a pure "
                                 "v_fma_f32 stream sustains on this chip (tools/ubench/valu_rate.hip), reported beside it, never instead of it"
                                 + (" (the neural SDF's f32 MFMA has the same peak rate and does not overlap with VALU work)" if wl.family == "bunny" else "")},
            # the view the metric's name asks for: HBM GB/s of the dominant kernel vs 8 TB/s
            "hbm": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "achieved_algorithmic": round(achieved_gbs, 4), "frac_algorithmic": round(achieved_gbs / HBM_PEAK_GBS, 8),
                    "algorithmic_bytes_per_launch": round(alg_bytes),
                    "achieved_counters": round(traffic / avg_launch_s / 1e9, 2) if traffic else None,
                    "frac_counters": round(traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                    "traffic": traffic, "kernel": kname,
                    "traffic_kind": "a COMMITTED constant from the PMC passes of profiles/hbm_traffic.json (rocprofv3 cannot run inside this process), not a measurement of this run" if traffic else None,
                    # what THIS run moved by construction, from its own counts (per step, all kernels): the complete-path form stages
                    # 12 B per sample (pool kernel W, accumulate R) and 5 B of primary record per sample (primary kernel W, pool
                    # kernel R) and read-modify-writes T7 once per pixel and launch; the src/ form moves T6 (40 B R + W) per pixel
                    # and residency, T7 (16 B R + W) per deposit and 16 B per sky lookup
                    "implementation_kind": "a MODEL from this run's own counts (records x bytes), not a counter measurement; the src/ form's figure assumes "
                                           "one T6 round trip per pixel and residency, which over-counts waves that keep their pixels resident",
                    "implementation_bytes_per_step": impl_bytes, "implementation_gbs": round(impl_bytes / max(dt / a.steps, 1e-9) / 1e9, 2),
                    "implementation_frac": round(impl_bytes / max(dt / a.steps, 1e-9) / 1e9 / HBM_PEAK_GBS, 5)},
        }
        if wl.family != "src":
            split = m["primary_launches"] > 0
            out["staging"] = {"bytes_per_step": int((12 + (5 if split else 0)) * c.samples),
                              "note": "transient device memory of one step: one 12-byte colour record per pixel-sample (+ one 5-byte primary record), "
                                      "reduced in sample order into image_buffer; reserved before the timed region (option reserve_spp)"}
        if multi:
            out["multi_gpu"] = multi
        r.close()
        if world == 1:
            # ---- what the run-time compiled kernels cost and buy (same workload, one step each)
            jit = {"first_use_s": round(first_use_s, 3), "first_use_minus_one_step_s": round(first_use_s - dt / a.steps, 3),
                   "first_use_note": "first rtpbr_sample() of a whole step: hipcc --genco of the scene's kernels into a fresh cache + module load "
                                     "+ first touch of the staging + the step itself" if not a.keep_jit_cache else "first rtpbr_sample() with the user's cache"}
            if not a.no_jit and not a.no_configs:
                for key, (j, b) in (("camera_free_value", (True, 1)), ("unbaked_value", (True, False)), ("aot_value", (False, False))):
                    r2 = make_renderer(wl, local_rank, a, jit=j, bake=b)
                    if wl.family != "src":
                        r2.set_option("reserve_spp", SPP)
                    r2.sample(SPP)
                    r2.sync()
                    m2 = measure(wl, r2, 1, 0)
                    jit[key] = round(W * H * SPP / m2["dt"] / 1e6, 1)
                    r2.close()
            if not a.no_jit and not a.no_configs and wl.family != "src":
                # the same step on a target WITHOUT a compiler: HIPCC points nowhere, the cache is empty, the code object comes
                # from the catalog built with the library (strict: jit = 2 — no silent fall-back to the ahead-of-time kernels)
                saved = {k: os.environ.get(k) for k in ("HIPCC", "RTPBR_JIT_CACHE", "RTPBR_JIT_CATALOG")}
                empty = tempfile.mkdtemp(prefix="rtpbr-bench-nocc-")
                os.environ["HIPCC"] = "/nonexistent/hipcc"
                os.environ["RTPBR_JIT_CACHE"] = empty
                if catalog_hidden:
                    del os.environ["RTPBR_JIT_CATALOG"]
                try:
                    r2 = make_renderer(wl, local_rank, a, jit=True, bake=True)
                    r2.set_option("jit", 2)
                    r2.set_option("reserve_spp", SPP)
                    t0 = time.perf_counter()
                    r2.sample(SPP)
                    r2.sync()
                    jit["no_compiler_first_use_s"] = round(time.perf_counter() - t0, 3)
                    m2 = measure(wl, r2, a.steps, 0)
                    jit["no_compiler_value"] = round(W * H * SPP * a.steps / m2["dt"] / 1e6, 1)
                    jit["no_compiler_run_time_kernels"] = bool(r2.counter("jit_active"))
                    r2.close()
                except Exception as e:      # noqa: BLE001
                    jit["no_compiler_error"] = str(e)[:300]
                finally:
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                    import shutil
                    shutil.rmtree(empty, True)
                jit["no_compiler_note"] = ("HIPCC=/nonexistent, empty code-object cache: the instance (scene, configuration, camera baked) comes from the catalog "
                                           "compiled at build time (raytracingpbr_amd/data/jit, python -m raytracingpbr_amd.prebuild); option jit = 2, so a missing "
                                           "instance would be an error, not a fall-back")
            jit["compiler_on_this_machine"] = have_compiler
            jit["note"] = ("value (the headline) = run-time instance with scene, configuration and camera frame baked (jit_bake 2: a moved camera "
                           "recompiles); camera_free_value = scene and configuration baked, camera a launch argument (jit_bake 1); unbaked_value = "
                           "run-time instance, nothing baked; aot_value = the ahead-of-time library alone (no hipcc on the target)")
            out["jit"] = jit
            if a.workload == "c2" and not a.no_configs and not a.no_jit:
                out["tolerance"] = tolerance_block(wl, a, local_rank, torch)
            if a.workload == "c2" and not a.no_configs:
                out["configs"] = {n: side_config(n, a, local_rank) for n in ("c1", "c3", "c3_valu", "c4", "c5", "src", "src_768", "src_4k", "src_1step",
                                                                                 "c3_fast", "c4_fast", "src_fast")}
            if a.workload == "c2" and not a.no_configs:
                out["frame"] = frame_latency(a, local_rank)
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(wl, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
