"""Summarise the rocprofv3 CSV outputs of tools/gpu_round.sh into profiles/<tag>_*.{txt,json}.

Reads gpurun_out/prof_<tag> (kernel trace + stats) and gpurun_out/pmc_*_<tag> (counter
collection).  Counter units: FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB
(x1024 -> bytes); on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide coalesced
streams (MI355X_MICROARCH.md §HBM) — reported raw and with the x2 read correction."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)


def find(d, pat):
    r = glob.glob(os.path.join(OUT, d, "**", pat), recursive=True)
    return r[0] if r else None


summary = {"tag": tag}
lines = []
# ---- kernel stats
f = find(f"prof_{tag}", "*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    lines.append("== rocprofv3 --kernel-trace --stats : kernel_stats")
    lines.append(f"{'kernel':70s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>10s} {'pct':>7s}")
    ks = []
    for r in rows:
        name = r.get("Name", r.get("KernelName", "?"))
        calls = int(r.get("Calls", 0))
        tot = float(r.get("TotalDurationNs", 0)) / 1e6
        avg = float(r.get("AverageNs", 0)) / 1e6
        pct = float(r.get("Percentage", 0))
        lines.append(f"{name[:70]:70s} {calls:6d} {tot:12.3f} {avg:10.4f} {pct:7.2f}")
        ks.append({"kernel": name, "calls": calls, "total_ms": tot, "avg_ms": avg, "pct": pct})
    summary["kernel_stats"] = ks
# ---- per-dispatch resources
f = find(f"prof_{tag}", "*kernel_trace.csv")
if f:
    seen = {}
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "?")
        if n not in seen:
            seen[n] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                             "Workgroup_Size", "Grid_Size")}
    lines.append("\n== per-kernel resources (first dispatch)")
    for n, v in seen.items():
        lines.append(f"{n[:70]:70s} {v}")
    summary["resources"] = seen
# ---- counters
ctr = {}
for d in (f"pmc_fetch_{tag}", f"pmc_write_{tag}", f"pmc_sq_{tag}"):
    f = find(d, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")
        c = r.get("Counter_Name")
        acc[k][c] += float(r.get("Counter_Value", 0))
        cnt[k][c] += 1
    for k in acc:
        for c in acc[k]:
            ctr.setdefault(k, {})[c] = {"sum": acc[k][c], "dispatches": cnt[k][c], "per_dispatch": acc[k][c] / max(cnt[k][c], 1)}
if ctr:
    lines.append("\n== PMC counters (separate passes), per dispatch")
    for k, v in ctr.items():
        for c, x in v.items():
            lines.append(f"{k[:60]:60s} {c:24s} per_dispatch={x['per_dispatch']:.4g} dispatches={x['dispatches']}")
    summary["counters"] = ctr
    for k, v in ctr.items():
        if ("trace_paths" in k or "rt_jit_trace" in k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fe, wr = v["FETCH_SIZE"]["per_dispatch"] * 1024, v["WRITE_SIZE"]["per_dispatch"] * 1024
            summary["trace_paths_hbm_bytes_per_launch"] = {"fetch_raw": fe, "fetch_x2_corrected": 2 * fe, "write": wr,
                                                           "total_corrected": 2 * fe + wr}
            lines.append(f"\ntrace_paths HBM bytes per launch: FETCH {fe:.4g} (x2 gfx950 correction: {2 * fe:.4g})  WRITE {wr:.4g}")
    for k, v in ctr.items():
        if ("primary_rays" in k or "rt_jit_primary" in k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fe, wr = v["FETCH_SIZE"]["per_dispatch"] * 1024, v["WRITE_SIZE"]["per_dispatch"] * 1024
            summary["primary_rays_hbm_bytes_per_launch"] = {"fetch_raw": fe, "fetch_x2_corrected": 2 * fe, "write": wr,
                                                            "total_corrected": 2 * fe + wr}
            lines.append(f"primary_rays HBM bytes per launch: FETCH {fe:.4g} (x2: {2 * fe:.4g})  WRITE {wr:.4g}")
for b in (f"bench_{tag}.json", f"prof_bench_{tag}.json"):
    p = os.path.join(OUT, b)
    if os.path.exists(p) and os.path.getsize(p):
        try:
            summary[b] = json.loads(open(p).read().strip().split("\n")[-1])
        except Exception:
            pass
for b in (f"host_{tag}.txt", f"pytest_gpu_{tag}.log", f"smoke_{tag}.log"):
    p = os.path.join(OUT, b)
    if os.path.exists(p):
        lines.append(f"\n== {b}\n" + open(p).read().strip())
# ---- profiles/hbm_traffic.json: what bench.py reports as roofline.traffic (dominant kernel, per launch)
bj = summary.get(f"prof_bench_{tag}.json")
tp = summary.get("trace_paths_hbm_bytes_per_launch")
if bj and tp and "--keep-traffic" not in sys.argv:
    import re
    m = re.search(r"(\d+)x(\d+), (\d+) spp, (\d+) bounces", bj["config"]["workload"])
    launches = max(bj["roofline"]["launches_timed"], 1)
    traffic = {
        "source": f"profiles/{tag}_rocprof_summary.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, kernel-trace only)",
        "command": "python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
        "workload": {"width": int(m.group(1)), "height": int(m.group(2)), "spp": int(m.group(3)), "bounces": int(m.group(4)),
                     "spp_per_launch": int(m.group(3)) * bj["steps"] / launches},
        "kernel": [k for k in ctr if "trace_paths" in k or "rt_jit_trace" in k][0],
        "fetch_bytes_raw": tp["fetch_raw"], "fetch_bytes_x2_gfx950": tp["fetch_x2_corrected"], "write_bytes": tp["write"],
        "hbm_bytes_per_launch": tp["total_corrected"],
        "primary_rays_hbm_bytes_per_launch": summary.get("primary_rays_hbm_bytes_per_launch"),
        "units": "FETCH_SIZE/WRITE_SIZE are KiB (x1024); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B "
                 "request on wide streams)",
    }
    json.dump(traffic, open(os.path.join(PROF, "hbm_traffic.json"), "w"), indent=1)
    lines.append("\nwrote profiles/hbm_traffic.json")
open(os.path.join(PROF, f"{tag}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(summary, open(os.path.join(PROF, f"{tag}_rocprof_summary.json"), "w"), indent=1)
print("\n".join(lines))
