#!/bin/bash
# SQ counters of the wavefront split's kernels (one bounce-step per launch): lane utilisation and issue rate of rt_jit_src_march.
# Usage (repo root, through gpurun):  bash tools/gpu_src_1step_pmc.sh <tag>   ->  gpurun_out/<tag>_1step_pmc.json
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for size in "768 432" "1920 1080"; do
  d=$OUT/${TAG}_1step_pmc_${size// /x}; rm -rf $d
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $d -o pmc -- python $R/tools/gpu_src_1step.py $size 64 > $OUT/${TAG}_1step_pmc_${size// /x}.log 2>&1
done
python - <<PY
import csv, glob, json, collections
out = {}
for d in sorted(glob.glob("$OUT/${TAG}_1step_pmc_*x*")):
    if not d.rsplit("_", 1)[1][0].isdigit(): continue
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    rows = list(csv.DictReader(open(fs[0])))
    last = {}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("rt_jit_src"): continue
        last.setdefault(k, []).append(int(r["Dispatch_Id"]))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if k in last and int(r["Dispatch_Id"]) >= sorted(set(last[k]))[-16]:      # the last 16 launches of each kernel
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    res = {}
    for k, c in acc.items():
        nl = 16.0
        res[k] = {kk: v / nl for kk, v in c.items()}
        if c.get("SQ_INSTS_VALU"):
            res[k]["lanes_per_valu_instruction"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] / 4.0, 2) if c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] > 64 else round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"], 2)
    out[d.rsplit("_", 1)[1]] = res
json.dump(out, open("$OUT/${TAG}_1step_pmc.json", "w"), indent=1)
for k, v in out.items():
    for kk, c in v.items(): print(k, kk, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in c.items()})
PY
