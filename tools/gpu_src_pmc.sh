#!/bin/bash
# src/-form kernels: kernel time + SQ counter passes at several frame sizes and both schedulers.
#   bash tools/gpu_src_pmc.sh <tag> [KEY=VALUE options for tools/gpu_src_prof.py]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-src}; shift
export TMPDIR=/tmp
cd /tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH")
: > $OUT/${TAG}_runs.jsonl
for size in "768 432" "1920 1080" "3840 2160"; do
 for sched in ${SCHEDS:-0 1}; do
  python $R/tools/gpu_src_prof.py $size $sched 256 "$@" >> $OUT/${TAG}_runs.jsonl 2>> $OUT/${TAG}.err
  i=0
  for set in "${SETS[@]}"; do
   i=$((i+1)); d=$OUT/${TAG}_pmc_${size// /x}_s${sched}_$i; rm -rf $d
   rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o pmc -- python $R/tools/gpu_src_prof.py $size $sched 256 "$@" > /dev/null 2>> $OUT/${TAG}.err
  done
 done
done
python - <<PY
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("$OUT/${TAG}_pmc_*")):
    key = d.split("_pmc_")[1].rsplit("_", 1)[0]
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "persistent" in r["Kernel_Name"]]
    if not rows: continue
    last = max(int(r["Dispatch_Id"]) for r in rows)      # the timed 256-step launch is the last dispatch
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            e = out.setdefault(key, {"kernel": r["Kernel_Name"].split("(")[0], "vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                                     "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size"), "grid": r.get("Grid_Size")})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
runs = [json.loads(l) for l in open("$OUT/${TAG}_runs.jsonl") if l.strip().startswith("{")]
json.dump({"runs": runs, "pmc": out}, open("$OUT/${TAG}_summary.json", "w"), indent=1)
for r in runs: print(r)
for k, v in out.items(): print(k, v)
PY
