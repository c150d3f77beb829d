#!/bin/bash
# Where do the wave cycles go: WAIT_ANY (parked on s_waitcnt/barrier) / WAIT_INST_ANY (issue stall) / ACTIVE_INST_ANY
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  rm -rf $OUT/pmcw_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmcw_$i -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $BENCH_OPTS > /dev/null 2> $OUT/pmcw_$i.err
  python - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/pmcw_$i/**/*counter_collection.csv", recursive=True)
if not fs: print("no output for set $i; see pmcw_$i.err"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if any(t in k for t in ("trace_paths", "primary", "rt_jit", "persistent")):
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    print(k[:40], {c: "%.4g" % (x / n[(k, c)]) for c, x in v.items()})
PY
done
