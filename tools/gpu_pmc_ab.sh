#!/bin/bash
# SQ counter pass of the bench with different option sets: bash tools/gpu_pmc_ab.sh "lazy_sqrt=0" "lazy_sqrt=1"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for o in "$@"; do
  tag=$(echo $o | tr '= ' '__')
  rm -rf $OUT/pmcab_$tag
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmcab_$tag -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --opt $o > /dev/null 2> $OUT/pmcab_$tag.err
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmcab_$tag/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "trace_paths" in k or "primary" in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    print("$o", k[:50], {c: "%.4g" % (x / n[(k, c)]) for c, x in v.items()})
PY
done
