#!/bin/bash
# HBM traffic of the bench's kernels: FETCH_SIZE and WRITE_SIZE in separate passes (KiB per dispatch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/traf_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/traf_$c -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $BENCH_OPTS > /dev/null 2> $OUT/traf_$c.err
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/traf_$c/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "trace_paths" in k or "primary" in k or "accumulate" in k:
        acc[k] += float(r["Counter_Value"]); n[k] += 1
for k in acc: print("$c", k[:48], "%.3f GB per dispatch" % (acc[k] / n[k] * 1024 / 1e9))
PY
done
