#!/bin/bash
# round 6: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the headline step's kernels for option sets:
#   bash tools/gpu_traffic_ab.sh <tag> "<bench options A>" "<bench options B>" ...     ("-" = none)      COUNTERS="WRITE_SIZE FETCH_SIZE"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r06_traffic}; shift
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/${TAG}_build.log 2>&1
export TMPDIR=/tmp
export RTPBR_JIT_CACHE=/tmp/rtpbr-traffic-cache      # (compile once: every pass finds the code objects)
cd /tmp
: > $OUT/${TAG}.txt
for o in "$@"; do
  [ "$o" = "-" ] && oo="" || oo="$o"
  for c in ${COUNTERS:-WRITE_SIZE FETCH_SIZE}; do
    rm -rf $OUT/${TAG}_pmc
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc -o pmc -- env RTPBR_JIT_EXTRA_FLAGS="$FLAGS" python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --keep-jit-cache ${WORKLOAD:+--workload $WORKLOAD} $oo > /dev/null 2> $OUT/${TAG}_pmc.err
    python - <<PY >> $OUT/${TAG}.txt
import csv, glob, collections
fs = glob.glob("$OUT/${TAG}_pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0]
    if r["Counter_Name"] == "$c" and ("trace" in k or "primary" in k or "accumulate" in k or "persistent" in k or "src_" in k or "chain" in k):
        acc[k].append(float(r["Counter_Value"]) * 1024)
for k, v in acc.items():
    print("[$o] $c %-28s %d dispatches, last %.3f GB, max %.3f GB" % (k[:28], len(v), v[-1] / 1e9, max(v) / 1e9))
PY
  done
done
rm -rf $OUT/${TAG}_pmc
cat $OUT/${TAG}.txt
