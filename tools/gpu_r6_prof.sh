#!/bin/bash
# round 6: instrumented one-step launches (per-wave records of the split march) with and without the object-parallel evaluation
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r06_prof}
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/${TAG}_build.log 2>&1
export RTPBR_JIT_EXTRA_FLAGS=-DRT_DEBUG_PHASE
: > $OUT/${TAG}.jsonl
for size in "768 432" "1920 1080"; do
  for o in ${OPTS:-"src_op=0" "src_op=1"}; do
    echo "{\"size\": \"$size\", \"opts\": \"$o\"}" >> $OUT/${TAG}.jsonl
    timeout 300 python tools/gpu_split_prof.py $size $o >> $OUT/${TAG}.jsonl 2>> $OUT/${TAG}.err
  done
done
unset RTPBR_JIT_EXTRA_FLAGS
if [ -n "$EXTRA" ]; then bash -c "$EXTRA" >> $OUT/${TAG}_extra.log 2>&1; fi
tail -c 30000 $OUT/${TAG}.jsonl
