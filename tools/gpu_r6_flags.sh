#!/bin/bash
# round 6: A/B of compile-time variants of the run-time instance (RTPBR_JIT_EXTRA_FLAGS) on one-step and fused src/ launches
#   bash tools/gpu_r6_flags.sh <tag> "<flags A>" "<flags B>" ...      ("-" = no extra flags)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r06_flags}; shift
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/${TAG}_build.log 2>&1
: > $OUT/${TAG}.jsonl
for rep in 1 2; do
for f in "$@"; do
  [ "$f" = "-" ] && export RTPBR_JIT_EXTRA_FLAGS="" || export RTPBR_JIT_EXTRA_FLAGS="$f"
  echo "{\"flags\": \"$f\"}" >> $OUT/${TAG}.jsonl
  for size in ${SIZES1:-"768 432" "1920 1080"}; do
    timeout 300 python tools/gpu_src_1step.py $size 256 $OPTS >> $OUT/${TAG}.jsonl 2>> $OUT/${TAG}.err
  done
  if [ $rep = 1 ]; then
  for size in ${SIZESF:-"768 432" "1920 1080"}; do
    timeout 300 python tools/gpu_src_conv.py $size 7 $OPTS >> $OUT/${TAG}.jsonl 2>> $OUT/${TAG}.err
  done
  fi
done
done
cat $OUT/${TAG}.jsonl
