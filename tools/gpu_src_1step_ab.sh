#!/bin/bash
# round 6: A/B of option sets on one-step launches (and optionally fused ones):  bash tools/gpu_src_1step_ab.sh <tag> "<K=V ...>" "<K=V ...>" ...   ("-" = defaults)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r06_ab}; shift
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/${TAG}_build.log 2>&1
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "$TESTS" > $OUT/${TAG}_pytest.log 2>&1; tail -3 $OUT/${TAG}_pytest.log; fi
: > $OUT/${TAG}.jsonl
for rep in 1 2; do
for o in "$@"; do
  [ "$o" = "-" ] && oo="" || oo="$o"
  for size in ${SIZES1:-"768 432" "1920 1080"}; do
    timeout 300 python tools/gpu_src_1step.py $size 256 $oo >> $OUT/${TAG}.jsonl 2>> $OUT/${TAG}.err
  done
  if [ $rep = 1 ] && [ -n "$FUSED" ]; then
  for size in $FUSED; do
    timeout 300 python tools/gpu_src_conv.py ${size/x/ } 7 $oo >> $OUT/${TAG}.jsonl 2>> $OUT/${TAG}.err
  done
  fi
done
done
cat $OUT/${TAG}.jsonl
