#!/bin/bash
# round 6: the object-parallel evaluation of sparse waves (option src_op) — parity subset, then A/B of one-step and fused launches
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r06_op}
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/${TAG}_build.log 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "persistent or adaptive or fuzz_random or backend" > $OUT/${TAG}_pytest.log 2>&1
  tail -5 $OUT/${TAG}_pytest.log
fi
: > $OUT/${TAG}_ab.jsonl
for size in "768 432" "1920 1080" "640 360"; do
  for o in "src_op=0" "src_op=1" "src_op=0" "src_op=1"; do
    timeout 300 python tools/gpu_src_1step.py $size 256 $o >> $OUT/${TAG}_ab.jsonl 2>> $OUT/${TAG}_ab.err
  done
done
for size in "768 432" "1024 576" "1920 1080"; do
  for o in "src_op=0" "src_op=1"; do
    timeout 300 python tools/gpu_src_conv.py $size 8 $o >> $OUT/${TAG}_ab.jsonl 2>> $OUT/${TAG}_ab.err
  done
done
cat $OUT/${TAG}_ab.jsonl
