#!/bin/bash
# Round-2 GPU-box visit: everything gpu_round.sh does, plus the K3 per-pixel comparison, and a rocprofv3
# kernel-trace summary of the OTHER BASELINE configs (bunny C3, Tokyo C4, 8K C5 share, src/ form).
# Usage (repo root, through gpurun):  bash tools/gpu_round2.sh <tag> [skip-tests]
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [ "$2" != "skip-tests" ]; then
  bash tools/gpu_round.sh $TAG > $OUT/round_$TAG.log 2>&1
  tail -60 $OUT/round_$TAG.log
fi
python tools/gpu_k3.py 16384 2>&1 | tail -5 | tee $OUT/k3_$TAG.log
export TMPDIR=/tmp
cd /tmp
export C3_SPP=${C3_SPP:-1024} C5_SPP=${C5_SPP:-512}
rm -f $OUT/configs.json $OUT/configs_$TAG.log
# one rocprofv3 run per config: the run-time compiled kernels of every scene carry the same names
for cfg in $(python $R/tools/gpu_configs.py --list); do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg_$TAG/$cfg -o trace -- \
     python $R/tools/gpu_configs.py $cfg >> $OUT/configs_$TAG.log 2> $OUT/prof_cfg_$TAG.$cfg.err
done
cd $R
cp $OUT/configs.json $OUT/configs_$TAG.json 2>/dev/null
python tools/prof_configs_summary.py $TAG 2>&1 | tail -30
