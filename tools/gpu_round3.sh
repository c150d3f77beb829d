#!/bin/bash
# Round-5 profile visit beyond gpu_round.sh: one rocprofv3 kernel trace per BASELINE config and src/ figure (tools/gpu_configs.py),
# the SQ counter passes of the src/ pool kernel, and the HBM counter passes of the tolerance flavour's headline step.
# Usage (repo root, through gpurun):  bash tools/gpu_round3.sh <tag>
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export C3_SPP=${C3_SPP:-1024} C5_SPP=${C5_SPP:-512}
rm -f $OUT/configs.json $OUT/configs_${TAG}c.log
for cfg in $(python $R/tools/gpu_configs.py --list); do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg_${TAG}c/$cfg -o trace -- \
     python $R/tools/gpu_configs.py $cfg >> $OUT/configs_${TAG}c.log 2> $OUT/prof_cfg_${TAG}c.$cfg.err
done
cp $OUT/configs.json $OUT/configs_${TAG}c.json 2>/dev/null
# tolerance flavour, headline step: FETCH / WRITE in separate passes (the guide's recipe)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_fast_${c}_$TAG -o pmc -- python $R/tools/gpu_fast_ab.py c2 256 > $OUT/pmc_fast_${c}_$TAG.log 2> $OUT/pmc_fast_${c}_$TAG.err
done
cd $R
SCHEDS=1 bash tools/gpu_src_pmc.sh ${TAG}_src > $OUT/src_pmc_$TAG.log 2>&1
tail -12 $OUT/src_pmc_$TAG.log
