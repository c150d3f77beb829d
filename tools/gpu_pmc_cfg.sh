#!/bin/bash
# SQ counters of one tools/gpu_configs.py config: bash tools/gpu_pmc_cfg.sh <config-name> [env assignments...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
CFG=$1
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1)); rm -rf $OUT/pmcc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmcc_$i -o pmc -- python $R/tools/gpu_configs.py $CFG > /dev/null 2> $OUT/pmcc_$i.err
  python - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/pmcc_$i/**/*counter_collection.csv", recursive=True)
if not fs: print("no output, see pmcc_$i.err"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    if any(t in k for t in ("trace_paths", "primary", "rt_jit", "persistent")): acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items(): print(k[:44], {c: "%.4g" % x for c, x in v.items()})
import json, os
path = "$OUT/pmc_cfg_$CFG.json"
old = json.load(open(path)) if os.path.exists(path) and $i > 1 else {}
for k, v in acc.items(): old.setdefault(k.split("(")[0][:60], {}).update(v)
json.dump(old, open(path, "w"), indent=1)
PY
done
