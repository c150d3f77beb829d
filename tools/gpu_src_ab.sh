#!/bin/bash
# src/-form pool kernel, A/B of option sets at the three frame sizes:  bash tools/gpu_src_ab.sh <tag> "<opts A>" "<opts B>" ...
# (an option set is a space-separated list of KEY=VALUE for tools/gpu_src_prof.py; "-" = defaults)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-ab}; shift
: > $OUT/${TAG}_ab.jsonl
for size in ${SIZES:-"768 432" "1920 1080" "3840 2160"}; do
 for opts in "$@"; do
  [ "$opts" = "-" ] && o="" || o="$opts"
  echo "{\"opts\": \"$opts\"}" >> $OUT/${TAG}_ab.jsonl
  python $R/tools/gpu_src_prof.py $size 1 256 $o >> $OUT/${TAG}_ab.jsonl 2>> $OUT/${TAG}_ab.err
 done
done
cat $OUT/${TAG}_ab.jsonl
