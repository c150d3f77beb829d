#!/bin/bash
# round 6, one full visit: build, pytest -m gpu, smoke(), the bench line (-> gpurun_out/<tag>_bench.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r06}
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/${TAG}_build.log 2>&1 || { tail -20 $OUT/${TAG}_build.log; exit 1; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/${TAG}_pytest.log
  grep -h "^\[fast\]\|^\[refpin\]" $OUT/${TAG}_pytest.log | head -40
fi
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; tail -1 $OUT/${TAG}_smoke.log
timeout 1500 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
j = json.load(open("$OUT/${TAG}_bench.json"))
print("value", j["value"], "ms", j["ms_per_step"], "frac", j["roofline"]["frac"])
print("jit", {k: v for k, v in j.get("jit", {}).items() if not k.endswith("note")})
for k, v in j.get("configs", {}).items():
    print(k, v["value"], v["ms_per_step"], v["frac"], v["run_time_kernels"])
print("frame", {k: v for k, v in j.get("frame", {}).items() if k.startswith(("ms_", "device", "sample", "frames"))})
print("cpu", j.get("cpu_baseline", {}).get("value"))
PY
