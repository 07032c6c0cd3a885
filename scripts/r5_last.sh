#!/usr/bin/env bash
# after the last source edits: counters + headline bench of THIS tree, the new C3-shape tests, the nonuniform cloud's legs
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
STEPS="profiles bench" bash scripts/final_measure.sh r5d > /dev/null 2>&1
O=gpurun_out/final_r5d
timeout 100 python -m pytest tests/test_gpu_c3shape.py tests/test_gpu_terrestrial.py -q -x --durations=8 -p no:cacheprovider > $O/pytest_c3_T.txt 2>&1; echo "c3 + T tests rc $?" >> $O/steps.log
timeout 60 python bench.py --config T --out $O/bench_T.json > /dev/null 2> $O/bench_T.err; echo "bench T rc $?" >> $O/steps.log
cat $O/steps.log; tail -14 $O/pytest_c3_T.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/final_r5d/bench_C4.json")); print("C4", d["value"], d["roofline"]["traffic"], d["roofline_match"]["traffic"], d["parity"]["ok"])
d = json.load(open("gpurun_out/final_r5d/bench_T.json"))
print("T", d["ms_per_step"], [(d[t]["ms_per_step"], d[t]["roofline"]["kernel"], d[t]["kernels_instrumented"]["match"]["avg_ms"], d[t]["parity"]["ok"]) for t in ("throughput_point", "throughput_point_q1000000")])
PY
