#!/usr/bin/env bash
# after the slot-invalidation fix: counters + headline of THIS tree, the regression test
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
STEPS="profiles bench" bash scripts/final_measure.sh r5e > /dev/null 2>&1
O=gpurun_out/final_r5e
timeout 40 python -m pytest tests/test_gpu_kernels.py -q -x -k "filter_slots_follow or many_queries_search_flavours" -p no:cacheprovider > $O/pytest_slots.txt 2>&1; echo "slot tests rc $?" >> $O/steps.log
cat $O/steps.log; tail -5 $O/pytest_slots.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/final_r5e/bench_C4.json")); print("C4", d["value"], d["roofline"]["traffic"], d["roofline_match"]["traffic"], d["parity"]["ok"])
PY
