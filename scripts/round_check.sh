#!/usr/bin/env bash
# One short GPU call that re-validates the tree:  scripts/round_check.sh  -> gpurun_out/check/
#   1. the operator-level tests and the C host (newest code first), 2. the rest of the GPU suite with durations,
#   3. the default bench line, 4. (EXTRAS=1) what the operator-by-operator road costs, the host's enqueue trace.  Every step under its own timeout; partial results survive a cut-off call.
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/check
mkdir -p "$OUT"
cd "$REPO"
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" >> "$OUT/steps.log"; }
stamp start
timeout ${T_OPS:-240} python -m pytest tests/test_gpu_operators.py tests/test_host_api.py -m gpu -q --durations=8 > "$OUT/ops.log" 2>&1
stamp "operator tests rc=$?"
timeout ${T_SUITE:-420} python -m pytest tests -m gpu -q --durations=15 --ignore=tests/test_gpu_operators.py > "$OUT/suite.log" 2>&1
stamp "suite rc=$?"
timeout ${T_BENCH:-240} python bench.py --out "$OUT/bench_C4.json" > "$OUT/bench.log" 2> "$OUT/bench.err"
stamp "bench rc=$?"
if [ "${EXTRAS:-0}" = "1" ]; then      # the two small records of profiles/r2 (operator_cost.txt, host_enqueue.txt)
  timeout ${T_OPCOST:-120} python scripts/operator_cost.py > "$OUT/operator_cost.txt" 2>&1
  stamp "operator cost rc=$?"
  SICP_SOLVE_TRACE=host timeout 120 python scripts/trace_c4.py 2>&1 | grep -E "\[host\]|iterations:" | tail -26 > "$OUT/host_enqueue.txt"
  stamp "host enqueue trace rc=$?"
fi
tail -3 "$OUT/ops.log"; tail -3 "$OUT/suite.log"; cat "$OUT/steps.log"
[ -f "$OUT/operator_cost.txt" ] && cat "$OUT/operator_cost.txt"
