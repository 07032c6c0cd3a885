#!/usr/bin/env bash
# round 6, call 2: windowed tail selection + key exchange over the callback -- tests, tail trace, headline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "tail_window or q_sweep or icp_run_equals or iteration_vs_oracle or too_few" -p no:cacheprovider > $O/pytest_tail.txt 2>&1; echo "pytest tail rc $?"; tail -5 $O/pytest_tail.txt
SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep -E "\[tail\]|iterations" | tail -26 > $O/tail_trace.txt; tail -24 $O/tail_trace.txt
timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --out $O/bench_C4_quick.json > $O/bench_quick.line 2> $O/bench_quick.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6c2/bench_C4_quick.json"))
print("C4", d["value"], d["ms_per_step"], d.get("steady_us_per_step"), d["parity"]["ok"])
PY
timeout 1500 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_run.py -q -m gpu -x -p no:cacheprovider --durations=8 > $O/pytest_xchg.txt 2>&1; echo "pytest exchange rc $?"; tail -15 $O/pytest_xchg.txt
