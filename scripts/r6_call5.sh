#!/usr/bin/env bash
# round 6, call 5: the few-workgroup rejection for 2048 < Q <= 16384 (k_reject_mb) -- tests, Q sweep, C3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c5; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_operators.py tests/test_gpu_c3shape.py tests/test_gpu_fuzz.py tests/test_gpu_run.py -q -m gpu -x -k "tail_window or q_sweep or icp_run_equals or iteration or rejection or one_launch or operator or c3 or fuzz or barrier or run" -p no:cacheprovider > $O/pytest_reject.txt 2>&1; echo "pytest rc $?"; tail -12 $O/pytest_reject.txt
timeout 300 python scripts/q_sweep.py 1e7 1000 2048 2049 4096 8192 10000 16384 32768 > $O/q_sweep.txt 2>&1; cat $O/q_sweep.txt
timeout 600 python bench.py --config C3 --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --out $O/bench_C3_quick.json > /dev/null 2> $O/bench_C3_quick.err; echo "bench C3 rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6c5/bench_C3_quick.json"))
print("C3", d["value"], d["ms_per_step"], d.get("steady_us_per_step"), d["parity"]["ok"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels_instrumented"].items()})
PY
