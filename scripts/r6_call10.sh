#!/usr/bin/env bash
# round 6, call 10: grid barrier with wrapping arrival counters, the tail's lazy range -- tests, sweeps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c10; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_c3shape.py tests/test_gpu_fuzz.py tests/test_gpu_run.py -q -m gpu -x -k "tail_window or windowed_rejection or longest_barrier or barrier_timeout or one_launch or q_sweep or large_q or duplicate or c3 or fuzz or icp_run_equals or run" -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
for i in 1 2 3; do python scripts/trace_c4.py 2>&1 | grep "it/s" | tail -1; done
SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep -E "\[tail\]" | tail -3
timeout 600 python scripts/q_sweep.py 1e7 1000 2049 10000 16384 32768 100000 1000000 > $O/q_sweep.txt 2>&1; cat $O/q_sweep.txt
timeout 600 python scripts/steady_sweep.py 1e7 32768 100000 1000000 > $O/steady_sweep.txt 2>&1; cat $O/steady_sweep.txt
