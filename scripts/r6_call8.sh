#!/usr/bin/env bash
# round 6, call 8: the single-workgroup tail with 512 lanes (8 waves) against 256 -- tests on the variant, tail trace, timeline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c8; mkdir -p $O
V=$PWD/simpleicp_amd/_obj/libsimpleicp_hip_tb512.so
SICP_LIBRARY=$V timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_run.py -q -m gpu -x -k "tail_window or q_sweep or icp_run_equals or iteration_vs_oracle or too_few or rejection or run" -p no:cacheprovider > $O/pytest_tb512.txt 2>&1; echo "pytest tb512 rc $?"; tail -3 $O/pytest_tb512.txt
for i in 1 2 3; do python scripts/trace_c4.py 2>&1 | grep "it/s" | tail -1; SICP_LIBRARY=$V python scripts/trace_c4.py 2>&1 | grep "it/s" | tail -1; done
SICP_LIBRARY=$V SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep -E "\[tail\]" | tail -22 > $O/tail_trace_tb512.txt; tail -4 $O/tail_trace_tb512.txt
SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep -E "\[tail\]" | tail -22 > $O/tail_trace_tb256.txt; tail -2 $O/tail_trace_tb256.txt
SICP_LIBRARY=$V scripts/kernel_timeline.sh c4_r6c8_tb512 scripts/trace_c4.py > $O/kernel_timeline_c4_tb512.txt 2>&1; python scripts/iter_timeline.py gpurun_out/kt_c4_r6c8_tb512 > $O/iter_timeline_tb512.txt 2>&1; cat $O/iter_timeline_tb512.txt
SICP_LIBRARY=$V timeout 300 python scripts/q_sweep.py 1e7 256 512 1000 1024 1025 2048 > $O/q_sweep_tb512.txt 2>&1; cat $O/q_sweep_tb512.txt
timeout 300 python scripts/q_sweep.py 1e7 256 512 1000 1024 1025 2048 > $O/q_sweep_tb256.txt 2>&1; cat $O/q_sweep_tb256.txt
