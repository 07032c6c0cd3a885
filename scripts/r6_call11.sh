#!/usr/bin/env bash
# round 6, call 11: loop state stored by wave 1 while wave 0 publishes the record -- tail tests + trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_run.py tests/test_gpu_fuzz.py -q -m gpu -x -k "tail_window or q_sweep or icp_run_equals or iteration_vs_oracle or too_few or run or fuzz" -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
for i in 1 2 3; do python scripts/trace_c4.py 2>&1 | grep "it/s" | tail -1; done
scripts/kernel_timeline.sh c4_r6c11 scripts/trace_c4.py > $O/kernel_timeline_c4.txt 2>&1; python scripts/iter_timeline.py gpurun_out/kt_c4_r6c11 > $O/iter_timeline.txt 2>&1; tail -5 $O/iter_timeline.txt
