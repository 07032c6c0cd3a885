#!/usr/bin/env bash
# round 6, the records again on the final tree (threshold change): kernel tests around the new crossover, then profiles + benches + sweeps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/final_r6
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_c3shape.py tests/test_gpu_fuzz.py tests/test_gpu_run.py -q -m gpu -x -k "tail_window or q_sweep or one_launch or c3 or fuzz or flavours or run or iteration" -p no:cacheprovider > gpurun_out/final_r6/pytest_subset_after_threshold.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/final_r6/pytest_subset_after_threshold.txt
STEPS="profiles bench configs exchange sweeps traces c5" bash scripts/final_measure.sh r6
PASSES="trace fetch write" scripts/gpu_profile.sh r6_C3 --config C3 > gpurun_out/gpu_profile_C3.log 2>&1
