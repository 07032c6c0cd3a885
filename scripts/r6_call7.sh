#!/usr/bin/env bash
# round 6, call 7: the call-sequence fuzz (32 seeds + the reverted-fix variant), then the whole GPU suite with durations
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c7; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --durations=5 > $O/pytest_fuzz.txt 2>&1; echo "fuzz rc $?"; tail -30 $O/pytest_fuzz.txt
