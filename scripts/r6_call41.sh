#!/usr/bin/env bash
# round 6: the full flavour behind the lean one works its marked slots off 64 per wave -- parity, then cold + steady at large Q
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -m gpu -x -k "filtered or filter or grid_knn or nn16 or many_queries or fuzz or large_q or q_sweep" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python scripts/cold_match.py 1e7 1e6 2>&1
timeout 300 python scripts/steady_sweep.py 1e7 262144 500000 1000000 2>&1 | cut -c1-150
timeout 300 python scripts/q_sweep.py 1e7 262144 1000000 2>&1 | cut -c1-150
