#!/usr/bin/env bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/r5h
O=gpurun_out/r5h
stampit() { while IFS= read -r line; do echo "$(date +%s) $line"; done; }
# headline first (the plain instantiation of k_grid_nn is back)
timeout 300 python scripts/q_sweep.py 1e7 1000 2048 10000 > $O/q_sweep_head.txt 2>&1
( timeout 700 python -m pytest tests/test_gpu_kernels.py -m gpu -v -p no:cacheprovider -k "not (transform or upload or non_finite or icp_run_equals or too_few or variants_and_overflow or movable_selection or massive_duplicate or one_launch_forms or barrier_timeout or download_both or knn1_bit_exact or ties_lowest or upper_bound or select_in_range_between)" 2>&1 | stampit ) > $O/pytest_kernels_main.txt
grep -c PASSED $O/pytest_kernels_main.txt; cat $O/q_sweep_head.txt
