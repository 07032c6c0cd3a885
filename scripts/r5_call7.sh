#!/usr/bin/env bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/r5g
O=gpurun_out/r5g
stampit() { while IFS= read -r line; do echo "$(date +%s) $line"; done; }
( timeout 420 python -m pytest tests/test_asan.py tests/test_gpu_c5size.py tests/test_gpu_exchange.py -m gpu -v -p no:cacheprovider 2>&1 | stampit ) > $O/pytest_files.txt
( timeout 420 python -m pytest tests/test_gpu_kernels.py -m gpu -v -p no:cacheprovider -k "transform or upload or non_finite or icp_run_equals or too_few or variants_and_overflow or movable_selection or massive_duplicate or one_launch_forms or barrier_timeout or download_both or knn1_bit_exact or ties_lowest or upper_bound or select_in_range_between" 2>&1 | stampit ) > $O/pytest_kernels_rest.txt
grep -c PASSED $O/pytest_files.txt $O/pytest_kernels_rest.txt
