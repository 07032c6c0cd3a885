#!/usr/bin/env bash
# round 5, first GPU call: the new search kernels against the oracle, then what they cost
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --maxfail=20 -p no:cacheprovider \
  -k "filtered_scan_equals_brute_force or filtered_iteration_uses_previous or many_queries_search_flavours or longest_barrier or windowed_rejection or large_q_iteration_multi or grid_knn_equals" > $O/pytest_kernels.txt 2>&1
echo "kernels rc $?" >> $O/pytest_kernels.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_run.py -q --maxfail=20 -p no:cacheprovider > $O/pytest_full.txt 2>&1
echo "full rc $?" >> $O/pytest_full.txt
timeout 600 python scripts/match_ab.py 1e7 1e6 > $O/match_ab_q1m.txt 2>&1
timeout 300 python scripts/match_ab.py 1e7 1e5 "exact:SICP_NN16=exact" "far:SICP_NN16=far" "near:SICP_NN16=near" > $O/match_ab_q100k.txt 2>&1
( SICP_BOXES=0 timeout 200 python scripts/cold_match.py; echo "--- boxes on"; timeout 200 python scripts/cold_match.py ) > $O/cold_match.txt 2>&1
timeout 200 python scripts/dataset_profile.py webots bunny > $O/dataset_profile.txt 2>&1
timeout 200 bash scripts/pmc_any.sh calib "FETCH_SIZE" $REPO/scripts/ubench/gather_calib; cp gpurun_out/pmc_calib.txt $O/ 2>/dev/null; cp gpurun_out/pmc_calib/cmd.out $O/calib_cmd.out 2>/dev/null
tail -3 $O/pytest_kernels.txt $O/pytest_full.txt
