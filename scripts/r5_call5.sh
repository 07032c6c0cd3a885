#!/usr/bin/env bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/r5e
O=gpurun_out/r5e
timeout 900 python -m pytest tests/test_gpu_terrestrial.py tests/test_gpu_run.py -q --maxfail=20 -p no:cacheprovider > $O/pytest_a.txt 2>&1
echo "a rc $?" >> $O/pytest_a.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --maxfail=20 -p no:cacheprovider -k "many_queries_search_flavours or filtered_scan_equals or filtered_iteration_uses or knn1_ or select_in_range or q_sweep_every_tail" > $O/pytest_b.txt 2>&1
echo "b rc $?" >> $O/pytest_b.txt
timeout 400 python bench.py --config T --no-cpu-baseline --throughput-q 0 --out $O/bench_T.json > $O/bench_T.line 2> $O/bench_T.err
SICP_GRID_POINTWISE=0 timeout 400 python bench.py --config T --no-cpu-baseline --throughput-q 0 --no-end-to-end --no-bruteforce-leg --out $O/bench_T_nopointwise.json > $O/bench_T_np.line 2> $O/bench_T_np.err
timeout 400 python bench.py --config T --correspondences 1000 --no-cpu-baseline --throughput-q 0 --no-end-to-end --no-bruteforce-leg --out $O/bench_T_q1000.json > $O/bench_T_q1000.line 2> $O/bench_T_q1000.err
tail -n 3 $O/pytest_a.txt $O/pytest_b.txt; tail -2 $O/bench_T.err
