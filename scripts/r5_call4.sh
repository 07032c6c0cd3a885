#!/usr/bin/env bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/r5d
O=gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --maxfail=20 -p no:cacheprovider -k "many_queries_search_flavours or filtered_scan_equals or filtered_iteration_uses" > $O/pytest_b.txt 2>&1
echo "b rc $?" >> $O/pytest_b.txt
timeout 900 python -m pytest tests/test_gpu_terrestrial.py tests/test_gpu_fullsize.py -q --maxfail=20 -p no:cacheprovider -k "flavours or large_q or select_in_range or overlap" > $O/pytest_a.txt 2>&1
echo "a rc $?" >> $O/pytest_a.txt
AB_EARLY=4 timeout 600 python scripts/match_ab.py 1e7 1e6 "near:SICP_NN16=near" "far:SICP_NN16=far" > $O/match_ab_q1m.txt 2>&1
timeout 400 python bench.py --config T --no-cpu-baseline --throughput-q 0 --out $O/bench_T.json > $O/bench_T.line 2> $O/bench_T.err
timeout 400 python bench.py --config C3 --no-cpu-baseline --throughput-q 0 --out $O/bench_C3.json > $O/bench_C3.line 2> $O/bench_C3.err
tail -n 3 $O/pytest_a.txt $O/pytest_b.txt; tail -2 $O/bench_T.err
