#!/usr/bin/env bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/r5c
O=gpurun_out/r5c
timeout 900 python -m pytest tests/test_gpu_terrestrial.py tests/test_gpu_run.py tests/test_gpu_operators.py -q --maxfail=20 -p no:cacheprovider > $O/pytest_a.txt 2>&1
echo "a rc $?" >> $O/pytest_a.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --maxfail=20 -p no:cacheprovider -k "many_queries_search_flavours or filtered_scan_equals or q_sweep_every_tail or large_q_iteration or normals_vs_oracle or iteration_vs_oracle or knn_sweep_equals" > $O/pytest_b.txt 2>&1
echo "b rc $?" >> $O/pytest_b.txt
timeout 400 python bench.py --config T --no-cpu-baseline --throughput-q 0 --out $O/bench_T.json > $O/bench_T.line 2> $O/bench_T.err
timeout 400 python bench.py --config C3 --no-cpu-baseline --throughput-q 0 --out $O/bench_C3.json > $O/bench_C3.line 2> $O/bench_C3.err
AB_EARLY=4 timeout 600 python scripts/match_ab.py 1e7 1e6 "near:SICP_NN16=near" "far:SICP_NN16=far" > $O/match_ab_q1m.txt 2>&1
timeout 600 python scripts/q_sweep.py 1e7 1000 2048 2049 4096 8192 10000 16384 32768 100000 196608 1000000 > $O/q_sweep.txt 2>&1
AB_EARLY=7 timeout 900 python scripts/match_ab.py 1e8 1e6 "exact:SICP_NN16=exact" "near:SICP_NN16=near" > $O/match_ab_c5size.txt 2>&1
PMC_BY_GRID=1 timeout 400 bash scripts/pmc_any.sh iter0_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" python $REPO/scripts/cold_iter0.py 1e7 1e6 2
grep -E "grid_nn" gpurun_out/pmc_iter0_sq.txt > $O/pmc_iter0_sq.txt
tail -n 3 $O/pytest_a.txt $O/pytest_b.txt
