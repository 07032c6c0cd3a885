#!/usr/bin/env bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p gpurun_out/r5f
O=gpurun_out/r5f
timeout 900 python -m pytest tests/test_gpu_kernels.py -q --maxfail=20 -p no:cacheprovider -k "many_queries_search_flavours or filtered_scan_equals or filtered_iteration_uses" > $O/pytest_b.txt 2>&1
echo "b rc $?" >> $O/pytest_b.txt
AB_EARLY=3 timeout 600 python scripts/match_ab.py 1e7 1e6 "near:SICP_NN16=near" > $O/match_ab_q1m.txt 2>&1
AB_EARLY=3 SICP_LIBRARY=$REPO/simpleicp_amd/_obj/libsimpleicp_hip_nnocc6.so timeout 600 python scripts/match_ab.py 1e7 1e6 "near-occ6:SICP_NN16=near" > $O/match_ab_q1m_occ6.txt 2>&1
AB_EARLY=6 timeout 900 python scripts/match_ab.py 1e8 1e6 "near:SICP_NN16=near" > $O/match_ab_c5size.txt 2>&1
tail -n 3 $O/pytest_b.txt; cat $O/match_ab_q1m.txt $O/match_ab_q1m_occ6.txt $O/match_ab_c5size.txt
