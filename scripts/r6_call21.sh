#!/usr/bin/env bash
# round 6: the filtered search's full flavour with the cells' occupancy bitmap -- parity, then the cold iterations at Q = 1 M with and without
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/cold
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "filtered or filter or grid_knn or nn16 or many_queries" -p no:cacheprovider 2>&1 | tail -4
for occ in 1 0; do
  echo "SICP_GRID_OCC=$occ"
  SICP_GRID_OCC=$occ timeout 300 python scripts/cold_match.py 1e7 1e6 2>&1 | tee gpurun_out/cold/cold_match_q1m_occ$occ.txt
  SICP_GRID_OCC=$occ timeout 300 python scripts/q_sweep.py 1e7 1000000 2>&1 | tail -2 | tee gpurun_out/cold/q_sweep_1m_occ$occ.txt
done
