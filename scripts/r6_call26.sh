#!/usr/bin/env bash
# round 6: the mid-Q minimisation on ONE block (no grid barrier) against ten
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/midq
for lim in 0 3072 4096 8192 16384; do
  echo "SICP_LM_ONE_BLOCK_Q=$lim"
  SICP_LM_ONE_BLOCK_Q=$lim timeout 300 python scripts/q_sweep.py 1e7 2049 3000 4096 6000 8192 10000 16384 2>&1 | cut -c1-150 | tee gpurun_out/midq/q_sweep_lm_one_block_$lim.txt
done
