#!/usr/bin/env bash
# round 6: lanes per query in the cold search of a million queries
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/cold
for g in 8 16; do
  echo "SICP_NN_GROUP=$g"
  SICP_NN_GROUP=$g SICP_GRID_OCC=0 timeout 300 python scripts/cold_match.py 1e7 1e6 2>&1 | tee gpurun_out/cold/cold_match_q1m_group$g.txt
done
