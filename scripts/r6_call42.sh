#!/usr/bin/env bash
# round 6: when does the lean flavour go first?  (SICP_FAR_MOVE: cells per iteration the estimate may still move by; default 0.75)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 0.75 1.5 3 1000; do
  echo "SICP_FAR_MOVE=$v"
  SICP_FAR_MOVE=$v timeout 300 python scripts/q_sweep.py 1e7 262144 1000000 2>&1 | cut -c1-150
done
