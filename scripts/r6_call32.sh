#!/usr/bin/env bash
# round 6: k_lm_all's fold of the block partials with all loads in flight
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "" simpleicp_amd/_obj/libsimpleicp_hip_foldahead.so; do
  echo "SICP_LIBRARY=$lib"
  SICP_LIBRARY=$lib timeout 300 python scripts/q_sweep.py 1e7 196608 500000 1000000 2>&1 | cut -c1-150
done
