#!/usr/bin/env bash
# round 6: the lean flavour of the filtered search at 5 (default) / 6 / 8 waves per SIMD
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "" simpleicp_amd/_obj/libsimpleicp_hip_occ6.so simpleicp_amd/_obj/libsimpleicp_hip_occ8.so ""; do
  echo "SICP_LIBRARY=$lib"
  SICP_LIBRARY=$lib timeout 300 python scripts/steady_sweep.py 1e7 500000 1000000 2>&1 | cut -c1-150
  SICP_LIBRARY=$lib timeout 300 python scripts/q_sweep.py 1e7 1000000 2>&1 | cut -c1-150
done
