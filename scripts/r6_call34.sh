#!/usr/bin/env bash
# round 6: block_gram asks for a correspondence's data together with its verdict (one round trip instead of two)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in simpleicp_amd/_obj/libsimpleicp_hip_foldold.so "" simpleicp_amd/_obj/libsimpleicp_hip_foldold.so ""; do
  echo "SICP_LIBRARY=$lib"
  SICP_LIBRARY=$lib timeout 300 python scripts/q_sweep.py 1e7 2049 10000 32768 2>&1 | cut -c1-150
done
