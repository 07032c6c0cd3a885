#!/usr/bin/env bash
# round 6: the filtered search's full flavour without the second geometry pass (rows dropped by the hit's ball, x ranges not narrowed)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/cold
for lib in "" simpleicp_amd/_obj/libsimpleicp_hip_nonearest.so; do
  echo "SICP_LIBRARY=$lib"
  SICP_LIBRARY=$lib timeout 300 python scripts/cold_match.py 1e7 1e6 2>&1 | tee gpurun_out/cold/cold_match_narrow_${lib:+no}.txt
  SICP_LIBRARY=$lib timeout 300 python scripts/q_sweep.py 1e7 1000000 2>&1 | cut -c1-150
done
