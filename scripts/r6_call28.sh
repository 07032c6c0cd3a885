#!/usr/bin/env bash
# round 6: (a) parity of the filtered search without the second geometry pass; (b) the same question for the exact four-per-wave search
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/cold
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -m gpu -x -k "filtered or filter or grid_knn or nn16 or many_queries or fuzz or large_q or q_sweep" -p no:cacheprovider 2>&1 | tail -3
for lib in "" simpleicp_amd/_obj/libsimpleicp_hip_nonarrow16.so; do
  echo "SICP_LIBRARY=$lib"
  SICP_LIBRARY=$lib timeout 300 python scripts/cold_match.py 1e7 1e5 2>&1
  SICP_LIBRARY=$lib timeout 300 python scripts/q_sweep.py 1e7 8192 10000 32768 100000 2>&1 | cut -c1-150
done
