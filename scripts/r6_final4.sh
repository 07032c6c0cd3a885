#!/usr/bin/env bash
# round 6, the records on the tree with the overlapped run() (ABI 7): everything, tests last
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/final_measure.sh r6
PASSES="trace fetch write" scripts/gpu_profile.sh r6_C3 --config C3 > gpurun_out/gpu_profile_C3.log 2>&1
