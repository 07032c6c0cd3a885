#!/usr/bin/env bash
# round 6, call 12: the cold first match with a finer subsample (stride 16 / 32 against 64) at the headline workload and at Q = 1 M
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c12; mkdir -p $O
for v in "" sub32 sub16; do
  L=""; [ -n "$v" ] && L=$PWD/simpleicp_amd/_obj/libsimpleicp_hip_$v.so
  echo "== stride ${v:-64}"
  for i in 1 2; do SICP_LIBRARY=$L python scripts/trace_c4.py 2>&1 | grep "it/s" | tail -1; done
  SICP_LIBRARY=$L scripts/kernel_timeline.sh c4_r6c12_$v scripts/trace_c4.py > $O/kt_$v.txt 2>&1; python scripts/iter_timeline.py gpurun_out/kt_c4_r6c12_$v | head -3
  SICP_LIBRARY=$L timeout 300 python scripts/q_sweep.py 1e7 1000 10000 1000000 2>&1 | tail -3
done
