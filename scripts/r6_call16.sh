#!/usr/bin/env bash
# round 6, run() end to end: where the 25 ms go (per ABI call), what the host side of the download costs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/e2e
{ echo "THP: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>&1)"; echo "defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag 2>&1)"; echo "nproc $(nproc)"; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; python -c "import numpy; print(numpy.__version__)"; } > gpurun_out/e2e/host.txt 2>&1
g++ -O2 -pthread -o /tmp/host_mem scripts/ubench/host_mem.cpp && /tmp/host_mem > gpurun_out/e2e/host_mem.txt 2>&1
timeout 600 python scripts/e2e_probe.py > gpurun_out/e2e/e2e_probe.txt 2>&1
timeout 300 python scripts/download_probe.py > gpurun_out/e2e/download_probe.txt 2>&1
cat gpurun_out/e2e/host.txt; cat gpurun_out/e2e/host_mem.txt; head -12 gpurun_out/e2e/e2e_probe.txt; cat gpurun_out/e2e/download_probe.txt
