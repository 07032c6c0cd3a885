#!/usr/bin/env bash
# round 6: on the terrestrial stand-in, where does four-queries-per-wave overtake one wave per query?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/t
for q in 5200 20000 40000 100000; do
for v in 5120 1000000; do
  SICP_NN16_MIN_Q=$v SICP_NN16F_MIN_Q=100000000 timeout 300 python bench.py --config T --correspondences $q --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --no-parity --throughput-q 0 --out gpurun_out/t/T_${q}_$v.json > /dev/null 2>&1
  python - gpurun_out/t/T_${q}_$v.json $q $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("Q", sys.argv[2], "nn16 from", sys.argv[3], ":", round(d["ms_per_step"] * 1e3, 1), "us/it, steady", round(d["steady_us_per_step"], 1), "match", round(d["kernels_instrumented"]["match"]["avg_ms"] * 1e3, 1))
PY
done; done
