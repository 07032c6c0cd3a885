#!/usr/bin/env bash
# round 6: the terrestrial stand-in at 10 000 correspondences -- four queries per wave (default from 5 120) against one wave per query
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/t
for v in 5120 100000; do
  echo "SICP_NN16_MIN_Q=$v"
  SICP_NN16_MIN_Q=$v timeout 300 python bench.py --config T --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --out gpurun_out/t/T_$v.json > /dev/null 2>&1
  python - gpurun_out/t/T_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(round(d["value"]), round(d["ms_per_step"] * 1e3, 1), round(d["steady_us_per_step"], 1), d["parity"]["ok"], d.get("kernels_instrumented"))
PY
done
