#!/usr/bin/env bash
# grid granularity A/B: points per occupied cell (SICP_GRID_TARGET; the cell table's size limit needs a rebuild with another
# `cap` in grid_build, sicp_api.cpp)
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT=gpurun_out/ab_target; mkdir -p $OUT
for CAP in ${TARGETS:-16 8 4}; do
  echo "== SICP_GRID_TARGET=$CAP"
  SICP_GRID_TARGET=$CAP timeout 600 python bench.py --config C5size --repeats 3 --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --out $OUT/bench_C5_cap$CAP.json > /dev/null 2>&1
  python -c "
import json; d=json.load(open('$OUT/bench_C5_cap$CAP.json')); print('C5size', round(d['ms_per_step'],4), 'ms/step', 'parity', d['parity']['ok'], 'cand/query', round(d['roofline_match']['candidates_per_query'],1), 'rows', round(d['roofline_match']['grid_rows_per_query'],1), 'grid build', round(d['setup']['grid_build_ms'],2), 'normals', round(d['setup']['normals_ms'],2))"
done
