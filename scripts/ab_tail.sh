#!/usr/bin/env bash
# tail-side A/B on one GPU box: correctness of the small-Q path, then C4 timing + the tail's cycle split   -> gpurun_out/ab_tail/
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/ab_tail
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_run.py -m gpu -q -x > "$OUT/kernels.log" 2>&1; tail -3 "$OUT/kernels.log"
timeout 300 python scripts/q_sweep.py 1e7 1000 2048 10000 100000 > "$OUT/q_sweep.txt" 2>&1; cat "$OUT/q_sweep.txt"
SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep "\[tail\]" | tail -22 > "$OUT/tail_trace.txt"; head -9 "$OUT/tail_trace.txt"; tail -3 "$OUT/tail_trace.txt"
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --out "$OUT/bench.json" > /dev/null 2>&1
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['parity']['ok'])"
for D in dragon bunny; do :; done
timeout 300 python scripts/datasets_run.py 2>&1 | cut -c1-40,150-400 | tail -8
