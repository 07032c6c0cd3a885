#!/usr/bin/env bash
# A/B of this round's kernel changes on one GPU box:  scripts/ab_r3.sh  -> gpurun_out/ab_r3/
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/ab_r3
mkdir -p "$OUT"
cd "$REPO"
QS="1000 10000 32768 100000 1000000"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x > "$OUT/kernels.log" 2>&1; tail -3 "$OUT/kernels.log"
timeout 300 python scripts/q_sweep.py 1e7 $QS > "$OUT/q_sweep_default.txt" 2>&1
SICP_HSEL=launches timeout 300 python scripts/q_sweep.py 1e7 32768 100000 1000000 > "$OUT/q_sweep_hsel_launches.txt" 2>&1
SICP_MATCH_EPILOGUE=0 timeout 300 python scripts/q_sweep.py 1e7 $QS > "$OUT/q_sweep_no_epilogue.txt" 2>&1
SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep "\[tail\]" | tail -22 > "$OUT/tail_trace.txt"
SICP_MATCH_EPILOGUE=0 SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep "\[tail\]" | tail -22 > "$OUT/tail_trace_no_epilogue.txt"
timeout 700 python -m pytest tests/test_gpu_fullsize.py -m gpu -q > "$OUT/fullsize.log" 2>&1; tail -3 "$OUT/fullsize.log"
head -20 "$OUT"/q_sweep_*.txt; tail -4 "$OUT/tail_trace.txt" "$OUT/tail_trace_no_epilogue.txt"
