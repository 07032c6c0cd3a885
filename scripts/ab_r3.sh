#!/usr/bin/env bash
# A/B of this round's kernel changes on one GPU box:  scripts/ab_r3.sh  -> gpurun_out/ab_r3/
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/ab_r3
mkdir -p "$OUT"
cd "$REPO"
QS="${QS:-1000 10000 32768 100000 1000000}"
QL="${QL:-10000 32768 100000 1000000}"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x > "$OUT/kernels.log" 2>&1; tail -3 "$OUT/kernels.log"
timeout 300 python scripts/q_sweep.py 1e7 $QS > "$OUT/q_sweep_default.txt" 2>&1
SICP_HSEL=launches timeout 300 python scripts/q_sweep.py 1e7 $QL > "$OUT/q_sweep_hsel_launches.txt" 2>&1
SICP_LM=launches timeout 300 python scripts/q_sweep.py 1e7 $QL > "$OUT/q_sweep_lm_launches.txt" 2>&1
if [ "${FULL:-1}" = "1" ]; then
  timeout 700 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "large_q or mid_q or dataframes" > "$OUT/fullsize.log" 2>&1; tail -3 "$OUT/fullsize.log"
fi
head -20 "$OUT"/q_sweep_*.txt
SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep "\[tail\]" | tail -22 > "$OUT/tail_trace.txt"; tail -3 "$OUT/tail_trace.txt"
