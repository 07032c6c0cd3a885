#!/usr/bin/env bash
# after the k_grid_nn fix: the whole GPU suite with durations, the default line and the small configs again, the default line's profile passes
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; OUT=gpurun_out/final_r5b; mkdir -p $OUT
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" >> "$OUT/steps.log"; }
timeout 300 python bench.py --out "$OUT/bench_C4.json" > "$OUT/bench_C4.line" 2> "$OUT/bench_C4.err"; stamp "bench C4 rc=$?"
for C in C1 C2 C3 T; do timeout 300 python bench.py --config $C --out "$OUT/bench_$C.json" > /dev/null 2> "$OUT/bench_$C.err"; stamp "bench $C rc=$?"; done
timeout 300 python bench.py --force-exchange --partition cloud --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 100000 --out "$OUT/bench_C4_exchange_cloud.json" > /dev/null 2> "$OUT/bench_x1.err"; stamp "exchange cloud rc=$?"
timeout 300 python scripts/q_sweep.py 1e7 1000 2048 2049 4096 8192 10000 16384 32768 100000 196608 1000000 > "$OUT/q_sweep.txt" 2>&1; stamp "q_sweep rc=$?"
timeout 200 python scripts/datasets_run.py > "$OUT/datasets_run.txt" 2>&1; stamp "datasets rc=$?"
timeout 200 python scripts/cold_match.py > "$OUT/cold_match.txt" 2>&1; stamp "cold match"
SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep "\[tail\]" | tail -22 > "$OUT/tail_trace.txt"; stamp "tail trace"
scripts/kernel_timeline.sh c4_r5b scripts/trace_c4.py > "$OUT/kernel_timeline_c4.txt" 2>&1
python scripts/iter_timeline.py gpurun_out/kt_c4_r5b > "$OUT/iter_timeline.txt" 2>&1; stamp "iter timeline"
PASSES="trace fetch write sq1" scripts/gpu_profile.sh r5 > "$OUT/gpu_profile.log" 2>&1; stamp "profile default"
PASSES="trace fetch write sq1" scripts/gpu_profile.sh r5_q1000000 --correspondences 1000000 >> "$OUT/gpu_profile.log" 2>&1; stamp "profile Q=1M"
timeout 1100 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=30 > "$OUT/pytest_gpu.txt" 2>&1; stamp "pytest -m gpu rc=$?"
tail -45 "$OUT/pytest_gpu.txt"; cat "$OUT/steps.log"; cat "$OUT/bench_C4.line" | cut -c1-600
