#!/usr/bin/env bash
# Everything the round's records are made of, in one GPU call:  scripts/final_measure.sh <tag>  -> gpurun_out/final_<tag>/ (+ prof_<tag>/)
set -u
TAG=${1:-r2}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p "$OUT"
cd "$REPO"
for C in C4 C1 C2 C3; do
  timeout 900 python bench.py --config $C --out "$OUT/bench_$C.json" > /dev/null 2> "$OUT/bench_$C.err"
done
timeout 600 python bench.py --force-exchange --partition cloud --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --out "$OUT/bench_C4_exchange_cloud.json" > /dev/null 2> "$OUT/bench_x1.err"
timeout 600 python bench.py --force-exchange --partition queries --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --out "$OUT/bench_C4_exchange_queries.json" > /dev/null 2> "$OUT/bench_x2.err"
timeout 900 python scripts/q_sweep.py 1e7 1000 2048 2049 10000 16384 32768 100000 1000000 > "$OUT/q_sweep.txt" 2>&1
timeout 600 python scripts/datasets_run.py > "$OUT/datasets_run.txt" 2>&1
timeout 600 python scripts/cold_match.py > "$OUT/cold_match.txt" 2>&1
SICP_SOLVE_TRACE=1 timeout 600 python scripts/trace_c4.py 2>&1 | grep "\[tail\]" | tail -22 > "$OUT/tail_trace.txt"
scripts/kernel_timeline.sh c4_$TAG scripts/trace_c4.py > "$OUT/kernel_timeline_c4.txt" 2>&1
python scripts/iter_timeline.py gpurun_out/kt_c4_$TAG > "$OUT/iter_timeline.txt" 2>&1
scripts/kernel_timeline.sh q1m_$TAG scripts/q_sweep.py 1e7 1000000 > "$OUT/kernel_timeline_q1m.txt" 2>&1
timeout 900 python scripts/run_profile.py > "$OUT/run_profile.txt" 2>&1
timeout 900 python scripts/run_profile.py 1e7 1000 1.0 > "$OUT/run_profile_overlap.txt" 2>&1
scripts/gpu_profile.sh $TAG > "$OUT/gpu_profile.log" 2>&1
timeout 1500 python bench.py --config C5size --repeats 5 --no-parity --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --out "$OUT/bench_C5size.json" > /dev/null 2> "$OUT/bench_C5size.err"
ls "$OUT"
