#!/usr/bin/env bash
# Everything the round's records are made of, in one GPU call:  scripts/final_measure.sh <tag>  -> gpurun_out/final_<tag>/ (+ prof_<tag>*/)
# Order: counters first (the bench lines quote them), measurements, the tests last.  Every step under its own timeout; partial results survive a cut-off call.  STEPS selects (default: all).
set -u
TAG=${1:-r6}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/final_$TAG
STEPS=${STEPS:-"tests bench configs exchange sweeps traces profiles c5"}
mkdir -p "$OUT"
cd "$REPO"
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" >> "$OUT/steps.log"; }
has() { case " $STEPS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has profiles; then
  PASSES="trace fetch write sq1" scripts/gpu_profile.sh $TAG > "$OUT/gpu_profile.log" 2>&1; stamp "profile default"
  PASSES="trace fetch write sq1" scripts/gpu_profile.sh ${TAG}_q1000000 --correspondences 1000000 >> "$OUT/gpu_profile.log" 2>&1; stamp "profile Q=1M"
  # the bench lines below carry the counters' traffic only when profiles/latest_pmc.json was made from THIS tree's kernels
  python scripts/summarize_profile.py $TAG > "$OUT/summarize_profile.txt" 2>&1; stamp "summarize rc=$?"
fi
if has bench; then
  timeout 600 python bench.py --out "$OUT/bench_C4.json" > /dev/null 2> "$OUT/bench_C4.err"; stamp "bench C4 rc=$?"
fi
if has configs; then
  for C in C1 C2 C3 T; do
    timeout 600 python bench.py --config $C --out "$OUT/bench_$C.json" > /dev/null 2> "$OUT/bench_$C.err"; stamp "bench $C rc=$?"
  done
fi
if has exchange; then
  timeout 600 python bench.py --force-exchange --partition cloud --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 100000 --out "$OUT/bench_C4_exchange_cloud.json" > /dev/null 2> "$OUT/bench_x1.err"; stamp "exchange cloud rc=$?"
fi
if has exchange; then
  # two ranks sharing the one GPU (gloo group, collectives staged through host memory): cloud shards, the throughput legs' winners by the
  # three reductions on 8-byte keys over the callback (ABI 6)
  SICP_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --partition cloud --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 100000,1000000 --repeats 10 --throughput-repeats 3 --out "$OUT/bench_C4_two_ranks_one_gpu_cloud.json" > /dev/null 2> "$OUT/bench_x2.err"; stamp "two ranks, cloud shards rc=$?"
fi
if has sweeps; then
  timeout 600 python scripts/q_sweep.py 1e7 1000 2048 2049 4096 8192 10000 16384 32768 100000 196608 1000000 > "$OUT/q_sweep.txt" 2>&1; stamp "q_sweep rc=$?"
  timeout 600 python scripts/steady_sweep.py 1e7 32768 100000 500000 1000000 > "$OUT/steady_sweep.txt" 2>&1; stamp "steady_sweep rc=$?"
  timeout 600 python scripts/datasets_run.py > "$OUT/datasets_run.txt" 2>&1; stamp "datasets rc=$?"
  timeout 600 python scripts/cold_match.py > "$OUT/cold_match.txt" 2>&1; stamp "cold match rc=$?"
  for n in 2049 10000 16384; do ./scripts/ubench/reject_trace $n; done > "$OUT/reject_trace_cycles.txt" 2>&1; stamp "reject trace"
fi
if has traces; then
  SICP_SOLVE_TRACE=1 timeout 600 python scripts/trace_c4.py 2>&1 | grep "\[tail\]" | tail -22 > "$OUT/tail_trace.txt"; stamp "tail trace"
  scripts/kernel_timeline.sh c4_$TAG scripts/trace_c4.py > "$OUT/kernel_timeline_c4.txt" 2>&1
  python scripts/iter_timeline.py gpurun_out/kt_c4_$TAG > "$OUT/iter_timeline.txt" 2>&1; stamp "iter timeline"
fi
if has c5; then
  timeout 1200 python bench.py --config C5size --repeats 5 --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --out "$OUT/bench_C5size.json" > /dev/null 2> "$OUT/bench_C5size.err"; stamp "C5size rc=$?"
fi
if has tests; then
  # what the driver runs (the full-size legs marked `slow` are skipped there), then those legs on their own
  timeout ${TESTS_TIMEOUT:-1500} python -m pytest ${TESTS_ARGS:-tests/} -q -m gpu -p no:cacheprovider --durations=25 > "$OUT/pytest_gpu.txt" 2>&1; stamp "pytest -m gpu rc=$?"
  tail -3 "$OUT/pytest_gpu.txt" >> "$OUT/steps.log"
  SICP_TEST_SLOW=1 timeout 900 python -m pytest tests/test_gpu_c5size.py tests/test_gpu_fullsize.py -q -m "gpu and slow" -p no:cacheprovider --durations=10 > "$OUT/pytest_gpu_slow.txt" 2>&1; stamp "pytest slow legs rc=$?"
  tail -3 "$OUT/pytest_gpu_slow.txt" >> "$OUT/steps.log"
fi
cat "$OUT/steps.log"; ls "$OUT"
