#!/usr/bin/env bash
# round 6, call 3: quickselect window pick (tail + k_reject) -- tests, tail trace, record-placement experiment, C4/C3 quick
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "tail_window or q_sweep or icp_run_equals or iteration_vs_oracle or rejection" -p no:cacheprovider > $O/pytest_tail.txt 2>&1; echo "pytest tail rc $?"; tail -4 $O/pytest_tail.txt
SICP_SOLVE_TRACE=1 timeout 300 python scripts/trace_c4.py 2>&1 | grep -E "\[tail\]|iterations" | tail -24 > $O/tail_trace.txt; tail -8 $O/tail_trace.txt
scripts/kernel_timeline.sh c4_r6c3 scripts/trace_c4.py > $O/kernel_timeline_c4.txt 2>&1; python scripts/iter_timeline.py gpurun_out/kt_c4_r6c3 > $O/iter_timeline.txt 2>&1; cat $O/iter_timeline.txt
SICP_EXP_REC_DEVICE=1 scripts/kernel_timeline.sh c4_r6c3_recdev scripts/trace_c4.py > $O/kernel_timeline_c4_recdev.txt 2>&1; python scripts/iter_timeline.py gpurun_out/kt_c4_r6c3_recdev > $O/iter_timeline_recdev.txt 2>&1; cat $O/iter_timeline_recdev.txt
for C in C4 C3; do
timeout 600 python bench.py --config $C --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --out $O/bench_${C}_quick.json > /dev/null 2> $O/bench_${C}_quick.err; echo "bench $C rc $?"
done
python - <<'PY'
import json
for C in ("C4", "C3"):
    d = json.load(open(f"gpurun_out/r6c3/bench_{C}_quick.json"))
    print(C, d["value"], d["ms_per_step"], d.get("steady_us_per_step"), d["parity"]["ok"])
PY
timeout 300 python scripts/q_sweep.py 1e7 1000 2048 2049 4096 10000 16384 > $O/q_sweep.txt 2>&1; cat $O/q_sweep.txt
timeout 900 python -m pytest tests/test_gpu_exchange.py -q -m gpu -x -p no:cacheprovider -k "disjoint or key_exchange" > $O/pytest_xchg.txt 2>&1; echo "pytest exchange rc $?"; tail -5 $O/pytest_xchg.txt
