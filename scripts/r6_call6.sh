#!/usr/bin/env bash
# round 6, call 6: record publish without a system fence (A/B), nn16 epilogue below 65 536 queries, k_lm_all's finish on its LDS state
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_operators.py tests/test_gpu_c3shape.py tests/test_gpu_fuzz.py tests/test_gpu_run.py -q -m gpu -x -k "tail_window or q_sweep or icp_run_equals or iteration or rejection or one_launch or operator or c3 or fuzz or barrier or run" -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
scripts/kernel_timeline.sh c4_r6c6 scripts/trace_c4.py > $O/kernel_timeline_c4.txt 2>&1; python scripts/iter_timeline.py gpurun_out/kt_c4_r6c6 > $O/iter_timeline.txt 2>&1; tail -6 $O/iter_timeline.txt; grep "it/s" gpurun_out/kt_c4_r6c6/stderr.txt | tail -2
SICP_LIBRARY=$PWD/simpleicp_amd/_obj/libsimpleicp_hip_pubfence.so scripts/kernel_timeline.sh c4_r6c6_fence scripts/trace_c4.py > $O/kernel_timeline_c4_fence.txt 2>&1; python scripts/iter_timeline.py gpurun_out/kt_c4_r6c6_fence > $O/iter_timeline_fence.txt 2>&1; tail -6 $O/iter_timeline_fence.txt; grep "it/s" gpurun_out/kt_c4_r6c6_fence/stderr.txt | tail -2
for i in 1 2 3; do python scripts/trace_c4.py 2>&1 | grep "it/s" | tail -1; SICP_LIBRARY=$PWD/simpleicp_amd/_obj/libsimpleicp_hip_pubfence.so python scripts/trace_c4.py 2>&1 | grep "it/s" | tail -1; done
timeout 300 python scripts/q_sweep.py 1e7 1000 2048 2049 4096 8192 10000 16384 32768 100000 > $O/q_sweep.txt 2>&1; cat $O/q_sweep.txt
timeout 600 python bench.py --config C3 --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --out $O/bench_C3_quick.json > /dev/null 2> $O/bench_C3_quick.err; echo "bench C3 rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6c6/bench_C3_quick.json"))
print("C3", d["value"], d["ms_per_step"], d.get("steady_us_per_step"), d["parity"]["ok"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels_instrumented"].items()})
PY
for n in 2049 10000 16384; do ./scripts/ubench/reject_trace $n; done > $O/reject_trace.txt 2>&1
