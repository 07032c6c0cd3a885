#!/usr/bin/env bash
# round 6, call 13: long rows of the exact search (EXT instantiation) with the next 64 records in flight -- terrestrial stand-in
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_terrestrial.py tests/test_gpu_kernels.py -q -m gpu -x -k "terrestrial or flavours or filtered_scan or knn1 or run_equals" -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
timeout 600 python bench.py --config T --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 100000 --out $O/bench_T.json > /dev/null 2> $O/bench_T.err; echo "bench T rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6c13/bench_T.json"))
tp = d.get("throughput_point")
print("T", d["value"], d["ms_per_step"], d.get("steady_us_per_step"), d["parity"]["ok"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels_instrumented"].items()}, tp and tp["ms_per_step"])
PY
