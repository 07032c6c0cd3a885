#!/usr/bin/env bash
# round 6, call 1: today's baseline (bench default line) + the selection's dynamics
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r6c1
timeout 300 python scripts/sel_dynamics.py 1e7 1000 2048 10000 > gpurun_out/r6c1/sel_dynamics.txt 2>&1; echo "dyn rc $?"
timeout 600 python bench.py --out gpurun_out/r6c1/bench_C4.json > gpurun_out/r6c1/bench_C4.line 2> gpurun_out/r6c1/bench_C4.err; echo "bench rc $?"
timeout 300 python bench.py --config C3 --no-cpu-baseline --out gpurun_out/r6c1/bench_C3.json > /dev/null 2> gpurun_out/r6c1/bench_C3.err; echo "bench C3 rc $?"
cat gpurun_out/r6c1/sel_dynamics.txt | head -80
python - <<'PY'
import json
for n in ("C4", "C3"):
    try:
        d = json.load(open(f"gpurun_out/r6c1/bench_{n}.json"))
        print(n, d["value"], d["ms_per_step"], d.get("steady_us_per_step"))
    except Exception as e: print(n, "failed", e)
PY
