#!/usr/bin/env bash
# round 6, run() end to end: packed chunks (one piece per chunk), background upload tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/e2e
timeout 300 python scripts/download_parts.py > gpurun_out/e2e/download_parts_streamed.txt 2>&1; cat gpurun_out/e2e/download_parts_packed.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "download or transform or run or upload or finite" -p no:cacheprovider 2>&1 | tail -3
timeout 600 python scripts/e2e_probe.py > gpurun_out/e2e/e2e_probe_streamed.txt 2>&1; head -3 gpurun_out/e2e/e2e_probe_packed.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/e2e/bench_streamed.json 2> gpurun_out/e2e/bench_streamed.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/e2e/bench_streamed.json'))
print(d['value'], d['ms_per_step'], d.get('run_end_to_end'))
PY
