#!/usr/bin/env bash
# round 6, run() end to end: link rates by piece size, the ring download + background upload in run()
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/e2e
timeout 120 scripts/ubench/d2h_rate > gpurun_out/e2e/d2h_rate.txt 2>&1; cat gpurun_out/e2e/d2h_rate.txt
timeout 600 python scripts/e2e_probe.py > gpurun_out/e2e/e2e_probe_bg_upload.txt 2>&1; head -6 gpurun_out/e2e/e2e_probe_bg_upload.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "download or transform or run or upload or nonfinite or non_finite or nan" -p no:cacheprovider 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/e2e/bench_bg_upload.json 2> gpurun_out/e2e/bench_bg_upload.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/e2e/bench_bg_upload.json'))
print(d['value'], d['ms_per_step'], d.get('run_end_to_end'))
PY
