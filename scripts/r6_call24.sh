#!/usr/bin/env bash
# round 6: is C3's run() end to end (28.8 ms in the last record, 8.3 before) a hiccup or the background upload?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/e2e
for i in 1 2 3; do
  timeout 300 python bench.py --config C3 --no-cpu-baseline --no-bruteforce-leg --throughput-q 0 --out gpurun_out/e2e/c3_$i.json > /dev/null 2>&1
  python - gpurun_out/e2e/c3_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["run_end_to_end"]
print({k: round(v, 5) if isinstance(v, float) else v for k, v in r.items() if k != "note"})
PY
done
timeout 300 python scripts/e2e_probe.py 1340000 2>&1 | head -6
