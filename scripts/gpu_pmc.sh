#!/usr/bin/env bash
# usage: scripts/gpu_pmc.sh <tag> "<counters...>" [bench args]   -- one PMC pass, summarised per kernel
set -u
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "?").split("(")[0][-40:]
            a = agg[k][r.get("Counter_Name", "?")]
            a[0] += float(r.get("Counter_Value", 0)); a[1] += 1
    for k, d in agg.items():
        if not any(x in k for x in ("frec", "fscan", "fixup", "grid_nn", "icp_solve")): continue
        print(k, {c: f"{v/n:.4g}" for c, (v, n) in d.items()})
PY
