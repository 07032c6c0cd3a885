#!/usr/bin/env bash
# Run on the GPU box (through gpurun): kernel-trace stats + HBM traffic counters for bench.py.
# usage: scripts/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err"
find "$OUT" -name '*.csv' | head -50
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("==", f)
    print(open(f).read()[:3000])
for tag in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(out + f"/{tag}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = (r.get("Kernel_Name", "?")[:60], r.get("Counter_Name", "?"))
                agg[k][0] += float(r.get("Counter_Value", 0)); agg[k][1] += 1
        print("==", f)
        for (k, c), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
            print(f"{k:60s} {c:12s} total={v:.4g} launches={n} per_launch={v/n:.4g}")
PY
