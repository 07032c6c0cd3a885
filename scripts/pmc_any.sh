#!/usr/bin/env bash
# One rocprofv3 counter pass over an arbitrary command, aggregated per kernel (per-launch averages).
# usage (on the GPU box): scripts/pmc_any.sh <tag> "<counters>" <command...>   -> gpurun_out/pmc_<tag>.txt
set -u
TAG=$1; CNT=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d "$OUT" -- "$@" > "$OUT/cmd.out" 2> "$OUT/cmd.err"
python - "$OUT" > "$REPO/gpurun_out/pmc_$TAG.txt" <<'PY'
import collections, csv, glob, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sicp::", "")
        if os.environ.get("PMC_BY_GRID"):
            k += "@grid" + r.get("Grid_Size", "?")                  # (one kernel launched at several sizes: keep them apart)
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(agg.items()):
    print(k, " ".join(f"{c}={v / n:.6g}(x{n})" for c, (v, n) in sorted(cs.items())))
PY
rm -rf "$OUT"/*/  # raw counter dumps are large; the summary is what travels back
