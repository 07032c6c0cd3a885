#!/usr/bin/env bash
# usage: scripts/kernel_timeline.sh <tag> <python script + args>   -- rocprofv3 kernel trace, prints stats + the last launches' timeline
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/kt_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SCRIPT=$1; shift
[ -f "$REPO/$SCRIPT" ] && SCRIPT="$REPO/$SCRIPT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python "$SCRIPT" "$@" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
st = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0]
for i, r in enumerate(csv.DictReader(open(st))):
    if i < 14:
        print("%-60s calls %5s avg %10.2f us  min %9.2f max %9.2f" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
tr = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-24:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-40s dur %8.2f us  gap %8.2f us" % (r["Kernel_Name"].split("(")[0][-40:], (e - s) / 1e3, ((s - prev) / 1e3) if prev else 0.0))
    prev = e
PY
