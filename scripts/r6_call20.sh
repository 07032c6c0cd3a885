#!/usr/bin/env bash
# round 6: what the first (cold) match of a million queries is made of
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/cold
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cold -- python "$GRAFT_REPO_ROOT/scripts/cold_iter0.py" 1e7 1e6 3 > /tmp/prof_cold.log 2>&1
cd "$GRAFT_REPO_ROOT"; tail -3 /tmp/prof_cold.log; find /tmp/prof_cold -name "*.csv" | head
f=$(find /tmp/prof_cold -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/cold/kernel_stats_cold_q1m.csv 2>/dev/null
t=$(find /tmp/prof_cold -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY' > gpurun_out/cold/kernel_trace_cold_q1m.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last repetition: from the last k_gather_queries (icp_setup) on
last = max(i for i, r in enumerate(rows) if "k_gather_queries" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    name = r["Kernel_Name"].split("(")[0][:70]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}  {name}")
PY
head -60 gpurun_out/cold/kernel_trace_cold_q1m.txt
