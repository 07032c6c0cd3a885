#!/usr/bin/env python3
"""Per-iteration (match, tail) kernel durations of the LAST run in a rocprofv3 kernel trace made by scripts/kernel_timeline.sh."""
import csv, glob, sys
tr = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "grid_nn" in r["Kernel_Name"] or "icp_tail" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
last = rows[-2 * n:]
t0 = int(last[0]["Start_Timestamp"])
for i in range(0, 2 * n, 2):
    a, b = last[i], last[i + 1]
    da = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    db = (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3
    print(f"it {i // 2:2d}: match {da:6.2f} us  tail {db:6.2f} us   end at {(int(b['End_Timestamp']) - t0) / 1e3:8.2f} us")
