#!/usr/bin/env python3
"""Per-iteration (match, tail) kernel durations of the LAST run in a rocprofv3 kernel trace made by scripts/kernel_timeline.sh.
An iteration = the grid-search launches up to a tail launch (a cold iteration has two: the subsample's bound, then the search)."""
import csv, glob, sys
tr = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "grid_nn" in r["Kernel_Name"] or "icp_tail" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
its, cur = [], []
for r in rows:
    cur.append(r)
    if "icp_tail" in r["Kernel_Name"]:
        its.append(cur); cur = []
last = its[-n:]
t0 = int(last[0][0]["Start_Timestamp"])
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for i, it in enumerate(last):
    m = [dur(r) for r in it[:-1]]
    print(f"it {i:2d}: match {sum(m):6.2f} us{' (' + ' + '.join(f'{x:.2f}' for x in m) + ')' if len(m) > 1 else ''}  tail {dur(it[-1]):6.2f} us   "
          f"end at {(int(it[-1]['End_Timestamp']) - t0) / 1e3:8.2f} us")
