#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (scripts/gpu_profile.sh) into the committed summaries under
profiles/<tag>/ and profiles/latest_pmc.json (per-kernel HBM bytes per launch, read by bench.py).

HBM bytes per launch = FETCH_SIZE * cal + WRITE_SIZE  (both counters are in KiB).  On gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section); `cal` is
measured in the same run on k_aos_to_soa, whose read volume is known (24 B per point)."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles" / tag
dst.mkdir(parents=True, exist_ok=True)


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("sicp::", "")
    return n.split("<")[0]


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: take the most recent run's file."""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=lambda f: Path(f).stat().st_mtime)] if files else []


stats = newest(str(src / "trace" / "**" / "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], dst / "kernel_stats.csv")
for f in ("bench_trace.json",):
    if (src / f).exists():
        shutil.copy(src / f, dst / "bench_under_rocprof.json")

per = collections.defaultdict(dict)
for tagc, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = newest(str(src / tagc / "**" / "*counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(dst / f"{tagc}_summary.csv", "w") as o:
        o.write("kernel,counter,launches,total_KiB,per_launch_KiB\n")
        for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            o.write(f"{k},{counter},{n},{v:.6g},{v / n:.6g}\n")
            per[k][counter] = v / n * 1024.0

# SQ counter passes -> one text summary (per-launch averages of the kernels of the default path and the brute-force leg)
lines = []
for tagc in ("pmc_sq1", "pmc_sq2"):
    files = newest(str(src / tagc / "**" / "*counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        if not any(x in k for x in ("grid_nn", "icp_tail", "knn1_f", "lm_eval", "lm_finish", "k_reject", "k_scatter", "k_cell_ids", "k_cloud_stats")):
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    lines.append(f"# {tagc}: rocprofv3 --pmc pass over bench.py (scripts/gpu_profile.sh), per-launch averages")
    for k, d in sorted(agg.items()):
        lines.append(k + " " + str({c: f"{v / n:.4g}" for c, (v, n) in sorted(d.items())}))
if lines:
    (dst / "pmc_sq_summary.txt").write_text("\n".join(lines) + "\n")

bench = {}
try:
    bench = json.loads((src / "bench_trace.json").read_text().strip().splitlines()[-1])
except Exception:
    pass
n_pts = bench.get("config", {}).get("n_fixed", 10_000_000)
cal = 2.0
if "k_aos_to_soa" in per and per["k_aos_to_soa"].get("FETCH_SIZE"):
    cal = n_pts * 24.0 / per["k_aos_to_soa"]["FETCH_SIZE"]
out = {"_note": f"HBM bytes per launch = FETCH_SIZE*{cal:.3f} + WRITE_SIZE (KiB counters); cal from k_aos_to_soa; profile tag {tag}"}
for k, d in per.items():
    out[k] = d.get("FETCH_SIZE", 0.0) * cal + d.get("WRITE_SIZE", 0.0)
(ROOT / "profiles" / "latest_pmc.json").write_text(json.dumps(out, indent=1))
(dst / "hbm_bytes_per_launch.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
