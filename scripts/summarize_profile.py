#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>[_q<Q>]/ (scripts/gpu_profile.sh) into the committed summaries under profiles/<tag>/ and
profiles/latest_pmc.json (per-kernel HBM bytes per launch, read by bench.py).

    python scripts/summarize_profile.py r3            # prof_r3 (default line) + every prof_r3_q<Q> (large-Q legs) + prof_r3_C<config>

HBM bytes per launch = FETCH_SIZE * factor + WRITE_SIZE  (both counters are in KiB).  On gfx950 FETCH_SIZE counts 64 bytes per
128-byte line requested: wide coalesced reads are under-reported by 2x (MI355X_MICROARCH.md, HBM section; `cal` is measured in the
same run on k_aos_to_soa, whose read volume is known: 24 B per point), the grid searches' row gathers by 1.64x (fetch_factor below).  Kernels of a large-Q leg are keyed "<kernel>@Q<Q>".  The file is stamped with the hash of the kernel
sources it was measured on (bench.csrc_hash): bench.py withholds `traffic` when the tree has moved on."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
dst = ROOT / "profiles" / tag
dst.mkdir(parents=True, exist_ok=True)
WATCH = ("grid_nn", "icp_tail", "knn1_f", "lm_eval", "lm_finish", "lm_advance", "lm_all", "k_reject", "k_scatter", "k_cell_ids", "k_cloud_stats",
         "hsel", "keep_stats", "postmatch", "grid_knn", "knn_sweep", "cov_normals", "k_normals", "query_order", "pack_best", "lexmin", "slot_queries", "k_recf", "cell_boxes")


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("sicp::", "")
    return n.split("<")[0]


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: take the most recent run's file."""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=lambda f: Path(f).stat().st_mtime)] if files else []


def counters(src, tagc):
    """(kernel, counter) -> (total, launches) of one PMC pass (aggregated on the GPU box, or the raw dump of older runs)."""
    agg = collections.defaultdict(lambda: [0.0, 0])
    files = newest(str(src / tagc / "**" / "counter_summary.csv"))
    if files:
        for r in csv.DictReader(open(files[0])):
            a = agg[(short(r["Kernel_Name"]), r["Counter_Name"])]
            a[0] += float(r["total"]); a[1] += int(r["launches"])
        return agg
    for f in newest(str(src / tagc / "**" / "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            a = agg[(short(r["Kernel_Name"]), r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg


def one(src, suffix):
    """summaries of one profile directory; returns {kernel: HBM bytes per launch}"""
    sfx = f"_{suffix}" if suffix else ""
    stats = newest(str(src / "trace" / "**" / "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], dst / f"kernel_stats{sfx}.csv")
    if (src / "bench_trace.json").exists():
        # the line bench.py printed under the kernel trace (its timings are the trace's, slower than the bench's own).  Lines written
        # before bench.py stopped doing so priced an untallied pruned search on the brute-force bytes (frac > 1): that field is voided
        text = (src / "bench_trace.json").read_text()
        try:
            rec = json.loads(text.strip().splitlines()[-1])
            for key in ("roofline", "roofline_match"):
                r = rec.get(key) or {}
                if r.get("kernel", "").startswith("k_grid_nn") and r.get("bytes_alg_per_launch") == r.get("bytes_bruteforce_per_launch"):
                    r.update(achieved=None, frac=None, bytes_alg_per_launch=None, pruning_ratio=None,
                             bytes_alg_source="none: no in-kernel tallies in the traced run (--no-work-pass) and no counter summary then")
            text = json.dumps(rec) + "\n"
        except Exception:  # noqa: BLE001
            pass
        (dst / f"bench_under_rocprof{sfx}.json").write_text(text)
    per = collections.defaultdict(dict)
    for tagc, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        agg = {k: v for (k, c), v in counters(src, tagc).items() if c == counter}
        if not agg:
            continue
        with open(dst / f"{tagc}_summary{sfx}.csv", "w") as o:
            o.write("kernel,counter,launches,total_KiB,per_launch_KiB\n")
            for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                o.write(f"{k},{counter},{n},{v:.6g},{v / n:.6g}\n")
                per[k][counter] = v / n * 1024.0
    lines = []
    for tagc in ("pmc_sq1", "pmc_sq2"):
        agg = collections.defaultdict(dict)
        for (k, c), (v, n) in counters(src, tagc).items():
            if any(x in k for x in WATCH):
                agg[k][c] = v / n
        if agg:
            lines.append(f"# {tagc}: rocprofv3 --pmc pass over bench.py (scripts/gpu_profile.sh {src.name[5:]}), per-launch averages")
            for k, d in sorted(agg.items()):
                lines.append(k + " " + str({c: f"{v:.4g}" for c, v in sorted(d.items())}))
    if lines:
        (dst / f"pmc_sq_summary{sfx}.txt").write_text("\n".join(lines) + "\n")
    bench = {}
    try:
        bench = json.loads((src / "bench_trace.json").read_text().strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        pass
    n_pts = bench.get("config", {}).get("n_fixed", 10_000_000)
    cal = 2.0
    if "k_aos_to_soa" in per and per["k_aos_to_soa"].get("FETCH_SIZE"):
        cal = n_pts * 24.0 / per["k_aos_to_soa"]["FETCH_SIZE"]
    return {k: d.get("FETCH_SIZE", 0.0) * fetch_factor(k, cal) + d.get("WRITE_SIZE", 0.0) for k, d in per.items()}, cal


# What FETCH_SIZE counts on gfx950, calibrated with scripts/ubench/gather_calib.hip (profiles/r5/README.md section 4): 64 bytes per
# 128-byte LINE requested.  Streaming kernels request whole lines (x2, measured per run on k_aos_to_soa); the grid searches read rows of
# 8-24 records at random positions -- 2.75 lines but 4.5 64-byte sectors per 256-byte run: x1.64; isolated gathers would be x1.
GRID_SEARCHES = ("grid_nn", "grid_knn", "knn_sweep")


def fetch_factor(kernel, cal):
    return 1.64 if any(g in kernel for g in GRID_SEARCHES) else cal


out = {}
base = ROOT / "gpurun_out" / f"prof_{tag}"
hashes = set()
notes = []
for src in [base] + sorted(ROOT.glob(f"gpurun_out/prof_{tag}_q*")) + sorted(ROOT.glob(f"gpurun_out/prof_{tag}_C*")):
    if not src.exists():
        continue
    suffix = src.name[len(f"prof_{tag}"):].lstrip("_")
    per, cal = one(src, suffix)
    # "q<n>": the default clouds with n correspondences -> "<kernel>@Q<n>";  "C<k>...": another configuration -> "<kernel>@<config>"
    key = "" if not suffix else (f"@Q{suffix[1:]}" if suffix.startswith("q") else f"@{suffix}")
    for k, v in per.items():
        out[k + key] = v
    notes.append(f"{src.name}: cal {cal:.3f}")
    h = src / "csrc_hash.txt"
    if h.exists():
        hashes.add(h.read_text().strip())
out["_note"] = ("HBM bytes per launch = FETCH_SIZE*factor + WRITE_SIZE (KiB counters; factor = cal from k_aos_to_soa of the same run for "
                "streaming kernels -- gfx950 tallies a 128-byte line at 64 bytes --, 1.64 for the grid searches' row gathers: "
                f"profiles/r5/README.md section 4); {'; '.join(notes)}; '<kernel>@Q<n>' = the leg with n correspondences")
out["_csrc_hash"] = hashes.pop() if len(hashes) == 1 else None
(ROOT / "profiles" / "latest_pmc.json").write_text(json.dumps(out, indent=1) + "\n")
(dst / "hbm_bytes_per_launch.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
