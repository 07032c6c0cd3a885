#!/usr/bin/env bash
# Run on the GPU box (through gpurun): kernel-trace stats + HBM traffic + SQ counters for bench.py.
# usage: scripts/gpu_profile.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/
#   e.g. scripts/gpu_profile.sh r3                                            the default line's kernels (C4, Q = 1000)
#        scripts/gpu_profile.sh r3_q1000000 --correspondences 1000000         the large-Q kernels at Q = 1 M on the same clouds
# PMC passes are separate runs without any trace domain (gpurun refuses --pmc with sys/hip traces); FETCH_SIZE and
# WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md, PMC slots); those two passes include the brute-force leg (k_knn1_frec's traffic).  PASSES="trace fetch write sq1 sq2" selects.
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
PASSES=${PASSES:-"trace fetch write sq1 sq2"}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd "$REPO" && python -c "import bench; print(bench.csrc_hash())" ) > "$OUT/csrc_hash.txt"
BENCH="python $REPO/bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --no-parity --no-end-to-end --throughput-q 0 $*"
for P in $PASSES; do
  case $P in
    trace) rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --no-work-pass > "$OUT/bench_trace.json" 2> "$OUT/trace.err" ;;
    fetch) rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err" ;;
    write) rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err" ;;
    sq1)   rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq1" -- $BENCH --no-bruteforce-leg > "$OUT/bench_pmc_sq1.json" 2> "$OUT/pmc_sq1.err" ;;
    sq2)   rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $BENCH --no-bruteforce-leg > "$OUT/bench_pmc_sq2.json" 2> "$OUT/pmc_sq2.err" ;;
  esac
done
# the raw per-dispatch dumps are large (gpurun_out/ travels back, 64 MiB cap): keep per-kernel aggregates only
python - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        a = agg[(r["Kernel_Name"], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(os.path.join(os.path.dirname(f), "counter_summary.csv"), "w") as o:
        w = csv.writer(o); w.writerow(["Kernel_Name", "Counter_Name", "launches", "total"])
        for (k, c), (v, n) in sorted(agg.items()):
            w.writerow([k, c, n, repr(v)])
    os.remove(f)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    os.remove(f)
PY
ls "$OUT"/*/* | head
