#!/usr/bin/env bash
# round 6: HBM traffic and wave statistics of the cold search's launches at Q = 1 M (per dispatch)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/cold
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_cold_$tag -- python "$GRAFT_REPO_ROOT/scripts/cold_iter0.py" 1e7 1e6 2 > /tmp/pmc_cold_$tag.log 2>&1
  f=$(find /tmp/pmc_cold_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' >> "$GRAFT_REPO_ROOT/gpurun_out/cold/pmc_cold_q1m.txt"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][:60])
    by.setdefault(k, {})[r["Counter_Name"]] = by.get(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# the last repetition: after the last k_gather_queries
keys = list(by.keys())
last = max(i for i, k in enumerate(keys) if "k_gather_queries" in k[1])
for k in keys[last:]:
    if "k_grid_nn" in k[1] or "k_hsel" in k[1] or "k_lm_all" in k[1]:
        print(k[0], k[1], " ".join(f"{c}={v:.4g}" for c, v in by[k].items()))
PY
done
cat "$GRAFT_REPO_ROOT/gpurun_out/cold/pmc_cold_q1m.txt"
