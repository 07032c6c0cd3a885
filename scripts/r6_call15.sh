#!/usr/bin/env bash
# round 6, call 15: nonuniform grids with their cells' records in x order, long rows trimmed by 64 probes (k_grid_nn<EXT>, NN_XSORTED)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_terrestrial.py tests/test_gpu_kernels.py tests/test_gpu_run.py -q -m gpu -x -k "terrestrial or flavours or filtered_scan or knn1 or select_in_range or run" -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
timeout 600 python bench.py --config T --no-cpu-baseline --no-bruteforce-leg --throughput-q 100000,1000000 --out $O/bench_T.json > /dev/null 2> $O/bench_T.err; echo "bench T rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6c15/bench_T.json"))
tp = d.get("throughput_point"); tp2 = d.get("throughput_point_q1000000")
print("T", round(d["value"]), f'{d["ms_per_step"]*1e3:.1f} us', d.get("steady_us_per_step"), d["parity"]["ok"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels_instrumented"].items()},
      "cand/q", d["roofline"].get("candidates_per_query"), "tp", tp and (tp["ms_per_step"], tp["parity"]["ok"]), tp2 and (tp2["ms_per_step"], tp2["parity"]["ok"]), "e2e", (d.get("run_end_to_end") or {}).get("seconds"), "setup", d.get("setup"))
PY
timeout 300 python scripts/t_deciles.py > $O/t_deciles.txt 2>&1; cat $O/t_deciles.txt
timeout 300 python scripts/datasets_run.py > $O/datasets_run.txt 2>&1; tail -8 $O/datasets_run.txt
