#!/usr/bin/env bash
# the operator replay after the oracle's tiny calls stopped waking a 256-thread team; the cell table's limit for nonuniform clouds
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; O=gpurun_out/r5j; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_operators.py -q -x --durations=6 > $O/pytest_operators.txt 2>&1; echo "operators rc $?"; tail -12 $O/pytest_operators.txt
for k in 27 28 29 30; do
  SICP_GRID_CAP_NONUNIFORM=$k timeout 120 python bench.py --config T --steps 20 --warmup 3 --out $O/bench_T_cap$k.json > $O/bench_T_cap$k.line 2> $O/bench_T_cap$k.err; echo "cap $k rc $?"
done
SICP_GRID_CAP_NONUNIFORM=30 timeout 200 python -m pytest tests/test_gpu_terrestrial.py -q -x > $O/pytest_T_cap30.txt 2>&1; echo "T tests cap30 rc $?"; tail -3 $O/pytest_T_cap30.txt
python - <<'PY'
import json
for k in (27, 28, 29, 30):
    try:
        d = json.load(open(f"gpurun_out/r5j/bench_T_cap{k}.json"))
    except Exception as e:
        print(k, "no record", e); continue
    m = d["kernels_instrumented"]["match"]["avg_ms"]; s = d["setup"]
    print(f"cap 2^{k}: {d['ms_per_step']*1e3:.1f} us/step  steady {d.get('steady_us_per_step')}  match {m*1e3:.1f} us  grid build {s.get('grid_build_ms'):.2f} ms  run() {d['run_end_to_end']['seconds']*1e3:.2f} ms"
          f"  Q=100k {d['throughput_point']['ms_per_step']:.3f} ms  Q=1M {d['throughput_point_q1000000']['ms_per_step']:.3f} ms  parity {d['parity']['ok']}")
PY
