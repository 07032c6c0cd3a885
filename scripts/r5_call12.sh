#!/usr/bin/env bash
# T's many-queries legs under the grid rules / search flavours; the download's fan-out threads at C4
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; O=gpurun_out/r5k; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 150 python bench.py --config T --steps 20 --warmup 3 --out $O/bench_T_$tag.json > $O/bench_T_$tag.line 2> $O/bench_T_$tag.err; echo "T $tag rc $?"; }
run default SICP_DUMMY=1
run avgrule SICP_GRID_POINTWISE=0
run exact16 SICP_NN16=exact
run faronly SICP_NN16=far
run filterall SICP_NN16F_MIN_Q=1
for t in 8 16 32; do
  SICP_DL_THREADS=$t timeout 150 python bench.py --steps 20 --warmup 3 --out $O/bench_C4_dl$t.json > $O/bench_C4_dl$t.line 2> $O/bench_C4_dl$t.err; echo "C4 dl$t rc $?"
done
python - <<'PY'
import json
for tag in ("default", "avgrule", "exact16", "faronly", "filterall"):
    try: d = json.load(open(f"gpurun_out/r5k/bench_T_{tag}.json"))
    except Exception as e: print(tag, "no record", e); continue
    legs = "  ".join(f"{tp[17:] or 'q100000'}: {d[tp]['ms_per_step']:.3f} ms ({d[tp]['roofline']['kernel']}, match {d[tp]['kernels_instrumented']['match']['avg_ms']:.3f}, tallied {d[tp]['roofline']['bytes_alg_per_launch']/1e6:.0f} MB)" for tp in ("throughput_point", "throughput_point_q1000000"))
    print(f"T {tag}: {d['ms_per_step']*1e3:.1f} us/step ({d['roofline_match']['kernel']}) match {d['kernels_instrumented']['match']['avg_ms']*1e3:.1f} us  grid {d['setup']['grid_build_ms']:.2f} ms  {legs}  parity {d['parity']['ok']}")
for t in (8, 16, 32):
    try: d = json.load(open(f"gpurun_out/r5k/bench_C4_dl{t}.json"))
    except Exception as e: print(t, "no record", e); continue
    r = d["run_end_to_end"]
    print(f"C4 download threads {t}: run() {r['seconds']*1e3:.2f} ms / {r['seconds_frame_owns_its_array']*1e3:.2f} ms; {d['ms_per_step']*1e3:.2f} us/step")
PY
