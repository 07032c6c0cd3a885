#!/usr/bin/env bash
# round 6, call 14: terrestrial stand-in at Q = 10 000 -- the filtered many-queries search from 8 192 queries (instead of one wave per query up to 65 536)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c14; mkdir -p $O
for v in default f8192 f8192far; do
  E=""; [ $v = f8192 ] && E="SICP_NN16F_MIN_Q=8192"; [ $v = f8192far ] && E="SICP_NN16F_MIN_Q=8192 SICP_NN16=far"
  env $E timeout 600 python bench.py --config T --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --repeats 10 --out $O/bench_T_$v.json > /dev/null 2> $O/bench_T_$v.err; echo "bench T $v rc $?"
done
python - <<'PY'
import json
for v in ("default", "f8192", "f8192far"):
    d = json.load(open(f"gpurun_out/r6c14/bench_T_{v}.json"))
    print(v, round(d["value"]), f'{d["ms_per_step"]*1e3:.1f} us', d.get("steady_us_per_step"), d["parity"]["ok"], d["roofline"]["kernel"], {k: round(x["avg_ms"] * 1e3, 1) for k, x in d["kernels_instrumented"].items()})
PY
