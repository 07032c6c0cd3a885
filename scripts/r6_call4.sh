#!/usr/bin/env bash
# round 6, call 4: the register-resident one-workgroup rejection (sicp_reject.hip) -- tests, Q sweep, C3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c4; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_operators.py tests/test_gpu_c3shape.py tests/test_gpu_fuzz.py -q -m gpu -x -k "tail_window or q_sweep or icp_run_equals or iteration or rejection or one_launch or operator or c3 or fuzz or mid" -p no:cacheprovider > $O/pytest_reject.txt 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_reject.txt
timeout 300 python scripts/q_sweep.py 1e7 1000 2048 2049 4096 8192 10000 16384 32768 > $O/q_sweep.txt 2>&1; cat $O/q_sweep.txt
timeout 600 python bench.py --config C3 --no-cpu-baseline --no-end-to-end --no-bruteforce-leg --throughput-q 0 --out $O/bench_C3_quick.json > /dev/null 2> $O/bench_C3_quick.err; echo "bench C3 rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6c4/bench_C3_quick.json"))
print("C3", d["value"], d["ms_per_step"], d.get("steady_us_per_step"), d["parity"]["ok"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels_instrumented"].items()})
PY
cat > /tmp/trace_c3.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench
from simpleicp_amd import _lib
Xf, Xm, H_true = bench.synthetic_pair(1_340_000)
sel = np.unique(np.round(np.linspace(0, len(Xf) - 1, 10000)).astype(np.int64))
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
z = np.zeros(6)
for rep in range(3):
    c.icp_setup(sel, nv, pl)
    r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=20, min_change=0.0)
PY
scripts/kernel_timeline.sh c3_r6c4 /tmp/trace_c3.py > $O/kernel_timeline_c3.txt 2>&1
python - <<'PY'
import csv, glob
tr = glob.glob("gpurun_out/kt_c3_r6c4/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("grid_nn", "reject", "lm_all", "postmatch"))]
last = rows[-80:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s start %8.2f dur %7.2f" % (r["Kernel_Name"].split("(")[0][-44:], (s - t0) / 1e3, (e - s) / 1e3))
PY
