"""Where does the float32-filtered many-queries search overtake the exact four-per-wave kernel?  Iteration time from cold (20
iterations behind sicp_icp_run, median of 5) and the steady match, per correspondence count and flavour.
    python scripts/q_cross.py [n_points] [Q ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
QS = [int(float(a)) for a in sys.argv[2:]] or [16384, 32768, 65536, 131072, 262144, 524288]
Xf, Xm, H_true = bench.synthetic_pair(N)
z = np.zeros(6)
ctxs = {}
for mode, env in (("wave", {"SICP_NN16_MIN_Q": "1000000000"}), ("exact", {"SICP_NN16": "exact", "SICP_NN16_MIN_Q": "1"}),
                  ("near", {"SICP_NN16": "near", "SICP_NN16_MIN_Q": "1", "SICP_NN16F_MIN_Q": "1"})):
    os.environ.update(env)
    try:
        c = _lib.Context(0)
    finally:
        for k in env:
            os.environ.pop(k, None)
    c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
    ctxs[mode] = c
for Q in QS:
    sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
    nv, pl = ctxs["exact"].estimate_normals(_lib.FIX, sel, 10)
    for mode, c in ctxs.items():
        c.icp_setup(sel, nv, pl)
        c.icp_run(z, z, z, 0.3, 1.0, max_iterations=20, min_change=0.0)
        ts = []
        for rep in range(5):
            c.icp_setup(sel, nv, pl)
            t0 = time.perf_counter()
            r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=20, min_change=0.0)
            ts.append(time.perf_counter() - t0)
        xs = np.array(r[-1].x[:])
        c.timing_enable(True); c.timing_reset()
        c.icp_run(xs, z, z, 0.3, 1.0, max_iterations=30, min_change=0.0)
        tm = c.timing(); c.timing_enable(False)
        print(f"N={N} Q={len(sel):8d} {mode:6s} {c.last_match_kernel():13s}: {float(np.median(ts)) / 20 * 1e3:8.4f} ms/it from cold   steady match {tm['match']['ms'] / 30 * 1e3:7.1f} us  "
              f"reject {tm['reject_select']['ms'] / 30 * 1e3:6.1f} us  solve {tm['solve']['ms'] / 30 * 1e3:6.1f} us", flush=True)
