"""Iteration time vs correspondences on one GPU (20 iterations behind sicp_icp_run from the cold state, 5 repeats, median).
    python scripts/q_sweep.py [n_points] [Q ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
QS = [int(float(a)) for a in sys.argv[2:]] or [1000, 2048, 2049, 10_000, 100_000, 1_000_000]
Xf, Xm, H_true = bench.synthetic_pair(N)
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
z = np.zeros(6)
for Q in QS:
    sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
    nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
    c.icp_setup(sel, nv, pl)
    c.icp_run(z, z, z, 0.3, 1.0, max_iterations=20, min_change=0.0)
    ts = []
    for rep in range(5):
        c.icp_setup(sel, nv, pl)
        t0 = time.perf_counter()
        r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=20, min_change=0.0)
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    c.timing_enable(True); c.timing_reset()
    c.icp_setup(sel, nv, pl)
    r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=20, min_change=0.0)
    tm = c.timing(); c.timing_enable(False)
    ev = sum(x.ne_evals for x in r) / 20
    err = np.abs(_lib.params_to_H(np.array(r[-1].x[:])) - H_true).max()
    print(f"N={N} Q={len(sel):8d}: {dt / 20 * 1e3:8.4f} ms/it  {len(sel) * 20 / dt / 1e6:8.2f} Mcorr/s   match {tm['match']['ms'] / 20 * 1e3:7.1f} us  "
          f"solve {tm['solve']['ms'] / 20 * 1e3:7.1f} us ({ev:.1f} evals)  reject {tm['reject_select']['ms'] / 20 * 1e3:7.1f} us   |H-H_true| {err:.1e}", flush=True)
