"""Per-iteration match cost of a run's FIRST iterations (from the cold state): time and the search's own tallies, by differencing
runs of 1, 2, ... iterations.    python scripts/early_iters.py [n_points] [Q] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
Xf, Xm, H_true = bench.synthetic_pair(N)
sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
z = np.zeros(6)
c.icp_setup(sel, nv, pl); c.icp_run(z, z, z, 0.3, 1.0, max_iterations=2, min_change=0.0)      # grids, allocations
prev_t, prev_w = 0.0, {"candidates": 0, "rows": 0}
for it in range(1, K + 1):
    c.timing_enable(True); c.timing_reset()
    c.icp_setup(sel, nv, pl); r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=it, min_change=0.0)
    t = c.timing()["match"]["ms"]
    c.timing_enable(True, count_work=True); c.timing_reset()
    c.icp_setup(sel, nv, pl); c.icp_run(z, z, z, 0.3, 1.0, max_iterations=it, min_change=0.0)
    w = c.match_work()
    c.timing_enable(False)
    x = np.array(r[-1].x[:])
    print(f"iteration {it - 1}: match {1e3 * (t - prev_t):9.1f} us   candidates/query {(w['candidates'] - prev_w['candidates']) / len(sel):8.1f}   "
          f"rows/query {(w['rows'] - prev_w['rows']) / len(sel):6.1f}   |H - H_true| after it {np.abs(_lib.params_to_H(x) - H_true).max():.2e}", flush=True)
    prev_t, prev_w = t, w
