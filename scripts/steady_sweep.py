"""Steady-state kernel split at large Q: 12 iterations from cold (untimed), then 30 more from where they ended, HIP events on.
    python scripts/steady_sweep.py [n_points] [Q ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
QS = [int(float(a)) for a in sys.argv[2:]] or [32768, 100_000, 1_000_000]
Xf, Xm, H_true = bench.synthetic_pair(N)
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
z = np.zeros(6)
for Q in QS:
    sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
    nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
    c.icp_setup(sel, nv, pl)
    r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=12, min_change=0.0)
    x = np.array(r[-1].x[:])
    c.timing_enable(True); c.timing_reset()
    r = c.icp_run(x, z, z, 0.3, 1.0, max_iterations=30, min_change=0.0)
    tm = c.timing(); c.timing_enable(False)
    print(f"N={N} Q={len(sel):8d} steady: match {tm['match']['ms'] / 30 * 1e3:7.1f} us  solve {tm['solve']['ms'] / 30 * 1e3:7.1f} us  "
          f"reject {tm['reject_select']['ms'] / 30 * 1e3:7.1f} us", flush=True)
