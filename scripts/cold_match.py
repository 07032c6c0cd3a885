"""The FIRST iteration's match (no previous match to bound the search): time and the search's own work, C4 sizes.
    python scripts/cold_match.py [n_points] [Q]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000
Xf, Xm, H_true = bench.synthetic_pair(N)
sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
z = np.zeros(6)
for it in (1, 2):
    c.icp_setup(sel, nv, pl); c.icp_run(z, z, z, 0.3, 1.0, max_iterations=it, min_change=0.0)
    c.timing_enable(True, count_work=True); c.timing_reset()
    c.icp_setup(sel, nv, pl); c.icp_run(z, z, z, 0.3, 1.0, max_iterations=it, min_change=0.0)
    w = c.match_work(); tm = c.timing()
    c.timing_enable(True); c.timing_reset()
    c.icp_setup(sel, nv, pl); c.icp_run(z, z, z, 0.3, 1.0, max_iterations=it, min_change=0.0)
    tm = c.timing(); c.timing_enable(False)
    print(f"first {it} iteration(s): match {tm['match']['ms'] * 1e3:.1f} us total, candidates/query {w['candidates'] / len(sel):.0f}, "
          f"rows/query {w['rows'] / len(sel):.1f} (summed over the launches)")
