"""Where does the terrestrial stand-in's match spend its time?  The fixed cloud's points sorted by their distance from the scanner, ten
deciles of 1000 queries each: per decile the steady match time of a chained run, the search's own tallies and the slowest query's row /
candidate counts would show whether a few queries at the scanner's feet hold the kernel up.   python scripts/t_deciles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

Xf, Xm, H_true = bench.terrestrial_pair(1_250_000)
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
r = np.linalg.norm(Xf, axis=1)
order = np.argsort(r)
z = np.zeros(6)
x0 = None
for dec in range(10):
    lo, hi = dec * len(Xf) // 10, (dec + 1) * len(Xf) // 10
    sel = np.sort(order[lo:hi][np.round(np.linspace(0, hi - lo - 1, 1000)).astype(np.int64)])
    nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
    c.icp_setup(sel, nv, pl)
    c.icp_run(z, z, z, 0.3, 1.0, max_iterations=12, min_change=0.0)          # settle (bounds from previous matches)
    c.timing_enable(True, count_work=True); c.timing_reset()
    c.icp_setup(sel, nv, pl)
    R = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=10, min_change=0.0)           # (from cold: 10 iterations, the first two far)
    t = c.timing(); w = c.match_work(); c.timing_enable(False)
    n = max(1, t["match"]["launches"])
    print(f"decile {dec}: r = {r[order[lo]]:6.2f} .. {r[order[hi - 1]]:6.2f} m   match {t['match']['ms'] / n * 1e3:7.1f} us per launch   "
          f"candidates/query {w['candidates'] / (1000 * n):8.1f}   rows/query {w['rows'] / (1000 * n):6.1f}   kept {R[-1].n_kept}", flush=True)
