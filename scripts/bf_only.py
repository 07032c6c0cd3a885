"""Brute-force (filtered scan) iterations only, for profiling one flavour:  SICP_FSCAN=record|inline python scripts/bf_only.py [n] [Q]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000
Xf, Xm, H_true = bench.synthetic_pair(N)
os.environ["SICP_KNN1"] = "filter"
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
c.icp_setup(sel, nv, pl)
z = np.zeros(6)
c.icp_run(z, z, z, 0.3, 1.0, max_iterations=2, min_change=0.0)
c.timing_enable(True); c.timing_reset()
t0 = time.perf_counter()
c.icp_run(z, z, z, 0.3, 1.0, max_iterations=6, min_change=0.0)
dt = time.perf_counter() - t0
tm = c.timing()["match"]
print(c.last_match_kernel(), "avg ms", tm["ms"] / tm["launches"], "pairs/s", N * len(sel) / (tm["ms"] / tm["launches"] * 1e-3), file=sys.stderr)
