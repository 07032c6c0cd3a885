"""What the operator-by-operator road costs against the chained loop: the same 20 iterations of the C4 workload
(10 M-vs-10 M, 1000 correspondences) once behind sicp_icp_run and once as sicp_corr_match / sicp_corr_reject_planarity /
sicp_corr_reject_distances / sicp_estimate_parameters per iteration (one host round trip per operator).
    python scripts/operator_cost.py [n_points] [Q]"""
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
K = 20


def chained():
    c.icp_setup(sel, nv, pl)
    t0 = time.perf_counter()
    r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=K, min_change=0.0)
    return time.perf_counter() - t0, np.array(r[-1].x[:])


def operators():
    c.icp_setup(sel, nv, pl)
    x = z.copy()
    t0 = time.perf_counter()
    for _ in range(K):
        c.corr_match(_lib.params_to_H(x))
        c.corr_reject_planarity(0.3, pl, None)
        c.corr_reject_distances()
        x = np.array(c.estimate_parameters(x, z, z, 1.0).x[:])
    return time.perf_counter() - t0, x


for name, fn in (("chained (sicp_icp_run)", chained), ("operator by operator", operators)):
    fn()
    ts, x = [], None
    for _ in range(5):
        dt, x = fn()
        ts.append(dt)
    dt = float(np.median(ts))
    print(f"N={N} Q={len(sel)} {name:24s}: {dt / K * 1e6:8.1f} us per iteration   |H - H_true| {np.abs(_lib.params_to_H(x) - H_true).max():.1e}",
          flush=True)
