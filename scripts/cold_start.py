"""Where a fresh process spends its first second: import, library load, context, first calls.
    python scripts/cold_start.py [n_points]"""
import os, sys, time
t00 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t_np = time.perf_counter()
import simpleicp_amd
from simpleicp_amd import _lib
t_imp = time.perf_counter()
_lib.load()
t_load = time.perf_counter()
ctx = _lib.Context(0)
t_ctx = time.perf_counter()
print(f"numpy {t_np - t00:.3f} s, import simpleicp_amd (+pandas) {t_imp - t_np:.3f} s, dlopen {t_load - t_imp:.3f} s, "
      f"sicp_ctx_create {t_ctx - t_load:.3f} s")
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(0)
X = rng.uniform(0, 1000, (N, 3))
X[:, 2] *= 0.01
q = X[:: N // 1000][:1000] + 0.01


def t(label, fn):
    t0 = time.perf_counter()
    r = fn()
    print(f"  {label:38s} {(time.perf_counter() - t0) * 1e3:9.2f} ms")
    return r


for rep in range(2):
    print(f"pass {rep}:")
    t("upload MOV (n,3)", lambda: ctx.upload(_lib.MOV, X))
    t("upload FIX (n,3)", lambda: ctx.upload(_lib.FIX, X))
    t("knn k=1 (grid build + search)", lambda: ctx.knn(_lib.MOV, q, k=1))
    t("knn k=1 again", lambda: ctx.knn(_lib.MOV, q, k=1))
    sel = np.arange(0, N, N // 1000)[:1000]
    nv, pl = t("estimate_normals (FIX grid build)", lambda: ctx.estimate_normals(_lib.FIX, sel, 10))
    t("icp_setup", lambda: ctx.icp_setup(sel, nv, pl))
    z = np.zeros(6)
    t("icp_run 3 iterations", lambda: ctx.icp_run(z, z, z, max_iterations=3, min_change=0.0))
    t("transform + download", lambda: (ctx.transform(_lib.MOV, np.eye(4)), ctx.download(_lib.MOV)))
