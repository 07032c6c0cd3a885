"""ms per ICP iteration against the number of correspondences Q (C3/C4 stand-ins), with the per-kernel split the
library's own HIP events report.   python scripts/q_sweep.py [n_points] [Q ...]"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_340_000
QS = [int(float(a)) for a in sys.argv[2:]] or [1000, 2048, 2049, 4096, 10_000, 30_000, 100_000]
Xf, Xm, _ = bench.synthetic_pair(N)
ctx = _lib.Context(0)
ctx.upload(_lib.FIX, Xf)
ctx.upload(_lib.MOV, Xm)
z = np.zeros(6)
for Q in QS:
    sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
    nv, pl = ctx.estimate_normals(_lib.FIX, sel, 10)
    ctx.icp_setup(sel, nv, pl)
    ctx.icp_run(z, z, z, max_iterations=3, min_change=0.0)
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    res = ctx.icp_run(z, z, z, max_iterations=20, min_change=0.0)
    dt = (time.perf_counter() - t0) / 20
    tm = ctx.timing()
    ctx.timing_enable(False)
    split = ", ".join(f"{k} {v['ms'] / 20 * 1e3:.0f} us/{v['launches'] / 20:.1f}x" for k, v in tm.items() if v["launches"])
    print(f"N={N} Q={len(sel):7d}: {dt * 1e3:8.3f} ms/iteration  ({len(sel) / dt / 1e6:7.2f} M corr/s)  evals/it "
          f"{sum(r.ne_evals for r in res) / 20:.2f}  [{split}]", flush=True)
