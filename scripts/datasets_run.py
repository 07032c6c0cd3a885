"""Wall time of SimpleICP.run() (normals estimated here, not injected) on the bundled data sets, warm, next to the
iteration count and the distance of H from the reference's fixture.   python scripts/datasets_run.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import GOLDEN_CASES, load_golden, load_cloud
from simpleicp_amd import PointCloud, SimpleICP

for name in GOLDEN_CASES:
    g, files, kw = load_golden(name)
    Xf, Xm = load_cloud(files[0]), load_cloud(files[1])
    best = None
    for rep in range(3):
        pc_fix = PointCloud(Xf, columns=["x", "y", "z"])
        pc_mov = PointCloud(Xm.copy(), columns=["x", "y", "z"])
        icp = SimpleICP(verbose=False)
        icp.add_point_clouds(pc_fix, pc_mov)
        t0 = time.perf_counter()
        H, X, rbp, res = icp.run(**kw)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{name:14s} N_fix={len(Xf):7d} N_mov={len(Xm):7d} kwargs={kw}: {best * 1e3:7.2f} ms, "
          f"{icp.last_run_info['iterations']} iterations (reference {int(g['iterations'])}), "
          f"max|H - H_ref| = {np.abs(H - g['H']).max():.1e}", flush=True)
