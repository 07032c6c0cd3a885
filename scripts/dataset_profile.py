"""Where the time of SimpleICP.run() goes on a BUNDLED data set (normals estimated here): wall time per ABI call, the kernel classes'
HIP-event times, the searches' own tallies.    python scripts/dataset_profile.py [name ...]      (default: webots)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_golden, load_cloud
from simpleicp_amd import PointCloud, SimpleICP, _lib, backend

acc = {}


def timed(name):
    fn = getattr(_lib.Context, name)

    def wrap(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return fn(self, *a, **k)
        finally:
            acc.setdefault(name, []).append(time.perf_counter() - t0)
    setattr(_lib.Context, name, wrap)


for m in ("upload", "upload_columns", "download", "download_columns", "download_both", "transform", "knn", "select_in_range",
          "estimate_normals", "icp_setup", "icp_run", "icp_iterate", "icp_state", "icp_uncertainties", "set_planarity"):
    if hasattr(_lib.Context, m):
        timed(m)

for name in (sys.argv[1:] or ["webots"]):
    g, files, kw = load_golden(name)
    Xf, Xm = load_cloud(files[0]), load_cloud(files[1])
    print(f"=== {name}: N_fix {len(Xf)} N_mov {len(Xm)} kwargs {kw}", flush=True)
    for rep in range(3):
        acc.clear()
        pc_fix = PointCloud(Xf, columns=["x", "y", "z"])
        pc_mov = PointCloud(Xm.copy(), columns=["x", "y", "z"])
        icp = SimpleICP(verbose=False)
        icp.add_point_clouds(pc_fix, pc_mov)
        ctx = backend.get_context()
        instrument = rep == 2
        if instrument:
            ctx.timing_enable(True, count_work=True); ctx.timing_reset()
        t0 = time.perf_counter()
        H, X, rbp, res = icp.run(**kw)
        dt = time.perf_counter() - t0
        print(f"--- pass {rep}{' (kernel events + tallies on: slower)' if instrument else ''}: run() {dt * 1e3:.2f} ms, "
              f"{icp.last_run_info['iterations']} iterations, max|H - H_ref| = {np.abs(H - g['H']).max():.1e}")
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            print(f"    ctx.{k:18s} {sum(v) * 1e3:9.3f} ms in {len(v)} call(s)")
        print(f"    {'host (pandas/numpy)':22s} {(dt - sum(sum(v) for v in acc.values())) * 1e3:9.3f} ms")
        if instrument:
            tm = ctx.timing()
            for k, v in tm.items():
                if v["launches"]:
                    print(f"    kernels {k:16s} {v['ms']:9.3f} ms in {v['launches']} timed scope(s)")
            print(f"    match tallies {ctx.match_work()}   k-NN tallies {ctx.knn_work()}   last match kernel {ctx.last_match_kernel()}")
            ctx.timing_enable(False)
