"""Why SimpleICP.run() costs what it costs inside bench.py's process: the same call as bench.end_to_end, several passes, with
the time inside each ABI call, (a) alone in the process, (b) on a fresh copy of the movable array per pass as bench.py does.  python scripts/e2e_probe.py [n_points]"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from simpleicp_amd import PointCloud, SimpleICP, _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Xf, Xm, H_true = bench.synthetic_pair(N)
acc = {}


def timed(name):
    fn = getattr(_lib.Context, name)

    def wrap(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return fn(self, *a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    setattr(_lib.Context, name, wrap)


for m in ("upload", "upload_columns", "upload_start", "upload_wait", "download_both", "transform", "estimate_normals", "icp_setup", "icp_run", "icp_state", "icp_uncertainties"):
    timed(m)


def passes(label, n, copy):
    for rep in range(n):
        acc.clear()
        pc_fix = PointCloud(Xf, columns=["x", "y", "z"])
        pc_mov = PointCloud(Xm.copy() if copy else Xm, columns=["x", "y", "z"])
        icp = SimpleICP(verbose=False)
        icp.add_point_clouds(pc_fix, pc_mov)
        t0 = time.perf_counter()
        icp.run(correspondences=1000)
        dt = time.perf_counter() - t0
        parts = " ".join(f"{k}={v * 1e3:.1f}" for k, v in sorted(acc.items(), key=lambda kv: -kv[1]))
        print(f"{label} pass {rep}: run() {dt * 1e3:.1f} ms  host {1e3 * (dt - sum(acc.values())):.1f}  | {parts}", flush=True)


passes("alone", 3, False)
passes("alone+copy", 2, True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
passes("alone+copy, profiled", 1, True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
