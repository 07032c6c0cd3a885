"""End-to-end wall time of SimpleICP.run() on the bench workload (10 M-vs-10 M synthetic surface), split
by ABI call: where a caller's time goes once the iterations themselves cost microseconds.
    python scripts/run_profile.py [n_points] [correspondences] [max_overlap_distance]"""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from simpleicp_amd import PointCloud, SimpleICP, _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000
OVERLAP = float(sys.argv[3]) if len(sys.argv) > 3 else np.inf       # max_overlap_distance (finite: the pre-pass runs)
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


for m in ("upload", "upload_columns", "download", "download_columns", "download_both", "transform", "knn", "select_in_range", "estimate_normals", "icp_setup", "icp_run", "icp_iterate",
          "icp_state", "icp_uncertainties"):
    timed(m)

for rep in range(2):                      # second pass = warm (context, allocator, page cache)
    acc.clear()
    t0 = time.perf_counter()
    pc_fix = PointCloud(Xf, columns=["x", "y", "z"])
    pc_mov = PointCloud(Xm, columns=["x", "y", "z"])
    t_pc = time.perf_counter() - t0
    icp = SimpleICP(verbose=False)
    icp.add_point_clouds(pc_fix, pc_mov)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    H, X_t, rbp, res = icp.run(correspondences=Q, max_overlap_distance=OVERLAP)
    pr.disable()
    t_run = time.perf_counter() - t0
    print(f"--- pass {rep}: PointCloud() x2 {t_pc:.3f} s, run() {t_run:.3f} s, {icp.last_run_info['iterations']} iterations, "
          f"|H - H_true| = {np.abs(H - H_true).max():.2e}")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"    ctx.{k:20s} {v * 1e3:10.2f} ms")
    print(f"    {'host (pandas/numpy)':24s} {(t_run - sum(acc.values())) * 1e3:10.2f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
