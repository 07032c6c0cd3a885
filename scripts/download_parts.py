"""Where sicp_cloud_download_both spends its time (SICP_SOLVE_TRACE=host prints the split): fresh destination arrays as run() hands
them over, the same arrays a second time (pages touched), and how much of a fresh array the kernel backed with huge pages.
python scripts/download_parts.py [n_points]"""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["SICP_SOLVE_TRACE"] = "host"
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000


def huge_kb():
    for line in open("/proc/self/smaps_rollup"):
        if line.startswith("AnonHugePages"):
            return int(line.split()[1])
    return -1


Xf, Xm, H = bench.synthetic_pair(N)
with _lib.Context(0) as c:
    c.upload(_lib.MOV, Xm)
    L, h = c._L, c._h
    keep = []
    for rep in range(4):
        h0 = huge_kb()
        out = np.empty((N, 3)); cols = [np.empty(N) for _ in range(3)]
        t0 = time.perf_counter()
        c._chk(L.sicp_cloud_download_both(h, _lib.MOV, _lib._ptr(out), *[_lib._ptr(v) for v in cols]))
        t1 = time.perf_counter()
        c._chk(L.sicp_cloud_download_both(h, _lib.MOV, _lib._ptr(out), *[_lib._ptr(v) for v in cols]))
        t2 = time.perf_counter()
        print(f"rep {rep}: fresh arrays {1e3 * (t1 - t0):.2f} ms, the same arrays again {1e3 * (t2 - t1):.2f} ms; huge pages +{(huge_kb() - h0) / 1024:.0f} MiB of 458", flush=True)
        if rep < 2:
            keep.append((out, cols))          # (reps 2, 3 free theirs: the allocator may hand the same addresses out again)
        t0 = time.perf_counter(); del out, cols; print(f"        freeing them {1e3 * (time.perf_counter() - t0):.2f} ms")
    for rows_only in (True, False):
        out = np.empty((N, 3)); cols = [np.empty(N) for _ in range(3)]
        t0 = time.perf_counter()
        if rows_only:
            c._chk(L.sicp_cloud_download_both(h, _lib.MOV, _lib._ptr(out), None, None, None))
        else:
            c._chk(L.sicp_cloud_download_both(h, _lib.MOV, None, *[_lib._ptr(v) for v in cols]))
        print(f"{'rows' if rows_only else 'columns'} only, fresh: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
        keep.append((out, cols))
