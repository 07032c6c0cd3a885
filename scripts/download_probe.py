"""How fast can the transformed cloud come back?  sequential download + download_columns vs the two from two host threads
(ctypes releases the GIL; experiment only: the context is not thread-safe in general)."""
import sys, time, threading
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Xf, Xm, H = bench.synthetic_pair(N)
with _lib.Context(0) as c:
    c.upload(_lib.MOV, Xm)
    for rep in range(3):
        t0 = time.perf_counter(); a = c.download(_lib.MOV); t1 = time.perf_counter(); cols = c.download_columns(_lib.MOV); t2 = time.perf_counter()
        print(f"sequential: download {1e3 * (t1 - t0):.2f} ms, columns {1e3 * (t2 - t1):.2f} ms, total {1e3 * (t2 - t0):.2f} ms")
    for rep in range(3):
        out = {}
        th = threading.Thread(target=lambda: out.setdefault("a", c.download(_lib.MOV)))
        t0 = time.perf_counter(); th.start(); cols = c.download_columns(_lib.MOV); th.join(); t1 = time.perf_counter()
        ok = np.array_equal(out["a"], a) and all(np.array_equal(u, v) for u, v in zip(cols, (a[:, 0], a[:, 1], a[:, 2])))
        print(f"two threads: {1e3 * (t1 - t0):.2f} ms  equal={ok}")
    for rep in range(2):
        t0 = time.perf_counter(); cs = np.column_stack(cols); t1 = time.perf_counter()
        print(f"host column_stack: {1e3 * (t1 - t0):.2f} ms")
        t0 = time.perf_counter(); c.upload(_lib.MOV, Xm); t1 = time.perf_counter()
        print(f"upload: {1e3 * (t1 - t0):.2f} ms")
    for rep in range(4):
        t0 = time.perf_counter(); rows, cols2 = c.download_both(_lib.MOV); t1 = time.perf_counter()
        print(f"download_both: {1e3 * (t1 - t0):.2f} ms equal={np.array_equal(rows, a)}")
