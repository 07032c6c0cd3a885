"""Cost of the per-kernel HIP-event bookkeeping on the bench workload: the same 20 iterations with kernel timing
on and off.   python scripts/timing_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

N, Q = 10_000_000, 1000
Xf, Xm, _ = bench.synthetic_pair(N)
ctx = _lib.Context(0)
ctx.upload(_lib.FIX, Xf)
ctx.upload(_lib.MOV, Xm)
sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
nv, pl = ctx.estimate_normals(_lib.FIX, sel, 10)
ctx.icp_setup(sel, nv, pl)
z = np.zeros(6)
ctx.icp_run(z, z, z, max_iterations=3, min_change=0.0)
for rep in range(3):
    for on in (True, False):
        ctx.timing_enable(on)
        t0 = time.perf_counter()
        ctx.icp_run(z, z, z, max_iterations=20, min_change=0.0)
        dt = (time.perf_counter() - t0) / 20
        print(f"kernel timing {'on ' if on else 'off'}: {dt * 1e6:7.2f} us/iteration", flush=True)
