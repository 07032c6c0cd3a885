import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench
from simpleicp_amd import _lib
Xf, Xm, H_true = bench.synthetic_pair(10_000_000)
sel = np.unique(np.round(np.linspace(0, len(Xf) - 1, 1000)).astype(np.int64))
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
z = np.zeros(6)
c.icp_setup(sel, nv, pl)
c.icp_run(z, z, z, 0.3, 1.0, max_iterations=20, min_change=0.0)
for rep in range(3):
    c.icp_setup(sel, nv, pl)
    t0 = time.perf_counter()
    r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=20, min_change=0.0)
    dt = time.perf_counter() - t0
    print(f"20 iterations: {dt*1e6:.1f} us -> {dt/20*1e6:.2f} us/it, {20/dt:.0f} it/s", file=sys.stderr)
print(np.abs(_lib.params_to_H(np.array(r[-1].x[:])) - H_true).max(), file=sys.stderr)
