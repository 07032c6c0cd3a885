"""One cold iteration, several times over (for a kernel trace: which launch of the first match costs what).
    python scripts/cold_iter0.py [n_points] [Q] [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 4
Xf, Xm, H_true = bench.synthetic_pair(N)
sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
z = np.zeros(6)
for rep in range(R):
    c.icp_setup(sel, nv, pl)
    x = z.copy()
    for it in range(3):
        Rr = c.icp_iterate(x, z, z, 0.3, 1.0)
        x = np.array(Rr.x[:])
print("done", c.last_match_kernel())
