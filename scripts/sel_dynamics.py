"""How far median and MAD of the point-to-plane distances move from one iteration to the next (what a windowed selection in the
tail may assume).    python scripts/sel_dynamics.py [n_points] [Q ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
QS = [int(float(a)) for a in sys.argv[2:]] or [1000, 10_000]
Xf, Xm, H_true = bench.synthetic_pair(N)
c = _lib.Context(0)
c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
z = np.zeros(6)
for Q in QS:
    sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
    nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
    c.icp_setup(sel, nv, pl)
    r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=30, min_change=0.0)
    print(f"N={N} Q={len(sel)}")
    pm, pa = None, None
    for i, x in enumerate(r):
        dm = abs(x.median - pm) / x.mad if pm is not None else float("nan")
        da = abs(x.mad - pa) / x.mad if pa is not None else float("nan")
        print(f"  it {i:2d}: n_planar {x.n_planar:6d} kept {x.n_kept:6d} median {x.median:+.6e} mad {x.mad:.6e}  |dmed|/mad {dm:.2e}  |dmad|/mad {da:.2e}  lm_steps {x.lm_steps} evals {x.ne_evals}", flush=True)
        pm, pa = x.median, x.mad
