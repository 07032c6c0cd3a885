"""The match of an ICP run at large Q, flavour by flavour (environment switches read at context creation), on the bench clouds:
per-iteration kernel time and the search's own tallies for the first iterations from cold, then the steady state.
    python scripts/match_ab.py [n_points] [Q] [variant ...]       variant = name:ENV=V,ENV=V"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from simpleicp_amd import _lib

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
Q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
VARIANTS = sys.argv[3:] or ["exact:SICP_NN16=exact", "exact-noboxes:SICP_NN16=exact,SICP_BOXES=0", "far:SICP_NN16=far", "near:SICP_NN16=near",
                            "near-noboxes:SICP_NN16=near,SICP_BOXES=0", "near8:SICP_NN16=near,SICP_NN_GROUP=8", "near16:SICP_NN16=near,SICP_NN_GROUP=16"]
EARLY = int(os.environ.get("AB_EARLY", "7"))
Xf, Xm, H_true = bench.synthetic_pair(N)
sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
z = np.zeros(6)
nv = pl = None
ref = None
for var in VARIANTS:
    name, _, envs = var.partition(":")
    env = dict(e.split("=") for e in envs.split(",") if e)
    env.setdefault("SICP_NN16F_MIN_Q", "1")          # (the flavour asked for, whatever Q)
    os.environ.update(env)
    try:
        c = _lib.Context(0)
    finally:
        for k in env:
            os.environ.pop(k, None)
    with c:
        c.upload(_lib.FIX, Xf); c.upload(_lib.MOV, Xm)
        if nv is None:
            nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
        c.icp_setup(sel, nv, pl); c.icp_run(z, z, z, 0.3, 1.0, max_iterations=2, min_change=0.0)      # grids, companions, allocations
        rows = []
        for tally in (False, True):
            c.icp_setup(sel, nv, pl)
            x = z.copy()
            for it in range(EARLY):
                c.timing_enable(True, count_work=tally); c.timing_reset()
                R = c.icp_iterate(x, z, z, 0.3, 1.0)
                if tally:
                    w = c.match_work()
                    rows[it] += (w["candidates"] / len(sel), w["rows"] / len(sel), w["deferred"])
                else:
                    rows.append((c.timing()["match"]["ms"],))
                x = np.array(R.x[:])
        c.timing_enable(False)
        idx, dist, keep, _ = c.icp_state(residual=False)
        if ref is None:
            ref = (idx, dist, x.copy())
        same = np.array_equal(idx, ref[0]) and np.array_equal(dist, ref[1]) and np.array_equal(x, ref[2])
        # steady state: 12 iterations from cold, then 30 more, events on
        c.icp_setup(sel, nv, pl)
        r = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=12, min_change=0.0)
        xs = np.array(r[-1].x[:])
        c.timing_enable(True); c.timing_reset()
        c.icp_run(xs, z, z, 0.3, 1.0, max_iterations=30, min_change=0.0)
        tm = c.timing()
        c.timing_enable(True, count_work=True); c.timing_reset()
        c.icp_run(xs, z, z, 0.3, 1.0, max_iterations=30, min_change=0.0)
        w = c.match_work(); c.timing_enable(False)
        print(f"[{name}] N={N} Q={len(sel)} kernel {c.last_match_kernel()} same answers as the first variant: {same}")
        for it, (ms, cand, rws, dfr) in enumerate(rows):
            print(f"    iteration {it}: match {ms * 1e3:9.1f} us   candidates/query {cand:7.1f}   rows/query {rws:5.1f}   left to the exact kernel {dfr}")
        print(f"    steady: match {tm['match']['ms'] / 30 * 1e3:8.1f} us  reject {tm['reject_select']['ms'] / 30 * 1e3:6.1f} us  solve {tm['solve']['ms'] / 30 * 1e3:6.1f} us   "
              f"candidates/query {w['candidates'] / 30 / len(sel):6.1f}   left to the exact kernel per iteration {w['deferred'] / 30:.0f}", flush=True)
