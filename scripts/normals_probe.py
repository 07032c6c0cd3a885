"""estimate_normals at large Q: kernel time, work tallies and wall time of the one-sweep k-NN + covariance kernel under a few
settings, next to the k-round search + k_normals (SICP_KNN_SWEEP=0).  Usage: python scripts/normals_probe.py [N] [Q] [k] [variants...]
Prints one JSON line per variant."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from simpleicp_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
VARIANTS = {
    "rounds": {"SICP_KNN_SWEEP": "0"},
    "sweep": {},
    "sweep_g1": {"SICP_KNN_GROUP": "1"},
    "sweep_g4_b4": {"SICP_KNN_GROUP": "4", "SICP_KNN_BATCH": "4"},
    "sweep_g4_b16": {"SICP_KNN_GROUP": "4", "SICP_KNN_BATCH": "16"},
    "sweep_g4_b1": {"SICP_KNN_GROUP": "4", "SICP_KNN_BATCH": "1"},
    "sweep_b1": {"SICP_KNN_BATCH": "1"},
    "sweep_b2": {"SICP_KNN_BATCH": "2"},
    "sweep_b4": {"SICP_KNN_BATCH": "4"},
    "sweep_b8": {"SICP_KNN_BATCH": "8"},
    "sweep_b16": {"SICP_KNN_BATCH": "16"},
    "sweep_b32": {"SICP_KNN_BATCH": "32"},
    "sweep_b64": {"SICP_KNN_BATCH": "64"},
    "sweep_unordered": {"SICP_ORDER_MIN_Q": "0"},
    "sweep_unordered_b1": {"SICP_ORDER_MIN_Q": "0", "SICP_KNN_BATCH": "1"},
}
names = sys.argv[4:] or ["rounds", "sweep", "sweep_b16", "sweep_unordered"]

Xf, _, _ = bench.synthetic_pair(N)
sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
ref = None
for name in names:
    env = VARIANTS[name]
    os.environ.update(env)
    try:
        c = _lib.Context(0)
    finally:
        for key in env:
            del os.environ[key]
    with c:
        c.upload(_lib.FIX, Xf)
        c.estimate_normals(_lib.FIX, sel[:1000], K)             # grid build + warm-up
        walls = []
        for rep in range(3):
            t0 = time.perf_counter()
            nv, pl = c.estimate_normals(_lib.FIX, sel, K)
            walls.append(time.perf_counter() - t0)
        c.timing_enable(True, count_work=True); c.timing_reset()
        c.estimate_normals(_lib.FIX, sel, K)
        work = c.knn_work()
        c.timing_enable(True); c.timing_reset()
        c.estimate_normals(_lib.FIX, sel, K)
        t = c.timing()["knnk_scan"]
        same = None if ref is None else bool(np.array_equal(ref[0], nv, equal_nan=True) and np.array_equal(ref[1], pl, equal_nan=True))
        if ref is None:
            ref = (nv, pl)
        print(json.dumps({"variant": name, "N": N, "Q": len(sel), "k": K, "knn_kernel_ms": t["ms"], "launches": t["launches"],
                          "wall_ms_min": min(walls) * 1e3, "work": work, "cand_per_query": work["candidates"] / len(sel),
                          "sweeps_per_query": work["sweeps"] / len(sel), "same_as_first": same}), flush=True)
