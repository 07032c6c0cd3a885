#!/usr/bin/env python3
"""Time the UNMODIFIED reference (/root/reference/python/simpleicp, with oracle/shim/lmfit standing in for the absent
third-party lmfit) on the bench configurations, in the BUILD CONTAINER (the GPU box has no /root/reference).

    python scripts/time_reference.py [C1 C2 C3 C4]      -> profiles/cpu_reference.json

Methodology of the reference's own scripts/benchmark.sh:5-8: the algorithm's time is the `Finished in ... seconds!`
line of python/simpleicp/simpleicp.py:322 (I/O and process start excluded).  C3 / C4 are the synthetic stand-ins of
bench.py; at C4 the normals are injected through the reference's own bypass (simpleicp.py:176: attribute columns
already present) because its estimate_normals alone needs ~10 min at 10 M points, and the run is capped at a few
iterations -- the per-iteration time is what is compared.
"""
import io
import json
import logging
import os
import re
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle" / "shim"))
sys.path.insert(0, "/root/reference/python")

import pandas as pd  # noqa: E402
import simpleicp as ref  # noqa: E402  (the unmodified reference)

import bench  # noqa: E402


def run_ref(Xf, Xm, kwargs, inject=None):
    pc_fix = ref.PointCloud(Xf, columns=["x", "y", "z"])
    pc_mov = ref.PointCloud(Xm.copy(), columns=["x", "y", "z"])
    if inject is not None:
        sel, nv, pl = inject
        for j, c in enumerate(("nx", "ny", "nz")):
            v = np.full(len(pc_fix), np.nan, np.float32)
            v[sel] = nv[:, j]
            pc_fix[c] = pd.arrays.SparseArray(v)
        v = np.full(len(pc_fix), np.nan, np.float32)
        v[sel] = pl
        pc_fix["planarity"] = pd.arrays.SparseArray(v)
    buf = io.StringIO()
    h = logging.StreamHandler(buf)
    h.setFormatter(logging.Formatter("%(message)s"))
    log = logging.getLogger("simpleicp")
    log.setLevel(logging.INFO)
    log.addHandler(h)
    t0 = time.perf_counter()
    try:
        icp = ref.SimpleICP(verbose=False)
        icp.add_point_clouds(pc_fix, pc_mov)
        icp.run(**kwargs)
    finally:
        log.removeHandler(h)
    wall = time.perf_counter() - t0
    text = buf.getvalue()
    fin = float(re.search(r"Finished in ([0-9.]+) seconds", text).group(1))
    # the table has one row per iteration except the one that met the convergence test (simpleicp.py:256-261)
    its = len(re.findall(r"^\s+\d+ \|", text, flags=re.M)) + (1 if "Convergence criteria fulfilled" in text else 0)
    return fin, its, wall


def main():
    names = sys.argv[1:] or ["C1", "C2", "C3", "C4"]
    out_file = ROOT / "profiles" / "cpu_reference.json"
    out = json.loads(out_file.read_text()) if out_file.exists() else {"configs": {}}
    out["measured_on"] = f"build container, {os.cpu_count()} host cores (cKDTree.query workers=-1), {time.strftime('%Y-%m-%d')}"
    out["method"] = "unmodified /root/reference/python/simpleicp + oracle/shim/lmfit; time = its own 'Finished in' line"
    for name in names:
        Xf, Xm, H_true, Q, k, kw, desc = bench.load_workload(name)
        kwargs = dict(correspondences=Q, neighbors=k, **kw)
        inject = None
        if name == "C4":
            from oracle import orc
            sel = np.unique(np.round(np.linspace(0, len(Xf) - 1, Q)).astype(np.int64))
            nn, _ = orc.knn(Xf, Xf[sel], k=k)
            nv, pl = orc.normals(Xf, nn)
            inject = (sel, nv, pl)
            kwargs.update(max_iterations=3, min_change=0.0)
        fin, its, wall = run_ref(Xf, Xm, kwargs, inject)
        out["configs"][name] = {"value": its / fin, "unit": "iterations/s", "seconds": fin, "iterations": its,
                                "cores": os.cpu_count(),
                                "sample": f"SimpleICP.run({', '.join(f'{a}={b}' for a, b in kwargs.items())}) on {desc}"
                                          + ("; normals injected (simpleicp.py:176 bypass)" if inject else "")}
        print(name, out["configs"][name], flush=True)
        out_file.write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
