"""How much does the REFERENCE's own result depend on things its normals leave open -- the arbitrary sign of each
normal, and which of several equidistant points cKDTree happens to return as the k-th neighbour?

    python oracle/normal_sensitivity.py [--patterns N] [--cases dragon,bunny,...] [--out FILE]

TEST INFRASTRUCTURE (a fixture generator like make_golden.py, which is why it lives under oracle/): build container only --
it imports the unmodified reference from /root/reference with oracle/shim/lmfit.

`estimate_normals` takes each normal from `np.linalg.eig` (pointcloud.py:192-198): its sign is whatever LAPACK returns.
The point-to-plane distance d = (p2 - p1).n flips with it, and the rejection step is built on the SIGNED median and the MAD
about it (corrpts.py:165-188), so the kept set -- and through it H and the iteration count -- depends on the signs.
This script measures by how much: for each of the reference's own test configurations it re-runs the unmodified
reference with ITS normals (the fixture's, injected through the reference's own bypass, simpleicp.py:176) re-signed under

    * "lapack"      the signs LAPACK produced (= the committed fixture; must reproduce its H bit for bit),
    * "convention"  this package's deterministic rule: the component of largest magnitude is positive
                    (oracle/sicp_oracle.c orc_normals, csrc k_normals),
    * "random<i>"   N seeded random patterns (each normal flipped with probability 1/2),
    * "oracle"      not a re-signing: normals and planarity recomputed by the ORACLE (brute-force k-NN with the
                    deterministic (d2, index) tie rule, covariance + Jacobi eigen step, sign convention) -- what the HIP
                    path's own estimate_normals produces to one float32 ulp.  On quantised clouds (Webots, Multisensor)
                    many neighbours are exactly equidistant, cKDTree's pick among them is arbitrary, and the normals
                    differ by more than their sign,

and records H, the iteration count and the final correspondence count of every run.  The spread over the patterns is
what a comparison "own normals vs the reference's H" can be held to (tests/test_gpu_run.py::OWN_NORMALS_TOL derives its
tolerance from the committed output, tests/golden/normal_sensitivity.json); the "oracle" run is the tight pin: the
unmodified reference, fed the normals the HIP path computes, must land where the HIP path lands.
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
DEFAULT_OUT = ROOT / "tests" / "golden" / "normal_sensitivity.json"
CASES = ["dragon", "bunny", "webots", "multisensor"]


def reference():
    for p in (str(ROOT / "oracle" / "shim"), str(REF / "python")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import simpleicp as ref
    return ref


def convention_signs(normals):
    """+1 / -1 per normal so that the component of largest magnitude (first on ties) becomes positive."""
    j = np.argmax(np.abs(normals), axis=1)
    lead = normals[np.arange(len(normals)), j]
    return np.where(lead < 0, -1.0, 1.0).astype(np.float32)


def signs_for(pattern, normals):
    if pattern == "lapack":
        return np.ones(len(normals), np.float32)
    if pattern == "convention":
        return convention_signs(normals)
    seed = int(pattern[len("random"):])
    return np.where(np.random.default_rng(1000 + seed).random(len(normals)) < 0.5, -1.0, 1.0).astype(np.float32)


def oracle_normals(g, kwargs, Xf):
    """estimate_normals as the oracle (and, to a float32 ulp, the HIP path) does it, for the fixture's selected rows."""
    from oracle import orc
    sel = g["sel_idx"]
    nn, _ = orc.knn(Xf, Xf[sel], k=int(kwargs.get("neighbors", 10)))
    return orc.normals(Xf, nn)


def run_reference(ref, g, kwargs, Xf, Xm, normals, planarity):
    """The unmodified reference on (Xf, Xm) with the given per-selected-row normals injected (simpleicp.py:176)."""
    pc_fix = ref.PointCloud(Xf, columns=["x", "y", "z"])
    pc_mov = ref.PointCloud(Xm.copy(), columns=["x", "y", "z"])
    sel = g["sel_idx"]
    nrm = normals
    for j, c in enumerate(("nx", "ny", "nz")):
        v = np.full(len(Xf), np.nan, np.float32)
        v[sel] = nrm[:, j]
        pc_fix[c] = pd.arrays.SparseArray(v)
    v = np.full(len(Xf), np.nan, np.float32)
    v[sel] = planarity
    pc_fix["planarity"] = pd.arrays.SparseArray(v)
    counts = []
    orig = ref.corrpts.CorrPts.reject_wrt_point_to_plane_distances

    def rd(self):
        orig(self)
        counts.append(len(self._df))
    ref.corrpts.CorrPts.reject_wrt_point_to_plane_distances = rd
    try:
        icp = ref.SimpleICP(verbose=False)
        icp.add_point_clouds(pc_fix, pc_mov)
        H, _, rbp, res = icp.run(**kwargs)
    finally:
        ref.corrpts.CorrPts.reject_wrt_point_to_plane_distances = orig
    assert np.array_equal(pc_fix.idx_selected, sel)
    return {"H": H.tolist(), "iterations": len(counts), "final_correspondences": counts[-1],
            "x": [float(v) for v in rbp.get_parameter_attributes_as_list("estimated_value")]}


def measure(case, patterns):
    from conftest import load_cloud, load_golden
    ref = reference()
    g, files, kwargs = load_golden(case)
    Xf, Xm = load_cloud(files[0]), load_cloud(files[1])
    runs = {}
    for p in patterns:
        if p == "oracle":
            nv, pl = oracle_normals(g, kwargs, Xf)
            runs[p] = run_reference(ref, g, kwargs, Xf, Xm, nv, pl)
            dot = np.abs(np.sum(nv.astype(np.float64) * g["normals"], axis=1))
            runs[p]["normals_parallel_to_reference_frac"] = float(np.mean(dot > 1 - 1e-5))
            continue
        s = signs_for(p, g["normals"])
        runs[p] = run_reference(ref, g, kwargs, Xf, Xm, g["normals"] * s[:, None], g["planarity"])
        runs[p]["flipped"] = int(np.count_nonzero(s < 0))
    H0 = np.array(g["H"])
    dev = {p: float(np.abs(np.array(r["H"]) - H0).max()) for p, r in runs.items()}
    its = [r["iterations"] for r in runs.values()]
    return {"kwargs": repr(kwargs), "fixture_iterations": int(g["iterations"]),
            "max_abs_dH_vs_fixture": dev,
            "spread_H": max(dev.values()),
            "iterations_min": min(its), "iterations_max": max(its),
            "runs": runs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patterns", type=int, default=10)
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--out", default=str(DEFAULT_OUT))
    a = ap.parse_args()
    patterns = ["lapack", "convention", "oracle"] + [f"random{i}" for i in range(a.patterns)]
    out = {"_what": "unmodified reference re-run with its own normals re-signed / with the oracle's normals (oracle/normal_sensitivity.py); "
                    "max_abs_dH_vs_fixture = max |H - H_fixture| per sign pattern",
           "patterns": patterns, "cases": {}}
    for case in a.cases.split(","):
        r = measure(case, patterns)
        out["cases"][case] = r
        print(f"{case:12s} spread of H over {len(patterns)} patterns: {r['spread_H']:.2e}   "
              f"lapack {r['max_abs_dH_vs_fixture']['lapack']:.1e}  convention {r['max_abs_dH_vs_fixture']['convention']:.1e}  "
              f"oracle {r['max_abs_dH_vs_fixture']['oracle']:.1e}   "
              f"iterations {r['iterations_min']}..{r['iterations_max']} (fixture {r['fixture_iterations']})", flush=True)
    Path(a.out).write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
