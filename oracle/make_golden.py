"""Generate tests/golden/*.npz by running the UNMODIFIED reference package.

Run in the build container only (needs /root/reference; the GPU box has no copy):

    python oracle/make_golden.py [--out DIR]      (default: tests/golden)

What it does
  1. puts oracle/shim (a stand-in for the absent third-party ``lmfit``) and
     /root/reference/python on sys.path and imports the reference ``simpleicp``;
  2. checks the Bunny run against the known-answer output printed in
     /root/reference/python/README.md:44-73 (H, rbp, uncertainties to all printed
     digits) -- the only value-level pin the reference itself ships;
  3. runs the reference's own test configurations (python/simpleicp/tests/
     test_simpleicp.py:35-99: Dragon, Bunny, Multisensor, Webots; the Airborne /
     Terrestrial inputs are missing blobs) with non-invasive wrappers that record,
     per ICP iteration, what CorrPts.match / reject_* / estimate_parameters
     produced, and stores everything as small .npz fixtures;
  4. stores the bundled input clouds as int32 * 1e-4 (all files carry <= 4
     decimals; the script asserts the round trip is bit-exact) so the GPU-side
     tests can run the same datasets without /root/reference.
"""
import io
import logging
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT / "oracle" / "shim"))
sys.path.insert(0, str(REF / "python"))

import simpleicp as ref  # noqa: E402  (the unmodified reference package)
from simpleicp import corrpts as ref_corrpts, optimization as ref_optim  # noqa: E402

GOLD = ROOT / "tests" / "golden"
DATA = GOLD / "data"

CASES = {
    # name: (fixed file, movable file, run kwargs)  -- test_simpleicp.py:35-99
    "dragon": ("dragon1.xyz", "dragon2.xyz", {}),
    "bunny": ("bunny_part1.xyz", "bunny_part2.xyz", {"max_overlap_distance": 1}),
    "multisensor": ("multisensor_lidar.xyz", "multisensor_radar.xyz",
                    {"max_overlap_distance": 1,
                     "rbp_observed_values": (-0.5, 0.0, 0.0, 0.0, 0.0, 0.0),
                     "rbp_observation_weights": (np.inf, np.inf, 0.0, 0.0, 0.0, 0.0)}),
    "webots": ("webots1.xyz", "webots2.xyz",
               {"neighbors": 40, "max_overlap_distance": 0.5,
                "rbp_observed_values": (0.0, 0.0, -60.0, -0.05, -0.09, 0.0),
                "rbp_observation_weights": (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)}),
    # extra coverage of kwargs the reference supports but does not test:
    "dragon_q5000": ("dragon1.xyz", "dragon2.xyz", {"correspondences": 5000, "neighbors": 20}),
    "dragon_kw": ("dragon1.xyz", "dragon2.xyz", {"correspondences": 500, "neighbors": 7, "min_planarity": 0.5,
                                                 "distance_weights": 4.0, "min_change": 3.0, "max_iterations": 6}),
    "bunny_obs": ("bunny_part1.xyz", "bunny_part2.xyz",
                  {"max_overlap_distance": 1, "distance_weights": None,
                   "rbp_observed_values": (0.0, 0.0, 10.0, 0.0, 0.0, 0.0),
                   "rbp_observation_weights": (100.0, 100.0, 50.0, 10.0, 10.0, np.inf)}),
}


def store_cloud(name):
    X = np.genfromtxt(REF / "data" / name)
    q = np.rint(X * 1e4).astype(np.int64)
    assert np.abs(q).max() < 2 ** 31
    back = q.astype(np.float64) / 1e4
    assert np.array_equal(back, X), f"{name}: int32*1e-4 round trip is not bit-exact"
    DATA.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(DATA / (Path(name).stem + ".npz"), q=q.astype(np.int32))
    return X


# Chained runs: the movable cloud of the recorded run is the FIXED cloud of an earlier run, so it carries the
# sparse nx / ny / nz / planarity columns and (first case) a partial `selected` mask -- the two things
# CorrPts.match / reject_wrt_planarity read from pc2 (corrpts.py:131-135, 158-163).
#   name: (stage-1 fixed = recorded run's MOVABLE, stage-1 movable = recorded run's FIXED (fresh copy),
#          stage-1 kwargs, select_all_points() on the carried cloud before the recorded run, recorded kwargs)
CHAIN_CASES = {
    "dragon_chain": ("dragon1.xyz", "dragon2.xyz", {}, False, {}),
    "bunny_chain": ("bunny_part1.xyz", "bunny_part2.xyz", {"max_overlap_distance": 1, "correspondences": 30000},
                    True, {"max_overlap_distance": 1}),
}


def run_chain_case(name, fa, fb, kw1, select_all, kw2):
    Xa, Xb = store_cloud(fa), store_cloud(fb)
    carried = ref.PointCloud(Xa, columns=["x", "y", "z"])
    icp = ref.SimpleICP(verbose=False)
    icp.add_point_clouds(carried, ref.PointCloud(Xb.copy(), columns=["x", "y", "z"]))
    icp.run(**kw1)
    assert np.array_equal(carried.X, Xa) and "planarity" in carried      # the fixed cloud is never moved
    if select_all:
        carried.select_all_points()
    return run_case(name, fb, fa, kw2, pc_mov=carried)


def run_case(name, f1, f2, kwargs, pc_mov=None):
    X_fix, X_mov = store_cloud(f1), store_cloud(f2)
    pc_fix = ref.PointCloud(X_fix, columns=["x", "y", "z"])
    mov_extra = {}
    if pc_mov is None:
        pc_mov = ref.PointCloud(X_mov.copy(), columns=["x", "y", "z"])
    else:
        pl = pc_mov["planarity"].to_numpy()
        rows = np.flatnonzero(~np.isnan(pl))
        mov_extra = {"mov_sel_idx": pc_mov.idx_selected.astype(np.int64), "mov_planarity_rows": rows.astype(np.int64),
                     "mov_planarity_vals": pl[rows].astype(np.float32)}
    trace = []

    orig_match = ref_corrpts.CorrPts.match
    orig_rp = ref_corrpts.CorrPts.reject_wrt_planarity
    orig_rd = ref_corrpts.CorrPts.reject_wrt_point_to_plane_distances
    orig_est = ref_optim.SimpleICPOptimization.estimate_parameters

    def match(self):
        orig_match(self)
        trace.append({"pc1_idx": self._df["pc1_idx"].to_numpy().copy(),
                      "pc2_idx": self._df["pc2_idx"].to_numpy().copy(),
                      "dist": self.point_to_plane_distances.copy()})

    def rp(self, t):
        orig_rp(self, t)
        trace[-1]["after_planarity_pc1_idx"] = self._df["pc1_idx"].to_numpy().copy()

    def rd(self):
        orig_rd(self)
        trace[-1]["kept_pc1_idx"] = self._df["pc1_idx"].to_numpy().copy()
        trace[-1]["kept_pc2_idx"] = self._df["pc2_idx"].to_numpy().copy()

    def est(self):
        r = orig_est(self)
        trace[-1]["x"] = np.array(self.rbp.get_parameter_attributes_as_list("estimated_value"), float)
        trace[-1]["x0"] = np.array(self.rbp.get_parameter_attributes_as_list("initial_value"), float)
        trace[-1]["w"] = float(self._distance_weights)
        trace[-1]["residuals"] = np.asarray(r).copy()
        return r

    ref_corrpts.CorrPts.match = match
    ref_corrpts.CorrPts.reject_wrt_planarity = rp
    ref_corrpts.CorrPts.reject_wrt_point_to_plane_distances = rd
    ref_optim.SimpleICPOptimization.estimate_parameters = est
    buf = io.StringIO()
    handler = logging.StreamHandler(buf)
    handler.setFormatter(logging.Formatter("%(message)s"))
    log = logging.getLogger("simpleicp")
    log.setLevel(logging.INFO)
    log.addHandler(handler)
    try:
        icp = ref.SimpleICP(verbose=False)
        icp.add_point_clouds(pc_fix, pc_mov)
        H, X_out, rbp, residuals = icp.run(**kwargs)
    finally:
        ref_corrpts.CorrPts.match = orig_match
        ref_corrpts.CorrPts.reject_wrt_planarity = orig_rp
        ref_corrpts.CorrPts.reject_wrt_point_to_plane_distances = orig_rd
        ref_optim.SimpleICPOptimization.estimate_parameters = orig_est
        log.removeHandler(handler)

    sel = pc_fix.idx_selected
    normals = np.column_stack([pc_fix[c].to_numpy()[sel] for c in ("nx", "ny", "nz")]).astype(np.float32)
    planarity = pc_fix["planarity"].to_numpy()[sel].astype(np.float32)
    out = {
        "H": H, "residuals": residuals,
        "x": np.array(rbp.get_parameter_attributes_as_list("estimated_value"), float),
        "sigma": np.array(rbp.get_parameter_attributes_as_list("estimated_uncertainty"), float),
        "sel_idx": sel.astype(np.int64), "normals": normals, "planarity": planarity,
        "iterations": np.int64(len(trace)),
        "counts": np.array([len(t["kept_pc1_idx"]) for t in trace], np.int64),
        "X_mov_transformed_head": X_out[:64].copy(),
        "X_mov_transformed_sum": X_out.sum(axis=0),
        "log": np.array(buf.getvalue()),
        "kwargs": np.array(repr(kwargs)),
        "files": np.array([f1, f2]),
        **mov_extra,
    }
    for i, t in enumerate(trace):
        for k, v in t.items():
            out[f"it{i:03d}_{k}"] = np.asarray(v)
    GOLD.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLD / f"{name}.npz", **out)
    return H, rbp, buf.getvalue(), trace


def check_readme_kat(H, rbp, text):
    """/root/reference/python/README.md:62-73 (Bunny, max_overlap_distance=1)."""
    want_H = np.array([[0.984798, -0.173702, -0.000053, 0.000676],
                       [0.173702, 0.984798, 0.000084, -0.001150],
                       [0.000038, -0.000092, 1.000000, 0.000113],
                       [0, 0, 0, 1]])
    assert np.all(np.abs(H - want_H) < 5.0e-7 + 1e-12), "README H not reproduced"
    want = {"alpha1": (-0.004804, 0.004491), "alpha2": (-0.003061, 0.002104), "alpha3": (10.003124, 0.005680),
            "tx": (0.000676, 0.000418), "ty": (-0.001150, 0.000885), "tz": (0.000113, 0.000189)}
    for k, (v, s) in want.items():
        p = getattr(rbp, k)
        assert f"{p.estimated_value_scaled:.6f}" == f"{v:.6f}", (k, p.estimated_value_scaled, v)
        assert f"{p.estimated_uncertainty_scaled:.6f}" == f"{s:.6f}", (k, p.estimated_uncertainty_scaled, s)
    print("README.md:62-73 known-answer (Bunny H, rbp, uncertainties): reproduced to all printed digits")


def main():
    global GOLD, DATA
    if len(sys.argv) > 2 and sys.argv[1] == "--out":          # regenerate somewhere else (tests/test_fixtures_regenerate.py)
        GOLD = Path(sys.argv[2]).resolve()
        DATA = GOLD / "data"
    os.chdir(ROOT)
    for name, (f1, f2, kw) in CASES.items():
        H, rbp, text, trace = run_case(name, f1, f2, kw)
        print(f"{name}: {len(trace)} iterations, final n={len(trace[-1]['kept_pc1_idx'])}")
        if name == "bunny":
            check_readme_kat(H, rbp, text)
    for name, (fa, fb, kw1, select_all, kw2) in CHAIN_CASES.items():
        H, rbp, text, trace = run_chain_case(name, fa, fb, kw1, select_all, kw2)
        print(f"{name}: {len(trace)} iterations, final n={len(trace[-1]['kept_pc1_idx'])}")


if __name__ == "__main__":
    main()
