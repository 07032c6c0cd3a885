"""End-to-end: simpleicp_amd.SimpleICP.run on the bundled datasets against fixtures of the
unmodified reference (tests/golden, made by oracle/make_golden.py).  GPU only."""
import io
import logging

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN_CASES, GOLDEN_CHAIN, load_golden

pytestmark = pytest.mark.gpu


def _run(name, clouds, inject_normals, verbose=False, **extra):
    from simpleicp_amd import PointCloud, SimpleICP
    g, files, kw = load_golden(name)
    pc_fix = PointCloud(clouds(files[0]), columns=["x", "y", "z"])
    pc_mov = PointCloud(clouds(files[1]).copy(), columns=["x", "y", "z"])
    if "mov_sel_idx" in g.files:
        # the movable cloud was the fixed cloud of an earlier run: partial `selected` mask + sparse planarity column
        pc_mov.idx_selected = g["mov_sel_idx"]
        v = np.full(len(pc_mov), np.nan, np.float32)
        v[g["mov_planarity_rows"]] = g["mov_planarity_vals"]
        pc_mov["planarity"] = pd.arrays.SparseArray(v)
    if inject_normals:
        # the reference's own bypass (simpleicp.py:176): precomputed attribute columns
        sel = g["sel_idx"]
        for j, c in enumerate(("nx", "ny", "nz")):
            v = np.full(len(pc_fix), np.nan, np.float32)
            v[sel] = g["normals"][:, j]
            pc_fix[c] = pd.arrays.SparseArray(v)
        v = np.full(len(pc_fix), np.nan, np.float32)
        v[sel] = g["planarity"]
        pc_fix["planarity"] = pd.arrays.SparseArray(v)
    icp = SimpleICP(verbose=verbose)
    icp.add_point_clouds(pc_fix, pc_mov)
    out = icp.run(**{**kw, **extra})
    return g, kw, icp, pc_fix, pc_mov, out


@pytest.mark.parametrize("name", GOLDEN_CASES + GOLDEN_CHAIN)
def test_run_with_reference_normals(name, clouds):
    """Identical inputs to the loop (the reference's normals): H must match the reference to
    1e-7 absolute per entry (its own least_squares tolerance is 1e-8 relative), iteration count
    and correspondence counts must be the reference's (cKDTree tie picks may flip <= 2)."""
    g, kw, icp, pc_fix, pc_mov, (H, X, rbp, res) = _run(name, clouds, True)
    tol = 2e-6 if name == "bunny_obs" else 1e-7
    assert np.abs(H - g["H"]).max() < tol
    assert icp.last_run_info["iterations"] == int(g["iterations"])
    counts = np.array([s[0] for s in icp.last_run_info["stats"]])
    assert np.abs(counts - g["counts"]).max() <= 2
    x = np.array(rbp.get_parameter_attributes_as_list("estimated_value"))
    assert np.abs(x - g["x"]).max() < tol
    sig = np.array(rbp.get_parameter_attributes_as_list("estimated_uncertainty"))
    free = np.isfinite(np.array(kw.get("rbp_observation_weights", (0.,) * 6), float))
    assert np.allclose(sig[free], g["sigma"][free], rtol=2e-3) and np.all(np.isnan(sig[~free]))
    assert abs(len(res) - len(g["residuals"])) <= 2
    assert abs(res.std() - g["residuals"].std()) < 1e-6
    # side effects (simpleicp.py:316, :254): movable cloud transformed in place, selection kept
    assert X.shape == (len(pc_mov), 3) and np.array_equal(X, pc_mov.X)
    assert np.abs(X[:64] - g["X_mov_transformed_head"]).max() < 1e-5
    assert np.abs(X.sum(axis=0) - g["X_mov_transformed_sum"]).max() < 1e-4 * len(X)
    assert np.array_equal(pc_fix.idx_selected, g["sel_idx"])
    if name in GOLDEN_CHAIN:
        # the movable cloud's own selection survives the run (the reference never edits pc2's mask), and the pc2
        # planarity filter was active: some iteration lost correspondences to it (corrpts.py:158-163)
        assert np.array_equal(pc_mov.idx_selected, g["mov_sel_idx"])
        n_pl1 = int(np.count_nonzero(g["planarity"] >= np.float32(kw.get("min_planarity", 0.3))))
        assert min(len(g[f"it{i:03d}_after_planarity_pc1_idx"]) for i in range(int(g["iterations"]))) < n_pl1


# The reference's result depends on things its normals leave open: the (arbitrary, LAPACK-internal) SIGN of each normal --
# through the signed-median / MAD filter -- and cKDTree's pick among equidistant k-th neighbours.  oracle/normal_sensitivity.py
# measured by how much (the UNMODIFIED reference re-run with its own normals re-signed under 12 patterns, and with the
# normals this package computes; committed as tests/golden/normal_sensitivity.json, held together by
# tests/test_normal_sensitivity.py): H moves by 1e-6 (Dragon), 2e-4 (Bunny), 1.5e-3 (Webots), 6e-2 (Multisensor), the
# iteration count by several.  So two comparisons with EARNED tolerances:
#   * against the reference's published-style result (the fixture): inside the spread the reference itself shows;
#   * against the reference FED THE SAME NORMALS ("oracle" run of that file): tight, same iteration count.
import json as _json
from conftest import GOLDEN as _GOLDEN
SENS = _json.loads((_GOLDEN / "normal_sensitivity.json").read_text())["cases"]
OWN_NORMALS_TOL = {name: r["spread_H"] * 1.001 + 1e-7 for name, r in SENS.items()}
SAME_NORMALS_TOL = {"dragon": 1e-7, "bunny": 1e-7, "webots": 5e-7, "multisensor": 5e-7}


@pytest.mark.parametrize("name", ["dragon", "bunny", "multisensor", "webots"])
def test_run_own_normals(name, clouds):
    """Everything on the GPU including the normals."""
    from oracle import orc
    g, kw, icp, pc_fix, pc_mov, (H, X, rbp, res) = _run(name, clouds, False)
    o = orc.run(clouds(str(g["files"][0])), clouds(str(g["files"][1])), **kw)
    assert np.array_equal(pc_fix.idx_selected, o["sel"])
    assert np.array_equal(pc_fix.idx_selected, g["sel_idx"])          # overlap + sub-sampling parity
    assert icp.last_run_info["iterations"] == o["iterations"]
    assert [s[0] for s in icp.last_run_info["stats"]] == [s[0] for s in o["stats"]]
    assert np.abs(H - o["H"]).max() < 1e-9
    # the unmodified reference, handed these normals, takes the same number of iterations to the same H
    same = SENS[name]["runs"]["oracle"]
    assert icp.last_run_info["iterations"] == same["iterations"]
    assert abs(icp.last_run_info["stats"][-1][0] - same["final_correspondences"]) <= 2
    assert np.abs(H - np.array(same["H"])).max() < SAME_NORMALS_TOL[name]
    # and against the reference with its own (LAPACK-signed, cKDTree-tie-picked) normals: inside the reference's own spread
    assert np.abs(H - g["H"]).max() <= OWN_NORMALS_TOL[name]
    for c in ("nx", "ny", "nz", "planarity"):
        assert str(pc_fix[c].dtype) == "Sparse[float32, nan]"           # pointcloud.py:180-183,200-203
        assert np.isnan(pc_fix[c].to_numpy()).sum() == len(pc_fix) - len(g["sel_idx"])


def test_log_lines_match_reference_format(clouds):
    """Same lines as the reference prints (python/README.md:44-75); numbers to printed precision."""
    log = logging.getLogger("simpleicp_amd")
    buf = io.StringIO()
    h = logging.StreamHandler(buf)
    h.setFormatter(logging.Formatter("%(message)s"))
    log.addHandler(h)
    log.setLevel(logging.INFO)
    try:
        g, kw, icp, *_ = _run("bunny", clouds, True)
    finally:
        log.removeHandler(h)
    ours = buf.getvalue().splitlines()
    # normals were injected (the reference's own bypass), so its "Estimate normals ..." line is not due
    ref = [l for l in str(g["log"]).splitlines() if not l.startswith("Estimate normals")]
    assert len(ours) == len(ref)
    same = sum(a == b for a, b in zip(ours[:-1], ref[:-1]))
    assert same >= len(ref) - 4                    # counts can differ by one on a tie flip
    assert ours[-1].startswith("Finished in ") and ours[-1].endswith(" seconds!")
    assert ours[3] == ref[3] and ours[4] == ref[4]          # table header + 'orig:0' row


def test_exceptions(clouds):
    from simpleicp_amd import PointCloud, SimpleICP, SimpleICPException
    X = clouds("bunny_part1")
    icp = SimpleICP(verbose=False)
    icp.add_point_clouds(PointCloud(X, columns=["x", "y", "z"]), PointCloud(X + 1000.0, columns=["x", "y", "z"]))
    with pytest.raises(SimpleICPException, match="do not overlap"):
        icp.run(max_overlap_distance=0.5)
    with pytest.raises(SimpleICPException, match="distance_weights"):
        icp.run(distance_weights=0)
    with pytest.raises(SimpleICPException, match="exactly 6"):
        icp.run(rbp_observed_values=(0, 0, 0))
    with pytest.raises(SimpleICPException, match="finite"):
        icp.run(rbp_observation_weights=(np.inf,) * 6)


def test_edge_cases(clouds):
    """Inputs the reference handles (or rejects) at its API: tiny clouds, Q below 6, identical clouds,
    k larger than the cloud, selection preserved across runs."""
    from simpleicp_amd import PointCloud, SimpleICP, SimpleICPException, _lib
    rng = np.random.default_rng(1)
    X = np.column_stack((rng.uniform(0, 10, 400), rng.uniform(0, 10, 400), rng.normal(0, 0.01, 400)))
    # identical clouds: residuals are exactly zero, convergence test sees 0/0 -> 0 % change (simpleicp.py:364-366)
    icp = SimpleICP(verbose=False)
    icp.add_point_clouds(PointCloud(X, columns=["x", "y", "z"]), PointCloud(X.copy(), columns=["x", "y", "z"]))
    H, Xt, rbp, res = icp.run(correspondences=100)
    assert np.abs(H - np.eye(4)).max() < 1e-12 and np.all(res == 0) and icp.last_run_info["iterations"] == 2
    # fewer than 6 correspondences -> the reference's exception text (simpleicp.py:209-214)
    icp = SimpleICP(verbose=False)
    icp.add_point_clouds(PointCloud(X, columns=["x", "y", "z"]), PointCloud(X + 0.01, columns=["x", "y", "z"]))
    with pytest.raises(SimpleICPException, match="Too few correspondences"):
        icp.run(correspondences=5)
    # neighbors > number of points -> loud error, not garbage
    pc = PointCloud(X[:8], columns=["x", "y", "z"])
    with pytest.raises(_lib.BackendError, match="exceeds the number of points"):
        pc.estimate_normals(20)
    # standalone operators of the PointCloud mirror
    pc = PointCloud(X, columns=["x", "y", "z"])
    pc.select_in_range(X[:50] + 0.001, max_range=0.05)
    from oracle import orc
    idx, _ = orc.knn(X[:50] + 0.001, X, k=1, max_dist=0.05)
    assert np.array_equal(pc.idx_selected, np.flatnonzero(idx[:, 0] >= 0))
    Hm = orc.params_to_H(np.array([0.1, 0.2, 0.3, 1, 2, 3]))
    pc.transform_by_H(Hm)
    assert np.array_equal(pc.X, orc.transform(Hm, X))


def test_debug_dirpath_dumps(tmp_path, clouds):
    """simpleicp.py:141-143,189-200,216-221,317-320: same file names and formats as the reference."""
    from simpleicp_amd import PointCloud, SimpleICP, io
    g, files, kw = load_golden("bunny")
    _run("bunny", clouds, True, max_iterations=3, debug_dirpath=str(tmp_path / "dbg"))   # reference normals injected
    names = sorted(p.name for p in (tmp_path / "dbg").iterdir())
    assert names == ["iteration000_preoptim_correspondences.xyz", "iteration000_preoptim_pcfix.xyz",
                     "iteration000_preoptim_pcmov.xyz", "iteration001_preoptim_correspondences.xyz",
                     "iteration001_preoptim_pcmov.xyz", "iteration002_postoptim_pcmov.xyz",
                     "iteration002_preoptim_correspondences.xyz", "iteration002_preoptim_pcmov.xyz"]
    head = (tmp_path / "dbg" / "iteration000_preoptim_correspondences.xyz").read_text().splitlines()
    assert head[0] == "//X1 Y1 Z1 X2 Y2 Z2 point_to_plane_distance" and len(head[1].split()) == 7
    assert (tmp_path / "dbg" / "iteration000_preoptim_pcfix.xyz").read_text().splitlines()[0] == "//X Y Z"
    assert io.read_xyz(tmp_path / "dbg" / "iteration002_postoptim_pcmov.xyz").shape == clouds(files[1]).shape
    # the correspondence dump is what the reference wrote at iteration 0 (same kept set, same %.18e distances)
    C = np.loadtxt(tmp_path / "dbg" / "iteration000_preoptim_correspondences.xyz", comments="//")
    assert len(C) == int(g["counts"][0])
    assert np.array_equal(C[:, 6], g["it000_dist"][np.isin(g["it000_pc1_idx"], g["it000_kept_pc1_idx"])])
