"""Host logic of the Python mirror without a GPU: ``SimpleICP.run``'s orchestration / log / side effects and the
bookkeeping of ``CorrPts`` / ``SimpleICPOptimization`` run against tests/oracle_backend.py (a stand-in context that
answers with the CPU oracle) and are held against the fixtures of the unmodified reference.  What is under test here is
simpleicp_amd/{icp,pointcloud,corrpts,optimization}.py; the kernels are checked by the ``-m gpu`` tests, which run the
same flows (tests/operator_flow.py) on the real library."""
import logging

import numpy as np
import pandas as pd
import pytest

import operator_flow as flow
import oracle_backend
from conftest import load_golden


@pytest.fixture
def ctx(monkeypatch):
    return oracle_backend.install(monkeypatch)


@pytest.mark.parametrize("name", ["dragon", "bunny_obs", "dragon_chain"])
def test_reference_loop_with_mirror_classes(ctx, name, clouds):
    flow.reference_loop(name, clouds)
    # one match / two rejections / one estimate per iteration went to the backend, nothing was computed on the side
    g, _, _ = load_golden(name)
    for op in ("corr_match", "corr_reject_distances", "estimate_parameters"):
        assert ctx.calls.count(op) == int(g["iterations"])


def test_corrpts_bookkeeping(ctx, clouds, tmp_path):
    flow.corrpts_object_semantics(clouds, tmp_path)


def test_backend_stand_in_agrees_with_itself(ctx):
    """(keeps the stand-in honest: its operator-level answers equal its fused iteration's)"""
    flow.abi_operators_vs_oracle(oracle_backend.OracleContext, 300, n=5000)
    flow.rejections_commute(oracle_backend.OracleContext, n=4000)


@pytest.mark.parametrize("name", ["dragon", "bunny"])
def test_run_orchestration_on_the_stand_in(ctx, name, clouds):
    """SimpleICP.run end to end (overlap pre-pass, selection, normals, loop, final transform): the reference's H,
    iteration count, log lines and side effects -- host logic only, the numbers come from the oracle."""
    from simpleicp_amd import PointCloud, SimpleICP
    g, files, kw = load_golden(name)
    pc_fix = PointCloud(clouds(files[0]), columns=["x", "y", "z"])
    pc_mov = PointCloud(clouds(files[1]).copy(), columns=["x", "y", "z"])
    sel = g["sel_idx"]
    for j, c in enumerate(("nx", "ny", "nz", "planarity")):
        v = np.full(len(pc_fix), np.nan, np.float32)
        v[sel] = g["planarity"] if c == "planarity" else g["normals"][:, j]
        pc_fix[c] = pd.arrays.SparseArray(v)
    records = []
    handler = logging.Handler()
    handler.emit = lambda r: records.append(r.getMessage())
    log = logging.getLogger("simpleicp_amd")
    log.addHandler(handler)
    old = log.level
    log.setLevel(logging.INFO)
    try:
        icp = SimpleICP(verbose=False)
        icp.add_point_clouds(pc_fix, pc_mov)
        X0 = pc_mov.X
        H, X, rbp, res = icp.run(**kw)
    finally:
        log.removeHandler(handler)
        log.setLevel(old)
    assert np.abs(H - g["H"]).max() < 1e-7
    assert icp.last_run_info["iterations"] == int(g["iterations"])
    assert np.abs(np.array([s[0] for s in icp.last_run_info["stats"]]) - g["counts"]).max() <= 2
    assert np.array_equal(pc_fix.idx_selected, sel)                       # selection left at the sub-sample (simpleicp.py:254)
    assert np.array_equal(X, pc_mov.X) and not np.array_equal(X, X0)      # movable cloud transformed in place (simpleicp.py:316)
    assert np.abs(X[:64] - g["X_mov_transformed_head"]).max() < 1e-5
    assert abs(len(res) - len(g["residuals"])) <= 2
    # normals were injected (the reference's own bypass, simpleicp.py:176), so its "Estimate normals ..." line is not due
    theirs = [m for m in str(g["log"]).splitlines() if not m.startswith(("Finished in", "Estimate normals"))]
    mine = [m for m in records if not m.startswith("Finished in")]
    assert len(mine) == len(theirs)
    assert sum(a == b for a, b in zip(mine, theirs)) >= len(theirs) - 4          # a count can differ by one on a tie flip
    assert [m for m in mine if "|" not in m and "[" not in m] == [m for m in theirs if "|" not in m and "[" not in m]
    assert records[-1].startswith("Finished in ") and records[-1].endswith(" seconds!")
    assert "icp_run" in ctx.calls and ctx.calls.count("upload") + ctx.calls.count("upload_start") >= 2


def _small_pair(clouds, n=4000):
    from simpleicp_amd import PointCloud
    X1, X2 = clouds("bunny_part1")[:n], clouds("bunny_part2")[:n]
    return PointCloud(X1, columns=["x", "y", "z"]), PointCloud(X2.copy(), columns=["x", "y", "z"])


def test_run_debug_dumps_and_iterate_path(ctx, clouds, tmp_path):
    """debug_dirpath (simpleicp.py:140-142,193-203,222-227,317-321): the loop then runs iteration by iteration
    (sicp_icp_iterate) and writes the reference's files under the reference's names."""
    from simpleicp_amd import SimpleICP
    pc_fix, pc_mov = _small_pair(clouds)
    icp = SimpleICP(verbose=False)
    icp.add_point_clouds(pc_fix, pc_mov)
    out = tmp_path / "dbg" / "nested"
    H, X, rbp, res = icp.run(correspondences=300, max_iterations=2, min_change=0.0, debug_dirpath=str(out))
    names = sorted(p.name for p in out.iterdir())
    assert names == ["iteration000_preoptim_correspondences.xyz", "iteration000_preoptim_pcfix.xyz",
                     "iteration000_preoptim_pcmov.xyz", "iteration001_postoptim_pcmov.xyz",
                     "iteration001_preoptim_correspondences.xyz", "iteration001_preoptim_pcmov.xyz"]
    assert ctx.calls.count("icp_iterate") == 2 and "icp_run" not in ctx.calls
    corr = np.loadtxt(out / "iteration001_preoptim_correspondences.xyz", comments="//")
    assert corr.shape[1] == 7 and 6 <= len(corr) <= 300
    # column 7 is the point-to-plane distance of the row's two points along pc1's normal: recompute it from the frame
    sel = pc_fix.idx_selected
    p1 = pc_fix.X[sel]
    row = {tuple(p): i for i, p in enumerate(np.round(p1, 12))}
    at = np.array([row[tuple(p)] for p in np.round(corr[:, :3], 12)])
    nv = np.column_stack([np.asarray(pc_fix[c].to_numpy(), dtype=np.float64)[sel][at] for c in ("nx", "ny", "nz")])
    d = np.sum((corr[:, 3:6] - corr[:, :3]) * nv, axis=1)
    assert np.abs(d - corr[:, 6]).max() < 1e-12
    post = np.loadtxt(out / "iteration001_postoptim_pcmov.xyz", comments="//")
    assert post.shape == X.shape and np.abs(post - X).max() <= 5.01e-4          # "%.3f"
    assert (out / "iteration000_preoptim_pcfix.xyz").read_text().splitlines()[0] == "//X Y Z"


def test_run_exceptions_on_the_stand_in(ctx, clouds):
    """Messages of simpleicp.py:165-170,209-214 reach the caller as SimpleICPException from either loop flavour."""
    from simpleicp_amd import PointCloud, SimpleICP, SimpleICPException
    pc_fix, pc_mov = _small_pair(clouds)
    icp = SimpleICP(verbose=False)
    icp.add_point_clouds(pc_fix, pc_mov)
    with pytest.raises(SimpleICPException, match=r"Too few correspondences! .* number of correspondences is 0\."):
        icp.run(correspondences=200, min_planarity=2.0)
    far = PointCloud(pc_mov.X + 1000.0, columns=["x", "y", "z"])
    icp.add_point_clouds(pc_fix, far)
    with pytest.raises(SimpleICPException, match="do not overlap within max_overlap_distance = 0.50000"):
        icp.run(max_overlap_distance=0.5)
    # automatic distance weight (simpleicp.py:229-234): frozen after the first iteration
    pc_fix, pc_mov = _small_pair(clouds)
    icp.add_point_clouds(pc_fix, pc_mov)
    H, X, rbp, res = icp.run(correspondences=300, max_iterations=3, distance_weights=None)
    assert np.isfinite(H).all() and len(res) >= 6


@pytest.mark.parametrize("dataset", ["Bunny", "Multisensor"])
def test_reference_test_module_on_the_stand_in(ctx, dataset, clouds, tmp_path):
    """tests/test_reference_suite.py (the reference's own test module, package name swapped) -- here with the stand-in
    context: .xyz round trip through the native reader / writer, debug dumps, fixed and observed parameters."""
    import test_reference_suite as suite
    for case in suite.test_simpleicp.pytestmark[0].args[1]:
        if case[0] == dataset:
            suite.test_simpleicp(*case, clouds, tmp_path)
            assert ctx.calls.count("icp_iterate") >= 2 and "icp_run" not in ctx.calls
            return
    raise AssertionError(dataset)


def test_cli_in_process_on_the_stand_in(ctx, clouds, tmp_path, capsys):
    """python -m simpleicp_amd (cli.main) end to end: .xyz in, the README's H line out (python/README.md:62), the
    transformed cloud written; a missing file and a failed run come back as exit code 1 with the reference CLI's text."""
    from simpleicp_amd import cli, io
    g, files, kw = load_golden("bunny")
    io.write_xyz(tmp_path / "f.xyz", clouds(files[0]), decimals=4, header=None)
    io.write_xyz(tmp_path / "m.xyz", clouds(files[1]), decimals=4, header=None)
    rc = cli.main(["-f", str(tmp_path / "f.xyz"), "-m", str(tmp_path / "m.xyz"), "-o", "1", "--quiet",
                   "--output", str(tmp_path / "out.xyz")])
    out = capsys.readouterr().out.splitlines()
    assert rc == 0 and len(out) == 4
    Hq = np.array([[float(v) for v in row.split()] for row in out])
    assert np.abs(Hq - g["H"]).max() < 1e-4 and out[3].split() == ["0.000000000", "0.000000000", "0.000000000", "1.000000000"]
    assert io.read_xyz(tmp_path / "out.xyz").shape == clouds(files[1]).shape
    assert cli.main(["-f", str(tmp_path / "nope.xyz"), "-m", str(tmp_path / "m.xyz"), "--quiet"]) == 1
    assert "Caught exception:" in capsys.readouterr().err
    assert cli.main(["-f", str(tmp_path / "f.xyz"), "-m", str(tmp_path / "m.xyz"), "-p", "2", "--quiet"]) == 1
    assert "Too few correspondences" in capsys.readouterr().err


def test_pointcloud_operators_on_the_stand_in(ctx, clouds):
    """The PointCloud mirror's own operators as a caller uses them (pointcloud.py:132-217): the overlap pre-pass +
    sub-sampling reproduce the reference's selection, estimate_normals leaves the reference's column layout, the
    transform is contract (T)."""
    from oracle import orc
    from simpleicp_amd import PointCloud
    g, files, kw = load_golden("bunny")
    X_fix, X_mov = clouds(files[0]), clouds(files[1])
    pc = PointCloud(X_fix, columns=["x", "y", "z"])
    pc.select_in_range(X_mov, max_range=kw["max_overlap_distance"])        # simpleicp.py:161-163 with H0 = identity
    assert 0 < pc.num_selected_points < pc.num_points
    pc.select_n_points(1000)
    assert np.array_equal(pc.idx_selected, g["sel_idx"])
    pc.estimate_normals(10)
    sel = pc.idx_selected
    nn, _ = orc.knn(X_fix, X_fix[sel], k=10)
    nv, pl = orc.normals(X_fix, nn)
    for j, c in enumerate(("nx", "ny", "nz")):
        assert str(pc[c].dtype) == "Sparse[float32, nan]"
        dense = pc[c].to_numpy()
        assert np.array_equal(dense[sel], nv[:, j]) and np.isnan(np.delete(dense, sel)).all()
    assert np.array_equal(pc["planarity"].to_numpy()[sel], pl)
    # empty selection: nothing to do, nothing breaks (the reference's loops are empty then)
    pc.unselect_all_points()
    pc.select_in_range(X_mov, max_range=1.0)
    assert pc.num_selected_points == 0
    pc.select_all_points()
    Hm = orc.params_to_H(np.array([0.1, 0.2, 0.3, 1, 2, 3]))
    pc.transform_by_H(Hm)
    assert np.array_equal(pc.X, orc.transform(Hm, X_fix))
    assert pc["x"].to_numpy().flags.c_contiguous and np.array_equal(pc.x, pc.X[:, 0])


def test_frames_a_caller_may_hand_over(ctx, clouds):
    """PointCloud is a DataFrame subclass: the mirror must take what pandas takes (pointcloud.py:15-49) -- a frame with
    a foreign index, extra columns and another column order gives the SAME registration as the plain array; float32
    input and a preset selection run through; the reference's error for a missing coordinate column."""
    from simpleicp_amd import PointCloud, PointCloudException, SimpleICP
    X1, X2 = clouds("bunny_part1"), clouds("bunny_part2")

    def run(pc1, pc2):
        icp = SimpleICP(verbose=False)
        icp.add_point_clouds(pc1, pc2)
        return icp.run(max_overlap_distance=1)[0]
    H0 = run(PointCloud(X1, columns=["x", "y", "z"]), PointCloud(X2.copy(), columns=["x", "y", "z"]))
    df1 = pd.DataFrame({"x": X1[:, 0], "y": X1[:, 1], "z": X1[:, 2], "intensity": np.arange(len(X1))},
                       index=np.arange(len(X1))[::-1] + 1000)
    df2 = pd.DataFrame({"z": X2[:, 2], "y": X2[:, 1], "x": X2[:, 0]})
    assert np.array_equal(run(PointCloud(df1), PointCloud(df2)), H0)
    H32 = run(PointCloud(X1.astype(np.float32), columns=["x", "y", "z"]), PointCloud(X2.astype(np.float32), columns=["x", "y", "z"]))
    assert np.abs(H32 - H0).max() < 1e-3
    pc1 = PointCloud(X1, columns=["x", "y", "z"])
    pc1.select_by_indices(np.arange(0, len(X1), 2))
    assert np.abs(run(pc1, PointCloud(X2.copy(), columns=["x", "y", "z"])) - H0).max() < 5e-3 and pc1.num_selected_points == 1000
    assert PointCloud(X1[:50].tolist(), columns=["x", "y", "z"]).X.dtype == np.float64
    with pytest.raises(PointCloudException, match='Column "z" is missing'):
        PointCloud(X1[:5, :2], columns=["x", "y"])
