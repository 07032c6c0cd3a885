"""Flows shared by tests/test_gpu_operators.py (the real library on the GPU box) and tests/test_host_mirror.py (the
host logic of the mirror on the oracle-backed stand-in context, CPU): the reference's iteration written out with ITS
OWN classes -- CorrPts.match / reject_wrt_planarity / reject_wrt_point_to_plane_distances,
SimpleICPOptimization.estimate_parameters (simpleicp.py:184-250) -- compared stage by stage with what the unmodified
reference recorded for the same runs (tests/golden) and with the CPU oracle."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_golden
from oracle import orc

def _sparse(n, rows, values):
    v = np.full(n, np.nan, np.float32)
    v[rows] = values
    return pd.arrays.SparseArray(v)


def point_clouds(name, clouds):
    """pc_fix as the reference had it when its loop started (selection, its own normals / planarity), pc_mov."""
    from simpleicp_amd import PointCloud
    g, files, kw = load_golden(name)
    pc_fix = PointCloud(clouds(files[0]), columns=["x", "y", "z"])
    pc_mov = PointCloud(clouds(files[1]).copy(), columns=["x", "y", "z"])
    sel = g["sel_idx"]
    pc_fix.idx_selected = sel
    for j, c in enumerate(("nx", "ny", "nz")):
        pc_fix[c] = _sparse(len(pc_fix), sel, g["normals"][:, j])
    pc_fix["planarity"] = _sparse(len(pc_fix), sel, g["planarity"])
    if "mov_sel_idx" in g.files:
        pc_mov.idx_selected = g["mov_sel_idx"]
        pc_mov["planarity"] = _sparse(len(pc_mov), g["mov_planarity_rows"], g["mov_planarity_vals"])
    return g, kw, pc_fix, pc_mov


def reference_loop(name, clouds):
    """simpleicp.py:184-250 with the mirror's classes, fed the reference's own estimate after every iteration so that
    each one starts from the reference's state -- including the few-ulp drift its in-place transform / inverse
    transform of pc2 leaves (simpleicp.py:188,202), which PointCloud.transform_by_H reproduces bit for bit.  Then:
    match == cKDTree's picks (exact ties aside: ours is the lower index), distances BIT-equal, both rejections leave
    the reference's rows, the minimiser is the reference's to its own tolerance and at least as good."""
    from simpleicp_amd.corrpts import CorrPts
    from simpleicp_amd.optimization import SimpleICPOptimization
    from simpleicp_amd.rbp import H_from_params
    g, kw, pc_fix, pc_mov = point_clouds(name, clouds)
    obs = np.array(kw.get("rbp_observed_values", (0.,) * 6), float)
    obs[:3] *= np.pi / 180
    ow = np.array(kw.get("rbp_observation_weights", (0.,) * 6), float)
    min_planarity = kw.get("min_planarity", 0.3)
    distance_weights = kw.get("distance_weights", 1)
    H = H_from_params(obs)
    if np.isfinite(kw.get("max_overlap_distance", np.inf)):      # simpleicp.py:161-163 moved pc2 there and back once already
        pc_mov.transform_by_H(H)
        pc_mov.transform_by_H(np.linalg.inv(H))
    X_fix = pc_fix.X
    optim = None
    for it in range(int(g["iterations"])):
        tag = f"it{it:03d}_"
        cp = CorrPts(pc_fix, pc_mov)
        pc_mov.transform_by_H(H)
        cp.match()
        X_mov_T = pc_mov.X
        pc_mov.transform_by_H(np.linalg.inv(H))
        assert cp.num_corr_pts == len(g[tag + "pc1_idx"])
        assert np.array_equal(cp._df["pc1_idx"].to_numpy(), g[tag + "pc1_idx"])
        mine, ref = cp._df["pc2_idx"].to_numpy(), g[tag + "pc2_idx"]
        same = mine == ref
        assert np.array_equal(cp.point_to_plane_distances[same], g[tag + "dist"][same])
        for j in np.flatnonzero(~same):                           # exact tie in d2, lower index taken
            d2 = orc.knn(X_mov_T[[mine[j], ref[j]]], X_fix[g[tag + "pc1_idx"][j]][None, :], k=2)[1][0]
            assert d2[0] == d2[1] and mine[j] < ref[j]
        exact = bool(same.all())

        cp.reject_wrt_planarity(min_planarity)
        if exact:
            assert np.array_equal(cp._df["pc1_idx"].to_numpy(), g[tag + "after_planarity_pc1_idx"])
        cp.reject_wrt_point_to_plane_distances()
        if exact:
            assert np.array_equal(cp._df["pc1_idx"].to_numpy(), g[tag + "kept_pc1_idx"])
            assert np.array_equal(cp._df["pc2_idx"].to_numpy(), g[tag + "kept_pc2_idx"])
        assert abs(cp.num_corr_pts - int(g["counts"][it])) <= (0 if exact else 2)

        if distance_weights is None:
            distance_weights = 1 / (np.std(cp.point_to_plane_distances) ** 2)
        w = float(g[tag + "w"])
        assert abs(distance_weights - w) <= (1e-12 if exact else 1e-2) * w
        optim = SimpleICPOptimization(cp, w, g[tag + "x0"], obs, ow)
        residuals = optim.estimate_parameters()
        x = np.array(optim.rbp.get_parameter_attributes_as_list("estimated_value"))
        assert len(residuals) == cp.num_corr_pts
        assert np.abs(x - g[tag + "x"]).max() < 5e-6              # the reference stops at ftol = xtol = gtol = 1e-8
        assert np.array_equal(x[~np.isfinite(ow)], g[tag + "x0"][~np.isfinite(ow)])       # fixed parameters do not move
        # the residuals ARE the point-to-plane distances at the estimate, from pc2's current (original) coordinates
        p1, n1 = X_fix[cp._df["pc1_idx"].to_numpy()], np.column_stack((cp.pc1_nx, cp.pc1_ny, cp.pc1_nz))
        p2 = np.column_stack((cp.pc2_x, cp.pc2_y, cp.pc2_z))
        assert np.allclose(residuals, orc.residuals(x, p1, n1, p2), rtol=0, atol=1e-13)
        if exact:
            def cost(xx):
                o = ow[(ow > 0) & np.isfinite(ow)] * (xx - obs)[(ow > 0) & np.isfinite(ow)]
                return np.sum((w * orc.residuals(xx, p1, n1, p2)) ** 2) + np.sum(o * o)
            assert cost(x) <= cost(g[tag + "x"]) * (1 + 1e-9)
            assert abs(residuals.std() - g[tag + "residuals"].std()) < 1e-5 * (1 + g[tag + "residuals"].std())
        H = H_from_params(g[tag + "x"])                           # rbp.H of the REFERENCE's estimate (simpleicp.py:247)

    optim.estimate_parameter_uncertainties()
    sigma = np.array(optim.rbp.get_parameter_attributes_as_list("estimated_uncertainty"))
    free = np.isfinite(ow)
    assert np.allclose(sigma[free], g["sigma"][free], rtol=5e-3) and np.all(np.isnan(sigma[~free]))


def _synthetic(n, Q, seed):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-20, 20, (n, 2))
    P = np.column_stack((xy, 2 * np.sin(xy[:, 0] / 5) * np.cos(xy[:, 1] / 7) + rng.normal(0, 0.01, n)))
    x_true = np.array([0.002, -0.001, 0.003, 0.05, -0.03, 0.02])
    Xm = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), P + rng.normal(0, 0.01, P.shape))
    return P, Xm, np.sort(rng.choice(n, Q, replace=False))


def abi_operators_vs_oracle(make_ctx, Q, n=60_000):
    """The C ABI's operators one by one (fused distances, the single-workgroup and the multi-workgroup selection,
    host-driven LM on the fused 6x6 reductions) against the oracle's iteration, bit for bit up to the solver, and
    against sicp_icp_iterate (the same iteration behind one call) on the same inputs."""
    from simpleicp_amd import _lib
    P, Xm, sel = _synthetic(n, Q, Q)
    z = np.zeros(6)
    x0 = np.array([0.001, 0.0, -0.001, 0.01, 0.0, 0.02])
    with make_ctx() as ctx:
        ctx.upload(_lib.FIX, P)
        ctx.upload(_lib.MOV, Xm)
        nv, pl = ctx.estimate_normals(_lib.FIX, sel, 10)
        pl[::7] = np.nan
        ctx.icp_setup(sel, nv, pl)
        o = orc.icp_iteration(Xm, P[sel], nv, pl, x0, x0, None, z, z, 0.3)

        idx, dist = ctx.corr_match(orc.params_to_H(x0))
        assert np.array_equal(idx, o["nn"]) and np.array_equal(dist, o["dist"])
        _, _, alive, _ = ctx.icp_state(pc2_idx=False, dist=False, residual=False)
        assert alive.all()
        n_pl = ctx.corr_reject_planarity(0.3, pl, None)
        assert n_pl == int(np.count_nonzero(pl >= np.float32(0.3)))
        med, mad, n = ctx.corr_reject_distances()
        _, _, alive, _ = ctx.icp_state(pc2_idx=False, dist=False, residual=False)
        assert np.array_equal(alive, o["keep"]) and n == o["n"] and med == o["median"] and mad == o["mad"]
        R = ctx.estimate_parameters(x0, z, z, distance_weight=None)
        x = np.array(R.x[:])
        assert np.abs(x - o["x"]).max() < 1e-9 and abs(R.weight_used - o["w"]) <= 1e-12 * o["w"]
        assert R.n_kept == o["n"]
        _, _, _, resid = ctx.icp_state(pc2_idx=False, dist=False, keep=False)
        assert np.allclose(resid[alive], orc.residuals(x, P[sel], nv, Xm[idx], alive), rtol=0, atol=1e-13)
        assert np.all(resid[~alive] == 0)
        s = ctx.icp_uncertainties()
        assert np.allclose(s, orc.uncertainties(x, R.weight_used, z, z, P[sel], nv, Xm[idx], alive), rtol=1e-9)

        # the same iteration behind ONE call
        ctx.icp_setup(sel, nv, pl)
        F = ctx.icp_iterate(x0, z, z, 0.3, None)
        fidx, fdist, fkeep, _ = ctx.icp_state()
        assert np.array_equal(fidx, idx) and np.array_equal(fdist, dist) and np.array_equal(fkeep, alive)
        assert F.median == med and F.mad == mad and np.abs(np.array(F.x[:]) - x).max() < 2e-9      # (each within 1e-9 of the oracle)
        # ... after which the operators have no correspondences of their own any more
        with pytest.raises(_lib.BackendError):
            ctx.corr_reject_distances()


def rejections_commute(make_ctx, n=30_000):
    """The reference's rejections are row filters on a DataFrame (corrpts.py:156,163,188): any order, any number of
    times.  MAD rejection first (median / MAD over ALL rows), then planarity; a second MAD pass works on the survivors."""
    from simpleicp_amd import _lib
    P, Xm, sel = _synthetic(n, 900, 3)
    with make_ctx() as ctx:
        ctx.upload(_lib.FIX, P)
        ctx.upload(_lib.MOV, Xm)
        nv, pl = ctx.estimate_normals(_lib.FIX, sel, 10)
        pl2 = np.random.default_rng(1).uniform(0, 1, len(sel)).astype(np.float32)      # a pc2 column, per correspondence
        pl2[::11] = np.nan
        ctx.icp_setup(sel, nv, pl)
        idx, dist = ctx.corr_match()
        keep1, n1, med1, mad1 = orc.reject(dist, np.ones(len(sel), np.float32), 0.0)       # nothing fails planarity
        med, mad, n = ctx.corr_reject_distances()
        assert (med, mad, n) == (med1, mad1, n1)
        n = ctx.corr_reject_planarity(0.4, pl, pl2)
        want = keep1 & (pl >= np.float32(0.4)) & (pl2 >= np.float32(0.4))
        _, _, alive, _ = ctx.icp_state(pc2_idx=False, dist=False, residual=False)
        assert np.array_equal(alive, want) and n == int(want.sum())
        # second distance rejection: statistics of the survivors only
        keep2, n2, med2, mad2 = orc.reject(dist, np.where(want, np.float32(1), np.float32(np.nan)), 0.0)
        med, mad, n = ctx.corr_reject_distances()
        _, _, alive, _ = ctx.icp_state(pc2_idx=False, dist=False, residual=False)
        assert (med, mad, n) == (med2, mad2, n2) and np.array_equal(alive, keep2)
        # everything rejected: median / MAD undefined, estimate refuses (simpleicp.py:209-214)
        assert ctx.corr_reject_planarity(2.0, pl, None) == 0
        med, mad, n = ctx.corr_reject_distances()
        assert n == 0 and np.isnan(med) and np.isnan(mad)
        with pytest.raises(_lib.BackendError) as e:
            ctx.estimate_parameters(np.zeros(6), np.zeros(6), np.zeros(6))
        assert e.value.code == _lib.ERR_TOO_FEW


def corrpts_object_semantics(clouds, tmp_path):
    """Bookkeeping of the mirror class: views, row labels, write_xyz, ownership of the device state."""
    from simpleicp_amd.corrpts import CorrPts, CorrPtsException
    from simpleicp_amd.optimization import SimpleICPOptimization
    g, kw, pc_fix, pc_mov = point_clouds("dragon", clouds)
    cp = CorrPts(pc_fix, pc_mov)
    assert cp.pc1_x is None and cp.num_corr_pts == 0                       # corrpts.py:30-36,119-122 before match()
    with pytest.raises(NotImplementedError):
        cp.reject_wrt_to_angle_between_normals()
    cp.match()
    n0 = cp.num_corr_pts
    assert n0 == len(g["sel_idx"]) and list(cp._df.columns) == ["pc1_idx", "pc2_idx", "point_to_plane_distances"]
    d = (cp.pc2_x - cp.pc1_x) * cp.pc1_nx + (cp.pc2_y - cp.pc1_y) * cp.pc1_ny + (cp.pc2_z - cp.pc1_z) * cp.pc1_nz
    assert np.array_equal(d, cp.point_to_plane_distances)                  # contract (P) == the reference's expression
    with pytest.raises(KeyError):                                          # pc2 has no normals: the reference's view raises too
        cp.pc2_nx
    cp.reject_wrt_planarity(0.3)
    cp.reject_wrt_point_to_plane_distances()
    assert 6 <= cp.num_corr_pts < n0
    assert cp._df.index.max() >= cp.num_corr_pts                           # rows dropped, labels kept (df.loc[keep])
    f = tmp_path / "corr.xyz"
    cp.write_xyz(f)
    back = np.loadtxt(f, comments="//")
    assert back.shape == (cp.num_corr_pts, 7) and np.array_equal(back[:, 6], cp.point_to_plane_distances)
    assert f.read_text().splitlines()[0] == "//X1 Y1 Z1 X2 Y2 Z2 point_to_plane_distance"
    optim = SimpleICPOptimization(cp, 1, (0.,) * 6, (0.,) * 6, (0.,) * 6)
    with pytest.raises(AttributeError):
        optim.estimate_parameter_uncertainties()
    res = optim.estimate_parameters()
    assert len(res) == cp.num_corr_pts and np.isfinite(optim.rbp.H).all()
    optim.estimate_parameter_uncertainties()
    assert all(np.isfinite(optim.rbp.get_parameter_attributes_as_list("estimated_uncertainty")))
    # a later match() takes the device state over
    other = CorrPts(pc_fix, pc_mov)
    other.match()
    with pytest.raises(CorrPtsException):
        cp.reject_wrt_point_to_plane_distances()
    with pytest.raises(CorrPtsException):
        optim.estimate_parameters()
    other.reject_wrt_point_to_plane_distances()
    # ... and so does a whole run
    from simpleicp_amd import SimpleICP
    icp = SimpleICP(verbose=False)
    icp.add_point_clouds(pc_fix, pc_mov)
    icp.run(max_iterations=1)
    with pytest.raises(CorrPtsException):
        other.reject_wrt_planarity(0.3)
