"""Parity of the HIP kernels (through the C ABI) against the CPU oracle.  GPU only."""
import os
import numpy as np
import pytest

from conftest import GOLDEN_CHAIN, load_golden, movable_columns
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from simpleicp_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _H(seed=0):
    rng = np.random.default_rng(seed)
    x = np.concatenate((rng.uniform(-0.3, 0.3, 3), rng.uniform(-1, 1, 3)))
    return orc.params_to_H(x)


@pytest.mark.parametrize("n,q", [(1, 1), (5, 3), (1000, 7), (1024, 1024), (1025, 1025), (50_000, 1000), (333_333, 2500)])
def test_knn1_bit_exact_random(ctx, n, q):
    from simpleicp_amd import _lib
    rng = np.random.default_rng(n + q)
    P = rng.uniform(-50, 50, (n, 3))
    Qp = rng.uniform(-50, 50, (q, 3))
    ctx.upload(_lib.MOV, P)
    for H in (None, _H(1)):
        idx, d2 = ctx.knn(_lib.MOV, Qp, k=1, H=H)
        ridx, rd2 = orc.knn(P, Qp, k=1, H=H)
        assert np.array_equal(idx, ridx)
        assert np.array_equal(d2, rd2)          # contract (T)+(D): bit-exact squared distances


def test_knn1_ties_lowest_index(ctx):
    """Quantised coordinates + exact duplicates: lexicographic (d2, idx) must hold."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(5)
    P = np.round(rng.uniform(-3, 3, (40_000, 3)), 1)
    P[20_000:] = P[:20_000]                      # every point twice
    Qp = np.round(rng.uniform(-3, 3, (900, 3)), 1)
    ctx.upload(_lib.MOV, P)
    idx, d2 = ctx.knn(_lib.MOV, Qp, k=1)
    ridx, rd2 = orc.knn(P, Qp, k=1)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
    assert np.all(idx < 20_000)


def test_knn1_upper_bound_strict(ctx):
    """select_in_range semantics (pointcloud.py:163-167): d < max_range strictly."""
    from simpleicp_amd import _lib
    P = np.array([[0.0, 0, 0], [3.0, 0, 0], [0, 4.0, 0]])
    Qp = np.array([[1.0, 0, 0], [3.0, 4.0, 0], [100.0, 0, 0]])
    ctx.upload(_lib.MOV, P)
    idx, d2 = ctx.knn(_lib.MOV, Qp, k=1, max_dist=3.0)
    ridx, rd2 = orc.knn(P, Qp, k=1, max_dist=3.0)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
    assert idx[:, 0].tolist() == [0, -1, -1]     # query 1 is at distance exactly 3 -> excluded
    assert np.isinf(d2[1, 0])


@pytest.mark.parametrize("nf,nm", [(3, 3), (900, 40_000), (30_000, 60_000)])
def test_select_in_range_between_resident_clouds(ctx, nf, nm):
    """sicp_select_in_range (both clouds resident, only the verdicts come back) == the oracle's bounded 1-NN
    from the same points handed over as host queries; strict bound; all points or a subset."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(nf + nm)
    F = rng.uniform(-20, 20, (nf, 3))
    M = rng.uniform(-20, 20, (nm, 3))
    M[: nm // 3, 0] += 25.0                               # part of the searched cloud out of reach
    ctx.upload(_lib.FIX, F)
    ctx.upload(_lib.MOV, M)
    r = 1.5 if nm > 1000 else 12.0
    for H in (None, _H(2)):
        want = orc.knn(M, F, k=1, H=H, max_dist=r)[0][:, 0] >= 0
        got = ctx.select_in_range(_lib.FIX, _lib.MOV, None, H, r)
        assert got.dtype == np.bool_ and np.array_equal(got, want) and (nf < 100 or 0 < want.sum() < nf)
        sel = np.unique(rng.integers(0, nf, max(1, nf // 3)))
        assert np.array_equal(ctx.select_in_range(_lib.FIX, _lib.MOV, sel, H, r), want[sel])
    assert not ctx.select_in_range(_lib.FIX, _lib.MOV, None, None, 0.0).any()
    assert ctx.select_in_range(_lib.FIX, _lib.MOV, None, None, np.inf).all()
    with pytest.raises(_lib.BackendError):
        ctx.select_in_range(_lib.FIX, _lib.MOV, np.array([nf]), None, 1.0)
    with pytest.raises(_lib.BackendError):
        ctx.select_in_range(_lib.MOV, _lib.MOV, None, None, 1.0)


@pytest.mark.parametrize("n,q,k", [(20, 5, 2), (5000, 300, 10), (5000, 300, 16), (30_000, 1100, 40),
                                   (3000, 64, 70), (100, 3, 100), (50, 4, 60)])
def test_knnk_bit_exact(ctx, n, q, k):
    from simpleicp_amd import _lib
    rng = np.random.default_rng(k)
    P = np.round(rng.uniform(-5, 5, (n, 3)), 2)   # quantised: plenty of exact ties
    Qp = P[rng.choice(n, q, replace=False)]
    ctx.upload(_lib.FIX, P)
    idx, d2 = ctx.knn(_lib.FIX, Qp, k=k)
    ridx, rd2 = orc.knn(P, Qp, k=k)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)


def test_upload_columns_equals_upload(ctx):
    """Column-wise upload (a frame whose x, y, z were assigned) fills the same device cloud as the (n,3) one."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(11)
    for n in (1, 1023, 1024, 70_001):
        X = rng.normal(size=(n, 3)) * 50
        q = rng.normal(size=(64, 3)) * 50
        ctx.upload(_lib.MOV, X)
        ref = ctx.knn(_lib.MOV, q, k=1)
        ctx.upload_columns(_lib.MOV, X[:, 0].copy(), X[:, 1].copy(), X[:, 2].copy())
        assert ctx.size(_lib.MOV) == n and np.array_equal(ctx.download(_lib.MOV), X)
        got = ctx.knn(_lib.MOV, q, k=1)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    with pytest.raises(ValueError):
        ctx.upload_columns(_lib.MOV, np.zeros(3), np.zeros(4), np.zeros(3))


@pytest.mark.parametrize("form", ["rows", "columns"])
def test_background_upload_equals_the_plain_one(ctx, form):
    """sicp_cloud_upload_start (ABI 7): the movable cloud travels on the library's helper thread while THIS thread builds the fixed
    cloud's grid and normals (what run() does, simpleicp.py:161-178); afterwards the slot holds exactly what the plain upload leaves
    -- coordinates, bounding box (the grid the match builds), statistics -- and searches answer bit for bit.  Sizes on both sides of the
    staged-upload limit (2^19 points: those are uploaded on the spot)."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(5)
    for n in (700_001, 40_000):
        F = rng.uniform(-20, 20, (n, 3)); F[:, 2] = 0.3 * np.sin(F[:, 0]) + 0.01 * rng.normal(size=n)
        M = F[rng.permutation(n)] + rng.normal(scale=0.01, size=(n, 3))
        sel = np.arange(0, n, 37)
        Qp = F[::211]
        ctx.upload(_lib.FIX, F)
        ctx.upload(_lib.MOV, M)
        ref_knn = ctx.knn(_lib.MOV, Qp, k=1, H=_H(3))
        ref_rows = ctx.download(_lib.MOV)
        ref_nv = ctx.estimate_normals(_lib.FIX, sel, 10)
        ctx.upload(_lib.MOV, F[:10])                           # (something else in the slot in between)
        if form == "rows":
            ctx.upload_start(_lib.MOV, xyz=M)
        else:
            ctx.upload_start(_lib.MOV, columns=[np.ascontiguousarray(M[:, j]) for j in range(3)])
        ctx.upload(_lib.FIX, F)                                # the other slot: upload, grid and normals beside the helper thread
        nv = ctx.estimate_normals(_lib.FIX, sel, 10)
        assert np.array_equal(nv[0], ref_nv[0], equal_nan=True) and np.array_equal(nv[1], ref_nv[1], equal_nan=True)
        got = ctx.knn(_lib.MOV, Qp, k=1, H=_H(3))              # (no explicit wait: the first call naming the slot joins)
        assert np.array_equal(got[0], ref_knn[0]) and np.array_equal(got[1], ref_knn[1])
        ctx.upload_wait(_lib.MOV)                              # idempotent
        assert ctx.size(_lib.MOV) == n and np.array_equal(ctx.download(_lib.MOV), ref_rows)


def test_background_upload_is_joined_by_every_entry_that_names_the_slot(ctx):
    """No explicit wait anywhere: transform, the planarity column, select_in_range (either role), the normals, a whole chained run
    and the download each find the cloud complete -- the same results as behind plain uploads."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(7)
    n = 640_000
    F = _surface(n, 21)
    M = orc.transform(np.linalg.inv(_H(5)), _surface(n, 22))
    sel = np.arange(0, n, 640)
    z = np.zeros(6)

    def flow(start):
        out = {}
        start(_lib.MOV, M)
        ctx.transform(_lib.MOV, _H(6))                                   # joins MOV
        out["moved"] = ctx.download(_lib.MOV)
        start(_lib.MOV, M)
        ctx.set_planarity(_lib.MOV, np.linspace(0, 1, n).astype(np.float32))
        start(_lib.FIX, F)
        out["near"] = ctx.select_in_range(_lib.FIX, _lib.MOV, sel, _H(5), 0.5)    # joins FIX (queries); MOV is there
        start(_lib.MOV, M)                                               # (drops the planarity column, like every upload)
        out["near2"] = ctx.select_in_range(_lib.FIX, _lib.MOV, sel, _H(5), 0.5)   # joins MOV (searched)
        start(_lib.FIX, F)
        nv, pl = ctx.estimate_normals(_lib.FIX, sel, 10)                 # joins FIX
        out["nv"], out["pl"] = nv, pl
        start(_lib.MOV, M)
        ctx.icp_setup(sel, nv, pl)                                       # (names FIX only)
        R = ctx.icp_run(z, z, z, 0.3, 1.0, max_iterations=3, min_change=0.0)      # joins MOV
        out["x"] = np.array([r.x[:] for r in R])
        out["state"] = ctx.icp_state()
        return out

    plain = flow(lambda slot, X: ctx.upload(slot, X))
    behind = flow(lambda slot, X: ctx.upload_start(slot, xyz=X))
    cols = flow(lambda slot, X: ctx.upload_start(slot, columns=[np.ascontiguousarray(X[:, j]) for j in range(3)]))
    for other in (behind, cols):
        for k, v in plain.items():
            if k == "state":
                assert all(np.array_equal(a, b) for a, b in zip(v, other[k]))
            else:
                assert np.array_equal(v, other[k], equal_nan=True), k
    assert np.array_equal(plain["moved"], orc.transform(_H(6), M))


def test_background_upload_hands_its_error_to_the_next_call_on_the_slot(ctx):
    """A non-finite movable cloud: the start returns, the verdict comes from upload_wait (or whatever names the slot next), the slot
    is empty afterwards and usable again; a second start waits for the first; the other slot is never disturbed."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(6)
    n = 600_000
    F = rng.normal(size=(n, 3))
    X = rng.normal(size=(n, 3)); X[n - 3, 2] = np.nan
    ctx.upload(_lib.FIX, F)
    ctx.upload_start(_lib.MOV, xyz=X)
    assert ctx.size(_lib.FIX) == n                             # (FIX calls do not wait and do not fail)
    with pytest.raises(_lib.BackendError, match="non-finite"):
        ctx.upload_wait(_lib.MOV)
    with pytest.raises(_lib.BackendError, match="empty"):
        ctx.knn(_lib.MOV, np.zeros((2, 3)), k=1)
    ctx.upload_start(_lib.MOV, xyz=X)
    with pytest.raises(_lib.BackendError, match="non-finite"):
        ctx.knn(_lib.MOV, np.zeros((2, 3)), k=1)               # the implicit join delivers it too
    good = rng.normal(size=(n, 3))
    ctx.upload_start(_lib.MOV, xyz=X)                          # a failing one in flight ...
    ctx.upload_start(_lib.FIX, xyz=good)                       # ... is waited for by a start on the other slot (one helper at a time), its verdict stays with ITS slot
    with pytest.raises(_lib.BackendError, match="non-finite"):
        ctx.size(_lib.MOV)
    assert ctx.knn(_lib.FIX, good[:5], k=1)[0][:, 0].tolist() == [0, 1, 2, 3, 4]
    ctx.upload_start(_lib.MOV, xyz=good)
    ctx.upload_start(_lib.FIX, xyz=F)                          # both arrive
    i0 = ctx.knn(_lib.MOV, good[:5], k=1)[0][:, 0]
    i1 = ctx.knn(_lib.FIX, F[:5], k=1)[0][:, 0]
    assert np.array_equal(i0, np.arange(5)) and np.array_equal(i1, np.arange(5))
    c2 = _lib.Context(0)                                       # a context destroyed with an upload in flight joins it first
    c2.upload_start(_lib.MOV, xyz=good)
    c2.close()


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_non_finite_clouds_are_rejected_at_upload(ctx, bad):
    """cKDTree refuses NaN / inf data (pointcloud.py:161,185 would raise); so does the upload, either flavour,
    and the slot is left empty instead of holding a cloud no search can be trusted on."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(0)
    for n in (10, 200_000):                                   # small (scan paths) and large (grid path)
        X = rng.normal(size=(n, 3))
        X[n // 2, 1] = bad
        with pytest.raises(_lib.BackendError, match="non-finite"):
            ctx.upload(_lib.MOV, X)
        with pytest.raises(_lib.BackendError, match="empty"):
            ctx.knn(_lib.MOV, X[:2] * 0, k=1)
        with pytest.raises(_lib.BackendError, match="non-finite"):
            ctx.upload_columns(_lib.MOV, X[:, 0].copy(), X[:, 1].copy(), X[:, 2].copy())
    ctx.upload(_lib.MOV, rng.normal(size=(10, 3)))           # and the slot is usable again
    assert ctx.knn(_lib.MOV, np.zeros((1, 3)), k=1)[0][0, 0] >= 0


def test_transform_bit_exact(ctx):
    from simpleicp_amd import _lib
    rng = np.random.default_rng(2)
    P = rng.uniform(-500, 500, (100_001, 3))
    H = _H(3)
    ctx.upload(_lib.MOV, P)
    ctx.transform(_lib.MOV, H)
    got = ctx.download(_lib.MOV)
    assert np.array_equal(got, orc.transform(H, P))
    # and that is numpy's own `H @ Xh.T` (pointcloud.py:205-217) on this image
    Xh = np.column_stack((P, np.ones(len(P))))
    ref = (H @ Xh.T).T
    assert np.array_equal(got, ref[:, :3] / ref[:, 3:4])


@pytest.mark.parametrize("name", ["dragon", "webots", "bunny"])
def test_normals_vs_oracle(ctx, name, clouds):
    from simpleicp_amd import _lib
    g, files, kw = load_golden(name)
    Xf = clouds(files[0])
    k = kw.get("neighbors", 10)
    sel = g["sel_idx"]
    ctx.upload(_lib.FIX, Xf)
    nv, pl, nn = ctx.estimate_normals(_lib.FIX, sel, k, want_nn=True)
    rnn, _ = orc.knn(Xf, Xf[sel], k=k)
    assert np.array_equal(nn, rnn)
    rnv, rpl = orc.normals(Xf, rnn)
    # same operation order on both sides; allow 1 ulp(float32) for the fp64 sqrt/div paths
    assert np.abs(nv - rnv).max() <= 2e-7
    assert np.abs(pl - rpl).max() <= 2e-6
    # and against the reference's normals (LAPACK) up to sign, where neighbour sets are untied
    dot = np.abs(np.sum(nv.astype(np.float64) * g["normals"], axis=1))
    assert np.mean(dot > 1 - 1e-5) > (0.70 if name == "webots" else 0.995)


@pytest.mark.parametrize("name", ["dragon", "bunny", "multisensor", "webots", "bunny_obs", "dragon_q5000"])
def test_iteration_vs_oracle(ctx, name, clouds):
    """Whole iterations (match -> reject -> LM on fused reductions) against the oracle, fed with
    the reference's normals; indices / masks bit-exact, parameters to 1e-9."""
    from simpleicp_amd import _lib
    g, files, kw = load_golden(name)
    Xf, Xm = clouds(files[0]), clouds(files[1])
    obs = np.array(kw.get("rbp_observed_values", (0.,) * 6), float)
    obs[:3] *= np.pi / 180
    ow = np.array(kw.get("rbp_observation_weights", (0.,) * 6), float)
    sel = g["sel_idx"]
    ctx.upload(_lib.FIX, Xf)
    ctx.upload(_lib.MOV, Xm)
    ctx.icp_setup(sel, g["normals"], g["planarity"])
    x = obs.copy()
    w = kw.get("distance_weights", 1)
    for it in range(min(int(g["iterations"]), 6)):
        R = ctx.icp_iterate(x, obs, ow, kw.get("min_planarity", 0.3), w)
        o = orc.icp_iteration(Xm, Xf[sel], g["normals"], g["planarity"], x, x, w, obs, ow, kw.get("min_planarity", 0.3))
        idx, dist, keep, resid = ctx.icp_state()
        assert np.array_equal(idx, o["nn"])
        assert np.array_equal(dist, o["dist"])               # contract (P): bit-exact
        assert np.array_equal(keep, o["keep"])
        assert R.n_kept == o["n"] and R.median == o["median"] and R.mad == o["mad"]
        xg = np.array(R.x[:])
        assert np.abs(xg - o["x"]).max() < 1e-9     # solver stops once the proposed undamped step is < 1e-10 (oracle iterates to 1e-13)
        assert np.allclose(resid[keep], orc.residuals(xg, Xf[sel], g["normals"], Xm[idx], keep), rtol=0, atol=1e-13)   # device sin/cos vs libm: ulp-level
        assert abs(R.res_mean - resid[keep].mean()) < 1e-15 and abs(R.res_std - resid[keep].std()) < 1e-14
        assert abs(R.dist_std - dist[keep].std()) < 1e-14
        ne = ctx.icp_normal_equations(xg)
        one = orc.normal_equations(xg, Xf[sel], g["normals"], Xm[idx], keep)
        assert np.allclose(ne, one, rtol=1e-12, atol=1e-13 * np.abs(one).max())   # J^T r ~ 0 at the optimum
        w = R.weight_used if w is None else w
        x = xg
    s = ctx.icp_uncertainties()
    so = orc.uncertainties(x, w, obs, ow, Xf[sel], g["normals"], Xm[idx], keep)
    free = np.isfinite(ow)
    assert np.allclose(s[free], so[free], rtol=1e-9) and np.all(np.isnan(s[~free]))


@pytest.mark.parametrize("name,w", [("dragon", 1.0), ("bunny", None)])
def test_icp_run_equals_iterate_loop(ctx, name, w, clouds):
    """sicp_icp_run (whole loop behind one call, iterations chained on the device) == the same loop driven from the
    host through sicp_icp_iterate: identical iteration count (same convergence test), identical correspondence
    counts, estimates equal to rounding (the chained loop carries sin / cos of the angles forward on the device
    with the addition theorem, the host-driven loop takes them from libm for every x it is handed)."""
    from simpleicp_amd import _lib
    g, files, kw = load_golden(name)
    Xf, Xm = clouds(files[0]), clouds(files[1])
    sel = g["sel_idx"]
    ctx.upload(_lib.FIX, Xf)
    ctx.upload(_lib.MOV, Xm)
    ctx.icp_setup(sel, g["normals"], g["planarity"])
    z = np.zeros(6)
    x, ww, loop = z.copy(), w, []
    for it in range(100):
        R = ctx.icp_iterate(x, z, z, 0.3, ww)
        loop.append(R)
        x = np.array(R.x[:])
        ww = R.weight_used if ww is None else ww
        ch = lambda a, b: abs((a - b) / b * 100)
        if it > 0 and ch(R.res_mean, loop[-2].res_mean) < 1 and ch(R.res_std, loop[-2].res_std) < 1:
            break
    whole = ctx.icp_run(z, z, z, 0.3, w, max_iterations=100, min_change=1.0)
    assert len(whole) == len(loop) and (w is None or len(loop) == int(g["iterations"]))
    for a, b in zip(whole, loop):
        assert np.abs(np.array(a.x[:]) - np.array(b.x[:])).max() < 1e-13 and np.abs(np.array(a.H[:]) - np.array(b.H[:])).max() < 1e-13
        assert a.n_kept == b.n_kept and abs(a.res_std - b.res_std) < 1e-13 and abs(a.weight_used - b.weight_used) <= 1e-12 * abs(b.weight_used)
    assert len(ctx.icp_run(z, z, z, max_iterations=3, min_change=0.0)) == 3
    assert ctx.icp_run(z, z, z, max_iterations=0) == []


def test_too_few_correspondences(ctx):
    from simpleicp_amd import _lib
    rng = np.random.default_rng(0)
    P = rng.uniform(0, 1, (100, 3))
    ctx.upload(_lib.FIX, P)
    ctx.upload(_lib.MOV, P + 0.01)
    sel = np.arange(10)
    pl = np.full(10, np.nan, np.float32)
    pl[:4] = 1.0
    ctx.icp_setup(sel, np.tile(np.float32([0, 0, 1]), (10, 1)), pl)
    with pytest.raises(_lib.BackendError) as e:
        ctx.icp_iterate(np.zeros(6), np.zeros(6), np.zeros(6))
    assert e.value.code == _lib.ERR_TOO_FEW and "Too few correspondences" in str(e.value)
    with pytest.raises(_lib.BackendError) as e:
        ctx.icp_run(np.zeros(6), np.zeros(6), np.zeros(6))
    assert e.value.code == _lib.ERR_TOO_FEW and len(e.value.results) == 1 and e.value.results[0].n_kept < 6


# ---- filtered scan (FP32 conservative filter + exact FP64 verification) == plain brute force ----
@pytest.fixture(scope="module", params=["filter", "grid", "grid16", "grid16far", "grid16exact"])
def ctx_filter(request):
    """Contexts that FORCE the filtered brute-force scan / the grid search (one query per wave with the cells' tight boxes, or the
    flavours large query sets get: the float32 filter -- lean kernel first, full kernel for what it leaves, exact kernel for the ties;
    the full kernel alone; the exact four-per-wave kernel) even for small clouds."""
    import os
    from simpleicp_amd import _lib
    env = {"SICP_KNN1": "grid" if request.param.startswith("grid") else request.param, "SICP_BOXES": "2"}
    if request.param.startswith("grid16"):
        env["SICP_NN16_MIN_Q"] = "1"
        env["SICP_ORDER_MIN_Q"] = "1"        # ... and the iteration's queries in cell order, one eighth per XCD
        env["SICP_COARSE_MIN_N"] = "1"       # ... and a cold iteration bounded by the subsample's nearest point whatever the cloud size
        env["SICP_NN16"] = {"grid16": "near", "grid16far": "far", "grid16exact": "exact"}[request.param]
        env["SICP_NN16F_MIN_Q"] = "1"        # ... through the float32 filter whatever the query count
        env["SICP_FAR_MOVE"] = "1e9"         # ... the lean flavour first in every iteration but the cold one
    os.environ.update(env)
    try:
        c = _lib.Context(0)
    finally:
        for k in env:
            del os.environ[k]
    c.mode = request.param
    yield c
    c.close()


def _surface(n, seed, L=None):
    rng = np.random.default_rng(seed)
    L = L or np.sqrt(n / 10.0)
    x, y = rng.uniform(0, L, n), rng.uniform(0, L, n)
    z = 20 * np.sin(2 * np.pi * x / 200) * np.cos(2 * np.pi * y / 300) + rng.normal(0, 0.02, n)
    return np.column_stack((x, y, z))


@pytest.mark.parametrize("case", ["uniform_small", "surface_1m", "offset_utm", "quantised_ties", "tiny_coords",
                                  "clustered", "q_gt_1024", "far_queries", "wide_balls", "degenerate_line", "identical_points",
                                  "single_point"])
def test_filtered_scan_equals_brute_force(ctx_filter, case):
    from simpleicp_amd import _lib
    rng = np.random.default_rng(11)
    H = _H(4)
    if case == "uniform_small":
        P, Qp = rng.uniform(-50, 50, (5000, 3)), rng.uniform(-50, 50, (300, 3))
    elif case == "surface_1m":
        P = _surface(1_000_000, 1); P -= P.mean(0)
        Qp = _surface(1_000_000, 2)[::1000] - _surface(1_000_000, 1).mean(0)
    elif case == "offset_utm":       # large common offset: the FP32 filter is nearly blind, still exact
        P = _surface(300_000, 3) + np.array([4.5e5, 5.2e6, 300.0])
        Qp = P[::300] + rng.normal(0, 0.05, (1000, 3))
    elif case == "quantised_ties":
        P = np.round(rng.uniform(-20, 20, (400_000, 3)), 1); P[200_000:] = P[:200_000]
        Qp = np.round(rng.uniform(-20, 20, (1000, 3)), 1)
    elif case == "tiny_coords":
        P, Qp = rng.uniform(-1e-6, 1e-6, (300_000, 3)), rng.uniform(-1e-6, 1e-6, (500, 3))
    elif case == "clustered":
        P = np.concatenate([rng.normal(c, 0.01, (100_000, 3)) for c in ((0, 0, 0), (100, 0, 0), (0, 1000, 5))])
        Qp = np.concatenate([rng.normal(c, 0.02, (300, 3)) for c in ((0, 0, 0), (100, 0, 0), (50, 500, 0))])
    elif case == "far_queries":      # queries far outside the cloud's box: expanding search / huge radii
        P = rng.uniform(-1, 1, (100_000, 3))
        Qp = np.concatenate((rng.uniform(-1, 1, (50, 3)), rng.uniform(500, 600, (50, 3)), [[1e6, -1e6, 3.0]]))
    elif case == "wide_balls":       # a cold iteration's searches: the answer lies several cells away, most rows of the ball are empty
        P = _surface(400_000, 8)
        Qp = _surface(400_000, 9)[::160] + np.column_stack((rng.uniform(-3, 3, 2500), rng.uniform(-3, 3, 2500), rng.uniform(1, 9, 2500)))
    elif case == "degenerate_line":  # zero extent on two axes
        P = np.zeros((70_000, 3)); P[:, 0] = np.round(rng.uniform(0, 100, 70_000), 2)
        Qp = np.column_stack((rng.uniform(-10, 110, 400), rng.normal(0, 1, 400), rng.normal(0, 1, 400)))
    elif case == "identical_points":
        P = np.tile([[1.5, -2.5, 3.25]], (5000, 1)); Qp = rng.uniform(-5, 5, (100, 3))
    elif case == "single_point":
        P = np.array([[1.0, 2.0, 3.0]]); Qp = rng.uniform(-5, 5, (10, 3))
    else:
        P, Qp = _surface(500_000, 5), _surface(500_000, 6)[::100]      # 5000 queries -> R = 8 blocks
    ctx_filter.upload(_lib.MOV, P)
    A = H.copy(); A[:3, :3] = A[:3, :3] @ np.diag([1.5, 0.7, 1.0]) + 0.01     # NOT rigid: must still be exact
    for Hm, md in ((None, np.inf), (H, np.inf), (None, 0.5), (H, 2.0), (A, np.inf)):
        idx, d2 = ctx_filter.knn(_lib.MOV, Qp, k=1, H=Hm, max_dist=md)
        ridx, rd2 = orc.knn(P, Qp, k=1, H=Hm, max_dist=md)
        assert np.array_equal(idx, ridx)
        assert np.array_equal(d2, rd2)


def test_filtered_iteration_uses_previous_match_bound(ctx_filter, clouds):
    """Iterations 2+ take the filter bound from the previous match; all of it must stay bit-exact."""
    from simpleicp_amd import _lib
    g, files, kw = load_golden("dragon")
    Xf, Xm = clouds(files[0]), clouds(files[1])
    sel = g["sel_idx"]
    ctx_filter.upload(_lib.FIX, Xf)
    ctx_filter.upload(_lib.MOV, Xm)
    ctx_filter.icp_setup(sel, g["normals"], g["planarity"])
    x = np.zeros(6)
    for it in range(5):
        R = ctx_filter.icp_iterate(x, np.zeros(6), np.zeros(6), 0.3, 1.0)
        o = orc.icp_iteration(Xm, Xf[sel], g["normals"], g["planarity"], x, x, 1.0, np.zeros(6), np.zeros(6), 0.3)
        idx, dist, keep, _ = ctx_filter.icp_state()
        assert np.array_equal(idx, o["nn"]) and np.array_equal(dist, o["dist"]) and np.array_equal(keep, o["keep"])
        x = np.array(R.x[:])
        assert np.abs(x - o["x"]).max() < 1e-9


@pytest.mark.parametrize("n,q,k", [(20, 5, 2), (5000, 300, 10), (30_000, 500, 40), (3000, 64, 70), (100, 3, 100),
                                   (200_000, 1000, 10), (50, 4, 60)])
def test_grid_knn_equals_brute_force(ctx_filter, n, q, k):
    """k-NN on the grid (forced by SICP_KNN1=grid; the `filter` context runs the brute-force kernels)."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(k + n)
    if n >= 100_000:
        P = _surface(n, 9)
    else:
        P = np.round(rng.uniform(-5, 5, (n, 3)), 2)          # quantised: plenty of exact ties
    Qp = P[rng.choice(n, q, replace=False)]
    ctx_filter.upload(_lib.FIX, P)
    idx, d2 = ctx_filter.knn(_lib.FIX, Qp, k=k)
    ridx, rd2 = orc.knn(P, Qp, k=k)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)


@pytest.mark.parametrize("quantised,odd", [(False, False), (True, False), (False, True)])
def test_large_q_iteration_multi_kernel_path(ctx, quantised, odd):
    """Q > 16384: multi-workgroup radix selection (median / MAD, even and odd counts, duplicate distances from a
    quantised cloud) + multi-block reductions + host LM must agree with the oracle."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(8)
    n = 60_000
    P = _surface(n, 21)
    x_true = np.array([0.002, -0.001, 0.003, 0.05, -0.03, 0.02])
    Xm = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), P + rng.normal(0, 0.01, P.shape))
    if quantised:
        P, Xm = np.round(P, 2), np.round(Xm, 2)
    sel = np.arange(0, n - (1 if odd else 0), 2)[: 30000 - (1 if odd else 0)]     # Q = 30000 / 29999
    ctx.upload(_lib.FIX, P)
    ctx.upload(_lib.MOV, Xm)
    nv, pl = ctx.estimate_normals(_lib.FIX, sel, 10)
    ctx.icp_setup(sel, nv, pl)
    x = np.zeros(6)
    for it in range(3):
        R = ctx.icp_iterate(x, np.zeros(6), np.zeros(6), 0.3, 1.0)
        o = orc.icp_iteration(Xm, P[sel], nv, pl, x, x, 1.0, np.zeros(6), np.zeros(6), 0.3)
        idx, dist, keep, resid = ctx.icp_state()
        assert np.array_equal(idx, o["nn"]) and np.array_equal(dist, o["dist"]) and np.array_equal(keep, o["keep"])
        assert R.n_kept == o["n"] and R.median == o["median"] and R.mad == o["mad"]
        x = np.array(R.x[:])
        assert np.abs(x - o["x"]).max() < 1e-9
        # multi-workgroup two-pass statistics (count / mean / population std) against numpy's
        assert abs(R.dist_mean - dist[keep].mean()) < 1e-15 and abs(R.dist_std - dist[keep].std()) < 1e-14
        assert abs(R.res_mean - resid[keep].mean()) < 1e-15 and abs(R.res_std - resid[keep].std()) < 1e-14


@pytest.mark.parametrize("variant,cap", [("inline", None), ("record", 1), ("record", None)])
def test_filtered_scan_variants_and_overflow_fallback(variant, cap):
    """Both filtered-scan kernels (exact work inline / recorded + fixed up) and the overflow fallback (candidate lists
    forced to one entry per query) return the brute-force answer."""
    import os
    from simpleicp_amd import _lib
    env = {"SICP_KNN1": "filter", "SICP_FSCAN": variant}
    if cap:
        env["SICP_FSCAN_CAP"] = str(cap)
    os.environ.update(env)
    try:
        c = _lib.Context(0)
    finally:
        for k in env:
            del os.environ[k]
    P = _surface(300_000, 31)
    Qp = _surface(300_000, 32)[::200]
    c.upload(_lib.MOV, P)
    for Hm in (None, _H(7)):
        idx, d2 = c.knn(_lib.MOV, Qp, k=1, H=Hm)
        ridx, rd2 = orc.knn(P, Qp, k=1, H=Hm)
        assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
        if cap is None:
            assert c.last_match_kernel() == {"inline": "k_knn1_fscan", "record": "k_knn1_frec"}[variant]
    c.close()


@pytest.mark.parametrize("Q", [1, 6, 63, 512, 513, 1024, 1025, 1500, 2048, 2049, 5000, 10_000, 16_384, 16_385, 40_000])
@pytest.mark.parametrize("quantised", [False, True])
def test_iteration_q_sweep_every_tail_path(ctx, Q, quantised):
    """Every instantiation of the iteration's tail against the oracle with bit-level assertions: the fused
    single-launch tail with 1 / 2 / 4 correspondences per lane (Q <= 512 / 1024 / 2048), the single-workgroup
    LDS selection (2048 < Q <= 16384) and the multi-workgroup digit selection above (40 000: with the four-queries-per-wave
    search over cell-ordered queries that large sets get by default), including both hand-over points and even / odd
    survivor counts; a quantised cloud supplies duplicate distances.  Also
    sicp_icp_uncertainties on each path (optimization.py:126-170)."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(Q)
    n = 70_000
    P = _surface(n, 40 + (Q % 7))
    x_true = np.array([0.002, -0.001, 0.003, 0.05, -0.03, 0.02])
    Xm = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), P + rng.normal(0, 0.01, P.shape))
    if quantised:
        P, Xm = np.round(P, 2), np.round(Xm, 2)
    sel = np.sort(rng.choice(n, Q, replace=False))
    ctx.upload(_lib.FIX, P)
    ctx.upload(_lib.MOV, Xm)
    nv, pl = ctx.estimate_normals(_lib.FIX, sel, 10)
    if Q >= 6:
        pl[:: max(1, Q // 5)] = np.nan                      # a few rows without planarity (corrpts.py:153-155)
    ctx.icp_setup(sel, nv, pl)
    z = np.zeros(6)
    x, w = z.copy(), None
    for it in range(3):
        o = orc.icp_iteration(Xm, P[sel], nv, pl, x, x, w, z, z, 0.3) if Q >= 6 else None
        if o is None or o["n"] < 6:
            with pytest.raises(_lib.BackendError) as e:
                ctx.icp_iterate(x, z, z, 0.3, w)
            assert e.value.code == _lib.ERR_TOO_FEW
            return
        R = ctx.icp_iterate(x, z, z, 0.3, w)
        idx, dist, keep, resid = ctx.icp_state()
        assert np.array_equal(idx, o["nn"]) and np.array_equal(dist, o["dist"]) and np.array_equal(keep, o["keep"])
        assert R.n_kept == o["n"] and R.median == o["median"] and R.mad == o["mad"]
        assert R.n_planar == int(np.count_nonzero(pl >= np.float32(0.3)))
        x = np.array(R.x[:])
        assert np.abs(x - o["x"]).max() < 1e-9
        assert abs(R.weight_used - o["w"]) <= 1e-12 * abs(o["w"])
        w = R.weight_used
        assert abs(R.dist_mean - dist[keep].mean()) < 1e-15 and abs(R.dist_std - dist[keep].std()) < 1e-14
        assert abs(R.res_mean - resid[keep].mean()) < 1e-15 and abs(R.res_std - resid[keep].std()) < 1e-14
        assert np.allclose(resid[keep], orc.residuals(x, P[sel], nv, Xm[idx], keep), rtol=0, atol=1e-13)
    s = ctx.icp_uncertainties()
    so = orc.uncertainties(x, w, z, z, P[sel], nv, Xm[idx], keep)
    assert np.allclose(s, so, rtol=1e-9)


@pytest.mark.parametrize("name", GOLDEN_CHAIN)
def test_iteration_with_movable_selection_and_planarity(ctx, name, clouds):
    """corrpts.py:131-135 (only pc2's selected points are searched) and :158-163 (pc2's planarity column filters
    too): the searched cloud is the selected subset, sicp_cloud_set_planarity carries its column; indices (mapped
    back through the subset), distances and masks bit-exact against the oracle, which reproduces the reference's
    own trace of these runs on the CPU (tests/test_oracle_golden.py)."""
    from simpleicp_amd import _lib
    g, files, kw = load_golden(name)
    Xf, Xm = clouds(files[0]), clouds(files[1])
    msel, pl2 = movable_columns(g, len(Xm))
    rows = np.arange(len(Xm)) if msel is None else msel
    sel = g["sel_idx"]
    z = np.zeros(6)
    ctx.upload(_lib.FIX, Xf)
    ctx.upload(_lib.MOV, Xm[rows])
    sub = pl2[rows]
    at = np.flatnonzero(~np.isnan(sub))
    for dense in (False, True):
        if dense:
            ctx.set_planarity(_lib.MOV, sub)
        else:
            ctx.set_planarity(_lib.MOV, sub[at], rows=at)
        ctx.icp_setup(sel, g["normals"], g["planarity"])
        x = z.copy()
        for it in range(4):
            R = ctx.icp_iterate(x, z, z, 0.3, 1.0)
            o = orc.icp_iteration(Xm, Xf[sel], g["normals"], g["planarity"], x, x, 1.0, z, z, 0.3, mov_sel=msel, planarity_mov=pl2)
            idx, dist, keep, resid = ctx.icp_state()
            assert np.array_equal(rows[idx], o["nn"]) and np.array_equal(dist, o["dist"]) and np.array_equal(keep, o["keep"])
            assert R.n_kept == o["n"] and R.median == o["median"] and R.mad == o["mad"]
            x = np.array(R.x[:])
            assert np.abs(x - o["x"]).max() < 1e-9
    # without the column the filter is off again (and an upload clears it)
    ctx.set_planarity(_lib.MOV, None)
    R0 = ctx.icp_iterate(z, z, z, 0.3, 1.0)
    assert R0.n_planar == int(np.count_nonzero(g["planarity"] >= np.float32(0.3)))
    ctx.set_planarity(_lib.MOV, sub)
    ctx.upload(_lib.MOV, Xm[rows])
    assert ctx.icp_iterate(z, z, z, 0.3, 1.0).n_planar == R0.n_planar
    with pytest.raises(_lib.BackendError):
        ctx.set_planarity(_lib.MOV, np.zeros(3, np.float32), rows=np.array([0, 1, len(rows)]))
    with pytest.raises(_lib.BackendError):
        ctx.icp_setup(np.array([0, len(Xf)]), np.zeros((2, 3), np.float32), np.zeros(2, np.float32))
    with pytest.raises(_lib.BackendError):
        ctx.estimate_normals(_lib.FIX, np.array([-1, 3]), 5)


@pytest.mark.parametrize("Q", [1500, 9000, 20_000])
@pytest.mark.parametrize("layers", [1, 2, 3])
def test_rejection_with_massive_duplicate_distances(ctx, Q, layers):
    """Thousands of EXACTLY equal distances (planes at exact offsets, exact normals): every selection flavour must walk its
    digits / bins down to a single key value and still get rank, second middle value and count right -- the fused tail
    (Q <= 2048), the one-workgroup LDS selection, and the multi-workgroup digit selection whose survivors no longer fit its
    candidate list (> 256 equal keys).  All parameters fixed (infinite weights): the iteration is match + rejection only."""
    from simpleicp_amd import _lib
    g = np.arange(300) * 0.1
    P = np.column_stack([a.ravel() for a in np.meshgrid(g, g)] + [np.zeros(90_000)])          # lattice plane z = 0
    off = np.array([0.5, 0.25, 1.0])[:layers]
    z = off[np.minimum((P[:, 0] / 30.0 * layers).astype(int), layers - 1)]                      # 1, 2 or 3 exact offsets, by strip
    Xm = P + np.column_stack((np.full(len(P), 0.003), np.full(len(P), 0.002), z))
    rng = np.random.default_rng(Q + layers)
    sel = np.sort(rng.choice(len(P), Q, replace=False))
    nv = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (Q, 1))
    pl = np.ones(Q, dtype=np.float32)
    pl[::7] = 0.1                                                                               # some rows fail the planarity test
    ctx.upload(_lib.FIX, P)
    ctx.upload(_lib.MOV, Xm)
    ctx.icp_setup(sel, nv, pl)
    z6, fixed = np.zeros(6), np.full(6, np.inf)
    o = orc.icp_iteration(Xm, P[sel], nv, pl, z6, z6, 1.0, z6, fixed, 0.3)
    R = ctx.icp_iterate(z6, z6, fixed, 0.3, 1.0)
    idx, dist, keep, _ = ctx.icp_state()
    assert np.array_equal(idx, o["nn"]) and np.array_equal(dist, o["dist"])
    assert len(np.unique(dist)) == layers                                                       # the duplicates are real
    assert R.median == o["median"] and R.mad == o["mad"] and R.n_kept == o["n"]
    assert np.array_equal(keep, o["keep"])
    assert np.array_equal(np.array(R.x[:]), z6)


@pytest.mark.parametrize("Q", [16_385, 40_000])
@pytest.mark.parametrize("quantised", [False, True])
def test_one_launch_forms_equal_launch_per_phase_forms(Q, quantised):
    """The large-Q minimisation as ONE launch (k_lm_all: evaluations as phases meeting at grid barriers) against its
    launch-per-evaluation form (SICP_LM=launches: k_lm_eval x E + k_lm_finish -- what a sharded 6x6 reduction runs): the same
    sums in the same order -- estimate, residuals and statistics bit for bit, over three chained iterations and through
    sicp_icp_run.  (The rejection's launch-per-phase twin was removed in round 4; its windowed and general forms are held
    against each other and the oracle in test_windowed_rejection_equals_the_general_form.)"""
    import os
    from simpleicp_amd import _lib
    rng = np.random.default_rng(Q + 17)
    n = 70_000
    P = _surface(n, 43)
    x_true = np.array([0.002, -0.001, 0.003, 0.05, -0.03, 0.02])
    Xm = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), P + rng.normal(0, 0.01, P.shape))
    if quantised:
        P, Xm = np.round(P, 2), np.round(Xm, 2)                 # thousands of exactly equal distances
    sel = np.sort(rng.choice(n, Q, replace=False))
    z = np.zeros(6)
    out = {}
    for form in ("one", "launches"):
        if form == "launches":
            os.environ["SICP_LM"] = "launches"
        try:
            c = _lib.Context(0)
        finally:
            os.environ.pop("SICP_LM", None)
        with c:
            c.upload(_lib.FIX, P); c.upload(_lib.MOV, Xm)
            nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
            c.icp_setup(sel, nv, pl)
            x, rec = z.copy(), []
            for it in range(3):
                R = c.icp_iterate(x, z, z, 0.3, 1.0)
                idx, dist, keep, resid = c.icp_state()
                rec.append((np.array(R.x[:]), R.n_kept, R.median, R.mad, R.dist_mean, R.dist_std, R.res_mean, R.res_std, keep, resid, R.ne_evals))
                x = np.array(R.x[:])
            c.icp_setup(sel, nv, pl)
            whole = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=4, min_change=0.0)
            rec.append(tuple(np.array(w.x[:]) for w in whole))
        out[form] = rec
    for a, b in zip(out["one"][:3], out["launches"][:3]):
        assert np.array_equal(a[0], b[0]) and a[1:8] == b[1:8] and np.array_equal(a[8], b[8]) and np.array_equal(a[9], b[9]) and a[10] == b[10]
    assert all(np.array_equal(p, q) for p, q in zip(out["one"][3], out["launches"][3]))


@pytest.mark.parametrize("quantised", [False, True])
def test_many_queries_search_flavours_agree(quantised):
    """The many-queries search in every flavour -- through the float32 filter (lean kernel + full kernel + exact kernel for the
    ties; the full kernel alone), 16 or 8 lanes per query, with and without the cells' tight boxes, and the exact four-per-wave
    kernel it replaced -- against each other and the oracle: same indices, same distances, bit for bit, over chained iterations
    (cold, loosely and tightly bounded searches), on a cloud with exact ties (quantised: every query ties -- the exact kernel
    answers them all)."""
    import os
    from simpleicp_amd import _lib
    rng = np.random.default_rng(99)
    n, Q = 90_000, 40_000
    P = _surface(n, 47)
    x_true = np.array([0.004, -0.003, 0.006, 0.25, -0.15, 0.1])       # far enough off that early searches are wide
    Xm = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), P + rng.normal(0, 0.01, P.shape))
    if quantised:
        P, Xm = np.round(P, 2), np.round(Xm, 2)
    sel = np.sort(rng.choice(n, Q, replace=False))
    z = np.zeros(6)
    out = {}
    # (mode, lanes per query, boxes, cells the estimate may move per iteration before every search goes to the full flavour)
    flavours = [("near", "16", "1", "1e9"), ("near", "8", "1", "1e9"), ("near", "8", "1", "0.75"), ("far", "16", "1", "0.75"), ("far", "8", "0", "0.75"),
                ("near", "16", "0", "1e9"), ("exact", "16", "1", "0.75"), ("exact", "8", "1", "0.75")]
    for mode, gs, boxes, far_move in flavours:
        env = {"SICP_NN_GROUP": gs, "SICP_NN16": mode, "SICP_BOXES": boxes, "SICP_NN16F_MIN_Q": "1", "SICP_FAR_MOVE": far_move}
        os.environ.update(env)
        try:
            c = _lib.Context(0)
        finally:
            for k in env:
                os.environ.pop(k, None)
        with c:
            c.upload(_lib.FIX, P); c.upload(_lib.MOV, Xm)
            nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
            c.icp_setup(sel, nv, pl)
            x, rec = z.copy(), []
            for it in range(4):
                R = c.icp_iterate(x, z, z, 0.3, 1.0)
                assert c.last_match_kernel() == ("k_grid_nn16" if mode == "exact" else "k_grid_nn16f")
                idx, dist, keep, _ = c.icp_state(residual=False)
                rec.append((x.copy(), idx, dist, keep, np.array(R.x[:])))
                x = np.array(R.x[:])
            # ... and the chained loop (iterations enqueued back to back: the slots' bounds travel from launch to launch)
            c.icp_setup(sel, nv, pl)
            whole = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=6, min_change=0.0)
            idx, dist, keep, _ = c.icp_state(residual=False)
            rec.append((tuple(tuple(w.x[:]) for w in whole), idx, dist, keep))
        out[(mode, gs, boxes, far_move)] = rec
    ref = out[("exact", "16", "1", "0.75")]
    for key, rec in out.items():
        for a, b in zip(ref[:4], rec[:4]):
            assert all(np.array_equal(u, v) for u, v in zip(a, b)), key
        assert ref[4][0] == rec[4][0] and all(np.array_equal(u, v) for u, v in zip(ref[4][1:], rec[4][1:])), key
    for x, idx, dist, keep, xn in ref[:2]:
        nn, _ = orc.knn(Xm, P[sel], k=1, H=orc.params_to_H(x))
        assert np.array_equal(idx, nn[:, 0])


def test_filter_slots_follow_the_setup_through_the_operator_route():
    """The filtered search keeps the queries (and each one's last match) in slot order between iterations.  A NEW setup with as many
    queries, then a match through the operator route (sicp_corr_match marks "there is an earlier match"), then a fused iteration: the slots
    must be the new setup's, not the earlier one's; likewise a new movable cloud between two iterations."""
    import os
    from simpleicp_amd import _lib
    rng = np.random.default_rng(7)
    n, Q = 60_000, 9_000
    P = _surface(n, 3)
    Xm = orc.transform(np.linalg.inv(orc.params_to_H(np.array([0.002, -0.001, 0.003, 0.05, -0.03, 0.02]))), P + rng.normal(0, 0.01, P.shape))
    sel_a = np.sort(rng.choice(n, Q, replace=False))
    sel_b = np.sort(rng.choice(n, Q, replace=False))
    z = np.zeros(6)
    env = {"SICP_NN16F_MIN_Q": "1", "SICP_NN16_MIN_Q": "1"}
    os.environ.update(env)
    try:
        c = _lib.Context(0)
    finally:
        for k in env:
            os.environ.pop(k, None)
    with c:
        c.upload(_lib.FIX, P); c.upload(_lib.MOV, Xm)
        nva, pla = c.estimate_normals(_lib.FIX, sel_a, 10)
        nvb, plb = c.estimate_normals(_lib.FIX, sel_b, 10)
        c.icp_setup(sel_a, nva, pla)
        c.icp_iterate(z, z, z, 0.3, 1.0)
        assert c.last_match_kernel() == "k_grid_nn16f"
        c.icp_setup(sel_b, nvb, plb)                      # same count, other queries
        idx_op, _ = c.corr_match()
        c.icp_iterate(z, z, z, 0.3, 1.0)
        assert c.last_match_kernel() == "k_grid_nn16f"
        idx, dist, keep, _ = c.icp_state(residual=False)
        nn, _ = orc.knn(Xm, P[sel_b], k=1)
        assert np.array_equal(idx_op, nn[:, 0]) and np.array_equal(idx, nn[:, 0])
        # a new movable cloud: the bounds kept by slot were points of the old one
        Xm2 = Xm[::-1].copy() + np.array([0.3, -0.2, 0.1])
        c.upload(_lib.MOV, Xm2)
        idx_op, _ = c.corr_match()
        c.icp_iterate(z, z, z, 0.3, 1.0)
        idx, _, _, _ = c.icp_state(residual=False)
        nn2, _ = orc.knn(Xm2, P[sel_b], k=1)
        assert np.array_equal(idx_op, nn2[:, 0]) and np.array_equal(idx, nn2[:, 0])


# ---- the one-sweep k-NN + covariance kernel (k_grid_knn_sweep) == k extraction rounds + k_normals == oracle ----
def _knn_cases():
    rng = np.random.default_rng(2024)
    surf = _surface(60_000, 5)
    line = np.zeros((4000, 3)); line[:, 0] = np.round(rng.uniform(0, 50, 4000), 3)
    clustered = np.concatenate([rng.normal(c, 0.05, (3000, 3)) for c in rng.uniform(-20, 20, (6, 3))] + [rng.uniform(-30, 30, (2000, 3))])
    dup = np.concatenate((np.tile(np.array([[1.0, 2.0, 3.0]]), (3000, 1)), rng.uniform(-1, 5, (3000, 3))))       # 3000 coincident points
    return {
        "surface": (surf, 10), "surface_k40": (surf, 40), "surface_k128": (surf[:20_000], 128), "surface_k129": (surf[:20_000], 129),
        "quantised_ties": (np.round(rng.uniform(-5, 5, (20_000, 3)), 1), 10),
        "line": (line, 10), "clustered": (clustered, 16), "coincident": (dup, 10), "coincident_k70": (dup, 70),
        "utm_offset": (surf[:30_000] + np.array([4.3e5, 5.2e6, 300.0]), 10),
        "fewer_points_than_k": (rng.uniform(0, 1, (7, 3)), 10),
        "volume": (rng.uniform(0, 30, (50_000, 3)), 12),
    }


@pytest.mark.parametrize("case", list(_knn_cases()))
@pytest.mark.parametrize("batch,ordered,group", [("0", False, "1"), ("64", True, "1"), ("5", True, "1"), ("0", True, "4"), ("16", False, "4"),
                                                 ("3", True, "4")])
def test_knn_sweep_equals_rounds_and_oracle(case, batch, ordered, group):
    """sicp_knn(k > 1) and sicp_estimate_normals through the one-sweep kernels -- one query per wave (batches of 1 / 5 / 64 queries in
    cell order whose starting radius follows the k-th distances met so far), and FOUR queries per wave (16 lanes each: the common
    case; what it leaves -- short balls, dense clusters, coincident points, k > 32 -- goes through the one-query-per-wave kernel in
    a second launch) -- against the k-round search + k_normals (SICP_KNN_SWEEP=0) and the oracle: indices and squared distances
    bit for bit, normals / planarity bit for bit against the other kernel pair and to 1 ulp(f32) against the oracle
    (pointcloud.py:185-203)."""
    import os
    from simpleicp_amd import _lib
    P, k = _knn_cases()[case]
    n = len(P)
    rng = np.random.default_rng(k + n)
    sel = np.sort(rng.choice(n, min(n, 700), replace=False))
    out = {}
    for sweep in ("1", "0"):
        env = {"SICP_KNN1": "grid", "SICP_KNN_SWEEP": sweep, "SICP_KNN_BATCH": batch, "SICP_ORDER_MIN_Q": "1" if ordered else "0",
               "SICP_KNN_GROUP": group}
        os.environ.update(env)
        try:
            c = _lib.Context(0)
        finally:
            for key in env:
                del os.environ[key]
        with c:
            c.upload(_lib.FIX, P)
            c.timing_enable(True, count_work=True); c.timing_reset()
            idx, d2 = c.knn(_lib.FIX, P[sel], k=k)
            if k <= n:
                nv, pl, nn = c.estimate_normals(_lib.FIX, sel, k, want_nn=True)
                nv2, pl2 = c.estimate_normals(_lib.FIX, sel, k)           # (no index lists leave the kernel)
                assert np.array_equal(nv, nv2, equal_nan=True) and np.array_equal(pl, pl2, equal_nan=True)
            else:
                nv = pl = nn = None
            out[sweep] = (idx, d2, nv, pl, nn, c.knn_work())
    ridx, rd2 = orc.knn(P, P[sel], k=k)
    for key in ("1", "0"):
        idx, d2, nv, pl, nn, work = out[key]
        assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
        if nn is not None:
            assert np.array_equal(nn, ridx)
    w = out["1"][5]
    if k <= 128:
        assert w["sweeps"] >= len(sel) and w["candidates"] > 0, w           # the sweep kernel ran ...
        if case.startswith("coincident"):
            assert w["slow_queries"] > 0, w                                    # ... and its k-round path where the LDS cannot hold the ball
    else:
        assert w["sweeps"] == 0, w                                            # k > 128: the k-round kernel
    assert out["0"][5]["sweeps"] == 0
    if k <= n:
        a, b = out["1"], out["0"]
        assert np.array_equal(a[2], b[2], equal_nan=True) and np.array_equal(a[3], b[3], equal_nan=True)
        rnv, rpl = orc.normals(P, ridx)
        ok = np.isfinite(rpl)
        assert np.abs(a[2] - rnv)[np.isfinite(rnv)].max() <= 2e-7
        assert np.abs(a[3] - rpl)[ok].max() <= 2e-6 * max(1.0, np.abs(rpl[ok]).max())


@pytest.mark.parametrize("fault", ["1", "2"])
def test_grid_barrier_timeout_is_an_error_not_a_hang(fault):
    """The one-launch rejection (k_hsel_all) and minimisation (k_lm_all) meet at a grid barrier that is only correct when every block
    is resident.  SICP_TEST_BARRIER_FAULT makes the barrier of one of them expect a block that never arrives: the bounded wait must
    end in an error the caller sees (sicp_lanes.h: GridBar::error -> record status 4 -> SICP_ERR_HIP), the run must stop, later
    phases must not each wait out the limit again, and the context must stay usable (the barrier state is reset, nothing sticks)."""
    import os
    import time
    from simpleicp_amd import _lib
    rng = np.random.default_rng(5)
    n, Q = 60_000, 40_000                                           # > 16 384 correspondences: the many-workgroup path
    P = _surface(n, 31)
    Xm = orc.transform(np.linalg.inv(orc.params_to_H(np.array([0.002, -0.001, 0.003, 0.05, -0.03, 0.02]))), P + rng.normal(0, 0.01, P.shape))
    sel = np.sort(rng.choice(n, Q, replace=False))
    z = np.zeros(6)
    os.environ["SICP_TEST_BARRIER_FAULT"] = fault
    try:
        c = _lib.Context(0)
    finally:
        del os.environ["SICP_TEST_BARRIER_FAULT"]
    with c:
        c.upload(_lib.FIX, P); c.upload(_lib.MOV, Xm)
        nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
        for attempt in range(2):                                    # twice: the first failure must not poison the second call
            c.icp_setup(sel, nv, pl)
            t0 = time.perf_counter()
            with pytest.raises(_lib.BackendError) as e:
                if attempt == 0:
                    c.icp_iterate(z, z, z, 0.3, 1.0)
                else:
                    c.icp_run(z, z, z, 0.3, 1.0, max_iterations=5, min_change=0.0)
            assert e.value.code == _lib.ERR_HIP and "barrier" in str(e.value)
            assert time.perf_counter() - t0 < 20.0                  # one bounded wait, not one per phase and launch
        # the same context on the path without grid barriers still works, and agrees with the oracle
        small = sel[:1000]
        c.icp_setup(small, nv[:1000], pl[:1000])
        R = c.icp_iterate(z, z, z, 0.3, 1.0)
        o = orc.icp_iteration(Xm, P[small], nv[:1000], pl[:1000], z, z, 1.0, z, z, 0.3)
        assert np.abs(np.array(R.x[:]) - o["x"]).max() < 1e-9
    # and a context without the fault runs the same iteration through the barriers
    with _lib.Context(0) as c2:
        c2.upload(_lib.FIX, P); c2.upload(_lib.MOV, Xm)
        c2.icp_setup(sel, nv, pl)
        R = c2.icp_iterate(z, z, z, 0.3, 1.0)
        assert R.n_kept > 6


@pytest.mark.parametrize("n", [1, 7, 524_288, 524_289, 1_600_003])
def test_download_both_equals_the_two_downloads(ctx, n):
    """sicp_cloud_download_both (pinned double buffer, host threads fan out and transpose) == sicp_cloud_download and
    sicp_cloud_download_columns, bit for bit, across its 512 Ki-point chunk boundaries; each destination also on its own."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(n)
    X = rng.uniform(-1e3, 1e3, (n, 3))
    ctx.upload(_lib.MOV, X)
    ctx.transform(_lib.MOV, _H(3))
    rows, cols = ctx.download_both(_lib.MOV)
    assert np.array_equal(rows, ctx.download(_lib.MOV))
    for a, b in zip(cols, ctx.download_columns(_lib.MOV)):
        assert np.array_equal(a, b)
    only_rows = np.empty((n, 3))
    ctx._chk(ctx._L.sicp_cloud_download_both(ctx._h, _lib.MOV, _lib._ptr(only_rows), None, None, None))
    assert np.array_equal(only_rows, rows)
    x, y, z = np.empty(n), np.empty(n), np.empty(n)
    ctx._chk(ctx._L.sicp_cloud_download_both(ctx._h, _lib.MOV, None, _lib._ptr(x), _lib._ptr(y), _lib._ptr(z)))
    assert np.array_equal(np.column_stack((x, y, z)), rows)
    with pytest.raises(_lib.BackendError):
        ctx._chk(ctx._L.sicp_cloud_download_both(ctx._h, _lib.MOV, None, _lib._ptr(x), None, None))


@pytest.mark.parametrize("Q,kind", [(40_000, "plain"), (40_000, "quantised"), (40_000, "layers"), (150_000, "plain")])
def test_windowed_rejection_equals_the_general_form(Q, kind):
    """Large-Q rejection, windowed form (three grid barriers: a linear histogram around the previous iteration's median, the
    median bin's and the MAD shells' keys collected and sorted, premise checked on the keys -- tried from a run's third
    iteration on) against the general digit selection (SICP_HSEL_WINDOW=0) and the oracle: median, MAD, counts and keep masks
    bit for bit over ten chained iterations, on plain data, on quantised data (thousands of exactly equal distances) and on
    data whose distances sit in a few layers (windows that must miss and fall back)."""
    import os
    from simpleicp_amd import _lib
    rng = np.random.default_rng(Q + len(kind))
    n = max(2 * Q, 120_000)
    P = _surface(n, 77)
    x_true = np.array([0.003, -0.002, 0.004, 0.2, -0.1, 0.08])
    noise = rng.normal(0, 0.01, P.shape)
    if kind == "layers":
        noise[:, 2] = rng.choice([-0.03, 0.0, 0.03], len(P))          # distances in three thin layers: holes where the MAD lies
    Xm = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), P + noise)
    if kind == "quantised":
        P, Xm = np.round(P, 2), np.round(Xm, 2)
    sel = np.sort(rng.choice(n, Q, replace=False))
    z = np.zeros(6)
    out = {}
    for window in ("1", "0"):
        os.environ["SICP_HSEL_WINDOW"] = window
        try:
            c = _lib.Context(0)
        finally:
            del os.environ["SICP_HSEL_WINDOW"]
        with c:
            c.upload(_lib.FIX, P); c.upload(_lib.MOV, Xm)
            nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
            c.icp_setup(sel, nv, pl)
            whole = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=10, min_change=0.0)
            _, _, keep, resid = c.icp_state()
            # ... and a host-driven loop on the same context (one launch per call; priors carried over from the run above)
            x, host = z.copy(), []
            c.icp_setup(sel, nv, pl)
            for it in range(5):
                R = c.icp_iterate(x, z, z, 0.3, 1.0)
                host.append((R.n_planar, R.median, R.mad, R.n_kept, tuple(R.x[:])))
                x = np.array(R.x[:])
            out[window] = ([(w.n_planar, w.median, w.mad, w.n_kept, w.dist_mean, w.dist_std, tuple(w.x[:])) for w in whole], keep, resid, host)
    a, b = out["1"], out["0"]
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
    # the oracle on the host-driven iterations (every one of them a launch that may use the window: the run before left its priors;
    # the chained run carries sin / cos forward on the device, so only the host-driven loop meets the oracle's H bit for bit)
    x = z.copy()
    for it in range(3 if Q <= 40_000 else 1):            # (the oracle's brute-force match: 4.5e10 pairs per iteration at 150 000 queries)
        o = orc.icp_iteration(Xm, P[sel], nv, pl, x, x, 1.0, z, z, 0.3)
        assert (a[3][it][1], a[3][it][2], a[3][it][3]) == (o["median"], o["mad"], int(o["keep"].sum()))
        x = np.array(a[3][it][4])


def test_rejection_longest_barrier_road_over_chained_launches():
    """The one-launch rejection's LONGEST road, several launches in a row (barrier numbers are absolute: a launch that goes
    through more barriers than the host books for it leaves the next launch's first barrier open).  Distances are exact values:
    1 500 equal keys at the median (the window's analysis accepts the bin, but one block alone finds more of them than its list
    holds: the window misses after its SECOND sweep, three barriers), and 1 400 equal keys at the MAD's rank -- so the general
    form that follows walks all six digit passes for both statistics (more than 256 equal keys never fit the candidate list):
    3 + 2 x 7 = 17 barriers per launch from the third launch of a run on.  All parameters fixed (infinite weights): every
    iteration sees the same distances, so every launch must return the oracle's numbers."""
    from simpleicp_amd import _lib
    Q = 40_000
    rng = np.random.default_rng(4242)
    g = np.arange(300) * 5.0                                       # lattice spacing 5: a point's nearest neighbour is its own copy
    P = np.column_stack([a.ravel() for a in np.meshgrid(g, g)] + [np.zeros(90_000)])
    sel = np.sort(rng.choice(len(P), Q, replace=False))
    c, D = 0.5, 0.1875
    off = np.empty(Q)
    n_bulk = Q - 1500 - 1400
    off[:1500] = c
    off[1500:2200] = c - D
    off[2200:2900] = c + D
    off[2900:2900 + n_bulk // 2] = c - rng.uniform(1e-3, 0.4, n_bulk // 2)
    off[2900 + n_bulk // 2:] = c + rng.uniform(1e-3, 0.4, n_bulk - n_bulk // 2)
    rng.shuffle(off)
    zoff = np.full(len(P), 0.25); zoff[sel] = off
    Xm = P + np.column_stack((np.full(len(P), 0.003), np.full(len(P), 0.002), zoff))
    nv = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (Q, 1))
    pl = np.ones(Q, dtype=np.float32)
    z6, fixed = np.zeros(6), np.full(6, np.inf)
    o = orc.icp_iteration(Xm, P[sel], nv, pl, z6, z6, 1.0, z6, fixed, 0.3)
    assert o["median"] == c and o["mad"] == D and np.array_equal(o["dist"], off)        # the construction is what it claims
    with _lib.Context(0) as ctx:
        ctx.upload(_lib.FIX, P); ctx.upload(_lib.MOV, Xm)
        ctx.icp_setup(sel, nv, pl)
        whole = ctx.icp_run(z6, z6, fixed, 0.3, 1.0, max_iterations=8, min_change=0.0)
        assert len(whole) == 8
        for R in whole:
            assert (R.median, R.mad, R.n_kept) == (o["median"], o["mad"], o["n"])
        idx, dist, keep, _ = ctx.icp_state()
        assert np.array_equal(idx, o["nn"]) and np.array_equal(dist, o["dist"]) and np.array_equal(keep, o["keep"])
        # ... and host-driven launches on the same state buffer right behind them
        for it in range(3):
            R = ctx.icp_iterate(z6, z6, fixed, 0.3, 1.0)
            assert (R.median, R.mad, R.n_kept) == (o["median"], o["mad"], o["n"])


@pytest.mark.parametrize("seed", range(12))
def test_knn_sweep_random_clouds(seed):
    """Seeded random clouds of every shape the sweep kernels branch on -- volumes, planes, lines, clusters with exact duplicates,
    a few points, coordinates far from the origin -- with random k (2..32) and query subsets, four queries per wave: indices,
    squared distances and normals against the oracle."""
    import os
    from simpleicp_amd import _lib
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(40, 6000))
    kind = seed % 6
    if kind == 0:
        P = rng.uniform(-3, 3, (n, 3))
    elif kind == 1:
        P = np.column_stack((rng.uniform(0, 20, n), rng.uniform(0, 20, n), np.zeros(n)))           # an exact plane
    elif kind == 2:
        P = np.zeros((n, 3)); P[:, 1] = np.round(rng.uniform(0, 100, n), 2)                         # a line with ties
    elif kind == 3:
        c = rng.uniform(-10, 10, (5, 3))
        P = np.round(c[rng.integers(0, 5, n)] + rng.normal(0, 0.02, (n, 3)), 2)                     # clusters, many duplicates
    elif kind == 4:
        P = rng.uniform(0, 1, (n, 3)) * np.array([100.0, 0.01, 1.0])                                # a needle-shaped box
    else:
        P = rng.normal(0, 1, (n, 3)) + np.array([6.5e5, 5.1e6, 400.0])                              # far from the origin
    k = int(rng.integers(2, min(32, n) + 1))
    sel = np.sort(rng.choice(n, int(rng.integers(1, min(n, 400) + 1)), replace=False))
    env = {"SICP_KNN1": "grid", "SICP_KNN_GROUP": "4", "SICP_ORDER_MIN_Q": str(int(rng.integers(0, 2))), "SICP_KNN_BATCH": str(int(rng.integers(0, 17)))}
    os.environ.update(env)
    try:
        c = _lib.Context(0)
    finally:
        for key in env:
            del os.environ[key]
    with c:
        c.upload(_lib.FIX, P)
        idx, d2 = c.knn(_lib.FIX, P[sel], k=k)
        nv, pl, nn = c.estimate_normals(_lib.FIX, sel, k, want_nn=True)
    ridx, rd2 = orc.knn(P, P[sel], k=k)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2) and np.array_equal(nn, ridx)
    rnv, rpl = orc.normals(P, ridx)
    # degenerate neighbourhoods (coincident or collinear points) have no defined normal: where the oracle's is finite, ours equals it
    ok = np.isfinite(rnv).all(axis=1) & np.isfinite(rpl) & np.isfinite(nv).all(axis=1)
    if kind in (0, 4, 5):
        assert ok.all()
        assert np.abs(nv - rnv).max() <= 2e-7 and np.abs(pl - rpl).max() <= 2e-6 * max(1.0, np.abs(rpl).max())


@pytest.mark.parametrize("Q", [40, 300, 1000, 1500, 2048, 2049, 6000, 16_384])
@pytest.mark.parametrize("kind", ["plain", "quantised", "layers"])
def test_tail_window_selection_equals_the_histogram_selection(Q, kind):
    """The single-workgroup tail of a chained run looks for median and MAD in a window around the last iteration's values first
    (sicp_tail.hip: window_collect / window_pick) and falls back to the range-histogram selection when the wanted rank is not inside.
    Same keys, same order statistic: a run with the windows on must reproduce the run with them off (SICP_TAIL_WINDOW=0) bit for
    bit -- every iteration's median, MAD, counts, estimate and residual statistics -- and the windows must actually have been used
    once the estimate has settled.  `layers`: thousands of exactly equal distances (a window full of one key value: more members
    than it holds -> the fallback, every iteration).  Above 2048 correspondences the one-workgroup rejection (k_reject, keys in LDS)
    does the same from the statistics its last launch left."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(Q)
    n = 60_000
    if kind == "layers":
        P = np.column_stack((rng.uniform(0, 30, n), rng.uniform(0, 30, n), np.zeros(n)))
        Xm = P + np.array([0.0, 0.0, 0.25]) * rng.integers(0, 3, n)[:, None]
        Xm[:, :2] += 0.001
    else:
        P = _surface(n, 50 + (Q % 5))
        x_true = np.array([0.004, -0.003, 0.006, 0.08, -0.05, 0.03])
        Xm = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), P + rng.normal(0, 0.01, P.shape))
        if kind == "quantised":
            P, Xm = np.round(P, 2), np.round(Xm, 2)
    sel = np.sort(rng.choice(n, Q, replace=False))
    z = np.zeros(6)
    ow = np.full(6, np.inf) if kind == "layers" else z            # (layers: match + rejection only)
    runs = {}
    for window in ("1", "0"):
        os.environ["SICP_TAIL_WINDOW"] = window
        try:
            c = _lib.Context(0)
        finally:
            del os.environ["SICP_TAIL_WINDOW"]
        c.upload(_lib.FIX, P); c.upload(_lib.MOV, Xm)
        if kind == "layers":
            nv = np.tile(np.array([0, 0, 1], np.float32), (Q, 1)); pl = np.ones(Q, np.float32)
        else:
            nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
        c.icp_setup(sel, nv, pl)
        R = c.icp_run(z, z, ow, 0.3, None if kind != "layers" else 1.0, max_iterations=16, min_change=0.0)
        runs[window] = ([(r.median, r.mad, r.n_planar, r.n_kept, tuple(r.x[:]), r.res_mean, r.res_std, r.dist_mean, r.dist_std, r.lm_steps)
                         for r in R], c.icp_state(), c.tail_selection())
        c.close()
    on, off = runs["1"], runs["0"]
    assert len(on[0]) == len(off[0]) == 16 and on[0] == off[0]
    for a, b in zip(on[1], off[1]):
        assert np.array_equal(a, b)
    assert off[2]["window_iterations"] == 0
    if Q > 2048:
        return                                                      # (k_reject keeps no tally: equality is the test)
    if kind == "layers":
        assert on[2]["window_iterations"] == 0                     # a window cannot hold thousands of equal keys: always the fallback
    else:
        assert on[2]["window_iterations"] >= 6, on[2]
        assert on[2]["median_rounds"] == 0 and on[2]["mad_rounds"] == 0
