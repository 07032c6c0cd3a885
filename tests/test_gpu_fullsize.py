"""BASELINE.json's full size (10M-vs-10M, Q = 1000):
  * the HIP match == the oracle's brute-force CPU check (orc.knn, 1e10 pairs per pass: seconds on the GPU box's
    host cores), indices AND squared distances bit for bit, for the identity and for a rigid H;
  * three full iterations (match -> distances -> rejection -> solve) against orc.icp_iteration: indices,
    distances and keep mask bit-exact, median / MAD equal, parameters to 1e-9 -- the config the metric is quoted on;
  * three independent exact kernels agree bit for bit (grid search == filtered brute-force scan;
    the exact FP64 scan on a 1M-point slice == both);
  * sharding is invisible: the lexicographic merge of two half-cloud searches == the full search;
  * rigid round trip: a cloud moved by a known H is registered back to that H to 1e-9, with zero residuals.
"""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
N, Q = 10_000_000, 1000


@pytest.fixture(scope="module")
def data():
    import bench
    Xf, Xm, H_true = bench.synthetic_pair(N)
    sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
    return Xf, Xm, H_true, sel


def _ctx(mode):
    from simpleicp_amd import _lib
    if mode:
        os.environ["SICP_KNN1"] = mode
    try:
        return _lib.Context(0)
    finally:
        os.environ.pop("SICP_KNN1", None)


def test_three_exact_kernels_agree(data):
    from simpleicp_amd import _lib
    from simpleicp_amd.rbp import H_from_params
    Xf, Xm, H_true, sel = data
    H = H_from_params(np.array([0.004, -0.002, 0.006, 0.1, -0.1, 0.05]))
    q = Xf[sel]
    out = {}
    for mode in ("grid", "filter"):
        with _ctx(mode) as c:
            c.upload(_lib.MOV, Xm)
            out[mode] = c.knn(_lib.MOV, q, k=1, H=H)
            assert c.last_match_kernel() == {"grid": "k_grid_nn", "filter": "k_knn1_frec"}[mode]
    assert np.array_equal(out["grid"][0], out["filter"][0]) and np.array_equal(out["grid"][1], out["filter"][1])
    # exact FP64 scan on a slice that contains every winner's neighbourhood is too big; use a 1M-point prefix
    with _ctx("exact") as c:
        c.upload(_lib.MOV, Xm[:1_000_000])
        ex = c.knn(_lib.MOV, q, k=1, H=H)
    with _ctx("grid") as c:
        c.upload(_lib.MOV, Xm[:1_000_000])
        gr = c.knn(_lib.MOV, q, k=1, H=H)
    assert np.array_equal(ex[0], gr[0]) and np.array_equal(ex[1], gr[1])


def test_sharding_is_invisible(data):
    """index_base + lexicographic merge (what the multi-GPU exchange does) at full size."""
    from simpleicp_amd import _lib
    Xf, Xm, H_true, sel = data
    q = Xf[sel]
    with _ctx(None) as c:
        c.upload(_lib.MOV, Xm)
        full_idx, full_d2 = c.knn(_lib.MOV, q, k=1)
        parts = []
        for lo, hi in ((0, 3_333_333), (3_333_333, 7_000_001), (7_000_001, N)):
            c.upload(_lib.MOV, Xm[lo:hi], index_base=lo)
            i, d = c.knn(_lib.MOV, q, k=1)
            parts.append(np.column_stack((d[:, 0], i[:, 0].view(np.float64), np.zeros((len(q), 3)))))
        d2, idx, _ = c.lexmin_gathered(np.stack(parts))
    assert np.array_equal(idx, full_idx[:, 0]) and np.array_equal(d2, full_d2[:, 0])


def test_rigid_round_trip(data):
    """movable := H_true^-1 (fixed): ICP must return H_true and vanishing residuals."""
    from simpleicp_amd import _lib
    from oracle import orc
    Xf, Xm, H_true, sel = data
    Xm2 = orc.transform(np.linalg.inv(H_true), Xf)                 # same points, moved rigidly (fp64 rounding only)
    with _ctx(None) as c:
        c.upload(_lib.FIX, Xf)
        c.upload(_lib.MOV, Xm2)
        nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
        c.icp_setup(sel, nv, pl)
        x = np.zeros(6)
        for _ in range(30):
            R = c.icp_iterate(x, np.zeros(6), np.zeros(6), 0.3, 1.0)
            x = np.array(R.x[:])
        idx, dist, keep, res = c.icp_state()
    assert np.abs(_lib.params_to_H(x) - H_true).max() < 1e-9
    assert np.array_equal(idx[keep], sel[keep])                     # every kept query found its own twin
    assert np.abs(res[keep]).max() < 1e-9


def test_match_equals_oracle_brute_force_at_full_size(data):
    """north_star: "correspondence indices are bit-exact against a brute-force CPU check" -- at the size the
    metric is quoted on (corrpts.py:131-135: 1-NN of the Q selected fixed points in the whole movable cloud)."""
    from simpleicp_amd import _lib
    from simpleicp_amd.rbp import H_from_params
    from oracle import orc
    Xf, Xm, H_true, sel = data
    q = Xf[sel]
    with _ctx(None) as c:
        c.upload(_lib.MOV, Xm)
        for H in (None, H_from_params(np.array([0.004, -0.002, 0.006, 0.1, -0.1, 0.05])), H_true):
            idx, d2 = c.knn(_lib.MOV, q, k=1, H=H)
            assert c.last_match_kernel() == "k_grid_nn"                       # the default (product) path
            ridx, rd2 = orc.knn(Xm, q, k=1, H=H)
            assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)


def test_iterations_equal_oracle_at_full_size(data):
    """Three whole iterations of the C4 workload against the oracle (corrpts.py:124-188, optimization.py:65-124)."""
    from simpleicp_amd import _lib
    from oracle import orc
    Xf, Xm, H_true, sel = data
    z = np.zeros(6)
    with _ctx(None) as c:
        c.upload(_lib.FIX, Xf)
        c.upload(_lib.MOV, Xm)
        nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
        c.icp_setup(sel, nv, pl)
        x = z.copy()
        for it in range(3):
            R = c.icp_iterate(x, z, z, 0.3, 1.0)
            o = orc.icp_iteration(Xm, Xf[sel], nv, pl, x, x, 1.0, z, z, 0.3)
            idx, dist, keep, resid = c.icp_state()
            assert np.array_equal(idx, o["nn"]) and np.array_equal(dist, o["dist"]) and np.array_equal(keep, o["keep"])
            assert R.n_kept == o["n"] and R.median == o["median"] and R.mad == o["mad"]
            x = np.array(R.x[:])
            assert np.abs(x - o["x"]).max() < 1e-9
        whole = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=3, min_change=0.0)
        assert np.abs(np.array(whole[-1].x[:]) - x).max() < 1e-13


# ---- round 3: the operators that had no oracle check at the headline size ------------------------------------------

def test_normals_knn_equals_oracle_at_full_size(data):
    """estimate_normals at 10 M (pointcloud.py:185-198): the k = 10 search itself (k_grid_knn) against the oracle's
    brute-force k-NN -- indices bit for bit incl. their (d2, idx) order -- then normals / planarity against the
    oracle's covariance + eigen step on those neighbours (float32 store, one ulp of a unit vector's component)."""
    from simpleicp_amd import _lib
    from oracle import orc
    Xf, Xm, H_true, sel = data
    with _ctx(None) as c:
        c.upload(_lib.FIX, Xf)
        nv, pl, nn = c.estimate_normals(_lib.FIX, sel, 10, want_nn=True)
    onn, _ = orc.knn(Xf, Xf[sel], k=10)
    assert np.array_equal(nn, onn)
    assert np.array_equal(nn[:, 0], sel)                         # the query itself is its own nearest neighbour
    onv, opl = orc.normals(Xf, onn)
    # same operation order on both sides; 1 ulp(float32) of a unit vector's components for the fp64 sqrt / div paths
    assert np.abs(nv - onv).max() <= 2e-7 and np.abs(pl - opl).max() <= 2e-6
    # and another neighbourhood size through the same kernel (webots' k = 40, pointcloud.py:173)
    sub = sel[::10]
    with _ctx(None) as c:
        c.upload(_lib.FIX, Xf)
        _, _, nn40 = c.estimate_normals(_lib.FIX, sub, 40, want_nn=True)
    assert np.array_equal(nn40, orc.knn(Xf, Xf[sub], k=40)[0])


@pytest.mark.slow
def test_select_in_range_at_full_size(data):
    """select_in_range with ALL 10 M fixed points as queries against the 10 M movable cloud under a rigid H
    (pointcloud.py:161-167, strict `<`): 2000 sampled verdicts against orc.knn(max_dist), plus a bound chosen ON a
    sampled nearest-neighbour distance (strictness at full size) and the two trivial bounds."""
    from simpleicp_amd import _lib
    from oracle import orc
    Xf, Xm, H_true, sel = data
    pick = np.unique(np.round(np.linspace(0, N - 1, 2000)).astype(np.int64))
    oidx, od2 = orc.knn(Xm, Xf[pick], k=1, H=H_true)
    dist = np.sqrt(od2[:, 0])
    with _ctx(None) as c:
        c.upload(_lib.FIX, Xf)
        c.upload(_lib.MOV, Xm)
        for bound in (float(np.median(dist)), float(np.sort(dist)[len(dist) // 10]), float(dist[7]), float(np.nextafter(dist[11], 0))):
            near = c.select_in_range(_lib.FIX, _lib.MOV, None, H_true, bound)
            assert near.shape == (N,)
            ridx, _ = orc.knn(Xm, Xf[pick], k=1, H=H_true, max_dist=bound)
            assert np.array_equal(near[pick], ridx[:, 0] >= 0)
            assert 0 < near.sum() < N
        # a bound whose SQUARE is exactly a nearest-neighbour's squared distance: that point is NOT in range (the rule
        # is d2 < max_range * max_range, strict, like cKDTree's distance_upper_bound); one ulp more and it is
        j = int(np.flatnonzero((dist * dist == od2[:, 0]) & (dist > 0))[0])
        assert not c.select_in_range(_lib.FIX, _lib.MOV, pick[j:j + 1], H_true, float(dist[j]))[0]
        assert c.select_in_range(_lib.FIX, _lib.MOV, pick[j:j + 1], H_true, float(np.nextafter(dist[j], np.inf)))[0]
        # a sub-selection gives the same verdicts as the full pass
        sub = c.select_in_range(_lib.FIX, _lib.MOV, pick, H_true, float(np.median(dist)))
        full = c.select_in_range(_lib.FIX, _lib.MOV, None, H_true, float(np.median(dist)))
        assert np.array_equal(sub, full[pick])
        assert c.select_in_range(_lib.FIX, _lib.MOV, None, H_true, np.inf).all()


def test_run_on_dataframes_at_full_size(data):
    """SimpleICP.run() itself on C4 DataFrames (simpleicp.py:135-324) against the oracle's whole-loop driver on the same
    arrays: selection, iteration count, per-iteration correspondence counts, H to 1e-9; the result against H_true;
    side effects (simpleicp.py:254,316: pc_fix keeps the sub-sample and gains sparse float32 columns, pc_mov is
    transformed in place)."""
    from simpleicp_amd import PointCloud, SimpleICP
    from oracle import orc
    Xf, Xm, H_true, sel = data
    pc_fix = PointCloud(Xf, columns=["x", "y", "z"])
    pc_mov = PointCloud(Xm.copy(), columns=["x", "y", "z"])
    icp = SimpleICP(verbose=False)
    icp.add_point_clouds(pc_fix, pc_mov)
    H, X, rbp, res = icp.run(correspondences=Q, neighbors=10)
    o = orc.run(Xf, Xm, correspondences=Q, neighbors=10)
    assert np.array_equal(pc_fix.idx_selected, o["sel"]) and np.array_equal(o["sel"], sel)
    assert icp.last_run_info["iterations"] == o["iterations"]
    assert [s[0] for s in icp.last_run_info["stats"]] == [s[0] for s in o["stats"]]
    assert np.abs(H - o["H"]).max() < 1e-9
    assert np.abs(np.array(rbp.get_parameter_attributes_as_list("estimated_value")) - o["x"]).max() < 1e-9
    assert np.allclose(np.array(rbp.get_parameter_attributes_as_list("estimated_uncertainty")), o["sigma"], rtol=1e-6)
    # (a rotation that differs by 1e-9 rad moves a point R metres from the origin by 1e-9 R: the residuals' bound carries the
    # cloud's radius, the parameters' does not)
    assert len(res) == len(o["residuals"]) and np.abs(res - o["residuals"]).max() < 1e-9 * (1 + np.abs(Xm).max())
    # the two samplings of the surface are independent: H_true is met to the data's noise, not to rounding
    assert np.abs(H - H_true).max() < 2e-2 and np.abs(H[:3, :3] - H_true[:3, :3]).max() < 1e-4
    # side effects
    assert X.shape == (N, 3) and np.array_equal(X, pc_mov.X)
    assert np.array_equal(X[::9973], orc.transform(H, Xm[::9973]))
    for col in ("nx", "ny", "nz", "planarity"):
        assert str(pc_fix[col].dtype) == "Sparse[float32, nan]"
    got = np.column_stack([pc_fix[c].to_numpy()[sel] for c in ("nx", "ny", "nz")])
    assert np.abs(got - o["normals"]).max() <= 2e-7


@pytest.fixture(scope="module")
def big_q(data):
    """The throughput regime on the same clouds: 1 M selected points -> k_grid_nn16, k_hsel_*, k_keep_stats,
    k_lm_eval / k_lm_finish (none of which run at Q = 1000)."""
    from simpleicp_amd import _lib
    Xf, Xm, H_true, _ = data
    sel = np.unique(np.round(np.linspace(0, N - 1, 1_000_000)).astype(np.int64))
    c = _ctx(None)
    c.upload(_lib.FIX, Xf)
    c.upload(_lib.MOV, Xm)
    nv, pl = c.estimate_normals(_lib.FIX, sel, 10)
    yield c, sel, nv, pl
    c.close()


def check_large_q_iteration(c, Xf, Xm, sel, nv, pl, x, R, n_sample, min_planarity=0.3):
    """One large-Q iteration against the oracle: the match on a sample of the queries (brute force over the WHOLE
    movable cloud), then everything downstream on ALL correspondences -- distances, keep mask, median, MAD, the
    minimiser -- from the device's matched indices (each stage is per-correspondence or a reduction the oracle
    recomputes in full; only the 1e13-pair match itself has to be sampled)."""
    from oracle import orc
    idx, dist, keep, resid = c.icp_state()
    H = orc.params_to_H(x)
    pick = np.unique(np.round(np.linspace(0, len(sel) - 1, n_sample)).astype(np.int64))
    nn, _ = orc.knn(Xm, Xf[sel[pick]], k=1, H=H)
    assert np.array_equal(idx[pick], nn[:, 0])
    p1, p2 = Xf[sel], Xm[idx]
    od = orc.point_to_plane(p1, nv, p2, H)
    assert np.array_equal(dist, od)
    okeep, on, omed, omad = orc.reject(od, pl, min_planarity)
    assert np.array_equal(keep, okeep) and R.n_kept == on and R.median == omed and R.mad == omad
    ox, _ = orc.solve(x, 1.0, np.zeros(6), np.zeros(6), p1, nv, p2, okeep)
    assert np.abs(np.array(R.x[:]) - ox).max() < 1e-9
    ores = orc.residuals(ox, p1, nv, p2, okeep)
    assert np.abs(resid[keep] - ores).max() < 1e-9 * (1 + np.abs(Xm).max())      # 1e-9 rad at the cloud's radius
    return ox


@pytest.mark.slow
def test_large_q_iterations_equal_oracle_at_full_size(data, big_q):
    """10 M clouds, Q = 1 M: three iterations of the many-workgroup path (a6-a10 at C5-class Q)."""
    Xf, Xm, H_true, _ = data
    c, sel, nv, pl = big_q
    z = np.zeros(6)
    c.icp_setup(sel, nv, pl)
    x = z.copy()
    for it in range(3):
        R = c.icp_iterate(x, z, z, 0.3, 1.0)
        assert c.last_match_kernel() == "k_grid_nn16f"
        check_large_q_iteration(c, Xf, Xm, sel, nv, pl, x, R, 3000)
        x = np.array(R.x[:])
    # the chained loop lands on the same estimates
    c.icp_setup(sel, nv, pl)
    whole = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=3, min_change=0.0)
    # (the chained loop carries sin / cos forward on the device, the iterate loop takes them from libm: equal to rounding)
    assert np.abs(np.array(whole[-1].x[:]) - x).max() < 1e-12
    assert whole[-1].n_kept == R.n_kept and abs(whole[-1].median - R.median) < 1e-12 and abs(whole[-1].mad - R.mad) < 1e-12


@pytest.mark.slow
def test_normals_at_one_million_queries_equal_oracle(data, big_q):
    """estimate_normals at Q = 1 M on 10 M points (pointcloud.py:173-203; C5-class Q): the one-sweep k-NN + covariance kernel with
    a few cell-ordered queries per wave.  A 3 000-query sample against the oracle's brute-force k-NN over the whole cloud -- indices
    bit for bit incl. their (d2, idx) order -- and the normals / planarity the kernel formed WITHOUT writing those lists
    against the oracle's covariance + eigen step on them (float32 store, 1 ulp of a unit vector's component)."""
    from simpleicp_amd import _lib
    from oracle import orc
    Xf, _, _, _ = data
    c, sel, nv, pl = big_q
    pick = np.unique(np.round(np.linspace(0, len(sel) - 1, 3000)).astype(np.int64))
    onn, _ = orc.knn(Xf, Xf[sel[pick]], k=10)
    onv, opl = orc.normals(Xf, onn)
    assert np.abs(nv[pick] - onv).max() <= 2e-7 and np.abs(pl[pick] - opl).max() <= 2e-6
    assert np.isfinite(nv).all() and np.isfinite(pl).all()
    # the same call with the index lists asked for: same normals, and the lists are the oracle's
    c.timing_enable(True, count_work=True); c.timing_reset()
    nv2, pl2, nn = c.estimate_normals(_lib.FIX, sel, 10, want_nn=True)
    work = c.knn_work()
    c.timing_enable(False)
    assert np.array_equal(nv2, nv) and np.array_equal(pl2, pl)
    assert np.array_equal(nn[pick], onn) and np.array_equal(nn[:, 0], sel)
    assert work["sweeps"] >= len(sel) and work["slow_queries"] == 0, work
    assert work["candidates"] <= 200 * len(sel), work              # (round 3's k-round search read ~20 x that)


@pytest.mark.slow
def test_mid_q_iteration_equals_oracle_at_full_size(data, big_q):
    """... and Q = 100 000 (SURVEY 8d's throughput point) on the same resident clouds."""
    Xf, Xm, H_true, _ = data
    c, sel, nv, pl = big_q
    z = np.zeros(6)
    s = slice(None, None, 10)
    c.icp_setup(sel[s], nv[s], pl[s])
    x = z.copy()
    for it in range(2):
        R = c.icp_iterate(x, z, z, 0.3, 1.0)
        check_large_q_iteration(c, Xf, Xm, sel[s], nv[s], pl[s], x, R, 3000)
        x = np.array(R.x[:])
