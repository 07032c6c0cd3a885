"""The oracle against the third-party routines the reference itself calls, on seeded random inputs (beyond the fixtures
of tests/test_oracle_golden.py): scipy.spatial.cKDTree for both k-NN uses, np.median / scipy.stats.median_abs_deviation
for the rejection, np.cov / np.linalg.eig for the normals, numpy's matrix product for the transform, the reference's own
residual expression, scipy.optimize.least_squares for the minimiser and the dense covariance formula for the
uncertainties (pointcloud.py:161-203, corrpts.py:124-211, optimization.py:65-288).  CPU only."""
import numpy as np
import pytest
from scipy import optimize, spatial, stats

from oracle import orc


def _H(rng):
    x = np.concatenate((rng.uniform(-0.5, 0.5, 3), rng.uniform(-3, 3, 3)))
    return x, orc.params_to_H(x)


@pytest.mark.parametrize("seed", range(6))
def test_knn_equals_ckdtree_without_ties(seed):
    """corrpts.py:131-132 / pointcloud.py:161-165,185-186: random doubles have no distance ties, so cKDTree's picks are
    determined and the brute-force (d2, idx) order must reproduce them, with and without the strict upper bound."""
    rng = np.random.default_rng(seed)
    n, q = int(rng.integers(50, 4000)), int(rng.integers(1, 300))
    P = rng.normal(0, 1, (n, 3)) * rng.uniform(0.1, 100)
    Q = rng.normal(0, 1, (q, 3)) * rng.uniform(0.1, 100)
    tree = spatial.cKDTree(P)
    for k in (1, int(rng.integers(2, min(n, 24)))):
        d, i = tree.query(Q, k=k, p=2)
        idx, d2 = orc.knn(P, Q, k=k)
        assert np.array_equal(idx, i.reshape(q, k))
        assert np.allclose(np.sqrt(d2), d.reshape(q, k), rtol=1e-14, atol=0)
    r = float(np.median(tree.query(Q, k=1)[0]))
    d, i = tree.query(Q, k=1, p=2, distance_upper_bound=r)
    idx, d2 = orc.knn(P, Q, k=1, max_dist=r)
    assert np.array_equal(idx[:, 0] >= 0, np.isfinite(d))               # strict bound: the same points are "in range"
    assert np.array_equal(idx[:, 0][np.isfinite(d)], i[np.isfinite(d)])
    # the transform folded into the search == transforming first (contract (T) is numpy's H @ Xh.T)
    _, H = _H(rng)
    Pt = (H @ np.column_stack((P, np.ones(n))).T).T[:, :3]
    assert np.array_equal(orc.transform(H, P), Pt)
    assert np.array_equal(orc.knn(P, Q, k=1, H=H)[0], orc.knn(Pt, Q, k=1)[0])


@pytest.mark.parametrize("seed", range(8))
def test_rejection_equals_numpy_and_scipy(seed):
    """corrpts.py:139-188 written with the reference's own calls: float32 planarity compare (NaN fails), np.median,
    stats.median_abs_deviation (scale 1.0), |d - median| <= 3 MAD -- on even / odd counts, duplicates, heavy tails."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(7, 3000))
    d = rng.normal(0, 0.05, n) + (rng.random(n) < 0.1) * rng.normal(0, 2, n)
    if seed % 2:
        d = np.round(d, 2)                                                 # many exactly equal distances
    pl = rng.uniform(0, 1, n).astype(np.float32)
    pl[rng.random(n) < 0.1] = np.nan
    thr = float(rng.uniform(0.1, 0.6))
    keep, cnt, med, mad = orc.reject(d, pl, thr)
    alive = pl >= thr                                                      # (numpy 2: the python float is cast to float32)
    dd = d[alive]
    m_ref, s_ref = np.median(dd), stats.median_abs_deviation(dd)
    want = np.zeros(n, bool)
    want[np.flatnonzero(alive)[[abs(v - m_ref) <= 3 * s_ref for v in dd]]] = True
    assert med == m_ref and mad == s_ref
    assert np.array_equal(keep, want) and cnt == int(want.sum())


@pytest.mark.parametrize("seed", range(4))
def test_normals_equal_cov_and_eig(seed):
    """pointcloud.py:188-198: np.cov of the k neighbours, eigen-decomposition, normal = eigenvector of the smallest
    eigenvalue (sign: the oracle's rule), planarity = (l_mid - l_min) / l_max; stored as float32."""
    rng = np.random.default_rng(200 + seed)
    n, k = 3000, int(rng.integers(4, 30))
    xy = rng.uniform(-10, 10, (n, 2))
    P = np.column_stack((xy, np.sin(xy[:, 0]) + 0.3 * xy[:, 1] + rng.normal(0, 0.02, n)))
    sel = rng.choice(n, 200, replace=False)
    nn, _ = orc.knn(P, P[sel], k=k)
    nv, pl = orc.normals(P, nn)
    for j in range(len(sel)):
        C = np.cov(P[nn[j]].T)
        w, V = np.linalg.eigh(C)
        v = V[:, 0]
        assert abs(abs(float(np.dot(v, nv[j].astype(np.float64)))) - 1) < 1e-6
        assert abs(float(pl[j]) - (w[1] - w[0]) / w[2]) < 2e-6
    big = np.argmax(np.abs(nv), axis=1)
    assert np.all(nv[np.arange(len(nv)), big] > 0)


def _problem(rng, n=400):
    xy = rng.uniform(-10, 10, (n, 2))
    p1 = np.column_stack((xy, 2 * np.sin(xy[:, 0] / 3) * np.cos(xy[:, 1] / 4)))
    nrm = np.column_stack((-2 / 3 * np.cos(xy[:, 0] / 3) * np.cos(xy[:, 1] / 4), 2 / 4 * np.sin(xy[:, 0] / 3) * np.sin(xy[:, 1] / 4),
                           np.ones(n)))
    n1 = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    x_true = np.concatenate((rng.uniform(-0.02, 0.02, 3), rng.uniform(-0.2, 0.2, 3)))
    p2 = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), p1 + rng.normal(0, 0.01, p1.shape))
    return p1, n1, p2, x_true


def _ref_residuals(x, p1, n1, p2):
    """optimization.py:213-262 in numpy, literally: H @ Xh.T, then dx * nx + dy * ny + dz * nz."""
    H = orc.params_to_H(x)
    Xt = (H @ np.column_stack((p2, np.ones(len(p2)))).T).T
    Xe = Xt[:, :3] / Xt[:, 3:4]
    d = Xe - p1
    return d[:, 0] * n1[:, 0] + d[:, 1] * n1[:, 1] + d[:, 2] * n1[:, 2]


@pytest.mark.parametrize("seed", range(4))
def test_residuals_solver_and_uncertainties(seed):
    rng = np.random.default_rng(300 + seed)
    p1, n1, p2, x_true = _problem(rng)
    keep = rng.random(len(p1)) < 0.8
    x0 = np.zeros(6)
    # residual expression: bit-equal to the reference's numpy arithmetic
    assert np.array_equal(orc.residuals(x_true, p1, n1, p2, keep), _ref_residuals(x_true, p1, n1, p2)[keep])
    assert np.array_equal(orc.point_to_plane(p1, n1, p2, orc.params_to_H(x_true)), _ref_residuals(x_true, p1, n1, p2))
    # minimiser: scipy's least_squares (what lmfit's "least_squares" method calls) on the same weighted objective,
    # with one observed and one fixed parameter
    w = float(rng.uniform(0.5, 20))
    obs = np.array([0.0, 0.0, 0.0, 0.0, 0.05, 0.0])
    ow = np.array([0.0, 0.0, np.inf, 0.0, 30.0, 0.0])
    free = np.isfinite(ow)

    def fun(xf):
        x = x0.copy()
        x[free] = xf
        r = w * _ref_residuals(x, p1[keep], n1[keep], p2[keep])
        return np.concatenate((r, [ow[4] * (x[4] - obs[4])]))
    sol = optimize.least_squares(fun, x0[free], method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-14)
    x, steps = orc.solve(x0, w, obs, ow, p1, n1, p2, keep)
    assert steps < 40 and x[2] == 0.0
    assert np.abs(x[free] - sol.x).max() < 1e-8
    assert np.sum(fun(x[free]) ** 2) <= np.sum(sol.fun ** 2) * (1 + 1e-10)
    # uncertainties: optimization.py:139-170 with a dense weight matrix and a central-difference Jacobian
    J = np.empty((len(sol.fun), int(free.sum())))
    for j in range(J.shape[1]):
        e = np.zeros(J.shape[1])
        e[j] = 1e-6
        J[:, j] = (fun(x[free] + e) - fun(x[free] - e)) / 2e-6
    weights = np.concatenate((np.full(int(keep.sum()), w), [ow[4]]))
    A = J / weights[:, None]
    v = fun(x[free]) / weights
    N = A.T @ np.diag(weights) @ A
    s0 = np.sqrt(np.sum(weights * v ** 2) / (A.shape[0] - A.shape[1]))
    sigma_ref = s0 * np.sqrt(np.diag(np.linalg.inv(N)))
    sigma = orc.uncertainties(x, w, obs, ow, p1, n1, p2, keep)
    assert np.allclose(sigma[free], sigma_ref, rtol=1e-5) and np.isnan(sigma[2])
    # the 30 sums of the fused reduction against their definition
    ne = orc.normal_equations(x, p1, n1, p2, keep)
    r = _ref_residuals(x, p1[keep], n1[keep], p2[keep])
    assert abs(ne[27] - r.sum()) < 1e-10 and abs(ne[28] - np.sum(r * r)) < 1e-10 and ne[29] == keep.sum()
    Jd = np.empty((int(keep.sum()), 6))
    for j in range(6):
        e = np.zeros(6)
        e[j] = 1e-6
        Jd[:, j] = (_ref_residuals(x + e, p1[keep], n1[keep], p2[keep]) - _ref_residuals(x - e, p1[keep], n1[keep], p2[keep])) / 2e-6
    JTJ = Jd.T @ Jd
    iu = np.triu_indices(6)
    assert np.allclose(ne[:21], JTJ[iu], rtol=1e-6, atol=1e-6 * np.abs(JTJ).max())
    assert np.allclose(ne[21:27], Jd.T @ r, rtol=1e-5, atol=1e-6 * np.abs(Jd.T @ r).max() + 1e-9)
