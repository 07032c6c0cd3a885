"""CPU port of the reference's Python ICP loop (TEST INFRASTRUCTURE / CPU BASELINE ONLY).

A pandas-free restatement of ``/root/reference/python/simpleicp`` that keeps the
reference's own third-party calls -- ``scipy.spatial.cKDTree`` (rebuilt on the
transformed movable cloud every iteration, corrpts.py:131), ``scipy.stats``-style
raw MAD, and ``scipy.optimize.least_squares`` (what lmfit's "least_squares" method
forwards to, optimization.py:93-101) -- so that

  * tests can check it against fixtures produced by the unmodified reference
    (tests/golden/*.npz, made by oracle/make_golden.py), and
  * bench.py can time "the reference's algorithm on the host cores" on the GPU
    box, where /root/reference itself does not exist (cpu_baseline.kind = "port").

It is NOT imported by the product package ``simpleicp_amd``.

Reference lines restated:
  simpleicp.py:135-324,356-379  run loop / convergence
  pointcloud.py:132-217         select_n_points, select_in_range, estimate_normals, transform_by_H
  corrpts.py:124-211            match, rejections, point-to-plane distances
  optimization.py:65-288        NLLS over absolute (alpha, t), observation rows, uncertainties
  mathutils.py:39-93            Euler convention
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np
from scipy.optimize import least_squares
from scipy.spatial import cKDTree


def euler_to_R(a1, a2, a3):
    """mathutils.py:39-68."""
    c1, s1, c2, s2, c3, s3 = np.cos(a1), np.sin(a1), np.cos(a2), np.sin(a2), np.cos(a3), np.sin(a3)
    return np.array([
        [c2 * c3, -c2 * s3, s2],
        [c1 * s3 + s1 * s2 * c3, c1 * c3 - s1 * s2 * s3, -s1 * c2],
        [s1 * s3 - c1 * s2 * c3, s1 * c3 + c1 * s2 * s3, c1 * c2],
    ])


def params_to_H(x):
    """mathutils.py:81-93 applied to optimization.py:334-350."""
    H = np.eye(4)
    H[:3, :3] = euler_to_R(x[0], x[1], x[2])
    H[:3, 3] = x[3:6]
    return H


def transform(X, H):
    """pointcloud.py:205-217 (homogeneous matmul, divide by w)."""
    Xh = np.column_stack((X, np.ones(len(X))))
    Y = (H @ Xh.T).T
    return np.column_stack((Y[:, 0] / Y[:, 3], Y[:, 1] / Y[:, 3], Y[:, 2] / Y[:, 3]))


def select_n_points(sel_idx, n):
    """pointcloud.py:132-147 -> new selected indices (sorted, duplicates collapsed)."""
    if len(sel_idx) > n:
        pos = np.round(np.linspace(0, len(sel_idx) - 1, n)).astype(int)
        return np.unique(sel_idx[pos])
    return sel_idx


def select_in_range(X_fix, sel_idx, X_other, max_range):
    """pointcloud.py:149-171 (strict upper bound)."""
    d, _ = cKDTree(X_other).query(X_fix[sel_idx], k=1, p=2, distance_upper_bound=max_range, workers=-1)
    return sel_idx[np.isfinite(d)]


def estimate_normals(X, sel_idx, k):
    """pointcloud.py:173-203 -> (normals f32 (q,3), planarity f32 (q,), nn idx (q,k))."""
    _, nn = cKDTree(X).query(X[sel_idx], k=k, p=2, workers=-1)
    normals = np.empty((len(sel_idx), 3), np.float32)
    planarity = np.empty(len(sel_idx), np.float32)
    for i, idx in enumerate(nn):
        C = np.cov(X[idx, :].T, bias=False)
        w, V = np.linalg.eig(C)
        order = w.argsort()[::-1]
        w, V = w[order], V[:, order]
        normals[i] = V[:, 2]
        planarity[i] = (w[1] - w[2]) / w[0]
    return normals, planarity, nn


@dataclass
class PortResult:
    H: np.ndarray
    x: np.ndarray                      # alpha1..3 [rad], tx, ty, tz
    sigma: np.ndarray                  # uncertainties (NaN for fixed parameters)
    residuals: np.ndarray
    iterations: int
    counts: list = field(default_factory=list)         # correspondences per iteration
    n_initial: int = 0
    per_iter: list = field(default_factory=list)       # dicts with timing
    seconds: float = 0.0


def run(X_fix, X_mov, correspondences=1000, neighbors=10, min_planarity=0.3,
        max_overlap_distance=np.inf, min_change=1.0, max_iterations=100,
        distance_weights=1, rbp_observed_values=(0.,) * 6, rbp_observation_weights=(0.,) * 6,
        normals=None, planarity=None, sel_idx=None, record=None):
    """simpleicp.py:75-324.  ``normals/planarity/sel_idx`` inject precomputed
    attributes of the selected fixed points (the reference's bypass, simpleicp.py:176)."""
    t_start = time.perf_counter()
    X_fix = np.ascontiguousarray(X_fix, dtype=np.float64)
    X_mov = np.ascontiguousarray(X_mov, dtype=np.float64)
    obs = np.array(rbp_observed_values, dtype=float)
    obs[:3] = obs[:3] * np.pi / 180
    ow = np.array(rbp_observation_weights, dtype=float)
    H = params_to_H(obs)

    if sel_idx is None:
        sel = np.arange(len(X_fix))
        if np.isfinite(max_overlap_distance):
            sel = select_in_range(X_fix, sel, transform(X_mov, H), max_overlap_distance)
            if len(sel) == 0:
                raise RuntimeError("no overlap")
        sel = select_n_points(sel, correspondences)
    else:
        sel = np.asarray(sel_idx)
    if normals is None:
        normals, planarity, _ = estimate_normals(X_fix, sel, neighbors)
    p1 = X_fix[sel]
    n1 = normals.astype(np.float64)
    free = np.isfinite(ow)
    observed = (ow > 0) & np.isfinite(ow)

    res_hist, counts, per_iter = [], [], []
    x_est = obs.copy()
    n_initial = 0
    w = distance_weights
    for it in range(max_iterations):
        t0 = time.perf_counter()
        Xt = transform(X_mov, H)                              # simpleicp.py:188
        t_xf = time.perf_counter() - t0
        tree = cKDTree(Xt)                                    # corrpts.py:131 (rebuilt every iteration)
        t_build = time.perf_counter() - t0 - t_xf
        _, nn = tree.query(p1, k=1, p=2, workers=-1)          # corrpts.py:132
        t_query = time.perf_counter() - t0 - t_xf - t_build
        p2t = Xt[nn]
        dist = (p2t[:, 0] - p1[:, 0]) * n1[:, 0] + (p2t[:, 1] - p1[:, 1]) * n1[:, 1] + (p2t[:, 2] - p1[:, 2]) * n1[:, 2]
        _ = transform(Xt, np.linalg.inv(H))                   # simpleicp.py:202 (cost only)
        t_match = time.perf_counter() - t0
        t_rej0 = time.perf_counter()
        keep = planarity >= np.float32(min_planarity)         # corrpts.py:139-163
        dk = dist[keep]
        med = np.median(dk)
        mad = np.median(np.abs(dk - med))                     # corrpts.py:186, scale 1.0
        keep2 = np.abs(dk - med) <= 3 * mad
        kidx = np.flatnonzero(keep)[keep2]
        if len(kidx) < 6:
            raise RuntimeError("too few correspondences")
        if it == 0:
            n_initial = len(kidx)
            initial_dist = dist[kidx]
        if w is None:
            w = 1 / (np.std(dist[kidx]) ** 2)                 # simpleicp.py:233-234
        q1, qn, q2 = p1[kidx], n1[kidx], X_mov[nn[kidx]]

        x0 = obs.copy() if it == 0 else x_est.copy()
        t_reject = time.perf_counter() - t_rej0
        t_sol0 = time.perf_counter()

        def resid(xf, x0=x0, q1=q1, qn=qn, q2=q2, w=w):
            x = x0.copy()
            x[free] = xf
            p = transform(q2, params_to_H(x))
            r = w * ((p[:, 0] - q1[:, 0]) * qn[:, 0] + (p[:, 1] - q1[:, 1]) * qn[:, 1] + (p[:, 2] - q1[:, 2]) * qn[:, 2])
            o = ow[observed] * (x[observed] - obs[observed])
            return np.concatenate((r, o))

        sol = least_squares(resid, x0[free], jac="2-point", method="trf", ftol=1e-8, xtol=1e-8, gtol=1e-8)
        x_est = x0.copy()
        x_est[free] = sol.x
        H = params_to_H(x_est)
        r_final = resid(sol.x)
        res = r_final[:len(kidx)] / w
        res_hist.append(res)
        counts.append(len(kidx))
        per_iter.append({"match_s": t_match, "transform_s": t_xf, "tree_build_s": t_build, "tree_query_s": t_query,
                         "reject_s": t_reject, "solve_s": time.perf_counter() - t_sol0, "total_s": time.perf_counter() - t0})
        if record is not None:
            record.append({"nn": nn.copy(), "dist": dist.copy(), "kept": kidx.copy(), "x": x_est.copy(),
                           "median": med, "mad": mad})
        if it > 0 and _converged(res_hist[it], res_hist[it - 1], min_change):
            break

    # optimization.py:126-170
    weights = np.concatenate((np.full(len(kidx), w), ow[observed]))
    A = sol.jac / weights[:, None]
    ru = r_final / weights
    N = A.T @ np.diag(weights) @ A
    Cxx = np.sum(weights * ru ** 2) / (A.shape[0] - A.shape[1]) * np.linalg.inv(N)
    sigma = np.full(6, np.nan)
    sigma[free] = np.sqrt(np.diag(Cxx))
    return PortResult(H=H, x=x_est, sigma=sigma, residuals=res_hist[-1], iterations=len(res_hist),
                      counts=counts, n_initial=n_initial, per_iter=per_iter,
                      seconds=time.perf_counter() - t_start)


def _converged(new, old, min_change):
    """simpleicp.py:356-379."""
    def change(a, b):
        if b == 0:
            return 0.0 if a == 0 else np.inf
        return abs((a - b) / b * 100)
    return change(np.mean(new), np.mean(old)) < min_change and change(np.std(new), np.std(old)) < min_change


def synthetic_pair(n, seed_fix=0, seed_mov=1, dtype=np.float64):
    """SURVEY.md section 8(d) generator (C4/C5): two independent samplings of one
    analytic surface, 10 pts/m^2, movable = H_true^-1 applied; returns
    (X_fix, X_mov, H_true).  Centroid of the fixed cloud is subtracted from both."""
    L = np.sqrt(n / 10.0)

    def sample(seed):
        rng = np.random.default_rng(seed)
        x = rng.uniform(0, L, n)
        y = rng.uniform(0, L, n)
        z = 20 * np.sin(2 * np.pi * x / 200) * np.cos(2 * np.pi * y / 300) \
            + 5 * np.sin(2 * np.pi * x / 37 + 1) * np.sin(2 * np.pi * y / 53) + rng.normal(0, 0.02, n)
        return np.column_stack((x, y, z))

    Xf = sample(seed_fix)
    Xm = sample(seed_mov)
    c = Xf.mean(axis=0)
    Xf -= c
    Xm -= c
    x_true = np.array([np.deg2rad(0.5), np.deg2rad(-0.3), np.deg2rad(0.8), 0.30, -0.20, 0.10])
    H_true = params_to_H(x_true)
    Hinv = np.linalg.inv(H_true)
    Xm = Xm @ Hinv[:3, :3].T + Hinv[:3, 3]
    return Xf.astype(dtype), Xm.astype(dtype), H_true
