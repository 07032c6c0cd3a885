"""ctypes front end of the C oracle (oracle/sicp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Builds ``oracle/_build/libsicp_oracle.so`` on first use (gcc is in the image).
Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libsicp_oracle.so"
_lib = None

_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_bp = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build():
    src = _HERE / "sicp_oracle.c"
    if not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(str(build()))
        L.orc_transform.argtypes = [_dp, _dp, C.c_int64, _dp]
        L.orc_knn.argtypes = [_dp, C.c_int64, C.c_void_p, _dp, C.c_int64, C.c_int, C.c_double, C.c_int64, _ip, _dp]
        L.orc_select_n_points.argtypes = [C.c_int64, C.c_int64, _ip]
        L.orc_select_n_points.restype = C.c_int64
        L.orc_normals.argtypes = [_dp, _ip, C.c_int64, C.c_int, _fp, _fp]
        L.orc_point_to_plane.argtypes = [_dp, _fp, _dp, _dp, C.c_int64, _dp]
        L.orc_reject.argtypes = [_dp, _fp, C.c_int64, C.c_double, _bp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_reject.restype = C.c_int64
        L.orc_params_to_H.argtypes = [_dp, _dp]
        L.orc_residuals.argtypes = [_dp, _dp, _fp, _dp, C.c_void_p, C.c_int64, _dp]
        L.orc_normal_equations.argtypes = [_dp, _dp, _fp, _dp, C.c_void_p, C.c_int64, _dp]
        L.orc_solve.argtypes = [_dp, C.c_double, _dp, _dp, _dp, _fp, _dp, C.c_void_p, C.c_int64, _dp, C.POINTER(C.c_int)]
        L.orc_uncertainties.argtypes = [_dp, C.c_double, _dp, _dp, _dp, _fp, _dp, C.c_void_p, C.c_int64, _dp]
        _lib = L
    return _lib


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


def transform(H, X):
    X = _c(X)
    out = np.empty_like(X)
    lib().orc_transform(_c(H).reshape(16), X, len(X), out)
    return out


def knn(P, Q, k=1, H=None, max_dist=np.inf, idx_base=0):
    """Brute-force k-NN, ascending (d2, idx).  Returns (idx (q,k) int64, d2 (q,k) f64)."""
    P, Q = _c(P), _c(Q)
    idx = np.empty((len(Q), k), np.int64)
    d2 = np.empty((len(Q), k), np.float64)
    Hp = None if H is None else _c(H).reshape(16).ctypes.data_as(C.c_void_p)
    Hk = None if H is None else _c(H).reshape(16)
    if Hk is not None:
        Hp = Hk.ctypes.data_as(C.c_void_p)
    lib().orc_knn(P, len(P), Hp, Q, len(Q), k, float(max_dist), idx_base, idx, d2)
    return idx, d2


def select_n_points(ns, n):
    pos = np.empty(max(n, 1), np.int64)
    m = lib().orc_select_n_points(ns, n, pos)
    return None if m < 0 else pos[:m]


def normals(P, nn_idx):
    nn_idx = _c(nn_idx, np.int64)
    q, k = nn_idx.shape
    nv = np.empty((q, 3), np.float32)
    pl = np.empty(q, np.float32)
    lib().orc_normals(_c(P), nn_idx, q, k, nv, pl)
    return nv, pl


def point_to_plane(p1, n1, p2, H):
    d = np.empty(len(p1))
    lib().orc_point_to_plane(_c(p1), _c(n1, np.float32), _c(p2), _c(H).reshape(16), len(p1), d)
    return d


def reject(d, planarity, min_planarity):
    keep = np.empty(len(d), np.uint8)
    med, mad = C.c_double(), C.c_double()
    n = lib().orc_reject(_c(d), _c(planarity, np.float32), len(d), float(min_planarity), keep,
                         C.byref(med), C.byref(mad))
    return keep.astype(bool), int(n), med.value, mad.value


def params_to_H(x):
    H = np.empty(16)
    lib().orc_params_to_H(_c(x), H)
    return H.reshape(4, 4)


def _keep_ptr(keep, q):
    if keep is None:
        return None, None
    k = _c(keep, np.uint8)
    assert len(k) == q
    return k, k.ctypes.data_as(C.c_void_p)


def residuals(x, p1, n1, p2, keep=None):
    k, kp = _keep_ptr(keep, len(p1))
    n = len(p1) if k is None else int(k.sum())
    r = np.empty(n)
    lib().orc_residuals(_c(x), _c(p1), _c(n1, np.float32), _c(p2), kp, len(p1), r)
    return r


def normal_equations(x, p1, n1, p2, keep=None):
    k, kp = _keep_ptr(keep, len(p1))
    out = np.empty(30)
    lib().orc_normal_equations(_c(x), _c(p1), _c(n1, np.float32), _c(p2), kp, len(p1), out)
    return out


def solve(x0, w, obs, ow, p1, n1, p2, keep=None):
    k, kp = _keep_ptr(keep, len(p1))
    x = np.empty(6)
    steps = C.c_int()
    lib().orc_solve(_c(x0), float(w), _c(obs), _c(ow), _c(p1), _c(n1, np.float32), _c(p2), kp, len(p1), x,
                    C.byref(steps))
    return x, steps.value


def uncertainties(x, w, obs, ow, p1, n1, p2, keep=None):
    k, kp = _keep_ptr(keep, len(p1))
    s = np.empty(6)
    rc = lib().orc_uncertainties(_c(x), float(w), _c(obs), _c(ow), _c(p1), _c(n1, np.float32), _c(p2), kp,
                                 len(p1), s)
    if rc != 0:
        raise RuntimeError("singular normal matrix")
    return s


def load_cloud(name):
    """tests/golden/data/<name>.npz -> (n,3) float64, bit-identical to np.genfromtxt of the .xyz."""
    q = np.load(_HERE.parent / "tests" / "golden" / "data" / f"{name}.npz")["q"]
    return q.astype(np.float64) / 1e4


def icp_iteration(X_mov, p1, n1, planarity, x_prev, x0, w, obs, ow, min_planarity, mov_sel=None, planarity_mov=None):
    """One ICP iteration per simpleicp.py:184-250 with the oracle's deterministic
    brute-force match.  Returns dict(nn, dist, keep, n, median, mad, x, residuals).
      mov_sel        rows of X_mov that are `selected` (corrpts.py:131-135 searches pc2.X_selected only and maps
                     the hits back through pc2.idx_selected); None = all
      planarity_mov  (len(X_mov),) float32 `planarity` column of the movable cloud, NaN where absent
                     (corrpts.py:158-163: a correspondence also needs pc2's planarity >= min_planarity when pc2
                     has that column); None = pc2 has no such column"""
    H = params_to_H(x_prev)
    if mov_sel is None:
        nn, _ = knn(X_mov, p1, k=1, H=H)
        nn = nn[:, 0]
    else:
        mov_sel = np.asarray(mov_sel, dtype=np.int64)
        nn, _ = knn(_c(X_mov)[mov_sel], p1, k=1, H=H)
        nn = mov_sel[nn[:, 0]]
    p2 = _c(X_mov)[nn]
    dist = point_to_plane(p1, n1, p2, H)
    if planarity_mov is not None:
        # both filters are row filters applied one after the other: a row survives iff both columns pass
        # (float32 compare, NaN fails) -- fold pc2's verdict into the planarity handed to orc_reject
        ok2 = np.asarray(planarity_mov, dtype=np.float32)[nn] >= np.float32(min_planarity)
        planarity = np.where(ok2, np.asarray(planarity, dtype=np.float32), np.float32(np.nan))
    keep, n, med, mad = reject(dist, planarity, min_planarity)
    if w is None:
        w = 1.0 / (np.std(dist[keep]) ** 2)
    x, steps = solve(x0, w, obs, ow, p1, n1, p2, keep)
    res = residuals(x, p1, n1, p2, keep)
    return dict(nn=nn, dist=dist, keep=keep, n=n, median=med, mad=mad, x=x, residuals=res, w=w, steps=steps)


def run(X_fix, X_mov, correspondences=1000, neighbors=10, min_planarity=0.3, max_overlap_distance=np.inf,
        min_change=1.0, max_iterations=100, distance_weights=1, rbp_observed_values=(0.,) * 6,
        rbp_observation_weights=(0.,) * 6, normals=None, planarity=None):
    """Whole simpleicp.py:135-324 loop on the oracle primitives (deterministic brute-force kNN,
    own normals with the fixed sign convention unless injected).  Returns a dict."""
    X_fix, X_mov = _c(X_fix), _c(X_mov)
    obs = np.array(rbp_observed_values, float)
    obs[:3] *= np.pi / 180
    ow = np.array(rbp_observation_weights, float)
    sel = np.arange(len(X_fix))
    if np.isfinite(max_overlap_distance):
        idx, _ = knn(X_mov, X_fix[sel], k=1, H=params_to_H(obs), max_dist=max_overlap_distance)
        sel = sel[idx[:, 0] >= 0]
    pos = select_n_points(len(sel), correspondences)
    if pos is not None:
        sel = np.unique(sel[pos])
    if normals is None:
        nn, _ = knn(X_fix, X_fix[sel], k=neighbors)
        normals, planarity = _normals(X_fix, nn)
    x, w, stats, r = obs.copy(), distance_weights, [], None
    for it in range(max_iterations):
        r = icp_iteration(X_mov, X_fix[sel], normals, planarity, x, x, w, obs, ow, min_planarity)
        w, x = r["w"], r["x"]
        stats.append((r["n"], r["residuals"].mean(), r["residuals"].std()))
        if it > 0:
            def ch(a, b):
                return (0.0 if a == 0 else np.inf) if b == 0 else abs((a - b) / b * 100)
            if ch(stats[it][1], stats[it - 1][1]) < min_change and ch(stats[it][2], stats[it - 1][2]) < min_change:
                break
    sigma = uncertainties(x, w, obs, ow, X_fix[sel], normals, X_mov[r["nn"]], r["keep"])
    return dict(H=params_to_H(x), x=x, sigma=sigma, sel=sel, normals=normals, planarity=planarity,
                iterations=len(stats), stats=stats, residuals=r["residuals"])


_normals = normals
