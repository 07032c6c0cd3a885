"""A stand-in for ``simpleicp_amd._lib.Context`` that answers every call with the CPU oracle (oracle/orc.py).

TEST INFRASTRUCTURE ONLY.  It lets the CPU test run (``-m "not gpu"``) exercise the HOST logic of the Python mirror --
``SimpleICP.run``'s orchestration, logging and side effects, ``PointCloud``'s operators, the bookkeeping of
``CorrPts`` / ``SimpleICPOptimization`` -- against the fixtures of the unmodified reference without a GPU.  The
product never sees this file: ``simpleicp_amd`` has no CPU path (tests/test_host_api.py pins that), and the HIP
kernels themselves are checked on the GPU box by the ``-m gpu`` tests, which run the same flows on the real library.
"""
import numpy as np

from oracle import orc
from simpleicp_amd import _lib

TOO_FEW = ("Too few correspondences! At least 6 correspondences are needed to estimate the 6 rigid body "
           "transformation parameters. The current number of correspondences is {}.")


class OracleContext:
    device = 0

    def __init__(self):
        self.cloud = {}                  # slot -> (xyz (n,3) f64, index_base)
        self.pl = {}                     # slot -> dense float32 planarity by global index
        self.calls = []                  # names of the entry points used, in order
        self._have_corr = False
        self._have_iter = False

    # ---- clouds ---------------------------------------------------------------------------------
    def _log(self, name):
        self.calls.append(name)

    def upload(self, slot, xyz, index_base=0):
        self._log("upload")
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        if not np.isfinite(xyz).all():
            raise _lib.BackendError("cloud has non-finite coordinates", _lib.ERR_INVALID)
        self.cloud[slot] = (xyz.copy(), int(index_base))
        self.pl.pop(slot, None)

    def upload_columns(self, slot, x, y, z, index_base=0):
        self.upload(slot, np.column_stack((x, y, z)), index_base)

    def upload_start(self, slot, xyz=None, columns=None, index_base=0):
        """Like the library: the verdict of a background upload is handed over by upload_wait / the next call on the slot."""
        self._log("upload_start")
        try:
            self.upload(slot, xyz if xyz is not None else np.column_stack(columns), index_base)
            self.calls.pop()                      # (the inner "upload")
            self.__dict__.setdefault("_bg_err", {}).pop(slot, None)
        except _lib.BackendError as e:
            self.calls.pop()
            self.__dict__.setdefault("_bg_err", {})[slot] = e

    def upload_wait(self, slot):
        e = self.__dict__.setdefault("_bg_err", {}).pop(slot, None)
        if e is not None:
            raise e

    def size(self, slot):
        return len(self.cloud[slot][0])

    def transform(self, slot, H):
        self._log("transform")
        X, base = self.cloud[slot]
        self.cloud[slot] = (orc.transform(np.asarray(H, dtype=np.float64).reshape(4, 4), X), base)

    def download(self, slot):
        return self.cloud[slot][0].copy()

    def download_columns(self, slot):
        X = self.cloud[slot][0]
        return [np.ascontiguousarray(X[:, j]) for j in range(3)]

    def download_both(self, slot):
        return self.download(slot), self.download_columns(slot)

    def set_planarity(self, slot, planarity=None, rows=None, n_global=None):
        self._log("set_planarity")
        if planarity is None:
            self.pl.pop(slot, None)
            return
        pl = np.asarray(planarity, dtype=np.float32)
        if rows is None:
            self.pl[slot] = pl.copy()
        else:
            dense = np.full(int(n_global if n_global is not None else self.size(slot)), np.nan, np.float32)
            dense[np.asarray(rows, dtype=np.int64)] = pl
            self.pl[slot] = dense

    # ---- nearest neighbours / attributes ------------------------------------------------------------
    def knn(self, slot, q_xyz, k=1, H=None, max_dist=np.inf):
        X, base = self.cloud[slot]
        return orc.knn(X, np.asarray(q_xyz, dtype=np.float64), k=k, H=H, max_dist=max_dist, idx_base=base)

    def select_in_range(self, query_slot, search_slot, sel=None, H=None, max_range=np.inf):
        self._log("select_in_range")
        Xq = self.cloud[query_slot][0]
        if sel is not None:
            Xq = Xq[np.asarray(sel, dtype=np.int64)]
        idx, _ = orc.knn(self.cloud[search_slot][0], Xq, k=1, H=H, max_dist=max_range)
        return idx[:, 0] >= 0

    def estimate_normals(self, slot, sel_idx, k, want_nn=False):
        self._log("estimate_normals")
        X, _ = self.cloud[slot]
        nn, _ = orc.knn(X, X[np.asarray(sel_idx, dtype=np.int64)], k=int(k))
        nv, pl = orc.normals(X, nn)
        return (nv, pl, nn) if want_nn else (nv, pl)

    # ---- the iteration -------------------------------------------------------------------------------
    def icp_setup(self, sel_idx, normals, planarity):
        self._log("icp_setup")
        self._sel = np.asarray(sel_idx, dtype=np.int64)
        self._p1 = self.cloud[_lib.FIX][0][self._sel]
        self._n1 = np.ascontiguousarray(normals, dtype=np.float32)
        self._pl1 = np.ascontiguousarray(planarity, dtype=np.float32)
        self._Q = len(self._sel)
        self._have_corr = self._have_iter = False

    def _result(self, **kw):
        R = _lib.IterResult()
        for k, v in kw.items():
            if k in ("x", "H"):
                for i, e in enumerate(np.asarray(v, dtype=float).ravel()):
                    getattr(R, k)[i] = float(e)
            else:
                setattr(R, k, v)
        return R

    def icp_iterate(self, x, obs, obs_weight, min_planarity=0.3, distance_weight=1.0, max_lm_steps=0):
        self._log("icp_iterate")
        return self._iterate(np.array(x, float), np.array(obs, float), np.array(obs_weight, float), min_planarity, distance_weight)

    def _iterate(self, x, obs, ow, min_planarity, w):
        Xm, base = self.cloud[_lib.MOV]
        H = orc.params_to_H(x)
        nn, _ = orc.knn(Xm, self._p1, k=1, H=H)
        nn = nn[:, 0]
        p2 = Xm[nn]
        dist = orc.point_to_plane(self._p1, self._n1, p2, H)
        pl = self._pl1
        if _lib.MOV in self.pl:
            ok2 = self.pl[_lib.MOV][nn + base] >= np.float32(min_planarity)
            pl = np.where(ok2, pl, np.float32(np.nan))
        keep, n, med, mad = orc.reject(dist, pl, min_planarity)
        self._idx, self._p2, self._dist, self._alive = nn + base, p2, dist, keep
        self._resid = np.zeros(self._Q)
        self._have_iter, self._have_corr = True, False
        self._last = None
        n_planar = int(np.count_nonzero(pl >= np.float32(min_planarity)))
        if n < 6:
            err = _lib.BackendError(TOO_FEW.format(n), _lib.ERR_TOO_FEW)
            err.result = self._result(x=x, n_queries=self._Q, n_planar=n_planar, n_kept=int(n), median=med, mad=mad)
            raise err
        R = self._solve(x, obs, ow, w)
        R.n_planar, R.median, R.mad = n_planar, med, mad
        return R

    def _solve(self, x0, obs, ow, w):
        keep, dist = self._alive, self._dist
        n = int(np.count_nonzero(keep))
        if w is None or not (w > 0):
            w = 1.0 / (np.std(dist[keep]) ** 2)
        x, steps = orc.solve(x0, w, obs, ow, self._p1, self._n1, self._p2, keep)
        res = orc.residuals(x, self._p1, self._n1, self._p2, keep)
        self._resid = np.zeros(self._Q)
        self._resid[keep] = res
        self._last = (x, w, obs, ow)
        o = ow[(ow > 0) & np.isfinite(ow)] * (x - obs)[(ow > 0) & np.isfinite(ow)]
        return self._result(x=x, H=orc.params_to_H(x), n_queries=self._Q, n_planar=n, n_kept=n, median=np.nan, mad=np.nan,
                            dist_mean=float(dist[keep].mean()), dist_std=float(dist[keep].std()),
                            res_mean=float(res.mean()), res_std=float(res.std()), weight_used=float(w),
                            cost=float(np.sum((w * res) ** 2) + np.sum(o * o)), lm_steps=int(steps), ne_evals=int(steps) + 2)

    def icp_run(self, x, obs, obs_weight, min_planarity=0.3, distance_weight=1.0, max_iterations=100, min_change=1.0,
                max_lm_steps=0):
        self._log("icp_run")
        x, obs, ow = np.array(x, float), np.array(obs, float), np.array(obs_weight, float)
        w, out = distance_weight, []

        def change(now, before):
            if before == 0:
                return 0.0 if now == 0 else np.inf
            return abs((now - before) / before * 100.0)
        for it in range(int(max_iterations)):
            try:
                R = self._iterate(x, obs, ow, min_planarity, w)
            except _lib.BackendError as e:
                e.results = out + [e.result]
                raise
            out.append(R)
            x = np.array(R.x[:])
            if w is None:
                w = R.weight_used
            if it > 0 and change(R.res_mean, out[-2].res_mean) < min_change and change(R.res_std, out[-2].res_std) < min_change:
                break
        return out

    def icp_state(self, pc2_idx=True, dist=True, keep=True, residual=True):
        if not (self._have_iter or self._have_corr):
            raise _lib.BackendError("no iteration has run yet", _lib.ERR_INVALID)
        return (self._idx.copy() if pc2_idx else None, self._dist.copy() if dist else None,
                self._alive.copy() if keep else None, self._resid.copy() if residual else None)

    def icp_uncertainties(self):
        if not self._have_iter or self._last is None:
            raise _lib.BackendError("no iteration has run yet", _lib.ERR_INVALID)
        x, w, obs, ow = self._last
        return orc.uncertainties(x, w, obs, ow, self._p1, self._n1, self._p2, self._alive)

    # ---- the iteration's operators one by one ------------------------------------------------------------
    def corr_match(self, H=None):
        self._log("corr_match")
        Xm, base = self.cloud[_lib.MOV]
        Hm = np.eye(4) if H is None else np.asarray(H, dtype=np.float64).reshape(4, 4)
        nn, _ = orc.knn(Xm, self._p1, k=1, H=Hm)
        nn = nn[:, 0]
        self._idx, self._p2 = nn + base, Xm[nn]
        self._dist = orc.point_to_plane(self._p1, self._n1, self._p2, Hm)
        self._alive = np.ones(self._Q, bool)
        self._resid = np.zeros(self._Q)
        self._have_corr, self._have_iter, self._last = True, False, None
        return self._idx.copy(), self._dist.copy()

    def _need_corr(self):
        if not self._have_corr:
            raise _lib.BackendError("call sicp_corr_match first", _lib.ERR_INVALID)

    def corr_reject_planarity(self, min_planarity, pc1_planarity=None, pc2_planarity=None):
        self._log("corr_reject_planarity")
        self._need_corr()
        for col in (pc1_planarity, pc2_planarity):
            if col is not None:
                self._alive &= np.asarray(col, dtype=np.float32) >= np.float32(min_planarity)
        return int(self._alive.sum())

    def corr_reject_distances(self):
        self._log("corr_reject_distances")
        self._need_corr()
        pl = np.where(self._alive, np.float32(1), np.float32(np.nan))
        keep, n, med, mad = orc.reject(self._dist, pl, 0.0)
        self._alive = keep
        return med, mad, int(n)

    def estimate_parameters(self, x, obs, obs_weight, distance_weight=1.0, pc2_xyz=None, max_lm_steps=0):
        self._log("estimate_parameters")
        self._need_corr()
        if pc2_xyz is not None:
            self._p2 = np.ascontiguousarray(pc2_xyz, dtype=np.float64)
        n = int(self._alive.sum())
        if n < 6:
            err = _lib.BackendError(TOO_FEW.format(n), _lib.ERR_TOO_FEW)
            err.result = self._result(x=x, n_queries=self._Q, n_kept=n)
            raise err
        R = self._solve(np.array(x, float), np.array(obs, float), np.array(obs_weight, float), distance_weight)
        self._have_iter = True
        return R

    # ---- plumbing the mirror touches -------------------------------------------------------------------------
    def set_exchange(self, fn, rank, world, gn_shard=False):
        pass

    def comm_destroy(self):
        pass

    def comm_activate(self, on=True, gn_shard=False):
        pass

    def set_partition(self, mode):
        pass

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass


def install(monkeypatch):
    """Route ``backend.get_context()`` (what PointCloud / SimpleICP / CorrPts use) to a fresh OracleContext."""
    from simpleicp_amd import backend
    ctx = OracleContext()
    monkeypatch.setattr(backend, "get_context", lambda: ctx)
    return ctx
