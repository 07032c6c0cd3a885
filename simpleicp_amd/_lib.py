"""ctypes binding of libsimpleicp_hip.so (C ABI: include/simpleicp_hip.h).

There is NO CPU fallback anywhere in this package: if the shared library is missing, or no
gfx950 device is visible, the operations raise ``BackendError`` -- they never silently run on
the host.  (The CPU oracle under oracle/ is test infrastructure and is not imported here.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
# SICP_LIBRARY: load another build of the same ABI (the sanitizer build of tests/test_asan.py)
LIB_PATH = Path(os.environ["SICP_LIBRARY"]) if os.environ.get("SICP_LIBRARY") else PKG / "libsimpleicp_hip.so"

FIX, MOV = 0, 1
OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_TOO_FEW, ERR_NUMERIC, ERR_EXCHANGE = 0, -1, -2, -3, -4, -5, -6
XCHG_ALLGATHER_F64, XCHG_SUM_F64, XCHG_MIN_U64, XCHG_MAX_U64 = 1, 2, 3, 4
PART_CLOUD, PART_QUERIES = 0, 1
K_KNN1, K_KNNK, K_NORMALEQ, K_SELECT, K_XCHG = 0, 1, 2, 3, 4
ABI_VERSION = 7          # include/simpleicp_hip.h SICP_ABI_VERSION this binding was written for
KERNEL_NAMES = {K_KNN1: "match", K_KNNK: "knnk_scan", K_NORMALEQ: "solve", K_SELECT: "reject_select", K_XCHG: "exchange"}
MATCH_KERNELS = {0: "k_knn1_scan", 1: "k_knn1_fscan", 2: "k_grid_nn", 3: "k_knn1_frec", 5: "k_grid_nn16", 6: "k_grid_nn16f"}

EXPORTS = [
    "sicp_abi_version", "sicp_last_error", "sicp_device_count", "sicp_ctx_create", "sicp_ctx_destroy",
    "sicp_ctx_device_name", "sicp_cloud_upload", "sicp_cloud_upload_columns", "sicp_cloud_upload_start", "sicp_cloud_upload_wait", "sicp_cloud_size", "sicp_cloud_transform",
    "sicp_cloud_download", "sicp_cloud_download_columns", "sicp_cloud_download_both", "sicp_cloud_set_planarity", "sicp_knn", "sicp_select_in_range", "sicp_estimate_normals", "sicp_icp_setup", "sicp_icp_iterate",
    "sicp_icp_run", "sicp_icp_get_state", "sicp_icp_uncertainties", "sicp_icp_normal_equations", "sicp_params_to_H",
    "sicp_corr_match", "sicp_corr_reject_planarity", "sicp_corr_reject_distances", "sicp_estimate_parameters",
    "sicp_set_exchange", "sicp_comm_unique_id", "sicp_comm_init", "sicp_comm_destroy", "sicp_comm_activate", "sicp_comm_info", "sicp_device_memory", "sicp_set_partition", "sicp_ctx_stream", "sicp_lexmin_gathered", "sicp_timing_enable", "sicp_timing_reset", "sicp_timing_get", "sicp_match_work", "sicp_match_deferred", "sicp_tail_cycles", "sicp_tail_selection", "sicp_exchange_info", "sicp_knn_work", "sicp_last_match_kernel",
    "sicp_xyz_count", "sicp_xyz_read", "sicp_xyz_write",
]


class BackendError(RuntimeError):
    """The HIP backend is unavailable or a HIP call failed."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


class IterParams(C.Structure):
    _fields_ = [("x", C.c_double * 6), ("obs", C.c_double * 6), ("obs_weight", C.c_double * 6),
                ("min_planarity", C.c_double), ("distance_weight", C.c_double), ("max_lm_steps", C.c_int64)]


class IterResult(C.Structure):
    _fields_ = [("x", C.c_double * 6), ("H", C.c_double * 16), ("n_queries", C.c_int64),
                ("n_planar", C.c_int64), ("n_kept", C.c_int64), ("median", C.c_double), ("mad", C.c_double),
                ("dist_mean", C.c_double), ("dist_std", C.c_double), ("res_mean", C.c_double),
                ("res_std", C.c_double), ("weight_used", C.c_double), ("cost", C.c_double),
                ("lm_steps", C.c_int64), ("ne_evals", C.c_int64)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)

_lib = None


def load():
    """dlopen the library (building it first if hipcc is around and it is stale)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        try:
            from . import build as _build
            _build.build()
        except Exception as exc:  # noqa: BLE001
            raise BackendError(f"{LIB_PATH} is missing and could not be built ({exc}); "
                               "run `python -m simpleicp_amd.build`") from exc
    try:
        L = C.CDLL(str(LIB_PATH))
    except OSError as exc:
        raise BackendError(f"cannot load {LIB_PATH}: {exc}") from exc
    vp, i64, dbl, cint = C.c_void_p, C.c_int64, C.c_double, C.c_int
    L.sicp_abi_version.restype = cint
    if L.sicp_abi_version() != ABI_VERSION:
        # (SICP_LIBRARY makes it easy to point at a stale build: its entry points would be called with the wrong arguments)
        raise BackendError(f"{LIB_PATH} implements ABI version {L.sicp_abi_version()}, this binding needs {ABI_VERSION}; "
                           "rebuild with `python -m simpleicp_amd.build`")
    L.sicp_last_error.restype = C.c_char_p
    L.sicp_device_count.argtypes = [C.POINTER(cint)]
    L.sicp_ctx_create.argtypes = [cint, C.POINTER(vp)]
    L.sicp_ctx_destroy.argtypes = [vp]
    L.sicp_ctx_device_name.argtypes = [vp, C.c_char_p, cint]
    L.sicp_cloud_upload.argtypes = [vp, cint, vp, i64, i64]
    L.sicp_cloud_upload_columns.argtypes = [vp, cint, vp, vp, vp, i64, i64]
    L.sicp_cloud_upload_start.argtypes = [vp, cint, vp, vp, vp, vp, i64, i64]
    L.sicp_cloud_upload_wait.argtypes = [vp, cint]
    L.sicp_cloud_size.argtypes = [vp, cint, C.POINTER(i64)]
    L.sicp_cloud_transform.argtypes = [vp, cint, vp]
    L.sicp_cloud_download.argtypes = [vp, cint, vp]
    L.sicp_cloud_download_columns.argtypes = [vp, cint, vp, vp, vp]
    L.sicp_cloud_download_both.argtypes = [vp, cint, vp, vp, vp, vp]
    L.sicp_cloud_set_planarity.argtypes = [vp, cint, vp, vp, i64, i64]
    L.sicp_knn.argtypes = [vp, cint, vp, i64, cint, vp, dbl, vp, vp]
    L.sicp_select_in_range.argtypes = [vp, cint, cint, vp, i64, vp, dbl, vp]
    L.sicp_estimate_normals.argtypes = [vp, cint, vp, i64, cint, vp, vp, vp]
    L.sicp_icp_setup.argtypes = [vp, vp, i64, vp, vp]
    L.sicp_icp_iterate.argtypes = [vp, C.POINTER(IterParams), C.POINTER(IterResult)]
    L.sicp_icp_run.argtypes = [vp, C.POINTER(IterParams), i64, dbl, C.POINTER(IterResult), C.POINTER(i64)]
    L.sicp_icp_get_state.argtypes = [vp, vp, vp, vp, vp]
    L.sicp_icp_uncertainties.argtypes = [vp, vp]
    L.sicp_icp_normal_equations.argtypes = [vp, vp, vp]
    L.sicp_params_to_H.argtypes = [vp, vp]
    L.sicp_corr_match.argtypes = [vp, vp, vp, vp]
    L.sicp_corr_reject_planarity.argtypes = [vp, dbl, vp, vp, C.POINTER(i64)]
    L.sicp_corr_reject_distances.argtypes = [vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(i64)]
    L.sicp_estimate_parameters.argtypes = [vp, C.POINTER(IterParams), vp, C.POINTER(IterResult)]
    L.sicp_set_exchange.argtypes = [vp, EXCHANGE_FN, vp, cint, cint, cint]
    L.sicp_comm_unique_id.argtypes = [vp]
    L.sicp_comm_init.argtypes = [vp, vp, cint, cint, cint]
    L.sicp_comm_destroy.argtypes = [vp]
    L.sicp_comm_activate.argtypes = [vp, cint, cint]
    L.sicp_comm_info.argtypes = [vp, C.POINTER(cint * 6)]
    L.sicp_device_memory.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.sicp_set_partition.argtypes = [vp, cint]
    L.sicp_ctx_stream.argtypes = [vp, C.POINTER(vp)]
    L.sicp_lexmin_gathered.argtypes = [vp, vp, cint, i64, vp, vp, vp]
    L.sicp_timing_enable.argtypes = [vp, cint]
    L.sicp_timing_reset.argtypes = [vp]
    L.sicp_match_work.argtypes = [vp, vp]
    L.sicp_match_deferred.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.sicp_tail_cycles.argtypes = [vp, vp]
    L.sicp_tail_selection.argtypes = [vp, vp]
    L.sicp_exchange_info.argtypes = [vp, vp]
    L.sicp_knn_work.argtypes = [vp, vp]
    L.sicp_last_match_kernel.argtypes = [vp, C.POINTER(cint)]
    L.sicp_xyz_count.argtypes = [C.c_char_p, C.POINTER(i64)]
    L.sicp_xyz_read.argtypes = [C.c_char_p, vp, i64, C.POINTER(i64), cint]
    L.sicp_xyz_write.argtypes = [C.c_char_p, vp, i64, cint, cint, C.c_char_p, cint]
    L.sicp_timing_get.argtypes = [vp, cint, C.POINTER(dbl), C.POINTER(i64)]
    for name in EXPORTS:
        if name != "sicp_last_error":
            getattr(L, name).restype = cint
    _lib = L
    return L


def _ptr(a):
    """numpy array / torch tensor / None -> void* (host-or-device pointer)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):          # torch.Tensor (host or device)
        return C.c_void_p(a.data_ptr())
    raise TypeError(f"unsupported buffer type {type(a)}")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = C.c_int(0)
    load().sicp_device_count(C.byref(n))
    return n.value


def params_to_H(x):
    H = np.empty(16)
    load().sicp_params_to_H(_ptr(_f64(x)), _ptr(H))
    return H.reshape(4, 4)


class Context:
    """One GPU context (= one `sicp_ctx`): device-resident clouds + ICP iteration state."""

    def __init__(self, device=0):
        self._L = load()
        self._h = C.c_void_p()
        self._cb = None
        rc = self._L.sicp_ctx_create(int(device), C.byref(self._h))
        if rc != OK:
            self._h = C.c_void_p()
            raise BackendError(self._L.sicp_last_error().decode(), rc)
        self.device = int(device)
        self._corr_owner = None      # the CorrPts object whose correspondences the context holds (simpleicp_amd/corrpts.py)
        self._bg_src = {}            # slot -> the host arrays an upload_start is still reading (kept alive until the slot's next upload)

    # -- plumbing --
    def _chk(self, rc):
        if rc != OK:
            raise BackendError(self._L.sicp_last_error().decode(), rc)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.sicp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._chk(self._L.sicp_ctx_device_name(self._h, buf, 256))
        return buf.value.decode()

    # -- clouds --
    def upload(self, slot, xyz, index_base=0):
        """xyz: (n,3) float64 numpy array or CUDA/host torch tensor."""
        if isinstance(xyz, np.ndarray):
            xyz = _f64(xyz)
        n = int(xyz.shape[0])
        if tuple(xyz.shape) != (n, 3):
            raise ValueError("cloud must have shape (n, 3)")
        self._chk(self._L.sicp_cloud_upload(self._h, slot, _ptr(xyz), n, int(index_base)))

    def upload_columns(self, slot, x, y, z, index_base=0):
        """x, y, z: contiguous float64 vectors of one length (no (n,3) gather on the host)."""
        cols = [_f64(np.asarray(v)) for v in (x, y, z)]
        n = len(cols[0])
        if any(v.ndim != 1 or len(v) != n for v in cols):
            raise ValueError("x, y, z must be vectors of the same length")
        self._chk(self._L.sicp_cloud_upload_columns(self._h, slot, _ptr(cols[0]), _ptr(cols[1]), _ptr(cols[2]), n,
                                                    int(index_base)))

    def upload_start(self, slot, xyz=None, columns=None, index_base=0):
        """The same upload running BEHIND the caller (sicp_cloud_upload_start): (n,3) rows OR three contiguous columns.  Returns once
        the device arrays are sized; any later call naming the slot waits for it first (and raises its error).  The source arrays are
        kept alive here; the caller must not write to them before `upload_wait` / the next call on the slot."""
        if (xyz is None) == (columns is None):
            raise ValueError("xyz OR columns")
        if xyz is not None:
            src = [_f64(xyz) if isinstance(xyz, np.ndarray) else xyz]
            n = int(src[0].shape[0])
            if tuple(src[0].shape) != (n, 3):
                raise ValueError("cloud must have shape (n, 3)")
            args = (_ptr(src[0]), None, None, None)
        else:
            src = [_f64(np.asarray(v)) for v in columns]
            n = len(src[0])
            if len(src) != 3 or any(v.ndim != 1 or len(v) != n for v in src):
                raise ValueError("x, y, z must be vectors of the same length")
            args = (None, _ptr(src[0]), _ptr(src[1]), _ptr(src[2]))
        self._bg_src[slot] = src
        self._chk(self._L.sicp_cloud_upload_start(self._h, slot, *args, n, int(index_base)))

    def upload_wait(self, slot):
        try:
            self._chk(self._L.sicp_cloud_upload_wait(self._h, slot))
        finally:
            self._bg_src.pop(slot, None)

    def size(self, slot):
        n = C.c_int64()
        self._chk(self._L.sicp_cloud_size(self._h, slot, C.byref(n)))
        return n.value

    def transform(self, slot, H):
        self._chk(self._L.sicp_cloud_transform(self._h, slot, _ptr(_f64(H).reshape(16))))

    def download(self, slot):
        out = np.empty((self.size(slot), 3))
        self._chk(self._L.sicp_cloud_download(self._h, slot, _ptr(out)))
        return out

    def download_columns(self, slot):
        """The cloud as three contiguous float64 vectors (x, y, z)."""
        n = self.size(slot)
        cols = [np.empty(n) for _ in range(3)]
        self._chk(self._L.sicp_cloud_download_columns(self._h, slot, _ptr(cols[0]), _ptr(cols[1]), _ptr(cols[2])))
        return cols

    def download_both(self, slot):
        """((n, 3) array, [x, y, z] vectors) in ONE pass over the link (pinned, pipelined, transposed by host threads)."""
        n = self.size(slot)
        out = np.empty((n, 3))
        cols = [np.empty(n) for _ in range(3)]
        self._chk(self._L.sicp_cloud_download_both(self._h, slot, _ptr(out), _ptr(cols[0]), _ptr(cols[1]), _ptr(cols[2])))
        return out, cols

    def set_planarity(self, slot, planarity=None, rows=None, n_global=None):
        """The cloud's `planarity` column (corrpts.py:158-163 tests the movable cloud's too): a dense float32 vector
        by global point index, or (rows, values) pairs with NaN elsewhere; None = no such column."""
        if planarity is None:
            self._chk(self._L.sicp_cloud_set_planarity(self._h, slot, None, None, 0, 0))
            return
        pl = np.ascontiguousarray(planarity, dtype=np.float32)
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.int64)
        if r is not None and len(r) != len(pl):
            raise ValueError("rows and planarity must have the same length")
        n = int(n_global if n_global is not None else (len(pl) if r is None else self.size(slot)))
        dummy = np.zeros(1, np.float32)
        self._chk(self._L.sicp_cloud_set_planarity(self._h, slot, _ptr(r), _ptr(pl if len(pl) else dummy), len(pl), n))

    # -- nearest neighbours --
    def knn(self, slot, q_xyz, k=1, H=None, max_dist=np.inf):
        q = _f64(q_xyz)
        Q = len(q)
        idx = np.empty((Q, k), np.int64)
        d2 = np.empty((Q, k), np.float64)
        Hc = None if H is None else _f64(H).reshape(16)
        self._chk(self._L.sicp_knn(self._h, slot, _ptr(q), Q, int(k), _ptr(Hc), float(max_dist), _ptr(idx), _ptr(d2)))
        return idx, d2

    def select_in_range(self, query_slot, search_slot, sel=None, H=None, max_range=np.inf):
        """bool mask over `sel` (all points of query_slot when None): nearest neighbour in search_slot (under H)
        closer than max_range -- both clouds already resident (sicp_select_in_range)."""
        if sel is not None:
            sel = np.ascontiguousarray(sel, dtype=np.int64)
        Q = self.size(query_slot) if sel is None else len(sel)
        out = np.empty(Q, dtype=np.uint8)
        Hp = None if H is None else _f64(H).reshape(16)
        self._chk(self._L.sicp_select_in_range(self._h, query_slot, search_slot, _ptr(sel), Q, _ptr(Hp), float(max_range),
                                               _ptr(out)))
        return out.view(np.bool_)

    def estimate_normals(self, slot, sel_idx, k, want_nn=False):
        sel = np.ascontiguousarray(sel_idx, dtype=np.int64)
        Q = len(sel)
        nv = np.empty((Q, 3), np.float32)
        pl = np.empty(Q, np.float32)
        nn = np.empty((Q, k), np.int64) if want_nn else None
        self._chk(self._L.sicp_estimate_normals(self._h, slot, _ptr(sel), Q, int(k), _ptr(nv), _ptr(pl), _ptr(nn)))
        return (nv, pl, nn) if want_nn else (nv, pl)

    # -- ICP iteration --
    def icp_setup(self, sel_idx, normals, planarity):
        sel = np.ascontiguousarray(sel_idx, dtype=np.int64)
        nv = np.ascontiguousarray(normals, dtype=np.float32)
        pl = np.ascontiguousarray(planarity, dtype=np.float32)
        if nv.shape != (len(sel), 3) or pl.shape != (len(sel),):
            raise ValueError("normals must be (Q,3) and planarity (Q,)")
        self._chk(self._L.sicp_icp_setup(self._h, _ptr(sel), len(sel), _ptr(nv), _ptr(pl)))
        self._Q = len(sel)

    def icp_iterate(self, x, obs, obs_weight, min_planarity=0.3, distance_weight=1.0, max_lm_steps=0):
        """One iteration; distance_weight None = automatic (simpleicp.py:233-234).  Returns IterResult;
        raises BackendError(code=ERR_TOO_FEW) when fewer than 6 correspondences survive."""
        P = IterParams((C.c_double * 6)(*x), (C.c_double * 6)(*obs), (C.c_double * 6)(*obs_weight),
                       min_planarity, -1.0 if distance_weight is None else distance_weight, int(max_lm_steps))
        R = IterResult()
        rc = self._L.sicp_icp_iterate(self._h, C.byref(P), C.byref(R))
        if rc != OK:
            err = BackendError(self._L.sicp_last_error().decode(), rc)
            err.result = R
            raise err
        return R

    def icp_run(self, x, obs, obs_weight, min_planarity=0.3, distance_weight=1.0, max_iterations=100, min_change=1.0,
                max_lm_steps=0):
        """The whole loop in one ABI call (sicp_icp_run).  Returns the list of per-iteration IterResult;
        raises BackendError (with `.results`) like icp_iterate when an iteration fails."""
        P = IterParams((C.c_double * 6)(*x), (C.c_double * 6)(*obs), (C.c_double * 6)(*obs_weight),
                       min_planarity, -1.0 if distance_weight is None else distance_weight, int(max_lm_steps))
        n = int(max_iterations)
        res = (IterResult * max(n, 1))()
        done = C.c_int64()
        rc = self._L.sicp_icp_run(self._h, C.byref(P), n, float(min_change), res, C.byref(done))
        out = [res[i] for i in range(done.value)]
        if rc != OK:
            err = BackendError(self._L.sicp_last_error().decode(), rc)
            err.results = out
            raise err
        return out

    def icp_state(self, pc2_idx=True, dist=True, keep=True, residual=True):
        Q = self._Q
        a = np.empty(Q, np.int64) if pc2_idx else None
        b = np.empty(Q, np.float64) if dist else None
        c = np.empty(Q, np.uint8) if keep else None
        d = np.empty(Q, np.float64) if residual else None
        self._chk(self._L.sicp_icp_get_state(self._h, _ptr(a), _ptr(b), _ptr(c), _ptr(d)))
        return a, b, (None if c is None else c.astype(bool)), d

    def icp_uncertainties(self):
        s = np.empty(6)
        self._chk(self._L.sicp_icp_uncertainties(self._h, _ptr(s)))
        return s

    def icp_normal_equations(self, x):
        out = np.empty(30)
        self._chk(self._L.sicp_icp_normal_equations(self._h, _ptr(_f64(x)), _ptr(out)))
        return out

    # -- the iteration's operators one by one (CorrPts / SimpleICPOptimization, corrpts.py / optimization.py) --
    def corr_match(self, H=None):
        """CorrPts.match: (pc2_idx (Q) int64, point-to-plane distances (Q)) of the points declared by icp_setup in the
        movable slot under H (None = identity); every correspondence is alive afterwards."""
        Q = self._Q
        idx, dist = np.empty(Q, np.int64), np.empty(Q, np.float64)
        Hc = None if H is None else _f64(H).reshape(16)
        self._chk(self._L.sicp_corr_match(self._h, _ptr(Hc), _ptr(idx), _ptr(dist)))
        return idx, dist

    def corr_reject_planarity(self, min_planarity, pc1_planarity=None, pc2_planarity=None):
        """CorrPts.reject_wrt_planarity; the columns per correspondence ((Q) float32, None = the cloud has none).
        Returns the number of correspondences still alive."""
        def col(v):
            if v is None:
                return None
            v = np.ascontiguousarray(v, dtype=np.float32)
            if v.shape != (self._Q,):
                raise ValueError("planarity columns must have one value per correspondence")
            return v
        p1, p2 = col(pc1_planarity), col(pc2_planarity)
        n = C.c_int64()
        self._chk(self._L.sicp_corr_reject_planarity(self._h, float(min_planarity), _ptr(p1), _ptr(p2), C.byref(n)))
        return n.value

    def corr_reject_distances(self):
        """CorrPts.reject_wrt_point_to_plane_distances over the alive correspondences: (median, mad, n_alive)."""
        med, mad, n = C.c_double(), C.c_double(), C.c_int64()
        self._chk(self._L.sicp_corr_reject_distances(self._h, C.byref(med), C.byref(mad), C.byref(n)))
        return med.value, mad.value, n.value

    def estimate_parameters(self, x, obs, obs_weight, distance_weight=1.0, pc2_xyz=None, max_lm_steps=0):
        """SimpleICPOptimization.estimate_parameters over the alive correspondences, from x; pc2_xyz: (Q,3) current
        coordinates of the matched movable points (None = as matched).  Returns IterResult."""
        P = IterParams((C.c_double * 6)(*x), (C.c_double * 6)(*obs), (C.c_double * 6)(*obs_weight),
                       0.0, -1.0 if distance_weight is None else distance_weight, int(max_lm_steps))
        R = IterResult()
        p2 = None
        if pc2_xyz is not None:
            p2 = _f64(pc2_xyz)
            if p2.shape != (self._Q, 3):
                raise ValueError("pc2_xyz must be (Q, 3)")
        rc = self._L.sicp_estimate_parameters(self._h, C.byref(P), _ptr(p2), C.byref(R))
        if rc != OK:
            err = BackendError(self._L.sicp_last_error().decode(), rc)
            err.result = R
            raise err
        return R

    def stream_ptr(self):
        """Raw hipStream_t of the context (for torch.cuda.ExternalStream)."""
        p = C.c_void_p()
        self._chk(self._L.sicp_ctx_stream(self._h, C.byref(p)))
        return p.value or 0

    # -- multi-GPU exchange hook --
    def set_exchange(self, fn, rank, world, gn_shard=False):
        """fn(what, a_ptr, b_ptr, c_ptr, count) -> 0 on success; pointers are device addresses."""
        if fn is None:
            self._cb = EXCHANGE_FN(0)
        else:
            def tramp(_user, what, a, b, c, count):
                try:
                    return int(fn(what, a, b, c, count) or 0)
                except Exception:  # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = EXCHANGE_FN(tramp)
        self._chk(self._L.sicp_set_exchange(self._h, self._cb, None, int(rank), int(world), int(bool(gn_shard))))

    # -- the library's own RCCL communicator (no host callback) --
    @staticmethod
    def comm_unique_id():
        """128-byte ncclUniqueId (rank 0 creates it, every rank passes the same bytes to comm_init)."""
        buf = C.create_string_buffer(128)
        L = load()
        rc = L.sicp_comm_unique_id(buf)
        if rc != OK:
            raise BackendError(L.sicp_last_error().decode(), rc)
        return buf.raw

    def comm_init(self, unique_id, rank, world, gn_shard=False):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        self._cb = None
        self._chk(self._L.sicp_comm_init(self._h, C.c_char_p(bytes(unique_id)), int(rank), int(world), int(bool(gn_shard))))

    def comm_destroy(self):
        self._comm_key = self._comm_group = None          # (dist.attach's note of a parked communicator: there is none any more)
        self._chk(self._L.sicp_comm_destroy(self._h))

    def comm_activate(self, on=True, gn_shard=False):
        """Use (on) or park (off) the communicator the context already owns."""
        self._chk(self._L.sicp_comm_activate(self._h, int(bool(on)), int(bool(gn_shard))))

    def comm_info(self):
        """What exchange is in force: backend none / callback / rccl, ranks and rank (for rccl as RCCL counts them)."""
        out = (C.c_int * 6)()
        self._chk(self._L.sicp_comm_info(self._h, C.byref(out)))
        return {"backend": ("none", "callback", "rccl")[out[0]], "nranks": out[1], "rank": out[2],
                "partition": "queries" if out[3] == PART_QUERIES else "cloud", "gn_shard": bool(out[4]),
                "communicator": bool(out[5])}

    def device_memory(self):
        """(free, total) bytes of the context's device."""
        f, t = C.c_int64(), C.c_int64()
        self._chk(self._L.sicp_device_memory(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def set_partition(self, mode):
        """PART_CLOUD: ranks hold index ranges of the searched cloud; PART_QUERIES: ranks hold the whole cloud and
        match a slice of the queries each."""
        self._chk(self._L.sicp_set_partition(self._h, int(mode)))

    def lexmin_gathered(self, gathered):
        """gathered: (world, Q, 5) float64 records (d2, idx bits, x, y, z) -> (d2, idx, xyz)."""
        g = _f64(gathered)
        world, Q, five = g.shape
        assert five == 5
        d2, idx, xyz = np.empty(Q), np.empty(Q, np.int64), np.empty((Q, 3))
        self._chk(self._L.sicp_lexmin_gathered(self._h, _ptr(g), world, Q, _ptr(d2), _ptr(idx), _ptr(xyz)))
        return d2, idx, xyz

    # -- timing --
    def timing_enable(self, on=True, count_work=False):
        """Kernel timing with HIP events on the library's stream; count_work also makes the grid search tally the
        candidates and rows it touches (slower: a separate pass)."""
        self._chk(self._L.sicp_timing_enable(self._h, 2 if (on and count_work) else int(bool(on))))

    def timing_reset(self):
        self._chk(self._L.sicp_timing_reset(self._h))

    def last_match_kernel(self):
        k = C.c_int()
        self._chk(self._L.sicp_last_match_kernel(self._h, C.byref(k)))
        return MATCH_KERNELS[k.value]

    def match_work(self):
        """Work counters of the grid search since timing_reset (kept while timing is enabled)."""
        out = np.zeros(3, np.uint64)
        self._chk(self._L.sicp_match_work(self._h, _ptr(out)))
        d = C.c_uint64(0)
        self._chk(self._L.sicp_match_deferred(self._h, C.byref(d)))
        return {"candidates": int(out[0]), "rows": int(out[1]), "launches": int(out[2]), "deferred": int(d.value)}

    def tail_cycles(self):
        """k_icp_tail's own clock over its phases in the last iteration it ran (shader cycles)."""
        out = np.zeros(5)
        self._chk(self._L.sicp_tail_cycles(self._h, _ptr(out)))
        return dict(zip(("load", "select", "keep", "lm", "final"), (float(v) for v in out)))

    def tail_selection(self):
        """How the single-workgroup tail found median / MAD in its last iteration (histogram rounds; 0 = from the window around the
        previous iteration's value) and in how many iterations since icp_setup both came from their windows."""
        out = np.zeros(3, np.int64)
        self._chk(self._L.sicp_tail_selection(self._h, _ptr(out)))
        return {"median_rounds": int(out[0]), "mad_rounds": int(out[1]), "window_iterations": int(out[2])}

    def exchange_info(self):
        """What the chained iterations' exchanges did since icp_setup: form of the last one, how many ran."""
        out = np.zeros(4, np.int64)
        self._chk(self._L.sicp_exchange_info(self._h, _ptr(out)))
        return {"form": ("none", "records_allgather", "key_allreduces", "query_slices")[int(out[0])], "count": int(out[1]),
                "keys_min_q": int(out[2]), "callback_serves_u64": bool(out[3])}

    def knn_work(self):
        """Work counters of the one-sweep k-NN (normals) since timing_reset (kept while timing_enable(2) is in force)."""
        out = np.zeros(4, np.uint64)
        self._chk(self._L.sicp_knn_work(self._h, _ptr(out)))
        return {"candidates": int(out[0]), "sweeps": int(out[1]), "slow_queries": int(out[2]), "in_ball": int(out[3])}

    def timing(self):
        out = {}
        for k, name in KERNEL_NAMES.items():
            ms, n = C.c_double(), C.c_int64()
            self._chk(self._L.sicp_timing_get(self._h, k, C.byref(ms), C.byref(n)))
            out[name] = {"ms": ms.value, "launches": n.value}
        return out
