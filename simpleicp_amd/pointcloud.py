"""PointCloud -- API mirror of /root/reference/python/simpleicp/pointcloud.py:15-226.

A ``pandas.DataFrame`` with columns ``x, y, z`` (float64), a boolean ``selected`` column and,
after ``estimate_normals``, float32 sparse columns ``nx, ny, nz, planarity`` (NaN where not
estimated) -- exactly the layout the reference's callers see.  The three operators that are
on the ICP hot path run on the GPU through the C ABI (no host fallback):

  select_in_range   -> exact 1-NN with a strict upper bound (pruned grid search; brute force on small clouds)  (pointcloud.py:149-171)
  estimate_normals  -> exact k-NN + covariance + 3x3 eigen (one sweep per query on the grid)              (pointcloud.py:173-203)
  transform_by_H    -> in-place rigid transform, contract (T)       (pointcloud.py:205-217)
"""
from __future__ import annotations

from pathlib import Path
from typing import List

import numpy as np
import pandas as pd

from . import _lib, backend

_PANDAS_MAJOR = int(pd.__version__.split(".")[0])
_XYZ = ["x", "y", "z"]
_ALL = object()          # selection sentinel: "every point" (no index vector materialised)
_ATTRS = ("nx", "ny", "nz", "planarity")

try:                                       # O(selected) construction of the sparse attribute columns
    from pandas._libs.sparse import IntIndex as _IntIndex
except ImportError:                        # pragma: no cover - falls back to the dense constructor
    _IntIndex = None


class PointCloudException(Exception):
    """Raised when the PointCloud class is misused (pointcloud.py:229)."""


class PointCloud(pd.DataFrame):
    def __init__(self, *args, **kwargs) -> None:
        kwargs.pop("remapping", None)     # accepted and ignored, like the reference (pointcloud.py:25)
        super().__init__(*args, **kwargs)
        for c in _XYZ:
            if c not in self:
                raise PointCloudException(f'Column "{c}" is missing in DataFrame.')
        self._num_points = len(self)
        if "selected" not in self:
            self["selected"] = np.ones(self._num_points, dtype=bool)

    # ---- views (pointcloud.py:51-110) ---------------------------------------------------
    def _col(self, name, only_selected=False):
        v = self[name].to_numpy()
        return v[self["selected"].to_numpy()] if only_selected else v

    @property
    def x(self) -> np.ndarray:
        return self._col("x")

    @property
    def y(self) -> np.ndarray:
        return self._col("y")

    @property
    def z(self) -> np.ndarray:
        return self._col("z")

    @property
    def x_selected(self) -> np.ndarray:
        return self._col("x", True)

    @property
    def y_selected(self) -> np.ndarray:
        return self._col("y", True)

    @property
    def z_selected(self) -> np.ndarray:
        return self._col("z", True)

    @property
    def X(self) -> np.ndarray:
        """(n,3) float64 copy of the coordinates."""
        return self[_XYZ].to_numpy()

    def _xyz_buffers(self):
        """The coordinates WITHOUT a host copy where the frame's storage allows it (10 M points = 240 MB):
        ("aos", (n,3) C-contiguous read-only view) for a frame still backed by the caller's (n,3) array,
        ("soa", (x, y, z) contiguous vectors) once columns have been assigned; else a gathered copy."""
        cols = [self[c].to_numpy() for c in _XYZ]
        n = len(cols[0])
        if n == 0 or any(c.dtype != np.float64 or c.ndim != 1 for c in cols):
            return "aos", self.X
        ptr = [c.__array_interface__["data"][0] for c in cols]
        if all(c.strides == (24,) for c in cols) and ptr[1] == ptr[0] + 8 and ptr[2] == ptr[0] + 16:
            return "aos", np.lib.stride_tricks.as_strided(cols[0], (n, 3), (24, 8), writeable=False)
        if all(c.strides == (8,) for c in cols):
            return "soa", cols
        return "aos", self.X

    def _upload(self, ctx, slot, lo=0, hi=None, index_base=0, rows=None, background=False):
        """Rows [lo, hi) (of the subset ``rows`` when given) into the library's slot; returns a row getter
        ``rows(idx) -> (len(idx),3)``.  ``background``: the whole cloud on the library's helper thread (`Context.upload_start`); the
        next call that names the slot waits for it."""
        kind, buf = self._xyz_buffers()
        if background and rows is None and lo == 0 and hi is None:
            if kind == "aos":
                ctx.upload_start(slot, xyz=buf, index_base=index_base)
                return lambda idx: buf[idx]
            ctx.upload_start(slot, columns=buf, index_base=index_base)
            return lambda idx: np.column_stack([b[idx] for b in buf])
        if rows is not None:
            part = rows[lo:hi]
            sub = buf[part] if kind == "aos" else np.column_stack([b[part] for b in buf])
            ctx.upload(slot, sub, index_base=index_base)
            return lambda idx: sub[idx]
        hi = self._num_points if hi is None else hi
        if kind == "aos":
            ctx.upload(slot, buf[lo:hi], index_base=index_base)
            return lambda idx: buf[idx]
        ctx.upload_columns(slot, buf[0][lo:hi], buf[1][lo:hi], buf[2][lo:hi], index_base=index_base)
        return lambda idx: np.column_stack([b[idx] for b in buf])

    def _attributes_of(self, idx):
        """(normals (len(idx),3) f32, planarity f32) of the rows ``idx`` (sorted) -- O(len(idx)) on the
        sparse columns estimate_normals creates, dense fallback for columns a caller assigned."""
        out = []
        for c in _ATTRS:
            arr = self[c].array
            if isinstance(arr, pd.arrays.SparseArray) and np.isnan(arr.fill_value) and hasattr(arr.sp_index, "indices"):
                pos_all = arr.sp_index.indices
                v = np.full(len(idx), np.nan, dtype=np.float32)
                if len(pos_all):
                    pos = np.minimum(np.searchsorted(pos_all, idx), len(pos_all) - 1)
                    hit = pos_all[pos] == idx
                    v[hit] = np.asarray(arr.sp_values, dtype=np.float32)[pos[hit]]
                out.append(v)
            else:
                out.append(np.asarray(self[c].to_numpy(), dtype=np.float32)[idx])
        return np.column_stack(out[:3]), out[3]

    def _planarity_pairs(self, rows=None):
        """(positions int64, values float32) of the non-NaN entries of the `planarity` column; positions count
        within ``rows`` (sorted row subset) when given.  O(stored values) on the sparse column estimate_normals
        creates."""
        arr = self["planarity"].array
        if isinstance(arr, pd.arrays.SparseArray) and np.isnan(arr.fill_value) and hasattr(arr.sp_index, "indices"):
            at = np.asarray(arr.sp_index.indices, dtype=np.int64)
            vals = np.asarray(arr.sp_values, dtype=np.float32)
        else:
            dense = np.asarray(self["planarity"].to_numpy(), dtype=np.float32)
            at = np.flatnonzero(~np.isnan(dense))
            vals = dense[at]
        ok = ~np.isnan(vals)
        at, vals = at[ok], vals[ok]
        if rows is None:
            return at, vals
        pos = np.searchsorted(rows, at)
        hit = (pos < len(rows)) & (rows[np.minimum(pos, len(rows) - 1)] == at)
        return pos[hit].astype(np.int64), vals[hit]

    @property
    def X_selected(self) -> np.ndarray:
        return self.loc[self["selected"], _XYZ].to_numpy()

    @property
    def idx_selected(self) -> np.ndarray:
        return np.flatnonzero(self["selected"].to_numpy())

    @idx_selected.setter
    def idx_selected(self, idx_selected: List[int]) -> None:
        self._set_idx_selected(idx_selected)

    def _set_idx_selected(self, idx_selected) -> None:
        """(Internal callers use this instead of ``self.idx_selected = ...``: pandas' __setattr__ evaluates the property's
        GETTER first -- a flatnonzero over all N points -- before it reaches the setter.)"""
        mask = np.zeros(self._num_points, dtype=bool)
        mask[np.asarray(idx_selected, dtype=np.int64)] = True
        self["selected"] = mask

    @property
    def num_points(self) -> int:
        return self._num_points

    @property
    def num_selected_points(self) -> int:
        return int(np.count_nonzero(self["selected"].to_numpy()))

    # ---- selection (pointcloud.py:112-171) ----------------------------------------------
    def select_all_points(self) -> None:
        self["selected"] = np.ones(self._num_points, dtype=bool)

    def unselect_all_points(self) -> None:
        self["selected"] = np.zeros(self._num_points, dtype=bool)

    def select_by_indices(self, indices: List[int]) -> None:
        """Keeps the currently selected points whose index is in ``indices``."""
        self._set_idx_selected(np.intersect1d(self.idx_selected, indices))

    def select_n_points(self, n: int, _cur=None):
        """Equidistant sub-sampling of the current selection (np.round = half-to-even;
        duplicates collapse, so fewer than n points may remain) -- pointcloud.py:132-147.
        (Internal callers pass the selection they already hold -- ``_ALL`` for "every point", which spares a pass over
        the mask and an index vector of N entries -- and get the new one back.)"""
        if _cur is _ALL:
            if self._num_points > n:
                cur = np.unique(np.round(np.linspace(0, self._num_points - 1, n)).astype(np.int64))
                self._set_idx_selected(cur)
                return cur
            return np.arange(self._num_points, dtype=np.int64)
        cur = self.idx_selected if _cur is None else _cur
        if len(cur) > n:
            pos = np.round(np.linspace(0, len(cur) - 1, n)).astype(int)
            cur = np.unique(cur[pos])
            self._set_idx_selected(cur)
        return cur if _cur is not None else None

    def _selection(self):
        """``_ALL`` when every point is selected (one vectorised count over the mask), else the selected indices."""
        mask = self["selected"].to_numpy()
        if int(np.count_nonzero(mask)) == self._num_points:
            return _ALL
        return np.flatnonzero(mask)

    def _keep_selected(self, cur, near):
        """Narrows the selection ``cur`` (``_ALL`` or sorted indices) to the entries flagged in ``near`` (one 0/1 byte
        per entry of ``cur``); writes the `selected` column and returns the new index vector."""
        near = np.asarray(near).view(bool) if np.asarray(near).dtype.itemsize == 1 else np.asarray(near, dtype=bool)
        if cur is _ALL:
            idx = np.flatnonzero(near)
            self["selected"] = near                      # the verdicts ARE the new mask
            return idx
        idx = cur[near]
        self._set_idx_selected(idx)
        return idx

    def select_in_range(self, X: np.ndarray, max_range: float, _ctx=None, _slot=None) -> None:
        """Keeps selected points whose nearest neighbour in X is closer than max_range
        (strict, like cKDTree's distance_upper_bound)."""
        if np.shape(X)[1] != 3:
            raise PointCloudException("X must have 3 columns!")
        ctx = _ctx or backend.get_context()
        if _slot is None:
            ctx.upload(_lib.MOV, np.asarray(X, dtype=np.float64))
        cur = self.idx_selected
        if len(cur) == 0:
            return
        self._upload(ctx, _lib.FIX)
        near = ctx.select_in_range(_lib.FIX, _lib.MOV, None if len(cur) == self._num_points else cur,
                                   max_range=float(max_range))
        self._set_idx_selected(cur[near])

    # ---- attributes (pointcloud.py:173-203) ---------------------------------------------
    def estimate_normals(self, neighbors: int, _ctx=None, _uploaded=False, _sel=None) -> None:
        """Normal vector + planarity of every SELECTED point from its `neighbors` nearest
        points among ALL points (itself included)."""
        ctx = _ctx or backend.get_context()
        if not _uploaded:
            self._upload(ctx, _lib.FIX)
        sel = self.idx_selected if _sel is None else _sel
        vals = {c: np.empty(0, dtype=np.float32) for c in _ATTRS}
        if len(sel):
            nv, pl = ctx.estimate_normals(_lib.FIX, sel, int(neighbors))
            vals = {"nx": nv[:, 0], "ny": nv[:, 1], "nz": nv[:, 2], "planarity": pl}
        for c, v in vals.items():
            self[c] = self._sparse_column(sel, v)

    def _sparse_column(self, idx, values):
        """float32 sparse column, NaN everywhere but ``values`` at the (sorted) rows ``idx`` -- what
        ``SparseArray(dense)`` gives (NaN results are not stored either), without touching n elements."""
        values = np.ascontiguousarray(values, dtype=np.float32)
        if _IntIndex is not None and self._num_points < 2**31:
            ok = ~np.isnan(values)
            return pd.arrays.SparseArray(values[ok], sparse_index=_IntIndex(self._num_points, idx[ok].astype(np.int32)),
                                         fill_value=np.nan, dtype=pd.SparseDtype(np.float32, np.nan))
        dense = np.full(self._num_points, np.nan, dtype=np.float32)
        dense[idx] = values
        return pd.arrays.SparseArray(dense)

    # ---- geometry (pointcloud.py:205-217) -----------------------------------------------
    def transform_by_H(self, H: np.ndarray, _ctx=None, _slot=None) -> None:
        """x,y,z <- (H @ [x y z 1]^T)[:3]  in place."""
        self._transform(H, _ctx, _slot)

    def _transform(self, H, ctx=None, slot=None) -> np.ndarray:
        """transform_by_H; hands back the (n,3) array of new coordinates it downloaded (= self.X)."""
        ctx = ctx or backend.get_context()
        if slot is None:
            slot = _lib.MOV
            self._upload(ctx, slot)
        ctx.transform(slot, np.asarray(H, dtype=np.float64))
        # run() returns the (n, 3) rows, the frame wants three contiguous columns: both come out of ONE pass over the link
        # (sicp_cloud_download_both: the device's columns through a pinned double buffer, fanned out -- and transposed -- by
        # host threads) and the columns are handed to the frame WITHOUT pandas' defensive copy of freshly made arrays
        # nobody else holds
        Xt, cols = ctx.download_both(slot)
        for name, col in zip(_XYZ, cols):
            self._adopt_column(name, col)
        return Xt

    def _adopt_column(self, name, values):
        """``self[name] = values`` for an array this object just created: same result, no copy.
        The copy-free road is pandas-internal (`DataFrame._set_item_mgr`, what `__setitem__` calls after sanitising) and
        is taken only on the pandas line it was written and tested against (2.x: 2.0 ... 2.3, tests/test_host_mirror.py);
        every other pandas -- and every column that is not a plain float64 vector of the frame's length -- goes through
        the public `__setitem__` and pays the copy."""
        fast = (_PANDAS_MAJOR == 2 and hasattr(self, "_set_item_mgr") and name in self.columns
                and isinstance(values, np.ndarray) and values.dtype == np.float64 and values.ndim == 1
                and len(values) == len(self.index))
        if fast:
            try:
                self._set_item_mgr(name, values)
                return
            except (TypeError, AttributeError):           # another signature after all: the public road
                pass
        self[name] = values

    # ---- I/O (pointcloud.py:219-226) ------------------------------------------------------
    def write_xyz(self, file: Path):
        """CloudCompare-style text file: header `//X Y Z`, 3 decimals."""
        from . import io
        io.write_xyz(file, self.X, decimals=3, header="//X Y Z")     # same bytes as pandas' to_csv(float_format="%.3f")
