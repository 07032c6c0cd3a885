"""PointCloud -- API mirror of /root/reference/python/simpleicp/pointcloud.py:15-226.

A ``pandas.DataFrame`` with columns ``x, y, z`` (float64), a boolean ``selected`` column and,
after ``estimate_normals``, float32 sparse columns ``nx, ny, nz, planarity`` (NaN where not
estimated) -- exactly the layout the reference's callers see.  The three operators that are
on the ICP hot path run on the GPU through the C ABI (no host fallback):

  select_in_range   -> brute-force 1-NN with a strict upper bound   (pointcloud.py:149-171)
  estimate_normals  -> brute-force k-NN + covariance + 3x3 eigen    (pointcloud.py:173-203)
  transform_by_H    -> in-place rigid transform, contract (T)       (pointcloud.py:205-217)
"""
from __future__ import annotations

from pathlib import Path
from typing import List

import numpy as np
import pandas as pd

from . import _lib, backend

_XYZ = ["x", "y", "z"]


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

    @property
    def X_selected(self) -> np.ndarray:
        return self.loc[self["selected"], _XYZ].to_numpy()

    @property
    def idx_selected(self) -> np.ndarray:
        return np.flatnonzero(self["selected"].to_numpy())

    @idx_selected.setter
    def idx_selected(self, idx_selected: List[int]) -> None:
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
        self.idx_selected = np.intersect1d(self.idx_selected, indices)

    def select_n_points(self, n: int) -> None:
        """Equidistant sub-sampling of the current selection (np.round = half-to-even;
        duplicates collapse, so fewer than n points may remain) -- pointcloud.py:132-147."""
        cur = self.idx_selected
        if len(cur) > n:
            pos = np.round(np.linspace(0, len(cur) - 1, n)).astype(int)
            self.idx_selected = cur[pos]

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
        idx, _ = ctx.knn(_lib.MOV, self.X_selected, k=1, max_dist=float(max_range))
        self.idx_selected = cur[idx[:, 0] >= 0]

    # ---- attributes (pointcloud.py:173-203) ---------------------------------------------
    def estimate_normals(self, neighbors: int, _ctx=None, _uploaded=False) -> None:
        """Normal vector + planarity of every SELECTED point from its `neighbors` nearest
        points among ALL points (itself included)."""
        ctx = _ctx or backend.get_context()
        if not _uploaded:
            ctx.upload(_lib.FIX, self.X)
        sel = self.idx_selected
        cols = {c: np.full(self._num_points, np.nan, dtype=np.float32) for c in ("nx", "ny", "nz", "planarity")}
        if len(sel):
            nv, pl = ctx.estimate_normals(_lib.FIX, sel, int(neighbors))
            cols["nx"][sel], cols["ny"][sel], cols["nz"][sel] = nv[:, 0], nv[:, 1], nv[:, 2]
            cols["planarity"][sel] = pl
        for c, v in cols.items():
            self[c] = pd.arrays.SparseArray(v)

    # ---- geometry (pointcloud.py:205-217) -----------------------------------------------
    def transform_by_H(self, H: np.ndarray, _ctx=None, _slot=None) -> None:
        """x,y,z <- (H @ [x y z 1]^T)[:3]  in place."""
        ctx = _ctx or backend.get_context()
        slot = _lib.MOV if _slot is None else _slot
        if _slot is None:
            ctx.upload(slot, self.X)
        ctx.transform(slot, np.asarray(H, dtype=np.float64))
        Xt = ctx.download(slot)
        self["x"], self["y"], self["z"] = Xt[:, 0], Xt[:, 1], Xt[:, 2]

    # ---- I/O (pointcloud.py:219-226) ------------------------------------------------------
    def write_xyz(self, file: Path):
        """CloudCompare-style text file: header `//X Y Z`, 3 decimals."""
        from . import io
        io.write_xyz(file, self.X, decimals=3, header="//X Y Z")     # same bytes as pandas' to_csv(float_format="%.3f")
