"""CorrPts -- operator-level mirror of /root/reference/python/simpleicp/corrpts.py:14-237.

``SimpleICP.run`` runs a whole iteration behind one ABI call.  A caller that drives the reference's classes itself,

    cp = CorrPts(pc_fix, pc_mov); cp.match()
    cp.reject_wrt_planarity(0.3); cp.reject_wrt_point_to_plane_distances()
    optim = SimpleICPOptimization(cp, ...); residuals = optim.estimate_parameters()      (simpleicp.py:190-227)

gets the same kernels one operator at a time (``sicp_corr_match`` / ``sicp_corr_reject_*`` /
``sicp_estimate_parameters``): the exact 1-NN search, contract (P) distances, the float32 planarity test, the
median / raw-MAD selection and the fused 6x6 reductions all run on the GPU; this class only keeps the reference's
bookkeeping (a DataFrame with one row per correspondence that is still alive) next to the device state.

The device state belongs to the process-wide context, so it follows the LAST ``match()`` (or ``SimpleICP.run``):
using an older CorrPts object afterwards raises ``CorrPtsException`` instead of silently mixing two sets.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import pandas as pd

from . import _lib, backend
from .pointcloud import _ALL, PointCloud

_COLUMNS = ("pc1_idx", "pc2_idx", "point_to_plane_distances")


class CorrPtsException(Exception):
    """Raised when a CorrPts object is used after its device state was replaced."""


def _take(pc: PointCloud, name: str, idx: np.ndarray) -> np.ndarray:
    """``pc.iloc[idx][name].to_numpy()`` without densifying a sparse attribute column (O(len(idx)) on the columns
    estimate_normals creates)."""
    arr = pc[name].array
    idx = np.asarray(idx, dtype=np.int64)
    if isinstance(arr, pd.arrays.SparseArray) and hasattr(arr.sp_index, "indices"):
        at = arr.sp_index.indices
        out = np.full(len(idx), arr.fill_value, dtype=arr.dtype.subtype)
        if len(at) and len(idx):
            pos = np.minimum(np.searchsorted(at, idx), len(at) - 1)
            hit = at[pos] == idx
            out[hit] = np.asarray(arr.sp_values)[pos[hit]]
        return out
    return pc[name].to_numpy()[idx]


class CorrPts:
    """Corresponding points between two overlapping point clouds (corrpts.py:14-28)."""

    def __init__(self, pc1: PointCloud, pc2: PointCloud) -> None:
        self._pc1 = pc1
        self._pc2 = pc2
        self._df = pd.DataFrame()
        self._Q = 0                      # correspondences the device holds (= rows right after match)
        self._pos = np.empty(0, np.int64)   # device position of every row still in _df

    # ---- views (corrpts.py:30-122) ---------------------------------------------------------
    def _of(self, which: int, name: str):
        col = _COLUMNS[which - 1]
        if col not in self._df:
            return None
        return _take(self._pc1 if which == 1 else self._pc2, name, self._df[col].to_numpy())

    pc1_x = property(lambda self: self._of(1, "x"))
    pc1_y = property(lambda self: self._of(1, "y"))
    pc1_z = property(lambda self: self._of(1, "z"))
    pc1_nx = property(lambda self: self._of(1, "nx"))
    pc1_ny = property(lambda self: self._of(1, "ny"))
    pc1_nz = property(lambda self: self._of(1, "nz"))
    pc2_x = property(lambda self: self._of(2, "x"))
    pc2_y = property(lambda self: self._of(2, "y"))
    pc2_z = property(lambda self: self._of(2, "z"))
    pc2_nx = property(lambda self: self._of(2, "nx"))
    pc2_ny = property(lambda self: self._of(2, "ny"))
    pc2_nz = property(lambda self: self._of(2, "nz"))

    @property
    def point_to_plane_distances(self) -> np.ndarray:
        return self._df["point_to_plane_distances"].to_numpy()

    @property
    def num_corr_pts(self) -> int:
        return len(self._df)

    # ---- device state ------------------------------------------------------------------------
    def _device(self):
        """The context, provided its correspondence state is still this object's."""
        ctx = backend.get_context()
        if getattr(ctx, "_corr_owner", None) is not self:
            raise CorrPtsException("the device holds the correspondences of a later match() / run(); call match() again")
        return ctx

    def _drop_dead_rows(self, ctx) -> None:
        """The rejection kernels cleared entries of the device's alive mask: drop those rows here as well
        (the reference's ``self._df = self._df.loc[keep][:]``, corrpts.py:156,163,188)."""
        _, _, alive, _ = ctx.icp_state(pc2_idx=False, dist=False, residual=False)
        keep = alive[self._pos]
        self._df = self._df.loc[keep][:]
        self._pos = self._pos[keep]

    def _per_correspondence(self, values: np.ndarray, dtype) -> np.ndarray:
        """Row values -> one entry per device correspondence (entries of dropped rows are never looked at)."""
        full = np.zeros((self._Q,) + values.shape[1:], dtype=dtype)
        full[self._pos] = values
        return full

    # ---- operators ---------------------------------------------------------------------------
    def match(self) -> None:
        """For every SELECTED point of pc1 its nearest neighbour among the SELECTED points of pc2, and the signed
        point-to-plane distance to it along pc1's normal (corrpts.py:124-137,195-211)."""
        pc1, pc2 = self._pc1, self._pc2
        sel1 = pc1.idx_selected
        sel2 = pc2._selection()
        if sel2 is not _ALL and len(sel2) == 0:
            raise ValueError("pc2 has no selected points to search in")
        if len(sel1) == 0:
            self._df = pd.DataFrame({"pc1_idx": sel1, "pc2_idx": sel1, "point_to_plane_distances": np.empty(0)})
            self._Q, self._pos = 0, np.empty(0, np.int64)
            return
        normals = np.column_stack([_take(pc1, c, sel1) for c in ("nx", "ny", "nz")]).astype(np.float32, copy=False)
        ctx = backend.get_context()
        ctx._corr_owner = None
        pc1._upload(ctx, _lib.FIX)
        pc2._upload(ctx, _lib.MOV, rows=None if sel2 is _ALL else sel2)
        ctx.icp_setup(sel1, normals, np.zeros(len(sel1), np.float32))       # (planarity is handed over when it is tested)
        idx, dist = ctx.corr_match()
        self._df = pd.DataFrame({"pc1_idx": sel1, "pc2_idx": idx if sel2 is _ALL else sel2[idx],
                                 "point_to_plane_distances": dist})
        self._Q = len(sel1)
        self._pos = np.arange(self._Q, dtype=np.int64)
        ctx._corr_owner = self

    def reject_wrt_planarity(self, min_planarity: float) -> None:
        """Keeps the correspondences whose point has planarity >= min_planarity in pc1 and in pc2 -- each cloud is
        only tested if it carries the column; NaN fails (corrpts.py:139-163)."""
        if not len(self._df.columns):
            raise KeyError("pc1_idx")                          # what the reference raises before match()
        cols = []
        for which, pc in ((1, self._pc1), (2, self._pc2)):
            cols.append(self._per_correspondence(self._of(which, "planarity"), np.float32) if "planarity" in pc else None)
        if (cols[0] is None and cols[1] is None) or self._Q == 0:
            return
        ctx = self._device()
        ctx.corr_reject_planarity(min_planarity, cols[0], cols[1])
        self._drop_dead_rows(ctx)

    def reject_wrt_point_to_plane_distances(self) -> None:
        """Keeps |d - median(d)| <= 3 * MAD(d) (raw MAD; corrpts.py:165-188)."""
        if self._Q == 0:
            _ = self._df["point_to_plane_distances"]           # KeyError before match(), like the reference
            return
        ctx = self._device()
        ctx.corr_reject_distances()
        self._drop_dead_rows(ctx)

    def reject_wrt_to_angle_between_normals(self) -> None:
        raise NotImplementedError                               # as in the reference (corrpts.py:190-193)

    # ---- I/O (corrpts.py:213-237) ---------------------------------------------------------------
    def write_xyz(self, file: Path):
        """`X1 Y1 Z1 X2 Y2 Z2 point_to_plane_distance` per correspondence, np.savetxt's number format."""
        from . import io
        X = np.column_stack((self.pc1_x, self.pc1_y, self.pc1_z, self.pc2_x, self.pc2_y, self.pc2_z,
                             self.point_to_plane_distances))
        io.write_xyz(file, X, decimals=-1, header="//X1 Y1 Z1 X2 Y2 Z2 point_to_plane_distance")
