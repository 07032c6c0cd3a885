"""Fast .xyz text I/O (host side, native + multithreaded; no GPU needed).

``read_xyz`` returns what ``np.genfromtxt(path)[:, :3]`` returns for the reference's data files
(/root/reference/python/simpleicp/tests/test_simpleicp.py:102-103) -- bit-identical values, two
orders of magnitude faster; ``write_xyz`` produces the bytes of ``PointCloud.write_xyz``
(pointcloud.py:219-226) / ``np.savetxt``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib


def _chk(L, rc):
    if rc != _lib.OK:
        raise OSError(L.sicp_last_error().decode())


def read_xyz(path, threads: int = 0) -> np.ndarray:
    """(n, 3) float64 array of the first three columns of every data row."""
    L = _lib.load()
    p = os.fsencode(path)
    n = C.c_int64()
    _chk(L, L.sicp_xyz_count(p, C.byref(n)))
    out = np.empty((n.value, 3), dtype=np.float64)
    got = C.c_int64()
    if n.value:
        _chk(L, L.sicp_xyz_read(p, out.ctypes.data_as(C.c_void_p), n.value, C.byref(got), int(threads)))
        assert got.value == n.value
    return out


def write_xyz(path, X, decimals: int = 3, header: str | None = "//X Y Z", threads: int = 0) -> None:
    """decimals >= 0 -> '%.<decimals>f', decimals < 0 -> '%.18e' (np.savetxt's default)."""
    L = _lib.load()
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("X must be 2-D")
    h = None if header is None else header.encode()
    _chk(L, L.sicp_xyz_write(os.fsencode(path), X.ctypes.data_as(C.c_void_p), X.shape[0], X.shape[1], int(decimals),
                             h, int(threads)))
