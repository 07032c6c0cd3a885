"""Small host-side helpers under the names of /root/reference/python/simpleicp/mathutils.py:10-93, for callers
that import the reference's ``mathutils`` module.  The Euler convention and the matrix layout live in ``rbp.py``
(one definition for the whole package); the device side applies the same H through contract (T).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import rbp


def euler_coord_to_homogeneous_coord(Xe: np.ndarray) -> np.ndarray:
    """(n,3) -> (n,4) with a trailing column of ones (mathutils.py:10-16)."""
    Xe = np.asarray(Xe)
    Xh = np.ones((Xe.shape[0], 4), dtype=np.result_type(Xe.dtype, np.float64))
    Xh[:, :3] = Xe
    return Xh


def homogeneous_coord_to_euler_coord(Xh: np.ndarray) -> np.ndarray:
    """(n,4) -> (n,3), every row divided by its fourth entry (mathutils.py:19-26)."""
    Xh = np.asarray(Xh)
    return Xh[:, :3] / Xh[:, 3:4]


def euler_angles_to_linearized_rotation_matrix(alpha1: float, alpha2: float, alpha3: float) -> np.ndarray:
    """First-order rotation I + [alpha]x (mathutils.py:29-36)."""
    dR = np.eye(3)
    dR[0, 1], dR[0, 2] = -alpha3, alpha2
    dR[1, 0], dR[1, 2] = alpha3, -alpha1
    dR[2, 0], dR[2, 1] = -alpha2, alpha1
    return dR


def euler_angles_to_rotation_matrix(alpha1: float, alpha2: float, alpha3: float) -> np.ndarray:
    """R = Rx(alpha1) Ry(alpha2) Rz(alpha3) (mathutils.py:39-68)."""
    return rbp.rotation_from_euler(float(alpha1), float(alpha2), float(alpha3))


def rotation_matrix_to_euler_angles(R: np.ndarray) -> Tuple[float, float, float]:
    """Inverse of the above away from the gimbal lock (mathutils.py:71-78)."""
    return rbp.euler_from_rotation(np.asarray(R))


def create_homogeneous_transformation_matrix(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """[[R, t], [0 0 0 1]] (mathutils.py:81-93)."""
    return rbp.homogeneous(np.asarray(R), t)
