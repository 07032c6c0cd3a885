"""Rigid-body transformation parameters (host side).

API mirror of ``Parameter`` / ``RigidBodyParameters`` in
/root/reference/python/simpleicp/optimization.py:291-382: six parameters alpha1..3 (rad,
logged in degree) and tx, ty, tz, each with initial / observed / estimated value, observation
weight and a-posteriori uncertainty.  ``H`` follows the reference's Euler convention
(mathutils.py:39-68: R = Rx(alpha1) @ Ry(alpha2) @ Rz(alpha3)).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, fields
from typing import List

import numpy as np

NAMES = ("alpha1", "alpha2", "alpha3", "tx", "ty", "tz")
_RAD2DEG = 180 / np.pi


def rotation_from_euler(a1: float, a2: float, a3: float) -> np.ndarray:
    """mathutils.py:39-68."""
    c1, s1 = math.cos(a1), math.sin(a1)
    c2, s2 = math.cos(a2), math.sin(a2)
    c3, s3 = math.cos(a3), math.sin(a3)
    return np.array([
        [c2 * c3, -c2 * s3, s2],
        [c1 * s3 + s1 * s2 * c3, c1 * c3 - s1 * s2 * s3, -s1 * c2],
        [s1 * s3 - c1 * s2 * c3, s1 * c3 + c1 * s2 * s3, c1 * c2],
    ])


def euler_from_rotation(R: np.ndarray):
    """mathutils.py:71-78."""
    return math.atan2(-R[1, 2], R[2, 2]), math.asin(R[0, 2]), math.atan2(-R[0, 1], R[0, 0])


def homogeneous(R: np.ndarray, t) -> np.ndarray:
    """mathutils.py:81-93."""
    H = np.zeros((4, 4))
    H[:3, :3] = R
    H[:3, 3] = np.asarray(t, dtype=float)
    H[3, 3] = 1.0
    return H


def H_from_params(x) -> np.ndarray:
    return homogeneous(rotation_from_euler(float(x[0]), float(x[1]), float(x[2])), x[3:6])


@dataclass
class Parameter:
    """One optimisation parameter (optimization.py:291-320)."""

    initial_value: float = np.nan
    observed_value: float = np.nan
    observation_weight: float = np.nan
    estimated_value: float = np.nan
    estimated_uncertainty: float = np.nan
    scale_for_logging: float = 1

    def _scaled(self, v):
        return v * self.scale_for_logging

    @property
    def initial_value_scaled(self):
        return self._scaled(self.initial_value)

    @property
    def observed_value_scaled(self):
        return self._scaled(self.observed_value)

    @property
    def estimated_value_scaled(self):
        return self._scaled(self.estimated_value)

    @property
    def estimated_uncertainty_scaled(self):
        return self._scaled(self.estimated_uncertainty)


def _angle():
    return Parameter(scale_for_logging=_RAD2DEG)


@dataclass
class RigidBodyParameters:
    """The six rigid-body parameters (optimization.py:323-382)."""

    alpha1: Parameter = field(default_factory=_angle)
    alpha2: Parameter = field(default_factory=_angle)
    alpha3: Parameter = field(default_factory=_angle)
    tx: Parameter = field(default_factory=Parameter)
    ty: Parameter = field(default_factory=Parameter)
    tz: Parameter = field(default_factory=Parameter)

    @property
    def H(self) -> np.ndarray:
        """4x4 homogeneous matrix of the ESTIMATED values."""
        return H_from_params(self.get_parameter_attributes_as_list("estimated_value"))

    def set_parameter_attributes_from_list(self, attribute_name: str, array: List):
        if len(array) != len(NAMES):
            raise ValueError("expected six values")
        for name, value in zip(NAMES, array):
            setattr(getattr(self, name), attribute_name, value)

    def get_parameter_attributes_as_list(self, attribute_name: str) -> List:
        return [getattr(getattr(self, f.name), attribute_name) for f in fields(self)]
