"""SimpleICPOptimization -- operator-level mirror of /root/reference/python/simpleicp/optimization.py:18-170
(``Parameter`` / ``RigidBodyParameters`` of optimization.py:291-382 live in ``rbp.py`` and are re-exported here, so
``from simpleicp.optimization import RigidBodyParameters`` keeps working with the package name swapped).

The reference hands lmfit a residual callback over pandas gathers (optimization.py:93-101,172-288).  Here the
correspondences already sit in HBM (``CorrPts.match``): ``sicp_estimate_parameters`` minimises the same objective
    sum_i [w ((R(alpha) p2_i + t - p1_i) . n1_i)]^2 + sum_j [w_j (x_j - obs_j)]^2
by Levenberg-Marquardt on fused 6x6 normal-equation reductions (one launch per evaluation, 72 B per correspondence)
with the 6x6 solve on the host; the uncertainties come from the same reduction (optimization.py:126-170 without the
n x n weight matrix).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import _lib, corrpts
from .rbp import NAMES, Parameter, RigidBodyParameters

__all__ = ["SimpleICPOptimization", "Parameter", "RigidBodyParameters"]


class SimpleICPOptimization:
    def __init__(
        self,
        corr_pts: "corrpts.CorrPts",
        distance_weights: Optional[float],
        rbp_initial_values: Tuple[float],
        rbp_observed_values: Tuple[float],
        rbp_observation_weights: Tuple[float],
    ) -> None:
        """Same arguments as the reference (optimization.py:27-57): angles in rad; an infinite observation weight
        fixes the parameter, a positive finite one adds its observation.  ``distance_weights=None`` stands for
        1 / std(distances)^2, what SimpleICP.run substitutes before it gets here (simpleicp.py:233-234)."""
        self._cp = corr_pts
        self._rbp = RigidBodyParameters()
        self._rbp.set_parameter_attributes_from_list("initial_value", list(rbp_initial_values))
        self._rbp.set_parameter_attributes_from_list("observed_value", list(rbp_observed_values))
        self._rbp.set_parameter_attributes_from_list("observation_weight", list(rbp_observation_weights))
        self._distance_weights = distance_weights
        self._optim_results = None

    @property
    def rbp(self) -> RigidBodyParameters:
        return self._rbp

    def estimate_parameters(self) -> np.ndarray:
        """Estimates the varying parameters; returns the unweighted signed point-to-plane residuals of the
        correspondences, in their row order (optimization.py:65-124)."""
        cp = self._cp
        ctx = cp._device()
        x0 = np.array(self._rbp.get_parameter_attributes_as_list("initial_value"), dtype=float)
        obs = np.array(self._rbp.get_parameter_attributes_as_list("observed_value"), dtype=float)
        ow = np.array(self._rbp.get_parameter_attributes_as_list("observation_weight"), dtype=float)
        # the reference evaluates pc2's coordinates when it optimises: by then SimpleICP.run has undone the
        # transform the match was made under (simpleicp.py:202), so hand over the rows as they are NOW
        p2 = cp._per_correspondence(np.column_stack((cp.pc2_x, cp.pc2_y, cp.pc2_z)), np.float64)
        try:
            R = ctx.estimate_parameters(x0, obs, ow, distance_weight=self._distance_weights, pc2_xyz=p2)
        except _lib.BackendError as e:
            if e.code == _lib.ERR_TOO_FEW:
                raise ValueError(str(e)) from None      # (lmfit refuses fewer residuals than parameters, too)
            raise
        self._optim_results = R
        self._rbp.set_parameter_attributes_from_list("estimated_value", [float(v) for v in R.x[:]])
        _, _, _, residuals = ctx.icp_state(pc2_idx=False, dist=False, keep=False)
        return residuals[cp._pos]

    def estimate_parameter_uncertainties(self) -> None:
        """A-posteriori standard deviation of every varying parameter (optimization.py:126-170)."""
        if self._optim_results is None:
            raise AttributeError("estimate_parameters() has not run")       # the reference fails on None.params
        sigma = self._cp._device().icp_uncertainties()
        ow = self._rbp.get_parameter_attributes_as_list("observation_weight")
        for name, s, w in zip(NAMES, sigma, ow):
            if np.isfinite(w):
                getattr(self._rbp, name).estimated_uncertainty = float(s)
