"""simpleicp_amd -- MI355X-native (gfx950) ICP inner loop behind simpleICP's Python API.

Drop-in for ``from simpleicp import SimpleICP, PointCloud, RigidBodyParameters``
(/root/reference/python/simpleicp/__init__.py:12-14); the hot path runs in hand-written HIP
kernels (simpleicp_amd/csrc) loaded through a C ABI (include/simpleicp_hip.h).

The reference's other modules keep their names for callers that drive the loop themselves:
``simpleicp_amd.corrpts.CorrPts``, ``simpleicp_amd.optimization.SimpleICPOptimization``,
``simpleicp_amd.mathutils``, ``simpleicp_amd.simpleicp`` (operator by operator on the same kernels).
"""
import logging as _logging

__version__ = "0.1.0"

_logging.getLogger(__name__).addHandler(_logging.NullHandler())

from .pointcloud import PointCloud, PointCloudException          # noqa: E402
from .rbp import Parameter, RigidBodyParameters                  # noqa: E402
from .icp import SimpleICP, SimpleICPException                   # noqa: E402

from . import io                                                 # noqa: E402,F401

__all__ = ["SimpleICP", "SimpleICPException", "PointCloud", "PointCloudException",
           "RigidBodyParameters", "Parameter"]
