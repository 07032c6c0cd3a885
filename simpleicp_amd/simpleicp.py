"""Module alias: the reference keeps its driver class in ``simpleicp/simpleicp.py`` (simpleicp.py:41,382), so
``from simpleicp.simpleicp import SimpleICP, SimpleICPException`` keeps working with the package name swapped.
The implementation is ``icp.py``."""
from .icp import SimpleICP, SimpleICPException

__all__ = ["SimpleICP", "SimpleICPException"]
