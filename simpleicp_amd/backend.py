"""Process-wide default GPU context used by PointCloud / SimpleICP.

One process drives one GPU (``LOCAL_RANK`` picks it under torch.distributed.run); the context is
created lazily on first use and raises ``BackendError`` when no MI355X is visible -- there is no
host fallback.
"""
from __future__ import annotations

import os

from . import _lib

_ctx = None


def default_device() -> int:
    return int(os.environ.get("SIMPLEICP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def get_context() -> "_lib.Context":
    global _ctx
    if _ctx is None:
        _ctx = _lib.Context(default_device())
    return _ctx


def reset_context():
    global _ctx
    if _ctx is not None:
        _ctx.close()
    _ctx = None
