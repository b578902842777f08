"""``yaw_to_quat`` -- mirrors ``torchbox3d/math/linalg/lie/SO3.py:122-134``."""

from __future__ import annotations

import torch
from torch import Tensor

from .... import _lib as L
from ....engine import _require_cuda


def yaw_to_quat(yaw_rad: Tensor) -> Tensor:
    """(N,1) yaw -> (N,4) scalar-first quaternions [cos(y/2), 0, 0, sin(y/2)]."""
    _require_cuda(yaw_rad, "yaw_rad")
    n = yaw_rad.shape[0]
    y = yaw_rad[:, -1].float().contiguous()
    out = torch.empty((n, 4), dtype=torch.float32, device=y.device)
    L.call("rv_yaw_to_quat", L.ptr(y), L.i64(n), L.i64(1), L.ptr(out), L.stream_ptr())
    return out.to(yaw_rad.dtype)
