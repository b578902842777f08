"""Box decoding -- mirrors ``torchbox3d/math/ops/coding.py:79-144``."""

from __future__ import annotations

import torch
from torch import Tensor

from ... import _lib as L
from ...engine import _require_cuda


def decode_range_view(regressands: Tensor, cart: Tensor, enable_azimuth_invariant_targets: bool) -> Tensor:
    """(B,8,H,W) regressands + (B,3,H,W) points -> (B,7,H,W) [x,y,z,l,w,h,yaw].

    fp64 arithmetic on device, rounded once to the input dtype (``coding.py:126-128,144``).
    """
    _require_cuda(regressands, "regressands")
    dtype = regressands.dtype
    B, _, H, W = regressands.shape
    reg = regressands.detach().float().contiguous()
    c = cart.detach().float().contiguous()
    out = torch.empty((B, 7, H, W), dtype=torch.float32, device=reg.device)
    L.call("rv_decode_range_view", L.ptr(reg), L.ptr(c), L.i32(B), L.i32(H), L.i32(W),
           L.i32(1 if enable_azimuth_invariant_targets else 0), L.ptr(out), L.stream_ptr())
    return out.to(dtype)
