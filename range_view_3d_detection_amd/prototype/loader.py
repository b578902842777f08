"""The one piece of ``torchbox3d/prototype/loader.py`` that shapes the hot path's input: the W padding that makes
the range-image width divisible by 16 (``subsample_range_view``, ``loader.py:792-815``) -- on device."""

from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from .. import _lib as L
from ..engine import _require_cuda

_PAD = {("waymo", 1): 3, ("waymo", 4): 19, ("av2", 1): 4, ("av2", 4): 28}


def _pad(x: Tensor, mask, pad: int, circular: bool) -> Tensor:
    c, h, w = x.shape
    src = x.float().contiguous()
    m = mask.float().reshape(h, w).contiguous() if mask is not None else None
    out = torch.empty((c, h, w + 2 * pad), dtype=torch.float32, device=x.device)
    L.call("rv_pad_range_view", L.ptr(src), L.ptr(m), L.i32(c), L.i32(h), L.i32(w), L.i32(pad), L.i32(1 if circular else 0), L.ptr(out),
           L.stream_ptr())
    return out


def subsample_range_view(range_view: Tensor, mask: Tensor, cart: Tensor, dataset_name: str, x_stride: int, mode: str) -> Tuple[Tensor, Tensor, Tensor]:
    """``range_view *= mask``; pad W by [4,4] (AV2) / [3,3] (Waymo) (``constant`` or ``circular``); ``[:, :, ::x_stride]``."""
    _require_cuda(range_view, "range_view")
    pad = _PAD[(dataset_name, 4 if x_stride == 4 else 1)]
    circ = mode == "circular"
    rv = _pad(range_view, mask, pad, circ)[:, :, ::x_stride]
    m = _pad(mask.float(), None, pad, circ)[:, :, ::x_stride]
    c = _pad(cart, None, pad, circ)[:, :, ::x_stride]
    return rv, m, c
