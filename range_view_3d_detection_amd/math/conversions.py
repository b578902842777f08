"""Spherical <-> Cartesian conversions -- mirrors ``torchbox3d/math/conversions.py:28-81``."""

from __future__ import annotations

import torch
from torch import Tensor

from .. import _lib as L
from ..engine import _require_cuda


def _convert(name: str, x: Tensor) -> Tensor:
    _require_cuda(x, "coordinates")
    flat = x.reshape(-1, 3).contiguous()
    if flat.dtype not in (torch.float32, torch.float64):
        flat = flat.float()
    out = torch.empty_like(flat)
    L.call(name, L.ptr(flat), L.i64(flat.shape[0]), L.i32(1 if flat.dtype == torch.float64 else 0), L.ptr(out), L.stream_ptr())
    return out.reshape(x.shape).to(x.dtype if x.dtype.is_floating_point else out.dtype)


def cartesian_to_spherical_coordinates(coordinates_cartesian_m: Tensor) -> Tensor:
    """(...,3) xyz -> (...,3) [azimuth, inclination, radius]."""
    return _convert("rv_cart_to_sph", coordinates_cartesian_m)


def spherical_to_cartesian_coordinates(coordinates_spherical: Tensor) -> Tensor:
    """(...,3) [azimuth, inclination, radius] -> (...,3) xyz."""
    return _convert("rv_sph_to_cart", coordinates_spherical)
