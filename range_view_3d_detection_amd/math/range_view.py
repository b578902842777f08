"""Range-image projection on device -- mirrors ``torchbox3d/math/range_view.py:14-44`` and
``torchbox3d/math/numpy/conversions.py`` / ``converters/av2/utils.py:108-208``.

``build_range_view`` takes the per-point arrays the reference's converter holds in numpy
(fp64 Cartesian points in the sensor frame, fp64 per-point features, laser ids) as device
tensors and returns the (C,H,W) fp32 range image; bin indices and pixel ownership are
bit-exact with the reference's sequential z-buffer (see ``csrc/project.hip``).
"""

from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from .. import _lib as L
from ..engine import _require_cuda

VARIANTS = {"converter": 0, "library": 1}


def range_view_indices(cart: Tensor, laser_numbers: Tensor, laser_mapping: Tensor, height: int, width: int,
                       variant: str = "converter") -> Tuple[Tensor, Tensor, Tensor]:
    """(N,3) f64 points -> (rows i32, cols i32, range f64)."""
    _require_cuda(cart, "cart")
    n = cart.shape[0]
    cart = cart.double().contiguous()
    laser = laser_numbers.to(torch.int32).contiguous()
    mapping = laser_mapping.to(torch.int32).contiguous()
    rows = torch.empty(n, dtype=torch.int32, device=cart.device)
    cols = torch.empty(n, dtype=torch.int32, device=cart.device)
    rng = torch.empty(n, dtype=torch.float64, device=cart.device)
    L.call("rv_project_indices", L.ptr(cart), L.ptr(laser), L.ptr(mapping), L.i64(n), L.i32(height), L.i32(width),
           L.i32(VARIANTS[variant]), L.ptr(rows), L.ptr(cols), L.ptr(rng), L.stream_ptr())
    return rows, cols, rng


def atan2_cr(y: Tensor, x: Tensor) -> Tensor:
    """Correctly rounded fp64 ``atan2`` (the azimuth the binning uses; ``rv_atan2_cr``)."""
    _require_cuda(y, "y")
    y, x = y.double().contiguous(), x.double().contiguous()
    out = torch.empty_like(y)
    L.call("rv_atan2_cr", L.ptr(y), L.ptr(x), L.i64(y.numel()), L.ptr(out), L.stream_ptr())
    return out


def hypot_libc(x: Tensor, y: Tensor) -> Tensor:
    """fp64 ``hypot`` with glibc's bits (what ``np.hypot`` returns in the reference's converter; ``rv_hypot_libc``)."""
    _require_cuda(x, "x")
    x, y = x.double().contiguous(), y.double().contiguous()
    out = torch.empty_like(x)
    L.call("rv_hypot_libc", L.ptr(x), L.ptr(y), L.i64(x.numel()), L.ptr(out), L.stream_ptr())
    return out


def z_buffer(rows: Tensor, cols: Tensor, distances: Tensor, features: Tensor, height: int, width: int,
             min_distance: float = 1.0) -> Tuple[Tensor, Tensor]:
    """features (C,N) f64 -> (image (C,H,W) f32, winner (H,W) i64); reference z-buffer semantics."""
    _require_cuda(features, "features")
    c, n = features.shape
    dev = features.device
    feats = features.double().contiguous()
    keys = torch.empty(height * width, dtype=torch.int64, device=dev)
    image = torch.empty((c, height, width), dtype=torch.float32, device=dev)
    winner = torch.empty((height, width), dtype=torch.int64, device=dev)
    # temporaries must stay referenced until the launch is enqueued (the caching allocator would hand
    # their memory to the next temporary otherwise)
    rows_i, cols_i, dist = rows.to(torch.int32).contiguous(), cols.to(torch.int32).contiguous(), distances.double().contiguous()
    L.call("rv_z_buffer", L.ptr(rows_i), L.ptr(cols_i), L.ptr(dist), L.ptr(feats), L.i64(n), L.i32(c), L.i32(height), L.i32(width),
           L.f64(min_distance), L.ptr(keys), L.ptr(image), L.ptr(winner), L.stream_ptr())
    return image, winner


def build_range_view(cart: Tensor, features: Tensor, laser_numbers: Tensor, laser_mapping: Tensor, height: int = 64,
                     width: int = 2048, variant: str = "converter") -> Tuple[Tensor, Tensor]:
    rows, cols, rng = range_view_indices(cart, laser_numbers, laser_mapping, height, width, variant)
    return z_buffer(rows, cols, rng, features, height, width)
