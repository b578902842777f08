"""Oracle: range-image projection (cart->sph, binning, z-buffer) and the loader's W pad.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  numpy fp64 exactly where the reference
uses it; the sequential z-buffer runs in C (``oracle/c/oracle.c::rvo_z_buffer``).
Reference files followed (relative to ``/root/reference/``):

* ``converters/av2/utils.py:156-183`` == ``src/torchbox3d/math/numpy/conversions.py:46-73``  cart_to_sph
* ``converters/av2/utils.py:108-153``  build_range_view_coordinates, *converter* variant
  (``col = W - round((az+pi)*W/tau)``) -- produced the training data, the one to match
* ``src/torchbox3d/math/numpy/conversions.py:9-43``  *library* variant
  (``col = round(W - (az+pi)*W/tau - 1)``)
* ``converters/av2/utils.py:186-208`` == ``numpy/conversions.py:106-128``  z_buffer
* ``src/torchbox3d/math/numpy/conversions.py:76-103``  sph_to_cart
* ``src/torchbox3d/prototype/loader.py:792-815``  subsample_range_view (W padding rule)
"""

from __future__ import annotations

import ctypes
import math
from typing import Tuple

import numpy as np

from . import nms as _nms  # shares the ctypes handle on liboracle.so


def cart_to_sph(cart: np.ndarray) -> np.ndarray:
    """(N,3) xyz -> (N,3) [azimuth, inclination, radius] (fp64 in, fp64 out)."""
    x, y, z = cart[..., 0], cart[..., 1], cart[..., 2]
    hyp = np.hypot(x, y)
    out = np.zeros_like(cart)
    out[..., 0] = np.arctan2(y, x)
    out[..., 1] = np.arctan2(z, hyp)
    out[..., 2] = np.hypot(hyp, z)
    return out


def sph_to_cart(sph: np.ndarray) -> np.ndarray:
    az, inc, r = sph[..., 0], sph[..., 1], sph[..., 2]
    rc = r * np.cos(inc)
    out = np.zeros_like(sph)
    out[..., 0] = rc * np.cos(az)
    out[..., 1] = rc * np.sin(az)
    out[..., 2] = r * np.sin(inc)
    return out


def range_view_indices(
    sph: np.ndarray,
    laser_numbers: np.ndarray,
    laser_mapping: np.ndarray,
    height: int,
    width: int,
    variant: str = "converter",
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Row/column bin of every point.  Returns (rows i64, cols i64, radius f64).

    ``np.round`` is round-half-to-even; clipping to [0, W-1] happens *before* the integer
    cast.  Unlike the reference this does not modify ``sph`` in place
    (``converters/av2/utils.py:133`` does ``azimuth += pi`` on the caller's array).
    """
    az = (sph[..., 0] + math.pi) * (width / math.tau)
    if variant == "converter":
        col = width - np.round(az)
    elif variant == "library":
        col = np.round(width - az - 1)
    else:
        raise ValueError(variant)
    col = np.clip(col, 0, width - 1)
    row = height - laser_mapping[laser_numbers] - 1
    return row.astype(np.int64), col.astype(np.int64), sph[..., 2].copy()


def z_buffer(
    rows: np.ndarray,
    cols: np.ndarray,
    distances: np.ndarray,
    features: np.ndarray,
    height: int,
    width: int,
    min_distance: float = 1.0,
) -> Tuple[np.ndarray, np.ndarray]:
    """Sequential z-buffer.  ``features`` is (C,N) fp64.  Returns (image (C,H,W) f32, winner (H,W) i64).

    Points closer than ``min_distance`` are skipped; point i replaces pixel p iff
    ``dist_i (f64) < buffer[p] (f32)``; on exact ties the earliest point stays.
    ``winner`` (index of the point that owns each pixel, -1 if empty) is an extra the
    reference does not return; it makes index parity checkable directly.
    """
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    dist = np.ascontiguousarray(distances, dtype=np.float64)
    feat = np.ascontiguousarray(features, dtype=np.float64)
    c, n = feat.shape
    image = np.zeros((c, height * width), dtype=np.float32)
    buf = np.full(height * width, np.inf, dtype=np.float32)
    winner = np.full(height * width, -1, dtype=np.int64)
    P = ctypes.POINTER
    _nms.lib().rvo_z_buffer(
        rows.ctypes.data_as(P(ctypes.c_int64)), cols.ctypes.data_as(P(ctypes.c_int64)),
        dist.ctypes.data_as(P(ctypes.c_double)), feat.ctypes.data_as(P(ctypes.c_double)),
        ctypes.c_int64(n), ctypes.c_int(c), ctypes.c_int(height), ctypes.c_int(width), ctypes.c_double(min_distance),
        image.ctypes.data_as(P(ctypes.c_float)), buf.ctypes.data_as(P(ctypes.c_float)),
        winner.ctypes.data_as(P(ctypes.c_int64)),
    )
    return image.reshape(c, height, width), winner.reshape(height, width)


def build_range_view(
    cart: np.ndarray,
    features: np.ndarray,
    laser_numbers: np.ndarray,
    laser_mapping: np.ndarray,
    height: int = 64,
    width: int = 2048,
    variant: str = "converter",
) -> Tuple[np.ndarray, np.ndarray]:
    """points (N,3) f64 + per-point features (C,N) -> range image (C,H,W) f32 + winner map."""
    sph = cart_to_sph(cart)
    rows, cols, radius = range_view_indices(sph, laser_numbers, laser_mapping, height, width, variant)
    return z_buffer(rows, cols, radius, features, height, width)


def pad_range_view(image: np.ndarray, dataset_name: str, mode: str = "constant") -> np.ndarray:
    """W padding that makes the width divisible by 16 (``loader.py:800-813``, x_stride 1).

    AV2: [4,4] (1800 -> 1808); Waymo: [3,3] (2650 -> 2656).  ``mode`` is "constant" (zeros)
    or "circular" (wrap-around in azimuth).
    """
    pad = {"av2": 4, "waymo": 3}[dataset_name]
    if mode == "constant":
        return np.pad(image, [(0, 0)] * (image.ndim - 1) + [(pad, pad)])
    if mode == "circular":
        return np.concatenate([image[..., -pad:], image, image[..., :pad]], axis=-1)
    raise ValueError(mode)
