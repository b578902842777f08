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


# --------------------------------------------------------------------------------------
# correctly rounded fp64 atan2 (what the device binning computes, csrc/project.hip)
# --------------------------------------------------------------------------------------
# ``np.arctan2`` is NOT a function of its inputs alone: numpy dispatches to a SIMD kernel (SVML on AVX512 hosts) whose
# result is within 1 ulp but not always the correctly rounded one (66 of the 6000 fixture points differ from glibc's
# atan2 on the box the goldens were made on), i.e. the reference's azimuth depends on the host CPU in the last bit.
# The device therefore computes THE correctly rounded atan2 -- the one value every faithful libm approximates -- and
# this function is its independent witness: 80-bit ``atan2l`` where that already decides the rounding, exact
# 80-digit decimal arithmetic where it does not.  A last-bit difference in the azimuth moves a point to another
# column only when (az + pi) * W / tau lies within one ulp (~2e-13) of a half-integer; the golden fixture's 112 exact
# ties are points where numpy and the correctly rounded value agree, so the columns are bit-exact on the goldens.
def _atan_decimal(q):
    """atan(q) for a Decimal 0 <= q <= 1 at the current context precision."""
    from decimal import Decimal

    halvings = 0
    while q > Decimal("0.1"):
        q = q / (1 + (1 + q * q).sqrt())
        halvings += 1
    s, p, q2, k = Decimal(0), q, q * q, 0
    while True:
        term = p / (2 * k + 1)
        if abs(term) < Decimal(10) ** -85:
            break
        s += term if k % 2 == 0 else -term
        p *= q2
        k += 1
    return s * (2 ** halvings)


def _atan2_exact_rn(y: float, x: float) -> float:
    """Correctly rounded atan2 of two finite non-zero doubles by 90-digit decimal arithmetic (slow: hard cases only)."""
    from decimal import Decimal, getcontext
    from fractions import Fraction

    getcontext().prec = 90
    ay, ax = Decimal(abs(y)), Decimal(abs(x))
    pi = 16 * _atan_decimal(Decimal(1) / 5) - 4 * _atan_decimal(Decimal(1) / 239)
    t = _atan_decimal(ay / ax) if ay <= ax else pi / 2 - _atan_decimal(ax / ay)
    if x < 0:
        t = pi - t
    frac = Fraction(t)
    lo = float(frac)  # RN of the 90-digit value; check the two neighbours for a closer double
    best = min((lo, np.nextafter(lo, -np.inf), np.nextafter(lo, np.inf)), key=lambda c: abs(Fraction(float(c)) - frac))
    return math.copysign(float(best), y)


def atan2_cr(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """Correctly rounded (round-to-nearest-even) fp64 ``atan2(y, x)``, elementwise."""
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    ld = np.arctan2(y.astype(np.longdouble), x.astype(np.longdouble))  # 64-bit mantissa, < 1 ulp_ld error
    out = ld.astype(np.float64)
    # the rounding is decided unless ld sits within 4 ulp_ld of the midpoint between two doubles
    lo = np.minimum(out, np.nextafter(out, -np.inf)).astype(np.longdouble)
    dist = np.minimum(np.abs(ld - (out.astype(np.longdouble) + np.nextafter(out, np.inf).astype(np.longdouble)) / 2),
                      np.abs(ld - (out.astype(np.longdouble) + np.nextafter(out, -np.inf).astype(np.longdouble)) / 2))
    hard = np.isfinite(ld) & (x != 0) & (y != 0) & np.isfinite(x) & np.isfinite(y) & (dist <= 4 * np.spacing(np.abs(ld)))
    flat_out, fy, fx = out.reshape(-1), y.reshape(-1), x.reshape(-1)
    for i in np.nonzero(hard.reshape(-1))[0]:
        flat_out[i] = _atan2_exact_rn(float(fy[i]), float(fx[i]))
    return out
