"""Oracle: the loader's augmentations on a range-view sweep + its annotations.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  numpy restatement of
``/root/reference/src/torchbox3d/prototype/loader.py``:

* ``:825-882``  random_rotation           -> :func:`rotate`      (roll by floor(theta / tau * W) columns, xyz and boxes by Rz(-theta))
* ``:885-915``  random_global_scale       -> :func:`scale`       (xyz, box centres and sizes times s; ``range`` := ||xyz||)
* ``:918-945``  random_global_translation -> :func:`translate`   (xyz and box centres plus t; ``range`` is NOT updated)
* ``:948-990``  flip_azimuth              -> :func:`flip`        (columns reversed, azimuth negated: y -> -y, yaw -> -yaw)

A sweep is a (C, H, W) array with named channels (the reference's table has H*W rows x named columns; every column is rolled /
flipped, then the ``x``/``y``/``z`` (and ``range``) columns are rewritten).  Annotations are (10, M): ``tx_m ty_m tz_m length_m
width_m height_m qw qx qy qz``.  The random draws are arguments (the reference draws them from ``random``).  Pinned by
``tests/golden/augment.npz`` (made by ``make_golden.py augment`` running the reference's own functions).
"""

from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np


def _idx(names: Sequence[str], *want: str):
    names = list(names)
    return [names.index(w) for w in want]


def _quat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Hamilton product of wxyz quaternion columns (4, M)."""
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def _canonical(q: np.ndarray) -> np.ndarray:
    """scipy's ``Rotation.as_quat()`` of a rotation built from a matrix / Euler angles is sign-ambiguous; q and -q are the same
    rotation.  Comparisons go through :func:`yaw_of`, never through raw quaternion components."""
    return q


def yaw_of(q_wxyz: np.ndarray) -> np.ndarray:
    w, x, y, z = q_wxyz
    return np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))


def rotate(sweep: np.ndarray, names: Sequence[str], ann: np.ndarray, theta: float) -> Tuple[np.ndarray, np.ndarray]:
    W = sweep.shape[-1]
    shift = math.floor(theta / math.tau * W)
    out = np.roll(sweep, shift=shift, axis=-1).astype(np.float64)
    ix, iy, _ = _idx(names, "x", "y", "z")
    c, s = math.cos(theta), math.sin(theta)
    x, y = out[ix].copy(), out[iy].copy()
    out[ix], out[iy] = c * x + s * y, -s * x + c * y  # rot.T @ cart with rot = Rz(theta)
    ann = ann.astype(np.float64).copy()
    if ann.shape[1] > 0:
        tx, ty = ann[0].copy(), ann[1].copy()
        ann[0], ann[1] = c * tx + s * ty, -s * tx + c * ty
        # mat = R_q @ rot.T = R_q @ Rz(-theta)  ->  q' = q * q_z(-theta)
        qz = np.stack([np.full(ann.shape[1], math.cos(-theta / 2)), np.zeros(ann.shape[1]), np.zeros(ann.shape[1]),
                       np.full(ann.shape[1], math.sin(-theta / 2))])
        ann[6:10] = _quat_mul(ann[6:10], qz)
    return out, ann


def scale(sweep: np.ndarray, names: Sequence[str], ann: np.ndarray, s: float) -> Tuple[np.ndarray, np.ndarray]:
    out = sweep.astype(np.float64).copy()
    ix, iy, iz, ir = _idx(names, "x", "y", "z", "range")
    for i in (ix, iy, iz):
        out[i] = s * out[i]
    out[ir] = np.sqrt(out[ix] ** 2 + out[iy] ** 2 + out[iz] ** 2)
    ann = ann.astype(np.float64).copy()
    ann[:6] = s * ann[:6]
    return out, ann


def translate(sweep: np.ndarray, names: Sequence[str], ann: np.ndarray, t: Sequence[float]) -> Tuple[np.ndarray, np.ndarray]:
    out = sweep.astype(np.float64).copy()
    for i, d in zip(_idx(names, "x", "y", "z"), t):
        out[i] = out[i] + d  # every pixel, empty ones (x = y = z = 0) included; ``range`` stays as it was
    ann = ann.astype(np.float64).copy()
    for i, d in enumerate(t):
        ann[i] = ann[i] + d
    return out, ann


def flip(sweep: np.ndarray, names: Sequence[str], ann: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    out = np.flip(sweep, axis=-1).astype(np.float64).copy()
    iy = _idx(names, "y")[0]
    out[iy] = -out[iy]  # cart -> sph, azimuth *= -1, sph -> cart
    ann = ann.astype(np.float64).copy()
    if ann.shape[1] > 0:
        ann[1] = -ann[1]
        yaw = -yaw_of(ann[6:10])  # the reference rebuilds the quaternion from the negated yaw alone (roll / pitch dropped)
        ann[6], ann[7], ann[8], ann[9] = np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)
    return out, ann
