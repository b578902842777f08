"""Oracle: from a RAW sweep (ego frame, motion-compensated) to the range image -- the converter's path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  numpy restatement of ``/root/reference/converters/av2/utils.py``:

* ``:231-295``  unmotion_compensate    -> :func:`unmotion_compensate`
* ``:211-228``  correct_laser_numbers  -> :func:`correct_laser_numbers`
* ``:32-105``   build_range_view       -> :func:`build_range_view` (ego -> sensor SE3, then ``oracle.project``)

Quirks kept on purpose (the converter produced the training data):
* points are kept iff ``min(pose ts) < ts < max(pose ts)`` (strict);
* the per-point rotation is scipy's ``Slerp`` (``R_i * exp(alpha * log(R_i^-1 R_{i+1}))``, i = searchsorted(left) - 1), the
  per-point translation is ``t_low * alpha + (1 - alpha) * t_high`` -- the interpolation weights are SWAPPED (``:275-276``);
* ``build_range_view`` stores the ORIGINAL ego-frame x, y, z as features; only the binning uses the un-compensated,
  sensor-frame points.
Pinned by ``tests/golden/raw_sweep.npz`` (``make_golden.py raw_sweep`` runs the reference's functions).
"""

from __future__ import annotations

from typing import Tuple

import numpy as np

from . import project as _project


def _qmul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    aw, ax, ay, az = np.moveaxis(a, -1, 0)
    bw, bx, by, bz = np.moveaxis(b, -1, 0)
    return np.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def quat_to_matrix(q: np.ndarray) -> np.ndarray:
    """(…,4) wxyz unit quaternions -> (…,3,3)."""
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = np.moveaxis(q, -1, 0)
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def slerp(pose_ts: np.ndarray, pose_q: np.ndarray, ts: np.ndarray) -> np.ndarray:
    """scipy ``Slerp(times, rotations)(ts)`` as wxyz quaternions."""
    ind = np.searchsorted(pose_ts, ts, side="left") - 1
    ind[ts == pose_ts[0]] = 0
    alpha = (ts - pose_ts[ind]) / (pose_ts[ind + 1] - pose_ts[ind])
    q0, q1 = pose_q[ind], pose_q[ind + 1]
    rel = _qmul(q0 * np.array([1.0, -1.0, -1.0, -1.0]), q1)
    rel = rel * np.where(rel[:, :1] < 0, -1.0, 1.0)  # rotation vector of the shorter arc
    vn = np.linalg.norm(rel[:, 1:], axis=1)
    angle = 2.0 * np.arctan2(vn, rel[:, 0])
    axis = rel[:, 1:] / np.where(vn > 0, vn, 1.0)[:, None]
    half = 0.5 * angle * alpha
    step = np.concatenate([np.cos(half)[:, None], axis * np.sin(half)[:, None]], axis=1)
    return _qmul(q0, step)


def unmotion_compensate(xyz: np.ndarray, offset_ns: np.ndarray, timestamp_ns: int, pose_ts: np.ndarray, pose_q_wxyz: np.ndarray,
                        pose_t: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (kept mask (N,), xyz_p of the kept points (K,3))."""
    ts = np.int64(timestamp_ns) + offset_ns.astype(np.int64)
    kept = (ts > pose_ts.min()) & (ts < pose_ts.max())
    ts, p = ts[kept], xyz[kept].astype(np.float64)
    idx = np.searchsorted(pose_ts, ts, side="left")
    ts_low, ts_high = pose_ts[idx - 1], pose_ts[idx]
    R_p = quat_to_matrix(slerp(pose_ts, pose_q_wxyz, ts))
    k = int(np.nonzero(pose_ts == timestamp_ns)[0][0])
    R_t, t_t = quat_to_matrix(pose_q_wxyz[k]), pose_t[k]
    alpha = ((ts - ts_low) / (ts_high - ts_low))[:, None]
    t_p = pose_t[idx - 1] * alpha + (1 - alpha) * pose_t[idx]  # (sic)
    R = np.einsum("bji,jk->bik", R_p, R_t)              # R_p^T R_target
    t = np.einsum("bji,bj->bi", R_p, t_t[None] - t_p)   # R_p^T (t_target - t_p)
    return kept, np.einsum("bij,bj->bi", R, p) + t


def correct_laser_numbers(laser: np.ndarray, affected: bool, laser_mapping: np.ndarray, row_mapping: np.ndarray) -> np.ndarray:
    laser = laser.astype(np.int64).copy()
    if affected:
        hi = laser >= 32
        laser[hi] = laser_mapping[laser[hi] - 32] + 32
        lo = laser < 32  # (evaluated AFTER the first assignment, as in the reference; values >= 32 stay >= 32)
        laser[lo] = laser_mapping[laser[lo]]
    return row_mapping[laser]


def build_range_view(xyz_p: np.ndarray, features: np.ndarray, laser: np.ndarray, offset_ns: np.ndarray, ext_q_wxyz: np.ndarray,
                     ext_t: np.ndarray, height: int, width: int) -> np.ndarray:
    """features (N,6) = [x, y, z, intensity, laser_number, is_within_roi] -> (8,H,W) fp32 image
    [x, y, z, intensity, laser_number, is_within_roi, timedelta_ns, range]."""
    R = quat_to_matrix(ext_q_wxyz)
    cart_lidar = (xyz_p - ext_t[None]) @ R  # sensor_SE3_egovehicle = inverse of (R, t): R^T (p - t), row-vector form
    sph = _project.cart_to_sph(cart_lidar)
    rows, cols, radius = _project.range_view_indices(sph, laser.astype(np.int64), np.arange(height), height, width, "converter")
    feats = np.concatenate([features.astype(np.float64), offset_ns.astype(np.float64)[:, None], radius[:, None]], axis=1).T
    image, _ = _project.z_buffer(rows, cols, radius, np.ascontiguousarray(feats), height, width)
    return image
