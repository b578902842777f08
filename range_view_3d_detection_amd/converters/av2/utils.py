"""Raw sweep -> range image on device -- mirrors ``/root/reference/converters/av2/utils.py`` (the offline converter).

The reference runs this path once per sweep on the host (polars + numpy + numba) and stores the result as a feather table;
here the same steps are device kernels, so a sweep can be projected online, right in front of the detector:

* :func:`unmotion_compensate`  (``utils.py:231-295``)  ``rv_unmotion_compensate``
* :func:`correct_laser_numbers` (``utils.py:211-228``)  ``rv_correct_laser_numbers`` -- the id tables (``LASER_MAPPING``,
  ``ROW_MAPPING_64`` / ``_32``, the list of affected logs: ``datasets/argoverse/constants.py:231-627``) are the dataset's
  data and are passed in, not shipped;
* :func:`build_range_view`      (``utils.py:32-105``)   ``rv_se3_inverse_apply`` + ``rv_project_indices`` + ``rv_z_buffer``.

Dropped points (outside the pose track's time span) are not compacted away: they keep their place (the z-buffer's tie rule
depends on the point order) and get range 0, which the z-buffer skips.  No CPU fallback.
"""

from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from ... import _lib as L
from ...engine import _require_cuda
from ...math import range_view as rv


def unmotion_compensate(xyz: Tensor, offset_ns: Tensor, timestamp_ns: int, pose_timestamps_ns: Tensor, pose_q_wxyz: Tensor, pose_t: Tensor) -> Tuple[Tensor, Tensor]:
    """(N,3) ego-frame points + per-point ``offset_ns`` + the pose track -> (``xyz_p`` (N,3) fp64, ``kept`` (N,) bool)."""
    _require_cuda(xyz, "xyz")
    dev = xyz.device
    xyz = xyz.double().contiguous()
    off = offset_ns.to(dev, torch.int32).contiguous()
    pts = pose_timestamps_ns.to(dev, torch.int64).contiguous()
    pq, pt = pose_q_wxyz.to(dev).double().contiguous(), pose_t.to(dev).double().contiguous()
    hit = (pose_timestamps_ns.cpu() == int(timestamp_ns)).nonzero()
    if hit.numel() != 1:
        raise L.RvError("unmotion_compensate: the sweep timestamp must be one of the pose timestamps (utils.py:259-268)")
    n = xyz.shape[0]
    out = torch.empty_like(xyz)
    kept = torch.empty(n, dtype=torch.uint8, device=dev)
    L.call("rv_unmotion_compensate", L.ptr(xyz), L.ptr(off), L.i64(n), L.i64(int(timestamp_ns)), L.ptr(pts), L.ptr(pq), L.ptr(pt), L.i32(pts.shape[0]),
           L.i32(int(hit[0, 0])), L.ptr(out), L.ptr(kept), L.stream_ptr())
    return out, kept.bool()


def correct_laser_numbers(laser_numbers: Tensor, affected: bool, laser_mapping: Tensor, row_mapping: Tensor) -> Tensor:
    """Laser ids -> image rows; ``affected`` = ``log_id in LOG_IDS`` (decided by the caller, who owns the dataset tables)."""
    _require_cuda(laser_numbers, "laser_numbers")
    dev = laser_numbers.device
    laser = laser_numbers.to(torch.int32).contiguous()
    lm = laser_mapping.to(dev, torch.int32).contiguous()
    rm = row_mapping.to(dev, torch.int32).contiguous()
    out = torch.empty_like(laser)
    L.call("rv_correct_laser_numbers", L.ptr(laser), L.i64(laser.numel()), L.i32(1 if affected else 0), L.ptr(lm), L.ptr(rm), L.i32(rm.numel()), L.ptr(out),
           L.stream_ptr())
    return out


def build_range_view(xyz_p: Tensor, kept: Tensor, features: Tensor, laser_number: Tensor, offset_ns: Tensor, ext_q_wxyz: Tensor, ext_t: Tensor,
                     height: int, width: int) -> Tensor:
    """``features`` (N,6) = [x, y, z, intensity, laser_number, is_within_roi] (ego frame, as the converter stores them) ->
    (8, H, W) fp32 image [x, y, z, intensity, laser_number, is_within_roi, timedelta_ns, range] (``RANGE_VIEW_SCHEMA``)."""
    _require_cuda(xyz_p, "xyz_p")
    dev = xyz_p.device
    n = xyz_p.shape[0]
    xyz_p = xyz_p.double().contiguous()
    q, t = ext_q_wxyz.to(dev).double().contiguous(), ext_t.to(dev).double().contiguous()
    k8 = kept.to(dev, torch.uint8).contiguous()
    cart = torch.empty_like(xyz_p)
    L.call("rv_se3_inverse_apply", L.ptr(xyz_p), L.i64(n), L.ptr(q), L.ptr(t), L.ptr(k8), L.ptr(cart), L.stream_ptr())
    mapping = torch.arange(height, dtype=torch.int32, device=dev)  # utils.py:66: rows were already corrected
    rows, cols, radius = rv.range_view_indices(cart, laser_number.to(dev), mapping, height, width, "converter")
    feats = torch.cat([features.to(dev).double().T, offset_ns.to(dev).double()[None], radius[None]], dim=0).contiguous()
    image, _ = rv.z_buffer(rows, cols, radius, feats, height, width)
    return image
