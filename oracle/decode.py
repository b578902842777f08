"""Oracle: per-pixel box decoding, range-stratified sampling and the decoder driver.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Reference files followed (relative to
``/root/reference/src/torchbox3d``):

* ``math/ops/coding.py:79-107``   egovehicle_from_azimuth
* ``math/ops/coding.py:110-144``  decode_range_view          -> :func:`decode_range_view`
* ``nn/decoders/range_decoder.py:127-156`` sample_by_range    -> :func:`sample_by_range`
* ``nn/decoders/range_decoder.py:29-124``  RangeDecoder.decode -> :func:`range_decode`
* ``math/linalg/lie/SO3.py:122-134``       yaw_to_quat        -> :func:`yaw_to_quat`
"""

from __future__ import annotations

from typing import Dict, Mapping, Sequence, Tuple

import torch
from torch import Tensor

from . import nms as _nms


def decode_range_view(regressands: Tensor, cart: Tensor, azimuth_invariant: bool = True) -> Tensor:
    """(B,8,H,W) regressands + (B,3,H,W) points -> (B,7,H,W) [x,y,z,l,w,h,yaw].

    fp64 internally, cast back to the input dtype (``coding.py:126-128,144``).  With
    azimuth-invariant targets the xy offset is rotated by the pixel's azimuth and the
    yaw is shifted by it, *without* wrapping to (-pi, pi] (``coding.py:93-106``).
    """
    dtype = regressands.dtype
    r = regressands.double()
    p = cart.double()
    off_x, off_y, off_z = r[:, 0], r[:, 1], r[:, 2]
    lwh = r[:, 3:6].exp()
    yaw = torch.atan2(r[:, 6], r[:, 7])
    if azimuth_invariant:
        az = torch.atan2(p[:, 1], p[:, 0])
        s, c = az.sin(), az.cos()
        off_x, off_y = c * off_x - s * off_y, s * off_x + c * off_y
        yaw = yaw + az
    ctr = torch.stack([p[:, 0] + off_x, p[:, 1] + off_y, p[:, 2] + off_z], dim=1)
    return torch.cat([ctr, lwh, yaw[:, None]], dim=1).to(dtype)


def sample_by_range(
    scores: Tensor,
    categories: Tensor,
    cuboids: Tensor,
    cart: Tensor,
    lower_bounds: Sequence[float],
    upper_bounds: Sequence[float],
    subsampling_rates: Sequence[int],
) -> Tuple[Tensor, Tensor, Tensor]:
    """Range bands (lower, upper]; band i keeps columns ``::rate_i``.

    Only the *scores* are zeroed outside the band; categories and cuboids of the kept
    columns are passed through untouched (``range_decoder.py:146-152``).  Output order:
    band-major, then row-major over (H, kept columns).  Returns scores (B,K),
    categories (B,K), cuboids (B,K,7).
    """
    dist = cart.norm(dim=1, keepdim=True)
    s_list, c_list, b_list = [], [], []
    for lo, hi, rate in zip(lower_bounds, upper_bounds, subsampling_rates):
        band = torch.logical_and(dist > lo, dist <= hi)
        s_list.append((scores * band)[:, :, :, ::rate].flatten(2))
        c_list.append(categories[:, :, :, ::rate].flatten(2))
        b_list.append(cuboids[:, :, :, ::rate].flatten(2))
    s = torch.cat(s_list, dim=-1).squeeze(1)
    c = torch.cat(c_list, dim=-1).squeeze(1)
    b = torch.cat(b_list, dim=-1).transpose(2, 1)
    return s, c, b


def yaw_to_quat(yaw: Tensor) -> Tensor:
    """(N,1) yaw -> (N,4) wxyz = [cos(y/2), 0, 0, sin(y/2)] (roll = pitch = 0)."""
    half = yaw[:, -1:] * 0.5
    zero = torch.zeros_like(half)
    return torch.cat([half.cos(), zero, zero, half.sin()], dim=-1)


def dense_candidates(
    logits: Tensor,
    regressands: Tensor,
    cart: Tensor,
    mask: Tensor,
    lower_bounds: Sequence[float] = (0, 15, 30),
    upper_bounds: Sequence[float] = (15, 30, float("inf")),
    subsampling_rates: Sequence[int] = (8, 2, 1),
    azimuth_invariant: bool = True,
    enable_sample_by_range: bool = True,
    category_offset: int = 0,
) -> Tuple[Tensor, Tensor, Tensor]:
    """One (stride, task) leg of ``RangeDecoder.decode`` (``range_decoder.py:46-76``).

    score = max_c sigmoid(logit_c) * mask (ties -> lowest class index, CPU semantics).
    """
    scores, cats = (logits.sigmoid() * mask).max(dim=1, keepdim=True)
    boxes = decode_range_view(regressands, cart, azimuth_invariant)
    if enable_sample_by_range:
        s, c, b = sample_by_range(scores, cats, boxes, cart, lower_bounds, upper_bounds, subsampling_rates)
    else:
        s = scores.flatten(2).squeeze(1)
        c = cats.flatten(2).squeeze(1)
        b = boxes.flatten(2).transpose(2, 1)
    return s, c + category_offset, b


def range_decode(
    logits: Tensor,
    regressands: Tensor,
    cart: Tensor,
    mask: Tensor,
    post: Mapping[str, float],
    use_nms: bool = True,
    **kw,
) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """``RangeDecoder.decode`` for the one-stride / one-task layout of the rv-* configs.

    Returns (params (N,10) = [x,y,z,l,w,h,qw,qx,qy,qz], scores (N,), categories (N,),
    batch_index (N,)).  With NMS the reference returns categories and batch_index as
    *float* tensors (``nms.py:113,242``); without NMS they stay integer.
    """
    scores, cats, boxes = dense_candidates(logits, regressands, cart, mask, **kw)
    if use_nms:
        boxes, scores, cats, bidx = _nms.batched_multiclass_nms(
            boxes,
            scores,
            cats,
            num_pre_nms=int(post["num_pre_nms"]),
            num_post_nms=int(post["num_post_nms"]),
            iou_threshold=float(post["nms_threshold"]),
            min_confidence=float(post["min_confidence"]),
        )
    else:
        B, N, _ = boxes.shape
        bidx = torch.arange(B).repeat_interleave(N)
        keep = scores.flatten() >= post["min_confidence"]
        boxes, scores, cats, bidx = boxes.flatten(0, 1)[keep], scores.flatten()[keep], cats.flatten()[keep], bidx[keep]
    params = torch.cat([boxes[:, :-1], yaw_to_quat(boxes[:, -1:])], dim=-1)
    return params, scores, cats, bidx
