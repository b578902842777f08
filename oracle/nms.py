"""Oracle: weighted multi-class NMS.  TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Wrapper logic follows the reference exactly (relative to ``/root/reference/src/torchbox3d``):

* ``math/ops/nms.py:181-266``  batched_multiclass_nms   -> :func:`batched_multiclass_nms`
* ``math/ops/nms.py:64-123``   weighted_multiclass_nms  -> :func:`weighted_multiclass_nms`
* ``math/ops/nms.py:126-177``  weighted_nms (wrapper around ``weighted_nms_ext.wnms_gpu``)

**Wrapper logic PINNED** (round 3): ``tests/golden/nms_wrapper.npz`` holds what the reference's own
``batched_multiclass_nms`` / ``weighted_multiclass_nms`` / ``weighted_nms`` / ``RangeDecoder.decode(use_nms=True)``
returned in the build container over a ``wnms_gpu`` stand-in running ``rvo_weighted_nms``
(``tests/golden/make_golden.py::gen_nms_wrapper``); ``tests/test_oracle_golden.py::test_nms_wrapper_matches_the_reference_wrapper``
requires this file's functions to equal those arrays bit for bit.

**PARITY UNPINNED for the inner kernel.**  ``weighted_nms_ext`` is TorchEx
(github.com/Abyssaledge/TorchEx, installed ad hoc per ``README.md:22``, no version or
commit pin, source and binary absent from ``/root/reference``), and the reference holds
no test or golden vector at that boundary.  What *is* visible in-tree is the contract:
inputs sorted by score descending; ``boxes`` (N,5) = [x1,y1,x2,y2,ry]; ``data2merge_score``
(N,D) whose last column is the score; caller-allocated zeroed ``output`` (N,D), ``keep``
(N,) i64, ``count`` (N,) i64; returns ``num_out``; post-conditions
``output[num_out:] == 0`` and ``count[:num_out] > 0`` (``nms.py:173-174``).

Semantics declared by this build (implemented in ``oracle/c/oracle.c::rvo_weighted_nms``
and, bit for bit, by the HIP kernels):

1. walk boxes in score order; every box not yet suppressed becomes an output row;
2. its cluster is itself plus every later box that was not suppressed before it was
   visited and whose rotated BEV IoU with it exceeds ``merge_thresh``;
3. the output row is the score-weighted mean of *all* D columns over the cluster
   (weights = last column, accumulated in ascending index order in fp32);
   ``count`` = cluster size;
4. every later box whose IoU with it exceeds ``nms_thresh`` is suppressed.

Rotated IoU: rectangle A clipped against the four half-planes of rectangle B
(Sutherland-Hodgman), shoelace area, ``inter / (|A| + |B| - inter)``, all fp32 without
fused multiply-add; sin/cos of the yaw are the fp32 roundings of the fp64 values.
"""

from __future__ import annotations

import ctypes
from typing import List, Tuple

import numpy as np
import torch
from torch import Tensor

from . import build as _build

_LIB = None


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(_build.build())
        _LIB.rvo_weighted_nms.restype = ctypes.c_int64
        _LIB.rvo_rotated_iou.restype = ctypes.c_float
    return _LIB


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def pairwise_iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().rvo_pairwise_iou(
        _ptr(a, ctypes.c_float), ctypes.c_int64(a.shape[0]), _ptr(b, ctypes.c_float), ctypes.c_int64(b.shape[0]),
        _ptr(out, ctypes.c_float),
    )
    return out


def weighted_nms(
    boxes: Tensor, data2merge: Tensor, scores: Tensor, nms_threshold: float, merge_thresh: float
) -> Tuple[Tensor, Tensor, Tensor]:
    """Sort by score (desc), append the score column, run the declared kernel semantics.

    Returns (keep indices into the *unsorted* input, merged rows (num_out, D+1), count).
    """
    sorted_scores, order = scores.sort(0, descending=True)
    b = np.ascontiguousarray(boxes[order].float().numpy())
    d = np.ascontiguousarray(torch.cat([data2merge[order].float(), sorted_scores[:, None].float()], 1).numpy())
    n, dim = d.shape
    out = np.zeros_like(d)
    keep = np.zeros(n, dtype=np.int64)
    count = np.zeros(n, dtype=np.int64)
    num_out = lib().rvo_weighted_nms(
        _ptr(b, ctypes.c_float), _ptr(d, ctypes.c_float), ctypes.c_int64(n), ctypes.c_int(dim),
        ctypes.c_float(nms_threshold), ctypes.c_float(merge_thresh),
        _ptr(out, ctypes.c_float), _ptr(keep, ctypes.c_int64), _ptr(count, ctypes.c_int64),
    )
    assert out[num_out:].sum() == 0 and (count[:num_out] > 0).all()  # nms.py:173-174
    return order[torch.from_numpy(keep[:num_out])], torch.from_numpy(out[:num_out]), torch.from_numpy(count[:num_out])


def weighted_multiclass_nms(
    cuboids_i: Tensor, scores_i: Tensor, categories_i: Tensor, iou_threshold: float, num_pre_nms: int, num_post_nms: int
) -> Tuple[Tensor, Tensor, Tensor]:
    """Per class (ascending ``unique``): top-k pre, weighted NMS (merge 0.5), top-k post."""
    out_b: List[Tensor] = []
    out_s: List[Tensor] = []
    out_c: List[Tensor] = []
    for j in categories_i.unique():
        sel = categories_i == j
        s, b = scores_i[sel], cuboids_i[sel]
        s, rank = s.topk(k=min(len(s), num_pre_nms), dim=0)
        b = b[rank]
        half = b[:, 3:5] / 2
        rect = torch.cat([b[:, :2] - half, b[:, :2] + half, b[:, 6:7]], dim=-1)
        data = torch.cat([b[:, :6], b[:, 6:7].sin(), b[:, 6:7].cos()], dim=1)
        _, merged, _ = weighted_nms(rect, data, s, iou_threshold, 0.5)  # merge_thresh hard-coded nms.py:106
        box6, sn, cs, sc = merged.split([6, 1, 1, 1], dim=1)
        b = torch.cat([box6, torch.atan2(sn, cs)], dim=1)
        sc = sc.flatten()
        sc, rank = sc.topk(k=min(len(b), num_post_nms), dim=0)
        out_b.append(b[rank])
        out_s.append(sc)
        out_c.append(torch.full_like(sc, fill_value=float(j)))
    return torch.cat(out_b), torch.cat(out_s), torch.cat(out_c)


def batched_multiclass_nms(
    cuboids: Tensor,
    scores: Tensor,
    categories: Tensor,
    num_pre_nms: int,
    num_post_nms: int,
    iou_threshold: float,
    min_confidence: float,
) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Per sweep: keep ``score >= min_confidence`` (``nms.py:212``), then per-class NMS."""
    bs, ss, cs, ids = [], [], [], []
    for i in range(cuboids.shape[0]):
        m = scores[i] >= min_confidence
        if int(m.sum()) == 0:
            continue
        b, s, c = weighted_multiclass_nms(cuboids[i, m], scores[i, m], categories[i, m], iou_threshold, num_pre_nms, num_post_nms)
        bs.append(b)
        ss.append(s)
        cs.append(c)
        ids.append(torch.full_like(s, fill_value=float(i)))
    if not bs:
        return (
            cuboids.new_empty((0, cuboids.shape[-1])),
            scores.new_empty((0, 1)),  # the reference's empty results are (0,1) (nms.py:249-251)
            categories.new_empty((0, 1)),
            categories.new_empty((0, 1)),
        )
    return torch.cat(bs), torch.cat(ss), torch.cat(cs), torch.cat(ids)
