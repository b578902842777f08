"""Weighted multi-class NMS -- mirrors ``torchbox3d/math/ops/nms.py:64-266``.

``weighted_nms`` keeps the reference wrapper's signature and contract around the op-level FFI
``weighted_nms_ext.wnms_gpu`` (``nms.py:126-177``); here that FFI is ``rv_wnms`` of
``librv3d_hip.so``.  The kernel's arithmetic is not in the reference tree (third-party, un-pinned):
the semantics implemented are declared in ``DESIGN.md`` / ``oracle/nms.py`` -- parity unpinned.
"""

from __future__ import annotations

import ctypes
from typing import List, Tuple

import torch
from torch import Tensor

from ... import _lib as L
from ...engine import _require_cuda


def weighted_nms(boxes: Tensor, data2merge: Tensor, scores: Tensor, nms_threshold: float, merge_thresh: float) -> Tuple[Tensor, Tensor, Tensor]:
    """boxes (N,5) [x1,y1,x2,y2,ry], data2merge (N,C), scores (N,) -> (keep indices, merged rows (K,C+1), counts (K,))."""
    _require_cuda(boxes, "boxes")
    sorted_scores, order = scores.sort(0, descending=True)
    boxes = boxes[order].contiguous().float()
    data = torch.cat([data2merge[order].float(), sorted_scores[:, None].float()], 1).contiguous()
    n, d = data.shape
    output = torch.zeros_like(data)
    keep = torch.zeros(n, dtype=torch.long, device=boxes.device)
    count = torch.zeros(n, dtype=torch.long, device=boxes.device)
    ws = torch.empty(L.load().rv_wnms_workspace_bytes(L.i64(n)), dtype=torch.uint8, device=boxes.device)
    num_out = ctypes.c_int64(0)
    L.call("rv_wnms", L.ptr(boxes), L.ptr(data), L.i64(n), L.i32(d), L.f32(nms_threshold), L.f32(merge_thresh), L.ptr(output),
           L.ptr(keep), L.ptr(count), L.ptr(ws), ctypes.byref(num_out), L.stream_ptr())
    k = int(num_out.value)
    return order[keep[:k]].contiguous(), output[:k], count[:k]


# All classes of a sweep in ONE weighted-NMS launch (``rv_wnms_classes``) when the sweep has at most this many candidates
# (the pair masks take n^2 / 4 bytes); above it, or when a class could exceed ``num_pre_nms``, the per-class loop runs.
FUSED_CLASSES_MAX = 32768


def _weighted_multiclass_nms_fused(cuboids_i: Tensor, scores_i: Tensor, categories_i: Tensor, iou_threshold: float,
                                   num_post_nms: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Same rows as the per-class loop below (classes do not interact; the score order within a class is the same), without
    its ``unique`` / boolean-mask / per-class host syncs: sort by score, one class-aware launch, then per-class top-k and
    the loop's output order (classes ascending, merged score descending within a class) by two stable sorts."""
    dev = cuboids_i.device
    sorted_scores, order = scores_i.sort(0, descending=True)
    b = cuboids_i[order].float()
    cats = categories_i[order].to(torch.int32).contiguous()
    half = b[:, 3:5] / 2
    rect = torch.cat([b[:, :2] - half, b[:, :2] + half, b[:, 6:7]], dim=-1).contiguous()
    data = torch.cat([b[:, :6], b[:, 6:7].sin(), b[:, 6:7].cos(), sorted_scores[:, None].float()], dim=1).contiguous()
    n, d = data.shape
    output = torch.zeros_like(data)
    keep = torch.zeros(n, dtype=torch.long, device=dev)
    count = torch.zeros(n, dtype=torch.long, device=dev)
    ws = torch.empty(L.load().rv_wnms_workspace_bytes(L.i64(n)), dtype=torch.uint8, device=dev)
    num_out = ctypes.c_int64(0)
    L.call("rv_wnms_classes", L.ptr(rect), L.ptr(data), L.ptr(cats), L.i64(n), L.i32(d), L.f32(iou_threshold), L.f32(0.5), L.ptr(output),
           L.ptr(keep), L.ptr(count), L.ptr(ws), ctypes.byref(num_out), L.stream_ptr())
    k = int(num_out.value)
    merged, kc = output[:k], cats[keep[:k]]
    box6, sn, cs, sc = merged.split([6, 1, 1, 1], dim=1)
    boxes = torch.cat([box6, torch.atan2(sn, cs)], dim=1)
    sc = sc.flatten()
    o1 = torch.sort(sc, descending=True, stable=True).indices       # merged score descending ...
    o2 = torch.sort(kc[o1], stable=True).indices                    # ... then classes ascending, keeping that order
    sel = o1[o2]
    boxes, sc, kc = boxes[sel], sc[sel], kc[sel]
    if k > num_post_nms:  # only then can a class hold more than num_post_nms rows: drop ranks >= num_post_nms within a class
        idx = torch.arange(k, device=dev)
        start = torch.where(torch.cat([torch.ones(1, dtype=torch.bool, device=dev), kc[1:] != kc[:-1]]), idx, torch.zeros_like(idx))
        rank = idx - torch.cummax(start, 0).values
        m = rank < num_post_nms
        boxes, sc, kc = boxes[m], sc[m], kc[m]
    return boxes, sc, kc.to(sc.dtype)


def weighted_multiclass_nms(cuboids_i: Tensor, scores_i: Tensor, categories_i: Tensor, iou_threshold: float, num_pre_nms: int,
                            num_post_nms: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Per class (ascending ``unique``): top-k pre, weighted NMS with merge threshold 0.5 (``nms.py:105-106``), top-k post."""
    if 0 < scores_i.shape[0] <= min(FUSED_CLASSES_MAX, num_pre_nms):
        return _weighted_multiclass_nms_fused(cuboids_i, scores_i, categories_i, iou_threshold, num_post_nms)
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
        _, merged, _ = weighted_nms(rect, data, s, nms_threshold=iou_threshold, merge_thresh=0.5)
        box6, sn, cs, sc = merged.split([6, 1, 1, 1], dim=1)
        b = torch.cat([box6, torch.atan2(sn, cs)], dim=1)
        sc = sc.flatten()
        sc, rank = sc.topk(k=min(len(b), num_post_nms), dim=0)
        out_b.append(b[rank])
        out_s.append(sc)
        out_c.append(torch.full_like(sc, fill_value=float(j)))
    return torch.cat(out_b), torch.cat(out_s), torch.cat(out_c)


def batched_multiclass_nms(cuboids: Tensor, scores: Tensor, categories: Tensor, num_pre_nms: int, num_post_nms: int,
                           iou_threshold: float, min_confidence: float, nms_mode: str) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Per sweep: ``score >= min_confidence`` filter, then per-class NMS (``nms.py:181-266``)."""
    nms_mode = nms_mode.upper()
    if nms_mode != "WEIGHTED":
        raise NotImplementedError("NMS mode HARD (detectron2 nms_rotated) is not selected by the rv-* configs (conf/model/baseline.yaml:52)")
    bs, ss, cs, ids = [], [], [], []
    for i in range(cuboids.shape[0]):
        m = scores[i] >= min_confidence
        if not bool(m.any()):
            continue
        b, s, c = weighted_multiclass_nms(cuboids[i, m], scores[i, m], categories[i, m], iou_threshold, num_pre_nms, num_post_nms)
        bs.append(b)
        ss.append(s)
        cs.append(c)
        ids.append(torch.full_like(s, fill_value=float(i)))
    if not bs:
        return (cuboids.new_empty((0, cuboids.shape[-1])), scores.new_empty((0, 1)), categories.new_empty((0, 1)), categories.new_empty((0, 1)))
    return torch.cat(bs), torch.cat(ss), torch.cat(cs), torch.cat(ids)
