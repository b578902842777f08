"""Weighted multi-class NMS -- mirrors ``torchbox3d/math/ops/nms.py:64-266``.

``weighted_nms`` keeps the reference wrapper's signature and contract around the op-level FFI
``weighted_nms_ext.wnms_gpu`` (``nms.py:126-177``); here that FFI is ``rv_wnms`` of
``librv3d_hip.so``.  The kernel's arithmetic is not in the reference tree (third-party, un-pinned):
the semantics implemented are declared in ``DESIGN.md`` / ``oracle/nms.py`` -- parity unpinned.
"""

from __future__ import annotations

import ctypes
import math
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


# Device-resident path (``rv_nms_sweeps``, csrc/nms2.hip): all sweeps of a batch in one set of launches, one device->host
# read at the end.  The per-candidate arrays are sized for every candidate of the sweep (up to ``FUSED_CLASSES_MAX``; the
# decoder emits 212 992 per 64 x 2048 sweep); the class-relative pair masks get a word budget (``MASK_WORDS`` per sweep and
# mask: 32 MB each), and a batch in which some sweep needs more -- tens of thousands of candidates in ONE class -- is redone
# over a buffer of the size the kernels report (second read; the ordering stages are not repeated).  The reference's
# per-class pre-NMS cut (``topk(num_pre_nms)``, nms.py:83-84) is applied on device.  Only a sweep with more candidates than
# ``FUSED_CLASSES_MAX`` (or more than 64 classes) takes the reference-shaped per-class loop over the FFI below.
FUSED_CLASSES_MAX = 262144
MASK_WORDS = 4 * 1024 * 1024


def nms_sweeps(cuboids: Tensor, scores: Tensor, categories: Tensor, n_classes: int, iou_threshold: float, min_confidence: float,
               num_post_nms: int, cap: int, num_pre_nms: int = 2**31 - 1) -> Tuple[Tensor, Tensor, Tensor, List[int]]:
    """(B,K,7), (B,K), (B,K) -> padded (B,R,7) boxes, (B,R) scores, (B,R) int32 classes and the per-sweep row counts
    (-2: the sweep had more than ``cap`` candidates -- its rows are not valid)."""
    _require_cuda(cuboids, "cuboids")
    dev = cuboids.device
    B, K, _ = cuboids.shape
    cub = cuboids.detach().float().contiguous()
    sc = scores.detach().float().contiguous()
    ct = categories.detach().to(torch.int64).contiguous()
    out_cap = min(cap, n_classes * num_post_nms)
    ob = torch.empty((B, out_cap, 7), dtype=torch.float32, device=dev)
    os_ = torch.empty((B, out_cap), dtype=torch.float32, device=dev)
    oc = torch.empty((B, out_cap), dtype=torch.int32, device=dev)
    counts = torch.empty((B, 4), dtype=torch.int64, device=dev)
    ws = torch.empty(L.load().rv_nms_sweeps_workspace_bytes(L.i32(B), L.i32(cap)), dtype=torch.uint8, device=dev)
    num_pre = int(min(num_pre_nms, 2**31 - 1))

    def run(mask_words: int, resume: int) -> List[List[int]]:
        masks = torch.empty((B, 2, mask_words), dtype=torch.int64, device=dev)
        L.call("rv_nms_sweeps", L.ptr(sc), L.ptr(ct), L.ptr(cub), L.i32(B), L.i64(K), L.i32(n_classes), L.f32(min_confidence), L.f32(iou_threshold),
               L.f32(0.5), L.i32(num_pre), L.i32(num_post_nms), L.i32(cap), L.i32(out_cap), L.ptr(ob), L.ptr(os_), L.ptr(oc), L.ptr(counts), L.ptr(ws),
               L.ptr(masks), L.i64(mask_words), L.i32(resume), L.stream_ptr())
        return counts.tolist()  # the one device->host read of the batch (a second one only when the mask budget was exceeded)

    budget = int(min(MASK_WORDS, max(1, (cap // 64 + 1) * cap)))  # (small inputs: no more than one class could need)
    rows = run(budget, 0)
    if any(r[0] == -1 for r in rows):
        rows = run(max(r[3] for r in rows), 1)
    return ob, os_, oc, [int(r[0]) for r in rows]


def _capacity(k: int) -> int:
    return max(64, (min(k, FUSED_CLASSES_MAX) + 63) // 64 * 64)


def weighted_multiclass_nms(cuboids_i: Tensor, scores_i: Tensor, categories_i: Tensor, iou_threshold: float, num_pre_nms: int,
                            num_post_nms: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Per class (ascending ``unique``): top-k pre, weighted NMS with merge threshold 0.5 (``nms.py:105-106``), top-k post."""
    n = scores_i.shape[0]
    if 0 < n <= FUSED_CLASSES_MAX:
        n_cls = int(categories_i.max().item()) + 1
        if n_cls <= 64:
            b, s, c, cnt = nms_sweeps(cuboids_i[None], scores_i[None], categories_i[None], n_cls, iou_threshold, -math.inf, num_post_nms, _capacity(n),
                                      num_pre_nms)
            return b[0, : cnt[0]], s[0, : cnt[0]], c[0, : cnt[0]].to(s.dtype)
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
                           iou_threshold: float, min_confidence: float, nms_mode: str, n_classes: int = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Per sweep: ``score >= min_confidence`` filter, then per-class NMS (``nms.py:181-266``)."""
    nms_mode = nms_mode.upper()
    if nms_mode != "WEIGHTED":
        raise NotImplementedError("NMS mode HARD (detectron2 nms_rotated) is not selected by the rv-* configs (conf/model/baseline.yaml:52)")
    bs, ss, cs, ids = [], [], [], []
    B, K = scores.shape
    fast = None
    cap = _capacity(K)
    if FUSED_CLASSES_MAX > 0 and n_classes is not None and n_classes <= 64:
        # device-resident path for the whole batch; sweeps that overflow its capacity fall through to the loop below
        fast = nms_sweeps(cuboids, scores, categories, n_classes, iou_threshold, min_confidence, num_post_nms, cap, num_pre_nms)
    for i in range(B):
        if fast is not None and fast[3][i] >= 0:
            k = fast[3][i]
            if k == 0:
                continue
            b, s, c = fast[0][i, :k], fast[1][i, :k], fast[2][i, :k].to(scores.dtype)
            bs.append(b)
            ss.append(s)
            cs.append(c)
            ids.append(torch.full_like(s, fill_value=float(i)))
            continue
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
