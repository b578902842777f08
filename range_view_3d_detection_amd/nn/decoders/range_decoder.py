"""``RangeDecoder`` -- mirrors ``torchbox3d/nn/decoders/range_decoder.py:19-156``.

Same dataclass fields, same ``decode(multiscale_outputs, post_processing_config, task_config,
use_nms=True, **kwargs) -> (params (N,10), scores (N,), categories (N,), batch_index (N,))``
contract.  sigmoid / class max / fp64 box decode / range-band sampling run as one HIP kernel
(``rv_decode_candidates``); NMS is ``math.ops.nms.batched_multiclass_nms``.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Sequence, Tuple, Union

import torch
from torch import Tensor

from ... import _lib as L
from ...engine import _require_cuda
from ...math.linalg.lie.SO3 import yaw_to_quat


def decode_candidates(logits: Tensor, regressands: Tensor, cart: Tensor, mask: Tensor, azimuth_invariant: bool,
                      lower: Sequence[float], upper: Sequence[float], rates: Sequence[int],
                      category_offset: int = 0) -> Tuple[Tensor, Tensor, Tensor]:
    """scores (B,K) f32, categories (B,K) i64, boxes (B,K,7) f32 in ``sample_by_range`` order (dense if no bands)."""
    _require_cuda(logits, "logits")
    import ctypes

    B, C, H, W = logits.shape
    dev = logits.device
    lg = logits.detach().float().contiguous()
    rg = regressands.detach().float().contiguous()
    ct = cart.detach().float().contiguous()
    mk = mask.detach().to(torch.uint8).contiguous()
    nb = len(rates)
    lo = (ctypes.c_float * max(nb, 1))(*[float(v) for v in lower])
    hi = (ctypes.c_float * max(nb, 1))(*[float(v) for v in upper])
    rt = (ctypes.c_int32 * max(nb, 1))(*[int(v) for v in rates])
    K = L.load().rv_decode_num_candidates(L.i32(H), L.i32(W), L.i32(nb), rt)
    scores = torch.empty((B, K), dtype=torch.float32, device=dev)
    cats = torch.empty((B, K), dtype=torch.int64, device=dev)
    boxes = torch.empty((B, K, 7), dtype=torch.float32, device=dev)
    L.call("rv_decode_candidates", L.ptr(lg), L.ptr(rg), L.ptr(ct), L.ptr(mk), L.i32(B), L.i32(C), L.i32(H), L.i32(W),
           L.i32(1 if azimuth_invariant else 0), L.i32(nb), lo, hi, rt, L.i64(category_offset), L.ptr(scores), L.ptr(cats),
           L.ptr(boxes), L.stream_ptr())
    return scores, cats, boxes


@dataclass
class RangeDecoder:
    enable_azimuth_invariant_targets: bool
    enable_sample_by_range: bool
    lower_bounds: Sequence[float]
    upper_bounds: Sequence[float]
    subsampling_rates: Sequence[int]

    def decode(self, multiscale_outputs: Dict[Union[int, str], Dict[str, Tensor]], post_processing_config: Dict[str, Any],
               task_config: Dict[int, Sequence[str]], use_nms: bool = True, **kwargs: Any) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        scores_l, cats_l, boxes_l = [], [], []
        for _, outputs in multiscale_outputs.items():
            cart, mask = outputs["cart"], outputs["mask"]
            task_offset = 0
            for task_id, task_group in task_config.items():
                task = outputs[task_id]
                bands = (self.lower_bounds, self.upper_bounds, self.subsampling_rates) if self.enable_sample_by_range else ((), (), ())
                s, c, b = decode_candidates(task["logits"], task["regressands"], cart, mask,
                                            self.enable_azimuth_invariant_targets, *bands, category_offset=task_offset)
                task_offset += len(task_group)
                scores_l.append(s)
                cats_l.append(c)
                boxes_l.append(b)
        scores, cats, params = torch.cat(scores_l, 1), torch.cat(cats_l, 1), torch.cat(boxes_l, 1)
        if use_nms:
            from ...math.ops.nms import batched_multiclass_nms

            params, scores, cats, batch_index = batched_multiclass_nms(
                params, scores, cats,
                num_pre_nms=post_processing_config["num_pre_nms"], num_post_nms=post_processing_config["num_post_nms"],
                iou_threshold=post_processing_config["nms_threshold"], min_confidence=post_processing_config["min_confidence"],
                nms_mode=post_processing_config["nms_mode"], n_classes=sum(len(g) for g in task_config.values()),
            )
        else:
            B, N, _ = params.shape
            batch_index = torch.arange(0, B, device=params.device).repeat_interleave(N)
            keep = scores.flatten() >= post_processing_config["min_confidence"]
            params, scores, cats, batch_index = params.flatten(0, 1)[keep], scores.flatten()[keep], cats.flatten()[keep], batch_index[keep]
        quats = yaw_to_quat(params[:, -1:])
        return torch.cat([params[:, :-1], quats], dim=-1), scores, cats, batch_index
