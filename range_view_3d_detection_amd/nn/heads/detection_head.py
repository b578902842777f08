"""``DetectionHead`` -- mirrors ``torchbox3d/nn/heads/detection_head.py:42-449``.

Same constructor keywords (``conf/model/range_view.yaml:88-126``), same sub-module names
(``classification_head.{stride}.{task}`` / ``regression_head...`` => same state-dict keys) and the
same ``forward(input, data, return_loss) -> (multiscale_outputs, losses)`` contract, including
the side effect of writing the per-stride target dicts into ``data`` (``:197-198``).

What runs where: the towers are fused HIP tap-conv programs; target assignment
(``compute_targets`` :496-665) and the soft-target / varifocal / L1 loss (:202-449,
``math/ops/assignment.py:76-161``) are device kernels without host synchronisation (the
reference loops over sweeps, tasks and instances in Python with ``.unique()/.tolist()``).
"""

from __future__ import annotations

import ctypes
import importlib
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor, nn

from ... import _lib as L
from ...engine import _require_cuda
from .dense_head import DenseHead, forward_pair

COLS = ("tx_m", "ty_m", "tz_m", "length_m", "width_m", "height_m", "qw", "qx", "qy", "qz", "task_id", "offset", "batch_index")
FOCAL_PRIOR_PROB = 0.01


def _cfg_get(cfg: Any, key: str, default: Any = None) -> Any:
    if isinstance(cfg, Mapping):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def annotations_to_cuboids(annotations: Any) -> np.ndarray:
    """Annotation table -> (M,10) fp64 [x,y,z,l,w,h,yaw,task,offset,batch] (``utils/polars.py:9-22``).

    Accepts a polars frame (``select(COLS).to_numpy()``), a numpy array or a tensor of shape (M,13).
    """
    if isinstance(annotations, Tensor):
        arr = annotations.detach().cpu().numpy()
    elif hasattr(annotations, "select"):
        arr = annotations.select(list(COLS)).to_numpy()
    else:
        arr = np.asarray(annotations)
    arr = np.asarray(arr, dtype=np.float64).reshape(-1, 13)
    if arr.shape[0] == 0:
        return np.zeros((0, 10), dtype=np.float64)
    w, x, y, z = arr[:, 6], arr[:, 7], arr[:, 8], arr[:, 9]
    yaw = np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return np.concatenate([arr[:, :6], yaw[:, None], arr[:, 10:]], axis=1)


def compute_targets(x: Dict[str, Any], tasks_config: Mapping, fpn_strides: Sequence[int], targets_config: Any) -> Dict[int, Dict[int, Dict[str, Tensor]]]:
    """Dense targets per stride / task (``detection_head.py:496-665``), one stride-1 level and one task on device."""
    cart = x["cart"]
    _require_cuda(cart, "cart")
    strides, tasks = [int(s) for s in fpn_strides], list(tasks_config.keys())
    if strides != [1] or len(tasks) != 1:
        raise NotImplementedError("the HIP target kernels cover the one-stride (1) / one-task layout of the rv-* configs")
    if _cfg_get(targets_config, "fpn_assignment_method") is not None:
        raise NotImplementedError("fpn_assignment_method must be null (conf/model/range_view.yaml:124)")
    t_id = tasks[0]
    n_cls = len(tasks_config[t_id])
    az_inv = bool(_cfg_get(targets_config, "enable_azimuth_invariant_targets", True))
    B, _, H, W = cart.shape
    dev = cart.device
    cub = annotations_to_cuboids(x["annotations"])
    order = np.argsort(cub[:, -1], kind="stable") if cub.shape[0] else np.zeros(0, dtype=np.int64)
    cub = cub[order]
    counts_per = np.bincount(cub[:, -1].astype(np.int64), minlength=B) if cub.shape[0] else np.zeros(B, dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts_per)]).astype(np.int32)
    m = int(cub.shape[0])
    # ONE pinned staging buffer, uploaded without blocking the host: a pageable `.to(device)` waits until the stream has drained
    # (the whole forward pass, ~20 ms twice per step: the host lost its lead over the GPU right before the backward pass)
    stage = torch.empty(10 * max(m, 1) + (B + 1 + 1) // 2, dtype=torch.float64, pin_memory=True)
    stage[: 10 * m].view(m, 10).copy_(torch.from_numpy(np.ascontiguousarray(cub)))
    stage[10 * max(m, 1) :].view(torch.int32)[: B + 1].copy_(torch.from_numpy(offsets))
    stage_d = stage.to(dev, non_blocking=True)
    cub_d = stage_d[: 10 * m].view(m, 10)
    off_d = stage_d[10 * max(m, 1) :].view(torch.int32)[: B + 1]
    scratch = torch.empty((3, max(m, 1)), dtype=torch.int32, device=dev)
    cart32 = cart.detach().float().contiguous()
    labels = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    pan = torch.empty((B, 1, H, W), dtype=torch.int64, device=dev)
    reg = torch.empty((B, 8, H, W), dtype=torch.float32, device=dev)
    ppo = torch.empty((B, 1, H, W), dtype=torch.int64, device=dev)
    nobj = torch.empty(1, dtype=torch.int32, device=dev)
    L.call("rv_assign_targets", L.ptr(cub_d) if m else None, L.i32(m), L.ptr(off_d), L.ptr(cart32), L.i32(B), L.i32(H), L.i32(W),
           L.i32(n_cls), L.i32(1 if az_inv else 0), L.ptr(scratch[0]), L.ptr(scratch[1]), L.ptr(scratch[2]), L.ptr(labels),
           L.ptr(pan), L.ptr(reg), L.ptr(ppo), L.ptr(nobj), L.stream_ptr())
    return {1: {t_id: {"points_per_obj": ppo, "panoptics": pan, "classification_labels": labels, "regression_targets": reg,
                       "num_objects": nobj, "num_category": torch.ones((B, n_cls, 1, 1), device=dev)}}}


def _nhwc_f32(x: Tensor) -> Tuple[Tensor, int]:
    """(N,C,H,W) fp32 tensor -> (storage tensor to keep alive, channel stride) for the NHWC kernels (zero-copy for head outputs)."""
    n, c, h, w = x.shape
    if x.dtype == torch.float32 and x.stride(1) == 1 and x.stride(2) == w * x.stride(3) and x.stride(0) == h * w * x.stride(3):
        return x, x.stride(3)
    y = x.detach().float().permute(0, 2, 3, 1).contiguous()
    return y, c


class _DetectionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, regressands: Tensor, cart: Tensor, mask: Tensor, tg: Dict[str, Tensor], hp: Dict[str, Any]):
        dev = logits.device
        B, n_cls, H, W = logits.shape
        lg, ld_l = _nhwc_f32(logits)
        rg, ld_r = _nhwc_f32(regressands)
        cart32 = cart.detach().float().contiguous()
        mask8 = mask.detach().reshape(B, H, W).to(torch.uint8).contiguous()
        sums = torch.empty(24, dtype=torch.float64, device=dev)
        soft = torch.empty((B, n_cls, H, W), dtype=torch.float32, device=dev)
        fg = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        coding = (ctypes.c_float * 8)(*[float(v) for v in hp["coding_weights"]])
        args = (L.ptr(lg), L.i32(ld_l), L.ptr(rg), L.i32(ld_r), L.ptr(cart32), L.ptr(mask8), L.ptr(tg["classification_labels"]),
                L.ptr(tg["panoptics"]), L.ptr(tg["regression_targets"]), L.ptr(tg["points_per_obj"]), L.ptr(tg["num_objects"]),
                L.i32(B), L.i32(n_cls), L.i32(H), L.i32(W), coding, L.f32(hp["cls_weight"]), L.f32(hp["reg_weight"]),
                L.f32(hp["smoothing"]), L.f32(hp["sigma"]), L.f32(hp["alpha"]), L.f32(hp["gamma"]), L.i32(1 if hp["az_inv"] else 0))
        L.call("rv_detection_loss_forward", *args, L.ptr(sums), L.ptr(soft), L.ptr(fg), L.stream_ptr())
        ctx.args, ctx.keep = args, (lg, rg, cart32, mask8, tg, coding)
        ctx.sums, ctx.meta = sums, (B, n_cls, H, W, ld_l, ld_r, logits.dtype, regressands.dtype)
        ctx.mark_non_differentiable(sums, soft, fg)
        loss = sums[16].clone()  # (formed by the kernel: sums[0] / sums[13] + sum(sums[4:12]) / sums[12])
        return loss, sums, soft, fg

    @staticmethod
    def backward(ctx, g_loss, *_):
        B, n_cls, H, W, ld_l, ld_r, dt_l, dt_r = ctx.meta
        dev = ctx.sums.device
        # (padding channels beyond n_cls / 8 are never written and never read: the returned gradients are slices)
        d_l = torch.empty((B, H, W, ld_l), dtype=torch.float32, device=dev)
        d_r = torch.empty((B, H, W, ld_r), dtype=torch.float32, device=dev)
        ctx.sums[15:16].copy_(g_loss.reshape(1))  # the incoming gradient as the kernel's device-side factor: one 8-byte copy instead of two passes over the gradients
        L.call("rv_detection_loss_backward", *ctx.args, L.ptr(ctx.sums), L.f32(1.0), L.ptr(d_l), L.ptr(d_r), L.stream_ptr())
        return (d_l[..., :n_cls].permute(0, 3, 1, 2).to(dt_l), d_r[..., :8].permute(0, 3, 1, 2).to(dt_r), None, None, None, None)


def _instantiate(cfg: Any) -> Any:
    if cfg is None or isinstance(cfg, str):
        return None
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.pop("_recursive_", None)
    if target.startswith("torchbox3d."):
        target = "range_view_3d_detection_amd." + target[len("torchbox3d."):]
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(**cfg)


class DetectionHead(nn.Module):
    """Per-stride x per-task classification / regression towers + losses."""

    def __init__(self, fpn: Mapping, fpn_kernel_sizes: Mapping, targets_config: Any, num_classification_blocks: int,
                 num_regression_blocks: int, final_kernel_size: int, tasks_cfg: Mapping, task_in_channels: int,
                 classification_weight: float, regression_weight: float, coding_weights: Sequence[float],
                 classification_head_channels: int, regression_head_channels: int, classification_normalization_method: str,
                 additive_smoothing: float = 1.0, _cls_loss: Any = None, _regression_loss: Any = None, compile: bool = False) -> None:
        super().__init__()
        self.fpn, self.fpn_kernel_sizes, self.targets_config = fpn, fpn_kernel_sizes, targets_config
        self.num_classification_blocks, self.num_regression_blocks = num_classification_blocks, num_regression_blocks
        self.final_kernel_size, self.tasks_cfg, self.task_in_channels = final_kernel_size, tasks_cfg, task_in_channels
        self.classification_weight, self.regression_weight = classification_weight, regression_weight
        self.coding_weights = list(coding_weights)
        self.classification_head_channels, self.regression_head_channels = classification_head_channels, regression_head_channels
        self.classification_normalization_method = classification_normalization_method
        self.additive_smoothing, self.compile = additive_smoothing, compile
        self.classification_head = nn.ModuleDict({
            str(stride): nn.ModuleDict({
                str(k): DenseHead(num_channels, classification_head_channels, len(categories), kernel_size=fpn_kernel_sizes[stride],
                                  final_kernel_size=final_kernel_size, prior_prob=FOCAL_PRIOR_PROB, num_blocks=num_classification_blocks)
                for k, categories in tasks_cfg.items()})
            for stride, num_channels in fpn.items()})
        self.regression_head = nn.ModuleDict({
            str(stride): nn.ModuleDict({
                str(k): DenseHead(num_channels, regression_head_channels, 8, kernel_size=fpn_kernel_sizes[stride],
                                  final_kernel_size=final_kernel_size, num_blocks=num_regression_blocks)
                for k, _ in tasks_cfg.items()})
            for stride, num_channels in fpn.items()})
        self.cls_loss = _instantiate(_cls_loss)
        reg_target = str(dict(_regression_loss).get("_target_", "torch.nn.L1Loss")) if _regression_loss is not None and not isinstance(_regression_loss, str) else "torch.nn.L1Loss"
        if not reg_target.endswith("L1Loss"):
            raise NotImplementedError("the fused HIP loss implements the configured torch.nn.L1Loss regression loss")
        self.regression_loss = nn.L1Loss(reduction="none")

    def forward(self, input: Dict[int, Tensor], data: Dict[Any, Any], return_loss: bool = False):
        multiscale_outputs: Dict[int, Dict[Any, Any]] = {}
        for stride in self.fpn.keys():
            s = int(stride)
            feats = input[s]
            features = data["features"][:, :, ::1, ::s].clone()
            cart = data["cart"][:, :, ::1, ::s].clone()
            mask = data["mask"][:, :, ::1, ::s].clone()
            multiscale_outputs[s] = {"features": features, "cart": cart, "mask": mask}
            if _cfg_get(self.targets_config, "fpn_assignment_method") == "RANGE":
                raise NotImplementedError("RANGE fpn assignment is not selected by any shipped rv-* config")
            for task_id in self.tasks_cfg.keys():
                logits, regressands = forward_pair(self.classification_head[str(stride)][str(task_id)],
                                                   self.regression_head[str(stride)][str(task_id)], feats)
                multiscale_outputs[s][task_id] = {"logits": logits, "regressands": regressands}
        losses: Dict[str, Any] = {}
        if return_loss:
            targets = compute_targets(data, tasks_config=self.tasks_cfg, fpn_strides=list(self.fpn.keys()), targets_config=self.targets_config)
            for k, v in targets.items():
                data[k] = v
            losses = self.loss(multiscale_outputs, data)
        return multiscale_outputs, losses

    def loss(self, multiscale_outputs: Dict[int, Dict[Any, Any]], multiscale_data: Dict[Any, Any]) -> Dict[str, Any]:
        """``DetectionHead.loss`` + ``reduce_multiscale_loss`` (``detection_head.py:202-449``) for one stride / one task."""
        (stride,) = [int(s) for s in self.fpn.keys()]
        (task_id,) = list(self.tasks_cfg.keys())
        out = multiscale_outputs[stride]
        tg = multiscale_data[stride][task_id]
        tc = self.targets_config
        if str(_cfg_get(tc, "affinity_fn", "GAUSSIAN")).upper() != "GAUSSIAN" or _cfg_get(tc, "normalize_affinities", False):
            raise NotImplementedError("the HIP loss kernel implements the configured GAUSSIAN affinity without normalisation")
        k = _cfg_get(tc, "k", float("inf"))
        if k != float("inf"):
            raise NotImplementedError("top-k soft assignment with finite k is not configured by any shipped rv-* config")
        hp = {
            "coding_weights": self.coding_weights, "cls_weight": float(self.classification_weight), "reg_weight": float(self.regression_weight),
            "smoothing": float(self.additive_smoothing), "sigma": float(_cfg_get(tc, "sigma", 0.75)),
            "alpha": float(getattr(self.cls_loss, "alpha", 0.75)), "gamma": float(getattr(self.cls_loss, "gamma", 2.0)),
            "az_inv": bool(_cfg_get(tc, "enable_azimuth_invariant_targets", True)),
        }
        flat = {"classification_labels": tg["classification_labels"], "panoptics": tg["panoptics"], "regression_targets": tg["regression_targets"],
                "points_per_obj": tg["points_per_obj"], "num_objects": tg["num_objects"]}
        loss, sums, soft, fg = _DetectionLossFn.apply(out[task_id]["logits"], out[task_id]["regressands"], out["cart"], out["mask"], flat, hp)
        tg["targets"] = soft
        total_fg, total_obj = sums[13], sums[12]
        # (views of the scalars loss_finish_kernel formed on the device: no launches here)
        task = {
            "loss": loss, "classification_loss": sums[17], "foreground_loss": sums[18], "background_loss": sums[19],
            "regression_loss": sums[23], "coordinate_loss": sums[20], "dimension_loss": sums[21], "rotation_loss": sums[22],
            "total_fg": total_fg, "total_objects": total_obj,
        }
        losses: Dict[str, Any] = dict(task)
        for name, v in task.items():
            losses[f"{name}/s{stride}"] = v
        mask = out["mask"]
        bg = torch.logical_and(fg.logical_not(), mask)
        losses["aux"] = {stride: {task_id: {"targets": soft, "foreground": fg, "background": bg.float(), "mask": mask,
                                            "point_counts": tg["points_per_obj"]}}}
        return losses
