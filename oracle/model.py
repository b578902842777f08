"""Oracle: conv stack of the range-view detector (stem, backbone, dense heads) on CPU.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Functional PyTorch-CPU restatement,
NCHW fp32, driven by a ``state_dict`` whose keys are the reference's module paths, so
the very same weights can be fed to the reference (for the golden fixtures), to this
oracle and to the HIP engine.

Reference files followed (relative to ``/root/reference/src/torchbox3d``):

* ``nn/modules/conv.py:25-80``     Conv2dSame  -> :func:`conv2d_same`
* ``nn/blocks/__init__.py:13-81``  BasicBlock  -> :func:`basic_block`
* ``nn/blocks/__init__.py:84-126`` ResidualBlock -> :func:`residual_block`
* ``nn/blocks/__init__.py:129-182`` AggregationBlock -> :func:`aggregation_block`
* ``nn/stems/__init__.py:12-85``   MetaKernel  -> :func:`meta_kernel`
* ``nn/backbones/dla.py:15-131``   RangeBackbone -> :func:`range_backbone`
* ``nn/backbones/dla.py:134-208``  RangeNet    -> :func:`range_net`
* ``nn/heads/dense_head.py:13-76`` DenseHead   -> :func:`dense_head`

``Numerics`` lets the tests emulate the *storage* rounding points of the bf16 HIP path
(operands of every conv rounded to bf16, raw conv outputs stored as bf16, BatchNorm
statistics taken from the fp32 accumulators) so the GPU result can be compared at a
tight tolerance; with the default (identity) policy the functions reproduce the
reference's fp32 CPU path.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

StateDict = Dict[str, Tensor]
BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, used everywhere in the reference
BN_MOMENTUM = 0.1


def _identity(x: Tensor) -> Tensor:
    return x


def round_bf16(x: Tensor) -> Tensor:
    """Round-to-nearest-even to bf16 and back to fp32 (what the HIP path stores)."""
    return x.to(torch.bfloat16).to(torch.float32)


@dataclass
class Numerics:
    """Rounding policy + BatchNorm mode shared by all oracle functions."""

    train: bool = True  # BatchNorm uses batch statistics (and updates running stats)
    operand: Callable[[Tensor], Tensor] = _identity  # applied to conv inputs and weights
    store: Callable[[Tensor], Tensor] = _identity  # applied to every tensor written to HBM
    running: Optional[StateDict] = None  # receives updated running_mean/var when train
    trace: Optional[List[dict]] = None  # teacher-forcing record: one dict per conv (+ BatchNorm) unit, see ``_trace``
    # Summation order of the convolutions' channel reductions: 0 = as stored; k > 0 = every conv sums its input channels in a fixed
    # permuted order (1: reversed, k >= 2: a permutation seeded by k).  Mathematically the same network -- a different, equally valid
    # fp32 rounding of every accumulation, hence a different REALISATION of the bf16 storage points.  The envelope over a few of them
    # (tests/tools/emulation_yardstick.py envelope) is the yardstick the chaotic-regime gradient test is held to.
    sum_order: int = 0

    @staticmethod
    def bf16(train: bool = True, sum_order: int = 0) -> "Numerics":
        return Numerics(train=train, operand=round_bf16, store=round_bf16, running={}, sum_order=sum_order)


def round_fp16(x: Tensor) -> Tensor:
    """Round-to-nearest-even to fp16 and back (the storage points of the reference's evaluation under
    ``torch.autocast(dtype=torch.float16)``, ``nn/arch/detector.py:329-340``, and of the fp16-operand HIP build)."""
    return x.to(torch.float16).to(torch.float32)


def _numerics_fp16(train: bool = False) -> "Numerics":
    return Numerics(train=train, operand=round_fp16, store=round_fp16, running={})


Numerics.fp16 = staticmethod(_numerics_fp16)
FP32 = Numerics()


def _channel_order(c: int, sum_order: int) -> Optional[Tensor]:
    """Input-channel visiting order of a conv under ``Numerics.sum_order`` (None: as stored)."""
    if sum_order == 0 or c < 2:
        return None
    if sum_order == 1:
        return torch.arange(c - 1, -1, -1)
    return torch.randperm(c, generator=torch.Generator().manual_seed(1000 * sum_order + c))


def _trace(nm: Numerics, y: Tensor, **rec) -> None:
    """Per-layer record for the teacher-forced parity tests: the unit's input ``x`` (as the layer receives it, before the
    operand rounding), its weight key, geometry, BatchNorm prefix and raw fp32 output ``y``; ``dy`` (the gradient w.r.t. the
    raw output) is filled in by a hook when the run is differentiated."""
    if nm.trace is None:
        return
    rec["y"] = y.detach()
    rec["x"] = rec["x"].detach()
    if y.requires_grad:
        y.register_hook(lambda g, d=rec: d.__setitem__("dy", g.detach()))
    nm.trace.append(rec)


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def conv2d_same(
    x: Tensor,
    weight: Tensor,
    stride: Tuple[int, int] = (1, 1),
    bias: Optional[Tensor] = None,
    nm: Numerics = FP32,
) -> Tensor:
    """Zero "same" padding applied first, then a VALID strided conv.

    Follows ``nn/modules/conv.py:63-80``: per spatial dim total pad = d*(k-1) (d=1 here),
    left = total // 2, right = total - left; the stride acts on the padded tensor.
    """
    kh, kw = weight.shape[-2:]
    th, tw = kh - 1, kw - 1
    pad = [tw // 2, tw - tw // 2, th // 2, th - th // 2]  # F.pad order: W-left, W-right, H-top, H-bottom
    xp = F.pad(nm.operand(x), pad)
    w = nm.operand(weight)
    order = _channel_order(w.shape[1], nm.sum_order)
    if order is not None:
        xp, w = xp[:, order], w[:, order]
    return F.conv2d(xp, w, bias=bias, stride=stride)


def batch_norm(
    y: Tensor,
    sd: StateDict,
    prefix: str,
    nm: Numerics = FP32,
    y_stored: Optional[Tensor] = None,
    reduce_dims: Sequence[int] = (0, 2, 3),
) -> Tensor:
    """``nn.BatchNorm2d`` (eps 1e-5, momentum 0.1, affine).

    Training mode: biased batch variance for normalisation, unbiased for the running
    estimate (torch semantics).  ``y`` supplies the statistics (fp32 accumulators on the
    HIP path) while the affine map is applied to ``y_stored`` (the bf16 value written to
    HBM); for the reference-exact fp32 policy both are the same tensor.
    """
    gamma, beta = sd[prefix + ".weight"].double(), sd[prefix + ".bias"].double()
    y_app = y if y_stored is None else y_stored
    shape = [1, -1] + [1] * (y.dim() - 2)
    if nm.train:
        n = y.numel() // y.shape[1]
        mean = y.double().mean(dim=tuple(reduce_dims))
        var = (y.double() - mean.view(shape)).pow(2).mean(dim=tuple(reduce_dims))
        if nm.running is not None:
            unbiased = var * (n / max(n - 1, 1))
            rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
            nm.running[prefix + ".running_mean"] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean).float().detach()
            nm.running[prefix + ".running_var"] = ((1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unbiased).float().detach()
    else:
        mean, var = sd[prefix + ".running_mean"].double(), sd[prefix + ".running_var"].double()
    # The normalisation itself is evaluated in fp64 and rounded once: autograd through the
    # three coupled paths (value, mean, variance) would otherwise lose ~1e-2 relative
    # accuracy in fp32, which the reference's fused native BN backward does not.
    inv_std = torch.rsqrt(var + BN_EPS)
    z = (y_app.double() - mean.view(shape)) * (gamma * inv_std).view(shape) + beta.view(shape)
    return z.to(y.dtype)


def _conv_bn(
    x: Tensor,
    sd: StateDict,
    conv_key: str,
    bn_prefix: str,
    stride: Tuple[int, int],
    nm: Numerics,
) -> Tensor:
    y = conv2d_same(x, sd[conv_key], stride, nm=nm)
    _trace(nm, y, kind="conv", x=x, w=conv_key, stride=stride, bn=bn_prefix)
    return batch_norm(y, sd, bn_prefix, nm, y_stored=nm.store(y))


# --------------------------------------------------------------------------------------
# blocks  (nn/blocks/__init__.py)
# --------------------------------------------------------------------------------------
def basic_block(
    x: Tensor,
    sd: StateDict,
    prefix: str,
    stride: Tuple[int, int] = (1, 1),
    project: bool = False,
    nm: Numerics = FP32,
) -> Tensor:
    """conv-BN-ReLU-conv(stride)-BN, (+ 1x1(stride)-BN projection), add, ReLU.

    ``nn/blocks/__init__.py:30-81``; the stride sits on the *second* conv (``:43-50``).
    """
    h = F.relu(_conv_bn(x, sd, f"{prefix}.net.0.conv.weight", f"{prefix}.net.1", (1, 1), nm))
    h = _conv_bn(h, sd, f"{prefix}.net.3.conv.weight", f"{prefix}.net.4", stride, nm)
    if project:
        res = _conv_bn(
            x, sd, f"{prefix}.projection_block.0.conv.weight", f"{prefix}.projection_block.1", stride, nm
        )
    else:
        res = x
    return nm.store(F.relu(h + res))


def residual_block(
    x: Tensor,
    sd: StateDict,
    prefix: str,
    num_blocks: int,
    stride: Tuple[int, int] = (1, 1),
    nm: Numerics = FP32,
) -> Tensor:
    """First BasicBlock projects + strides, the rest are plain (``nn/blocks/__init__.py:98-126``)."""
    x = basic_block(x, sd, f"{prefix}.blocks.0", stride, project=True, nm=nm)
    for i in range(1, num_blocks):
        x = basic_block(x, sd, f"{prefix}.blocks.{i}", nm=nm)
    return x


def aggregation_block(
    x1: Tensor,
    x2: Tensor,
    sd: StateDict,
    prefix: str,
    stride: Tuple[int, int],
    padding: Tuple[int, int],
    num_blocks: int,
    nm: Numerics = FP32,
) -> Tensor:
    """ConvTranspose2d-BN-ReLU on x2, add to x1, ResidualBlock (``nn/blocks/__init__.py:146-182``)."""
    w = sd[f"{prefix}.upscale.weight"]  # (Cin, Cout, kh, kw)
    x2o, wo = nm.operand(x2), nm.operand(w)
    order = _channel_order(wo.shape[0], nm.sum_order)
    if order is not None:
        x2o, wo = x2o[:, order], wo[order]
    up = F.conv_transpose2d(x2o, wo, stride=stride, padding=padding)
    _trace(nm, up, kind="convT", x=x2, w=f"{prefix}.upscale.weight", stride=stride, padding=padding, bn=f"{prefix}.normalization")
    up = F.relu(batch_norm(up, sd, f"{prefix}.normalization", nm, y_stored=nm.store(up)))
    return residual_block(nm.store(x1 + up), sd, f"{prefix}.block", num_blocks, nm=nm)


# --------------------------------------------------------------------------------------
# stem  (nn/stems/__init__.py:12-85)
# --------------------------------------------------------------------------------------
def _conv_norm_act(x: Tensor, sd: StateDict, prefix: str, nm: Numerics) -> Tensor:
    """torchvision ``Conv2dNormActivation`` with k=1: conv(no bias) -> BN -> ReLU."""
    xo, wo = nm.operand(x), nm.operand(sd[f"{prefix}.0.weight"])
    order = _channel_order(wo.shape[1], nm.sum_order)
    if order is not None:
        xo, wo = xo[:, order], wo[:, order]
    y = F.conv2d(xo, wo)
    _trace(nm, y, kind="conv1x1", x=x, w=f"{prefix}.0.weight", stride=(1, 1), bn=f"{prefix}.1")
    return F.relu(batch_norm(y, sd, f"{prefix}.1", nm, y_stored=nm.store(y)))


def meta_kernel(
    features: Tensor,
    cart: Tensor,
    sd: StateDict,
    prefix: str,
    num_neighbors: int = 3,
    num_layers: int = 2,
    nm: Numerics = FP32,
) -> Tensor:
    """MetaKernel stem (``nn/stems/__init__.py:64-85``).

    ``F.unfold`` orders its output channels ``c * k*k + tap`` and zero-pads the border, so
    a border neighbour has relative coordinate ``-centre`` (``:74-80``); the positional
    MLP's BatchNorms therefore reduce over (B, taps, H*W).
    """
    k = num_neighbors
    f = basic_block(features, sd, f"{prefix}.projection", project=True, nm=nm)
    B, C, H, W = f.shape
    feat = F.unfold(f, k, padding=k // 2).view(B, C, k * k, H * W)
    nbr = F.unfold(cart, k, padding=k // 2).view(B, 3, k * k, H * W)
    centre = (k * k) // 2
    rel = nbr - nbr[:, :, centre : centre + 1]
    pos = rel
    for i in range(num_layers):
        pos = nm.store(_conv_norm_act(pos, sd, f"{prefix}.positional_kernel.{i}", nm))
    geo = nm.store(pos * feat).view(B, C * k * k, H, W)
    for i in range(num_layers):
        geo = _conv_norm_act(geo, sd, f"{prefix}.fusion_kernel.{i}", nm)
        geo = nm.store(geo)
    return geo


# --------------------------------------------------------------------------------------
# backbone  (nn/backbones/dla.py)
# --------------------------------------------------------------------------------------
_RES = (  # name, stride, num_blocks      (dla.py:37-63)
    ("res1", (1, 1), 2),
    ("res2a", (1, 2), 3),
    ("res2", (1, 2), 3),
    ("res3a", (1, 2), 5),
    ("res3", (1, 2), 5),
)
_AGG = {  # name: (stride, padding, num_blocks)   (dla.py:67-108)
    "agg2": ((1, 4), (1, 2), 2),
    "agg1": ((1, 4), (1, 2), 2),
    "agg2a": ((1, 2), (1, 1), 1),
    "agg3": ((1, 2), (1, 1), 2),
}


def range_partition(features: Tensor, cart: Tensor, mask: Tensor, sd: StateDict, prefix: str, nm: Numerics = FP32) -> Tensor:
    """``RangePartition.forward`` (``nn/stems/__init__.py:88-135``): six range bands [0,15] [10,20] [15,30] [20,40] [30,60] [45,inf]
    (closed on both sides, overlapping; ``lower_bounds`` is an int64 parameter, ``upper_bounds`` fp32 -- the comparison promotes to
    fp32), every input channel repeated once per band and zeroed outside it (channel index band * C_in + c after ``flatten(1, 2)``),
    times the validity mask, then a projecting BasicBlock (``project=True``; kernel size from the weight)."""
    dists = cart.norm(dim=1, keepdim=True)
    lower, upper = sd[f"{prefix}.lower_bounds"], sd[f"{prefix}.upper_bounds"]
    bands = torch.logical_and(dists >= lower, dists <= upper)
    x = (bands[:, :, None] * features[:, None]).flatten(1, 2) * mask
    return basic_block(x, sd, f"{prefix}.projection", project=True, nm=nm)


def range_backbone(x: Tensor, sd: StateDict, prefix: str, nm: Numerics = FP32) -> Dict[int, Tensor]:
    """DLA-style trunk (``nn/backbones/dla.py:110-131``)."""
    r: Dict[str, Tensor] = {}
    h = x
    for name, stride, n in _RES:
        h = residual_block(h, sd, f"{prefix}.{name}", n, stride, nm)
        r[name] = h

    def agg(name: str, a: Tensor, b: Tensor) -> Tensor:
        stride, padding, n = _AGG[name]
        return aggregation_block(a, b, sd, f"{prefix}.{name}", stride, padding, n, nm)

    agg2 = agg("agg2", r["res2"], r["res3"])
    agg1 = agg("agg1", r["res1"], r["res2"])
    agg2a = agg("agg2a", r["res2a"], agg2)
    agg3 = agg("agg3", agg1, agg2a)
    return {1: torch.cat([x, agg3], dim=1), 2: agg2a, 4: agg2, 16: r["res3"]}


def range_net(
    features: Tensor,
    cart: Tensor,
    sd: StateDict,
    prefix: str = "",
    stem_type: str = "META",
    num_neighbors: int = 3,
    num_layers: int = 2,
    nm: Numerics = FP32,
    mask: Optional[Tensor] = None,
) -> Dict[int, Tensor]:
    """Stem dispatch + trunk (``nn/backbones/dla.py:193-208``)."""
    p = prefix + "." if prefix else ""
    if stem_type == "META":
        stem = meta_kernel(features, cart, sd, p + "stem", num_neighbors, num_layers, nm)
    elif stem_type == "BASIC":
        stem = basic_block(features, sd, p + "stem", project=True, nm=nm)
    elif stem_type == "RANGE_PARTITION":
        stem = range_partition(features, cart, mask, sd, p + "stem", nm)
    else:
        raise NotImplementedError(stem_type)
    return range_backbone(stem, sd, p + "net", nm)


# --------------------------------------------------------------------------------------
# dense head  (nn/heads/dense_head.py:13-76)
# --------------------------------------------------------------------------------------
def dense_head(x: Tensor, sd: StateDict, prefix: str, num_blocks: int = 4, nm: Numerics = FP32) -> Tensor:
    """num_blocks x [conv kxk "same" (no bias), BN, ReLU] + final conv (bias, no norm/act)."""
    h = x
    for i in range(num_blocks):
        w = sd[f"{prefix}.blocks.{i}.0.weight"]
        y = conv2d_same(h, w, nm=nm)  # padding="same", odd k => symmetric
        _trace(nm, y, kind="conv", x=h, w=f"{prefix}.blocks.{i}.0.weight", stride=(1, 1), bn=f"{prefix}.blocks.{i}.1")
        h = nm.store(F.relu(batch_norm(y, sd, f"{prefix}.blocks.{i}.1", nm, y_stored=nm.store(y))))
    w = sd[f"{prefix}.blocks.{num_blocks}.0.weight"]
    b = sd[f"{prefix}.blocks.{num_blocks}.0.bias"]
    out = conv2d_same(h, w, bias=b, nm=nm)
    _trace(nm, out, kind="conv", x=h, w=f"{prefix}.blocks.{num_blocks}.0.weight", stride=(1, 1), bn=None, bias=f"{prefix}.blocks.{num_blocks}.0.bias")
    return out


def detector_forward(
    features: Tensor,
    cart: Tensor,
    sd: StateDict,
    stem_type: str = "META",
    num_blocks: int = 4,
    nm: Numerics = FP32,
    backbone_prefix: str = "backbone",
    cls_prefix: str = "head.classification_head.1.0",
    reg_prefix: str = "head.regression_head.1.0",
) -> Tuple[Dict[int, Tensor], Tensor, Tensor]:
    """Backbone + the stride-1 classification / regression towers.

    Mirrors ``Detector.forward`` -> ``DetectionHead.forward`` (``nn/arch/detector.py:196-210``,
    ``nn/heads/detection_head.py:163-187``) for the one-FPN-level, one-task layout every
    shipped rv-* config uses.
    """
    feats = range_net(features, cart, sd, backbone_prefix, stem_type, nm=nm)
    logits = dense_head(feats[1], sd, cls_prefix, num_blocks, nm)
    regressands = dense_head(feats[1], sd, reg_prefix, num_blocks, nm)
    return feats, logits, regressands
