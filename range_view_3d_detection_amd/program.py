"""Fused HIP programs of the reference's modules, expressed on the engine tape.

Each ``*_program`` function follows the reference forward it replaces (cited per function)
but emits fused ops: BatchNorm+ReLU of a producer are folded into the operand load of the
consuming conv (``Lazy``), block outputs are one element-wise pass, "same" padding lives in
the tap tables, the channel concat is written in place.  ``run`` wraps a program into ONE
``torch.autograd.Function`` node whose backward replays the tape in reverse.
"""

from __future__ import annotations

import ctypes
import os

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import _lib as L
from . import engine as E
from .engine import Act, Lazy, Operand, Tape


# ---------------------------------------------------------------------------------------------
# module programs
# ---------------------------------------------------------------------------------------------
# Block programs are written as GENERATORS: wherever a conv -> BatchNorm group can be launched (its inputs exist), the program
# yields the group's specs ``[(layer, input, bn, relu, need_input_grad), ...]`` and receives the results.  ``_drive`` runs ONE
# program (each yielded group -> ``engine.conv_bn_many``); ``_drive_many`` runs several INDEPENDENT programs side by side and
# merges the groups they yield at the same step into one ``conv_bn_many`` call -- under SyncBN the BatchNorm statistics of
# every layer of a merged group travel in ONE all-reduce, forward and backward (the two convs a BasicBlock applies to its
# input; layer i of the classification and regression towers; the layers of the backbone's independent agg1 / agg2 branches).
def _drive(t: Tape, gen):
    try:
        specs = next(gen)
        while True:
            specs = gen.send(E.conv_bn_many(t, specs))
    except StopIteration as stop:
        return stop.value


def _drive_many(t: Tape, gens):
    gens = list(gens)
    results = [None] * len(gens)
    pending = {}
    for i, g in enumerate(gens):
        try:
            pending[i] = next(g)
        except StopIteration as stop:
            results[i] = stop.value
    while pending:
        order = sorted(pending)
        flat = [spec for i in order for spec in pending[i]]
        outs = E.conv_bn_many(t, flat)
        nxt, o = {}, 0
        for i in order:
            n = len(pending[i])
            try:
                nxt[i] = gens[i].send(outs[o : o + n])
            except StopIteration as stop:
                results[i] = stop.value
            o += n
        pending = nxt
    return results


def _basic_block_gen(t: Tape, m: nn.Module, x: Operand, out: Optional[Act] = None, need_input_grad: bool = True):
    """``BasicBlock.forward`` (nn/blocks/__init__.py:68-81): relu_(net(x) + proj(x))."""
    c1, bn1, _, c2, bn2 = m.net
    l1 = E.tap_layer(c1.conv)
    if E._smallk_eligible(l1, x, True, need_input_grad):  # 5/6-channel stem projection: the small-K element-wise path
        h1 = E.conv_bn(t, l1, x, bn1, relu=True, need_input_grad=need_input_grad)
        res: Operand = x
        if m.projection_block is not None:
            pc, pbn = m.projection_block
            res = E.conv_bn(t, E.tap_layer(pc.conv), x, pbn, relu=False, need_input_grad=need_input_grad)
        fused = E.conv_bn_residual(t, E.tap_layer(c2.conv), h1, bn2, res, False, True, out=out)
        if fused is not None:
            return fused
        (h2,) = yield [(E.tap_layer(c2.conv), h1, bn2, False, True)]
        return E.CombineOp(t, h2, res, relu_out=True, out=out).out
    if m.projection_block is not None:
        # net.0 and the projection conv read the same input: both convs first, then both BatchNorms
        pc, pbn = m.projection_block
        h1, res = yield [(l1, x, bn1, True, need_input_grad), (E.tap_layer(pc.conv), x, pbn, False, need_input_grad)]
    else:
        (h1,) = yield [(l1, x, bn1, True, need_input_grad)]
        if isinstance(x, Lazy):
            x = E.CombineOp(t, x, None, relu_out=False).out
        res = x
    # inference: the block's sum and ReLU leave the second conv's own launch (rv_tap_residual)
    fused = E.conv_bn_residual(t, E.tap_layer(c2.conv), h1, bn2, res, False, True, out=out)
    if fused is not None:
        return fused
    (h2,) = yield [(E.tap_layer(c2.conv), h1, bn2, False, True)]
    return E.CombineOp(t, h2, res, relu_out=True, out=out).out


def _residual_block_gen(t: Tape, m: nn.Module, x: Operand, out: Optional[Act] = None):
    """``ResidualBlock.forward`` (nn/blocks/__init__.py:123-126)."""
    blocks = list(m.blocks)
    for i, b in enumerate(blocks):
        x = yield from _basic_block_gen(t, b, x, out=out if i == len(blocks) - 1 else None)
    return x


def _aggregation_block_gen(t: Tape, m: nn.Module, x1: Act, x2: Act, out: Optional[Act] = None):
    """``AggregationBlock.forward`` (nn/blocks/__init__.py:165-182): x1 + relu(bn(convT(x2))) -> ResidualBlock."""
    s = E.conv_bn_residual(t, E.tap_layer(m.upscale), x2, m.normalization, x1, True, False)  # (inference)
    if s is None:
        (up,) = yield [(E.tap_layer(m.upscale), x2, m.normalization, True, True)]
        s = E.CombineOp(t, x1, up, relu_out=False).out
    return (yield from _residual_block_gen(t, m.block, s, out=out))


def basic_block_program(t: Tape, m: nn.Module, x: Operand, out: Optional[Act] = None, need_input_grad: bool = True) -> Act:
    return _drive(t, _basic_block_gen(t, m, x, out, need_input_grad))


def residual_block_program(t: Tape, m: nn.Module, x: Operand, out: Optional[Act] = None) -> Act:
    return _drive(t, _residual_block_gen(t, m, x, out))


def aggregation_block_program(t: Tape, m: nn.Module, x1: Act, x2: Act, out: Optional[Act] = None) -> Act:
    return _drive(t, _aggregation_block_gen(t, m, x1, x2, out))


def meta_kernel_program(t: Tape, m: nn.Module, features: Act, cart: Tensor, out: Optional[Act] = None) -> Act:
    """``MetaKernel.forward`` (nn/stems/__init__.py:64-85) without materialising either ``F.unfold``."""
    f = basic_block_program(t, m.projection, features, need_input_grad=False)
    rel = E.MetaRelativeOp(t, cart).out
    pos: Operand = rel
    n_pos = len(m.positional_kernel)
    blocks = list(m.positional_kernel)
    geo: Optional[Operand] = None
    if n_pos == 2 and E.pos_modulate_eligible(t, E.tap_layer(blocks[0][0]), E.tap_layer(blocks[1][0]), rel, f):
        # inference: positional pair and modulation in one kernel, nothing of the 9x grid but `geo` itself is stored
        geo = E.PosModulateOp(t, E.tap_layer(blocks[0][0]), blocks[0][1], E.tap_layer(blocks[1][0]), blocks[1][1], rel, f).out
        blocks = []
    elif n_pos == 2 and E.pos_pair_eligible(E.tap_layer(blocks[0][0]), E.tap_layer(blocks[1][0]), rel):
        # 3 -> 256 -> 256: both layers in one persistent streaming kernel (csrc/posconv.hip)
        pos = E.pos_pair(t, E.tap_layer(blocks[0][0]), blocks[0][1], E.tap_layer(blocks[1][0]), blocks[1][1], rel)
        blocks = []
    for i, blk in enumerate(blocks):
        # the LAST positional layer feeds MetaModulateOp, which folds its BatchNorm+ReLU itself and needs the Lazy form
        # (num_layers == 1: that is the 3 -> C layer, which would otherwise take the small-K fast path)
        pos = E.conv_bn(t, E.tap_layer(blk[0]), pos, blk[1], relu=True, need_input_grad=(i > 0), smallk=(i + 1 < n_pos), fold_eval=(i + 1 < n_pos))
    if geo is None:
        geo = E.MetaModulateOp(t, pos, f).out
    c = m.out_channels
    last = len(m.fusion_kernel) - 1
    for i, blk in enumerate(m.fusion_kernel):
        kw = {"in_perm": (c, m.num_neighbors**2)} if i == 0 else {}
        geo = E.conv_bn(t, E.tap_layer(blk[0], **kw), geo, blk[1], relu=True, out=out if i == last else None)
    if out is not None and geo is out:  # (inference: the last fusion conv wrote the activation into its place)
        return out
    return E.CombineOp(t, geo, None, relu_out=False, out=out).out


def range_partition_program(t: Tape, m: nn.Module, features: Tensor, cart: Tensor, mask: Tensor, out: Optional[Act] = None) -> Act:
    """``RangePartition.forward`` (nn/stems/__init__.py:121-135): the banded, masked operand in one pass over the fp32 inputs
    (``rv_range_partition``), then the projecting BasicBlock.  No gradient leaves it: its inputs are the sweep itself."""
    n, c, h, w = features.shape
    bands = m.lower_bounds.numel()
    x = Act.empty(n, h, w, bands * c, t.device)
    feats32, cart32 = features.contiguous().float(), cart.contiguous().float()  # (temporaries referenced across the launch)
    if mask.dtype == torch.bool:
        mask8 = mask.contiguous().view(torch.uint8)
    else:
        # the reference multiplies (`features * mask`, nn/stems/__init__.py:128): only a 0/1 mask is the same thing as a gate
        if not bool(((mask == 0) | (mask == 1)).all()):
            raise L.RvError("RangePartition: a non-boolean mask must hold only 0 and 1 (the reference MULTIPLIES by it; the kernel gates)")
        mask8 = mask.to(torch.uint8).contiguous()
    bounds = m.__dict__.get("_rv_bounds")
    ver = (m.lower_bounds._version, m.upper_bounds._version, m.lower_bounds.data_ptr())
    if bounds is None or bounds[0] != ver:  # (host copies of the twelve frozen numbers: read back once, not per step)
        lo = (ctypes.c_float * bands)(*[float(v) for v in m.lower_bounds.detach().flatten().tolist()])
        hi = (ctypes.c_float * bands)(*[float(v) for v in m.upper_bounds.detach().flatten().tolist()])
        m.__dict__["_rv_bounds"] = bounds = (ver, lo, hi)
    L.call("rv_range_partition", L.ptr(feats32), L.ptr(cart32), L.ptr(mask8), L.i32(n), L.i32(c), L.i32(h), L.i32(w), bounds[1], bounds[2], L.i32(bands),
           x.ptr(), L.i32(x.ld), L.stream_ptr())
    return basic_block_program(t, m.projection, x, out=out, need_input_grad=False)


def range_backbone_program(t: Tape, m: nn.Module, stem: Act, feat1: Optional[Act]) -> Dict[int, Act]:
    """``RangeBackbone.forward`` (nn/backbones/dla.py:110-131)."""
    res1 = residual_block_program(t, m.res1, stem)
    res2a = residual_block_program(t, m.res2a, res1)
    res2 = residual_block_program(t, m.res2, res2a)
    res3a = residual_block_program(t, m.res3a, res2)
    res3 = residual_block_program(t, m.res3, res3a)
    # agg2 (res2, res3) and agg1 (res1, res2) are independent branches: run side by side, their BatchNorm groups merged
    agg2, agg1 = _drive_many(t, [_aggregation_block_gen(t, m.agg2, res2, res3), _aggregation_block_gen(t, m.agg1, res1, res2)])
    agg2a = aggregation_block_program(t, m.agg2a, res2a, agg2)
    c0 = stem.cp
    if feat1 is not None:  # concat written in place: stem already sits in channels [0, c0)
        agg3 = aggregation_block_program(t, m.agg3, agg1, agg2a, out=feat1.slice(c0, 2 * c0))
        cat = feat1
        cat.c = 2 * c0
    else:
        agg3 = aggregation_block_program(t, m.agg3, agg1, agg2a)
        cat = E.ConcatOp(t, [stem, agg3]).out
    return {1: cat, 2: agg2a, 4: agg2, 16: res3}


def range_net_program(t: Tape, m: nn.Module, features: Tensor, cart: Tensor, mask: Optional[Tensor] = None) -> Dict[int, Act]:
    """``RangeNet.forward`` (nn/backbones/dla.py:193-208)."""
    n, c, h, w = features.shape
    c0 = m.layers[0]
    in_place = c0 % 32 == 0
    feat1 = Act.empty(n, h, w, 2 * c0, t.device) if in_place else None
    stem_out = feat1.slice(0, c0) if in_place else None
    if m.stem_type == "RANGE_PARTITION":
        if mask is None:
            raise L.RvError("RangeNet(stem_type=RANGE_PARTITION) needs the sweep's validity mask (x['mask'])")
        stem = range_partition_program(t, m.stem, features, cart, mask, out=stem_out)
        return range_backbone_program(t, m.net, stem, feat1)
    x = Act.empty(n, h, w, c, t.device, zero=True)
    feats32 = features.contiguous().float()  # keep the temporary referenced across the launch
    L.call("rv_nchw_f32_to_nhwc_bf16", L.ptr(feats32), L.i32(n), L.i32(c), L.i32(h), L.i32(w), x.ptr(),
           L.i32(x.ld), L.i32(0), L.stream_ptr())
    if m.stem_type == "META":
        stem = meta_kernel_program(t, m.stem, x, cart, out=stem_out)
    elif m.stem_type == "BASIC":
        stem = basic_block_program(t, m.stem, x, out=stem_out, need_input_grad=False)
    else:
        raise NotImplementedError("This stem type is not implemented!")
    return range_backbone_program(t, m.net, stem, feat1)


def dense_head_program(t: Tape, m: nn.Module, x: Act) -> E.ConvOp:
    """``DenseHead.forward`` (nn/heads/dense_head.py:74-76): towers of conv-BN-ReLU + a biased final conv (fp32 out)."""
    blocks = list(m.blocks)
    h: Operand = x
    for blk in blocks[:-1]:
        h = E.conv_bn(t, E.tap_layer(blk[0]), h, blk[1], relu=True)
    return E.ConvOp(t, E.tap_layer(blocks[-1][0]), h, stats=False, out_f32=True)


def dense_head_pair_program(t: Tape, cls_head: nn.Module, reg_head: nn.Module, x: Act) -> Tuple[E.ConvOp, E.ConvOp]:
    """The classification and the regression tower of one (stride, task), layer by layer side by side: layer i of both towers
    is launched before either BatchNorm is finalised, so that under SyncBN the two layers share one all-reduce in each
    direction (``engine.conv_bn_many``).  Same arithmetic as two ``dense_head_program`` calls."""
    ca, cb = list(cls_head.blocks), list(reg_head.blocks)
    if len(ca) != len(cb):
        return dense_head_program(t, cls_head, x), dense_head_program(t, reg_head, x)
    ha: Operand = x
    hb: Operand = x
    for ba, bb in zip(ca[:-1], cb[:-1]):
        ha, hb = E.conv_bn_many(t, [(E.tap_layer(ba[0]), ha, ba[1], True, True), (E.tap_layer(bb[0]), hb, bb[1], True, True)])
    return (E.ConvOp(t, E.tap_layer(ca[-1][0]), ha, stats=False, out_f32=True),
            E.ConvOp(t, E.tap_layer(cb[-1][0]), hb, stats=False, out_f32=True))


# ---------------------------------------------------------------------------------------------
# autograd bridge
# ---------------------------------------------------------------------------------------------
# Operand type of an eval-mode program.  The reference evaluates under ``torch.autocast(device_type="cuda", dtype=torch.float16)``
# (nn/arch/detector.py:329-340 with conf/model/range_view.yaml:26 ``eval_precision: 16``) and trains under bf16 autocast
# (``precision: bf16-mixed``).  None: follow the caller's autocast state -- an eval program inside an fp16 autocast region runs
# on the fp16-operand build of the library (librv3d_hip_f16.so: v_mfma_f32_16x16x32_f16, fp16 activations); anything else runs
# bf16.  "f16" / "bf16": force it (bench.py's forward_only leg, tests).
EVAL_OPERAND: Optional[str] = None


def operand_for(training: bool) -> str:
    fp16_autocast = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16
    if training:
        if fp16_autocast:
            raise NotImplementedError("training under torch.autocast(float16): the reference trains in bf16-mixed (conf/trainer/train.yaml:14); "
                                      "fp16 operands are built for inference only (no loss scaling on this path)")
        return "bf16"
    if EVAL_OPERAND is not None:
        return EVAL_OPERAND
    return "f16" if fp16_autocast else "bf16"


_MATERIALIZE_GRADS = False  # (True: autograd's default zero-filled gradients; A/B in-process)


class _ProgramFn(torch.autograd.Function):
    """One autograd node for a whole fused program.

    ``build(tape, *tensor_inputs)`` -> (list of input Acts wanting gradients (or None), list of outputs), where an
    output is an ``Act`` (bf16 NHWC, returned as a channels_last NCHW view) or a ``ConvOp`` with an fp32 result.
    """

    @staticmethod
    def forward(ctx, build: Callable, training: bool, n_in: int, *args):
        inputs = args[:n_in]
        for x in inputs:
            if isinstance(x, Tensor):
                E._require_cuda(x, "input tensor")
        dev = next(x.device for x in inputs if isinstance(x, Tensor))
        tape = Tape(training, dev)
        if training:
            E.prepack_stale()  # weight images of every layer the optimiser touched, one launch
        with L.operand(operand_for(training)):
            in_acts, outs = build(tape, *inputs)
        if tape.bn_counters:
            torch._foreach_add_(tape.bn_counters, 1)  # num_batches_tracked of every BatchNorm on the tape, one launch
            tape.bn_counters = []
        # an output nobody differentiates (the backbone's stride-2/4/16 maps when the head reads stride 1 only) gets NO gradient
        # instead of a materialised zero tensor, which would be converted, copied and then accumulated into by every real writer
        ctx.set_materialize_grads(_MATERIALIZE_GRADS)
        ctx.tape, ctx.in_acts, ctx.outs, ctx.n_in = tape, in_acts, outs, n_in
        ctx.params = args[n_in:]
        ctx.in_meta = [(x.dtype, x.shape) if isinstance(x, Tensor) else None for x in inputs]
        result = []
        for o in outs:
            if isinstance(o, Act):
                result.append(o.nchw())
            else:  # fp32 conv output (N,H,W,cp) -> (N,c,H,W) view
                result.append(o.out_t[..., : o.layer.c_out].permute(0, 3, 1, 2))
        return tuple(result)

    @staticmethod
    def backward(ctx, *gouts):
        from . import engine_bwd

        tape: Tape = ctx.tape
        for o, g in zip(ctx.outs, gouts):
            if g is None:
                continue
            if isinstance(o, Act):
                tape.set_grad(o, engine_bwd.grad_act_like(o, g))
            else:
                engine_bwd.seed_f32_output_grad(tape, o, g)
        tape.backward()
        gin: List[Optional[Tensor]] = []
        for a, meta in zip(ctx.in_acts, ctx.in_meta):
            if a is None or meta is None or id(a) not in tape.grads:
                gin.append(None)
            else:
                gin.append(tape.grads[id(a)].nchw().to(meta[0]))
        gparams = [tape.param_grads.get(id(p)) for p in ctx.params]
        if E.GRAD_SYNC is not None:  # data parallel: this node's gradients into the flat buffer, its all-reduce under way
            gparams = E.GRAD_SYNC.reduce_node(ctx.params, gparams)
        ctx.tape = None  # free activations
        return (None, None, None, *gin, *gparams)


def run(build: Callable, module: nn.Module, inputs: Sequence[Tensor]) -> Tuple[Tensor, ...]:
    params = [p for p in module.parameters()]
    return _ProgramFn.apply(build, module.training, len(inputs), *inputs, *params)


def standalone(m: nn.Module, *inputs: Tensor):
    """Run a single block-level module on NCHW tensors (API parity with the reference's per-module ``forward``)."""
    from .nn.blocks import AggregationBlock, BasicBlock, ResidualBlock
    from .nn.modules.conv import Conv2dSame
    from .nn.stems import MetaKernel, RangePartition

    def build(t: Tape, *xs: Tensor):
        if isinstance(m, RangePartition):
            feats, cart, mask = xs
            return [None, None, None], [range_partition_program(t, m, feats, cart, mask)]
        if isinstance(m, MetaKernel):
            feats, cart = xs
            n, c, h, w = feats.shape
            x = Act.empty(n, h, w, c, t.device, zero=True)
            x.data[..., :c].copy_(feats.permute(0, 2, 3, 1))
            return [None, None], [meta_kernel_program(t, m, x, cart)]
        acts = [Act.from_nchw(x) for x in xs]
        if isinstance(m, Conv2dSame):
            return acts, [E.ConvOp(t, E.tap_layer(m.conv), acts[0]).out]
        if isinstance(m, BasicBlock):
            return acts, [basic_block_program(t, m, acts[0])]
        if isinstance(m, ResidualBlock):
            return acts, [residual_block_program(t, m, acts[0])]
        if isinstance(m, AggregationBlock):
            return acts, [aggregation_block_program(t, m, acts[0], acts[1])]
        raise NotImplementedError(type(m))

    out = run(build, m, inputs)
    return out[0]
