"""Backward halves of the engine ops (hand-derived; every kernel goes through the C ABI).

Gradient flow on the tape, in reverse op order:

* ``CombineOp``   dOut -> (a) plain operands: ``rv_ew_mask_grad``; (b) Lazy operands: hand
  ``(dOut, OUT)`` to the BatchNorm that produced them;
* ``BnOp``        ``rv_bn_bwd_reduce`` -> ``rv_bn_bwd_finalize`` (dgamma, dbeta, coefficients)
  -> ``rv_bn_bwd_apply`` = gradient w.r.t. the raw conv output (bf16);
* ``ConvOp``      input gradient = the opposite tap form (gather <-> scatter) on the same
  geometry; weight gradient = ``rv_tap_wgrad`` (split-K TN GEMM) -> ``rv_unpack_weight_grad``.
"""

from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from . import engine as E
from .engine import Act, Lazy, Tape, pad32


def grad_act_like(o: Act, g: Tensor) -> Act:
    """Incoming autograd gradient (N,C,H,W), any layout/dtype -> NHWC bf16 Act shaped like ``o``."""
    nhwc = g.permute(0, 2, 3, 1)
    if g.dtype == torch.bfloat16 and g.shape[1] == o.cp and nhwc.is_contiguous():
        return Act(nhwc, o.c)
    out = Act.empty(o.N, o.H, o.W, o.c, g.device, zero=(g.shape[1] != o.cp))
    out.data[..., : g.shape[1]].copy_(nhwc)
    return out


def seed_f32_output_grad(t: Tape, op: "E.ConvOp", g: Tensor) -> None:
    """Gradient of an fp32 conv output (final head conv): bf16 copy for the kernels + the bias gradient."""
    c = op.layer.c_out
    n, h, w = op.out_t.shape[:3]
    act = Act.empty(n, h, w, c, g.device, zero=(c % 32 != 0))
    act.data[..., :c].copy_(g.permute(0, 2, 3, 1))
    t.raw_grad[id(op)] = act
    if op.layer.bias is not None:
        if c < 8:
            # (torch's reduction over (N, H, W) of a tensor with fewer than 8 channels takes 740 us on the 3-class Waymo head -- 35 us
            #  as a sum over W followed by a sum over the rows: profiles/tools/mb_bias_sum.py)
            t.add_param_grad(op.layer.bias, g.float().permute(0, 2, 3, 1).sum(dim=2).sum(dim=(0, 1)))
        else:
            t.add_param_grad(op.layer.bias, g.float().sum(dim=(0, 2, 3)))


def _conv_out_grad(op: "E.ConvOp", t: Tape) -> Optional[Act]:
    if op.out_f32:
        return t.raw_grad.get(id(op))
    g = t.raw_grad.get(id(op.out))
    if g is None and id(op.out) in t.grads and id(op.out) in t.written:
        g = t.grads[id(op.out)]
    return g


def conv_backward(op: "E.ConvOp", t: Tape) -> None:
    dout = _conv_out_grad(op, t)
    if dout is None:
        return
    _release_held_wgrads(t)  # (weight gradients of earlier layers that met no all-reduce since)
    layer, g = op.layer, op.layer.geom
    src, sc, sh, in_flags = E._operand_parts(op.x)
    if op.x_plain is not None:
        # the forward pass wrote relu(bn(.)) out for the DMA kernel: the weight gradient reads it too
        src, sc, sh, in_flags = op.x_plain, None, None, 0
    fwd = layer.fwd_form
    bwd = "scatter" if fwd == "gather" else "gather"
    sp = op.shape
    # ---- input gradient: the opposite tap form -------------------------------------------------
    fused_first = POS_BWD_FUSE and op.pos_first is not None and op.need_input_grad and t.training
    early_ready = None
    if fused_first:
        # second positional layer of the MetaKernel stem: dh1 = dy2 W2 is consumed in registers by the first layer's BatchNorm
        # backward + weight gradient (rv_pos_backward_sums) -- no input-gradient tensor, no second pass over it
        _wait_chained_wgrad(t)
        _pos_pair_backward(op, t, dout)
    elif op.need_input_grad and _head_final_eligible(op, t, dout):
        # final 1x1 conv of a head tower behind conv -> BatchNorm -> ReLU: no input-gradient tensor -- the BatchNorm backward of that unit
        # recomputes dA = W^T dY in both of its passes (rv_head_final_bwd_sums / _apply, csrc/headfinal.hip)
        _head_final_sums(op, t, dout)
        return  # (the final conv's weight gradient came out of the same pass)
    elif op.need_input_grad:
        if isinstance(op.x, Lazy):
            dst, accumulate = t.lazy_grad_target(op.x)
            dst._rv_owned = True
        else:
            dst, accumulate = t.grad_buffer(op.x)
        shape = L.TapShape(sp.N, sp.H, sp.Wu, sp.Wv, dout.ld, dst.ld, L.OUT_ACCUM if accumulate else 0)
        bg, bshape = g, shape
        gf = layer.fold_geom() if (bwd == "gather" and g.stride_w > 1) else None
        if (gf is not None and dout.ld == dout.cp and dout.W == g.stride_w * sp.Wu
                and E._dma_eligible(gf, sp.N, sp.H, sp.Wu, sp.Wu, g.stride_w * dout.ld, dst.ld, False)):
            # backward-data of a ConvTranspose2d = a strided gather over dOut: stride-1 on the folded view (engine.FOLD_STRIDED)
            bg = gf
            bshape = L.TapShape(sp.N, sp.H, sp.Wu, sp.Wu, g.stride_w * dout.ld, dst.ld, shape.flags)
            wp = layer.packed_folded()
        elif (E.FOLD_STRIDED and bwd == "scatter" and g.stride_w == 2 and g.kh == 1 and g.kw == 1 and g.pad_w == 0 and dst.ld == dst.cp
              and dst.W == 2 * sp.Wu):
            # backward-data of a 1x1 stride-2 conv: only the even columns receive anything -- a STRIDE-1 1x1 launch into the
            # even-column view of the gradient (pixel pitch 2 ld; the one-tap scatter image is the same bytes), odd columns zeroed
            # unless the buffer already holds another consumer's contribution
            if not accumulate:
                dst.data.zero_()
            bg = L.TapGeom(1, 1, 1, 0, 0, g.cu, g.cv)
            bshape = L.TapShape(sp.N, sp.H, sp.Wu, sp.Wu, dout.ld, 2 * dst.ld, shape.flags)
            wp = layer.packed(bwd)
        else:
            wp = layer.packed(bwd)
        call = lambda: L.call("rv_tap_" + bwd, ctypes.byref(bg), ctypes.byref(bshape), dout.ptr(), None, None, L.ptr(wp), None,
                              dst.ptr(), None, L.stream_ptr())
        if E.BNB_FUSE and bg is g and isinstance(op.x, Lazy) and not accumulate and op.x.bn.mean is not None:
            # first (often only) consumer of relu(bn(y)): this launch can form that BatchNorm's backward sums on the way out
            rows = L.load().rv_tap_bnb_rows(ctypes.byref(g), ctypes.byref(shape), L.i32(1 if bwd == "scatter" else 0))
            if rows > 0:
                lz, st = op.x, op.x.bn
                partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, dst.cp), dtype=torch.float32, device=t.device)
                epi = L.BnbEpilogue(lz.raw.ptr().value, lz.raw.ld, L.BNB_RELU_Z if lz.relu else 0, L.ptr(st.scale).value, L.ptr(st.shift).value,
                                    L.ptr(st.mean).value, L.ptr(st.invstd).value, L.ptr(partial).value)
                call = lambda: L.call("rv_tap_data_grad_bnb", ctypes.byref(g), ctypes.byref(shape), L.i32(1 if bwd == "scatter" else 0), dout.ptr(),
                                      L.ptr(wp), dst.ptr(), ctypes.byref(epi), L.stream_ptr())
                t.lazy_sums[id(lz)] = (partial, rows, dst)
        if E.OVERLAP_WGRAD and E.EARLY_WGRAD_FILL > 0:
            # a persistent backward-data launch whose LAST round of tiles fills only part of the chip (1328 tiles on 256 CUs: five full
            # rounds and 48 tiles): the weight gradient of the same layer -- it reads dOut and the layer input, both complete -- is
            # released on an event recorded BEFORE this launch instead of after it, so both are eligible at once.  Nothing orders the
            # two on the hardware queues: the intent is that the weight gradient's workgroups take CUs as the last round leaves them idle;
            # when they take some earlier, this launch's rounds stretch by the same amount.  Kept on the measurement alone: rv-waymo
            # -1.0 ms chained / -0.15 ms free-running, rv-av2 (no ragged launch) unchanged (profiles/r04_ab_notes.md, r05_ab_notes.md)
            info = (ctypes.c_int32 * 4)()
            if (L.load().rv_tap_launch_info(ctypes.byref(bg), ctypes.byref(bshape), 1 if bwd == "scatter" else 0, info) == 0 and info[0] == 6):
                wg_tiles, cus = info[2] * info[3], E.cu_count(t.device)
                if wg_tiles / (cus * ((wg_tiles + cus - 1) // cus)) < E.EARLY_WGRAD_FILL:
                    early_ready = torch.cuda.Event()
                    early_ready.record()
                    t.chained_wgrad = None  # ... and this launch does not wait for the weight gradient before it either: its own tail is where that one ends
        _wait_chained_wgrad(t)
        if E.PROFILE is not None:
            E._launch(E.tap_kernel_name(bg, bshape, bwd == "scatter"), E.tap_flops(g, shape), call, E.tap_bytes(g, shape))
        else:
            call()
        if not isinstance(op.x, Lazy):
            t.mark_written(op.x)
    # ---- weight gradient ---------------------------------------------------------------------------
    if fwd == "gather":
        u, v, v_affine = dout, src, 1
    else:
        u, v, v_affine = src, dout, 0
    wshape = L.TapShape(sp.N, sp.H, sp.Wu, sp.Wv, 0, 0, in_flags | L.WGRAD_TORCH_LAYOUT)
    # strided layers: the weight gradient on the stride-1 FOLDED view of the fine tensor (wgrad3 instead of the generic kernel),
    # then rv_unfold_weight_grad picks the kernel's own entries out of the folded gradient
    wg, wsh, ld_v = g, wshape, v.ld
    gfw = layer.fold_geom() if (g.stride_w > 1 and sc is None and in_flags == 0 and v.ld == v.cp and v.W == g.stride_w * sp.Wu) else None
    if gfw is not None:
        wg, wsh, ld_v = gfw, L.TapShape(sp.N, sp.H, sp.Wu, sp.Wu, 0, 0, L.WGRAD_TORCH_LAYOUT), g.stride_w * v.ld

    def run_wgrad() -> None:
        ws_bytes = L.load().rv_tap_wgrad_workspace_bytes(ctypes.byref(wg), ctypes.byref(wsh))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=t.device)
        grad = torch.empty((wg.cu, wg.cv, wg.kh, wg.kw), dtype=torch.float32, device=t.device)
        wname = "wgrad_kernel(+reduce)"
        if E.PROFILE is not None:
            winfo = (ctypes.c_int32 * 4)()
            L.call("rv_tap_wgrad_info", ctypes.byref(wg), ctypes.byref(wsh), winfo)
            wname = ("wgrad_kernel", "wgrad_kernel", "wgrad2_kernel", "wgrad3_kernel")[winfo[0]] + "(+reduce)"
        if os.environ.get("RV3D_PROFILE_SHAPES"):
            wname += f" k{g.kh}x{g.kw}s{g.stride_w} {g.cu}<->{g.cv} {wshape.N}x{wshape.H}x{wshape.Wu}"
        # the split-K reduction (one small launch behind the kernel) writes the torch layout dT[cu][cv][kh][kw] itself
        # (RV_WGRAD_TORCH_LAYOUT): no unpack pass
        E._launch(wname, E.tap_flops(g, wshape),
                  lambda: L.call("rv_tap_wgrad", ctypes.byref(wg), ctypes.byref(wsh), u.ptr(), L.i32(u.ld), v.ptr(), L.i32(ld_v),
                                 L.ptr(sc), L.ptr(sh), L.i32(v_affine), L.ptr(grad), L.ptr(ws), L.stream_ptr()),
                  E.tap_bytes(g, wshape, wgrad=True))
        if wg is not g:
            folded, grad = grad, torch.empty((g.cu, g.cv, g.kh, g.kw), dtype=torch.float32, device=t.device)
            L.call("rv_unfold_weight_grad", ctypes.byref(g), L.ptr(folded), L.ptr(grad), L.i32(0), L.stream_ptr())
        t.add_param_grad(layer.weight, layer.unpermute_grad(grad))

    small = E.OVERLAP_MAX_TFLOP is None or E.tap_flops(g, wshape) < 1e12 * E.OVERLAP_MAX_TFLOP
    if E.OVERLAP_WGRAD and E.HOLD_WGRAD_FOR_SYNC_BN and E.sync_bn_active():
        # Synchronised BatchNorm statistics over real ranks: the all-reduce of the NEXT BatchNorm backward is an RCCL kernel that needs a
        # CU slot, and it becomes ready at the same moment as this weight gradient (both wait for the backward-data launch above).  A
        # weight gradient that reaches the CUs first holds all of them for its whole duration and the collective -- the critical chain --
        # waits behind it.  So the launch is HELD until that all-reduce has been enqueued and waited for on the main stream
        # (_release_held_wgrads: bn_backward_finish / the next conv / the end of the pass); it then runs beside the finalize + apply passes
        # exactly as in the local case.
        t.held_wgrads.append(run_wgrad)
    elif E.OVERLAP_WGRAD and (small or E.OVERLAP_CHAIN):
        side = E.side_stream(t.device)
        ready = early_ready
        if ready is None:
            ready = torch.cuda.Event()
            ready.record()  # dout (and everything before it on the main stream) is complete at this point of the main stream
        side.wait_event(ready)
        with torch.cuda.stream(side):
            run_wgrad()
            if E.OVERLAP_CHAIN and not small:  # chained: the main stream's next MFMA-bound launch waits for this one (_wait_chained_wgrad)
                t.chained_wgrad = torch.cuda.Event()
                t.chained_wgrad.record()
        t.used_side_stream = True
    else:
        run_wgrad()


HEAD_FINAL_FUSE = True  # (module attribute: tests flip it in-process)
# (Round 5 built the same idea for the stem's first fusion conv -- its backward-data GEMM recomputed inside both passes of the modulation
#  backward, `csrc/metachain.hip` -- parity-green and 0.4-1.0 ms per step SLOWER than the launches it replaced; round 6 removed it from the
#  library: profiles/r05_metachain.md is the record.)


def _head_final_eligible(op: "E.ConvOp", t: Tape, dout: Act) -> bool:
    """``op`` is a tower's final conv (1x1, stride 1, at most 32 output channels, fp32 output) whose ONLY input is the activated
    output of a conv -> BatchNorm (-> ReLU) unit in training mode that nobody else has contributed a gradient to."""
    x, lay = op.x, op.layer
    g = lay.geom
    return (HEAD_FINAL_FUSE and t.training and op.out_f32 and isinstance(x, Lazy) and x.relu and x.bn.mean is not None and id(x) not in t.lazy_in
            and id(x) not in t.head_final and lay.fwd_form == "gather" and g.kh == 1 and g.kw == 1 and g.stride_w == 1 and lay.in_perm is None
            and pad32(g.cu) == 32 and x.raw.cp % 256 == 0 and x.raw.cp == pad32(g.cv) and dout.ld >= 32 and dout.pixels == x.raw.pixels)


def _head_final_sums(op: "E.ConvOp", t: Tape, dY: Act) -> None:
    """First pass of the fused form: the BatchNorm-backward sums of the unit in front of the final conv (kept on the tape for
    ``bn_backward_begin``) and the final conv's own weight gradient, from ONE pass over the unit's raw output."""
    lazy, layer = op.x, op.layer
    st, raw = lazy.bn, lazy.raw
    wp = layer.packed("scatter")
    rows = L.load().rv_head_final_bwd_rows(L.i64(raw.pixels))
    partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, raw.cp), dtype=torch.float32, device=t.device)
    dw_partial = torch.empty((rows, 32 * raw.cp), dtype=torch.float32, device=t.device)
    head = (L.i64(raw.pixels), L.i32(raw.cp), raw.ptr(), L.i32(raw.ld), dY.ptr(), L.i32(dY.ld), L.ptr(wp), L.ptr(st.scale), L.ptr(st.shift),
            L.ptr(st.mean), L.ptr(st.invstd), L.i32(1))
    L.call("rv_head_final_bwd_sums", *head, L.ptr(partial), L.ptr(dw_partial), L.stream_ptr())
    dw = torch.empty((32, raw.cp), dtype=torch.float32, device=t.device)
    L.call("rv_reduce_rows", L.ptr(dw_partial), L.i32(rows), L.i32(32 * raw.cp), L.ptr(dw), L.stream_ptr())
    g = layer.geom
    t.add_param_grad(layer.weight, dw[: g.cu, : g.cv].reshape(g.cu, g.cv, 1, 1))
    t.head_final[id(lazy)] = (head, partial, rows, dY, wp)


def _release_held_wgrads(t: Tape) -> None:
    """Launch the weight gradients held back for a SyncBN all-reduce (see conv_backward) on the side stream, behind everything the
    main stream has been given so far."""
    if not t.held_wgrads:
        return
    side = E.side_stream(t.device)
    ready = torch.cuda.Event()
    ready.record()
    side.wait_event(ready)
    with torch.cuda.stream(side):
        for run in t.held_wgrads:
            run()
    t.held_wgrads = []
    t.used_side_stream = True


def _wait_chained_wgrad(t: Tape) -> None:
    """RV3D_OVERLAP=chain: a big weight gradient is running on the side stream -- the main stream's next backward-data launch starts
    after it (two persistent MFMA kernels only take CUs from each other); the bandwidth-bound passes in between do not wait."""
    if t.chained_wgrad is not None:
        torch.cuda.current_stream().wait_event(t.chained_wgrad)
        t.chained_wgrad = None


POS_BWD_FUSE = True  # (module attribute: tests and A/B tools flip it in-process; no environment switch)


def _pos_pair_backward(op: "E.ConvOp", t: Tape, dy2: Act) -> None:
    """Parameter gradients of the FIRST positional layer (conv 3 -> C, BatchNorm) from the second layer's output gradient."""
    sk = op.pos_first
    lay0, bn0, rel = sk.layer, sk.bn, sk.x
    l1 = op.layer
    cp, cin, pixels = sk.out.cp, lay0.c_in, sk.out.pixels
    dev = t.device
    ws = torch.empty(L.load().rv_bn_bwd_smallk_workspace_bytes(L.i64(pixels), L.i32(cp), L.i32(cin)), dtype=torch.uint8, device=dev)
    sums = torch.empty((2 + 4) * cp, dtype=torch.float64, device=dev)
    moms = torch.empty(4 + 16, dtype=torch.float64, device=dev)
    wp0 = lay0.packed("gather")
    call = lambda: L.call("rv_pos_backward_sums", L.i64(pixels), L.i32(cp), dy2.ptr(), L.ptr(l1.packed("scatter")), rel.ptr(), L.i32(rel.ld), L.i32(cin),
                          L.ptr(wp0), L.i32(E.pad32(cin)), L.ptr(sk.scale), L.ptr(sk.shift), L.ptr(sk.mean), L.ptr(sk.invstd), L.ptr(sums),
                          L.ptr(moms), L.ptr(ws), L.stream_ptr())
    if E.PROFILE is not None:
        E._launch("pos_bwd_kernel", 2.0 * pixels * cp * cp, call)
    else:
        call()
    dgamma = torch.empty(cp, dtype=torch.float32, device=dev)
    dbeta = torch.empty(cp, dtype=torch.float32, device=dev)
    dw = torch.empty((cp, cin), dtype=torch.float32, device=dev)
    g01 = None
    if sk.sync_world > 1:  # SyncBN: the normalisation needs the GLOBAL (sum g, sum g*xhat); the other sums stay this rank's (as _smallk_grads)
        g01 = _global_s01(sums, cp, pixels)
    L.call("rv_bn_bwd_smallk_from_sums", L.i32(cp), L.i32(cin), L.ptr(sums), L.ptr(moms), L.ptr(g01), L.ptr(wp0), L.i32(E.pad32(cin)),
           L.ptr(sk.gamma_p), L.ptr(sk.mean), L.ptr(sk.invstd), L.i64(sk.count), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dw), L.stream_ptr())
    c = bn0.num_features
    t.add_param_grad(bn0.weight, dgamma[:c])
    t.add_param_grad(bn0.bias, dbeta[:c])
    t.add_param_grad(lay0.weight, lay0.unpermute_grad(dw[: lay0.c_out].reshape(lay0.c_out, cin, 1, 1).contiguous()))
    sk.grads_done = True


def _global_s01(sums: Tensor, cp: int, pixels: int) -> Tensor:
    """All-reduced (sum g, sum g*xhat) of a small-K layer with this rank's pixel count in the slot behind them: the finalize
    kernel reads the GLOBAL count there (``count = -1``), so no value comes back to the host."""
    g01 = torch.empty(2 * cp + 1, dtype=torch.float64, device=sums.device)
    g01[: 2 * cp] = sums[: 2 * cp]
    g01[2 * cp : 2 * cp + 1].fill_(float(pixels))  # (a kernel argument; `g01[i] = python_float` is a pageable host-to-device copy that blocks the
                                        #  host until the stream has drained -- 25 ms twice per step, profiles/r04_syncbn_collectives.md)
    E.COLLECTIVES.add(g01)
    E.all_reduce_(g01)
    return g01


def _smallk_grads(t: Tape, pixels: int, cp: int, dout: Act, mask: Optional[Act], y: Optional[Act], scale, shift, mean, invstd, flags: int, v: Act,
                  lay, gamma_p, stat_mean, stat_invstd, count: int, sync: bool):
    """(dgamma, dbeta, dW) of a small-K conv + BatchNorm from one pass (``rv_bn_bwd_smallk``); under SyncBN the two-phase
    form with the all-reduce of (sum g, sum g*xhat) in between."""
    cin = lay.c_in
    dev = t.device
    ws = torch.empty(L.load().rv_bn_bwd_smallk_workspace_bytes(L.i64(pixels), L.i32(cp), L.i32(cin)), dtype=torch.uint8, device=dev)
    dgamma = torch.empty(cp, dtype=torch.float32, device=dev)
    dbeta = torch.empty(cp, dtype=torch.float32, device=dev)
    dw = torch.empty((cp, cin), dtype=torch.float32, device=dev)
    wp = lay.packed("gather")
    head = (L.i64(pixels), L.i32(cp), dout.ptr(), L.i32(dout.ld), mask.ptr() if mask is not None else None,
            L.i32(mask.ld if mask is not None else 0), y.ptr() if y is not None else None, L.i32(y.ld if y is not None else 0), L.ptr(scale),
            L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.i32(flags), v.ptr(), L.i32(v.ld), L.i32(cin), L.ptr(wp), L.i32(E.pad32(cin)))
    if sync:
        cin_pad = 4 if cin <= 4 else 8
        sums = torch.empty((2 + cin_pad) * cp, dtype=torch.float64, device=dev)
        moms = torch.empty(cin_pad + cin_pad * cin_pad, dtype=torch.float64, device=dev)
        L.call("rv_bn_bwd_smallk_sums", *head, L.ptr(sums), L.ptr(moms), L.ptr(ws), L.stream_ptr())
        g01 = _global_s01(sums, cp, pixels)  # SyncBN: global (sum g, sum g*xhat, pixel count); the other sums stay this rank's
        L.call("rv_bn_bwd_smallk_from_sums", L.i32(cp), L.i32(cin), L.ptr(sums), L.ptr(moms), L.ptr(g01), L.ptr(wp), L.i32(E.pad32(cin)),
               L.ptr(gamma_p), L.ptr(stat_mean), L.ptr(stat_invstd), L.i64(-1), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dw), L.stream_ptr())
    else:
        L.call("rv_bn_bwd_smallk", *head, L.ptr(gamma_p), L.ptr(stat_mean), L.ptr(stat_invstd), L.i64(count),
               L.ptr(dgamma), L.ptr(dbeta), L.ptr(dw), L.ptr(ws), L.stream_ptr())
    return dgamma, dbeta, dw


SMALLK_FUSION = True


def bn_backward(op: "E.BnOp", t: Tape) -> None:
    rec = bn_backward_begin(op, t)
    if rec is not None:
        bn_backward_finish([rec], t)


def bn_backward_begin(op: "E.BnOp", t: Tape):
    """First half of a BatchNorm backward: this rank's (sum g, sum g*xhat) rows (from the backward-data epilogue that wrote the
    gradient, or a reduce pass).  Returns a record for ``bn_backward_finish``, or None when the op was completed here (no
    gradient arrived; the MetaKernel / small-K fused forms, which carry their own collectives)."""
    lazy = op.lazy
    meta = t.meta_in.pop(id(lazy), None)
    if meta is not None:
        assert id(lazy) not in t.lazy_in, "the positional Lazy behind the MetaKernel modulation has exactly one consumer"
        _bn_backward_meta(op, t, meta)
        return None
    hf = t.head_final.pop(id(lazy), None)
    if hf is not None:
        assert id(lazy) not in t.lazy_in, "the last unit of a head tower feeds the final conv only"
        head, partial, rows, dY, wp = hf  # (the sums were formed with the final conv's weight gradient: _head_final_sums)
        return (op, None, partial, rows, lazy.raw.pixels, 0, None, ("head_final", head, dY, wp))
    entry = t.lazy_in.pop(id(lazy), None)
    if entry is None:
        return None
    dout, mask, res = entry if len(entry) == 3 else (entry[0], entry[1], None)
    st, raw = lazy.bn, lazy.raw
    if st.mean is None:
        raise L.RvError("BatchNorm backward needs batch statistics (module was run in eval mode)")
    cp, pixels = raw.cp, raw.pixels
    flags = L.BNB_RELU_Z if lazy.relu else 0
    conv = op.conv
    lay, geo = conv.layer, conv.layer.geom
    if (SMALLK_FUSION and res is None and not conv.need_input_grad and not conv.out_f32 and not isinstance(conv.x, Lazy)
            and geo.kh == 1 and geo.kw == 1 and geo.stride_w == 1 and lay.fwd_form == "gather" and lay.c_in <= 8 and lay.in_perm is None
            and lay.bias is None):
        # 1x1 conv with a handful of input channels and no input gradient (stem): BatchNorm backward and the conv's
        # weight gradient from one pass over (dOut, y, input) -- neither dy nor a separate wgrad pass (rv_bn_bwd_smallk)
        dgamma, dbeta, dw = _smallk_grads(t, pixels, cp, dout, mask, raw, st.scale, st.shift, st.mean, st.invstd, flags, conv.x, lay,
                                          op.gamma_p, st.mean, st.invstd, st.count, op.sync_world > 1)
        c, cin = st.module.num_features, lay.c_in
        t.add_param_grad(st.module.weight, dgamma[:c])
        t.add_param_grad(st.module.bias, dbeta[:c])
        t.add_param_grad(lay.weight, lay.unpermute_grad(dw[: lay.c_out].reshape(lay.c_out, cin, 1, 1).contiguous()))
        return None
    common = (L.i64(pixels), L.i32(cp), dout.ptr(), L.i32(dout.ld), mask.ptr() if mask is not None else None,
              L.i32(mask.ld if mask is not None else 0), raw.ptr(), L.i32(raw.ld), L.ptr(st.scale), L.ptr(st.shift),
              L.ptr(st.mean), L.ptr(st.invstd))
    sums = t.lazy_sums.pop(id(lazy), None)
    if sums is not None and sums[2] is dout and ((len(sums) == 3 and mask is None and res is None) or (len(sums) == 4 and mask is not None)):
        partial, rows = sums[0], sums[1]  # formed by the backward-data launch that wrote dout (rv_tap_data_grad_bnb)
    else:
        rows = L.load().rv_bn_bwd_rows(L.i64(pixels))
        partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, cp), dtype=torch.float32, device=t.device)
        L.call("rv_bn_bwd_reduce", *common, L.i32(flags), L.ptr(partial), L.stream_ptr())
    return (op, common, partial, rows, pixels, flags, res, (dout, mask))  # (dout / mask referenced until the apply pass has been issued)


def bn_backward_finish(recs, t: Tape) -> None:
    """Second half for a run of independent BatchNorm layers: ONE all-reduce of all their sums under SyncBN, then per layer
    the coefficients and the apply pass (gradient w.r.t. the raw conv output)."""
    glob = [None] * len(recs)
    local = [None] * len(recs)
    sync = [i for i, r in enumerate(recs) if r[0].sync_world > 1]
    if sync:
        for i in sync:  # this rank's own (sum g, sum g*xhat) = (dbeta, dgamma): written by the same launch that fills the all-reduce buffer
            local[i] = torch.empty((2, recs[i][0].lazy.raw.cp), dtype=torch.float32, device=t.device)
        views = E.allreduce_partial_rows_many([(recs[i][2], recs[i][3], recs[i][4], local[i]) for i in sync])
        for i, v in zip(sync, views):
            glob[i] = v
        _release_held_wgrads(t)  # (the collective is enqueued and the main stream waits for it: the held weight gradients may take the CUs now)
    for (op, common, partial, rows, pixels, flags, res, _keep), g, loc in zip(recs, glob, local):
        lazy = op.lazy
        st, raw = lazy.bn, lazy.raw
        dgamma, dbeta, coef = _bn_finalize(op, t, partial, rows, pixels, glob=g, local=loc)
        dy = raw.like()
        if common is None:  # the head-final form: dA recomputed from the final conv's output gradient (bn_backward_begin)
            L.call("rv_head_final_bwd_apply", *_keep[1], L.ptr(coef), dy.ptr(), L.i32(dy.ld), L.stream_ptr())
        elif res is not None:
            rg, racc = res
            L.call("rv_bn_bwd_apply", *common, L.ptr(coef), L.i32(flags | (L.BNB_RES_ACCUM if racc else 0)), dy.ptr(), L.i32(dy.ld),
                   rg.ptr(), L.i32(rg.ld), L.stream_ptr())
        else:
            L.call("rv_bn_bwd_apply", *common, L.ptr(coef), L.i32(flags), dy.ptr(), L.i32(dy.ld), None, L.i32(0), L.stream_ptr())
        t.raw_grad[id(raw)] = dy
        c = st.module.num_features
        t.add_param_grad(st.module.weight, dgamma[:c])
        t.add_param_grad(st.module.bias, dbeta[:c])


def _bn_finalize(op: "E.BnOp", t: Tape, partial: Tensor, rows: int, pixels: int, glob: Optional[Tensor] = None, local: Optional[Tensor] = None):
    """(dgamma, dbeta, coef) from the partial (sum g, sum g*xhat) rows; SyncBN: the coefficients from the all-reduced totals
    (``glob`` / ``local``: already reduced in a group, else reduced here)."""
    st, cp = op.lazy.bn, op.lazy.raw.cp
    coef = torch.empty((3, cp), dtype=torch.float32, device=t.device)
    if op.sync_world > 1:
        # SyncBN backward: the normalisation coefficients need the GLOBAL (sum g, sum g*xhat) and count -> one RCCL
        # all-reduce; dgamma / dbeta stay LOCAL sums (DDP averages parameter gradients over ranks, as under torch
        # SyncBatchNorm): they are this rank's totals, copied out by the launch that filled the all-reduce buffer
        if glob is None:
            local = torch.empty((2, cp), dtype=torch.float32, device=t.device)
            glob = E.allreduce_partial_rows(partial, rows, pixels, local)
            _release_held_wgrads(t)
        L.call("rv_bn_bwd_finalize", L.ptr(glob), L.i32(1), L.i32(cp), L.i64(-1), L.ptr(op.gamma_p), L.ptr(st.invstd),
               None, None, L.i32(0), L.ptr(coef), L.stream_ptr())
        return local[1], local[0], coef
    dgamma = torch.empty(cp, dtype=torch.float32, device=t.device)
    dbeta = torch.empty(cp, dtype=torch.float32, device=t.device)
    L.call("rv_bn_bwd_finalize", L.ptr(partial), L.i32(rows), L.i32(cp), L.i64(st.count), L.ptr(op.gamma_p), L.ptr(st.invstd),
           L.ptr(dgamma), L.ptr(dbeta), L.i32(0), L.ptr(coef), L.stream_ptr())
    return dgamma, dbeta, coef


def _bn_backward_meta(op: "E.BnOp", t: Tape, meta) -> None:
    """BatchNorm backward of the positional layer behind the MetaKernel modulation (sums formed by modulate_backward)."""
    dgeo, feat, partial, rows = meta
    lazy = op.lazy
    st, raw = lazy.bn, lazy.raw
    dgamma, dbeta, coef = _bn_finalize(op, t, partial, rows, raw.pixels)
    dy = raw.like()
    L.call("rv_meta_modulate_bwd_apply", dgeo.ptr(), raw.ptr(), L.ptr(st.scale), L.ptr(st.shift), L.ptr(st.mean), L.ptr(st.invstd),
           L.ptr(coef), feat.ptr(), L.i32(feat.ld), L.i32(feat.N), L.i32(feat.H), L.i32(feat.W), L.i32(feat.cp), dy.ptr(), L.stream_ptr())
    t.raw_grad[id(raw)] = dy
    c = st.module.num_features
    t.add_param_grad(st.module.weight, dgamma[:c])
    t.add_param_grad(st.module.bias, dbeta[:c])


def combine_backward(op: "E.CombineOp", t: Tape) -> None:
    gout, have = t.grad_buffer(op.out)
    if not have:
        return
    mask = op.out if op.relu_out else None
    lazies = [x for x in (op.a, op.b) if isinstance(x, Lazy)]
    plains = [x for x in (op.a, op.b) if x is not None and not isinstance(x, Lazy)]
    # out = relu(bn(y) + x): the residual gradient dOut*[out>0] is exactly the `g` the BatchNorm-backward apply pass
    # already forms, so that pass also writes it (rv_bn_bwd_apply's dres output) instead of a separate masking pass.
    fuse_res = len(lazies) == 1 and len(plains) == 1 and not lazies[0].relu and t.training
    if (BNB_PAIR and len(lazies) == 2 and not plains and mask is not None and t.training and all(not x.relu and x.bn.mean is not None for x in lazies)
            and all(id(x) not in t.lazy_in for x in lazies) and lazies[0].raw.cp == lazies[1].raw.cp == gout.cp
            and lazies[0].raw.pixels == lazies[1].raw.pixels == gout.pixels):
        # out = relu(bn_a(ya) + bn_b(yb)) (a block with a projection): the sums of BOTH BatchNorm backwards from one pass
        # over (dOut, out, ya, yb); each bn_backward_begin then finds its rows and skips its own reduce pass
        la, lb = lazies
        rows = L.load().rv_bn_bwd_rows(L.i64(gout.pixels))
        pa = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, gout.cp), dtype=torch.float32, device=t.device)
        pb = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, gout.cp), dtype=torch.float32, device=t.device)
        L.call("rv_bn_bwd_reduce_pair", L.i64(gout.pixels), L.i32(gout.cp), gout.ptr(), L.i32(gout.ld), mask.ptr(), L.i32(mask.ld),
               la.raw.ptr(), L.i32(la.raw.ld), L.ptr(la.bn.mean), L.ptr(la.bn.invstd), lb.raw.ptr(), L.i32(lb.raw.ld), L.ptr(lb.bn.mean),
               L.ptr(lb.bn.invstd), L.ptr(pa), L.ptr(pb), L.stream_ptr())
        ops = [t.bn_of.get(id(x)) for x in lazies]
        if all(o is not None for o in ops) and (all(o.sync_world == 1 for o in ops) or (E.GROUP_SYNC_BN and all(o.sync_world > 1 for o in ops))):
            # ... and both apply passes as one: the gradients w.r.t. both raw conv outputs leave now; the two BatchNorm ops find
            # nothing pending when the tape reaches them.  SyncBN: the sums of both layers travel in ONE all-reduce first.
            globs, locs = [None, None], [None, None]
            if ops[0].sync_world > 1:
                locs = [torch.empty((2, gout.cp), dtype=torch.float32, device=t.device) for _ in ops]
                globs = E.allreduce_partial_rows_many([(pa, rows, gout.pixels, locs[0]), (pb, rows, gout.pixels, locs[1])])
                _release_held_wgrads(t)
            coefs = []
            for o, part, gl, lo in zip(ops, (pa, pb), globs, locs):
                dgamma, dbeta, coef = _bn_finalize(o, t, part, rows, gout.pixels, glob=gl, local=lo)
                c = o.lazy.bn.module.num_features
                t.add_param_grad(o.lazy.bn.module.weight, dgamma[:c])
                t.add_param_grad(o.lazy.bn.module.bias, dbeta[:c])
                coefs.append(coef)
            dya, dyb = la.raw.like(), lb.raw.like()
            L.call("rv_bn_bwd_apply_pair", L.i64(gout.pixels), L.i32(gout.cp), gout.ptr(), L.i32(gout.ld), mask.ptr(), L.i32(mask.ld),
                   la.raw.ptr(), L.i32(la.raw.ld), L.ptr(la.bn.mean), L.ptr(la.bn.invstd), L.ptr(coefs[0]), dya.ptr(), L.i32(dya.ld),
                   lb.raw.ptr(), L.i32(lb.raw.ld), L.ptr(lb.bn.mean), L.ptr(lb.bn.invstd), L.ptr(coefs[1]), dyb.ptr(), L.i32(dyb.ld), L.stream_ptr())
            t.raw_grad[id(la.raw)] = dya
            t.raw_grad[id(lb.raw)] = dyb
            return
        for x, part in ((la, pa), (lb, pb)):
            t.add_lazy_grad(x, gout, mask, None)
            t.lazy_sums[id(x)] = (part, rows, gout, True)
        return
    for x in lazies:
        res = None
        if fuse_res:
            g, have_x = t.grad_buffer(plains[0])
            res = (g, have_x)
            t.mark_written(plains[0])
        t.add_lazy_grad(x, gout, mask, res)
    if fuse_res:
        return
    for x in plains:
        g, have_x = t.grad_buffer(x)
        L.call("rv_ew_mask_grad", L.i64(gout.pixels), L.i32(gout.cp), gout.ptr(), L.i32(gout.ld),
               mask.ptr() if mask is not None else None, L.i32(mask.ld if mask is not None else 0), g.ptr(), L.i32(g.ld),
               L.i32(1 if have_x else 0), L.stream_ptr())
        t.mark_written(x)


BNB_PAIR = os.environ.get("RV3D_NO_BNB_PAIR") is None  # (environment switch kept for tests/test_gpu_ddp.py: isolates the SyncBN grouping from the one-pass pair apply)
META_BWD_FUSE = True


def modulate_backward(op: "E.MetaModulateOp", t: Tape) -> None:
    pos, feat = op.pos, op.feat
    st = pos.bn
    g, have = t.grad_buffer(op.out)
    if not have:
        return
    gf, have_f = t.grad_buffer(feat)
    assert not have_f, "MetaKernel projection output has a single consumer"
    if META_BWD_FUSE and pos.relu and st.mean is not None and id(pos) not in t.lazy_in:
        # fused with the BatchNorm(+ReLU) backward of the positional layer: this pass forms dfeat and the (sum z, sum z*xhat)
        # rows; bn_backward finalizes them and a second pass writes dy -- the activated-gradient tensor is never written
        rows = L.load().rv_meta_bwd_rows(L.i32(feat.N), L.i32(feat.H), L.i32(feat.W))
        partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, feat.cp), dtype=torch.float32, device=t.device)
        L.call("rv_meta_modulate_bwd_sums", g.ptr(), pos.raw.ptr(), L.ptr(st.scale), L.ptr(st.shift), L.ptr(st.mean), L.ptr(st.invstd),
               feat.ptr(), L.i32(feat.ld), L.i32(feat.N), L.i32(feat.H), L.i32(feat.W), L.i32(feat.cp), gf.ptr(), L.i32(gf.ld),
               L.ptr(partial), L.stream_ptr())
        t.mark_written(feat)
        t.meta_in[id(pos)] = (g, feat, partial, rows)
        return
    dpos = pos.raw.like()
    L.call("rv_meta_modulate_bwd", g.ptr(), pos.raw.ptr(), L.ptr(pos.bn.scale), L.ptr(pos.bn.shift), feat.ptr(), L.i32(feat.ld),
           L.i32(feat.N), L.i32(feat.H), L.i32(feat.W), L.i32(feat.cp), dpos.ptr(), gf.ptr(), L.i32(gf.ld), L.stream_ptr())
    t.mark_written(feat)
    dpos._rv_owned = True
    t.add_lazy_grad(pos, dpos, None)


def smallk_backward(op: "E.SmallKOp", t: Tape) -> None:
    """Backward of ``h = relu(bn(W x))``: the raw output is recomputed from the <= 8 input channels inside the one pass over
    dOut (RV_BNB_Y_FROM_INPUT) -- the stored activation is not read back."""
    if op.grads_done:  # the consumer's backward formed these gradients without materialising dOut (_pos_pair_backward)
        return
    dout, have = t.grad_buffer(op.out)
    if not have:
        return
    if not t.training:
        raise L.RvError("BatchNorm backward needs batch statistics (module was run in eval mode)")
    lay, bn, h, v = op.layer, op.bn, op.out, op.x
    cp, cin, pixels = h.cp, lay.c_in, h.pixels
    dgamma, dbeta, dw = _smallk_grads(t, pixels, cp, dout, None, None, op.scale, op.shift, op.mean, op.invstd, L.BNB_RELU_Z | L.BNB_Y_FROM_INPUT,
                                      v, lay, op.gamma_p, op.mean, op.invstd, op.count, op.sync_world > 1)
    c = bn.num_features
    t.add_param_grad(bn.weight, dgamma[:c])
    t.add_param_grad(bn.bias, dbeta[:c])
    t.add_param_grad(lay.weight, lay.unpermute_grad(dw[: lay.c_out].reshape(lay.c_out, cin, 1, 1).contiguous()))
