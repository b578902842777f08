"""Teacher-forced per-layer parity at the REAL channel widths: every conv (+ BatchNorm) unit of the rv-av2 and rv-waymo models.

The composed-model tests (test_gpu_realwidth.py) bound the END of ~60 bf16 layers; a wrong epilogue in one of the ~160 layer
launches could hide inside those margins.  Here the oracle's bf16 run (``oracle.model.Numerics.bf16`` with ``trace``) records,
for every conv unit, the input it received, its raw output and -- after ``loss.backward()`` -- the gradient w.r.t. that
output.  Each HIP layer launch is then fed the ORACLE's tensors (not its own predecessor's output) and compared with fp32
torch ops on the same bf16-rounded operands, so a failure names the layer:

* forward   raw conv output (bf16 store)           8e-3 of the tensor maximum (= two bf16 ulps at the maximum; the fp32
            accumulation order differs, the stored value may round the other way);
* BatchNorm batch mean / 1/sqrt(var + eps) from the fp32 accumulators   2e-5 relative to (|mean| + std) / to invstd
  (measured 6e-8 / 2.4e-6);
* backward-data (bf16 store)                       8e-3 of the maximum;
* weight gradient (fp32, split-K)                  1e-4 of the maximum (measured 7e-6).

Measured worst cases (MI355X, round 3): forward 5.3e-3, backward-data 4.0e-3 over the 80 units of either model.

The production kernels are forced on the small crop (``tapconv4_min_blocks`` = 1) and asserted from the launch records.
"""

from __future__ import annotations

import pytest
import torch
import torch.nn.functional as F

from test_gpu_forward import DEV, rel_err
from test_gpu_realwidth import _small_grids, _prepare

pytestmark = pytest.mark.gpu


def _ref_unit(rec, w):
    """fp32 torch ops on the bf16-rounded operands of one traced unit: y, and (dx, dw) for the traced dy."""
    from oracle import model as om

    xb = om.round_bf16(rec["x"]).requires_grad_(True)
    wb = om.round_bf16(w).requires_grad_(True)
    if rec["kind"] == "convT":
        y = F.conv_transpose2d(xb, wb, stride=rec["stride"], padding=rec["padding"])
    elif rec["kind"] == "conv1x1":
        y = F.conv2d(xb, wb)
    else:
        y = om.conv2d_same(xb, wb, rec["stride"])
    dx = dw = None
    if "dy" in rec:
        dx, dw = torch.autograd.grad(y, (xb, wb), om.round_bf16(rec["dy"]))
    return y.detach(), dx, dw


@pytest.mark.parametrize("widths,n_feat,n_cls,W", [("rv-av2", 5, 26, 256), ("rv-waymo", 6, 3, 336)])
def test_every_layer_teacher_forced(widths, n_feat, n_cls, W):
    from bench import Detector
    from oracle import model as om
    from oracle import targets as otgt
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd import engine_bwd

    backbone, head, sd, batch = _prepare(widths, n_feat, n_cls, W, 1.0)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    nm = om.Numerics.bf16(train=True)
    nm.trace = []
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    _, logits, reg = om.detector_forward(batch["features"], batch["cart"], {**sd, **params}, nm=nm)
    tg = otgt.compute_targets(batch["cart"], batch["annotations"], n_cls)
    otgt.detection_loss(logits, reg, batch["cart"], batch["mask"], tg, n_cls)["loss"].backward()
    trace = nm.trace
    assert len(trace) > 70 and all("dy" in r for r in trace)

    model = Detector(backbone, head).to(DEV).train()
    modules = dict(model.named_modules())
    C0 = modules["backbone.stem"].out_channels
    ran, worst = set(), {"fwd": (0.0, ""), "dgrad": (0.0, ""), "wgrad": (0.0, ""), "mean": (0.0, ""), "invstd": (0.0, "")}

    def note(kind, err, name):
        if err > worst[kind][0]:
            worst[kind] = (err, name)

    with _small_grids():
        E.PROFILE = E.KernelProfile()
        try:
            for rec in trace:
                name = rec["w"]
                mod = modules[name[: -len(".weight")]]
                x, kw = rec["x"], {}
                if name.endswith("stem.fusion_kernel.0.0.weight"):  # reference channel order c*9+k -> the engine's k*C+c (engine.TapLayer.in_perm)
                    b_, ck, h_, w_ = x.shape
                    x = x.view(b_, C0, 9, h_, w_).permute(0, 2, 1, 3, 4).reshape(b_, ck, h_, w_)
                    kw = {"in_perm": (C0, 9)}
                layer = E.tap_layer(mod, **kw)
                y_ref, dx_ref, dw_ref = _ref_unit(rec, sd[name])
                t = E.Tape(True, DEV)
                x_act = E.Act.from_nchw(om.round_bf16(x).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
                first = name.startswith("backbone.stem.projection") and x.shape[1] == n_feat or "positional_kernel.0" in name
                final = rec["bn"] is None
                conv = E.ConvOp(t, layer, x_act, stats=not final, out_f32=final, need_input_grad=not first)
                if final:
                    y = conv.out_t[..., : layer.c_out].permute(0, 3, 1, 2).float().cpu()
                    y_cmp = y_ref + sd[rec["bias"]].view(1, -1, 1, 1)
                    note("fwd", rel_err(y, y_cmp), name)
                    assert rel_err(y, y_cmp) < 1e-4, (name, rel_err(y, y_cmp))  # fp32 output of bf16 operands
                else:
                    y = conv.out.nchw().float().cpu()
                    e = rel_err(y, om.round_bf16(y_ref))
                    note("fwd", e, name)
                    assert e < 8e-3, (name, "forward", e)
                    bnop = E.BnOp(t, conv, modules[rec["bn"]], relu=True)
                    yd = y_ref.double()
                    dims = (0, 2, 3)
                    mean, var = yd.mean(dim=dims), yd.var(dim=dims, unbiased=False)
                    c = mean.shape[0]
                    em = float(((bnop.state.mean[:c].cpu().double() - mean).abs() / (mean.abs() + var.sqrt() + 1e-12)).max())
                    ei = float(((bnop.state.invstd[:c].cpu().double() - (var + 1e-5).rsqrt()).abs() * (var + 1e-5).sqrt()).max())
                    note("mean", em, name)
                    note("invstd", ei, name)
                    assert em < 2e-5 and ei < 2e-5, (name, "batch statistics", em, ei)
                # backward of the conv: the oracle's dy in, this layer's dx / dW out
                dy = om.round_bf16(rec["dy"]).to(DEV)
                if final:
                    engine_bwd.seed_f32_output_grad(t, conv, dy)
                else:
                    t.raw_grad[id(conv.out)] = E.Act.from_nchw(dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last))
                engine_bwd.conv_backward(conv, t)
                if t.used_side_stream:
                    torch.cuda.current_stream().wait_stream(E.side_stream(DEV))
                dw = t.param_grads[id(layer.weight)].float().cpu()
                e = rel_err(dw, dw_ref)
                note("wgrad", e, name)
                assert e < 1e-4, (name, "weight gradient", e)
                if not first:
                    dx = t.grads[id(x_act)].nchw().float().cpu()
                    ref = dx_ref
                    if kw:  # back to the engine's channel order
                        b_, ck, h_, w_ = ref.shape
                        ref = ref.view(b_, C0, 9, h_, w_).permute(0, 2, 1, 3, 4).reshape(b_, ck, h_, w_)
                    e = rel_err(dx, om.round_bf16(ref))
                    note("dgrad", e, name)
                    assert e < 8e-3, (name, "backward-data", e)
            torch.cuda.synchronize()
            ran = set(n for n, *_ in E.PROFILE.records)
        finally:
            E.PROFILE = None
    print(f"[{widths}] {len(trace)} units; worst " + "  ".join(f"{k} {v[0]:.2e} ({v[1].split('.weight')[0][-40:]})" for k, v in worst.items()))
    need = {"tapconv4_kernel<128>", "wgrad3_kernel(+reduce)"}
    assert need <= ran and any(n.startswith(("tapconv5_kernel<", "tapconv6_kernel<")) for n in ran), (need - ran, sorted(ran))


def test_rv_waymo_full_size_eval_forward_vs_oracle():
    """rv-waymo at its stated size (64 x 2656 x 6, [128]*5, towers 256, 3 classes), eval mode: logits / regressands against the
    oracle with the same bounds as the rv-av2 full-size test (test_gpu_realwidth.py::test_full_size_eval_forward_vs_oracle)."""
    from oracle import model as om
    from range_view_3d_detection_amd import engine as E
    from test_gpu_realwidth import _check_direction, _check_forward

    backbone, head, sd, batch = _prepare("rv-waymo", 6, 3, 2656, 0.5)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        _, lg16, rg16 = om.detector_forward(batch["features"], batch["cart"], sd, nm=om.Numerics.bf16(train=False))
        _, lg32, rg32 = om.detector_forward(batch["features"], batch["cart"], sd, nm=om.Numerics(train=False))
    backbone, head = backbone.to(DEV).eval(), head.to(DEV).eval()
    data = {k: (v.to(DEV) if k != "annotations" else v) for k, v in batch.items()}
    E.PROFILE = E.KernelProfile()
    try:
        with torch.no_grad():
            outputs, _ = head(backbone(data), data, return_loss=False)
        torch.cuda.synchronize()
        ran = set(name for name, *_ in E.PROFILE.records)
    finally:
        E.PROFILE = None
    assert "tapconv6_kernel<128>" in ran, sorted(ran)  # (full size: every 3x3 layer has at least one round of 512 x 128 tiles)
    logits, reg = outputs[1][0]["logits"].float().cpu(), outputs[1][0]["regressands"].float().cpu()
    m = {"logits~bf16": rel_err(logits, lg16), "logits~fp32": rel_err(logits, lg32), "emu~fp32": rel_err(lg16, lg32),
         "reg~bf16": rel_err(reg, rg16), "reg~fp32": rel_err(reg, rg32), "reg emu~fp32": rel_err(rg16, rg32)}
    print("[rv-waymo eval 1x64x2656] " + "  ".join(f"{k} {v:.3e}" for k, v in m.items()))
    _check_forward(m)
    _check_direction(logits, lg16, lg32, reg, rg16, rg32)
