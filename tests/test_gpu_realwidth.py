"""Composed-model parity at the REAL channel widths (the programs bench.py runs), against the pinned oracle.

* rv-av2 widths  (layers [256,128,128,128,128], towers 512, 26 classes, 5 features) on a 1 x 64 x 256 crop;
* rv-waymo widths ([128]*5, towers 256, 3 classes, 6 features) on a 1 x 64 x 336 crop (width not a multiple of 64);
* one full-size (1 x 64 x 2048) eval forward of the rv-av2 model.

On crops this small the library's speed heuristic would pick the register-staged kernels (too few tiles to fill 256 CUs), so
the tests lift it per call (``RV_SEL_SMALL_GRIDS`` in the shape flags, ``_lib.select``) and ASSERT from the launch records that the production
kernels -- ``tapconv4_kernel<256>``, ``tapconv4_kernel<128>`` and ``wgrad3_kernel`` -- are what ran.

Tolerances (stated next to the asserts) are relative to the tensor maximum unless noted:
forward vs the oracle with the same bf16 storage points (``Numerics.bf16``), forward vs the fp32 oracle (= the reference's
CPU path, pinned by tests/golden), loss relative error, per-parameter gradient cosine against the fp32 oracle with the CPU
bf16 emulation as the yardstick of what bf16 storage costs (see test_gpu_model.py::test_detector_gradients_vs_oracle).
"""

from __future__ import annotations

import ctypes

import numpy as np
import pytest
import torch

from test_gpu_backward import _cos
from test_gpu_forward import DEV, rel_err

pytestmark = pytest.mark.gpu


def _small_grids(on: bool = True):
    """Per-call selection hint (include/rv3d.h RV_SEL_SMALL_GRIDS): generations 4 / 5 also take grids below one round of CUs."""
    from range_view_3d_detection_amd import _lib as L

    return L.select(L.SEL_SMALL_GRIDS if on else 0)


def _check_forward(m):
    for got, emu in (("logits", "emu~fp32"), ("reg", "reg emu~fp32")):
        assert m[f"{got}~bf16"] < max(3e-2, 1.5 * m[emu] + 1e-2), m
        assert m[f"{got}~fp32"] < 1.5 * m[emu] + 1e-2, m


def _check_direction(logits, lg16, lg32, reg, rg16, rg32):
    """Cosine against the fp32 oracle: > 0.99, and no worse than the CPU bf16 emulation's by more than 2e-3.  ``lg16`` / ``rg16``:
    the emulation's tensors, or its RECORDED cosines (floats: the full-size tests' yardstick, tests/tools/emulation_yardstick.py)."""
    for name, got, emu, ref in (("logits", logits, lg16, lg32), ("regressands", reg, rg16, rg32)):
        c, c_emu = _cos(got, ref), (emu if isinstance(emu, float) else _cos(emu, ref))
        print(f"    {name}: cosine vs fp32 oracle {c:.5f} (CPU bf16 emulation {c_emu:.5f})")
        assert c > 0.99 and c > c_emu - 2e-3, (name, c, c_emu)


def _envelope(key):
    """Recorded envelope of the CPU bf16 emulation's gradient cosines for a crop case (tests/golden/emulation_envelope.json), or None."""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "emulation_envelope.json")
    with open(path) as f:
        return json.load(f)["cases"].get(key)


def _prepare(widths, n_feat, n_cls, W, bn_bias_shift, B=1, H=64, boxes=12):
    from bench import build_model, synthetic_batch

    torch.manual_seed(0)
    backbone, head = build_model(widths, n_cls, n_feat)
    gen = torch.Generator().manual_seed(1)
    for m in list(backbone.modules()) + list(head.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=gen)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=gen) + bn_bias_shift
            m.running_mean.data = 0.1 * torch.randn(m.running_mean.shape, generator=gen)
            m.running_var.data = 0.5 + torch.rand(m.running_var.shape, generator=gen)
    sd = {**{f"backbone.{k}": v.clone() for k, v in backbone.state_dict().items()}, **{f"head.{k}": v.clone() for k, v in head.state_dict().items()}}
    batch = synthetic_batch(B, H, W, seed=3, device="cpu", n_feat=n_feat, boxes_per_sweep=boxes, n_cls=n_cls)
    return backbone, head, sd, batch


@pytest.mark.parametrize("widths,n_feat,n_cls,W,bn_bias_shift", [("rv-av2", 5, 26, 256, 3.0), ("rv-av2", 5, 26, 256, 0.0),
                                                                 ("rv-waymo", 6, 3, 336, 3.0)])
def test_real_width_train_step_vs_oracle(widths, n_feat, n_cls, W, bn_bias_shift):
    _train_step_vs_oracle(widths, n_feat, n_cls, W, bn_bias_shift, small_grids=True)


def _train_step_vs_oracle(widths, n_feat, n_cls, W, bn_bias_shift, small_grids, emulation=None):
    """One training step (forward, targets, loss, backward) of the composed model against the oracle in fp32 and with bf16
    storage points.  small_grids: crops -- lift the library's tile-count heuristic so that the production kernels run; full-size
    images (tests/test_gpu_fullsize_train.py) take the library's own selection.  ``emulation``: the yardstick of the CPU bf16
    emulation RECORDED for this exact case (tests/tools/emulation_yardstick.py) instead of a second 60-80 s oracle pass -- every
    bound that is stated relative to the emulation then uses the recorded scalar; the element-wise comparison against the
    emulation's tensors is what the crop cases keep."""
    from bench import Detector
    from oracle import model as om
    from oracle import targets as otgt
    from range_view_3d_detection_amd import engine as E

    backbone, head, sd, batch = _prepare(widths, n_feat, n_cls, W, bn_bias_shift)
    torch.set_num_threads(min(32, torch.get_num_threads()))

    def oracle_run(nm):
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
        _, logits, reg = om.detector_forward(batch["features"], batch["cart"], {**sd, **params}, nm=nm)
        tg = otgt.compute_targets(batch["cart"], batch["annotations"], n_cls)
        loss = otgt.detection_loss(logits, reg, batch["cart"], batch["mask"], tg, n_cls)["loss"]
        loss.backward()
        return logits.detach(), reg.detach(), float(loss.detach()), {k: p.grad for k, p in params.items()}, tg

    lg32, rg32, loss32, g32, tg = oracle_run(om.Numerics(train=True))
    if emulation is None:
        lg16, rg16, loss16, g16, _ = oracle_run(om.Numerics.bf16(train=True))
    else:
        # the oracle's fp32 pass must be the one the yardstick was recorded against
        assert abs(loss32 - emulation["loss32"]) < 1e-4 * abs(loss32), (loss32, emulation["loss32"])
        lg16, rg16, loss16, g16 = emulation["cos_logits"], emulation["cos_reg"], emulation["loss16"], None

    model = Detector(backbone, head).to(DEV).train()
    data = {k: (v.to(DEV) if k != "annotations" else v) for k, v in batch.items()}
    with _small_grids(small_grids):
        E.PROFILE = E.KernelProfile()
        try:
            feats = model.backbone(data)
            outputs, losses = model.head(feats, data, return_loss=True)
            losses["loss"].backward()
            torch.cuda.synchronize()
            ran = set(name for name, *_ in E.PROFILE.records)
        finally:
            E.PROFILE = None
    # ---- the production kernels are what ran ----
    # (rv-waymo has no 256-channel pointwise conv: its only 256-channel layers are the 3x3 towers, i.e. tapconv5)
    need = {"tapconv4_kernel<128>", "wgrad3_kernel(+reduce)"} | ({"tapconv4_kernel<256>"} if widths == "rv-av2" else set())
    assert need <= ran and any(n.startswith(("tapconv5_kernel<", "tapconv6_kernel<")) for n in ran), (need - ran, sorted(ran))
    if not small_grids:
        assert "tapconv6_kernel<128>" in ran, sorted(ran)

    logits, reg = outputs[1][0]["logits"].float().cpu(), outputs[1][0]["regressands"].float().cpu()
    loss = float(losses["loss"].detach())
    if emulation is None:
        m = {"logits~bf16": rel_err(logits, lg16), "logits~fp32": rel_err(logits, lg32), "emu~fp32": rel_err(lg16, lg32),
             "reg~bf16": rel_err(reg, rg16), "reg~fp32": rel_err(reg, rg32), "reg emu~fp32": rel_err(rg16, rg32)}
    else:  # (no emulation tensors: the comparison against fp32 with the recorded yardstick)
        m = {"logits~bf16": 0.0, "logits~fp32": rel_err(logits, lg32), "emu~fp32": emulation["emu~fp32"],
             "reg~bf16": 0.0, "reg~fp32": rel_err(reg, rg32), "reg emu~fp32": emulation["reg emu~fp32"]}
    print(f"[{widths} shift {bn_bias_shift}] " + "  ".join(f"{k} {v:.3e}" for k, v in m.items()) +
          f"  loss {loss:.6f} / bf16-emu {loss16:.6f} / fp32 {loss32:.6f}")
    # targets are integer work: exact
    for k in ("classification_labels", "panoptics", "points_per_obj"):
        assert torch.equal(data[1][0][k].cpu(), tg[k]), k
    # forward, relative to the tensor maximum: what bf16 storage costs the CPU emulation of the same model against fp32
    # (emu~fp32) is the yardstick -- two bf16 realisations of the model (different summation order inside the fp32
    # accumulators, ReLU gates within one bf16 ulp of zero) differ from each other by about as much as each differs from
    # fp32.  Bounds: vs the bf16 emulation max(3e-2, 1.5 x yardstick + 1e-2); vs fp32 1.5 x yardstick + 1e-2.
    _check_forward(m)
    _check_direction(logits, lg16, lg32, reg, rg16, rg32)
    # loss: 1e-2 relative to the fp32 oracle (and no further from it than 2x the bf16 emulation + 2e-3)
    assert abs(loss - loss32) / abs(loss32) < 1e-2, (loss, loss32, loss16)
    assert abs(loss - loss32) <= 2.0 * abs(loss16 - loss32) + 2e-3 * abs(loss32), (loss, loss32, loss16)

    cos32, cos_emu, names = [], [], []
    for k, p in model.named_parameters():
        ref = g32[k]
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        if float(ref.norm()) < 1e-9:
            continue
        names.append(k)
        cos32.append(_cos(p.grad.cpu(), ref))
        if g16 is not None:
            cos_emu.append(_cos(g16[k], ref))
    cos32 = np.array(cos32)
    med32, q32 = float(np.median(cos32)), float(np.quantile(cos32, 0.05))
    if g16 is not None:
        cos_emu = np.array(cos_emu)
        med_emu, q_emu, min_emu = float(np.median(cos_emu)), float(np.quantile(cos_emu, 0.05)), float(cos_emu.min())
    else:
        med_emu, q_emu, min_emu = emulation["grad_cos_median"], emulation["grad_cos_q05"], emulation["grad_cos_min"]
    print(f"    {len(cos32)} parameters; gradient cosine vs fp32 oracle: HIP median {med32:.4f} q05 {q32:.4f} min {cos32.min():.4f};  "
          f"CPU bf16 emulation median {med_emu:.4f} q05 {q_emu:.4f} min {min_emu:.4f}")
    for i in np.argsort(cos32)[:4]:
        print(f"    worst: {names[i]:60s} HIP {cos32[i]:.4f}" + (f"  emulation {cos_emu[i]:.4f}" if g16 is not None else "") + f"  |g| {float(g32[names[i]].norm()):.2e}")
    # gradients: at least as close to the fp32 oracle as the CPU bf16 emulation (median - 0.02, 5 % quantile - 0.05);
    # in the well-conditioned regime (gates firmly open) additionally median > 0.99, 5 % quantile > 0.95.
    # The crop cases are held to a RECORDED ENVELOPE instead of to this run's one emulation (round-5 review, item 9): shift 0.0 is the chaotic
    # regime -- half of the ReLU gates sit within a bf16 ulp of zero and EVERY bf16 realisation of the model has a median cosine of only
    # ~0.61-0.64 against fp32; the emulation's own number moves by 0.03 with nothing but the order in which each conv sums its input
    # channels (tests/golden/emulation_envelope.json: eight orders, tests/tools/emulation_yardstick.py envelope -- median 0.6138 .. 0.6429, 5 %
    # quantile 0.4086 .. 0.4389), and three equally valid kernel selections of this library measured 0.6286 / 0.4088, 0.6199 / 0.4171
    # and 0.6025 / 0.3873 in rounds 3-5 (rounds 3-5 compared against ONE emulation and re-tuned the margin twice).  Bound: the envelope's
    # minimum - 0.02 (median) / - 0.03 (5 % quantile: the tenth-smallest of ~190 cosines is the noisier statistic); fixed numbers that no
    # kernel change moves.  The well-conditioned case, the full-size cases and the per-layer teacher-forced test are where an error shows.
    env = _envelope(f"{widths}/{W}/{bn_bias_shift}") if emulation is None else None
    if env is not None:
        assert abs(loss32 - env["loss32"]) < 1e-4 * abs(loss32), (loss32, env["loss32"])  # (the case the envelope was recorded for)
        print(f"    envelope of {len(env['realisations'])} CPU bf16 realisations: median {env['median_min']:.4f} .. {env['median_max']:.4f}, "
              f"q05 {env['q05_min']:.4f} .. {env['q05_max']:.4f}")
        assert med32 > env["median_min"] - 0.02 and q32 > env["q05_min"] - 0.03, (med32, q32, env["median_min"], env["q05_min"])
    else:
        assert med32 > med_emu - 0.02 and q32 > q_emu - 0.05, (med32, med_emu, q32, q_emu)
    if bn_bias_shift >= 3.0:
        assert med32 > 0.99 and q32 > 0.95, (med32, q32)


_EVAL_CASE = {}


def full_size_eval_case():
    """The rv-av2 model in eval mode on ONE full 64 x 2048 sweep and the fp32 oracle's outputs for it -- computed ONCE per test
    session and shared by this file's test and tests/test_gpu_fp16.py (an fp32 oracle forward of a whole sweep costs ~18 s of a
    16-core host).  The final classification bias is lifted to -1.5 so that the decoder of the fp16 test has boxes to work on."""
    if not _EVAL_CASE:
        from oracle import model as om

        backbone, head, sd, batch = _prepare("rv-av2", 5, 26, 2048, 0.5)
        head.classification_head["1"]["0"].blocks[-1][0].bias.data.fill_(-1.5)  # some scores above min_confidence
        sd["head.classification_head.1.0.blocks.4.0.bias"] = head.classification_head["1"]["0"].blocks[-1][0].bias.data.clone()
        torch.set_num_threads(min(32, torch.get_num_threads()))
        with torch.no_grad():
            _, lg32, rg32 = om.detector_forward(batch["features"], batch["cart"], sd, nm=om.Numerics(train=False))
        _EVAL_CASE.update(backbone=backbone, head=head, sd=sd, batch=batch, lg32=lg32, rg32=rg32)
    return _EVAL_CASE


def test_full_size_eval_forward_vs_oracle():
    """rv-av2 model, eval mode (running statistics), ONE full 64 x 2048 sweep: logits / regressands vs the oracle."""
    from oracle import model as om

    case = full_size_eval_case()
    backbone, head, sd, batch, lg32, rg32 = (case[k] for k in ("backbone", "head", "sd", "batch", "lg32", "rg32"))
    with torch.no_grad():
        _, lg16, rg16 = om.detector_forward(batch["features"], batch["cart"], sd, nm=om.Numerics.bf16(train=False))
    backbone, head = backbone.to(DEV).eval(), head.to(DEV).eval()
    data = {k: (v.to(DEV) if k != "annotations" else v) for k, v in batch.items()}
    from range_view_3d_detection_amd import engine as E

    E.PROFILE = E.KernelProfile()
    try:
        with torch.no_grad():
            feats = backbone(data)
            outputs, _ = head(feats, data, return_loss=False)
        torch.cuda.synchronize()
        ran = set(name for name, *_ in E.PROFILE.records)
    finally:
        E.PROFILE = None
    assert {"tapconv4_kernel<256>", "tapconv4_kernel<128>"} <= ran and any(n.startswith(("tapconv5_kernel<", "tapconv6_kernel<")) for n in ran), sorted(ran)
    logits, reg = outputs[1][0]["logits"].float().cpu(), outputs[1][0]["regressands"].float().cpu()
    m = {"logits~bf16": rel_err(logits, lg16), "logits~fp32": rel_err(logits, lg32), "emu~fp32": rel_err(lg16, lg32),
         "reg~bf16": rel_err(reg, rg16), "reg~fp32": rel_err(reg, rg32), "reg emu~fp32": rel_err(rg16, rg32)}
    print("[rv-av2 eval 1x64x2048] " + "  ".join(f"{k} {v:.3e}" for k, v in m.items()))
    # same bounds as the training-mode crop test
    _check_forward(m)
    _check_direction(logits, lg16, lg32, reg, rg16, rg32)
