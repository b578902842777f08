"""Inference: eval-mode BatchNorm folded into the conv's weight image and bias, ReLU in the epilogue (RV_OUT_RELU).

conv -> BatchNorm(eval) -> ReLU is ONE launch writing the activation (the reference: cuDNN conv + batch_norm + relu_).
* exact: integer activations / weights, BatchNorm parameters chosen so that scale = gamma and shift are exactly representable:
  the stored output must EQUAL relu(conv * gamma + shift) (rounded once to the operand type) on every kernel generation (tapconv5 3x3, tapconv4 1x1, the folded
  stride-2 form, the conv-transpose phases), in the bf16 and in the fp16 build;
* the same programs with the fold switched off (``engine.EVAL_FOLD = False``: BatchNorm applied as a folded operand of the
  consumer, as in training) give the same block outputs within bf16 rounding (2e-2 of max).
"""

from __future__ import annotations

import ctypes

import pytest
import torch
import torch.nn.functional as F

from test_gpu_forward import DEV, rel_err
from test_gpu_tapconv4 import _ints

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[5, 6], ids=["gen5", "gen6"])
def _small_grids_allowed(request):
    """Crops this small stay below the libraries' tile-count thresholds: lift them, once with the halo-resident layers on the fifth
    generation (256 x 256 / 256 x 128 tiles) and once on the sixth (512 x 128 tiles, where H >= 16), on both operand builds."""
    from range_view_3d_detection_amd import _lib as L

    with L.select(L.SEL_SMALL_GRIDS | (L.SEL_SMALL_GRIDS6 if request.param == 6 else L.SEL_NO_GEN6)):
        yield


def _exact_bn(c, g):
    bn = torch.nn.BatchNorm2d(c, eps=0.0)
    bn.weight.data = torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (c,), generator=g)]
    bn.bias.data = _ints((c,), g, -4, 5)
    bn.running_mean.data = _ints((c,), g, -2, 3)
    bn.running_var.data = torch.ones(c)  # eps = 0, var = 1: scale = gamma and shift = beta - mean * gamma exactly
    return bn.eval()


@pytest.mark.parametrize("tag", ["bf16", "f16"])
@pytest.mark.parametrize("kind", ["3x3", "1x1", "3x3s2", "convT"])
def test_eval_fold_exact(kind, tag):
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd.nn.modules.conv import Conv2dSame

    g = torch.Generator().manual_seed(hash(kind) % 1000)
    cin, cout = 128, 256
    x = _ints((2, cin, 16, 256), g, -1, 2)
    if kind == "convT":
        m = torch.nn.ConvTranspose2d(cin, cout, kernel_size=(3, 4), stride=(1, 2), padding=(1, 1), bias=False)
        m.weight.data = _ints(m.weight.shape, g, -1, 2)
        ref = F.conv_transpose2d(x, m.weight.data, stride=(1, 2), padding=(1, 1))
        conv = m
    else:
        k, s = (1, (1, 1)) if kind == "1x1" else (3, (1, 2) if kind == "3x3s2" else (1, 1))
        m = Conv2dSame(cin, cout, k, stride=s, bias=False)
        m.conv.weight.data = _ints(m.conv.weight.shape, g, -1, 2)
        t_ = k - 1
        ref = F.conv2d(F.pad(x, [t_ // 2, t_ - t_ // 2, t_ // 2, t_ - t_ // 2]), m.conv.weight.data, stride=s)
        conv = m.conv
    bn = _exact_bn(cout, g)
    scale = bn.weight.data
    want = F.relu(ref * scale.view(1, -1, 1, 1) + (bn.bias.data - bn.running_mean.data * scale).view(1, -1, 1, 1))
    assert float(want.abs().max()) < 2048
    conv, bn = conv.to(DEV), bn.to(DEV)
    with L.operand(tag):
        t = E.Tape(False, DEV)
        out = E.conv_bn(t, E.tap_layer(conv), E.Act.from_nchw(x.to(DEV)), bn, relu=True)
    assert isinstance(out, E.Act) and len(t.ops) == 1  # one launch, a plain activation
    # every partial sum is exact in fp32: the stored value is relu(conv * gamma + shift) rounded ONCE to the operand type
    got, exp = out.nchw().float().cpu(), want.to(torch.float16 if tag == "f16" else torch.bfloat16).float()
    assert torch.equal(got, exp), (float((got - exp).abs().max()), int((got != exp).sum()), got.numel())


def test_eval_fold_equals_the_unfolded_programs(golden):
    """Block programs and the tiny detector in eval mode: folded (inference default) against unfolded (BatchNorm as a folded
    operand of the consumer) and against the reference's eval outputs."""
    from range_view_3d_detection_amd import engine as E
    from test_gpu_model import load_tiny

    g = golden("tiny_model")
    backbone, head = load_tiny(g)
    backbone.eval()
    head.eval()
    data = {"features": g["features"].to(DEV), "cart": g["cart"].to(DEV), "mask": g["mask"].to(DEV)}

    def run():
        with torch.no_grad():
            out, _ = head(backbone(data), data, return_loss=False)
        return out[1][0]["logits"].float().cpu(), out[1][0]["regressands"].float().cpu()

    lf, rf = run()
    old = E.EVAL_FOLD
    E.EVAL_FOLD = False
    try:
        lu, ru = run()
    finally:
        E.EVAL_FOLD = old
    assert rel_err(lf, lu) < 2e-2 and rel_err(rf, ru) < 2e-2, (rel_err(lf, lu), rel_err(rf, ru))
    # and no further from the reference's fp32 eval outputs than the unfolded form is (+ 20 %)
    ef, eu = rel_err(lf, g["eval/logits"]), rel_err(lu, g["eval/logits"])
    print(f"tiny model eval logits vs the reference: folded {ef:.3e}, unfolded {eu:.3e}")
    assert ef < 4e-2 and ef < 1.2 * eu + 1e-3


@pytest.mark.parametrize("tag", ["bf16", "f16"])
@pytest.mark.parametrize("kind,hw", [("3x3", (16, 256)), ("1x1", (16, 256)), ("3x3s2", (16, 256)), ("convT", (16, 256)), ("3x3", (4, 32)), ("1x1", (4, 32))])
def test_residual_epilogue_equals_the_separate_pass(kind, hw, tag):
    """rv_tap_residual -- relu(bn(conv(x)) + res) and res + relu(bn(convT(x))) leaving the conv's own launch -- against the
    folded conv followed by the element-wise pass it replaces: EQUAL bit for bit on random (non-integer) data, because the
    conv result is rounded to the storage type before the residual is added, exactly as the stored intermediate was.  Shapes
    pick every kernel generation: tapconv5 (3x3), tapconv4 (1x1), the folded stride-2 form, the conv-transpose phases on the
    16 x 256 images; the register-staged generic kernels on the 4 x 32 ones."""
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd.nn.modules.conv import Conv2dSame

    g = torch.Generator().manual_seed(7 + sum(map(ord, kind)) + hw[0])
    cin, cout = 128, 256
    h, w = hw
    x = torch.randn(2, cin, h, w, generator=g)
    if kind == "convT":
        conv = torch.nn.ConvTranspose2d(cin, cout, kernel_size=(3, 4), stride=(1, 2), padding=(1, 1), bias=False)
        wo = 2 * w
    else:
        k, s = (1, (1, 1)) if kind == "1x1" else (3, (1, 2) if kind == "3x3s2" else (1, 1))
        conv = Conv2dSame(cin, cout, k, stride=s, bias=False).conv
        wo = w // s[1]
    bn = torch.nn.BatchNorm2d(cout)
    bn.weight.data = 0.5 + torch.rand(cout, generator=g)
    bn.bias.data = 0.2 * torch.randn(cout, generator=g)
    bn.running_mean.data = 0.1 * torch.randn(cout, generator=g)
    bn.running_var.data = 0.5 + torch.rand(cout, generator=g)
    conv, bn = conv.to(DEV), bn.eval().to(DEV)
    res = torch.randn(2, cout, h, wo, generator=g)
    for relu_conv, relu_out in ((False, True), (True, False)):
        with L.operand(tag):
            t = E.Tape(False, DEV)
            xa, ra = E.Act.from_nchw(x.to(DEV)), E.Act.from_nchw(res.to(DEV))
            fused = E.conv_bn_residual(t, E.tap_layer(conv), xa, bn, ra, relu_conv, relu_out)
            assert fused is not None and len(t.ops) == 1
            y = E.conv_bn(t, E.tap_layer(conv), xa, bn, relu=relu_conv)
            sep = E.CombineOp(t, y, ra, relu_out=relu_out).out
            a, b = fused.nchw().float().cpu(), sep.nchw().float().cpu()
        assert torch.equal(a, b), (kind, relu_conv, relu_out, float((a - b).abs().max()), int((a != b).sum()))


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_residual_epilogue_in_the_tiny_detector(golden, precision):
    """The tiny detector in eval mode with the block sums in the conv epilogues (default) against the separate passes
    (``engine.EVAL_RES_FUSE = False``): every output EQUAL, and fewer element-wise launches on the tape."""
    from range_view_3d_detection_amd import engine as E
    from test_gpu_model import load_tiny

    g = golden("tiny_model")
    backbone, head = load_tiny(g)
    backbone.eval()
    head.eval()
    data = {"features": g["features"].to(DEV), "cart": g["cart"].to(DEV), "mask": g["mask"].to(DEV)}

    def run():
        n0 = E.COMBINE_LAUNCHES
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=precision == "f16"):
            out, _ = head(backbone(data), data, return_loss=False)
        return out[1][0]["logits"].float().cpu(), out[1][0]["regressands"].float().cpu(), E.COMBINE_LAUNCHES - n0

    lf, rf, nf = run()
    E.EVAL_RES_FUSE = False
    try:
        lu, ru, nu = run()
    finally:
        E.EVAL_RES_FUSE = True
    assert torch.equal(lf, lu) and torch.equal(rf, ru)
    assert nf < nu, (nf, nu)


@pytest.mark.parametrize("tag", ["bf16", "f16"])
@pytest.mark.parametrize("C", [256, 128])
@pytest.mark.parametrize("N,H,W", [(2, 5, 37), (3, 16, 160)])
def test_pos_modulate_forward_equals_the_two_kernels(N, H, W, C, tag):
    """rv_pos_modulate_forward (inference: positional pair + modulation in one persistent kernel, neither h1 nor y2 stored)
    against rv_pos_forward followed by rv_meta_modulate: EQUAL bit for bit (y2 is rounded to the storage type before the
    BatchNorm, as the stored tensor was).  2 x 5 x 37: rows that wrap image rows and images inside a step, a partial last
    step, every border neighbour; 3 x 16 x 160: 540 (C = 256) / 270 (C = 128) steps over 256 persistent workgroups."""
    from range_view_3d_detection_amd import _lib as L

    dt = torch.float16 if tag == "f16" else torch.bfloat16
    gen = torch.Generator().manual_seed(N * 1000 + W + C)
    P = N * H * W * 9
    rel = torch.zeros(P, 32, dtype=dt)
    rel[:, :3] = (torch.randn(P, 3, generator=gen) * 2).to(dt)
    w1 = torch.zeros(C, 32, dtype=dt)
    w1[:, :3] = torch.randn(C, 3, generator=gen).to(dt)
    w2 = (torch.randn(C, C, generator=gen) / 16).to(dt)
    s1, t1 = 0.5 + torch.rand(C, generator=gen), 0.3 * torch.randn(C, generator=gen)
    s2, t2 = 0.5 + torch.rand(C, generator=gen), 0.3 * torch.randn(C, generator=gen)
    feat = torch.randn(N * H * W, C, generator=gen).to(dt)
    rel, w1, w2, s1, t1, s2, t2, feat = (x.to(DEV) for x in (rel, w1, w2, s1, t1, s2, t2, feat))
    with L.operand(tag):
        h1 = torch.empty((P, C), dtype=dt, device=DEV)
        y2 = torch.empty((P, C), dtype=dt, device=DEV)
        L.call("rv_pos_forward", L.ptr(rel), L.i32(32), L.i32(3), L.i64(P), L.ptr(w1), L.i32(32), L.ptr(s1), L.ptr(t1), L.ptr(w2), L.i32(C),
               L.ptr(h1), L.ptr(y2), None, L.stream_ptr())
        want = torch.full((N * H * W, 9 * C), float("nan"), dtype=dt, device=DEV)
        L.call("rv_meta_modulate", L.ptr(y2), L.ptr(s2), L.ptr(t2), L.ptr(feat), L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(want),
               L.stream_ptr())
        got = torch.full((N * H * W, 9 * C), float("nan"), dtype=dt, device=DEV)
        L.call("rv_pos_modulate_forward", L.ptr(rel), L.i32(32), L.i32(3), L.ptr(w1), L.i32(32), L.ptr(s1), L.ptr(t1), L.ptr(w2), L.i32(C),
               L.ptr(s2), L.ptr(t2), L.ptr(feat), L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.ptr(got), L.stream_ptr())
        torch.cuda.synchronize()
    assert not bool(torch.isnan(want.float()).any()) and float(want.float().abs().max()) > 0
    a, b = got.float().cpu(), want.float().cpu()
    assert torch.equal(a, b), (float((a - b).abs().max()), int((a != b).sum()), a.numel())


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("C", [256, 128])
def test_meta_kernel_eval_with_the_fused_stem_kernel(C, precision):
    """MetaKernel in eval mode at the rv-av2 / rv-waymo stem widths: the fused inference kernel (default) against the
    positional pair + separate modulation (``engine.POS_MOD_FUSE = False``): outputs EQUAL."""
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd.nn.stems import MetaKernel

    gen = torch.Generator().manual_seed(31)
    m = MetaKernel(5, C, 3, 2).to(DEV).eval()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.data = 0.1 * torch.randn(mod.num_features, generator=gen).to(DEV)
            mod.running_var.data = (0.5 + torch.rand(mod.num_features, generator=gen)).to(DEV)
    feats = torch.randn(2, 5, 16, 160, generator=gen).to(DEV)
    cart = (torch.randn(2, 3, 16, 160, generator=gen) * 5).to(DEV)

    def run():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=precision == "f16"):
            return m(feats, cart).float().cpu()

    a = run()
    E.POS_MOD_FUSE = False
    try:
        b = run()
    finally:
        E.POS_MOD_FUSE = True
    assert float(a.abs().max()) > 0 and torch.equal(a, b), float((a - b).abs().max())


def test_eval_mode_backward_fails_loudly(golden):
    """Eval-mode programs are inference-only (BatchNorm backward is built on batch statistics): the forward outside
    ``torch.no_grad()`` still takes the fused inference forms and equals the no-grad run bit for bit; asking for gradients
    raises ``RvError`` instead of returning something silently wrong (round-3 ADVICE, documented limitation)."""
    from range_view_3d_detection_amd import _lib as L
    from test_gpu_model import load_tiny

    g = golden("tiny_model")
    backbone, head = load_tiny(g)
    backbone.eval()
    head.eval()
    data = {"features": g["features"].to(DEV), "cart": g["cart"].to(DEV), "mask": g["mask"].to(DEV)}
    with torch.no_grad():
        ref, _ = head(backbone(data), data, return_loss=False)
    out, _ = head(backbone(data), data, return_loss=False)
    logits = out[1][0]["logits"]
    assert torch.equal(logits.detach(), ref[1][0]["logits"])
    with pytest.raises(L.RvError):
        logits.float().square().mean().backward()
