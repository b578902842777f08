"""tapconv6 (512-pixel x 128-channel tiles, 32-channel chunks, input halo resident in LDS across the taps): exact checks.

Same method as test_gpu_tapconv4/5.py: small-integer activations and weights make every partial sum an integer below 2^24, so
the bf16 output must equal the CPU convolution rounded once to bf16, bit for bit, whatever the summation order.  Every case
asserts (``rv_tap_launch_info``) that generation 6 runs.  Shapes cover ragged tile rows (H % 16 != 0) and columns
(W % 32 != 0), one / two / many 32-channel chunks (halo double-buffering, every position of the counted waits), one to four
channel tiles, bias, batch statistics (four partial rows per tile), the conv-transpose phases (2 x 3 taps), 3 x 2 kernels,
the accumulate epilogue, the BatchNorm-backward-sum epilogue and a race screen on random data.
"""

from __future__ import annotations

import ctypes

import pytest
import torch
import torch.nn.functional as F

from test_gpu_forward import DEV
from test_gpu_tapconv4 import _ints

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_grids_allowed():
    """The library keeps grids below one round of CUs on the smaller tiles (speed heuristic); lifted here per call."""
    from range_view_3d_detection_amd import _lib as L

    with L.select(L.SEL_SMALL_GRIDS | L.SEL_SMALL_GRIDS6):
        yield


def _run(module, x, stats=False):
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E

    t = E.Tape(True, x.device)
    layer = E.tap_layer(module)
    op = E.ConvOp(t, layer, E.Act.from_nchw(x), stats=stats)
    info = (ctypes.c_int32 * 4)()
    assert L.load().rv_tap_launch_info(ctypes.byref(layer.geom), ctypes.byref(op.shape), 1 if layer.fwd_form == "scatter" else 0, info) == 0
    assert info[0] == 6, list(info)
    return op.out.data[..., : layer.c_out].permute(0, 3, 1, 2).float(), op


@pytest.mark.parametrize("cin,cout,N,H,W,bias", [(32, 128, 4, 30, 520, False),    # ONE chunk (no halo refill), ragged rows / columns
                                                 (64, 256, 2, 64, 256, True),     # two chunks, two channel tiles, bias
                                                 (96, 128, 4, 17, 1030, False),   # three chunks, one-row last tile row
                                                 (512, 512, 1, 64, 288, False),   # sixteen chunks, four channel tiles
                                                 (64, 128, 8, 16, 32, False),     # a single tile per image
                                                 (320, 128, 3, 33, 1030, False),  # ten chunks, ragged
                                                 (128, 384, 1, 16, 96, True)])    # three channel tiles
def test_gather_3x3_exact(cin, cout, N, H, W, bias):
    g = torch.Generator().manual_seed(cin + W)
    m = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=bias)
    m.weight.data = _ints(m.weight.shape, g, -2, 3)
    if bias:
        m.bias.data = _ints(m.bias.shape, g, -8, 9)
    x = _ints((N, cin, H, W), g)
    ref = F.conv2d(x, m.weight.data, m.bias.data if bias else None, padding=1)
    out, op = _run(m.to(DEV), x.to(DEV), stats=not bias)
    assert torch.equal(out.cpu(), ref.bfloat16().float())
    if not bias:
        rows = op.partial[: op.rows].double().sum(dim=0).cpu()  # (2, C)
        assert torch.allclose(rows[0, :cout], ref.double().sum(dim=(0, 2, 3)), rtol=1e-6, atol=1e-3)
        assert torch.allclose(rows[1, :cout], (ref.double() ** 2).sum(dim=(0, 2, 3)), rtol=1e-5)


def test_gather_3x2_exact():
    """Even-width kernel through Conv2dSame's asymmetric padding rule (six taps: the shortest chunk the kernel takes)."""
    from range_view_3d_detection_amd.nn.modules.conv import Conv2dSame

    g = torch.Generator().manual_seed(32)
    m = Conv2dSame(128, 256, (3, 2), bias=False)
    m.conv.weight.data = _ints(m.conv.weight.shape, g, -2, 3)
    x = _ints((2, 128, 24, 200), g)
    ref = F.conv2d(F.pad(x, [0, 1, 1, 1]), m.conv.weight.data)
    out, _ = _run(m.conv.to(DEV), x.to(DEV))
    assert torch.equal(out.cpu(), ref.bfloat16().float())


@pytest.mark.parametrize("kernel,stride,padding,N,H,W", [((3, 4), (1, 2), (1, 1), 4, 16, 512), ((3, 8), (1, 4), (1, 2), 4, 16, 256),
                                                         ((3, 4), (1, 2), (1, 1), 3, 21, 600)])
@pytest.mark.parametrize("cout", [256, 128])
def test_scatter_conv_transpose_exact(kernel, stride, padding, N, H, W, cout):
    g = torch.Generator().manual_seed(W)
    m = torch.nn.ConvTranspose2d(128, cout, kernel_size=kernel, stride=stride, padding=padding, bias=False)
    m.weight.data = _ints(m.weight.shape, g, -2, 3)
    x = _ints((N, 128, H, W), g)
    ref = F.conv_transpose2d(x, m.weight.data, stride=stride, padding=padding)
    out, _ = _run(m.to(DEV), x.to(DEV))
    assert torch.equal(out.cpu(), ref.bfloat16().float())


def test_input_gradient_and_accumulate_exact():
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E

    g = torch.Generator().manual_seed(5)
    N, H, W, cin, cout = 4, 32, 512, 256, 128
    m = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False)
    m.weight.data = _ints(m.weight.shape, g, -2, 3)
    dy = _ints((N, cout, H, W), g)
    old = _ints((N, cin, H, W), g, -20, 21)
    ref = F.conv_transpose2d(dy, m.weight.data, padding=1)  # d/dx of conv2d(x, w, padding=1)
    layer = E.tap_layer(m.to(DEV))
    gact = E.Act.from_nchw(dy.to(DEV))
    for accumulate in (False, True):
        dst = E.Act.from_nchw(old.to(DEV)) if accumulate else E.Act.empty(N, H, W, cin, DEV)
        shape = L.TapShape(N, H, W, W, gact.ld, dst.ld, L.OUT_ACCUM if accumulate else 0)
        info = (ctypes.c_int32 * 4)()
        assert L.load().rv_tap_launch_info(ctypes.byref(layer.geom), ctypes.byref(shape), 1, info) == 0 and info[0] == 6, list(info)
        L.call("rv_tap_scatter", ctypes.byref(layer.geom), ctypes.byref(shape), gact.ptr(), None, None, L.ptr(layer.packed("scatter")), None,
               dst.ptr(), None, L.stream_ptr())
        got = dst.data[..., :cin].permute(0, 3, 1, 2).float().cpu()
        want = ref.bfloat16().float()
        if accumulate:
            want = (want + old).bfloat16().float()
        assert torch.equal(got, want)


@pytest.mark.parametrize("cin,cout", [(512, 512), (128, 128)])
def test_repeatable_on_random_data(cin, cout):
    """Race screen (see test_gpu_tapconv4.py): fixed summation order => repeated launches on random data agree bit for bit;
    a halo or weight piece read before its DMA landed, or overwritten while still being read, shows up as a difference."""
    g = torch.Generator().manual_seed(9)
    m = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False)
    m.weight.data = torch.randn(m.weight.shape, generator=g) * 0.05
    m = m.to(DEV)
    x = torch.randn(4, cin, 64, 1024, generator=g).bfloat16().float().to(DEV)
    first, _ = _run(m, x)
    ref = F.conv2d(x[:1, :, :6, :96], m.weight.data.bfloat16().float(), padding=1)[:, :, 1:5, 1:95]
    assert float((first[:1, :, 1:5, 1:95] - ref).abs().max()) / float(ref.abs().max()) < 1e-2  # bf16 output rounding
    for _ in range(30):
        again, _ = _run(m, x)
        assert torch.equal(first, again)


def test_matches_generation_5_on_random_data():
    """Both generations accumulate in fp32 (in different orders): outputs equal to one bf16 ulp, statistics to 1e-5."""
    from range_view_3d_detection_amd import _lib as L

    g = torch.Generator().manual_seed(11)
    m = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False)
    m.weight.data = torch.randn(m.weight.shape, generator=g) * 0.03
    m = m.to(DEV)
    x = torch.randn(2, 256, 64, 512, generator=g).bfloat16().float().to(DEV)
    out6, op6 = _run(m, x, stats=True)
    with L.select(L.SEL_NO_GEN6):
        from range_view_3d_detection_amd import engine as E

        t = E.Tape(True, x.device)
        op5 = E.ConvOp(t, E.tap_layer(m), E.Act.from_nchw(x), stats=True)
        out5 = op5.out.data[..., :256].permute(0, 3, 1, 2).float()
    assert float((out6 - out5).abs().max()) <= 2 ** -7 * float(out5.abs().max())
    s6 = op6.partial[: op6.rows].double().sum(dim=0)
    s5 = op5.partial[: op5.rows].double().sum(dim=0)
    assert torch.allclose(s6, s5, rtol=1e-5, atol=1e-3 * float(s5.abs().max()))


@pytest.mark.parametrize("cin,c,N,H,W,relu", [(64, 256, 2, 24, 200, True), (64, 512, 1, 17, 96, True), (128, 128, 2, 16, 64, False)])
def test_data_grad_launch_forms_the_batchnorm_backward_sums(cin, c, N, H, W, relu, monkeypatch):
    """conv -> BatchNorm(+ReLU) -> conv: the second conv's backward-data launch (``rv_tap_data_grad_bnb``, now generation 6)
    also emits sum(g), sum(g*xhat) of the BatchNorm between them -- same checks as the generation-5 test."""
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd import engine_bwd

    gen = torch.Generator().manual_seed(c + W)
    c1 = torch.nn.Conv2d(cin, c, 3, padding=1, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(c).to(DEV).train()
    c2 = torch.nn.Conv2d(c, 256, 3, padding=1, bias=False).to(DEV)
    bn.weight.data = (0.5 + torch.rand(c, generator=gen)).to(DEV)
    bn.bias.data = (0.3 * torch.randn(c, generator=gen)).to(DEV)
    x = torch.randn(N, cin, H, W, generator=gen).to(DEV)
    gout = torch.randn(N, 256, H, W, generator=gen).to(DEV)

    def run(fuse):
        calls = []
        real = L.call
        monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
        monkeypatch.setattr(E, "BNB_FUSE", fuse)
        t = E.Tape(True, x.device)
        h = E.conv_bn(t, E.tap_layer(c1), E.Act.from_nchw(x), bn, relu=relu)
        op2 = E.ConvOp(t, E.tap_layer(c2), h, stats=False)
        t.set_grad(op2.out, engine_bwd.grad_act_like(op2.out, gout.to(torch.bfloat16)))
        t.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(L, "call", real)
        return calls, t.param_grads[id(bn.weight)].float().cpu(), t.param_grads[id(bn.bias)].float().cpu(), t.param_grads[id(c1.weight)].float().cpu()

    calls_f, dg_f, db_f, dw_f = run(True)
    calls_s, dg_s, db_s, dw_s = run(False)
    assert "rv_tap_data_grad_bnb" in calls_f and "rv_bn_bwd_reduce" not in calls_f, calls_f
    assert "rv_bn_bwd_reduce" in calls_s and "rv_tap_data_grad_bnb" not in calls_s
    assert torch.allclose(dg_f, dg_s, rtol=1e-4, atol=1e-3 * float(dg_s.abs().max()))
    assert torch.allclose(db_f, db_s, rtol=1e-4, atol=1e-3 * float(db_s.abs().max()))
    assert float((dw_f - dw_s).abs().max()) <= 2e-2 * float(dw_s.abs().max())
    cos = float((dw_f * dw_s).sum() / (dw_f.norm() * dw_s.norm()))
    assert cos > 0.99999, cos
