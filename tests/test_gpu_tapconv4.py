"""tapconv4 (256 x 256 tiles, LDS-DMA staging, counted waits): exact checks with integer data.

With small-integer activations and weights every product and every partial sum is an integer below 2^24, so the fp32
accumulators hold the exact result whatever the summation order; the kernel's bf16 output must therefore equal the
CPU convolution rounded once to bf16, bit for bit.  Every case asserts (through ``rv_tap_launch_info``) that it is this
kernel that runs, not one of the register-staged ones.
"""

from __future__ import annotations

import ctypes

import pytest
import torch
import torch.nn.functional as F

from test_gpu_forward import DEV

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _generation4_only():
    """Multi-tap layers with 256-channel tiles go to tapconv5 by default, 1x1 C -> C layers to the pointwise streaming GEMM (generation 7,
    tests/test_gpu_pointwise.py); these tests pin generation 4."""
    from range_view_3d_detection_amd import _lib as L

    with L.select(L.SEL_NO_GEN5 | L.SEL_NO_POINTWISE):
        yield


def _ints(shape, g, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=g).float()


def _run(module, x, stats=False, expect_kernel=4):
    """bf16-output tap-conv launch through the engine; returns (NCHW float output, ConvOp)."""
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E

    t = E.Tape(True, x.device)
    layer = E.tap_layer(module)
    op = E.ConvOp(t, layer, E.Act.from_nchw(x), stats=stats)
    info = (ctypes.c_int32 * 4)()
    assert L.load().rv_tap_launch_info(ctypes.byref(layer.geom), ctypes.byref(op.shape), 1 if layer.fwd_form == "scatter" else 0, info) == 0
    assert info[0] == expect_kernel, list(info)
    return op.out.data[..., : layer.c_out].permute(0, 3, 1, 2).float(), op


@pytest.mark.parametrize("cin,cout,N,H,W,bias", [(64, 256, 4, 30, 520, False),   # ragged rows and columns, one channel tile
                                                 (128, 512, 2, 64, 256, True),   # two channel tiles, two K chunks, bias
                                                 (192, 256, 4, 17, 1030, False),  # three K chunks, one-row last tile
                                                 (64, 128, 4, 30, 520, False),    # 128-channel variant (three-piece K tiles)
                                                 (128, 384, 3, 32, 300, True)])   # 128-channel variant, three channel tiles, bias
def test_gather_3x3_exact(cin, cout, N, H, W, bias):
    _gather_exact(cin, cout, 3, N, H, W, bias)


@pytest.mark.parametrize("cin,cout,N,H,W", [(256, 256, 4, 32, 520), (64, 512, 2, 64, 300), (576, 256, 2, 33, 1000), (128, 128, 4, 32, 520)])
def test_gather_1x1_exact(cin, cout, N, H, W):
    _gather_exact(cin, cout, 1, N, H, W, False)


def _gather_exact(cin, cout, k, N, H, W, bias):
    g = torch.Generator().manual_seed(cin + W)
    m = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=bias)
    m.weight.data = _ints(m.weight.shape, g, -2, 3)
    if bias:
        m.bias.data = _ints(m.bias.shape, g, -8, 9)
    x = _ints((N, cin, H, W), g)
    ref = F.conv2d(x, m.weight.data, m.bias.data if bias else None, padding=k // 2)
    out, op = _run(m.to(DEV), x.to(DEV), stats=not bias)
    assert torch.equal(out.cpu(), ref.bfloat16().float())
    if not bias:
        rows = op.partial[: op.rows].double().sum(dim=0).cpu()  # (2, C)
        assert torch.allclose(rows[0, :cout], ref.double().sum(dim=(0, 2, 3)), rtol=1e-6, atol=1e-3)
        assert torch.allclose(rows[1, :cout], (ref.double() ** 2).sum(dim=(0, 2, 3)), rtol=1e-5)


@pytest.mark.parametrize("kernel,stride,padding,N,H,W,cout", [((3, 4), (1, 2), (1, 1), 4, 16, 512, 256), ((3, 8), (1, 4), (1, 2), 4, 16, 256, 256),
                                                              ((3, 4), (1, 2), (1, 1), 3, 21, 600, 256), ((3, 8), (1, 4), (1, 2), 4, 16, 300, 128)])
def test_scatter_conv_transpose_exact(kernel, stride, padding, N, H, W, cout):
    g = torch.Generator().manual_seed(W)
    m = torch.nn.ConvTranspose2d(64, cout, kernel_size=kernel, stride=stride, padding=padding, bias=False)
    m.weight.data = _ints(m.weight.shape, g, -2, 3)
    x = _ints((N, 64, H, W), g)
    ref = F.conv_transpose2d(x, m.weight.data, stride=stride, padding=padding)
    out, _ = _run(m.to(DEV), x.to(DEV))
    assert torch.equal(out.cpu(), ref.bfloat16().float())


@pytest.mark.parametrize("cin", [256, 128])
def test_input_gradient_and_accumulate_exact(cin):
    """Backward-data of a 3x3 conv is the scatter form with one phase (plain bf16 gradient in => this kernel), once into a
    fresh buffer and once accumulating into an existing gradient (RV_OUT_ACCUM: bf16(bf16(conv) + old))."""
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E

    g = torch.Generator().manual_seed(5)
    N, H, W, cout = 4, 32, 512, 64
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
        assert L.load().rv_tap_launch_info(ctypes.byref(layer.geom), ctypes.byref(shape), 1, info) == 0 and info[0] == 4, list(info)
        L.call("rv_tap_scatter", ctypes.byref(layer.geom), ctypes.byref(shape), gact.ptr(), None, None, L.ptr(layer.packed("scatter")), None,
               dst.ptr(), None, L.stream_ptr())
        got = dst.data[..., :cin].permute(0, 3, 1, 2).float().cpu()
        want = ref.bfloat16().float()
        if accumulate:
            want = (want + old).bfloat16().float()
        assert torch.equal(got, want)


def test_repeatable_on_random_data():
    """Race screen: the kernel is deterministic by construction (fixed summation order), so repeated launches on random
    data must agree bit for bit; a staged piece read before its DMA landed, or overwritten while still being read, shows
    up as a difference between runs."""
    g = torch.Generator().manual_seed(9)
    m = torch.nn.Conv2d(512, 512, 3, padding=1, bias=False)
    m.weight.data = torch.randn(m.weight.shape, generator=g) * 0.05
    m = m.to(DEV)
    x = torch.randn(4, 512, 64, 1024, generator=g).bfloat16().float().to(DEV)
    first, _ = _run(m, x)
    ref = F.conv2d(x[:1, :, :6, :96], m.weight.data.bfloat16().float(), padding=1)[:, :, 1:5, 1:95]
    assert float((first[:1, :, 1:5, 1:95] - ref).abs().max()) / float(ref.abs().max()) < 1e-2  # bf16 output rounding
    for _ in range(30):
        again, _ = _run(m, x)
        assert torch.equal(first, again)


@pytest.mark.parametrize("C,N,H,W", [(128, 2, 8, 2656), (128, 1, 16, 333), (256, 1, 8, 520)])
def test_gather_1x1_into_a_channel_slice_exact(C, N, H, W):
    """1x1 C -> C on a plain operand, written into a channel slice of a wider tensor (row pitch 2 C: the stem's last conv writes the backbone's
    in-place concat) with the batch statistics, widths that are not multiples of the 64-column tile (2656 = rv-waymo): the output must equal the CPU
    convolution, the statistics its sums, and the OTHER slice must stay untouched.  (Round 6: written while hunting a fault of free-running rv-waymo
    steps whose stem conv took this launch on a written-out operand -- profiles/r06_ab_notes.md section 4.)"""
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E

    g = torch.Generator().manual_seed(C + W)
    m = torch.nn.Conv2d(C, C, 1, bias=False)
    m.weight.data = _ints(m.weight.shape, g, -2, 3)
    x = _ints((N, C, H, W), g)
    ref = F.conv2d(x, m.weight.data)
    t = E.Tape(True, DEV)
    layer = E.tap_layer(m.to(DEV))
    wide = E.Act.empty(N, H, W, 2 * C, DEV)
    wide.data.fill_(7.0)
    guard = torch.full((4096,), 5.0, dtype=wide.data.dtype, device=DEV)  # (allocated right behind: a write past the end of `wide` would land here)
    with L.select(L.SEL_SMALL_GRIDS):  # (a crop: lift the tile-count heuristic so that generation 4 runs)
        op = E.ConvOp(t, layer, E.Act.from_nchw(x.to(DEV)), stats=True, out=wide.slice(0, C))
    info = (ctypes.c_int32 * 4)()
    assert L.load().rv_tap_launch_info(ctypes.byref(layer.geom), ctypes.byref(op.shape), 0, info) == 0 and info[0] == 4, list(info)
    torch.cuda.synchronize()
    assert torch.equal(wide.data[..., :C].permute(0, 3, 1, 2).float().cpu(), ref.bfloat16().float())
    assert bool((wide.data[..., C:] == 7.0).all()) and bool((guard == 5.0).all())
    rows = op.partial[: op.rows].double().sum(dim=0).cpu()
    assert torch.allclose(rows[0, :C], ref.double().sum(dim=(0, 2, 3)), rtol=1e-6, atol=1e-3)
    assert torch.allclose(rows[1, :C], (ref.double() ** 2).sum(dim=(0, 2, 3)), rtol=1e-5)
