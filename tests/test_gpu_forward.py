"""GPU parity of the forward HIP path against the oracle and the reference's golden vectors.

All tests here need a real MI355X (``-m gpu``) and call the kernels through the C ABI
(``librv3d_hip.so`` via ``range_view_3d_detection_amd._lib``).

Tolerances (stated per the north-star "stated fp tolerance for conv/box regression"):

* raw tap-conv with bf16-representable operands and fp32 output: products are exact in fp32,
  only the accumulation order differs from the CPU conv => 2e-5 of the output's max;
* bf16 module programs vs the oracle run with the SAME storage rounding points
  (``oracle.model.Numerics.bf16``): 1.5e-2 of max (a bf16 ulp is 7.8e-3; isolated values that
  sit on a rounding boundary may land on the neighbouring bf16 value);
* bf16 module programs vs the reference's own fp32 golden outputs: 4e-2 of max;
* decode: 1e-5 (fp64 arithmetic rounded to fp32; device libm vs glibc differ in the last ulp);
* range-image bins and z-buffer ownership: bit-exact.
"""

from __future__ import annotations

import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def bf16r(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).float()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)


def run_conv_f32(module, x, in_affine=None, relu=False):
    """One tap-conv launch with fp32 output (no rounding of the result)."""
    from range_view_3d_detection_amd import engine as E

    t = E.Tape(False, x.device)
    a = E.Act.from_nchw(x)
    operand = a
    if in_affine is not None:
        scale, shift = in_affine
        cp = a.cp
        sc = torch.zeros(cp, device=x.device)
        sh = torch.zeros(cp, device=x.device)
        sc[: scale.numel()] = scale
        sh[: shift.numel()] = shift
        operand = E.Lazy(a, E.BnState(None, sc, sh), relu)
    op = E.ConvOp(t, E.tap_layer(module), operand, out_f32=True)
    return op.out_t[..., : op.layer.c_out].permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize(
    "cin,cout,k,stride,H,W",
    [
        (40, 24, 3, 1, 5, 48),     # ragged channels, partial M tile
        (32, 32, 3, 2, 4, 64),     # strided 3x3 (pad first, then stride)
        (64, 48, 1, 2, 3, 96),     # strided 1x1 projection
        (5, 16, 1, 1, 6, 32),      # stem projection: C_in = 5
        (128, 128, 3, 1, 4, 256),  # full 128x128 tile, two K chunks... (4 chunks)
        (96, 160, 3, 1, 2, 128),   # C_out not a multiple of the N tile
        (64, 128, 3, 1, 5, 192),   # tapconv2: odd H (masked second tile row), 3 column tiles, 64-channel chunks
        (32, 64, 3, 1, 4, 128),    # tapconv2<1>: 32-channel chunks
        (128, 64, 1, 1, 6, 128),   # tapconv2: 1x1 => double-buffered A tile
        (192, 256, 3, 1, 3, 100),  # tapconv2: ragged width (partial column tile), 3 chunks
    ],
)
def test_gather_matches_conv2d(cin, cout, k, stride, H, W):
    from oracle import model as om
    from range_view_3d_detection_amd.nn.modules.conv import Conv2dSame

    g = torch.Generator().manual_seed(cin * 131 + cout)
    m = Conv2dSame(cin, cout, kernel_size=k, stride=(1, stride), bias=False)
    m.conv.weight.data = bf16r(torch.randn(m.conv.weight.shape, generator=g) * 0.2)
    x = bf16r(torch.randn(2, cin, H, W, generator=g))
    ref = om.conv2d_same(x, m.conv.weight.data, (1, stride))
    out = run_conv_f32(m.to(DEV).conv, x.to(DEV))
    assert rel_err(out, ref) < 2e-5


def test_gather_folded_bn_relu_prologue():
    """Operand = relu(scale*x+shift), zero *after* the transform at the padded border."""
    from oracle import model as om
    from range_view_3d_detection_amd.nn.modules.conv import Conv2dSame

    g = torch.Generator().manual_seed(7)
    m = Conv2dSame(64, 32, kernel_size=3, stride=1, bias=False)
    m.conv.weight.data = bf16r(torch.randn(m.conv.weight.shape, generator=g) * 0.2)
    x = bf16r(torch.randn(2, 64, 5, 40, generator=g))
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    operand = bf16r(F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)))
    ref = om.conv2d_same(operand, m.conv.weight.data)
    out = run_conv_f32(m.to(DEV).conv, x.to(DEV), (scale.to(DEV), shift.to(DEV)), relu=True)
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("kernel,stride,padding,W,cout", [((3, 8), (1, 4), (1, 2), 24, 40), ((3, 4), (1, 2), (1, 1), 40, 40),
                                                          ((3, 4), (1, 2), (1, 1), 160, 40), ((3, 8), (1, 4), (1, 2), 64, 96),
                                                          ((3, 4), (1, 2), (1, 1), 96, 128)])
def test_scatter_matches_conv_transpose2d(kernel, stride, padding, W, cout):
    g = torch.Generator().manual_seed(W)
    m = torch.nn.ConvTranspose2d(48, cout, kernel_size=kernel, stride=stride, padding=padding, bias=False)
    m.weight.data = bf16r(torch.randn(m.weight.shape, generator=g) * 0.2)
    x = bf16r(torch.randn(2, 48, 5, W, generator=g))
    ref = F.conv_transpose2d(x, m.weight.data, stride=stride, padding=padding)
    out = run_conv_f32(m.to(DEV), x.to(DEV))
    assert rel_err(out, ref) < 2e-5


def test_bias_and_stats_epilogue():
    """fp32 bias add; BatchNorm partial sums equal the sums of the fp32 result."""
    from range_view_3d_detection_amd import engine as E

    g = torch.Generator().manual_seed(11)
    m = torch.nn.Conv2d(32, 26, 1, bias=True)
    m.weight.data = bf16r(torch.randn(m.weight.shape, generator=g) * 0.3)
    m.bias.data = torch.randn(26, generator=g)
    x = bf16r(torch.randn(2, 32, 4, 80, generator=g))
    ref = F.conv2d(x, m.weight.data, m.bias.data)
    out = run_conv_f32(m.to(DEV), x.to(DEV))
    assert rel_err(out, ref) < 2e-5

    m2 = torch.nn.Conv2d(32, 64, 3, padding=1, bias=False)
    m2.weight.data = bf16r(torch.randn(m2.weight.shape, generator=g) * 0.3)
    y = F.conv2d(x, m2.weight.data, padding=1)
    t = E.Tape(True, DEV)
    op = E.ConvOp(t, E.tap_layer(m2.to(DEV)), E.Act.from_nchw(x.to(DEV)), stats=True)
    part = op.partial[: op.rows].double().sum(0).cpu()  # (2, C)
    assert rel_err(part[0], y.double().sum((0, 2, 3))) < 1e-4
    assert rel_err(part[1], (y.double() ** 2).sum((0, 2, 3))) < 1e-5
    assert rel_err(op.out.nchw().float(), bf16r(y)) < 4e-3  # stored as bf16 (1 ulp = 2^-8 relative)


# ---------------------------------------------------------------------------------------------
# module programs vs oracle (bf16 storage emulation) and vs the reference's golden outputs
# ---------------------------------------------------------------------------------------------
def _load(module, sd, dev=DEV):
    module.load_state_dict({k: v for k, v in sd.items()})
    return module.to(dev)


def _module_cases():
    from range_view_3d_detection_amd.nn.blocks import AggregationBlock, BasicBlock, ResidualBlock
    from oracle import model as om

    return {
        "basic_plain": (lambda: BasicBlock(8, 8), lambda x, sd, nm: om.basic_block(x, sd, "m", nm=nm), 1),
        "basic_proj_s12": (lambda: BasicBlock(8, 16, stride=(1, 2), project=True), lambda x, sd, nm: om.basic_block(x, sd, "m", (1, 2), True, nm), 1),
        "basic_k1_proj": (lambda: BasicBlock(5, 16, kernel_size=1, project=True), lambda x, sd, nm: om.basic_block(x, sd, "m", (1, 1), True, nm), 1),
        "residual_s12_n3": (lambda: ResidualBlock(8, 16, num_blocks=3, stride=(1, 2)), lambda x, sd, nm: om.residual_block(x, sd, "m", 3, (1, 2), nm), 1),
        "agg_k8_s4": (lambda: AggregationBlock(8, 16, 8, kernel_size=(3, 8), stride=(1, 4), padding=(1, 2), num_blocks=2),
                      lambda a, b, sd, nm: om.aggregation_block(a, b, sd, "m", (1, 4), (1, 2), 2, nm), 2),
        "agg_k4_s2": (lambda: AggregationBlock(8, 16, 8, kernel_size=(3, 4), stride=(1, 2), padding=(1, 1), num_blocks=1),
                      lambda a, b, sd, nm: om.aggregation_block(a, b, sd, "m", (1, 2), (1, 1), 1, nm), 2),
    }


@pytest.mark.parametrize("name", ["basic_plain", "basic_proj_s12", "basic_k1_proj", "residual_s12_n3", "agg_k8_s4", "agg_k4_s2"])
@pytest.mark.parametrize("train", [True, False])
def test_block_forward(golden, name, train):
    from oracle import model as om

    g = golden("conv_blocks")
    make, ofn, n_in = _module_cases()[name]
    sd = g.sub(f"{name}/sd")
    m = _load(make(), sd)
    m.train(train)
    xs = [g[f"{name}/in{i}"] for i in range(n_in)]
    with torch.no_grad():
        out = m(*[x.to(DEV) for x in xs]).float()
    nm = om.Numerics.bf16(train=train)
    with torch.no_grad():
        exp = ofn(*xs, {f"m.{k}": v for k, v in sd.items()}, nm)
    assert rel_err(out, exp) < 1.5e-2, "vs oracle with bf16 storage emulation"
    ref = g[f"{name}/out"] if train else g[f"{name}/out_eval"]
    assert rel_err(out, ref) < 4e-2, "vs the reference's fp32 output"
    if train:  # running statistics were updated in place like nn.BatchNorm2d does
        after = {k: v for k, v in m.state_dict().items() if "running_" in k}
        for k, v in g.sub(f"{name}/sd_after").items():
            assert rel_err(after[k], v) < 2e-2, k


@pytest.mark.parametrize("train", [True, False])
def test_meta_kernel_forward(golden, train):
    from oracle import model as om
    from range_view_3d_detection_amd.nn.stems import MetaKernel

    g = golden("meta_kernel")
    sd = g.sub("meta/sd")
    m = _load(MetaKernel(5, 16, 3, 2), sd)
    m.train(train)
    f, c = g["meta/in0"], g["meta/in1"]
    with torch.no_grad():
        out = m(f.to(DEV), c.to(DEV)).float()
    ref = g["meta/out"] if train else g["meta/out_eval"]
    assert rel_err(out, ref) < 5e-2


def test_range_partition_operand_is_exact():
    """rv_range_partition: the banded, masked 16-bit operand of the RangePartition stem equals bf16(the reference's expression) element for
    element -- returns exactly ON the closed band edges (10, 15, 20, 30, 40, 45, 60 m) included, 6 input channels (rv-waymo), a masked
    pixel on an edge, and the padding channels zero."""
    import ctypes

    from range_view_3d_detection_amd import _lib as L

    gen = torch.Generator().manual_seed(5)
    N, C, H, W = 2, 6, 5, 48
    feats = torch.randn(N, C, H, W, generator=gen)
    cart = torch.randn(N, 3, H, W, generator=gen) * 25
    mask = torch.rand(N, 1, H, W, generator=gen) > 0.2
    for i, d in enumerate((0.0, 10.0, 15.0, 20.0, 30.0, 40.0, 45.0, 60.0, 15.0)):
        cart[1, :, 2, 4 * i] = torch.tensor([0.0, -d, 0.0])
        mask[1, 0, 2, 4 * i] = i != 8  # (the last one: on an edge but without a return)
    lower = torch.as_tensor((0, 10, 15, 20, 30, 45)).view(1, -1, 1, 1)
    upper = torch.as_tensor((15, 20, 30, 40, 60, torch.inf)).view(1, -1, 1, 1)
    dists = cart.norm(dim=1, keepdim=True)
    want = ((torch.logical_and(dists >= lower, dists <= upper)[:, :, None] * feats[:, None]).flatten(1, 2) * mask).to(torch.bfloat16)
    ld = 64
    out = torch.full((N, H, W, ld), 7.0, dtype=torch.bfloat16, device=DEV)
    f_d, c_d, m_d = feats.to(DEV), cart.to(DEV), mask.to(DEV).view(torch.uint8)
    lo = (ctypes.c_float * 6)(*[float(v) for v in lower.flatten()])
    hi = (ctypes.c_float * 6)(*[float(v) for v in upper.flatten()])
    L.call("rv_range_partition", L.ptr(f_d), L.ptr(c_d), L.ptr(m_d), L.i32(N), L.i32(C), L.i32(H), L.i32(W), lo, hi, L.i32(6), L.ptr(out), L.i32(ld), L.stream_ptr())
    got = out.cpu()
    assert torch.equal(got[..., : 6 * C].permute(0, 3, 1, 2).contiguous().view(torch.int16), want.contiguous().view(torch.int16))
    assert not got[..., 6 * C :].float().any()


@pytest.mark.parametrize("tag", ["k1", "k3"])
@pytest.mark.parametrize("train", [True, False])
def test_range_partition_stem_forward(golden, tag, train):
    """RangePartition (nn/stems/__init__.py:88-135) through the HIP path against the oracle with bf16 storage emulation and against the
    reference's own fp32 output; running statistics updated as nn.BatchNorm2d does."""
    from oracle import model as om
    from range_view_3d_detection_amd.nn.stems import RangePartition

    g = golden("range_partition")
    sd = g.sub(f"{tag}/sd")
    m = _load(RangePartition(5, 16, 3, int(tag[1])), sd)
    m.train(train)
    f, c, k = g["features"], g["cart"], g["mask"]
    with torch.no_grad():
        out = m(f.to(DEV), c.to(DEV), k.to(DEV)).float()
        exp = om.range_partition(f, c, k, {f"m.{n}": v for n, v in sd.items()}, "m", om.Numerics.bf16(train=train))
    assert rel_err(out, exp) < 1.5e-2, "vs oracle with bf16 storage emulation"
    assert rel_err(out, g[f"{tag}/out"] if train else g[f"{tag}/out_eval"]) < 4e-2, "vs the reference's fp32 output"
    if train:
        after = {n: v for n, v in m.state_dict().items() if "running_" in n}
        for n, v in g.sub(f"{tag}/sd_after").items():
            assert rel_err(after[n], v) < 2e-2, n


def test_range_net_dispatches_to_the_range_partition_stem(golden):
    """RangeNet(stem_type="RANGE_PARTITION") (nn/backbones/dla.py:164-171, 200-201: the stem reads x["mask"]) in eval mode against the
    reference's feature maps at all four strides."""
    from range_view_3d_detection_amd.nn.backbones.dla import RangeNet

    g = golden("range_partition")
    sd = g.sub("net/sd")
    C = 16
    net = RangeNet(in_channels=5, layers=[C] * 5, out_channels=C, projection_kernel_size=1, dataset_name="av2", num_neighbors=3, num_layers=2,
                   stem_type="RANGE_PARTITION", _net={"_target_": "torchbox3d.nn.backbones.dla.RangeBackbone", "in_channels": 5, "layers": [C] * 5, "out_channels": C})
    net = _load(net, sd).eval()
    with torch.no_grad():
        out = net({"features": g["features"].to(DEV), "cart": g["cart"].to(DEV), "mask": g["mask"].to(DEV)})
    for s_, ref in g.sub("net/eval_feat").items():
        assert rel_err(out[int(s_)].float(), ref) < 5e-2, s_
    with pytest.raises(Exception, match="mask"):
        net({"features": g["features"].to(DEV), "cart": g["cart"].to(DEV)})


# ---------------------------------------------------------------------------------------------
# decode / projection
# ---------------------------------------------------------------------------------------------
def test_decode_range_view_and_candidates(golden):
    from range_view_3d_detection_amd.math.ops.coding import decode_range_view
    from range_view_3d_detection_amd.nn.decoders.range_decoder import RangeDecoder

    g = golden("decode")
    reg, cart, mask, logits = (g[k].to(DEV) for k in ("regressands", "cart", "mask", "logits"))
    assert rel_err(decode_range_view(reg, cart, True), g["decoded_inv"]) < 1e-5
    assert rel_err(decode_range_view(reg, cart, False), g["decoded_plain"]) < 1e-5
    post = {"num_pre_nms": 50000, "num_post_nms": 1000, "nms_threshold": 0.3, "min_confidence": 0.1, "nms_mode": "WEIGHTED"}
    mo = {1: {"cart": cart, "mask": mask, 0: {"logits": logits, "regressands": reg}}}
    for sample, tag in ((True, "dec"), (False, "dense")):
        dec = RangeDecoder(True, sample, [0, 15, 30], [15, 30, math.inf], [8, 2, 1])
        p, s, c, b = dec.decode(mo, post, {0: ["c"] * 7}, use_nms=False)
        assert torch.equal(c.cpu(), g[f"{tag}_categories"]) and torch.equal(b.cpu(), g[f"{tag}_batch_index"])
        assert rel_err(p, g[f"{tag}_params"]) < 1e-5
        assert rel_err(s, g[f"{tag}_scores"]) < 1e-6


def test_projection_bit_exact(golden):
    from oracle import project as oproj
    from range_view_3d_detection_amd.math import range_view as rv

    g = golden("projection")
    cart = g["cart"].to(DEV)
    laser = g["laser_numbers"].to(DEV)
    mapping = g["row_mapping_64"].to(DEV)
    feats = g["features"].to(DEV)
    H, W = g.np("image_converter").shape[1:]
    for variant in ("converter", "library"):
        rows, cols, rng = rv.range_view_indices(cart, laser, mapping, H, W, variant)
        ref_idx = g.np(f"indices_{variant}")
        rows_c, cols_c = rows.cpu().numpy(), cols.cpu().numpy()
        assert np.array_equal(rows_c, ref_idx[0])
        # integer work: bit-exact, including the fixture's 112 exact half-bin azimuths (round-half-to-even)
        assert np.array_equal(cols_c, ref_idx[1]), int((cols_c != ref_idx[1]).sum())
        # the range is the C library's hypot, restated on the device: the reference's own bits
        assert np.array_equal(rng.cpu().numpy(), g.np(f"hybrid_{variant}")[:, 2])
        # z-buffer on the REFERENCE's indices: image and ownership bit-exact
        r_t, c_t = torch.from_numpy(ref_idx[0]).to(DEV), torch.from_numpy(ref_idx[1]).to(DEV)
        dist = torch.from_numpy(g.np(f"hybrid_{variant}")[:, 2]).to(DEV)
        image, winner = rv.z_buffer(r_t, c_t, dist, feats, H, W)
        assert np.array_equal(image.cpu().numpy(), g.np(f"image_{variant}")), variant
        _, win_o = oproj.z_buffer(ref_idx[0], ref_idx[1], g.np(f"hybrid_{variant}")[:, 2], g.np("features"), H, W)
        assert np.array_equal(winner.cpu().numpy(), win_o)


def test_atan2_correctly_rounded_and_bins_on_a_million_points():
    """Device azimuth == THE correctly rounded fp64 atan2 (witness: oracle.project.atan2_cr), bit for bit, on random,
    near-axis, tiny-ratio and extreme-exponent inputs; columns of 1.2e6 random points == the oracle's numpy binning."""
    from oracle import project as oproj
    from range_view_3d_detection_amd.math import range_view as rv

    rng = np.random.default_rng(7)
    n = 1_200_000
    x, y = rng.normal(size=n) * 50, rng.normal(size=n) * 50
    x[:2000] = rng.uniform(1, 100, 2000)
    y[:2000] = x[:2000] * 2.0 ** rng.uniform(-60, -20, 2000) * rng.choice([-1, 1], 2000)  # tiny angles (series branch + its edge)
    x[2000:3000] = -rng.uniform(1, 100, 1000)
    y[2000:3000] = 2.0 ** rng.uniform(-80, -10, 1000)  # just below +pi
    y[3000:4000] = rng.uniform(1, 100, 1000)
    x[3000:4000] = 2.0 ** rng.uniform(-80, -10, 1000) * rng.choice([-1, 1], 1000)  # around pi/2
    x[4000:4500] *= 2.0 ** 300
    y[4000:4500] *= 2.0 ** 300
    x[4500:5000] *= 2.0 ** -300
    y[4500:5000] *= 2.0 ** -300
    x[5000:5010] = [0.0, -0.0, 1.0, -1.0, 0.0, -0.0, 3.0, -3.0, np.inf, -np.inf]
    y[5000:5010] = [0.0, 0.0, 0.0, 0.0, 2.0, -2.0, -0.0, -0.0, 1.0, 1.0]
    az = rv.atan2_cr(torch.from_numpy(y).to(DEV), torch.from_numpy(x).to(DEV)).cpu().numpy()
    ref = oproj.atan2_cr(y, x)
    assert np.array_equal(az.view(np.int64), ref.view(np.int64)), int((az.view(np.int64) != ref.view(np.int64)).sum())
    # the range: np.hypot's bits (glibc's corrected sqrt, which is NOT the correctly rounded value), also on huge / tiny /
    # lopsided / non-finite operands where glibc rescales or returns ax + ay
    hx, hy = x.copy(), y.copy()
    hx[5010:5500] = rng.normal(size=490) * 2.0 ** rng.integers(480, 1020, 490)
    hy[5010:5500] = rng.normal(size=490) * 2.0 ** rng.integers(400, 1020, 490)
    hx[5500:6000] = rng.normal(size=490 + 10) * 2.0 ** rng.integers(-1070, -400, 500)
    hy[5500:6000] = rng.normal(size=500) * 2.0 ** rng.integers(-1074, -440, 500)
    hy[6000:6500] = hx[6000:6500] * 2.0 ** rng.integers(-56, -50, 500)
    hx[6500:6506] = [np.inf, np.nan, np.nan, -np.inf, 0.0, 5e-324]
    hy[6500:6506] = [np.nan, np.inf, 1.0, 2.0, -0.0, 5e-324]
    with np.errstate(all="ignore"):
        h_ref = np.hypot(hx, hy)
    h = rv.hypot_libc(torch.from_numpy(hx).to(DEV), torch.from_numpy(hy).to(DEV)).cpu().numpy()
    assert np.array_equal(h.view(np.int64)[~np.isnan(h_ref)], h_ref.view(np.int64)[~np.isnan(h_ref)]), int((h != h_ref).sum())
    assert np.array_equal(np.isnan(h), np.isnan(h_ref))
    # binning of sensor-frame points: columns of both variants equal the oracle's (numpy) result
    cart = np.stack([x, y, rng.normal(size=n) * 3], axis=1)
    cart[4000:5010] = rng.normal(size=(1010, 3)) * 30
    laser = rng.integers(0, 64, n)
    mapping = rng.permutation(64)
    H, W = 64, 2048
    sph = oproj.cart_to_sph(cart)
    for variant in ("converter", "library"):
        r_o, c_o, d_o = oproj.range_view_indices(sph, laser, mapping, H, W, variant)
        r, c, d = rv.range_view_indices(torch.from_numpy(cart).to(DEV), torch.from_numpy(laser).to(DEV), torch.from_numpy(mapping).to(DEV), H, W, variant)
        assert np.array_equal(r.cpu().numpy(), r_o) and np.array_equal(c.cpu().numpy(), c_o), variant
        assert np.array_equal(d.cpu().numpy(), d_o), variant


def test_z_buffer_fp64_vs_fp32_quirk():
    """Later point whose fp64 range is below the fp32-rounded stored range still replaces the owner."""
    from oracle import project as oproj
    from range_view_3d_detection_amd.math import range_view as rv

    base = float(np.float32(10.0) + np.float32(2.0) ** -20)  # exactly representable in fp32
    d = np.array([base + 1e-9, base - 1e-9 + 2e-9 * 0 + 4e-10, 0.5, base + 3e-10, 25.0, base - 1e-7], dtype=np.float64)
    # all but the last round to `base` in fp32; index 1 and 3 are both < fp32(base+1e-9)?  compare with the oracle
    rows = np.zeros(6, dtype=np.int64)
    cols = np.array([3, 3, 3, 3, 3, 5], dtype=np.int64)
    feats = np.arange(12, dtype=np.float64).reshape(2, 6)
    img_o, win_o = oproj.z_buffer(rows, cols, d, feats, 1, 8)
    img, win = rv.z_buffer(torch.from_numpy(rows).to(DEV), torch.from_numpy(cols).to(DEV), torch.from_numpy(d).to(DEV),
                           torch.from_numpy(feats).to(DEV), 1, 8)
    assert np.array_equal(win.cpu().numpy(), win_o) and np.array_equal(img.cpu().numpy(), img_o)
    rng = np.random.default_rng(0)
    n = 5000
    d = (5.0 + rng.integers(0, 4, n) * 1e-7 + rng.uniform(-3e-7, 3e-7, n)).astype(np.float64)
    d[rng.integers(0, n, 50)] = 0.3
    rows = rng.integers(0, 4, n)
    cols = rng.integers(0, 16, n)
    feats = rng.normal(size=(3, n))
    img_o, win_o = oproj.z_buffer(rows, cols, d, feats, 4, 16)
    img, win = rv.z_buffer(torch.from_numpy(rows).to(DEV), torch.from_numpy(cols).to(DEV), torch.from_numpy(d).to(DEV),
                           torch.from_numpy(feats).to(DEV), 4, 16)
    assert np.array_equal(win.cpu().numpy(), win_o) and np.array_equal(img.cpu().numpy(), img_o)


@pytest.mark.parametrize("case", ["no_points", "all_below_min_distance", "one_pixel", "exact_ties", "seam_and_origin"])
def test_projection_degenerate_sweeps(case):
    """The sweeps a converter can be handed at the edges: none at all, every return inside the 1 m ego mask, every return in one
    pixel, equal ranges (the earliest point keeps the pixel) and points on the azimuth seam / at the origin -- image, ownership and
    bins equal the oracle's sequential z-buffer, bit for bit."""
    from oracle import project as oproj
    from range_view_3d_detection_amd.math import range_view as rv

    rng = np.random.default_rng(11)
    H, W, C = 8, 64, 4
    mapping = rng.permutation(H)
    if case == "no_points":
        cart = np.zeros((0, 3))
    elif case == "all_below_min_distance":
        cart = rng.normal(size=(500, 3))
        cart *= rng.uniform(0.0, 0.999, (500, 1)) / np.linalg.norm(cart, axis=1, keepdims=True)
    elif case == "one_pixel":
        cart = np.array([[7.0, 0.01, 0.2]]) * rng.uniform(1.0, 6.0, (3000, 1))
    elif case == "exact_ties":
        cart = np.repeat(rng.normal(size=(40, 3)) * 20, 25, axis=0)  # 25 copies of each point: index order decides
    else:
        cart = np.array([[-5.0, 0.0, 0.1], [-5.0, -0.0, 0.1], [-5.0, 1e-300, 0.0], [-5.0, -1e-300, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 4.0],
                         [0.0, 3.0, 0.0], [0.0, -3.0, 0.0], [5.0, 0.0, 0.0], [-0.0, 0.0, 2.0], [1e-200, 1e-200, 3.0]])
    n = len(cart)
    laser = np.zeros(n, dtype=np.int64) if case == "one_pixel" else rng.integers(0, H, n)
    if case == "exact_ties":
        laser = np.repeat(laser[::25], 25)
    feats = rng.normal(size=(C, n))
    sph = oproj.cart_to_sph(cart)
    for variant in ("converter", "library"):
        r_o, c_o, d_o = oproj.range_view_indices(sph, laser, mapping, H, W, variant)
        r, c, d = rv.range_view_indices(torch.from_numpy(cart).to(DEV), torch.from_numpy(laser).to(DEV), torch.from_numpy(mapping).to(DEV), H, W, variant)
        assert np.array_equal(r.cpu().numpy(), r_o) and np.array_equal(c.cpu().numpy(), c_o), (case, variant)
        assert np.array_equal(d.cpu().numpy(), d_o)
        img_o, win_o = oproj.z_buffer(r_o, c_o, d_o, feats, H, W)
        img, win = rv.build_range_view(torch.from_numpy(cart).to(DEV), torch.from_numpy(feats).to(DEV), torch.from_numpy(laser).to(DEV),
                                       torch.from_numpy(mapping).to(DEV), H, W, variant)
        assert np.array_equal(win.cpu().numpy(), win_o.reshape(H, W)), (case, variant)
        assert np.array_equal(img.cpu().numpy(), img_o.reshape(C, H, W)), (case, variant)
        if case in ("no_points", "all_below_min_distance"):
            assert (win_o == -1).all() and not img_o.any()
        if case == "one_pixel":
            assert (win_o >= 0).sum() == 1


def test_device_loader_item_matches_the_reference(golden, tmp_path):
    """``range_view_from_table`` (rv_table_to_range_view + the padding kernel) against the reference's own ``__getitem__`` output:
    everything but tanh(intensity) bit-exact (copies, 0/1 products, mask, circular / constant padding), tanh within 2 fp32 ulps
    (device tanhf vs numpy).  Also through a feather file written and read back (the on-disk form of the contract)."""
    import pyarrow as pa

    from range_view_3d_detection_amd.prototype import loader as ld
    from test_oracle_golden import _loader_case

    g = golden("loader_item")
    row_map = golden("raw_sweep").np("tables/ROW_MAPPING_64")
    for tag, ds in (("av2", "av2"), ("waymo", "waymo"), ("av2_view", "av2")):  # av2_view: the `view` feature (loader.py:605-624)
        names, table, roi, mode = _loader_case(g, tag)
        cfg = {"feature_column_names": names, "filter_roi": roi, "height": 8, "width": 64, "row_mapping_64": row_map}
        path = tmp_path / f"{tag}.feather"
        with pa.OSFile(str(path), "wb") as sink:
            t = pa.table(table)
            with pa.ipc.new_file(sink, t.schema) as w:
                w.write_table(t)
        for src in (table, ld.read_sweep_table(path)):
            got = ld.range_view_from_table(src, cfg, ds, 1, mode, device=DEV)
            assert got["mask"].dtype == torch.bool and torch.equal(got["mask"].cpu(), torch.from_numpy(g.np(f"{tag}/mask")))
            assert torch.equal(got["cart"].cpu(), torch.from_numpy(g.np(f"{tag}/cart")))
            ref = torch.from_numpy(g.np(f"{tag}/features"))
            for i, n in enumerate(names):
                if n == "intensity" and ds == "waymo":
                    assert float((got["features"][i].cpu() - ref[i]).abs().max()) <= 2.4e-7 * float(ref[i].abs().max()), (tag, n)
                else:
                    assert torch.equal(got["features"][i].cpu(), ref[i]), (tag, n)
    with pytest.raises(Exception):
        ld.range_view_from_table(table, cfg, ds, 1, mode, device="cpu")  # no CPU fallback


def test_device_augmentations_match_the_reference(golden):
    """Loader augmentations on device (prototype/loader.py of this package -> rv_augment) against fixtures produced by the
    reference's own functions: pixel placement (every non-geometry channel) bit-exact, geometry 1e-6 of the channel
    maximum, boxes 1e-9; and the seeded ``random`` draws pick the same parameters as the reference."""
    import random

    from oracle import augment as oaug
    from range_view_3d_detection_amd.prototype import loader as ld

    g = golden("augment")
    names = [str(n) for n in g.np("column_names")]
    s0 = torch.from_numpy(g.np("sweep/in")).float()
    a0 = torch.from_numpy(g.np("ann/in")).T.contiguous()  # (M, 10)
    ann = torch.cat([a0, torch.zeros(a0.shape[0], 3, dtype=torch.float64)], dim=1)  # + task_id, offset, batch_index
    H, W = s0.shape[1:]
    cart0 = s0[[names.index(n) for n in ("x", "y", "z")]]
    batch = {"features": s0[None].to(DEV), "cart": cart0[None].to(DEV), "mask": (s0[names.index("range")] > 0)[None, None].to(DEV),
             "annotations": ann}

    def check(out, tag):
        ref_s, ref_a = g.np(f"{tag}/sweep"), g.np(f"{tag}/ann")
        got = out["features"][0].cpu().numpy()
        for i, n in enumerate(names):
            if n not in ("x", "y", "z", "range"):
                assert np.array_equal(got[i], ref_s[i].astype(np.float32)), (tag, n)
            assert np.max(np.abs(got[i] - ref_s[i])) <= 1e-6 * max(1.0, np.max(np.abs(ref_s[i]))), (tag, n)
        gc = out["cart"][0].cpu().numpy()
        for j, n in enumerate(("x", "y", "z")):
            assert np.max(np.abs(gc[j] - ref_s[names.index(n)])) <= 1e-6 * max(1.0, np.max(np.abs(ref_s[names.index(n)]))), (tag, n)
        # the mask is what the reference derives from the AUGMENTED table (range > 0, loader.py:645-652): it travels with the
        # pixels, except that random_global_scale re-derives the range from the coordinates (the fixture's synthetic range is
        # negative at a few pixels: those become valid after a scale, in the reference and here)
        assert np.array_equal(out["mask"][0, 0].cpu().numpy(), ref_s[names.index("range")] > 0), tag
        tr = out["transforms"][0]
        if not tr.use_range:
            ws = (tr.a * np.arange(W) + tr.b) % W
            assert np.array_equal(out["mask"][0, 0].cpu().numpy(), (g.np("sweep/in")[names.index("range")] > 0)[:, ws]), tag
        ga = out["annotations"].numpy()[:, :10].T
        assert np.max(np.abs(ga[:6] - ref_a[:6])) <= 1e-9 * max(1.0, np.max(np.abs(ref_a[:6]))), tag
        dyaw = oaug.yaw_of(ga[6:10]) - oaug.yaw_of(ref_a[6:10])
        assert np.max(np.abs(np.arctan2(np.sin(dyaw), np.cos(dyaw)))) < 1e-9, tag

    rot_cfg = {"low": -0.78539816, "high": 0.78539816, "p": 1.0}
    random.seed(1)
    check(ld.augment_batch(batch, names, {"flip_azimuth": {"p": 1.0}}), "flip")
    random.seed(2)
    out = ld.augment_batch(batch, names, {"random_rotation": rot_cfg})
    assert out["transforms"][0].ops[0][1] == float(g.np("rotation/theta"))  # same draw as the reference
    check(out, "rotation")
    random.seed(5)
    check(ld.augment_batch(batch, names, {"random_rotation": {"low": -3.0, "high": -2.0, "p": 1.0}}), "rotation_neg")
    random.seed(3)
    check(ld.augment_batch(batch, names, {"random_global_scale": {"low": 0.95, "high": 1.05}}), "scale")
    random.seed(4)
    check(ld.augment_batch(batch, names, {"random_global_translation": {"std_x": 0.5, "std_y": 0.5, "std_z": 0.2}}), "translation")
    random.seed(6)
    chain = {"flip_azimuth": {"p": 1.0}, "random_rotation": rot_cfg, "random_global_scale": {"low": 0.95, "high": 1.05},
             "random_global_translation": {"std_x": 0.5, "std_y": 0.5, "std_z": 0.2}}
    check(ld.augment_batch(batch, names, chain), "chain")
    # a batch of two sweeps: each sweep gets its own draws; the second equals what a single-sweep call with those draws gives
    random.seed(9)
    two = {"features": torch.cat([batch["features"]] * 2), "cart": torch.cat([batch["cart"]] * 2), "mask": torch.cat([batch["mask"]] * 2),
           "annotations": torch.cat([ann, torch.cat([ann[:, :-1], torch.ones(ann.shape[0], 1, dtype=torch.float64)], dim=1)])}
    out2 = ld.augment_batch(two, names, chain)
    t0, t1 = out2["transforms"]
    assert t0.ops != t1.ops
    so, ao = g.np("sweep/in"), g.np("ann/in")
    for op in t1.ops:
        so, ao = {"flip": lambda s, a: oaug.flip(s, names, a), "rotate": lambda s, a: oaug.rotate(s, names, a, op[1]),
                  "scale": lambda s, a: oaug.scale(s, names, a, op[1]), "translate": lambda s, a: oaug.translate(s, names, a, op[1])}[op[0]](so, ao)
    got = out2["features"][1].cpu().numpy()
    for i, n in enumerate(names):
        assert np.max(np.abs(got[i] - so[i])) <= 1e-6 * max(1.0, np.max(np.abs(so[i]))), n
    # p = 0: nothing happens and nothing but the coin is drawn
    random.seed(1)
    same = ld.augment_batch(batch, names, {"flip_azimuth": {"p": 0.0}})
    assert torch.equal(same["features"], batch["features"]) and torch.equal(same["mask"], batch["mask"])


def test_raw_sweep_path_on_device(golden):
    """unmotion_compensate -> correct_laser_numbers -> build_range_view on device against the fixtures the reference's own
    converter functions produced: kept mask and rows exact, un-compensated points 1e-9 (fp64 rigid transforms), the final
    range image bit for bit."""
    from range_view_3d_detection_amd.converters.av2 import utils as cu

    g = golden("raw_sweep")
    t = lambda name: torch.from_numpy(g.np(name))
    xyz = t("sweep/xyz").to(DEV)
    xyz_p, kept = cu.unmotion_compensate(xyz, t("sweep/offset_ns"), int(g.np("sweep/timestamp_ns")), t("poses/timestamp_ns"), t("poses/q_wxyz"), t("poses/t"))
    assert np.array_equal(kept.cpu().numpy(), g.np("unmotion/kept"))
    ref = g.np("unmotion/xyz_p")
    got = xyz_p[kept].cpu().numpy()
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))
    lz = t("sweep/laser_number").to(DEV)
    rows64 = cu.correct_laser_numbers(lz, True, t("tables/LASER_MAPPING"), t("tables/ROW_MAPPING_64"))
    assert np.array_equal(rows64[kept].cpu().numpy(), g.np("laser/h64_affected"))
    assert np.array_equal(cu.correct_laser_numbers(lz, False, t("tables/LASER_MAPPING"), t("tables/ROW_MAPPING_64"))[kept].cpu().numpy(), g.np("laser/h64_plain"))
    assert np.array_equal(cu.correct_laser_numbers(lz % 32, True, t("tables/LASER_MAPPING"), t("tables/ROW_MAPPING_32"))[kept].cpu().numpy(), g.np("laser/h32_affected"))
    feats = torch.stack([xyz[:, 0], xyz[:, 1], xyz[:, 2], t("sweep/intensity").to(DEV).double(), rows64.double(), t("sweep/is_within_roi").to(DEV).double()], dim=1)
    # the un-compensated points of the reference (so that a 1e-16 difference in a rigid transform cannot move a bin): the
    # kept points take the fixture's values, dropped ones stay in place with range 0
    xyz_ref = xyz_p.clone()
    xyz_ref[kept] = t("unmotion/xyz_p").to(DEV)
    for pts in (xyz_ref, xyz_p):
        img = cu.build_range_view(pts, kept, feats, rows64, t("sweep/offset_ns"), t("extrinsics/q_wxyz"), t("extrinsics/t"), 64, 512)
        got = img.cpu().numpy().astype(np.float64)
        got[3], got[4], got[5] = got[3].astype(np.uint8), got[4].astype(np.uint8), got[5] != 0  # RANGE_VIEW_SCHEMA casts (utils.py:16-25)
        want = g.np("range_view/image")
        assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.parametrize("C", [256, 128])
@pytest.mark.parametrize("P", [999, 41472])
def test_pos_forward_kernel_vs_fp32(P, C):
    """rv_pos_forward (both positional layers of the MetaKernel stem in one persistent streaming GEMM: the first layer is
    generated in the second one's operand staging) against fp32 torch ops: h1 one bf16 rounding of relu(s1 (W1 rel) + t1)
    (4e-3 of max), y2 = W2 h1 from the kernel's own bf16 h1 one bf16 rounding (4e-3), the (sum, sum of squares) rows of the
    fp32 accumulators 1e-4 of their scale.  P = 999: one partial step per workgroup; 41472: 324 steps over 256 persistent
    workgroups (both LDS images, the generate-next-while-multiplying path).  C = 256 is rv-av2's stem, C = 128 rv-waymo's (256-pixel
    steps, two waves per channel slice: 162 steps, so some workgroups take one step and the rest none)."""
    from range_view_3d_detection_amd import _lib as L

    gen = torch.Generator().manual_seed(P)
    rel = torch.zeros(P, 32, dtype=torch.bfloat16)
    rel[:, :3] = (torch.randn(P, 3, generator=gen) * 2).to(torch.bfloat16)
    w1 = torch.zeros(C, 32, dtype=torch.bfloat16)
    w1[:, :3] = torch.randn(C, 3, generator=gen).to(torch.bfloat16)
    w2 = (torch.randn(C, C, generator=gen) / 16).to(torch.bfloat16)
    s1, t1 = 0.5 + torch.rand(C, generator=gen), 0.3 * torch.randn(C, generator=gen)
    rel, w1, w2, s1, t1 = (x.to(DEV) for x in (rel, w1, w2, s1, t1))
    h1 = torch.full((P, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    y2 = torch.full((P, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    rows = L.load().rv_pos_forward_rows(L.i64(P))
    partial = torch.zeros((rows + L.STATS_SCRATCH_ROWS, 2, C), dtype=torch.float32, device=DEV)
    L.call("rv_pos_forward", L.ptr(rel), L.i32(32), L.i32(3), L.i64(P), L.ptr(w1), L.i32(32), L.ptr(s1), L.ptr(t1), L.ptr(w2), L.i32(C),
           L.ptr(h1), L.ptr(y2), L.ptr(partial), L.stream_ptr())
    torch.cuda.synchronize()
    want_h1 = torch.relu((rel[:, :3].float() @ w1[:, :3].float().t()) * s1 + t1)
    assert rel_err(h1.float(), want_h1) < 4e-3, rel_err(h1.float(), want_h1)
    want_y2 = h1.float() @ w2.float().t()
    assert rel_err(y2.float(), want_y2) < 4e-3, rel_err(y2.float(), want_y2)
    got = partial[:rows].double().sum(0)
    scale = float(want_y2.abs().sum(0).max())
    assert float((got[0] - want_y2.double().sum(0)).abs().max()) < 1e-4 * scale
    assert float((got[1] - (want_y2.double() ** 2).sum(0)).abs().max()) < 1e-4 * float((want_y2.double() ** 2).sum(0).max())


@pytest.mark.parametrize("C", [256, 128])
def test_meta_kernel_positional_pair_fused_matches_unfused(C):
    """MetaKernel at the rv-av2 (C = 256) and rv-waymo (C = 128) stem widths: rv_pos_forward against the SmallKOp + 1x1 tap-conv pair it replaces --
    output and running statistics within bf16 rounding (2e-2 / 2e-3 of max), every parameter gradient (backward is shared; bf16
    roundings of h1 / y2 differ in the last place and flip a few ReLU gates) cosine 0.9999, 4e-2 of max."""
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd.nn.stems import MetaKernel

    gen = torch.Generator().manual_seed(21)
    m = MetaKernel(5, C, 3, 2).to(DEV).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    feats = torch.randn(1, 5, 16, 160, generator=gen).to(DEV)
    cart = (torch.randn(1, 3, 16, 160, generator=gen) * 5).to(DEV)
    probe = torch.randn(1, C, 16, 160, generator=gen).to(DEV)

    def run(fused: bool):
        E.POS_FUSE = fused
        try:
            m.load_state_dict(sd)
            m.zero_grad(set_to_none=True)
            out = m(feats, cart).float()
            (out * probe).sum().backward()
            stats = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if "running_" in k}
            return out.detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}, stats
        finally:
            E.POS_FUSE = True

    out_a, ga, st_a = run(True)
    out_b, gb, st_b = run(False)
    assert rel_err(out_a, out_b) < 2e-2, rel_err(out_a, out_b)
    for k in st_a:
        assert rel_err(st_a[k], st_b[k]) < 2e-3, (k, rel_err(st_a[k], st_b[k]))
    for k in ga:
        cos = float(torch.nn.functional.cosine_similarity(ga[k].flatten().double(), gb[k].flatten().double(), dim=0))
        assert cos > 0.9999 and rel_err(ga[k], gb[k]) < 4e-2, (k, cos, rel_err(ga[k], gb[k]))
