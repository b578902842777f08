"""GPU parity of the backward HIP path (input / weight / BatchNorm gradients).

Tolerances:
* weight-gradient kernel with bf16-representable operands (fp32 result): 2e-5 of max -- the
  products are exact, only the summation order differs;
* input-gradient kernel (bf16 result): one bf16 ulp, 4e-3 of max;
* block-level gradients vs the fp32 autograd of the oracle with bf16 storage emulation: 3e-2 of
  max (activation *gradients* are stored as bf16 between kernels, which the oracle does not
  emulate); vs the reference's own fp32 gradients: 5e-2 of max.
"""

from __future__ import annotations

import ctypes

import pytest
import torch
import torch.nn.functional as F

from test_gpu_forward import DEV, _load, _module_cases, bf16r, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize(
    "cin,cout,k,stride,H,W",
    [(40, 24, 3, 1, 5, 48), (32, 32, 3, 2, 4, 64), (64, 48, 1, 2, 3, 96), (128, 160, 3, 1, 4, 128), (256, 128, 3, 1, 2, 256),
     (64, 64, 3, 2, 5, 256), (128, 128, 1, 2, 4, 256), (64, 128, 1, 1, 3, 128),  # strided layers: bwd-data = tapconv2 scatter phases
     (128, 128, 3, 1, 3, 200), (64, 96, 3, 1, 4, 136), (256, 256, 1, 1, 2, 226)],  # widths that are not multiples of the 64-pixel chunk: wgrad3 (DMA), wgrad2, 1x1
)
def test_conv_input_and_weight_grad(cin, cout, k, stride, H, W):
    from range_view_3d_detection_amd.nn.modules.conv import Conv2dSame
    from oracle import model as om

    g = torch.Generator().manual_seed(cin + 7 * cout)
    m = Conv2dSame(cin, cout, kernel_size=k, stride=(1, stride), bias=False)
    m.conv.weight.data = bf16r(torch.randn(m.conv.weight.shape, generator=g) * 0.2)
    x = bf16r(torch.randn(2, cin, H, W, generator=g)).requires_grad_(True)
    w = m.conv.weight.data.clone().requires_grad_(True)
    y = om.conv2d_same(x, w, (1, stride))
    probe = bf16r(torch.randn(y.shape, generator=g))
    (y * probe).sum().backward()

    m = m.to(DEV)
    xd = x.detach().to(DEV).requires_grad_(True)
    yd = m(xd)
    # stored bf16 results: one ulp.  A bf16 ulp is 2^-8 .. 2^-7 of the value, so relative to the tensor's maximum a
    # single value that rounds to the neighbouring bf16 shows up as at most 7.8e-3.
    assert rel_err(yd.float(), bf16r(y.detach())) < 8e-3
    (yd.float() * probe.to(DEV)).sum().backward()
    assert rel_err(m.conv.weight.grad, w.grad) < 2e-5
    assert rel_err(xd.grad, bf16r(x.grad)) < 8e-3


@pytest.mark.parametrize("kernel,stride,padding,W", [((3, 8), (1, 4), (1, 2), 24), ((3, 4), (1, 2), (1, 1), 80)])
def test_conv_transpose_grads(kernel, stride, padding, W):
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd import program

    g = torch.Generator().manual_seed(W)
    m = torch.nn.ConvTranspose2d(48, 40, kernel_size=kernel, stride=stride, padding=padding, bias=False)
    m.weight.data = bf16r(torch.randn(m.weight.shape, generator=g) * 0.2)
    x = bf16r(torch.randn(2, 48, 5, W, generator=g)).requires_grad_(True)
    w = m.weight.data.clone().requires_grad_(True)
    y = F.conv_transpose2d(x, w, stride=stride, padding=padding)
    probe = bf16r(torch.randn(y.shape, generator=g))
    (y * probe).sum().backward()

    m = m.to(DEV)

    def build(t, xin):
        a = E.Act.from_nchw(xin)
        return [a], [E.ConvOp(t, E.tap_layer(m), a).out]

    xd = x.detach().to(DEV).requires_grad_(True)
    yd = program.run(build, m, [xd])[0]
    assert rel_err(yd.float(), bf16r(y.detach())) < 4e-3
    (yd.float() * probe.to(DEV)).sum().backward()
    assert rel_err(m.weight.grad, w.grad) < 2e-5
    assert rel_err(xd.grad, bf16r(x.grad)) < 4e-3


def _cos(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_bn_backward_kernels_exact():
    """rv_bn_bwd_reduce / finalize / apply against the closed-form BatchNorm+ReLU backward on identical inputs."""
    from range_view_3d_detection_amd import _lib as L

    g = torch.Generator().manual_seed(5)
    n_px, c = 3000, 64
    dout = bf16r(torch.randn(n_px, c, generator=g))
    y = bf16r(torch.randn(n_px, c, generator=g) * 2 + 0.5)
    out = bf16r(torch.randn(n_px, c, generator=g))
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.3
    mean, var = y.mean(0), y.var(0, unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    for flags, use_out in ((0, True), (L.BNB_RELU_Z, False), (L.BNB_RELU_Z, True), (0, False)):
        gg = dout.double().clone()
        if use_out:
            gg = gg * (out > 0)
        if flags & L.BNB_RELU_Z:
            gg = gg * ((y * scale + shift) > 0)
        xhat = (y.double() - mean.double()) * invstd.double()
        s0, s1 = gg.sum(0), (gg * xhat).sum(0)
        dy_ref = (gamma * invstd).double() * (gg - s0 / n_px - xhat * s1 / n_px)

        dv = lambda t_, dt=None: t_.to(DEV) if dt is None else t_.to(DEV, dt)
        d_dout, d_y, d_out = dv(dout, torch.bfloat16), dv(y, torch.bfloat16), dv(out, torch.bfloat16)
        d_sc, d_sh, d_mu, d_is, d_ga = dv(scale), dv(shift), dv(mean), dv(invstd), dv(gamma)
        rows = L.load().rv_bn_bwd_rows(L.i64(n_px))
        partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, c), dtype=torch.float32, device=DEV)
        common = (L.i64(n_px), L.i32(c), L.ptr(d_dout), L.i32(c), L.ptr(d_out) if use_out else None, L.i32(c), L.ptr(d_y), L.i32(c),
                  L.ptr(d_sc), L.ptr(d_sh), L.ptr(d_mu), L.ptr(d_is))
        L.call("rv_bn_bwd_reduce", *common, L.i32(flags), L.ptr(partial), L.stream_ptr())
        dgamma = torch.empty(c, device=DEV)
        dbeta = torch.empty(c, device=DEV)
        coef = torch.empty((3, c), device=DEV)
        L.call("rv_bn_bwd_finalize", L.ptr(partial), L.i32(rows), L.i32(c), L.i64(n_px), L.ptr(d_ga), L.ptr(d_is), L.ptr(dgamma),
               L.ptr(dbeta), L.i32(0), L.ptr(coef), L.stream_ptr())
        dy = torch.empty((n_px, c), dtype=torch.bfloat16, device=DEV)
        L.call("rv_bn_bwd_apply", *common, L.ptr(coef), L.i32(flags), L.ptr(dy), L.i32(c), None, L.i32(0), L.stream_ptr())
        assert rel_err(dbeta, s0) < 1e-5 and rel_err(dgamma, s1) < 1e-5
        assert rel_err(dy.float(), dy_ref) < 4e-3  # one bf16 ulp


def _smooth(sd):
    """Shift every BatchNorm bias so that all ReLUs are active: gradients become smooth in the inputs
    (no gate flips from bf16 rounding), which lets the tape's composition be checked tightly."""
    out = dict(sd)
    for k, v in sd.items():
        if k.endswith(".bias") and k[: -len(".bias")] + ".running_mean" in sd:
            out[k] = v + 8.0
    return out


@pytest.mark.parametrize("name", ["basic_plain", "basic_proj_s12", "basic_k1_proj", "residual_s12_n3", "agg_k8_s4", "agg_k4_s2"])
@pytest.mark.parametrize("smooth", [True, False])
def test_block_backward(golden, name, smooth):
    from oracle import model as om

    g = golden("conv_blocks")
    make, ofn, n_in = _module_cases()[name]
    sd = g.sub(f"{name}/sd")
    if smooth:
        sd = _smooth(sd)
    probe = g[f"{name}/probe"]
    # oracle with the HIP path's storage rounding points, fp32 autograd
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    full = {f"m.{k}": v for k, v in {**sd, **params}.items()}
    xs = [g[f"{name}/in{i}"].clone().requires_grad_(True) for i in range(n_in)]
    (ofn(*xs, full, om.Numerics.bf16(train=True)) * probe).sum().backward()

    m = _load(make(), sd).train()
    xd = [g[f"{name}/in{i}"].to(DEV).requires_grad_(True) for i in range(n_in)]
    out = m(*xd)
    (out.float() * probe.to(DEV)).sum().backward()
    pairs = [(f"gin{i}", xd[i].grad.float(), xs[i].grad, g[f"{name}/gin{i}"]) for i in range(n_in)]
    pairs += [(k, p.grad, params[k].grad, g[f"{name}/grad/{k}"]) for k, p in m.named_parameters()]
    for k, got, orc, ref in pairs:
        if smooth:
            # every ReLU active: the only noise is bf16 rounding of stored activations / gradients.  Gradients that
            # cancel analytically (a BatchNorm bias feeding a conv that is batch-normalised again is ~0) are sums
            # of n_px terms of size |probe| each carrying a bf16 rounding error => additive floor.
            n_px = probe.numel() // probe.shape[1]
            tol = 3e-2 * float(orc.abs().max()) + 4e-4 * n_px * float(probe.abs().max())
            err = float((got.detach().double().cpu() - orc.double()).abs().max())
            assert err < tol, (k, err, tol)
        else:
            # ReLU gates of values within one bf16 ulp of zero flip between the bf16 path and fp32 math; with only
            # 384 pixels per channel a flip moves a per-channel sum by several percent => statistical check
            assert _cos(got, orc) > 0.97 and _l2(got, orc) < 0.25, (k, _cos(got, orc), _l2(got, orc))
            assert _cos(got, ref) > 0.97 and _l2(got, ref) < 0.25, (k, _cos(got, ref), _l2(got, ref))


def test_meta_kernel_backward(golden):
    from range_view_3d_detection_amd.nn.stems import MetaKernel

    g = golden("meta_kernel")
    sd = g.sub("meta/sd")
    m = _load(MetaKernel(5, 16, 3, 2), sd).train()
    out = m(g["meta/in0"].to(DEV), g["meta/in1"].to(DEV))
    assert rel_err(out.float(), g["meta/out"]) < 5e-2
    (out.float() * g["meta/probe"].to(DEV)).sum().backward()
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        ref = g[f"meta/grad/{k}"]
        assert _cos(p.grad, ref) > 0.97 and _l2(p.grad, ref) < 0.25, (k, _cos(p.grad, ref), _l2(p.grad, ref))


@pytest.mark.parametrize("tag", ["k1", "k3"])
def test_range_partition_stem_backward(golden, tag):
    """RangePartition training step against the reference's parameter gradients (same bounds as the MetaKernel stem's: a 16-channel,
    randomly initialised stem in bf16); the frozen band edges receive no gradient."""
    from range_view_3d_detection_amd.nn.stems import RangePartition

    g = golden("range_partition")
    m = _load(RangePartition(5, 16, 3, int(tag[1])), g.sub(f"{tag}/sd")).train()
    out = m(g["features"].to(DEV), g["cart"].to(DEV), g["mask"].to(DEV))
    assert rel_err(out.float(), g[f"{tag}/out"]) < 5e-2
    (out.float() * g[f"{tag}/probe"].to(DEV)).sum().backward()
    assert m.lower_bounds.grad is None and m.upper_bounds.grad is None
    for k, p in m.named_parameters():
        if "bounds" in k:
            continue
        assert p.grad is not None, k
        ref = g[f"{tag}/grad/{k}"]
        assert _cos(p.grad, ref) > 0.97 and _l2(p.grad, ref) < 0.25, (k, _cos(p.grad, ref), _l2(p.grad, ref))


def test_small_k_fused_paths_match_unfused_and_oracle():
    """rv_smallk_forward (activated output in one element-wise pass, closed-form batch statistics) and rv_bn_bwd_smallk
    (BatchNorm backward + 1x1 weight gradient in one pass, dy never written) against the conv / statistics / reduce /
    apply / wgrad kernels they replace and against the fp32 oracle, on the stem's MetaKernel (3 -> C positional conv on
    the 9x grid, 5 -> C feature projection).  BatchNorm biases are shifted so that most ReLU gates are firmly open:
    with gates flipping under bf16 rounding every variant sits ~10 % (max-abs) from the fp32 gradients of this randomly
    initialised stem and nothing tight can be said (see test_gpu_model.py::test_detector_gradients_vs_oracle)."""
    from oracle import model as om
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd import engine_bwd
    from range_view_3d_detection_amd.nn.stems import MetaKernel

    torch.manual_seed(4)
    gen = torch.Generator().manual_seed(4)
    m = MetaKernel(5, 64, 3, 2).to(DEV).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data = (0.5 + torch.rand(mod.weight.shape, generator=gen)).to(DEV)
            mod.bias.data = (0.3 * torch.randn(mod.bias.shape, generator=gen) + 3.0).to(DEV)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    feats = torch.randn(2, 5, 16, 96, generator=gen)
    cart = torch.randn(2, 3, 16, 96, generator=gen) * 5
    probe = torch.randn(2, 64, 16, 96, generator=gen)

    def run(fused: bool):
        engine_bwd.SMALLK_FUSION = fused
        E.SMALLK_FORWARD = fused
        try:
            m.load_state_dict(sd)
            m.zero_grad(set_to_none=True)
            out = m(feats.to(DEV), cart.to(DEV)).float()
            (out * probe.to(DEV)).sum().backward()
            stats = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if "running_" in k}
            return out.detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}, stats
        finally:
            engine_bwd.SMALLK_FUSION = True
            E.SMALLK_FORWARD = True

    out_a, a, st_a = run(True)
    out_b, b, st_b = run(False)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    out_o = om.meta_kernel(feats, cart, {"m." + k: v for k, v in {**sd, **params}.items()}, "m", nm=om.Numerics(train=True))
    (out_o * probe).sum().backward()
    assert rel_err(out_a, out_b) < 2e-2 and rel_err(out_a, out_o) < 2e-2
    for k in st_a:  # closed-form batch statistics == statistics of the conv output
        assert rel_err(st_a[k], st_b[k]) < 1e-3, (k, rel_err(st_a[k], st_b[k]))
    for k in a:
        ref = params[k].grad
        assert _cos(a[k], ref) > 0.985 and _cos(a[k], b[k]) > 0.99, (k, _cos(a[k], ref), _cos(a[k], b[k]))
        assert rel_err(a[k], ref) < max(2.0 * rel_err(b[k], ref), 2e-2), (k, rel_err(a[k], ref), rel_err(b[k], ref))


@pytest.mark.parametrize("project", [False, True])
def test_basic_block_fed_a_lazy_input_accumulates_both_consumers(project):
    """A Lazy (unmaterialised relu(bn(conv))) feeding a BasicBlock has TWO consumers -- the block's first conv and either
    its projection conv or the identity branch -- whose gradients must add up (round-1 advisory: the second writer used to
    overwrite the first).  Checked against fp32 autograd of the oracle; BatchNorm biases shifted so that the ReLU gates are
    firmly open (see test_block_backward for why), tolerance = the block tests' (3e-2 of max + rounding floor)."""
    from oracle import model as om
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd import program
    from range_view_3d_detection_amd.nn.blocks import BasicBlock
    from range_view_3d_detection_amd.nn.stems import conv_norm_act

    gen = torch.Generator().manual_seed(11 + project)
    cin, c = 32, 64
    pre = conv_norm_act(cin, c, 1)  # conv -> BN -> ReLU: the Lazy producer
    blk = BasicBlock(c, c if not project else 96, kernel_size=3, project=project)
    holder = torch.nn.ModuleDict({"pre": pre, "blk": blk})
    for mod in holder.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data = 0.5 + torch.rand(mod.weight.shape, generator=gen)
            mod.bias.data = 0.2 * torch.randn(mod.bias.shape, generator=gen) + 3.0
        if isinstance(mod, torch.nn.Conv2d):
            mod.weight.data = bf16r(torch.randn(mod.weight.shape, generator=gen) * (2.0 / (mod.weight[0].numel())) ** 0.5)
    sd = {k: v.detach().clone() for k, v in holder.state_dict().items()}
    x = bf16r(torch.randn(2, cin, 8, 64, generator=gen))
    # oracle (fp32 autograd)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    full = {**sd, **params}
    xo = x.clone().requires_grad_(True)
    nm = om.Numerics(train=True)
    h = torch.relu(om.batch_norm(om.conv2d_same(xo, full["pre.0.weight"], nm=nm), full, "pre.1", nm=nm))
    out_o = om.basic_block(h, full, "blk", project=project, nm=nm)
    probe = bf16r(torch.randn(out_o.shape, generator=gen))
    (out_o * probe).sum().backward()

    holder = holder.to(DEV).train()

    def build(t, xin):
        a = E.Act.from_nchw(xin)
        lazy = E.conv_bn(t, E.tap_layer(holder["pre"][0]), a, holder["pre"][1], relu=True)
        assert isinstance(lazy, E.Lazy)
        return [a], [program.basic_block_program(t, holder["blk"], lazy)]

    xd = x.to(DEV).requires_grad_(True)
    out = program.run(build, holder, [xd])[0]
    assert rel_err(out.float(), out_o.detach()) < 3e-2
    (out.float() * probe.to(DEV)).sum().backward()
    n_px = probe.numel() // probe.shape[1]
    pairs = [("gin", xd.grad.float(), xo.grad)] + [(k, p.grad, params[k].grad) for k, p in holder.named_parameters()]
    for k, got, orc in pairs:
        # (two convs deep: 5e-2 of max + the rounding floor of test_block_backward; direction within 1e-3)
        tol = 5e-2 * float(orc.abs().max()) + 4e-4 * n_px * float(probe.abs().max())
        err = float((got.detach().double().cpu() - orc.double()).abs().max())
        assert err < tol and _cos(got, orc) > 0.999, (k, err, tol, _cos(got, orc))


def test_meta_kernel_single_positional_layer():
    """``MetaKernel(num_layers=1)`` (a legal reference constructor argument): the only positional layer is the 3 -> C one, which
    must stay on the generic conv path because MetaModulateOp folds its BatchNorm (round-1 advisory: AttributeError)."""
    from oracle import model as om
    from range_view_3d_detection_amd.nn.stems import MetaKernel

    gen = torch.Generator().manual_seed(3)
    m = MetaKernel(5, 32, 3, num_layers=1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data = 0.5 + torch.rand(mod.weight.shape, generator=gen)
            mod.bias.data = 0.3 * torch.randn(mod.bias.shape, generator=gen) + 2.0
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    feats = torch.randn(2, 5, 8, 64, generator=gen)
    cart = torch.randn(2, 3, 8, 64, generator=gen) * 5
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    out_o = om.meta_kernel(feats, cart, {"m." + k: v for k, v in {**sd, **params}.items()}, "m", num_layers=1, nm=om.Numerics(train=True))
    probe = torch.randn(out_o.shape, generator=gen)
    (out_o * probe).sum().backward()
    m = m.to(DEV).train()
    out = m(feats.to(DEV), cart.to(DEV)).float()
    assert rel_err(out, out_o.detach()) < 3e-2
    (out * probe.to(DEV)).sum().backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and _cos(p.grad, params[k].grad) > 0.98, (k, _cos(p.grad, params[k].grad))
    m.eval()
    with torch.no_grad():
        assert torch.isfinite(m(feats.to(DEV), cart.to(DEV)).float()).all()


@pytest.mark.parametrize("N,H,W,C", [(2, 5, 37, 32), (1, 4, 64, 256), (2, 3, 21, 96)])
def test_meta_modulate_backward_fused_kernels_vs_fp32(N, H, W, C):
    """rv_meta_modulate_bwd_sums / _apply (modulation backward fused with the positional layer's BatchNorm+ReLU backward:
    two passes over the 9x-grid tensors, the activated gradient never written) against the same formulas in fp32 torch ops
    on the bf16 inputs: dfeat and dy one bf16 rounding (4e-3 of max), the (sum z, sum z*xhat) rows 1e-4 of their scale.
    Widths that are not multiples of 8 exercise the XCD column strips, C = 96 a channel count that does not divide 256."""
    from range_view_3d_detection_amd import _lib as L

    gen = torch.Generator().manual_seed(N * 1000 + W)
    bf = lambda *s: torch.randn(*s, generator=gen).to(torch.bfloat16).to(DEV)
    dgeo, y, feat = bf(N, H, W, 9, C), bf(N, H, W, 9, C), bf(N, H, W, C)
    scale = (0.5 + torch.rand(C, generator=gen)).to(DEV)
    shift = (0.3 * torch.randn(C, generator=gen)).to(DEV)
    mean = (0.2 * torch.randn(C, generator=gen)).to(DEV)
    invstd = (0.5 + torch.rand(C, generator=gen)).to(DEV)
    coef = torch.stack([0.5 + torch.rand(C, generator=gen), 0.1 * torch.randn(C, generator=gen), 0.1 * torch.randn(C, generator=gen)]).to(DEV)
    lib = L.load()
    rows = lib.rv_meta_bwd_rows(L.i32(N), L.i32(H), L.i32(W))
    partial = torch.zeros((rows + L.STATS_SCRATCH_ROWS, 2, C), dtype=torch.float32, device=DEV)
    dfeat = torch.empty_like(feat)
    dy = torch.empty_like(y)
    L.call("rv_meta_modulate_bwd_sums", L.ptr(dgeo), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(feat), L.i32(C),
           L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(dfeat), L.i32(C), L.ptr(partial), L.stream_ptr())
    L.call("rv_meta_modulate_bwd_apply", L.ptr(dgeo), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(coef), L.ptr(feat),
           L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(dy), L.stream_ptr())
    torch.cuda.synchronize()
    # fp32 restatement: nbr[n,h,w,k] = feat[n, h+k//3-1, w+k%3-1] (zero outside), exactly F.unfold's neighbourhood order
    fp = F.pad(feat.float(), (0, 0, 1, 1, 1, 1))
    nbr = torch.stack([fp[:, k // 3 : k // 3 + H, k % 3 : k % 3 + W] for k in range(9)], dim=3)
    inside = F.pad(torch.ones(N, H, W, 1, device=DEV), (0, 0, 1, 1, 1, 1))
    inside = torch.stack([inside[:, k // 3 : k // 3 + H, k % 3 : k % 3 + W] for k in range(9)], dim=3)
    act = y.float() * scale + shift
    gate = (act > 0).float() * inside
    z = dgeo.float() * nbr * gate
    xhat = (y.float() - mean) * invstd
    s0, s1 = z.sum((0, 1, 2, 3)), (z * xhat).sum((0, 1, 2, 3))
    got = partial[:rows].double().sum(0)
    for a, b in ((got[0], s0), (got[1], s1)):
        assert float((a - b.double()).abs().max()) < 1e-4 * float(z.abs().sum((0, 1, 2, 3)).max()), (a, b)
    # dfeat[q] = sum_k dgeo[q - off_k][k] * relu(act)[q - off_k][k]: scatter the per-(p, k) products to the neighbour
    contrib = F.pad(torch.zeros(N, H, W, C, device=DEV), (0, 0, 1, 1, 1, 1))
    prod = dgeo.float() * torch.relu(act) * inside
    for k in range(9):
        contrib[:, k // 3 : k // 3 + H, k % 3 : k % 3 + W] += prod[:, :, :, k]
    want_dfeat = contrib[:, 1:-1, 1:-1]
    assert rel_err(dfeat.float(), want_dfeat) < 4e-3, rel_err(dfeat.float(), want_dfeat)
    want_dy = coef[0] * (z - coef[1] - xhat * coef[2])
    assert rel_err(dy.float(), want_dy) < 4e-3, rel_err(dy.float(), want_dy)


def test_meta_kernel_backward_fused_matches_unfused():
    """MetaKernel training step with the fused modulation/BatchNorm backward against the unfused chain it replaces
    (rv_meta_modulate_bwd -> rv_bn_bwd_reduce -> rv_bn_bwd_apply): same parameter gradients up to the one bf16 rounding of
    the activated gradient the unfused chain stores (cosine > 0.999, 2e-2 of max)."""
    from range_view_3d_detection_amd import engine_bwd
    from range_view_3d_detection_amd.nn.stems import MetaKernel

    gen = torch.Generator().manual_seed(11)
    m = MetaKernel(5, 64, 3, 2).to(DEV).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    feats = torch.randn(2, 5, 8, 72, generator=gen).to(DEV)
    cart = (torch.randn(2, 3, 8, 72, generator=gen) * 5).to(DEV)
    probe = torch.randn(2, 64, 8, 72, generator=gen).to(DEV)

    def run(fused: bool):
        engine_bwd.META_BWD_FUSE = fused
        try:
            m.load_state_dict(sd)
            m.zero_grad(set_to_none=True)
            (m(feats, cart).float() * probe).sum().backward()
            return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        finally:
            engine_bwd.META_BWD_FUSE = True

    a, b = run(True), run(False)
    for k in a:
        assert _cos(a[k], b[k]) > 0.999 and rel_err(a[k], b[k]) < 2e-2, (k, _cos(a[k], b[k]), rel_err(a[k], b[k]))


@pytest.mark.parametrize("C", [256, 128])
@pytest.mark.parametrize("P", [999, 41472])
def test_pos_backward_sums_kernel_vs_fp64(P, C):
    """rv_pos_backward_sums (second positional layer's backward-data GEMM fused with the first layer's small-K BatchNorm
    backward sums: dh1 = dy2 W2 never written) against the same sums in fp64 torch ops on the bf16 inputs:
    S0 = sum g, S1 = sum g xhat, R[d] = sum g rel[d] with g = dh1 [s1 y1 + t1 > 0], y1 = W1 rel -- 2e-4 of each plane's scale
    (fp32 MFMA accumulation and fp32 running sums against fp64); the data moments of rel exactly as rv_bn_bwd_smallk_sums."""
    from range_view_3d_detection_amd import _lib as L

    gen = torch.Generator().manual_seed(P + 1)
    rel = torch.zeros(P, 32, dtype=torch.bfloat16)
    rel[:, :3] = (torch.randn(P, 3, generator=gen) * 2).to(torch.bfloat16)
    w1 = torch.zeros(C, 32, dtype=torch.bfloat16)
    w1[:, :3] = torch.randn(C, 3, generator=gen).to(torch.bfloat16)
    w2s = (torch.randn(C, C, generator=gen) / 16).to(torch.bfloat16)  # scatter image [ci][co]
    dy2 = torch.randn(P, C, generator=gen).to(torch.bfloat16)
    s1, t1 = 0.5 + torch.rand(C, generator=gen), 0.3 * torch.randn(C, generator=gen)
    mu, isd = 0.2 * torch.randn(C, generator=gen), 0.5 + torch.rand(C, generator=gen)
    rel, w1, w2s, dy2, s1, t1, mu, isd = (x.to(DEV) for x in (rel, w1, w2s, dy2, s1, t1, mu, isd))
    ws = torch.empty(L.load().rv_bn_bwd_smallk_workspace_bytes(L.i64(P), L.i32(C), L.i32(3)), dtype=torch.uint8, device=DEV)
    sums = torch.zeros(6 * C, dtype=torch.float64, device=DEV)
    moms = torch.zeros(20, dtype=torch.float64, device=DEV)
    L.call("rv_pos_backward_sums", L.i64(P), L.i32(C), L.ptr(dy2), L.ptr(w2s), L.ptr(rel), L.i32(32), L.i32(3), L.ptr(w1), L.i32(32), L.ptr(s1), L.ptr(t1),
           L.ptr(mu), L.ptr(isd), L.ptr(sums), L.ptr(moms), L.ptr(ws), L.stream_ptr())
    torch.cuda.synchronize()
    r3 = rel[:, :3].double()
    y1 = r3 @ w1[:, :3].double().t()
    dh1 = dy2.double() @ w2s.double().t()  # dh1[p][ci] = sum_co dy2[p][co] * w2s[ci][co]
    gate = (y1.float() * s1 + t1 > 0).double()  # the kernel gates in fp32
    g = dh1 * gate
    xhat = (y1 - mu.double()) * isd.double()
    want = torch.stack([g.sum(0), (g * xhat).sum(0), (g * r3[:, 0:1]).sum(0), (g * r3[:, 1:2]).sum(0), (g * r3[:, 2:3]).sum(0), torch.zeros(C, dtype=torch.float64, device=DEV)])
    scale = torch.stack([g.abs().sum(0), (g * xhat).abs().sum(0), (g * r3[:, 0:1]).abs().sum(0), (g * r3[:, 1:2]).abs().sum(0), (g * r3[:, 2:3]).abs().sum(0),
                         torch.ones(C, dtype=torch.float64, device=DEV)])
    got = sums.view(6, C)
    err = ((got - want).abs() / scale.max(dim=1, keepdim=True).values).max(dim=1).values
    assert float(err.max()) < 2e-4, err
    m1 = torch.zeros(4, dtype=torch.float64, device=DEV)
    m1[:3] = r3.sum(0)
    assert float((moms[:4] - m1).abs().max()) < 1e-6 * float(r3.abs().sum())


@pytest.mark.parametrize("C", [256, 128])
def test_meta_kernel_positional_pair_backward_fused_matches_unfused(C):
    """MetaKernel at the rv-av2 (256) and rv-waymo (128) stem widths with the fused positional-pair backward against the chain it replaces
    (rv_tap_scatter of the second layer -> rv_bn_bwd_smallk of the first): the first layer's conv / BatchNorm gradients within
    the bf16 rounding of the input gradient the unfused chain stores (cosine 0.9999, 2e-2 of max); every other gradient
    bit-identical (same kernels, same inputs)."""
    from range_view_3d_detection_amd import engine_bwd
    from range_view_3d_detection_amd.nn.stems import MetaKernel

    gen = torch.Generator().manual_seed(23)
    m = MetaKernel(5, C, 3, 2).to(DEV).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    feats = torch.randn(1, 5, 16, 160, generator=gen).to(DEV)
    cart = (torch.randn(1, 3, 16, 160, generator=gen) * 5).to(DEV)
    probe = torch.randn(1, C, 16, 160, generator=gen).to(DEV)

    def run(fused: bool):
        engine_bwd.POS_BWD_FUSE = fused
        try:
            m.load_state_dict(sd)
            m.zero_grad(set_to_none=True)
            (m(feats, cart).float() * probe).sum().backward()
            return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        finally:
            engine_bwd.POS_BWD_FUSE = True

    a, b = run(True), run(False)
    for k in a:
        if k.startswith("positional_kernel.0."):
            assert _cos(a[k], b[k]) > 0.9999 and rel_err(a[k], b[k]) < 2e-2, (k, _cos(a[k], b[k]), rel_err(a[k], b[k]))
        else:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("kind,cin,cout,kernel,stride,pad,H,W", [
    ("conv", 128, 128, (3, 3), 2, None, 8, 256),          # stride-2 3x3 conv: forward + weight gradient on the folded view
    ("conv", 128, 256, (1, 1), 2, None, 8, 256),          # 1x1 stride-2 projection: weight gradient folded, backward-data into the even-column view
    ("convT", 128, 256, (3, 8), 4, (1, 2), 8, 64),        # the (3,8)/s4 up-sampler: backward-data + weight gradient folded
    ("convT", 128, 128, (3, 4), 2, (1, 1), 8, 128),       # the (3,4)/s2 up-sampler
])
def test_strided_layers_on_the_folded_view(kind, cin, cout, kernel, stride, pad, H, W):
    """The FOLDED stride-1 form of the strided layers (rv_fold_geom / rv_pack_weight_folded / rv_unfold_weight_grad; the LDS-DMA
    kernels are forced onto these small shapes with tapconv4_min_blocks = 1) against fp32 torch ops on bf16-representable
    operands -- forward one bf16 rounding (8e-3 of max), input gradient 8e-3, weight gradient 2e-5 (exact products, summation
    order only) -- and against the unfolded path of the same library (RV3D_NO_FOLD semantics: engine.FOLD_STRIDED = False)."""
    import ctypes

    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd import program

    g = torch.Generator().manual_seed(cin + cout + stride)
    if kind == "conv":
        m = torch.nn.Conv2d(cin, cout, kernel, stride=(1, stride), padding=((kernel[0] - 1) // 2, (kernel[1] - 1) // 2), bias=False)
    else:
        m = torch.nn.ConvTranspose2d(cin, cout, kernel, stride=(1, stride), padding=pad, bias=False)
    m.weight.data = bf16r(torch.randn(m.weight.shape, generator=g) * 0.1)
    x = bf16r(torch.randn(2, cin, H, W, generator=g)).requires_grad_(True)
    w = m.weight.data.clone().requires_grad_(True)
    if kind == "conv":
        y = F.conv2d(x, w, stride=(1, stride), padding=m.padding)
    else:
        y = F.conv_transpose2d(x, w, stride=(1, stride), padding=pad)
    probe = bf16r(torch.randn(y.shape, generator=g))
    (y * probe).sum().backward()
    m = m.to(DEV)

    def run(fold: bool):
        E.FOLD_STRIDED = fold
        m.__dict__.pop("_rv_layer", None)  # a fresh TapLayer: the folded geometry is cached on it
        m.zero_grad(set_to_none=True)
        sel = L.select(L.SEL_SMALL_GRIDS)
        sel.__enter__()
        try:
            E.PROFILE = E.KernelProfile()

            def build(t, xin):
                a = E.Act.from_nchw(xin)
                return [a], [E.ConvOp(t, E.tap_layer(m), a).out]

            xd = x.detach().to(DEV).requires_grad_(True)
            yd = program.run(build, m, [xd])[0]
            (yd.float() * probe.to(DEV)).sum().backward()
            torch.cuda.synchronize()
            ran = sorted(set(name for name, *_ in E.PROFILE.records))
            return yd.detach().float().cpu(), xd.grad.detach().float().cpu(), m.weight.grad.detach().cpu().clone(), ran
        finally:
            E.PROFILE = None
            E.FOLD_STRIDED = True
            sel.__exit__(None, None, None)

    y_f, dx_f, dw_f, ran_f = run(True)
    y_u, dx_u, dw_u, ran_u = run(False)
    # the folded run used the DMA kernels where the unfolded one used the generic strided ones
    assert any(n.startswith("wgrad3") for n in ran_f) and not any(n.startswith("wgrad3") for n in ran_u), (ran_f, ran_u)
    for got_y, got_dx, got_dw in ((y_f, dx_f, dw_f), (y_u, dx_u, dw_u)):
        assert rel_err(got_y, bf16r(y.detach())) < 8e-3
        assert rel_err(got_dx, bf16r(x.grad)) < 8e-3
        assert rel_err(got_dw, w.grad) < 2e-5
    assert rel_err(dw_f, dw_u) < 2e-5 and rel_err(y_f, y_u) < 8e-3 and rel_err(dx_f, dx_u) < 8e-3


@pytest.mark.parametrize("C,W", [(128, 256), (256, 128)])
def test_projection_block_sums_from_one_pass(C, W):
    """BasicBlock with a projection: out = relu(bn2(y2) + bn_p(yp)) puts two BatchNorms on one gradient; rv_bn_bwd_reduce_pair
    forms both sets of backward sums in one pass over (dOut, out, y2, yp).  Against two rv_bn_bwd_reduce passes
    (``engine_bwd.BNB_PAIR = False``): every gradient equal to 1e-6 of its max (the same values added in the same order)."""
    from range_view_3d_detection_amd import engine_bwd
    from range_view_3d_detection_amd.nn.blocks import BasicBlock

    gen = torch.Generator().manual_seed(C + W)
    m = BasicBlock(C // 2, C, project=True).to(DEV).train()
    x = torch.randn(2, C // 2, 16, W, generator=gen).to(DEV)
    probe = torch.randn(2, C, 16, W, generator=gen).to(DEV)

    def run(on: bool):
        engine_bwd.BNB_PAIR = on
        try:
            m.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            (m(xi).float() * probe).sum().backward()
            return {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}, xi.grad.detach().float().cpu()
        finally:
            engine_bwd.BNB_PAIR = True

    ga, dxa = run(True)
    gb, dxb = run(False)
    assert rel_err(dxa, dxb) < 1e-6, rel_err(dxa, dxb)
    for k in ga:
        assert rel_err(ga[k], gb[k]) < 1e-6, (k, rel_err(ga[k], gb[k]))


def test_a_training_step_frees_its_activations_without_the_cycle_collector():
    """The tape of a program (ops, Lazy operands, gradient buffers) must not contain reference cycles: the activations of a
    step are freed when its backward returns, not when Python's cycle collector next runs (a cycle BnOp <-> Lazy once kept
    every raw conv output alive across steps: 3.5x the step time from allocator pressure)."""
    import gc

    from range_view_3d_detection_amd.nn.blocks import ResidualBlock

    gen = torch.Generator().manual_seed(3)
    m = ResidualBlock(64, 128, 3).to(DEV).train()
    x = torch.randn(2, 64, 16, 256, generator=gen).to(DEV)

    def step():
        m.zero_grad(set_to_none=True)
        (m(x).float() ** 2).mean().backward()

    step()
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    assert grown < (1 << 20), f"{grown / 2**20:.1f} MiB still allocated after three steps with the cycle collector off"


@pytest.mark.parametrize("C,n_out,W", [(256, 26, 200), (512, 8, 96), (256, 3, 328)])
def test_head_final_conv_backward_fused_with_the_last_batchnorm(C, n_out, W):
    """A tower's final 1x1 conv behind conv -> BatchNorm -> ReLU (nn/heads/dense_head.py:44-57): rv_head_final_bwd_sums / _apply
    recompute the final conv's input gradient inside the BatchNorm backward instead of storing it.  Against (a) the unfused chain of
    the same library (backward-data launch + rv_bn_bwd_reduce + rv_bn_bwd_apply): parameter gradients of the unit and the gradient
    w.r.t. the tower input at bf16 level, (dgamma, dbeta) 5e-3 (the unfused chain rounds dA to bf16 before summing: measured 2.4e-3); (b) torch fp32
    autograd on the same bf16-rounded operands.  Pixel counts that are not multiples of 16 (W = 200: 3 x 5 x 200 = 3000 pixels,
    ragged last range) included."""
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine_bwd
    from range_view_3d_detection_amd.nn.heads.dense_head import DenseHead

    gen = torch.Generator().manual_seed(C + n_out)
    m = DenseHead(C, C, n_out, 3, 1, num_blocks=1, prior_prob=0.01)
    m.blocks[0][0].weight.data = 0.05 * torch.randn(m.blocks[0][0].weight.shape, generator=gen)
    m.blocks[1][0].weight.data = 0.2 * torch.randn(m.blocks[1][0].weight.shape, generator=gen)
    m.blocks[0][1].weight.data = 0.5 + torch.rand(C, generator=gen)
    m.blocks[0][1].bias.data = 0.3 * torch.randn(C, generator=gen)
    x = bf16r(torch.randn(3, C, 5, W, generator=gen))
    probe = torch.randn(3, n_out, 5, W, generator=gen)
    # (b) fp32 reference: train-mode BatchNorm, bf16-rounded conv operands as the kernels see them
    ref = DenseHead(C, C, n_out, 3, 1, num_blocks=1)
    ref.load_state_dict(m.state_dict())
    ref.blocks[0][0].weight.data = bf16r(ref.blocks[0][0].weight.data)
    ref.blocks[1][0].weight.data = bf16r(ref.blocks[1][0].weight.data)
    xr = x.clone().requires_grad_(True)
    y = F.conv2d(xr, ref.blocks[0][0].weight, padding=1)
    a = F.relu(F.batch_norm(y, None, None, ref.blocks[0][1].weight, ref.blocks[0][1].bias, training=True, eps=1e-5))
    out = F.conv2d(a, ref.blocks[1][0].weight, ref.blocks[1][0].bias)
    (out * probe).sum().backward()
    m = m.to(DEV).train()

    def run(fused: bool):
        old = engine_bwd.HEAD_FINAL_FUSE
        engine_bwd.HEAD_FINAL_FUSE = fused
        calls = []
        real = L._call

        def spy(name, *args):
            calls.append(name)
            return real(name, *args)

        L._call = spy
        try:
            m.zero_grad(set_to_none=True)
            xi = x.to(DEV).requires_grad_(True)
            (m(xi).float() * probe.to(DEV)).sum().backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}, xi.grad.detach().float().cpu(), calls
        finally:
            L._call = real
            engine_bwd.HEAD_FINAL_FUSE = old

    gf, dxf, cf = run(True)
    gu, dxu, cu = run(False)
    assert cf.count("rv_head_final_bwd_sums") == 1 and cf.count("rv_head_final_bwd_apply") == 1 and "rv_bn_bwd_apply" not in cf, cf
    assert "rv_head_final_bwd_sums" not in cu and "rv_bn_bwd_apply" in cu
    for k in gf:
        assert rel_err(gf[k], gu[k]) < (5e-3 if "blocks.0.1" in k else 2e-2) and _cos(gf[k], gu[k]) > 0.9999, (k, rel_err(gf[k], gu[k]))
    assert rel_err(dxf, dxu) < 2e-2 and _cos(dxf, dxu) > 0.9999
    refg = {"blocks.0.0.weight": ref.blocks[0][0].weight.grad, "blocks.0.1.weight": ref.blocks[0][1].weight.grad, "blocks.0.1.bias": ref.blocks[0][1].bias.grad,
            "blocks.1.0.weight": ref.blocks[1][0].weight.grad, "blocks.1.0.bias": ref.blocks[1][0].bias.grad}
    for k, r in refg.items():  # vs fp32 autograd: the fused chain is at least as close as the unfused one (+ a bf16 ulp of slack)
        ef, eu = rel_err(gf[k], r), rel_err(gu[k], r)
        assert ef < max(1.5 * eu, 4e-3) + 1e-3, (k, ef, eu)
    assert rel_err(dxf, xr.grad) < max(1.5 * rel_err(dxu, xr.grad), 8e-3) + 1e-3


@pytest.mark.parametrize("relu", [1, 0])
def test_head_final_sums_kernel_vs_fp64_with_and_without_relu(relu):
    """rv_head_final_bwd_sums at the C ABI (include/rv3d.h): the BatchNorm-backward sums and the final conv's weight gradient against
    fp64 torch ops on the same bf16 operands -- with a ReLU behind the BatchNorm and WITHOUT one (round-5 advice: with relu = 0 the
    kernel used to pack a constant 1.0 as the weight gradient's activated operand instead of scale * y + shift; the engine only ever
    passes relu = 1)."""
    from range_view_3d_detection_amd import _lib as L

    lib = L.load()
    gen = torch.Generator().manual_seed(77 + relu)
    P, C, n_out = 3000, 256, 8
    y = bf16r(torch.randn(P, C, generator=gen)).to(DEV).to(torch.bfloat16)
    dY = torch.zeros(P, 32)
    dY[:, :n_out] = bf16r(torch.randn(P, n_out, generator=gen))
    dY = dY.to(DEV).to(torch.bfloat16)
    W = bf16r(0.3 * torch.randn(n_out, C, generator=gen))
    wp = torch.zeros(C, 32)
    wp[:, :n_out] = W.t()
    wp = wp.to(DEV).to(torch.bfloat16)
    scale = (0.5 + torch.rand(C, generator=gen)).to(DEV)
    shift = (0.3 * torch.randn(C, generator=gen)).to(DEV)
    mean = (0.1 * torch.randn(C, generator=gen)).to(DEV)
    invstd = (0.8 + 0.4 * torch.rand(C, generator=gen)).to(DEV)
    rows = lib.rv_head_final_bwd_rows(L.i64(P))
    partial = torch.zeros((rows + L.STATS_SCRATCH_ROWS, 2, C), dtype=torch.float32, device=DEV)
    dw_partial = torch.zeros((rows, 32 * C), dtype=torch.float32, device=DEV)
    L.call("rv_head_final_bwd_sums", L.i64(P), L.i32(C), L.ptr(y), L.i32(C), L.ptr(dY), L.i32(32), L.ptr(wp), L.ptr(scale), L.ptr(shift),
           L.ptr(mean), L.ptr(invstd), L.i32(relu), L.ptr(partial), L.ptr(dw_partial), L.stream_ptr())
    dw = torch.empty((32, C), dtype=torch.float32, device=DEV)
    L.call("rv_reduce_rows", L.ptr(dw_partial), L.i32(rows), L.i32(32 * C), L.ptr(dw), L.stream_ptr())
    torch.cuda.synchronize()
    yd, dYd = y.double().cpu(), dY.double().cpu()
    t = yd * scale.double().cpu() + shift.double().cpu()
    dA = dYd[:, :n_out] @ W.double()
    g = dA * (t > 0) if relu else dA
    xhat = (yd - mean.double().cpu()) * invstd.double().cpu()
    s = partial[:rows].double().sum(0).cpu()
    assert rel_err(s[0].float(), g.sum(0).float()) < 1e-4 and rel_err(s[1].float(), (g * xhat).sum(0).float()) < 1e-4
    act = bf16r((t.clamp_min(0) if relu else t).float()).double()  # (the operand is packed to bf16 before the MFMA)
    want_dw = dYd[:, :n_out].t() @ act
    assert rel_err(dw[:n_out].cpu(), want_dw.float()) < 1e-4, rel_err(dw[:n_out].cpu(), want_dw.float())
    assert float(dw[n_out:].abs().max()) == 0.0
