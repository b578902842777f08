"""GPU parity of the full detector path: backbone + head + targets + loss + backward, decoder + weighted NMS.

Tolerances: see test_gpu_forward.py / test_gpu_backward.py; additionally
* dense targets: labels / panoptic ids / point counts bit-exact, regression targets 1e-5;
* loss scalars vs the oracle evaluated on the HIP path's own logits / regressands: 1e-4;
* weighted NMS vs the C oracle (declared semantics): kept indices and counts bit-exact, merged rows 1e-6.
"""

from __future__ import annotations

import math

import numpy as np
import pytest
import torch

from test_gpu_backward import _cos, _l2
from test_gpu_forward import DEV, rel_err

pytestmark = pytest.mark.gpu

NCLS = 5


def build_tiny(C=16, ncls=NCLS):
    from range_view_3d_detection_amd.nn.backbones.dla import RangeNet
    from range_view_3d_detection_amd.nn.heads.detection_head import DetectionHead

    layers = [C] * 5
    backbone = RangeNet(in_channels=5, layers=layers, out_channels=C, projection_kernel_size=1, dataset_name="av2",
                        num_neighbors=3, num_layers=2, stem_type="META",
                        _net={"_target_": "torchbox3d.nn.backbones.dla.RangeBackbone", "in_channels": 5, "layers": layers, "out_channels": C})
    tasks = {0: [f"C{i}" for i in range(ncls)]}
    tcfg = {"dataset_name": "av2", "tasks": tasks, "enable_azimuth_invariant_targets": True, "range_partitions": {1: [0.0, math.inf]},
            "fpn_assignment_method": None, "k": math.inf, "affinity_fn": "GAUSSIAN", "normalize_affinities": False, "sigma": 0.75}
    head = DetectionHead(fpn={1: 2 * C}, fpn_kernel_sizes={1: [3, 3]}, targets_config=tcfg, num_classification_blocks=4,
                         num_regression_blocks=4, final_kernel_size=1, tasks_cfg=tasks, task_in_channels=C, classification_weight=1.0,
                         regression_weight=1.0, coding_weights=[1.0] * 8, classification_head_channels=2 * C,
                         regression_head_channels=2 * C, classification_normalization_method="FOREGROUND",
                         _cls_loss={"_target_": "torchbox3d.nn.losses.classification.VarifocalLoss", "alpha": 0.75, "gamma": 2.0, "reduction": "none"},
                         _regression_loss={"_target_": "torch.nn.L1Loss", "reduction": "none"})
    return backbone, head


def load_tiny(g):
    backbone, head = build_tiny()
    sd = g.sub("sd")
    backbone.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")})
    head.load_state_dict({k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")})
    return backbone.to(DEV), head.to(DEV)


def test_state_dict_keys_match_reference(golden):
    g = golden("tiny_model")
    backbone, head = build_tiny()
    ref_keys = set(g.sub("sd").keys())
    ours = {f"backbone.{k}" for k in backbone.state_dict()} | {f"head.{k}" for k in head.state_dict()}
    assert ours == ref_keys


def test_targets_bit_exact(golden):
    from range_view_3d_detection_amd.nn.heads.detection_head import compute_targets

    g = golden("tiny_model")
    tasks = {0: ["c"] * NCLS}
    data = {"cart": g["cart"].to(DEV), "annotations": g["annotations"]}
    t = compute_targets(data, tasks, [1], {"enable_azimuth_invariant_targets": True, "fpn_assignment_method": None})[1][0]
    for k in ("classification_labels", "panoptics", "points_per_obj"):
        assert torch.equal(t[k].cpu(), g[f"targets/{k}"]), k
    assert rel_err(t["regression_targets"], g["targets/regression_targets"]) < 1e-5
    assert int(t["num_objects"].item()) == int(g["loss/total_objects"].item())


def test_loss_kernel_matches_oracle(golden):
    """Loss + gradients on the reference's own logits / regressands (fp32 in, fp64 reductions)."""
    from oracle import targets as otgt
    from range_view_3d_detection_amd.nn.heads.detection_head import _DetectionLossFn, compute_targets

    g = golden("tiny_model")
    logits = g["logits"].clone().requires_grad_(True)
    reg = g["regressands"].clone().requires_grad_(True)
    tg_o = {k: g[f"targets/{k}"] for k in ("classification_labels", "panoptics", "points_per_obj", "regression_targets")}
    lo = otgt.detection_loss(logits, reg, g["cart"], g["mask"], tg_o, NCLS)
    lo["loss"].backward()

    data = {"cart": g["cart"].to(DEV), "annotations": g["annotations"]}
    t = compute_targets(data, {0: ["c"] * NCLS}, [1], {"enable_azimuth_invariant_targets": True, "fpn_assignment_method": None})[1][0]
    hp = {"coding_weights": [1.0] * 8, "cls_weight": 1.0, "reg_weight": 1.0, "smoothing": 1.0, "sigma": 0.75, "alpha": 0.75, "gamma": 2.0, "az_inv": True}
    ld = g["logits"].to(DEV).requires_grad_(True)
    rd = g["regressands"].to(DEV).requires_grad_(True)
    loss, sums, soft, fg = _DetectionLossFn.apply(ld, rd, g["cart"].to(DEV), g["mask"].to(DEV), t, hp)
    loss.backward()
    assert rel_err(loss.reshape(()), g["loss/loss"].reshape(())) < 1e-4
    assert rel_err(soft, g["targets/soft"]) < 1e-5
    assert torch.equal(fg.cpu(), g["aux/foreground"])
    for key, val in (("classification_loss", sums[0] / sums[13]), ("foreground_loss", sums[1] / sums[13]), ("background_loss", sums[2] / sums[13]),
                     ("coordinate_loss", sums[4:7].sum() / sums[12]), ("dimension_loss", sums[7:10].sum() / sums[12]),
                     ("rotation_loss", sums[10:12].sum() / sums[12]), ("total_fg", sums[13]), ("total_objects", sums[12])):
        assert rel_err(val.reshape(()), g[f"loss/{key}"].reshape(())) < 1e-4, key
    assert rel_err(ld.grad, logits.grad) < 1e-4
    assert rel_err(rd.grad, reg.grad) < 1e-4
    # an incoming gradient other than one reaches the kernel as a DEVICE scalar (sums[15]: no pass over the gradients afterwards)
    ld2, rd2 = g["logits"].to(DEV).requires_grad_(True), g["regressands"].to(DEV).requires_grad_(True)
    loss2, *_ = _DetectionLossFn.apply(ld2, rd2, g["cart"].to(DEV), g["mask"].to(DEV), t, hp)
    (loss2 * -2.5).backward()
    assert rel_err(ld2.grad, -2.5 * ld.grad) < 1e-6 and rel_err(rd2.grad, -2.5 * rd.grad) < 1e-6


def test_tiny_detector_forward_backward(golden):
    g = golden("tiny_model")
    backbone, head = load_tiny(g)
    backbone.train()
    head.train()
    data = {"features": g["features"].to(DEV), "cart": g["cart"].to(DEV), "mask": g["mask"].to(DEV), "annotations": g["annotations"]}
    from oracle import model as om

    with torch.no_grad():
        feats_o, logits_o, reg_o = om.detector_forward(g["features"], g["cart"], g.sub("sd"), nm=om.Numerics.bf16(train=True))
    feats = backbone(data)
    for s in (1, 2, 4, 16):
        assert feats[s].shape == g[f"feat/{s}"].shape
        # ~60 conv+BN layers on 2x8x{4..64} pixels: vs the oracle with the same bf16 storage points, and (looser) vs fp32
        # (max-norm is dominated by single ReLU-gate flips that propagate through the 16-channel, 8x4..8x64-pixel layers)
        # The yardstick is the CPU bf16 emulation's OWN distance from the reference's fp32 output on this fixture (round 5's review: a constant
        # 2e-1 says nothing): the HIP result may be 1.5x that far from fp32 and 2x that far from the emulation (two bf16 realisations), + 1e-2.
        emu = rel_err(feats_o[s], g[f"feat/{s}"])
        e_ref, e_orc = rel_err(feats[s].float(), g[f"feat/{s}"]), rel_err(feats[s].float(), feats_o[s])
        assert e_orc < max(3e-2, 2.0 * emu + 1e-2) and _cos(feats[s].float(), feats_o[s]) > 0.99, (s, e_orc, emu, _cos(feats[s].float(), feats_o[s]))
        assert e_ref < max(3e-2, 1.5 * emu + 1e-2) and _cos(feats[s].float(), g[f"feat/{s}"]) > 0.985, (s, e_ref, emu, _cos(feats[s].float(), g[f"feat/{s}"]))
        print(f"tiny detector feat/{s}: emulation {emu:.3e}, HIP vs fp32 {e_ref:.3e}, HIP vs emulation {e_orc:.3e}")
    outputs, losses = head(feats, data, return_loss=True)
    # (BatchNorm over as few as 2x8x4 = 64 values amplifies bf16 rounding: the same yardstick)
    for got, orc, ref in ((outputs[1][0]["logits"], logits_o, g["logits"]), (outputs[1][0]["regressands"], reg_o, g["regressands"])):
        emu = rel_err(orc, ref)
        assert rel_err(got, orc) < max(3e-2, 2.0 * emu + 1e-2) and _cos(got, orc) > 0.99, (rel_err(got, orc), emu, _cos(got, orc))
        assert rel_err(got, ref) < max(3e-2, 1.5 * emu + 1e-2) and _cos(got, ref) > 0.985, (rel_err(got, ref), emu, _cos(got, ref))
        print(f"tiny detector head output: emulation {emu:.3e}, HIP vs fp32 {rel_err(got, ref):.3e}, HIP vs emulation {rel_err(got, orc):.3e}")
    for k in ("classification_labels", "panoptics", "points_per_obj"):
        assert torch.equal(data[1][0][k].cpu(), g[f"targets/{k}"])
    assert rel_err(losses["loss"].reshape(()), g["loss/loss"].reshape(())) < 3e-2
    assert losses["loss"].dtype == torch.float64
    losses["loss"].backward()
    # Gradients of THIS fixture are not comparable across precisions: its BatchNorms see as few as 64 values per channel
    # (2 x 8 x 4 pixels), and even the CPU oracle with bf16 storage emulation only reaches a median cosine of 0.68
    # against the reference's fp32 gradients.  Gradient parity of the composed model is checked on a better conditioned
    # model in test_detector_gradients_vs_oracle; here: every parameter gets a finite gradient.
    for prefix, mod in (("backbone", backbone), ("head", head)):
        for k, p in mod.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    # running statistics updated in place
    sd_after = {**{f"backbone.{k}": v for k, v in backbone.state_dict().items()}, **{f"head.{k}": v for k, v in head.state_dict().items()}}
    worst = max(rel_err(sd_after[k], v) for k, v in g.sub("sd_after").items())
    assert worst < 5e-2, worst


@pytest.mark.parametrize("bn_bias_shift,n_feat,n_cls,width", [(6.0, 5, 5, 256), (3.0, 5, 5, 256), (0.0, 5, 5, 256),
                                                               (3.0, 6, 3, 336)])  # Waymo-like: 6 features, 3 classes, W % 64 != 0
def test_detector_gradients_vs_oracle(bn_bias_shift, n_feat, n_cls, width):
    """Composed model (stem + backbone + towers + targets + loss), forward AND backward, against the pinned oracle.

    32-channel model on 2 x 16 x 256 sweeps.  HIP (bf16 storage) vs the oracle in fp32 and vs the oracle with bf16 storage
    emulation: per-parameter gradient cosine.

    Conditioning: with bf16 storage a ReLU gate whose pre-activation is within rounding error of zero flips, and a flipped
    gate is an all-or-nothing gradient error.  Over the ~45 BN-ReLU layers of this model that makes the randomly
    initialised network's gradient ill conditioned -- the CPU oracle's own bf16 emulation only reaches a median cosine of
    ~0.6 against its fp32 gradients.  So the composition (tape order, residual sums, gradient routing, every backward
    kernel in its place) is checked where the problem is well conditioned -- BatchNorm biases shifted so that (nearly)
    all / most gates are firmly open -- and in the natural regime (shift 0) the HIP path is required to be at least as
    close to the fp32 oracle as the CPU bf16 emulation is.  Gate masks themselves are checked exactly in
    test_gpu_backward.py.
    """
    from bench import Detector, build_model, synthetic_batch
    from oracle import model as om
    from oracle import targets as otgt

    torch.manual_seed(0)
    backbone, head = build_model("c32", n_cls, n_feat)
    gen = torch.Generator().manual_seed(1)
    for m in list(backbone.modules()) + list(head.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=gen)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=gen) + bn_bias_shift
    sd = {**{f"backbone.{k}": v.clone() for k, v in backbone.state_dict().items()}, **{f"head.{k}": v.clone() for k, v in head.state_dict().items()}}
    batch = synthetic_batch(2, 16, width, seed=3, device="cpu", n_feat=n_feat, boxes_per_sweep=8, n_cls=n_cls)
    torch.set_num_threads(min(16, torch.get_num_threads()))

    def oracle_grads(nm):
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
        _, logits, reg = om.detector_forward(batch["features"], batch["cart"], {**sd, **params}, nm=nm)
        tg = otgt.compute_targets(batch["cart"], batch["annotations"], n_cls)
        loss = otgt.detection_loss(logits, reg, batch["cart"], batch["mask"], tg, n_cls)["loss"]
        loss.backward()
        return float(loss.detach()), {k: p.grad for k, p in params.items()}

    loss32, g32 = oracle_grads(om.Numerics(train=True))
    loss16, g16 = oracle_grads(om.Numerics.bf16(train=True))
    model = Detector(backbone, head).to(DEV).train()
    data = {k: (v.to(DEV) if k != "annotations" else v) for k, v in batch.items()}
    loss = model(data)
    loss.backward()
    assert abs(float(loss) - loss32) / abs(loss32) < 2e-2, (float(loss), loss32, loss16)
    cos32, cos16, cos_emu, names = [], [], [], []
    for k, p in model.named_parameters():
        ref = g32[k]
        assert torch.isfinite(p.grad).all(), k
        if float(ref.norm()) < 1e-9:
            continue
        names.append(k)
        cos32.append(_cos(p.grad, ref))
        cos16.append(_cos(p.grad, g16[k]))
        cos_emu.append(_cos(g16[k], ref))
    cos32, cos_emu = np.array(cos32), np.array(cos_emu)
    med32, med16, med_emu = np.median(cos32), np.median(cos16), np.median(cos_emu)
    print(f"[shift {bn_bias_shift}] {len(cos32)} parameters; gradient cosine medians: HIP~fp32 {med32:.4f}  HIP~bf16-emulation {med16:.4f}  "
          f"emulation~fp32 {med_emu:.4f}; min {cos32.min():.4f} {min(cos16):.4f} {cos_emu.min():.4f}")
    for i in np.argsort(cos32)[:4]:
        print(f"    worst: {names[i]:56s} HIP~fp32 {cos32[i]:.4f}  emulation~fp32 {cos_emu[i]:.4f}  |g| {float(g32[names[i]].norm()):.2e}")
    # never worse than what bf16 storage costs the CPU emulation of the same model
    q32, q_emu = np.quantile(cos32, 0.05), np.quantile(cos_emu, 0.05)
    assert med32 > med_emu - 0.03 and q32 > q_emu - 0.05, (med32, med_emu, q32, q_emu)
    if bn_bias_shift >= 6.0:
        assert med32 > 0.995 and q32 > 0.99, (med32, q32)
    elif bn_bias_shift >= 3.0:
        assert med32 > 0.98 and q32 > 0.9, (med32, q32)


def test_baseline_config0_shape_vs_oracle():
    """BASELINE.json configs[0] on the HIP path (round-5 review, item 9): the debug-overfit configuration's own shape -- ONE synthetic
    64 x 512 x 5 sweep, the nearest reference-valid tiny backbone ``layers=[16]*5`` (towers 32, 5 classes; what bench.py times on the CPU as
    ``cpu_baseline.config1``) -- one training step against the oracle: logits / regressands against the oracle with the same bf16
    storage points (max(3e-2, 1.5 x the emulation's own distance from fp32 + 1e-2) of the maximum) and the
    fp32 oracle (direction), loss 2e-3 relative to the bf16-emulating oracle and 1e-2 to fp32, per-parameter gradient cosines against
    fp32 no worse than the CPU emulation's (median - 0.02, 5 % quantile - 0.05) and median > 0.99 (gates firmly open: BatchNorm shift
    3.0 as in test_detector_gradients_vs_oracle; 16 logical channels = one 32-channel padded slab per tensor)."""
    from bench import Detector, build_model, synthetic_batch
    from oracle import model as om
    from oracle import targets as otgt

    n_cls = 5
    torch.manual_seed(0)
    backbone, head = build_model("c16", n_cls)
    gen = torch.Generator().manual_seed(1)
    for m in list(backbone.modules()) + list(head.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=gen)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=gen) + 3.0
    sd = {**{f"backbone.{k}": v.clone() for k, v in backbone.state_dict().items()}, **{f"head.{k}": v.clone() for k, v in head.state_dict().items()}}
    batch = synthetic_batch(1, 64, 512, seed=0, device="cpu", boxes_per_sweep=4, n_cls=n_cls)  # (bench.py::cpu_baseline.config1's sweep)

    def oracle_run(nm):
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
        _, logits, reg = om.detector_forward(batch["features"], batch["cart"], {**sd, **params}, nm=nm)
        tg = otgt.compute_targets(batch["cart"], batch["annotations"], n_cls)
        loss = otgt.detection_loss(logits, reg, batch["cart"], batch["mask"], tg, n_cls)["loss"]
        loss.backward()
        return logits.detach(), reg.detach(), float(loss.detach()), {k: p.grad for k, p in params.items()}, tg

    lg32, rg32, loss32, g32, tg = oracle_run(om.Numerics(train=True))
    lg16, rg16, loss16, g16, _ = oracle_run(om.Numerics.bf16(train=True))
    model = Detector(backbone, head).to(DEV).train()
    data = {k: (v.to(DEV) if k != "annotations" else v) for k, v in batch.items()}
    feats = model.backbone(data)
    outputs, losses = model.head(feats, data, return_loss=True)
    losses["loss"].backward()
    torch.cuda.synchronize()
    for k in ("classification_labels", "panoptics", "points_per_obj"):
        assert torch.equal(data[1][0][k].cpu(), tg[k]), k
    logits, reg = outputs[1][0]["logits"].float().cpu(), outputs[1][0]["regressands"].float().cpu()
    assert logits.shape == (1, n_cls, 64, 512) and reg.shape == (1, 8, 64, 512)
    e_l, e_r = rel_err(logits, lg16), rel_err(reg, rg16)
    print(f"configs[0] shape: logits {e_l:.3e} / regressands {e_r:.3e} of max against the bf16-emulating oracle; cosine vs fp32 "
          f"{_cos(logits, lg32):.5f} / {_cos(reg, rg32):.5f}; loss {float(losses['loss']):.6f} (emulation {loss16:.6f}, fp32 {loss32:.6f})")
    # bounds as in tests/test_gpu_realwidth.py::_check_forward: against the emulation max(3e-2, 1.5 x what bf16 storage costs the emulation itself
    # against fp32 + 1e-2); against fp32 1.5 x that yardstick + 1e-2 (first run on an MI355X: logits 2.6e-3, regressands 2.8e-2 of the maximum)
    for got, emu, ref, e in ((logits, lg16, lg32, e_l), (reg, rg16, rg32, e_r)):
        yard = rel_err(emu, ref)
        assert e < max(3e-2, 1.5 * yard + 1e-2) and rel_err(got, ref) < 1.5 * yard + 1e-2, (e, rel_err(got, ref), yard)
    assert _cos(logits, lg32) > 0.999 and _cos(reg, rg32) > 0.999
    loss = float(losses["loss"])
    assert abs(loss - loss16) / abs(loss16) < 2e-3 and abs(loss - loss32) / abs(loss32) < 1e-2, (loss, loss16, loss32)
    cos32, cos_emu = [], []
    for k, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        if float(g32[k].norm()) < 1e-9:
            continue
        cos32.append(_cos(p.grad.cpu(), g32[k]))
        cos_emu.append(_cos(g16[k], g32[k]))
    cos32, cos_emu = np.array(cos32), np.array(cos_emu)
    med32, q32, med_emu, q_emu = np.median(cos32), np.quantile(cos32, 0.05), np.median(cos_emu), np.quantile(cos_emu, 0.05)
    print(f"    {len(cos32)} parameters; gradient cosine vs fp32: HIP median {med32:.4f} q05 {q32:.4f}; CPU bf16 emulation {med_emu:.4f} / {q_emu:.4f}")
    assert med32 > med_emu - 0.02 and q32 > q_emu - 0.05 and med32 > 0.99, (med32, med_emu, q32, q_emu)


def test_tiny_detector_eval_and_decode(golden):
    from range_view_3d_detection_amd.nn.decoders.range_decoder import RangeDecoder

    g = golden("tiny_model")
    backbone, head = load_tiny(g)
    backbone.eval()
    head.eval()
    data = {"features": g["features"].to(DEV), "cart": g["cart"].to(DEV), "mask": g["mask"].to(DEV)}
    with torch.no_grad():
        feats = backbone(data)
        outputs, _ = head(feats, data, return_loss=False)
    assert rel_err(outputs[1][0]["logits"], g["eval/logits"]) < 6e-2
    assert rel_err(outputs[1][0]["regressands"], g["eval/regressands"]) < 6e-2
    dec = RangeDecoder(True, True, [0, 15, 30], [15, 30, math.inf], [8, 2, 1])
    post = {"num_pre_nms": 50000, "num_post_nms": 1000, "nms_threshold": 0.3, "min_confidence": 0.1, "nms_mode": "WEIGHTED"}
    # decoder on the reference's own eval outputs: exact candidate set
    mo = {1: {"cart": data["cart"], "mask": data["mask"], 0: {"logits": g["eval/logits"].to(DEV), "regressands": g["eval/regressands"].to(DEV)}}}
    p, s, c, b = dec.decode(mo, post, {0: ["c"] * NCLS}, use_nms=False)
    assert torch.equal(c.cpu(), g["eval/dec_categories"]) and torch.equal(b.cpu(), g["eval/dec_batch_index"])
    assert rel_err(p, g["eval/dec_params"]) < 1e-5 and rel_err(s, g["eval/dec_scores"]) < 1e-6
    # full decode with weighted NMS on the HIP outputs runs and satisfies the wrapper's post-conditions
    p, s, c, b = dec.decode(outputs, post, {0: ["c"] * NCLS}, use_nms=True)
    assert p.shape[1] == 10 and p.shape[0] == s.shape[0] == c.shape[0] == b.shape[0] and p.shape[0] > 0
    assert torch.isfinite(p).all() and (s >= 0).all()


# ---------------------------------------------------------------------------------------------
# weighted NMS
# ---------------------------------------------------------------------------------------------
def _random_boxes(n, seed, spread=30.0):
    g = torch.Generator().manual_seed(seed)
    ctr = (torch.rand(n, 2, generator=g) - 0.5) * spread
    lw = 1.0 + 4.0 * torch.rand(n, 2, generator=g)
    yaw = (torch.rand(n, 1, generator=g) * 2 - 1) * math.pi
    z = torch.randn(n, 1, generator=g)
    h = 1.0 + torch.rand(n, 1, generator=g)
    cub = torch.cat([ctr, z, lw, h, yaw], dim=1)  # x,y,z,l,w,h,yaw
    scores = torch.rand(n, generator=g) * 0.9 + 0.1
    return cub, scores


@pytest.mark.parametrize("n,spread", [(1, 10.0), (65, 8.0), (700, 30.0), (3000, 60.0)])
def test_weighted_nms_matches_oracle_bit_exact(n, spread):
    from oracle import nms as onms
    from range_view_3d_detection_amd.math.ops import nms as hnms

    cub, scores = _random_boxes(n, n, spread)
    half = cub[:, 3:5] / 2
    rect = torch.cat([cub[:, :2] - half, cub[:, :2] + half, cub[:, 6:7]], dim=-1)
    data = torch.cat([cub[:, :6], cub[:, 6:7].sin(), cub[:, 6:7].cos()], dim=1)
    keep_o, out_o, cnt_o = onms.weighted_nms(rect, data, scores, 0.3, 0.5)
    keep, out, cnt = hnms.weighted_nms(rect.to(DEV), data.to(DEV), scores.to(DEV), 0.3, 0.5)
    assert torch.equal(keep.cpu(), keep_o) and torch.equal(cnt.cpu(), cnt_o)
    assert torch.equal(out.cpu(), out_o), float((out.cpu() - out_o).abs().max())
    assert (cnt > 0).all() and int(cnt.sum()) <= n  # nms.py:173-174 post-conditions; clusters are disjoint


def test_wnms_gpu_ffi_shim_under_the_reference_wrapper_logic():
    """``compat/weighted_nms_ext.wnms_gpu`` under the reference's own wrapper logic (``math/ops/nms.py:126-177`` restated line
    for line: sort, caller-allocated zero ``output`` / device ``count`` / HOST ``keep``, the two post-condition asserts)."""
    import sys

    from oracle import nms as onms
    from range_view_3d_detection_amd import compat
    from range_view_3d_detection_amd.math.ops import nms as hnms

    sys.path.insert(0, list(compat.__path__)[0])
    try:
        import weighted_nms_ext
    finally:
        sys.path.pop(0)

    def reference_shaped_weighted_nms(boxes, data2merge, scores, nms_threshold, merge_thresh):
        sorted_scores, order = scores.sort(0, descending=True)
        boxes = boxes[order].contiguous().float()
        data2merge = data2merge[order].contiguous().float()
        data2merge_score = torch.cat([data2merge, sorted_scores[:, None]], 1).contiguous().float()
        output = torch.zeros_like(data2merge_score)
        count = torch.zeros(boxes.size(0), dtype=torch.long, device=boxes.device)
        assert data2merge_score.dim() == 2
        keep = torch.zeros(boxes.size(0), dtype=torch.long)  # host
        num_out = weighted_nms_ext.wnms_gpu(boxes, data2merge_score, output, keep, count, nms_threshold, merge_thresh, boxes.device.index)
        keep = order[keep[:num_out].cuda(boxes.device)].contiguous()
        assert output[num_out:, :].sum() == 0
        assert (count[:num_out] > 0).all()
        return keep, output[:num_out, :], count[:num_out]

    for n, spread in ((1, 10.0), (700, 30.0), (3000, 60.0)):
        cub, scores = _random_boxes(n, n + 1, spread)
        half = cub[:, 3:5] / 2
        rect = torch.cat([cub[:, :2] - half, cub[:, :2] + half, cub[:, 6:7]], dim=-1)
        data = torch.cat([cub[:, :6], cub[:, 6:7].sin(), cub[:, 6:7].cos()], dim=1)
        keep, out, cnt = reference_shaped_weighted_nms(rect.to(DEV), data.to(DEV), scores.to(DEV), 0.3, 0.5)
        keep_o, out_o, cnt_o = onms.weighted_nms(rect, data, scores, 0.3, 0.5)
        assert torch.equal(keep.cpu(), keep_o) and torch.equal(cnt.cpu(), cnt_o) and torch.equal(out.cpu(), out_o)
        k2, o2, c2 = hnms.weighted_nms(rect.to(DEV), data.to(DEV), scores.to(DEV), 0.3, 0.5)
        assert torch.equal(keep, k2) and torch.equal(out, o2) and torch.equal(cnt, c2)
    # contract violations raise instead of falling back
    with pytest.raises(RuntimeError):
        weighted_nms_ext.wnms_gpu(rect, data, data, torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long), 0.3, 0.5, 0)


def test_rotated_iou_matches_oracle():
    from oracle import nms as onms
    from range_view_3d_detection_amd import _lib as L

    cub, _ = _random_boxes(200, 3, 12.0)
    half = cub[:, 3:5] / 2
    rect = torch.cat([cub[:, :2] - half, cub[:, :2] + half, cub[:, 6:7]], dim=-1).contiguous()
    ref = onms.pairwise_iou(rect.numpy(), rect.numpy())
    rd = rect.to(DEV)
    out = torch.empty((200, 200), dtype=torch.float32, device=DEV)
    L.call("rv_rotated_iou", L.ptr(rd), L.i64(200), L.ptr(rd), L.i64(200), L.ptr(out), L.stream_ptr())
    assert np.array_equal(out.cpu().numpy(), ref)
    assert abs(float(out.diagonal().min()) - 1.0) < 1e-5 and float(out.max()) <= 1.0 + 1e-6


def test_batched_multiclass_nms_matches_oracle():
    from oracle import nms as onms
    from range_view_3d_detection_amd.math.ops import nms as hnms

    cubs, scs, cats = [], [], []
    for b in range(2):
        cub, s = _random_boxes(500, 100 + b, 25.0)
        cubs.append(cub)
        scs.append(s * (torch.rand(500, generator=torch.Generator().manual_seed(b)) > 0.3))
        cats.append(torch.randint(0, 4, (500,), generator=torch.Generator().manual_seed(10 + b)))
    cub, sc, cat = torch.stack(cubs), torch.stack(scs), torch.stack(cats)
    bo, so, co, io = onms.batched_multiclass_nms(cub, sc, cat, 50000, 100, 0.3, 0.1)
    b, s, c, i = hnms.batched_multiclass_nms(cub.to(DEV), sc.to(DEV), cat.to(DEV), 50000, 100, 0.3, 0.1, "weighted")
    assert b.shape == bo.shape and torch.equal(c.cpu(), co) and torch.equal(i.cpu(), io)
    assert rel_err(b, bo) < 1e-6 and rel_err(s, so) < 1e-6


def test_fused_multiclass_nms_equals_per_class_loop():
    """One class-aware launch (rv_wnms_classes) == the reference-shaped per-class loop, row for row and bit for bit
    (classes do not interact; within a class the score order and therefore every merge sum is the same)."""
    from range_view_3d_detection_amd.math.ops import nms as hnms

    cub, s = _random_boxes(6000, 77, 60.0)
    cat = torch.randint(0, 26, (6000,), generator=torch.Generator().manual_seed(5))
    args = (cub.to(DEV), s.to(DEV), cat.to(DEV), 0.3, 50000, 40)  # num_post_nms = 40 < boxes kept per class: exercises the rank cut
    fused = hnms.weighted_multiclass_nms(*args)
    old = hnms.FUSED_CLASSES_MAX
    hnms.FUSED_CLASSES_MAX = 0
    try:
        loop = hnms.weighted_multiclass_nms(*args)
    finally:
        hnms.FUSED_CLASSES_MAX = old
    assert fused[0].shape == loop[0].shape and fused[0].shape[0] > 26 * 20
    for a, b in zip(fused, loop):
        assert torch.equal(a, b)


def test_detections_wire_format_from_device_decode(golden, tmp_path):
    """SURVEY §8f rank 2 end to end on the device path: eval forward -> RangeDecoder.decode (weighted NMS) -> build_dataframe
    -> one feather file per (log_id, timestamp_ns) -> read back.  The table holds exactly the decoder's rows (fp32 columns bit
    for bit), categories resolved through the task frame, rows of sweeps without a uuid dropped by the inner join.  (Parity
    with the reference's polars implementation stays unpinned: polars is not in the image; the schema / join rules are
    restated from math/ops/coding.py:11-76 and checked on CPU in test_host_cpu.py.)"""
    import pyarrow.feather as feather

    from range_view_3d_detection_amd.math.ops.coding import DETECTION_COLUMNS, build_dataframe, write_detections
    from range_view_3d_detection_amd.nn.decoders.range_decoder import RangeDecoder

    g = golden("tiny_model")
    backbone, head = load_tiny(g)
    backbone.eval()
    head.eval()
    data = {"features": g["features"].to(DEV), "cart": g["cart"].to(DEV), "mask": g["mask"].to(DEV)}
    with torch.no_grad():
        outputs, _ = head(backbone(data), data, return_loss=False)
    dec = RangeDecoder(True, True, [0, 15, 30], [15, 30, math.inf], [8, 2, 1])
    post = {"num_pre_nms": 50000, "num_post_nms": 1000, "nms_threshold": 0.3, "min_confidence": 0.1, "nms_mode": "WEIGHTED"}
    params, scores, cats, bidx = dec.decode(outputs, post, {0: ["c"] * NCLS}, use_nms=True)
    assert params.is_cuda and params.shape[0] > 0
    names = ["REGULAR_VEHICLE", "PEDESTRIAN", "BUS", "BICYCLE", "TRUCK"]
    uuids = {"batch_index": [0], "log_id": ["log-a"], "timestamp_ns": [315969904359876000]}  # sweep 1 has no uuid: dropped
    table = build_dataframe(params, scores, cats, bidx, uuids, names)
    keep = (bidx.int() == 0).cpu()
    assert table.num_rows == int(keep.sum()) and 0 < table.num_rows < params.shape[0]
    for j, col in enumerate(DETECTION_COLUMNS):
        assert np.array_equal(np.asarray(table.column(col)), params[:, j].float().cpu().numpy()[keep.numpy()]), col
    assert np.array_equal(np.asarray(table.column("score")), scores.float().cpu().numpy()[keep.numpy()])
    assert table.column("category").to_pylist() == [names[int(c)] for c in cats.int().cpu()[keep].tolist()]
    paths = write_detections(table, str(tmp_path), "run0")
    assert len(paths) == 1 and paths[0].endswith("predictions/run0/log-a/315969904359876000.feather")
    back = feather.read_table(paths[0])
    assert back.equals(table)


def test_batch_nms_device_path_equals_per_class_loop_and_oracle():
    """``rv_nms_sweeps`` (whole batch, device-resident: compaction, (class, score) ordering by counting, one scan workgroup
    per class, per-class top-k) returns the rows of the reference-shaped per-sweep / per-class loop bit for bit, and the
    oracle's rows to 1e-6; a sweep without candidates and a sweep that overflows the capacity take their fallbacks."""
    from oracle import nms as onms
    from range_view_3d_detection_amd.math.ops import nms as hnms

    cubs, scs, cats = [], [], []
    for b in range(4):
        cub, s = _random_boxes(3000, 200 + b, 45.0)
        keep = torch.rand(3000, generator=torch.Generator().manual_seed(b)) > (0.3 if b != 2 else 2.0)  # sweep 2: nothing above the threshold
        cubs.append(cub)
        scs.append(s * keep)
        cats.append(torch.randint(0, 26, (3000,), generator=torch.Generator().manual_seed(10 + b)))
    cub, sc, cat = torch.stack(cubs), torch.stack(scs), torch.stack(cats)
    args = (cub.to(DEV), sc.to(DEV), cat.to(DEV), 50000, 40, 0.3, 0.1, "weighted")
    fast = hnms.batched_multiclass_nms(*args, n_classes=26)
    old = hnms.FUSED_CLASSES_MAX
    hnms.FUSED_CLASSES_MAX = 0
    try:
        loop = hnms.batched_multiclass_nms(*args, n_classes=26)
    finally:
        hnms.FUSED_CLASSES_MAX = old
    assert fast[0].shape == loop[0].shape and fast[0].shape[0] > 3 * 26 * 20
    for a, b_ in zip(fast, loop):
        assert torch.equal(a, b_)
    assert set(fast[3].unique().tolist()) == {0.0, 1.0, 3.0}
    bo, so, co, io = onms.batched_multiclass_nms(cub, sc, cat, 50000, 40, 0.3, 0.1)
    assert torch.equal(fast[2].cpu(), co) and torch.equal(fast[3].cpu(), io)
    assert rel_err(fast[0], bo) < 1e-6 and rel_err(fast[1], so) < 1e-6
    # capacity overflow: a tiny capacity forces the fallback for every sweep -- same rows
    hnms.FUSED_CLASSES_MAX = 1024
    try:
        small = hnms.batched_multiclass_nms(*args, n_classes=26)
    finally:
        hnms.FUSED_CLASSES_MAX = old
    for a, b_ in zip(small, loop):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("n_cls,k", [(1, 2500), (3, 4001)])
def test_batch_nms_device_path_few_large_classes(n_cls, k):
    """Waymo-like class counts: one / three classes with thousands of boxes each (class segments spanning dozens of mask words,
    i.e. many blocks of the blocked scan) -- rows equal the per-class loop over the FFI bit for bit."""
    from range_view_3d_detection_amd.math.ops import nms as hnms

    cubs, scs, cats = [], [], []
    for b in range(2):
        cub, s = _random_boxes(k, 300 + b, 35.0)
        cubs.append(cub)
        scs.append(s)
        cats.append(torch.randint(0, n_cls, (k,), generator=torch.Generator().manual_seed(20 + b)))
    args = (torch.stack(cubs).to(DEV), torch.stack(scs).to(DEV), torch.stack(cats).to(DEV), 50000, 300, 0.3, 0.1, "weighted")
    fast = hnms.batched_multiclass_nms(*args, n_classes=n_cls)
    old = hnms.FUSED_CLASSES_MAX
    hnms.FUSED_CLASSES_MAX = 0
    try:
        loop = hnms.batched_multiclass_nms(*args, n_classes=n_cls)
    finally:
        hnms.FUSED_CLASSES_MAX = old
    assert fast[0].shape == loop[0].shape and fast[0].shape[0] >= 2 * n_cls * 100
    for a, b_ in zip(fast, loop):
        assert torch.equal(a, b_)


def test_batched_weight_repack_equals_the_per_layer_pack():
    """``engine.prepack_stale`` (every stale layer's gather + scatter image in ONE ``rv_pack_batch`` launch at the start of a
    training step) writes bit for bit what ``rv_pack_weight`` writes layer by layer, for every geometry of the model (3x3,
    1x1, strided, transposed (3,8)/s4 and (3,4)/s2), and leaves layers it does not cover (permuted weights) to the lazy path."""
    import ctypes

    from bench import Detector, build_model, synthetic_batch
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E

    torch.manual_seed(0)
    backbone, head = build_model("c32", 5)
    model = Detector(backbone, head).to(DEV).train()
    batch = synthetic_batch(1, 16, 128, seed=2, device=DEV, boxes_per_sweep=4, n_cls=5)
    model(batch).backward()  # first step: every layer packs itself (both forms)
    layers = [l for l in E._LAYERS if any(l.weight is p for p in model.parameters())]
    assert len(layers) > 40
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * torch.randn_like(p))  # an optimiser step: every image is stale now
    E.prepack_stale()
    covered = 0
    for l in layers:
        if l.in_perm is not None:
            assert l._version != (l.weight._version, l.weight.data_ptr())  # still stale: the lazy path repacks it
            continue
        assert l._version == (l.weight._version, l.weight.data_ptr()) and set(l._packed) == {"gather", "scatter"}
        n = L.load().rv_packed_weight_bytes(ctypes.byref(l.geom)) // 2
        ref = {f: torch.empty(n, dtype=torch.bfloat16, device=DEV) for f in ("gather", "scatter")}
        L.call("rv_pack_weight", ctypes.byref(l.geom), L.ptr(l.weight.detach().contiguous()), L.ptr(ref["gather"]), L.ptr(ref["scatter"]), L.stream_ptr())
        for f in ref:
            assert torch.equal(l._packed[f].view(torch.int16), ref[f].view(torch.int16)), (f, l.geom.kh, l.geom.kw, l.geom.stride_w)
        covered += 1
    assert covered > 40
    loss = model(batch)  # and the next step runs on the batched images
    assert torch.isfinite(loss)
    # inside a training step: an optimiser step that writes the parameters through raw pointers (optim.AdamW) must make
    # every image stale, and the next forward must re-pack them in the one batched launch
    from range_view_3d_detection_amd.optim import AdamW

    opt = AdamW(list(model.parameters()), lr=1e-2)
    loss.backward()
    before = {id(l): l._packed["gather"].clone() for l in layers if l.in_perm is None}
    opt.step()
    model(batch)
    changed = sum(int(not torch.equal(before[id(l)], l._packed["gather"])) for l in layers if l.in_perm is None)
    assert changed == len(before), (changed, len(before))


def test_fused_adamw_matches_torch_adamw_with_clipping():
    """``optim.AdamW`` (clip_grad_norm_ + AdamW of all parameters in two launches, ``rv_adamw_step``) against
    ``torch.nn.utils.clip_grad_norm_`` + ``torch.optim.AdamW`` under the recipe's ``OneCycleLR`` (which moves lr AND beta1
    every step): parameters and both moments within 2e-6 of their scale after five steps, the reported total norm 1e-6;
    tensor sizes straddle the 65536-element chunk, the 4-element vector width and a tensor of one element; the large
    gradients of step 2 make the clip active, the small ones of the other steps leave it inactive."""
    from range_view_3d_detection_amd.optim import AdamW

    gen = torch.Generator().manual_seed(0)
    sizes = [(1,), (3,), (7, 11), (1000,), (65536 + 5,), (3, 70001), (256, 256, 3, 3)]
    base = [torch.randn(s, generator=gen) for s in sizes]
    a = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    b = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    oa = AdamW(a, lr=1e-3, max_grad_norm=35.0)
    ob = torch.optim.AdamW(b, lr=1e-3)
    sa = torch.optim.lr_scheduler.OneCycleLR(oa, max_lr=0.0015, total_steps=20)
    sb = torch.optim.lr_scheduler.OneCycleLR(ob, max_lr=0.0015, total_steps=20)
    for it in range(5):
        grads = [torch.randn(s, generator=gen) * (5.0 if it == 2 else 0.01) for s in sizes]
        for p, q, g in zip(a, b, grads):
            p.grad = g.clone().to(DEV)
            q.grad = g.clone().to(DEV)
        want_norm = torch.nn.utils.clip_grad_norm_(b, 35.0)
        oa.step()
        ob.step()
        sa.step()
        sb.step()
        assert abs(float(oa.last_grad_norm) - float(want_norm)) < 1e-6 * float(want_norm)
        assert (float(want_norm) > 35.0) == (it == 2)
        assert oa.param_groups[0]["betas"] == ob.param_groups[0]["betas"] and oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
    for p, q in zip(a, b):
        for x, y in ((p, q), (oa.state[p]["exp_avg"], ob.state[q]["exp_avg"]), (oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"])):
            err = float((x.detach() - y.detach()).abs().max())
            assert err <= 2e-6 * max(float(y.detach().abs().max()), 1e-30), (tuple(p.shape), err)
        assert float(oa.state[p]["step"]) == float(ob.state[q]["step"]) == 5.0
    sd = oa.state_dict()  # the state keys / layout torch.optim.AdamW checkpoints carry
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["param_groups"][0]["weight_decay"] == 1e-2


def test_training_step_on_sweeps_without_annotations(golden):
    """Edge case of real data: no cuboid in any sweep of the batch (or in one of them).  The reference normalises by
    ``total_objects.clamp(1.0)`` and ``total_fg + additive_smoothing`` (nn/heads/detection_head.py:391-400), so the step is
    well defined: every target is background, the loss finite.  Asserted: what a training loop needs -- finite loss, finite
    gradients, all-background labels / no instance ids where no cuboid exists."""
    g = golden("tiny_model")
    backbone, head = load_tiny(g)
    backbone.train()
    head.train()
    ann = g["annotations"]
    for keep in (ann[:0], ann[ann[:, -1] == 0]):  # nothing at all; only the first sweep's cuboids
        backbone.zero_grad(set_to_none=True)
        head.zero_grad(set_to_none=True)
        data = {"features": g["features"].to(DEV), "cart": g["cart"].to(DEV), "mask": g["mask"].to(DEV), "annotations": keep}
        outputs, losses = head(backbone(data), data, return_loss=True)
        loss = losses["loss"]
        assert torch.isfinite(loss).all(), loss
        loss.backward()
        for mod in (backbone, head):
            for k, p in mod.named_parameters():
                assert p.grad is None or torch.isfinite(p.grad).all(), k
        labels = data[1][0]["classification_labels"]
        n_cls = outputs[1][0]["logits"].shape[1]  # (the background label is the class count)
        if keep.shape[0] == 0:
            assert int((labels != n_cls).sum()) == 0 and int((data[1][0]["panoptics"] != 0).sum()) == 0
        else:
            pan = data[1][0]["panoptics"]
            assert int((pan[1:] != 0).sum()) == 0 and int((pan[0] != 0).sum()) > 0  # sweep 1 has no cuboid, sweep 0 kept its own


def test_training_trajectory_vs_oracle_overfit():
    """The reference's only integration check is a workflow: ``scripts/debug-overfit.sh`` (overfit a tiny subsample, one device).
    Here as a parity test: SIX optimisation steps of the recipe (AdamW 1e-3, OneCycleLR stepped per step, gradient clipping at 35;
    ``nn/meta/arch.py:48-75``, ``conf/trainer/train.yaml:12``) on ONE synthetic batch, the HIP path -- fused clip + AdamW, head
    towers wide enough (256 channels) for the fused head-final backward -- against the ORACLE stepping the same model with
    ``torch.optim.AdamW`` + ``clip_grad_norm_`` on the CPU in fp32.  BatchNorm gates firmly open (well-conditioned regime, as
    in test_detector_gradients_vs_oracle); the per-step losses must track the oracle's (2e-3 at the first steps, 6e-2 at the
    sixth: bf16 storage against fp32, drifting apart as any two training runs do) and the batch must be overfitted (loss falls)."""
    from bench import Detector, build_model, synthetic_batch
    from oracle import model as om
    from oracle import targets as otgt
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

    n_cls, steps = 3, 6
    torch.manual_seed(0)
    backbone, head = build_model("c128", n_cls, 5)  # layers [128]*5, towers 256
    gen = torch.Generator().manual_seed(1)
    for m in list(backbone.modules()) + list(head.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=gen)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=gen) + 3.0
    sd = {**{f"backbone.{k}": v.clone() for k, v in backbone.state_dict().items()}, **{f"head.{k}": v.clone() for k, v in head.state_dict().items()}}
    batch = synthetic_batch(2, 16, 128, seed=11, device="cpu", boxes_per_sweep=6, n_cls=n_cls)
    torch.set_num_threads(min(16, torch.get_num_threads()))

    # oracle: the same recipe on CPU, fp32
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k]
    oparams = {k: torch.nn.Parameter(sd[k].clone()) for k in names}
    oopt, osched = configure_optimizers(list(oparams.values()), num_devices=1, batch_size=2, total_steps=steps + 8, fused=False)
    tg = otgt.compute_targets(batch["cart"], batch["annotations"], n_cls)
    oracle_losses = []
    for _ in range(steps):
        oopt.zero_grad(set_to_none=True)
        _, logits, reg = om.detector_forward(batch["features"], batch["cart"], {**sd, **oparams}, nm=om.Numerics(train=True))
        loss = otgt.detection_loss(logits, reg, batch["cart"], batch["mask"], tg, n_cls)["loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(oparams.values()), 35.0)
        oopt.step()
        osched.step()
        oracle_losses.append(float(loss.detach()))

    model = Detector(backbone, head).to(DEV).train()
    data = {k: (v.to(DEV) if k != "annotations" else v) for k, v in batch.items()}
    params = [p for _, p in model.named_parameters()]
    opt, sched = configure_optimizers(params, num_devices=1, batch_size=2, total_steps=steps + 8, fused=True, max_grad_norm=35.0)
    calls = []
    real = L._call

    def spy(name, *args):
        calls.append(name)
        return real(name, *args)

    L._call = spy
    try:
        hip_losses = []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = model(data)
            loss.backward()
            opt.step()
            sched.step()
            hip_losses.append(float(loss.detach()))
    finally:
        L._call = real
    print("oracle", [f"{v:.5f}" for v in oracle_losses])
    print("hip   ", [f"{v:.5f}" for v in hip_losses])
    assert calls.count("rv_head_final_bwd_sums") == 2 * steps and calls.count("rv_adamw_step") == steps  # both towers on the fused form; fused optimiser
    # two training runs in different arithmetic drift apart step by step (measured on an MI355X: 1.6e-4, 1.9e-4, 2.2e-3, 6.9e-3, 3.1e-2,
    # 2.8e-2 relative): the bound per step is ~10x the first steps' and ~2x the last steps' measured distance
    for i, (a, b, tol) in enumerate(zip(hip_losses, oracle_losses, (2e-3, 2e-3, 1e-2, 3e-2, 6e-2, 6e-2))):
        assert abs(a - b) < tol * abs(b), (i, hip_losses, oracle_losses)
    assert hip_losses[-1] < 0.9 * hip_losses[0] and oracle_losses[-1] < 0.9 * oracle_losses[0], (hip_losses, oracle_losses)
    # parameters after the six steps: direction of the total update against the oracle's
    upd_h = torch.cat([(p.detach().cpu().float() - sd[k]).flatten() for k, p in model.named_parameters()])
    upd_o = torch.cat([(oparams[k].detach() - sd[k]).flatten() for k, _ in model.named_parameters()])
    cos = float(torch.nn.functional.cosine_similarity(upd_h, upd_o, dim=0))
    print(f"    cosine of the six-step parameter update against the oracle's: {cos:.4f}")
    assert cos > 0.95, cos  # (measured 0.9905)


@pytest.mark.parametrize("case", ["one_sweep_without_boxes", "no_boxes_at_all"])
def test_training_step_with_empty_annotation_sets(case):
    """Edge case of ``compute_targets`` (nn/heads/detection_head.py:512-528): sweeps without annotations do not appear in
    ``annotations[:, -1].unique()`` and keep their initialised (background) targets; with no annotation at all the loop does not
    run.  The normalisers then clamp (``total_objects >= 1``, ``total_fg + 1``: :379-399).  HIP against the oracle: integer targets
    exact, loss 2e-2, every parameter gradient finite."""
    from bench import Detector, build_model, synthetic_batch
    from oracle import model as om
    from oracle import targets as otgt

    n_cls = 5
    torch.manual_seed(0)
    backbone, head = build_model("c32", n_cls, 5)
    gen = torch.Generator().manual_seed(1)
    for m in list(backbone.modules()) + list(head.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=gen)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=gen) + 3.0
    sd = {**{f"backbone.{k}": v.clone() for k, v in backbone.state_dict().items()}, **{f"head.{k}": v.clone() for k, v in head.state_dict().items()}}
    batch = synthetic_batch(2, 16, 128, seed=5, device="cpu", boxes_per_sweep=5, n_cls=n_cls)
    ann = batch["annotations"]
    batch["annotations"] = ann[ann[:, -1] == 0].clone() if case == "one_sweep_without_boxes" else ann[:0].clone()
    with torch.no_grad():
        _, logits_o, reg_o = om.detector_forward(batch["features"], batch["cart"], sd, nm=om.Numerics.bf16(train=True))
        tg = otgt.compute_targets(batch["cart"], batch["annotations"], n_cls)
        loss_o = float(otgt.detection_loss(logits_o, reg_o, batch["cart"], batch["mask"], tg, n_cls)["loss"])
    model = Detector(backbone, head).to(DEV).train()
    data = {k: (v.to(DEV) if k != "annotations" else v) for k, v in batch.items()}
    feats = model.backbone(data)
    _, losses = model.head(feats, data, return_loss=True)
    losses["loss"].backward()
    torch.cuda.synchronize()
    for k in ("classification_labels", "panoptics", "points_per_obj"):
        assert torch.equal(data[1][0][k].cpu(), tg[k]), k
    if case == "no_boxes_at_all":
        assert bool((tg["classification_labels"] == n_cls).all())
    else:
        assert bool((tg["classification_labels"][1] == n_cls).all()) and bool((tg["classification_labels"][0] != n_cls).any())
    loss = float(losses["loss"].detach())
    assert math.isfinite(loss) and abs(loss - loss_o) < 2e-2 * abs(loss_o), (loss, loss_o)
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
