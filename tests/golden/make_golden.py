"""Generate the golden fixtures in ``tests/golden/*.npz`` by running the REFERENCE itself.

Run in the build container only (needs ``/root/reference``):

    PYTORCH_JIT=0 python tests/golden/make_golden.py

The reference's own Python modules (``/root/reference/src/torchbox3d`` and
``/root/reference/converters/av2/utils.py``) are imported with stand-ins for the absent
third-party packages (``_ref_stubs.py``), executed on CPU in fp32 with fixed seeds, and
their inputs / weights / outputs / gradients are stored as small ``.npz`` files.  The
reference Python never travels to the GPU box; only these arrays do.  The oracle is
checked against them in ``tests/test_oracle_golden.py``.

Fixtures contain *data only* (arrays produced by or fed to the reference).
"""

from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()

from omegaconf import DictConfig, ListConfig  # noqa: E402  (stub)
from torchbox3d.math.numpy import conversions as ref_np  # noqa: E402
from torchbox3d.math.ops.coding import decode_range_view  # noqa: E402
from torchbox3d.nn.backbones.dla import RangeNet  # noqa: E402
from torchbox3d.nn.blocks import AggregationBlock, BasicBlock, ResidualBlock  # noqa: E402
from torchbox3d.nn.decoders.range_decoder import RangeDecoder, sample_by_range  # noqa: E402
from torchbox3d.nn.heads.dense_head import DenseHead  # noqa: E402
from torchbox3d.nn.heads.detection_head import COLS, DetectionHead, compute_targets  # noqa: E402
from torchbox3d.nn.modules.conv import Conv2dSame  # noqa: E402
from torchbox3d.nn.stems import MetaKernel, RangePartition  # noqa: E402
from torchbox3d.math.conversions import (  # noqa: E402
    cartesian_to_spherical_coordinates,
    spherical_to_cartesian_coordinates,
)
from torchbox3d.math.linalg.lie.SO3 import yaw_to_quat  # noqa: E402
from torchbox3d.nn.functional import varifocal_loss  # noqa: E402

ref_conv = _ref_stubs.load_converter_utils()


def npy(t):
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return a.astype(str) if a.dtype == object else a  # string columns read back from feather files


OUT_DIR = os.environ.get("RV3D_GOLDEN_OUT", HERE)  # (tests/test_oracle_golden.py regenerates into a temporary directory)


def _fixture(name: str) -> str:
    """A fixture another one is built from: the freshly generated file when this run has made it, else the committed one."""
    fresh = os.path.join(OUT_DIR, name + ".npz")
    return fresh if os.path.exists(fresh) else os.path.join(HERE, name + ".npz")


def save(name: str, **arrays) -> None:
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"{name}.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def randomize_bn(module: torch.nn.Module, g: torch.Generator) -> None:
    """Non-trivial affine + running statistics so that eval-mode parity means something."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=g)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=g)
            m.running_mean.data = 0.1 * torch.randn(m.running_mean.shape, generator=g)
            m.running_var.data = 0.5 + torch.rand(m.running_var.shape, generator=g)


def module_case(prefix: str, module: torch.nn.Module, inputs, g: torch.Generator, out: dict) -> None:
    """Train-mode forward + backward of ``sum(out * probe)``, then an eval-mode forward."""
    randomize_bn(module, g)
    sd0 = {k: v.clone() for k, v in module.state_dict().items()}
    module.train()
    xs = [x.clone().requires_grad_(True) for x in inputs]
    y = module(*xs)
    probe = torch.randn(y.shape, generator=g)
    (y * probe).sum().backward()
    for k, v in sd0.items():
        out[f"{prefix}/sd/{k}"] = v
    for i, x in enumerate(xs):
        out[f"{prefix}/in{i}"] = x
        out[f"{prefix}/gin{i}"] = x.grad
    out[f"{prefix}/out"] = y
    out[f"{prefix}/probe"] = probe
    for k, p in module.named_parameters():
        out[f"{prefix}/grad/{k}"] = p.grad
    for k, v in module.state_dict().items():
        if "running_" in k:
            out[f"{prefix}/sd_after/{k}"] = v.clone()
    module.load_state_dict(sd0)
    module.eval()
    with torch.no_grad():
        out[f"{prefix}/out_eval"] = module(*[x.detach() for x in xs])


# --------------------------------------------------------------------------------------
def gen_conv_blocks() -> None:
    g = torch.Generator().manual_seed(1)
    torch.manual_seed(1)  # module constructors draw their initial weights from the GLOBAL generator
    out: dict = {}
    x = torch.randn(2, 8, 6, 32, generator=g)
    for name, k, s in (("conv3_s11", 3, (1, 1)), ("conv3_s12", 3, (1, 2)), ("conv1_s12", 1, (1, 2)), ("conv1_s11", 1, 1)):
        m = Conv2dSame(8, 16, kernel_size=k, stride=s, bias=False)
        module_case(name, m, [x], g, out)
    module_case("basic_plain", BasicBlock(8, 8), [x], g, out)
    module_case("basic_proj_s12", BasicBlock(8, 16, stride=(1, 2), project=True), [x], g, out)
    module_case("basic_k1_proj", BasicBlock(5, 16, kernel_size=1, project=True), [torch.randn(2, 5, 6, 32, generator=g)], g, out)
    module_case("residual_s12_n3", ResidualBlock(8, 16, num_blocks=3, stride=(1, 2)), [x], g, out)
    x1 = torch.randn(2, 8, 6, 32, generator=g)
    x2a = torch.randn(2, 16, 6, 8, generator=g)
    x2b = torch.randn(2, 16, 6, 16, generator=g)
    module_case("agg_k8_s4", AggregationBlock(8, 16, 8, kernel_size=(3, 8), stride=(1, 4), padding=(1, 2), num_blocks=2), [x1, x2a], g, out)
    module_case("agg_k4_s2", AggregationBlock(8, 16, 8, kernel_size=(3, 4), stride=(1, 2), padding=(1, 1), num_blocks=1), [x1, x2b], g, out)
    save("conv_blocks", **out)


def synthetic_sweep(g: torch.Generator, B: int, H: int, W: int, n_feat: int = 5, drop: float = 0.1, smooth: bool = False):
    """Synthetic range image as SURVEY.md §8d defines it (row inclinations, column azimuths).

    ``smooth``: spatially coherent short ranges, so that boxes contain many pixels.
    """
    r = 1.5 + 78.5 * torch.rand(B, 1, H, W, generator=g)
    if smooth:
        az_ = torch.linspace(math.pi, -math.pi, W).view(1, 1, 1, W)
        inc_ = torch.linspace(0.2, -0.4, H).view(1, 1, H, 1)
        r = 6.0 + 1.5 * torch.sin(3 * az_) + 1.0 * torch.cos(9 * inc_) + 0.2 * torch.rand(B, 1, H, W, generator=g)
    mask = torch.rand(B, 1, H, W, generator=g) >= drop
    inc = torch.linspace(0.2, -0.4, H).view(1, 1, H, 1)
    az = torch.linspace(math.pi, -math.pi, W).view(1, 1, 1, W)
    cart = torch.cat([r * inc.cos() * az.cos(), r * inc.cos() * az.sin(), r * inc.sin().expand(B, 1, H, W)], dim=1) * mask
    intensity = torch.rand(B, 1, H, W, generator=g)
    feats = [intensity, r, cart[:, 0:1], cart[:, 1:2], cart[:, 2:3]]
    if n_feat == 6:
        feats = [torch.rand(B, 1, H, W, generator=g)] + feats
    features = torch.cat(feats, dim=1) * mask
    return features.float(), cart.float(), mask


def gen_meta_kernel() -> None:
    g = torch.Generator().manual_seed(2)
    torch.manual_seed(2)  # (as above: MetaKernel's conv / BatchNorm initialisation)
    out: dict = {}
    features, cart, _ = synthetic_sweep(g, 2, 6, 32)
    m = MetaKernel(in_channels=5, out_channels=16, num_neighbors=3, num_layers=2)
    module_case("meta", m, [features, cart], g, out)
    save("meta_kernel", **out)


def gen_range_partition() -> None:
    """RangePartition stem (nn/stems/__init__.py:88-135) with both projection kernel sizes, train forward + parameter gradients + eval
    forward; and RangeNet(stem_type="RANGE_PARTITION") end to end (the dispatch of nn/backbones/dla.py:164-171, 200-201).  Some ranges sit
    EXACTLY on a band edge (10, 15, 20, 30, 45, 60 m: the bands are closed and overlap)."""
    g = torch.Generator().manual_seed(12)
    torch.manual_seed(12)
    out: dict = {}
    features, cart, mask = synthetic_sweep(g, 2, 6, 32)
    edges = torch.tensor([10.0, 15.0, 20.0, 30.0, 40.0, 45.0, 60.0, 0.0])
    for i, d in enumerate(edges):  # points on the x axis: norm == d exactly in fp32
        cart[0, :, 1, 3 * i] = torch.tensor([float(d), 0.0, 0.0])
        mask[0, 0, 1, 3 * i] = True
        features[0, :, 1, 3 * i] = torch.randn(5, generator=g)
    out["features"], out["cart"], out["mask"] = features, cart, mask
    for tag, k in (("k1", 1), ("k3", 3)):
        m = RangePartition(in_channels=5, out_channels=16, num_neighbors=3, projection_kernel_size=k, num_layers=2)
        randomize_bn(m, g)
        sd0 = {n: v.clone() for n, v in m.state_dict().items()}
        m.train()
        y = m(features, cart, mask)
        probe = torch.randn(y.shape, generator=g)
        (y * probe).sum().backward()
        for n, v in sd0.items():
            out[f"{tag}/sd/{n}"] = v
        out[f"{tag}/out"], out[f"{tag}/probe"] = y, probe
        for n, p in m.named_parameters():
            if p.grad is not None:
                out[f"{tag}/grad/{n}"] = p.grad
        for n, v in m.state_dict().items():
            if "running_" in n:
                out[f"{tag}/sd_after/{n}"] = v.clone()
        m.load_state_dict(sd0)
        m.eval()
        with torch.no_grad():
            out[f"{tag}/out_eval"] = m(features, cart, mask)
    C = 16
    L = ListConfig([C] * 5)
    net = RangeNet(
        in_channels=5, layers=L, out_channels=C, projection_kernel_size=1, dataset_name="av2", num_neighbors=3,
        num_layers=2, stem_type="RANGE_PARTITION",
        _net=DictConfig(_target_="torchbox3d.nn.backbones.dla.RangeBackbone", in_channels=5, layers=L, out_channels=C),
    )
    randomize_bn(net, g)
    for n, v in net.state_dict().items():
        out[f"net/sd/{n}"] = v.clone()
    net.eval()
    with torch.no_grad():
        feats = net({"features": features, "cart": cart, "mask": mask})
    for s_, t in feats.items():
        out[f"net/eval_feat/{s_}"] = t
    save("range_partition", **out)


def make_annotations(g: torch.Generator, cart: torch.Tensor, mask: torch.Tensor, n_per: int, n_cls: int) -> np.ndarray:
    """(M,13) fp64 [xyz,lwh,qwxyz,task,offset,batch]; boxes centred on valid pixels, some nested."""
    rows = []
    B, _, H, W = cart.shape
    for b in range(B):
        valid = mask[b, 0].nonzero()
        pick = valid[torch.randperm(valid.shape[0], generator=g)[:n_per]]
        sweep_rows = []
        for i, (h, w) in enumerate(pick.tolist()):
            ctr = cart[b, :, h, w].double()
            lwh = torch.tensor([1.0, 1.0, 1.0]) + torch.rand(3, generator=g) * torch.tensor([5.0, 2.0, 2.0])
            if i % 3 == 1:  # a big box around the previous one => contested pixels
                ctr = torch.tensor(sweep_rows[-1][:3])
                lwh = torch.tensor(sweep_rows[-1][3:6]) * 2.5
            yaw = (torch.rand(1, generator=g).item() * 2 - 1) * math.pi
            q = [math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)]
            cat = int(torch.randint(0, n_cls, (1,), generator=g).item())
            sweep_rows.append(ctr.tolist() + lwh.double().tolist() + q + [0.0, float(cat), float(b)])
        sweep_rows.sort(key=lambda r: (r[10], r[11]))  # loader.py:700-704 sorts by (task_id, offset)
        rows += sweep_rows
    return np.asarray(rows, dtype=np.float64)


class Frame(_ref_stubs._PlFrame):
    pass


def gen_tiny_model() -> None:
    """RangeNet(META, layers=[16]*5) + DetectionHead, full fwd / targets / loss / bwd."""
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    B, H, W, C, NCLS = 2, 8, 64, 16, 5
    L = ListConfig([C] * 5)
    backbone = RangeNet(
        in_channels=5, layers=L, out_channels=C, projection_kernel_size=1, dataset_name="av2", num_neighbors=3,
        num_layers=2, stem_type="META",
        _net=DictConfig(_target_="torchbox3d.nn.backbones.dla.RangeBackbone", in_channels=5, layers=L, out_channels=C),
    )
    tasks = DictConfig({0: ListConfig([f"C{i}" for i in range(NCLS)])})
    tcfg = DictConfig(
        dataset_name="av2", tasks=tasks, enable_azimuth_invariant_targets=True,
        range_partitions=DictConfig({1: [0.0, math.inf]}), fpn_assignment_method=None, k=math.inf,
        affinity_fn="GAUSSIAN", normalize_affinities=False, sigma=0.75,
    )
    head = DetectionHead(
        fpn=DictConfig({1: 2 * C}), fpn_kernel_sizes=DictConfig({1: ListConfig([3, 3])}), targets_config=tcfg,
        num_classification_blocks=4, num_regression_blocks=4, final_kernel_size=1, tasks_cfg=tasks,
        task_in_channels=C, classification_weight=1.0, regression_weight=1.0,
        coding_weights=ListConfig([1.0] * 8), classification_head_channels=2 * C, regression_head_channels=2 * C,
        classification_normalization_method="FOREGROUND",
        _cls_loss=DictConfig(_target_="torchbox3d.nn.losses.classification.VarifocalLoss", alpha=0.75, gamma=2.0, reduction="none"),
        _regression_loss=DictConfig(_target_="torch.nn.L1Loss", reduction="none"),
    )
    randomize_bn(backbone, g)
    randomize_bn(head, g)
    # larger head weights than the N(0, 0.01) init so that logits / regressands are not ~constant
    for name, p in head.named_parameters():
        if name.endswith("0.weight"):
            p.data = 0.08 * torch.randn(p.shape, generator=g)
    head.classification_head["1"]["0"].blocks[-1][0].bias.data.fill_(-1.0)  # so that some scores pass 0.1
    features, cart, mask = synthetic_sweep(g, B, H, W, smooth=True)
    ann = make_annotations(g, cart, mask, n_per=6, n_cls=NCLS)
    frame = Frame({c: ann[:, i] for i, c in enumerate(COLS)})

    out: dict = {"features": features, "cart": cart, "mask": mask, "annotations": ann}
    for k, v in backbone.state_dict().items():
        out[f"sd/backbone.{k}"] = v.clone()
    for k, v in head.state_dict().items():
        out[f"sd/head.{k}"] = v.clone()

    backbone.train()
    head.train()
    data = {"features": features, "cart": cart, "mask": mask, "annotations": frame}
    feats = backbone(data)
    outputs, losses = head(feats, data, return_loss=True)
    losses["loss"].backward()
    for s, t in feats.items():
        out[f"feat/{s}"] = t
    out["logits"] = outputs[1][0]["logits"]
    out["regressands"] = outputs[1][0]["regressands"]
    for k in ("classification_labels", "panoptics", "regression_targets", "points_per_obj"):
        out[f"targets/{k}"] = data[1][0][k]
    out["targets/soft"] = data[1][0]["targets"]
    for k, v in losses.items():
        if isinstance(v, torch.Tensor) and "/" not in k:
            out[f"loss/{k}"] = v
    aux = losses["aux"][1][0]
    out["aux/foreground"] = aux["foreground"]
    out["aux/background"] = aux["background"]
    for k, p in backbone.named_parameters():
        out[f"grad/backbone.{k}"] = p.grad
    for k, p in head.named_parameters():
        out[f"grad/head.{k}"] = p.grad
    for k, v in list(backbone.state_dict().items()) + []:
        if "running_" in k:
            out[f"sd_after/backbone.{k}"] = v.clone()
    for k, v in head.state_dict().items():
        if "running_" in k:
            out[f"sd_after/head.{k}"] = v.clone()

    # eval-mode forward + decoder (no NMS: the NMS extension is absent)
    backbone.load_state_dict({k[len("sd/backbone."):]: torch.as_tensor(npy(v)) for k, v in out.items() if k.startswith("sd/backbone.")})
    head.load_state_dict({k[len("sd/head."):]: torch.as_tensor(npy(v)) for k, v in out.items() if k.startswith("sd/head.")})
    backbone.eval()
    head.eval()
    with torch.no_grad():
        data = {"features": features, "cart": cart, "mask": mask}
        feats = backbone(data)
        outputs, _ = head(feats, data, return_loss=False)
        out["eval/feat1"] = feats[1]
        out["eval/logits"] = outputs[1][0]["logits"]
        out["eval/regressands"] = outputs[1][0]["regressands"]
        dec = RangeDecoder(True, True, ListConfig([0, 15, 30]), ListConfig([15, 30, math.inf]), ListConfig([8, 2, 1]))
        params, scores, cats, bidx = dec.decode(
            outputs, DictConfig(num_pre_nms=50000, num_post_nms=1000, nms_threshold=0.3, min_confidence=0.1, nms_mode="WEIGHTED"),
            tasks, use_nms=False,
        )
        out["eval/dec_params"], out["eval/dec_scores"], out["eval/dec_categories"], out["eval/dec_batch_index"] = params, scores, cats, bidx
    save("tiny_model", **out)


def gen_decode() -> None:
    g = torch.Generator().manual_seed(4)
    B, H, W, NCLS = 2, 8, 64, 7
    _, cart, mask = synthetic_sweep(g, B, H, W, drop=0.2)
    reg = 0.5 * torch.randn(B, 8, H, W, generator=g)
    logits = 2.0 * torch.randn(B, NCLS, H, W, generator=g)
    logits[:, 2, :, ::5] = logits[:, 4, :, ::5]  # exact class ties -> lowest index must win
    out = {"cart": cart, "mask": mask, "regressands": reg, "logits": logits}
    out["decoded_inv"] = decode_range_view(reg.clone(), cart.clone(), True)
    out["decoded_plain"] = decode_range_view(reg.clone(), cart.clone(), False)
    scores, cats = (logits.sigmoid() * mask).max(dim=1, keepdim=True)
    s, c, b = sample_by_range(scores, cats, out["decoded_inv"], cart, (0, 15, 30), (15, 30, math.inf), (8, 2, 1))
    out["sampled_scores"], out["sampled_categories"], out["sampled_cuboids"] = s, c, b
    dec = RangeDecoder(True, True, ListConfig([0, 15, 30]), ListConfig([15, 30, math.inf]), ListConfig([8, 2, 1]))
    mo = {1: {"cart": cart, "mask": mask, 0: {"logits": logits, "regressands": reg}}}
    tasks = DictConfig({0: ListConfig(["c"] * NCLS)})
    post = DictConfig(num_pre_nms=50000, num_post_nms=1000, nms_threshold=0.3, min_confidence=0.1, nms_mode="WEIGHTED")
    p, sc, ca, bi = dec.decode(mo, post, tasks, use_nms=False)
    out["dec_params"], out["dec_scores"], out["dec_categories"], out["dec_batch_index"] = p, sc, ca, bi
    dec2 = RangeDecoder(True, False, ListConfig([0, 15, 30]), ListConfig([15, 30, math.inf]), ListConfig([8, 2, 1]))
    p, sc, ca, bi = dec2.decode(mo, post, tasks, use_nms=False)
    out["dense_params"], out["dense_scores"], out["dense_categories"], out["dense_batch_index"] = p, sc, ca, bi
    yaw = (torch.rand(33, 1, generator=g) * 2 - 1) * 4.0
    out["yaw"], out["quat"] = yaw, yaw_to_quat(yaw)
    x = 3.0 * torch.randn(4, 6, 5, 9, generator=g)
    t = torch.rand(4, 6, 5, 9, generator=g) * (torch.rand(4, 6, 5, 9, generator=g) > 0.7)
    out["vfl_x"], out["vfl_t"] = x, t
    out["vfl"] = varifocal_loss(x, t, 0.75, 2.0, "none")
    pts = 10 * torch.randn(257, 3, generator=g)
    out["s1_cart"] = pts
    out["s1_sph"] = cartesian_to_spherical_coordinates(pts)
    out["s1_back"] = spherical_to_cartesian_coordinates(out["s1_sph"])
    save("decode", **out)


def gen_projection() -> None:
    rng = np.random.default_rng(5)
    N, H, W = 6000, 64, 512
    out: dict = {}
    r = rng.uniform(0.3, 90.0, N)  # includes points closer than the 1 m cut
    az = rng.uniform(-math.pi, math.pi, N)
    inc = rng.uniform(-0.45, 0.25, N)
    cart = np.stack([r * np.cos(inc) * np.cos(az), r * np.cos(inc) * np.sin(az), r * np.sin(inc)], axis=1)
    # forced exact duplicates (ties -> earliest wins) and exact half-bin azimuths (round-half-even)
    cart[1000:1100] = cart[0:100]
    k = np.arange(200)
    az_half = (k + 0.5) * (math.tau / W) - math.pi
    cart[2000:2200] = np.stack([20 * np.cos(az_half), 20 * np.sin(az_half), np.zeros(200)], axis=1)
    laser = rng.integers(0, 64, N)
    intensity = rng.integers(0, 255, N).astype(np.float64)
    row_map = np.asarray(ref_conv.ROW_MAPPING_64)
    out["cart"], out["laser_numbers"], out["intensity"], out["row_mapping_64"] = cart, laser, intensity, row_map
    out["laser_mapping_32"] = np.asarray(ref_conv.LASER_MAPPING)
    sph = ref_conv.cart_to_sph(cart.copy())
    out["sph"] = sph.copy()
    hyb_c = ref_conv.build_range_view_coordinates(cart.copy(), sph.copy(), laser.copy(), row_map, H, W)
    hyb_l = ref_np.build_range_view_coordinates(cart.copy(), sph.copy(), laser.copy(), row_map, H, W)
    out["hybrid_converter"], out["hybrid_library"] = hyb_c, hyb_l
    feats = np.concatenate([sph, cart, intensity[:, None]], axis=1).transpose(1, 0)
    for tag, hyb in (("converter", hyb_c), ("library", hyb_l)):
        idx = np.ascontiguousarray(hyb[:, :2].transpose(1, 0).astype(int))
        out[f"indices_{tag}"] = idx
        out[f"image_{tag}"] = ref_conv.z_buffer(idx, hyb[:, 2], feats, height=H, width=W)
    out["features"] = feats
    # library sph<->cart twins
    out["np_sph_to_cart"] = ref_np.sph_to_cart(sph.copy())
    # loader W-pad rule (prototype/loader.py:792-815)
    try:
        from torchbox3d.prototype.loader import subsample_range_view

        g = torch.Generator().manual_seed(6)
        for ds, w in (("av2", 1800), ("waymo", 2650)):
            rv = torch.randn(2, 2, w, generator=g)
            m = torch.rand(1, 2, w, generator=g) > 0.2
            c = torch.randn(3, 2, w, generator=g)
            for mode in ("constant", "circular"):
                a, b_, c_ = subsample_range_view(rv.clone(), m.clone().float(), c.clone(), ds, 1, mode)
                out[f"pad/{ds}/{mode}/rv_in"], out[f"pad/{ds}/{mode}/mask_in"] = rv, m.float()
                out[f"pad/{ds}/{mode}/rv"], out[f"pad/{ds}/{mode}/mask"] = a, b_
    except Exception as exc:  # loader imports many absent packages
        print("loader.subsample_range_view not importable:", repr(exc))
    save("projection", **out)


def gen_augment() -> None:
    """Loader augmentations (prototype/loader.py:825-990) run by the reference itself on a synthetic 8 x 64 sweep table
    (Float32 columns, as the converter writes them) and a 6-box annotation table (Float64): one fixture per augmentation,
    plus the chain of the shipped recipe (conf/model/baseline.yaml:12-21: flip_azimuth, random_rotation, random_global_scale)
    followed by random_global_translation.  The random draws are reproduced by re-seeding ``random`` and drawing in the
    order the reference does, so the fixtures also hold the parameters."""
    import random

    import polars as pl  # stub
    from torchbox3d.prototype import loader as ref_loader

    H, W = 8, 64
    rng = np.random.default_rng(21)
    inc = np.linspace(0.2, -0.4, H)[:, None]
    az = np.linspace(math.pi, -math.pi, W)[None, :]
    r = (20.0 + 15.0 * np.sin(3 * az) + 10.0 * np.cos(7 * inc) + rng.random((H, W))).astype(np.float32)
    keep = rng.random((H, W)) >= 0.1
    r = r * keep
    cols = {
        "x": (r * np.cos(inc) * np.cos(az)).astype(np.float32), "y": (r * np.cos(inc) * np.sin(az)).astype(np.float32),
        "z": (r * np.sin(inc) * np.ones_like(az)).astype(np.float32), "intensity": (rng.random((H, W)) * keep).astype(np.float32),
        "laser_number": (np.arange(H)[:, None] * np.ones((1, W)) * keep).astype(np.float32), "range": r.astype(np.float32),
    }
    names = list(cols)
    sweep0 = pl.DataFrame({k: v.reshape(-1) for k, v in cols.items()})
    yaw = rng.uniform(-math.pi, math.pi, 6)
    ann = {"tx_m": rng.normal(0, 20, 6), "ty_m": rng.normal(0, 20, 6), "tz_m": rng.normal(0, 1, 6), "length_m": rng.uniform(1, 6, 6),
           "width_m": rng.uniform(1, 3, 6), "height_m": rng.uniform(1, 3, 6), "qw": np.cos(yaw / 2), "qx": np.zeros(6), "qy": np.zeros(6),
           "qz": np.sin(yaw / 2)}
    ann_names = list(ann)
    ann0 = pl.DataFrame(ann)
    cfg = DictConfig({"height": H, "width": W})

    def table(frame, nm, n):
        return np.stack([np.asarray(frame[k], dtype=np.float64) for k in nm]).reshape(len(nm), *n)

    out = {"sweep/in": table(sweep0, names, (H, W)), "ann/in": table(ann0, ann_names, (6,)), "column_names": np.array(names),
           "ann_column_names": np.array(ann_names)}

    def record(tag, sweep, annot, **params):
        out[f"{tag}/sweep"] = table(sweep.collect(), names, (H, W))
        out[f"{tag}/ann"] = table(annot.collect(), ann_names, (6,))
        out[f"{tag}/sweep_is_f64"] = np.array([sweep.collect()[k].dtype == np.float64 for k in names])
        for k, v in params.items():
            out[f"{tag}/{k}"] = np.asarray(v, dtype=np.float64)

    # --- each augmentation alone ---
    random.seed(1)
    s, a = ref_loader.flip_azimuth(sweep0.lazy(), ann0.lazy(), cfg, p=1.0)
    record("flip", s, a)
    random.seed(2)
    s, a = ref_loader.random_rotation(sweep0.lazy(), ann0.lazy(), cfg, low=-0.78539816, high=0.78539816, p=1.0)
    random.seed(2)
    random.random()
    record("rotation", s, a, theta=random.uniform(-0.78539816, 0.78539816))
    random.seed(3)
    s, a = ref_loader.random_global_scale(sweep0.lazy(), ann0.lazy(), low=0.95, high=1.05)
    random.seed(3)
    record("scale", s, a, scale=random.uniform(0.95, 1.05))
    random.seed(4)
    s, a = ref_loader.random_global_translation(sweep0.lazy(), ann0.lazy(), std_x=0.5, std_y=0.5, std_z=0.2)
    random.seed(4)
    record("translation", s, a, t=[random.normalvariate(0, 0.5), random.normalvariate(0, 0.5), random.normalvariate(0, 0.2)])
    # --- a negative rotation (roll to the left) and a not-applied flip (p = 0) ---
    random.seed(5)
    s, a = ref_loader.random_rotation(sweep0.lazy(), ann0.lazy(), cfg, low=-3.0, high=-2.0, p=1.0)
    random.seed(5)
    random.random()
    record("rotation_neg", s, a, theta=random.uniform(-3.0, -2.0))
    # --- the shipped chain (+ translation) ---
    random.seed(6)
    s, a = ref_loader.flip_azimuth(sweep0.lazy(), ann0.lazy(), cfg, p=1.0)
    s, a = ref_loader.random_rotation(s, a, cfg, low=-0.78539816, high=0.78539816, p=1.0)
    s, a = ref_loader.random_global_scale(s, a, low=0.95, high=1.05)
    s, a = ref_loader.random_global_translation(s, a, std_x=0.5, std_y=0.5, std_z=0.2)
    random.seed(6)
    random.random()
    random.random()
    theta = random.uniform(-0.78539816, 0.78539816)
    scale = random.uniform(0.95, 1.05)
    record("chain", s, a, theta=theta, scale=scale, t=[random.normalvariate(0, 0.5), random.normalvariate(0, 0.5), random.normalvariate(0, 0.2)])
    save("augment", **out)


def gen_loader_item() -> None:
    """The loader's per-sweep contract (prototype/loader.py:568-705, ``DataLoader.__getitem__``): a range-view feather table
    (H*W rows x named columns) -> ``features`` (F,H,W), ``cart`` (3,H,W), ``mask`` (1,H,W), W padded by ``subsample_range_view``.
    The reference's own ``__getitem__`` is run (unbound, on a stand-in ``self``: the constructor walks a dataset directory)
    on two synthetic tables written as feather files: AV2 columns with ``filter_roi`` and circular padding
    (conf/model/baseline.yaml:57, range_view.yaml:141-149), Waymo columns with tanh(intensity) and constant padding
    (conf/experiment/base-waymo.yaml:26-33)."""
    import tempfile
    import types
    from pathlib import Path

    import polars as pl  # stub
    import pyarrow as pa
    import pyarrow.feather as feather
    from torchbox3d.prototype import loader as ref_loader

    pl.scan_ipc = _ref_stubs._pl_scan_ipc
    H, W = 8, 64
    out = {}
    for tag, ds, names, roi, mode in (("av2", "av2", ["intensity", "range", "x", "y", "z"], True, "circular"),
                                      ("waymo", "waymo", ["elongation", "intensity", "range", "x", "y", "z"], False, "constant"),
                                      # the `view` feature (loader.py:605-624): laser rows through the reverse ROW_MAPPING_64, upper / lower view id
                                      ("av2_view", "av2", ["intensity", "laser_number", "view", "range", "x", "y", "z"], True, "circular")):
        rng = np.random.default_rng({"av2": 31, "waymo": 32}.get(tag, 34))
        inc = np.linspace(0.2, -0.4, H)[:, None]
        az = np.linspace(math.pi, -math.pi, W)[None, :]
        r = (20.0 + 15.0 * np.sin(3 * az) + 10.0 * np.cos(7 * inc) + rng.random((H, W))).astype(np.float32)
        keep = rng.random((H, W)) >= 0.1
        r = (r * keep).astype(np.float32)
        cols = {"x": (r * np.cos(inc) * np.cos(az)).astype(np.float32), "y": (r * np.cos(inc) * np.sin(az)).astype(np.float32),
                "z": (r * np.sin(inc) * np.ones_like(az)).astype(np.float32), "range": r,
                "intensity": ((rng.random((H, W)) * (255.0 if ds == "av2" else 3.0)) * keep).astype(np.float32),
                "elongation": (rng.random((H, W)) * keep).astype(np.float32),
                "laser_number": (np.arange(H)[:, None] * np.ones((1, W)) * keep).astype(np.float32),
                "is_within_roi": rng.random((H, W)) >= 0.25}
        if tag == "av2_view":  # laser ids over the whole 0..63 range, as the raw sweeps have them
            cols["laser_number"] = (rng.integers(0, 64, (H, W)) * keep).astype(np.float32)
        tmp = Path(tempfile.mkdtemp())
        lidar = tmp / "sweep.feather"
        feather.write_feather(pa.table({k: v.reshape(-1) for k, v in cols.items()}), str(lidar), compression="uncompressed")
        annp = tmp / "annotations.feather"
        feather.write_feather(pa.table({"timestamp_ns": np.array([7, 7], dtype=np.int64), "num_interior_pts": np.array([5, 0], dtype=np.int64),
                                        "category": np.array(["REGULAR_VEHICLE", "BUS"])}), str(annp), compression="uncompressed")
        me = types.SimpleNamespace(
            metadata=pl.DataFrame({"log_id": np.array(["log0"]), "timestamp_ns": np.array([7], dtype=np.int64)}),
            categories=["REGULAR_VEHICLE"], annotations_path=lambda log_id: annp, lidar_path=lambda log_id, ts: lidar,
            range_view_config=DictConfig({"feature_column_names": names, "filter_roi": roi, "height": H, "width": W}),
            split_name="val", augmentations_config=None, dataset_name=ds, enable_database=False, db_config=None, x_stride=1,
            padding_mode=mode, targets_config=None)
        datum = ref_loader.DataLoader.__getitem__(me, 0)
        for k in ("intensity", "range", "x", "y", "z", "elongation", "is_within_roi") + (("laser_number",) if tag == "av2_view" else ()):
            out[f"{tag}/table/{k}"] = cols[k].reshape(-1)
        out[f"{tag}/feature_column_names"] = np.array(names)
        out[f"{tag}/padding_mode"] = np.array(mode)
        out[f"{tag}/filter_roi"] = np.array(roi)
        out[f"{tag}/features"] = datum["features"].numpy()
        out[f"{tag}/cart"] = datum["cart"].numpy()
        out[f"{tag}/mask"] = datum["mask"].numpy()
        print(tag, "features", tuple(datum["features"].shape), "mask", datum["mask"].dtype, float(datum["mask"].float().mean()))
    save("loader_item", **out)


def gen_raw_sweep() -> None:
    """The converter's path from a RAW AV2 sweep to the range image (converters/av2/export.py:70-133 driving
    converters/av2/utils.py): unmotion_compensate (:231-295), correct_laser_numbers (:211-228), build_range_view (:32-105:
    ego -> sensor SE3, spherical binning, z-buffer), run by the reference itself on a synthetic sweep, a synthetic pose
    track and synthetic extrinsics.  The dataset tables the reference indexes (LASER_MAPPING, ROW_MAPPING_64 / _32) are
    stored as data: the device path takes them as arguments."""
    import polars as pl  # stub
    from scipy.spatial.transform import Rotation, Slerp
    from torchbox3d.datasets.argoverse.constants import LASER_MAPPING, LOG_IDS, ROW_MAPPING_32, ROW_MAPPING_64

    rng = np.random.default_rng(33)
    P, N, H, W = 14, 4000, 64, 512
    ts = (315969904359876000 + np.cumsum(rng.integers(95_000_000, 105_000_000, P))).astype(np.int64)
    yaw = 0.3 + 0.02 * np.arange(P) + 0.003 * rng.normal(size=P)
    rp = 0.01 * rng.normal(size=(P, 2))
    rots = Rotation.from_euler("xyz", np.stack([rp[:, 0], rp[:, 1], yaw], axis=1))
    q_xyzw = rots.as_quat()
    trans = np.stack([1000 + 1.2 * np.arange(P) + 0.01 * rng.normal(size=P), 500 + 0.3 * np.arange(P), 10 + 0.01 * rng.normal(size=P)], axis=1)
    poses = pl.DataFrame({"timestamp_ns": ts, "qw": q_xyzw[:, 3], "qx": q_xyzw[:, 0], "qy": q_xyzw[:, 1], "qz": q_xyzw[:, 2],
                          "tx_m": trans[:, 0], "ty_m": trans[:, 1], "tz_m": trans[:, 2]})
    slerp = Slerp(ts, Rotation.from_quat(poses.select(ref_conv.QUATERNION_WXYZ).to_numpy()))
    target = int(ts[6])
    offset_ns = rng.integers(-20_000_000, 95_000_000, N).astype(np.int32)
    offset_ns[:5] = [int(ts[0] - target) - 7, int(ts[-1] - target) + 7, int(ts[0] - target), int(ts[-1] - target), int(ts[7] - target)]  # outside / on the ends / on a pose
    r = rng.uniform(0.5, 80.0, N)
    az, inc = rng.uniform(-math.pi, math.pi, N), rng.uniform(-0.4, 0.25, N)
    xyz = np.stack([r * np.cos(inc) * np.cos(az) + 1.35, r * np.cos(inc) * np.sin(az), r * np.sin(inc) + 1.64], axis=1).astype(np.float32).astype(np.float64)
    laser = rng.integers(0, 64, N).astype(np.uint8)
    lidar = pl.DataFrame({"x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2], "intensity": rng.integers(0, 256, N).astype(np.uint8),
                          "laser_number": laser, "offset_ns": offset_ns, "is_within_roi": rng.random(N) > 0.3})
    out = {"poses/timestamp_ns": ts, "poses/q_wxyz": np.stack([q_xyzw[:, 3], q_xyzw[:, 0], q_xyzw[:, 1], q_xyzw[:, 2]], axis=1), "poses/t": trans,
           "sweep/timestamp_ns": np.int64(target), "sweep/xyz": xyz, "sweep/offset_ns": offset_ns, "sweep/laser_number": laser,
           "sweep/intensity": np.asarray(lidar["intensity"]), "sweep/is_within_roi": np.asarray(lidar["is_within_roi"]),
           "tables/LASER_MAPPING": np.asarray(LASER_MAPPING), "tables/ROW_MAPPING_64": np.asarray(ROW_MAPPING_64),
           "tables/ROW_MAPPING_32": np.asarray(ROW_MAPPING_32)}
    # ---- unmotion_compensate ----
    um = ref_conv.unmotion_compensate(lidar, poses, target, slerp)
    out["unmotion/kept"] = np.isin(np.arange(N), np.nonzero((target + offset_ns.astype(np.int64) > ts[0]) & (target + offset_ns.astype(np.int64) < ts[-1]))[0])
    out["unmotion/xyz_p"] = np.stack([np.asarray(um["x_p"]), np.asarray(um["y_p"]), np.asarray(um["z_p"])], axis=1)
    assert out["unmotion/xyz_p"].shape[0] == int(out["unmotion/kept"].sum())
    # ---- correct_laser_numbers: a log with the laser-ordering defect and one without, both image heights ----
    lz = np.asarray(um["laser_number"]).astype(np.int64)
    out["laser/in"] = lz
    out["laser/h64_affected"] = ref_conv.correct_laser_numbers(lz.copy(), LOG_IDS[0], 64)
    out["laser/h64_plain"] = ref_conv.correct_laser_numbers(lz.copy(), "not-a-listed-log", 64)
    out["laser/h32_affected"] = ref_conv.correct_laser_numbers(lz.copy() % 32, LOG_IDS[0], 32)
    # ---- build_range_view (ego -> up_lidar frame, binning, z-buffer) ----
    ext_rot = Rotation.from_euler("xyz", [[0.002, -0.003, 0.01], [3.1, 0.0, 0.0]])
    eq = ext_rot.as_quat()
    extrinsics = pl.DataFrame({"sensor_name": np.array(["up_lidar", "down_lidar"]), "qw": eq[:, 3], "qx": eq[:, 0], "qy": eq[:, 1], "qz": eq[:, 2],
                               "tx_m": np.array([1.35, 1.355]), "ty_m": np.array([0.01, -0.01]), "tz_m": np.array([1.64, 1.515])})
    um = um.with_columns(pl.Series("laser_number", out["laser/h64_affected"]))
    feats = um.select(("x", "y", "z", "intensity", "laser_number", "is_within_roi")).to_numpy()
    rv = ref_conv.build_range_view(um, extrinsics=extrinsics, features=feats, sensor_name="up_lidar", height=H, width=W)
    out["extrinsics/q_wxyz"] = np.array([eq[0, 3], eq[0, 0], eq[0, 1], eq[0, 2]])
    out["extrinsics/t"] = np.array([1.35, 0.01, 1.64])
    names = ["x", "y", "z", "intensity", "laser_number", "is_within_roi", "timedelta_ns", "range"]
    out["range_view/column_names"] = np.array(names)
    out["range_view/image"] = np.stack([np.asarray(rv[k]).astype(np.float64) for k in names]).reshape(len(names), H, W)
    save("raw_sweep", **out)


def _install_wnms_stand_in():
    """``weighted_nms_ext.wnms_gpu`` (TorchEx, absent) bound to a CPU callable with the extension's own signature
    (``math/ops/nms.py:161-170``): fills ``output`` / ``keep`` / ``count`` IN PLACE and returns ``num_out``, running the
    DECLARED kernel semantics (``oracle/c/oracle.c::rvo_weighted_nms``, ``oracle/nms.py`` header).  Everything AROUND
    that call -- the class loop over ``unique()``, both ``topk`` cuts, the ``[6,1,1,1]`` split, ``atan2``, float categories
    and batch index, the ``min_confidence`` filter, the empty-result shapes -- is then the REFERENCE's own Python."""
    import ctypes

    import weighted_nms_ext  # the stub module registered by _ref_stubs.install()

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import nms as onms

    lib = onms.lib()

    def wnms_gpu(boxes, data2merge_score, output, keep, count, nms_thresh, merge_thresh, device_index):
        assert boxes.dtype == torch.float32 and boxes.is_contiguous() and data2merge_score.is_contiguous()
        assert output.is_contiguous() and keep.dtype == torch.long and count.dtype == torch.long
        n, d = data2merge_score.shape
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        cast = lambda t, ty: ctypes.cast(t.data_ptr(), ty)  # noqa: E731
        return int(lib.rvo_weighted_nms(cast(boxes, fp), cast(data2merge_score, fp), ctypes.c_int64(n), ctypes.c_int(d),
                                        ctypes.c_float(nms_thresh), ctypes.c_float(merge_thresh), cast(output, fp), cast(keep, ip),
                                        cast(count, ip)))

    weighted_nms_ext.wnms_gpu = wnms_gpu
    # nms.py:171 moves ``keep`` with ``.cuda(boxes.device)``: on this CPU run that is the identity
    torch.Tensor.cuda = lambda self, *a, **k: self


def gen_nms_wrapper() -> None:
    """The reference's OWN weighted-NMS wrapper logic (``math/ops/nms.py:64-123, 126-177, 181-266``,
    ``nn/decoders/range_decoder.py:100-124``) run over the stand-in above.  The inner kernel's arithmetic stays
    declared (parity unpinned, TorchEx source absent); what these fixtures pin is every line of the reference that
    surrounds it."""
    from torchbox3d.math.ops import nms as ref_nms

    _install_wnms_stand_in()
    g = torch.Generator().manual_seed(41)
    out: dict = {}

    def candidates(B, K, n_cls, absent, heavy, cluster_frac=0.6):
        """(B,K,7) boxes in clusters around a few hundred centres (so that merging and suppression both happen), scores,
        int64 categories.  ``absent``: a class id that never occurs; ``heavy``: a class that gets half of the candidates."""
        centres = (torch.rand(B, 160, 2, generator=g) - 0.5) * 120.0
        which = torch.randint(0, 160, (B, K), generator=g)
        ctr = torch.gather(centres, 1, which[..., None].expand(B, K, 2))
        scattered = torch.rand(B, K, generator=g) > cluster_frac
        ctr = torch.where(scattered[..., None], (torch.rand(B, K, 2, generator=g) - 0.5) * 150.0, ctr + 0.35 * torch.randn(B, K, 2, generator=g))
        z = torch.randn(B, K, 1, generator=g)
        lwh = torch.tensor([4.5, 2.0, 1.7]) * (1.0 + 0.15 * torch.randn(B, K, 3, generator=g)).clamp(0.5, 1.6)
        yaw = (torch.rand(B, K, 1, generator=g) * 2 - 1) * math.pi
        yaw_c = torch.gather((torch.rand(B, 160, generator=g) * 2 - 1) * math.pi, 1, which)[..., None] + 0.05 * torch.randn(B, K, 1, generator=g)
        yaw = torch.where(scattered[..., None], yaw, yaw_c)
        cub = torch.cat([ctr, z, lwh, yaw], dim=-1).float()
        sc = torch.rand(B, K, generator=g).pow(2.0).float()
        cat = torch.randint(0, n_cls, (B, K), generator=g)
        cat = torch.where(torch.rand(B, K, generator=g) < 0.5, torch.full_like(cat, heavy), cat)
        cat = torch.where(cat == absent, torch.full_like(cat, (absent + 1) % n_cls), cat)
        return cub, sc, cat

    # ---- (a) 3 sweeps x 2000 candidates x 5 classes ----
    B, K, NCLS = 3, 2000, 5
    cub, sc, cat = candidates(B, K, NCLS, absent=3, heavy=1)
    sc[1] = sc[1] * 0.0999  # sweep 1: nothing reaches min_confidence = 0.1 (nms.py:218-220 skips it)
    # sweep 2: exact score ties inside a class, between overlapping boxes (members of one cluster) and between far-apart boxes
    tie = torch.arange(0, 400, 4)
    sc[2, tie] = sc[2, tie + 1]
    sc[2, 800:840] = 0.5
    out["a/cuboids"], out["a/scores"], out["a/categories"] = cub, sc, cat
    for tag, pre, post in (("post1000", 50000, 1000), ("post40", 50000, 40), ("pre150", 150, 1000)):
        p, s, c, b = ref_nms.batched_multiclass_nms(cub.clone(), sc.clone(), cat.clone(), num_pre_nms=pre, num_post_nms=post,
                                                    iou_threshold=0.3, min_confidence=0.1, nms_mode="weighted")
        out[f"a/{tag}/params"], out[f"a/{tag}/scores"], out[f"a/{tag}/categories"], out[f"a/{tag}/batch_index"] = p, s, c, b
        out[f"a/{tag}/cfg"] = np.array([pre, post, 0.3, 0.1])
        print("a", tag, tuple(p.shape), s.dtype, c.dtype, b.dtype, "rows per sweep", [int((b == i).sum()) for i in range(B)],
              "rows per class", [int((c == j).sum()) for j in range(NCLS)])
    # one sweep, one call of weighted_multiclass_nms (nms.py:64-123) and of weighted_nms (nms.py:126-177) themselves
    m = sc[0] >= 0.1
    p, s, c = ref_nms.weighted_multiclass_nms(cub[0, m], sc[0, m], cat[0, m], iou_threshold=0.3, num_pre_nms=50000, num_post_nms=40)
    out["a/multiclass/params"], out["a/multiclass/scores"], out["a/multiclass/categories"] = p, s, c
    sel = m & (cat[0] == 1)
    b1 = cub[0, sel]
    rect = torch.cat([b1[:, :2] - b1[:, 3:5] / 2, b1[:, :2] + b1[:, 3:5] / 2, b1[:, 6:7]], dim=-1)
    data = torch.cat([b1[:, :6], b1[:, 6:7].sin(), b1[:, 6:7].cos()], dim=1)
    keep, merged, count = ref_nms.weighted_nms(rect, data, sc[0, sel], nms_threshold=0.3, merge_thresh=0.5)
    out["a/wnms/boxes"], out["a/wnms/data"], out["a/wnms/scores"] = rect, data, sc[0, sel]
    out["a/wnms/keep"], out["a/wnms/output"], out["a/wnms/count"] = keep, merged, count
    # all sweeps empty: the (0,7) / (0,1) results of nms.py:248-251
    p, s, c, b = ref_nms.batched_multiclass_nms(cub[1:2].clone(), sc[1:2].clone(), cat[1:2].clone(), num_pre_nms=50000, num_post_nms=1000,
                                                iou_threshold=0.3, min_confidence=0.1, nms_mode="WEIGHTED")
    out["a/empty/params_shape"], out["a/empty/scores_shape"] = np.array(p.shape), np.array(s.shape)
    out["a/empty/categories_shape"], out["a/empty/batch_index_shape"] = np.array(c.shape), np.array(b.shape)
    out["a/empty/categories_is_int64"] = np.array(c.dtype == torch.int64)

    # ---- (b) RangeDecoder.decode(use_nms=True) on the tiny model's eval outputs (tests/golden/tiny_model.npz) ----
    tm = np.load(_fixture("tiny_model"))
    NC = tm["eval/logits"].shape[1]
    tasks = DictConfig({0: ListConfig([f"C{i}" for i in range(NC)])})
    dec = RangeDecoder(True, True, ListConfig([0, 15, 30]), ListConfig([15, 30, math.inf]), ListConfig([8, 2, 1]))
    post = DictConfig(num_pre_nms=50000, num_post_nms=1000, nms_threshold=0.3, min_confidence=0.1, nms_mode="WEIGHTED")
    mo = {1: {"cart": torch.as_tensor(tm["cart"]), "mask": torch.as_tensor(tm["mask"]),
              0: {"logits": torch.as_tensor(tm["eval/logits"]), "regressands": torch.as_tensor(tm["eval/regressands"])}}}
    p, s, c, b = dec.decode(mo, post, tasks, use_nms=True)
    out["b/tiny/params"], out["b/tiny/scores"], out["b/tiny/categories"], out["b/tiny/batch_index"] = p, s, c, b
    print("b tiny", tuple(p.shape), "from", int((torch.as_tensor(tm["eval/dec_scores"]) >= 0.1).sum()), "candidates")
    # and on the decode fixture's dense logits (7 classes, exact class ties, 20 % dropped pixels), dense + sampled decoders
    dg = np.load(_fixture("decode"))
    mo = {1: {"cart": torch.as_tensor(dg["cart"]), "mask": torch.as_tensor(dg["mask"]),
              0: {"logits": torch.as_tensor(dg["logits"]), "regressands": torch.as_tensor(dg["regressands"])}}}
    tasks7 = DictConfig({0: ListConfig(["c"] * dg["logits"].shape[1])})
    for tag, sample, postn in (("sampled", True, 1000), ("dense", False, 25)):
        d = RangeDecoder(True, sample, ListConfig([0, 15, 30]), ListConfig([15, 30, math.inf]), ListConfig([8, 2, 1]))
        cfg = DictConfig(num_pre_nms=50000, num_post_nms=postn, nms_threshold=0.3, min_confidence=0.1, nms_mode="WEIGHTED")
        p, s, c, b = d.decode(mo, cfg, tasks7, use_nms=True)
        out[f"b/{tag}/params"], out[f"b/{tag}/scores"], out[f"b/{tag}/categories"], out[f"b/{tag}/batch_index"] = p, s, c, b
        out[f"b/{tag}/num_post_nms"] = np.array(postn)
        print("b", tag, tuple(p.shape), "rows per class", [int((c == j).sum()) for j in range(7)])
    save("nms_wrapper", **out)


def gen_loader_train_item() -> None:
    """``DataLoader.__getitem__`` with ``split_name == "train"`` (prototype/loader.py:568-705): ROI filter -> augmentations on
    the UNPADDED table (:598-603, :514-549) -> table -> image -> ``range_view *= mask`` + W padding (``subsample_range_view``,
    :792-815) -> annotations joined with the task frame and sorted (:699-704).  Run by the reference itself on the synthetic
    tables of ``gen_loader_item``; the augmentation parameters are recovered by re-seeding ``random`` and drawing in the
    reference's order.  AV2: the shipped chain (conf/model/baseline.yaml:12-21) + a translation, circular padding; Waymo: a
    translation BEFORE the scale (the scale then recomputes ``range`` from translated coordinates: pixels without a return
    get range = |s t| > 0 and count as valid -- the reference's behaviour, kept), constant padding."""
    import random
    import tempfile
    import types
    from pathlib import Path

    import polars as pl  # stub
    import pyarrow as pa
    import pyarrow.feather as feather
    from torchbox3d.prototype import loader as ref_loader

    pl.scan_ipc = _ref_stubs._pl_scan_ipc
    H, W = 8, 64
    out = {}
    cats = ["BUS", "REGULAR_VEHICLE", "PEDESTRIAN"]
    for tag, ds, names, roi, mode, aug, seed in (
        ("av2", "av2", ["intensity", "range", "x", "y", "z"], True, "circular",
         {"flip_azimuth": {"p": 1.0}, "random_rotation": {"low": -0.78539816, "high": 0.78539816, "p": 1.0},
          "random_global_scale": {"low": 0.95, "high": 1.05}, "random_global_translation": {"std_x": 0.5, "std_y": 0.5, "std_z": 0.2}}, 11),
        ("waymo", "waymo", ["elongation", "intensity", "range", "x", "y", "z"], False, "constant",
         {"random_rotation": {"low": 2.0, "high": 3.0, "p": 1.0}, "random_global_translation": {"std_x": 0.5, "std_y": 0.5, "std_z": 0.2},
          "random_global_scale": {"low": 0.95, "high": 1.05}}, 12),
        # point_dropout (loader.py:506-512: EVERY column times a keep mask drawn from numpy's global generator) at the head of a chain
        ("av2_dropout", "av2", ["intensity", "range", "x", "y", "z"], True, "circular",
         {"point_dropout": {"p": 0.8}, "flip_azimuth": {"p": 1.0}, "random_global_scale": {"low": 0.95, "high": 1.05}}, 13)):
        rng = np.random.default_rng({"av2": 51, "waymo": 52}.get(tag, 53))
        inc = np.linspace(0.2, -0.4, H)[:, None]
        az = np.linspace(math.pi, -math.pi, W)[None, :]
        r = (20.0 + 15.0 * np.sin(3 * az) + 10.0 * np.cos(7 * inc) + rng.random((H, W))).astype(np.float32)
        keep = rng.random((H, W)) >= 0.1
        r = (r * keep).astype(np.float32)
        cols = {"x": (r * np.cos(inc) * np.cos(az)).astype(np.float32), "y": (r * np.cos(inc) * np.sin(az)).astype(np.float32),
                "z": (r * np.sin(inc) * np.ones_like(az)).astype(np.float32), "range": r,
                "intensity": ((rng.random((H, W)) * (255.0 if ds == "av2" else 3.0)) * keep).astype(np.float32),
                "elongation": (rng.random((H, W)) * keep).astype(np.float32)}
        if roi:
            cols["is_within_roi"] = rng.random((H, W)) >= 0.25
        tmp = Path(tempfile.mkdtemp())
        lidar = tmp / "sweep.feather"
        feather.write_feather(pa.table({k: v.reshape(-1) for k, v in cols.items()}), str(lidar), compression="uncompressed")
        M = 7
        yaw = rng.uniform(-math.pi, math.pi, M)
        ann = {"timestamp_ns": np.array([7, 7, 7, 8, 7, 7, 7], dtype=np.int64), "num_interior_pts": np.array([5, 3, 0, 9, 2, 4, 6], dtype=np.int64),
               "category": np.array(["REGULAR_VEHICLE", "BUS", "BUS", "BUS", "PEDESTRIAN", "BOLLARD", "BUS"]),
               "tx_m": rng.normal(0, 20, M), "ty_m": rng.normal(0, 20, M), "tz_m": rng.normal(0, 1, M), "length_m": rng.uniform(1, 6, M),
               "width_m": rng.uniform(1, 3, M), "height_m": rng.uniform(1, 3, M), "qw": np.cos(yaw / 2), "qx": np.zeros(M), "qy": np.zeros(M),
               "qz": np.sin(yaw / 2)}
        annp = tmp / "annotations.feather"
        feather.write_feather(pa.table(ann), str(annp), compression="uncompressed")
        tcfg = DictConfig({"tasks": DictConfig({0: ListConfig(["REGULAR_VEHICLE", "BUS"]), 1: ListConfig(["PEDESTRIAN"])})})
        me = types.SimpleNamespace(
            metadata=pl.DataFrame({"log_id": np.array(["log0"]), "timestamp_ns": np.array([7], dtype=np.int64)}),
            categories=cats, annotations_path=lambda log_id: annp, lidar_path=lambda log_id, ts: lidar,
            range_view_config=DictConfig({"feature_column_names": names, "filter_roi": roi, "height": H, "width": W}),
            split_name="train", augmentations_config=DictConfig({k: DictConfig(v) for k, v in aug.items()}), dataset_name=ds,
            enable_database=False, db_config=None, x_stride=1, padding_mode=mode, targets_config=tcfg)
        me.apply_augmentations = types.MethodType(ref_loader.DataLoader.apply_augmentations, me)
        me.tasks_frame = ref_loader.DataLoader.tasks_frame.func(me)
        me._point_dropout = types.MethodType(ref_loader.DataLoader._point_dropout, me)
        random.seed(seed)
        np.random.seed(seed)
        datum = ref_loader.DataLoader.__getitem__(me, 0)
        # the same draws, in the reference's order (flip: random(); rotation: random(), uniform; scale: uniform; translation: 3 x normalvariate)
        random.seed(seed)
        for k, v in aug.items():
            if k == "flip_azimuth":
                random.random()
            elif k == "random_rotation":
                random.random()
                out[f"{tag}/theta"] = np.float64(random.uniform(v["low"], v["high"]))
            elif k == "random_global_scale":
                out[f"{tag}/scale"] = np.float64(random.uniform(v["low"], v["high"]))
            elif k == "point_dropout":
                np.random.seed(seed)
                out[f"{tag}/keep"] = (np.random.rand(H * W, 1) <= v["p"]).reshape(-1)
            else:
                out[f"{tag}/t"] = np.array([random.normalvariate(0, v["std_x"]), random.normalvariate(0, v["std_y"]), random.normalvariate(0, v["std_z"])])
        for k, v in cols.items():
            out[f"{tag}/table/{k}"] = v.reshape(-1)
        for k, v in ann.items():
            out[f"{tag}/ann_in/{k}"] = v
        out[f"{tag}/feature_column_names"] = np.array(names)
        out[f"{tag}/padding_mode"], out[f"{tag}/filter_roi"] = np.array(mode), np.array(roi)
        out[f"{tag}/augmentation_order"] = np.array(list(aug))
        out[f"{tag}/seed"] = np.array(seed)
        out[f"{tag}/features"], out[f"{tag}/cart"], out[f"{tag}/mask"] = datum["features"].numpy(), datum["cart"].numpy(), datum["mask"].numpy()
        a = datum["annotations"]
        out[f"{tag}/ann_columns"] = np.array(a.columns)
        for k in a.columns:
            out[f"{tag}/ann_out/{k}"] = np.asarray(a[k])
        print(tag, "features", tuple(datum["features"].shape), "valid", float(datum["mask"].float().mean()), "annotations", a.shape, a.columns)
    save("loader_train_item", **out)


def gen_detections_frame() -> None:
    """``build_dataframe`` (math/ops/coding.py:31-76) run by the reference itself (over the polars stand-in: dtype casts of
    ``schema_overrides``, ``with_row_count`` + ``cast``, two inner joins, ``drop``) on the decoder rows of
    ``nms_wrapper.npz`` (b/tiny: float categories / batch index as ``decode(use_nms=True)`` returns them), a uuid frame that
    knows only sweep 0 (the rows of sweep 1 fall out of the inner join) and the loader's task frame."""
    import polars as pl  # stub
    from torchbox3d.math.ops.coding import build_dataframe

    nw = np.load(_fixture("nms_wrapper"))
    params, scores = torch.as_tensor(nw["b/tiny/params"]), torch.as_tensor(nw["b/tiny/scores"])
    cats, bidx = torch.as_tensor(nw["b/tiny/categories"]), torch.as_tensor(nw["b/tiny/batch_index"])
    names = ["REGULAR_VEHICLE", "PEDESTRIAN", "BUS", "BICYCLE", "TRUCK"]
    task_frame = pl.DataFrame({"task_id": np.zeros(5, dtype=np.int64), "offset": np.arange(5, dtype=np.int64), "category": np.array(names)})
    uuids = pl.DataFrame({"batch_index": np.array([0], dtype=np.int32), "log_id": np.array(["log-a"]),
                          "timestamp_ns": np.array([315969904359876000], dtype=np.uint64)})
    frame = build_dataframe(params, scores, cats, bidx, uuids, task_frame)
    out = {"columns": np.array(frame.columns), "dtypes": np.array([str(np.asarray(frame[c]).dtype) for c in frame.columns]),
           "category_names": np.array(names), "uuids/batch_index": np.array([0], dtype=np.int32), "uuids/log_id": np.array(["log-a"]),
           "uuids/timestamp_ns": np.array([315969904359876000], dtype=np.int64)}
    for c in frame.columns:
        out[f"col/{c}"] = np.asarray(frame[c])
    print("detections frame", frame.shape, frame.columns)
    save("detections_frame", **out)


ALL = ("conv_blocks", "meta_kernel", "range_partition", "decode", "projection", "tiny_model", "augment", "loader_item", "raw_sweep", "nms_wrapper",
       "loader_train_item", "detections_frame")  # in dependency order: nms_wrapper reads tiny_model / decode, detections_frame reads nms_wrapper


if __name__ == "__main__":
    torch.set_num_threads(8)
    names = sys.argv[1:] or ["conv_blocks", "meta_kernel", "decode", "projection", "tiny_model"]
    if names == ["all"]:
        names = list(ALL)
    for name in names:
        globals()["gen_" + name]()
