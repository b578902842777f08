"""The oracle (``oracle/``) against the golden vectors captured from the REFERENCE itself.

The fixtures in ``tests/golden/*.npz`` were produced by ``tests/golden/make_golden.py``
running the reference's own Python on CPU (fp32).  This pins the oracle for every row of
SURVEY.md §8a except the weighted-NMS kernel (parity unpinned, see ``oracle/nms.py``).
fp tolerance: 2e-5 relative-to-max for forward values, 2e-4 for gradients (fp32
accumulation-order noise between two CPU formulations); integer outputs exact.
"""

from __future__ import annotations

import math
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

from oracle import decode as odec
from oracle import model as om
from oracle import project as oproj
from oracle import targets as otgt


def close(a: torch.Tensor, b: torch.Tensor, tol: float = 2e-5, what: str = "") -> None:
    a, b = a.detach().double(), b.detach().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if b.numel() == 0:
        return
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f"{what}: rel-to-max error {err:.3e} > {tol}"


def run_case(g, prefix, fn, n_in, grad_tol=2e-4):
    """Train fwd/bwd + eval fwd of one module case against the reference's arrays."""
    sd = g.sub(f"{prefix}/sd")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    full = dict(sd)
    full.update(params)
    xs = [g[f"{prefix}/in{i}"].clone().requires_grad_(True) for i in range(n_in)]
    nm = om.Numerics(train=True, running={})
    y = fn(*xs, full, nm)
    close(y, g[f"{prefix}/out"], what=f"{prefix} out")
    (y * g[f"{prefix}/probe"]).sum().backward()
    for i, x in enumerate(xs):
        close(x.grad, g[f"{prefix}/gin{i}"], grad_tol, f"{prefix} gin{i}")
    for k, p in params.items():
        close(p.grad, g[f"{prefix}/grad/{k}"], grad_tol, f"{prefix} grad {k}")
    for k, v in g.sub(f"{prefix}/sd_after").items():
        close(nm.running[k], v, what=f"{prefix} running {k}")
    with torch.no_grad():
        ye = fn(*[x.detach() for x in xs], sd, om.Numerics(train=False))
    close(ye, g[f"{prefix}/out_eval"], what=f"{prefix} eval")


# ----------------------------------------------------------------------------- C1-C4
@pytest.mark.parametrize(
    "name,stride", [("conv3_s11", (1, 1)), ("conv3_s12", (1, 2)), ("conv1_s12", (1, 2)), ("conv1_s11", (1, 1))]
)
def test_conv2d_same(golden, name, stride):
    g = golden("conv_blocks")
    run_case(g, name, lambda x, sd, nm: om.conv2d_same(x, sd["conv.weight"], stride, nm=nm), 1)


def _pref(sd, p="m"):
    return {f"{p}.{k}": v for k, v in sd.items()}


def _wrap(fn):
    def inner(*args):
        *xs, sd, nm = args
        inner_nm = om.Numerics(train=nm.train, running={} if nm.running is not None else None)
        y = fn(*xs, _pref(sd), inner_nm)
        if nm.running is not None:
            nm.running.update({k[2:]: v for k, v in inner_nm.running.items()})
        return y

    return inner


def test_basic_blocks_running_stats(golden):
    g = golden("conv_blocks")
    run_case(g, "basic_plain", _wrap(lambda x, sd, nm: om.basic_block(x, sd, "m", nm=nm)), 1)
    run_case(g, "basic_proj_s12", _wrap(lambda x, sd, nm: om.basic_block(x, sd, "m", (1, 2), True, nm)), 1)
    run_case(g, "basic_k1_proj", _wrap(lambda x, sd, nm: om.basic_block(x, sd, "m", (1, 1), True, nm)), 1)


def test_residual_block(golden):
    g = golden("conv_blocks")
    run_case(g, "residual_s12_n3", _wrap(lambda x, sd, nm: om.residual_block(x, sd, "m", 3, (1, 2), nm)), 1)


def test_aggregation_blocks(golden):
    g = golden("conv_blocks")
    run_case(g, "agg_k8_s4", _wrap(lambda a, b, sd, nm: om.aggregation_block(a, b, sd, "m", (1, 4), (1, 2), 2, nm)), 2)
    run_case(g, "agg_k4_s2", _wrap(lambda a, b, sd, nm: om.aggregation_block(a, b, sd, "m", (1, 2), (1, 1), 1, nm)), 2)


# ----------------------------------------------------------------------------- C5
def test_meta_kernel(golden):
    g = golden("meta_kernel")
    run_case(g, "meta", _wrap(lambda f, c, sd, nm: om.meta_kernel(f, c, sd, "m", nm=nm)), 2, grad_tol=5e-4)


@pytest.mark.parametrize("tag", ["k1", "k3"])
def test_range_partition_stem(golden, tag):
    """RangePartition (the third stem RangeNet dispatches to, nn/stems/__init__.py:88-135) against the reference's own arrays: train
    forward, parameter gradients, running statistics, eval forward; the fixture holds returns exactly on the closed band edges."""
    g = golden("range_partition")
    feats, cart, mask = g["features"], g["cart"], g["mask"]
    d = cart.norm(dim=1)
    assert sum(int((d == e).sum()) for e in (10.0, 15.0, 20.0, 30.0, 40.0, 45.0, 60.0)) >= 7  # (the edge cases are in the data)
    sd = {"m." + k: v for k, v in g.sub(f"{tag}/sd").items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k and "bounds" not in k}
    full = dict(sd)
    full.update(params)
    nm = om.Numerics(train=True, running={})
    y = om.range_partition(feats, cart, mask, full, "m", nm)
    close(y, g[f"{tag}/out"], what=f"{tag} out")
    (y * g[f"{tag}/probe"]).sum().backward()
    grads = g.sub(f"{tag}/grad")
    assert set(grads) == {k[2:] for k in params}
    for k, v in grads.items():
        close(params["m." + k].grad, v, 2e-4, f"{tag} grad {k}")
    for k, v in g.sub(f"{tag}/sd_after").items():
        close(nm.running["m." + k], v, what=f"{tag} running {k}")
    with torch.no_grad():
        close(om.range_partition(feats, cart, mask, sd, "m", om.Numerics(train=False)), g[f"{tag}/out_eval"], what=f"{tag} eval")


def test_range_net_with_the_range_partition_stem(golden):
    g = golden("range_partition")
    sd = g.sub("net/sd")
    with torch.no_grad():
        out = om.range_net(g["features"], g["cart"], sd, stem_type="RANGE_PARTITION", nm=om.Numerics(train=False), mask=g["mask"])
    for k, v in g.sub("net/eval_feat").items():
        close(out[int(k)], v, what=f"RangeNet(RANGE_PARTITION) feature {k}")


# ----------------------------------------------------------------------------- D1-D3, Q1, L1, S1
def test_decode_range_view(golden):
    g = golden("decode")
    close(odec.decode_range_view(g["regressands"], g["cart"], True), g["decoded_inv"], 1e-6, "decode inv")
    close(odec.decode_range_view(g["regressands"], g["cart"], False), g["decoded_plain"], 1e-6, "decode plain")


def test_sample_by_range_and_decoder(golden):
    g = golden("decode")
    scores, cats = (g["logits"].sigmoid() * g["mask"]).max(dim=1, keepdim=True)
    s, c, b = odec.sample_by_range(scores, cats, g["decoded_inv"], g["cart"], (0, 15, 30), (15, 30, math.inf), (8, 2, 1))
    assert torch.equal(c, g["sampled_categories"])
    close(s, g["sampled_scores"], 1e-7, "sampled scores")
    close(b, g["sampled_cuboids"], 1e-7, "sampled cuboids")
    post = {"num_pre_nms": 50000, "num_post_nms": 1000, "nms_threshold": 0.3, "min_confidence": 0.1}
    p, sc, ca, bi = odec.range_decode(g["logits"], g["regressands"], g["cart"], g["mask"], post, use_nms=False)
    assert torch.equal(ca, g["dec_categories"]) and torch.equal(bi, g["dec_batch_index"])
    close(p, g["dec_params"], 1e-6, "dec params")
    close(sc, g["dec_scores"], 1e-7, "dec scores")
    p, sc, ca, bi = odec.range_decode(
        g["logits"], g["regressands"], g["cart"], g["mask"], post, use_nms=False, enable_sample_by_range=False
    )
    assert torch.equal(ca, g["dense_categories"]) and torch.equal(bi, g["dense_batch_index"])
    close(p, g["dense_params"], 1e-6, "dense params")


def test_yaw_to_quat_vfl_sph(golden):
    g = golden("decode")
    close(odec.yaw_to_quat(g["yaw"]), g["quat"], 1e-7, "quat")
    close(otgt.varifocal_loss(g["vfl_x"], g["vfl_t"]), g["vfl"], 1e-6, "vfl")
    sph = oproj.cart_to_sph(g.np("s1_cart").astype(np.float64))
    close(torch.from_numpy(sph), g["s1_sph"], 1e-6, "cart->sph")
    close(torch.from_numpy(oproj.sph_to_cart(sph)), g["s1_back"], 1e-5, "sph->cart")


# ----------------------------------------------------------------------------- R1, R2
def test_projection_bit_exact(golden):
    g = golden("projection")
    cart = g.np("cart")
    sph = oproj.cart_to_sph(cart)
    assert np.array_equal(sph, g.np("sph"))
    H, W = g.np("image_converter").shape[1:]
    for variant in ("converter", "library"):
        rows, cols, radius = oproj.range_view_indices(sph, g.np("laser_numbers"), g.np("row_mapping_64"), H, W, variant)
        ref_idx = g.np(f"indices_{variant}")
        assert np.array_equal(rows, ref_idx[0]) and np.array_equal(cols, ref_idx[1]), variant
        assert np.array_equal(radius, g.np(f"hybrid_{variant}")[:, 2])
        image, winner = oproj.z_buffer(rows, cols, radius, g.np("features"), H, W)
        assert np.array_equal(image, g.np(f"image_{variant}")), variant
        # winner map is consistent with the image
        filled = winner >= 0
        assert np.array_equal(image[2][filled], g.np("features")[2][winner[filled]].astype(np.float32))
    # the two binning variants differ by exactly one column almost everywhere
    d = g.np("indices_converter")[1] - g.np("indices_library")[1]
    assert (d == 1).mean() > 0.95  # (200 of 6000 points sit on forced half-bin ties)
    close(torch.from_numpy(oproj.sph_to_cart(g.np("sph"))), g["np_sph_to_cart"], 1e-12, "np sph->cart")


def test_correctly_rounded_azimuth_gives_the_reference_columns(golden):
    """The device bins with THE correctly rounded atan2 (witness ``oproj.atan2_cr``); numpy's SIMD arctan2 differs from it
    in the last bit on ~1 % of the fixture points, none of which changes a column -- the 112 exact half-bin ties included."""
    g = golden("projection")
    cart, sph = g.np("cart"), g.np("sph").copy()
    az = oproj.atan2_cr(cart[:, 1], cart[:, 0])
    assert np.max(np.abs(az - sph[:, 0])) <= np.spacing(np.pi)  # within one ulp of the reference's azimuth
    sph[:, 0] = az
    H, W = g.np("image_converter").shape[1:]
    t = (az + np.pi) * (W / math.tau)
    assert int((t - np.floor(t) == 0.5).sum()) >= 100  # the fixture really contains exact ties
    for variant in ("converter", "library"):
        _, cols, _ = oproj.range_view_indices(sph, g.np("laser_numbers"), g.np("row_mapping_64"), H, W, variant)
        assert np.array_equal(cols, g.np(f"indices_{variant}")[1]), variant
    # the decimal fallback alone (no long double) agrees with the fast path
    for i in (0, 17, 2000, 2100, 5999):
        assert oproj._atan2_exact_rn(float(cart[i, 1]), float(cart[i, 0])) == az[i]


def test_w_padding_rule(golden):
    g = golden("projection")
    for ds, w_out in (("av2", 1808), ("waymo", 2656)):
        for mode in ("constant", "circular"):
            rv = g.np(f"pad/{ds}/{mode}/rv_in") * g.np(f"pad/{ds}/{mode}/mask_in")
            out = oproj.pad_range_view(rv, ds, mode)
            assert out.shape[-1] == w_out and w_out % 16 == 0
            assert np.array_equal(out, g.np(f"pad/{ds}/{mode}/rv"))


# ----------------------------------------------------------------------------- full tiny model: C6-C9, T1, T2, L2
def _tiny_state(g):
    sd = g.sub("sd")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    full = dict(sd)
    full.update(params)
    return sd, params, full


def test_tiny_model_forward_backward(golden):
    g = golden("tiny_model")
    sd, params, full = _tiny_state(g)
    nm = om.Numerics(train=True, running={})
    feats, logits, reg = om.detector_forward(g["features"], g["cart"], full, nm=nm)
    for s in (1, 2, 4, 16):
        close(feats[s], g[f"feat/{s}"], 5e-5, f"feat {s}")
    close(logits, g["logits"], 5e-5, "logits")
    close(reg, g["regressands"], 5e-5, "regressands")

    tg = otgt.compute_targets(g["cart"], g["annotations"], num_classes=5)
    for k in ("classification_labels", "panoptics", "points_per_obj"):
        assert torch.equal(tg[k], g[f"targets/{k}"]), k
    close(tg["regression_targets"], g["targets/regression_targets"], 1e-6, "regression targets")
    assert int((tg["panoptics"] > 0).sum()) > 20  # the scene really has foreground

    losses = otgt.detection_loss(logits, reg, g["cart"], g["mask"], tg, num_classes=5)
    close(losses["targets"], g["targets/soft"], 1e-5, "soft targets")
    assert torch.equal(losses["foreground"], g["aux/foreground"])
    assert torch.equal(losses["background"].float(), g["aux/background"])
    for k in ("loss", "classification_loss", "foreground_loss", "background_loss", "regression_loss",
              "coordinate_loss", "dimension_loss", "rotation_loss", "total_fg", "total_objects"):
        close(losses[k].reshape(()), g[f"loss/{k}"].reshape(()), 5e-5, f"loss {k}")
    assert losses["loss"].dtype == torch.float64  # detection_head.py:349-353

    losses["loss"].backward()
    worst = 0.0
    for k, p in params.items():
        ref = g[f"grad/{k}"]
        scale = max(float(ref.abs().max()), 1e-7)
        worst = max(worst, float((p.grad.double() - ref.double()).abs().max()) / scale)
    assert worst < 2e-3, worst
    for k, v in g.sub("sd_after").items():
        close(nm.running[k], v, 5e-5, f"running {k}")


def test_tiny_model_eval_and_decode(golden):
    g = golden("tiny_model")
    sd = g.sub("sd")
    with torch.no_grad():
        feats, logits, reg = om.detector_forward(g["features"], g["cart"], sd, nm=om.Numerics(train=False))
    close(feats[1], g["eval/feat1"], 5e-5, "eval feat")
    close(logits, g["eval/logits"], 5e-5, "eval logits")
    close(reg, g["eval/regressands"], 5e-5, "eval regressands")
    post = {"num_pre_nms": 50000, "num_post_nms": 1000, "nms_threshold": 0.3, "min_confidence": 0.1}
    p, s, c, b = odec.range_decode(g["eval/logits"], g["eval/regressands"], g["cart"], g["mask"], post, use_nms=False)
    assert torch.equal(c, g["eval/dec_categories"]) and torch.equal(b, g["eval/dec_batch_index"])
    assert p.shape[0] > 10  # the fixture really has detections above min_confidence
    close(p, g["eval/dec_params"], 1e-6, "dec params")
    close(s, g["eval/dec_scores"], 1e-7, "dec scores")


def _check_augmented(got_sweep, got_ann, g, tag, names):
    from oracle import augment as oaug

    ref_s, ref_a = g.np(f"{tag}/sweep"), g.np(f"{tag}/ann")
    # index work (which pixel lands where) is exact: every non-geometry channel must match bit for bit
    for i, n in enumerate(names):
        if n not in ("x", "y", "z", "range"):
            assert np.array_equal(got_sweep[i].astype(np.float32), ref_s[i].astype(np.float32)), (tag, n)
    # geometry: 1e-6 of the channel maximum (the reference mixes float32 columns with float64 matrix products)
    for i, n in enumerate(names):
        assert np.max(np.abs(got_sweep[i] - ref_s[i])) <= 1e-6 * max(1.0, np.max(np.abs(ref_s[i]))), (tag, n)
    assert np.max(np.abs(got_ann[:6] - ref_a[:6])) <= 1e-9 * max(1.0, np.max(np.abs(ref_a[:6]))), tag
    dyaw = oaug.yaw_of(got_ann[6:10]) - oaug.yaw_of(ref_a[6:10])  # (q and -q are the same rotation)
    assert np.max(np.abs(np.arctan2(np.sin(dyaw), np.cos(dyaw)))) < 1e-9, tag


def test_augmentations_match_the_reference(golden):
    from oracle import augment as oaug

    g = golden("augment")
    names = [str(n) for n in g.np("column_names")]
    s0, a0 = g.np("sweep/in"), g.np("ann/in")
    _check_augmented(*oaug.flip(s0, names, a0), g, "flip", names)
    _check_augmented(*oaug.rotate(s0, names, a0, float(g.np("rotation/theta"))), g, "rotation", names)
    _check_augmented(*oaug.rotate(s0, names, a0, float(g.np("rotation_neg/theta"))), g, "rotation_neg", names)
    _check_augmented(*oaug.scale(s0, names, a0, float(g.np("scale/scale"))), g, "scale", names)
    _check_augmented(*oaug.translate(s0, names, a0, g.np("translation/t")), g, "translation", names)
    s, a = oaug.flip(s0, names, a0)
    s, a = oaug.rotate(s, names, a, float(g.np("chain/theta")))
    s, a = oaug.scale(s, names, a, float(g.np("chain/scale")))
    s, a = oaug.translate(s, names, a, g.np("chain/t"))
    _check_augmented(s, a, g, "chain", names)
    # the roll really moves columns: shift = floor(theta / tau * W), negative theta rolls left
    W = s0.shape[-1]
    sh = math.floor(float(g.np("rotation_neg/theta")) / math.tau * W)
    assert sh < 0 and np.array_equal(g.np("rotation_neg/sweep")[3], np.roll(s0[3], sh, axis=-1))


def _loader_case(g, tag):
    names = [str(n) for n in g.np(f"{tag}/feature_column_names")]
    table = {k: g.np(f"{tag}/table/{k}") for k in ("intensity", "range", "x", "y", "z", "elongation", "is_within_roi", "laser_number") if f"{tag}/table/{k}" in g}
    return names, table, bool(g.np(f"{tag}/filter_roi")), str(g.np(f"{tag}/padding_mode"))


def test_loader_item_matches_the_reference(golden):
    """Table -> (features, cart, mask) of DataLoader.__getitem__ (loader.py:568-705) against the reference's own output: AV2
    columns with the ROI filter and circular padding, Waymo columns with tanh(intensity) and constant padding.  Copies,
    products with 0/1 and the mask are exact; tanh is numpy's on both sides here."""
    from oracle import loader as old

    g = golden("loader_item")
    row_map = golden("raw_sweep").np("tables/ROW_MAPPING_64")  # (the dataset's id table, stored as data by the raw-sweep fixture)
    for tag, ds in (("av2", "av2"), ("waymo", "waymo"), ("av2_view", "av2")):
        names, table, roi, mode = _loader_case(g, tag)
        got = old.range_view_from_table(table, names, 8, 64, ds, roi, 1, mode, row_mapping_64=row_map)
        for k in ("features", "cart", "mask"):
            ref = g.np(f"{tag}/{k}")
            assert got[k].shape == ref.shape and got[k].dtype == ref.dtype, (tag, k, got[k].shape, ref.shape, got[k].dtype, ref.dtype)
            assert np.array_equal(got[k], ref), (tag, k)
        assert got["features"].shape[-1] == {"av2": 72, "waymo": 70}[ds]


def test_raw_sweep_path_matches_the_reference(golden):
    from oracle import rawsweep as oraw

    g = golden("raw_sweep")
    kept, xyz_p = oraw.unmotion_compensate(g.np("sweep/xyz"), g.np("sweep/offset_ns"), int(g.np("sweep/timestamp_ns")), g.np("poses/timestamp_ns"),
                                           g.np("poses/q_wxyz"), g.np("poses/t"))
    assert np.array_equal(kept, g.np("unmotion/kept")) and 0 < kept.sum() < kept.size
    ref = g.np("unmotion/xyz_p")
    assert np.max(np.abs(xyz_p - ref)) < 1e-9 * np.max(np.abs(ref))  # fp64 rigid transforms (scipy Slerp vs quaternion slerp)
    lz = g.np("laser/in")
    for tag, affected, rows in (("h64_affected", True, "ROW_MAPPING_64"), ("h64_plain", False, "ROW_MAPPING_64")):
        assert np.array_equal(oraw.correct_laser_numbers(lz, affected, g.np("tables/LASER_MAPPING"), g.np(f"tables/{rows}")), g.np(f"laser/{tag}")), tag
    assert np.array_equal(oraw.correct_laser_numbers(lz % 32, True, g.np("tables/LASER_MAPPING"), g.np("tables/ROW_MAPPING_32")), g.np("laser/h32_affected"))
    feats = np.stack([g.np("sweep/xyz")[kept][:, 0], g.np("sweep/xyz")[kept][:, 1], g.np("sweep/xyz")[kept][:, 2], g.np("sweep/intensity")[kept].astype(np.float64),
                      g.np("laser/h64_affected").astype(np.float64), g.np("sweep/is_within_roi")[kept].astype(np.float64)], axis=1)
    img = oraw.build_range_view(ref, feats, g.np("laser/h64_affected"), g.np("sweep/offset_ns")[kept], g.np("extrinsics/q_wxyz"), g.np("extrinsics/t"), 64, 512)
    want = g.np("range_view/image")
    # the reference casts intensity / laser_number to UInt8 and is_within_roi to Boolean when it builds the table
    got = img.astype(np.float64)
    got[3], got[4], got[5] = got[3].astype(np.uint8), got[4].astype(np.uint8), got[5] != 0
    assert np.array_equal(got, want), int((got != want).sum())


# ----------------------------------------------------------------------------- N1 (wrapper logic pinned; N2 stays declared)
def test_nms_wrapper_matches_the_reference_wrapper(golden):
    """``oracle.nms`` / ``oracle.decode.range_decode(use_nms=True)`` against what the REFERENCE's own
    ``batched_multiclass_nms`` / ``weighted_multiclass_nms`` / ``weighted_nms`` / ``RangeDecoder.decode(use_nms=True)`` returned
    (``math/ops/nms.py:64-123,126-177,181-266``, ``range_decoder.py:100-124``) when run in the build container over a
    ``wnms_gpu`` stand-in with the declared kernel semantics (``tests/golden/make_golden.py::gen_nms_wrapper``).  Pins the
    wrapper logic: class order, both top-k cuts, merged-score ranking, float categories / batch index, empty shapes.  The
    kernel arithmetic (N2) remains declared -- the same C function sits under both sides here."""
    from oracle import nms as onms

    g = golden("nms_wrapper")
    cub, sc, cat = g["a/cuboids"], g["a/scores"], g["a/categories"]
    for tag in ("post1000", "post40", "pre150"):
        pre, post, thr, conf = g.np(f"a/{tag}/cfg").tolist()
        p, s, c, b = onms.batched_multiclass_nms(cub, sc, cat, int(pre), int(post), thr, conf)
        assert torch.equal(c, g[f"a/{tag}/categories"]) and torch.equal(b, g[f"a/{tag}/batch_index"]), tag
        assert c.dtype == torch.float32 and b.dtype == torch.float32
        assert torch.equal(p, g[f"a/{tag}/params"]) and torch.equal(s, g[f"a/{tag}/scores"]), tag
    m = sc[0] >= 0.1
    p, s, c = onms.weighted_multiclass_nms(cub[0, m], sc[0, m], cat[0, m], 0.3, 50000, 40)
    assert torch.equal(p, g["a/multiclass/params"]) and torch.equal(s, g["a/multiclass/scores"]) and torch.equal(c, g["a/multiclass/categories"])
    keep, merged, count = onms.weighted_nms(g["a/wnms/boxes"], g["a/wnms/data"], g["a/wnms/scores"], 0.3, 0.5)
    assert torch.equal(keep, g["a/wnms/keep"]) and torch.equal(merged, g["a/wnms/output"]) and torch.equal(count, g["a/wnms/count"])
    p, s, c, b = onms.batched_multiclass_nms(cub[1:2], sc[1:2], cat[1:2], 50000, 1000, 0.3, 0.1)
    assert list(p.shape) == g.np("a/empty/params_shape").tolist() and list(s.shape) == g.np("a/empty/scores_shape").tolist()
    assert list(c.shape) == g.np("a/empty/categories_shape").tolist() and list(b.shape) == g.np("a/empty/batch_index_shape").tolist()
    assert (c.dtype == torch.int64) == bool(g.np("a/empty/categories_is_int64"))
    # RangeDecoder.decode(use_nms=True)
    t = golden("tiny_model")
    post = {"num_pre_nms": 50000, "num_post_nms": 1000, "nms_threshold": 0.3, "min_confidence": 0.1}
    p, s, c, b = odec.range_decode(t["eval/logits"], t["eval/regressands"], t["cart"], t["mask"], post, use_nms=True)
    assert torch.equal(c, g["b/tiny/categories"]) and torch.equal(b, g["b/tiny/batch_index"])
    close(p, g["b/tiny/params"], 1e-6, "tiny params")
    close(s, g["b/tiny/scores"], 1e-7, "tiny scores")
    d = golden("decode")
    for tag, sample in (("sampled", True), ("dense", False)):
        post["num_post_nms"] = int(g.np(f"b/{tag}/num_post_nms"))
        p, s, c, b = odec.range_decode(d["logits"], d["regressands"], d["cart"], d["mask"], post, use_nms=True, enable_sample_by_range=sample)
        assert torch.equal(c, g[f"b/{tag}/categories"]) and torch.equal(b, g[f"b/{tag}/batch_index"]), tag
        close(p, g[f"b/{tag}/params"], 1e-6, f"{tag} params")
        close(s, g[f"b/{tag}/scores"], 1e-7, f"{tag} scores")


def _train_item_case(g, tag):
    names = [str(n) for n in g.np(f"{tag}/feature_column_names")]
    table = {k[len(f"{tag}/table/"):]: g.np(k) for k in g.keys if k.startswith(f"{tag}/table/")}
    ann_in = {k[len(f"{tag}/ann_in/"):]: g.np(k) for k in g.keys if k.startswith(f"{tag}/ann_in/")}
    chain = []
    for k in [str(n) for n in g.np(f"{tag}/augmentation_order")]:
        if k == "point_dropout":
            chain.append(("dropout", g.np(f"{tag}/keep")))
            continue
        chain.append({"flip_azimuth": ("flip",), "random_rotation": ("rotate", float(g.np(f"{tag}/theta")) if f"{tag}/theta" in g else 0.0),
                      "random_global_scale": ("scale", float(g.np(f"{tag}/scale")) if f"{tag}/scale" in g else 1.0),
                      "random_global_translation": ("translate", tuple(g.np(f"{tag}/t").tolist()) if f"{tag}/t" in g else (0, 0, 0))}[k])
    return names, table, ann_in, chain, bool(g.np(f"{tag}/filter_roi")), str(g.np(f"{tag}/padding_mode"))


TRAIN_TASKS = {0: ["REGULAR_VEHICLE", "BUS"], 1: ["PEDESTRIAN"]}


def test_loader_train_item_augments_before_padding(golden):
    """The train-split ``__getitem__`` of the reference (augmentations on the unpadded table, THEN features *= mask and the W
    padding; annotations filtered, joined with the task frame and sorted) against ``oracle.loader.train_item_from_table``:
    placement channels exact, geometry 1e-6, mask exact -- including the Waymo chain whose scale follows a translation (empty
    pixels become valid there, as in the reference)."""
    from oracle import augment as oaug
    from oracle import loader as old

    g = golden("loader_train_item")
    for tag, ds in (("av2", "av2"), ("waymo", "waymo"), ("av2_dropout", "av2")):
        names, table, ann_in, chain, roi, mode = _train_item_case(g, tag)
        got = old.train_item_from_table(table, names, 8, 64, ds, roi, chain, 1, mode)
        assert np.array_equal(got["mask"], g.np(f"{tag}/mask")), tag
        ref_f, ref_c = g.np(f"{tag}/features"), g.np(f"{tag}/cart")
        assert got["features"].shape == ref_f.shape and got["features"].shape[-1] == 64 + (8 if ds == "av2" else 6)
        for i, n in enumerate(names):
            tol = 1e-6 * max(1.0, float(np.abs(ref_f[i]).max()))
            if n in ("x", "y", "z", "range") or (n == "intensity" and ds == "waymo"):
                assert np.max(np.abs(got["features"][i] - ref_f[i])) <= tol, (tag, n)
            else:
                assert np.array_equal(got["features"][i].astype(np.float32), ref_f[i]), (tag, n)
        assert np.max(np.abs(got["cart"] - ref_c)) <= 1e-6 * max(1.0, float(np.abs(ref_c).max())), tag
        # annotations: filter + join + sort, then the same chain
        rows, keys = old.annotations_for_sweep(ann_in, 7, TRAIN_TASKS)
        assert [k[0] for k in keys] == g.np(f"{tag}/ann_out/task_id").tolist() and [k[1] for k in keys] == g.np(f"{tag}/ann_out/offset").tolist()
        assert [str(ann_in["category"][i]) for i in rows] == [str(c) for c in g.np(f"{tag}/ann_out/category")]
        cols = ("tx_m", "ty_m", "tz_m", "length_m", "width_m", "height_m", "qw", "qx", "qy", "qz")
        a = np.stack([np.asarray(ann_in[c], dtype=np.float64)[rows] for c in cols])
        s = np.zeros((6, 1, 1))
        for op in [o for o in chain if o[0] != "dropout"]:
            _, a = {"flip": lambda: oaug.flip(s, ["x", "y", "z", "range", "i", "e"], a), "rotate": lambda: oaug.rotate(s, ["x", "y", "z", "range", "i", "e"], a, op[1]),
                    "scale": lambda: oaug.scale(s, ["x", "y", "z", "range", "i", "e"], a, op[1]),
                    "translate": lambda: oaug.translate(s, ["x", "y", "z", "range", "i", "e"], a, op[1])}[op[0]]()
        ref_a = np.stack([g.np(f"{tag}/ann_out/{c}") for c in cols])
        assert np.max(np.abs(a[:6] - ref_a[:6])) <= 1e-9 * max(1.0, np.max(np.abs(ref_a[:6]))), tag
        dyaw = oaug.yaw_of(a[6:10]) - oaug.yaw_of(ref_a[6:10])
        assert np.max(np.abs(np.arctan2(np.sin(dyaw), np.cos(dyaw)))) < 1e-9, tag


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/torchbox3d"), reason="build container only: regenerating the fixtures runs the reference")
def test_fixture_generator_reproduces_every_committed_fixture(tmp_path):
    """tests/golden/make_golden.py (the reference itself, run in the build container) regenerates all twelve fixtures BYTE for byte:
    every generator seeds its own torch.Generator and the global one (module constructors initialise from it)."""
    import subprocess
    import sys

    env = dict(os.environ, RV3D_GOLDEN_OUT=str(tmp_path), PYTORCH_JIT="0")
    out = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), "all"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    names = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert len(names) == 12
    for f in names:
        assert open(os.path.join(GOLDEN, f), "rb").read() == open(tmp_path / f, "rb").read(), f"{f} is not reproduced"
