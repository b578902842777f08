"""N1 -- both HIP weighted-NMS paths against fixtures made by the REFERENCE's own wrapper code.

``tests/golden/nms_wrapper.npz`` (``tests/golden/make_golden.py::gen_nms_wrapper``) holds what the reference's
``batched_multiclass_nms`` / ``weighted_multiclass_nms`` / ``weighted_nms`` (``math/ops/nms.py:64-123,126-177,181-266``) and
``RangeDecoder.decode(use_nms=True)`` (``nn/decoders/range_decoder.py:100-124``) returned on CPU over a ``wnms_gpu``
stand-in with the declared kernel semantics.  Compared here: the device-resident batch path (``rv_nms_sweeps``) AND the
reference-shaped per-class loop over the FFI (``rv_wnms``).  Bar: row order, categories, batch index, dtypes and shapes
exact; boxes / scores 1e-6 of max (device ``sin`` / ``cos`` / ``atan2`` against the CPU's).  The inner kernel's arithmetic
(N2) stays declared semantics: TorchEx's source is absent.
"""

from __future__ import annotations

import math

import pytest
import torch

from test_gpu_forward import DEV, rel_err

pytestmark = pytest.mark.gpu


def _both_paths(fn):
    """Run ``fn()`` on the device-resident path and with the per-class loop forced."""
    from range_view_3d_detection_amd.math.ops import nms as hnms

    fast = fn()
    old = hnms.FUSED_CLASSES_MAX
    hnms.FUSED_CLASSES_MAX = 0
    try:
        loop = fn()
    finally:
        hnms.FUSED_CLASSES_MAX = old
    return {"rv_nms_sweeps": fast, "per-class loop": loop}


def _canonical(p, s, c, b):
    """Rows with EXACTLY equal (sweep, class, merged score) come out of ``scores.topk`` (``nms.py:116``) in an order torch does
    not define (its CPU and device kernels differ); the device path breaks such ties by candidate index.  Everything the
    reference defines -- sweep order, class order, descending score -- is compared in place; inside a tie group the rows
    are compared as a set (ordered by their box centre here)."""
    import numpy as np

    p, s, c, b = (t.detach().cpu() for t in (p, s, c, b))
    if s.dim() != 1 or s.numel() == 0:
        return p, s, c, b
    same = (s[1:] == s[:-1]) & (c[1:] == c[:-1]) & (b[1:] == b[:-1])
    group = torch.cat([torch.zeros(1, dtype=torch.long), (~same).long().cumsum(0)])
    order = torch.from_numpy(np.lexsort((p[:, 1].numpy(), p[:, 0].numpy(), group.numpy())))
    return p[order], s, c, b


def _same_rows(got, ref, what):
    p, s, c, b = _canonical(*got)
    rp, rs, rc, rb = _canonical(*ref)
    assert p.shape == rp.shape and s.shape == rs.shape, (what, tuple(p.shape), tuple(rp.shape))
    assert c.dtype == rc.dtype and b.dtype == rb.dtype, (what, c.dtype, b.dtype)
    assert torch.equal(c.cpu(), rc), f"{what}: categories / row order differ"
    assert torch.equal(b.cpu(), rb), f"{what}: batch index differs"
    assert rel_err(p, rp) < 1e-6, (what, rel_err(p, rp))
    assert rel_err(s, rs) < 1e-6, (what, rel_err(s, rs))


@pytest.mark.parametrize("tag", ["post1000", "post40", "pre150"])
def test_batched_multiclass_nms_against_the_reference_wrapper(golden, tag):
    """3 sweeps x 2000 candidates x 5 classes: an absent class, a class holding half of the candidates, a sweep with nothing
    >= min_confidence, exact score ties (inside clusters and between far-apart boxes); ``num_post_nms`` 1000 / 40 (the
    merged-score top-k cuts) and ``num_pre_nms`` 150 (the pre-NMS top-k cuts: the per-class loop on both sides)."""
    from range_view_3d_detection_amd.math.ops import nms as hnms

    g = golden("nms_wrapper")
    pre, post, thr, conf = g.np(f"a/{tag}/cfg").tolist()
    cub, sc, cat = g["a/cuboids"].to(DEV), g["a/scores"].to(DEV), g["a/categories"].to(DEV)
    ref = tuple(g[f"a/{tag}/{k}"] for k in ("params", "scores", "categories", "batch_index"))
    for name, got in _both_paths(lambda: hnms.batched_multiclass_nms(cub, sc, cat, int(pre), int(post), thr, conf, "weighted", n_classes=5)).items():
        _same_rows(got, ref, f"{tag} / {name}")


def test_weighted_multiclass_and_weighted_nms_against_the_reference_wrapper(golden):
    from range_view_3d_detection_amd.math.ops import nms as hnms

    g = golden("nms_wrapper")
    cub, sc, cat = g["a/cuboids"].to(DEV), g["a/scores"].to(DEV), g["a/categories"].to(DEV)
    m = sc[0] >= 0.1
    ref = tuple(g[f"a/multiclass/{k}"] for k in ("params", "scores", "categories"))
    for name, (p, s, c) in _both_paths(lambda: hnms.weighted_multiclass_nms(cub[0, m], sc[0, m], cat[0, m], 0.3, 50000, 40)).items():
        assert torch.equal(c.cpu(), ref[2]) and c.dtype == ref[2].dtype, name
        assert p.shape == ref[0].shape and rel_err(p, ref[0]) < 1e-6 and rel_err(s, ref[1]) < 1e-6, name
    # the wrapper around the FFI itself (nms.py:126-177): same inputs the reference's wrapper was given
    keep, merged, count = hnms.weighted_nms(g["a/wnms/boxes"].to(DEV), g["a/wnms/data"].to(DEV), g["a/wnms/scores"].to(DEV), 0.3, 0.5)
    assert torch.equal(keep.cpu(), g["a/wnms/keep"]) and torch.equal(count.cpu(), g["a/wnms/count"])
    assert torch.equal(merged.cpu(), g["a/wnms/output"])  # same fp32 inputs, unfused arithmetic: bit for bit


def test_empty_batch_shapes_against_the_reference_wrapper(golden):
    from range_view_3d_detection_amd.math.ops import nms as hnms

    g = golden("nms_wrapper")
    cub, sc, cat = g["a/cuboids"][1:2].to(DEV), g["a/scores"][1:2].to(DEV), g["a/categories"][1:2].to(DEV)
    for name, (p, s, c, b) in _both_paths(lambda: hnms.batched_multiclass_nms(cub, sc, cat, 50000, 1000, 0.3, 0.1, "WEIGHTED", n_classes=5)).items():
        assert list(p.shape) == g.np("a/empty/params_shape").tolist() and list(s.shape) == g.np("a/empty/scores_shape").tolist(), name
        assert list(c.shape) == g.np("a/empty/categories_shape").tolist() and list(b.shape) == g.np("a/empty/batch_index_shape").tolist(), name
        assert (c.dtype == torch.int64) == bool(g.np("a/empty/categories_is_int64")), name


@pytest.mark.parametrize("tag,sample", [("tiny", True), ("sampled", True), ("dense", False)])
def test_range_decoder_with_nms_against_the_reference(golden, tag, sample):
    """``RangeDecoder.decode(use_nms=True)`` from the reference's fp32 logits / regressands: the tiny model's eval outputs, and
    the decode fixture (7 classes, exact class ties, dropped pixels) with the band-sampled and the dense decoder."""
    from range_view_3d_detection_amd.nn.decoders.range_decoder import RangeDecoder

    g = golden("nms_wrapper")
    if tag == "tiny":
        t = golden("tiny_model")
        logits, reg, cart, mask, postn = t["eval/logits"], t["eval/regressands"], t["cart"], t["mask"], 1000
    else:
        t = golden("decode")
        logits, reg, cart, mask, postn = t["logits"], t["regressands"], t["cart"], t["mask"], int(g.np(f"b/{tag}/num_post_nms"))
    mo = {1: {"cart": cart.to(DEV), "mask": mask.to(DEV), 0: {"logits": logits.to(DEV), "regressands": reg.to(DEV)}}}
    dec = RangeDecoder(True, sample, [0, 15, 30], [15, 30, math.inf], [8, 2, 1])
    post = {"num_pre_nms": 50000, "num_post_nms": postn, "nms_threshold": 0.3, "min_confidence": 0.1, "nms_mode": "WEIGHTED"}
    ref = tuple(g[f"b/{tag}/{k}"] for k in ("params", "scores", "categories", "batch_index"))
    assert ref[0].shape[0] > 50
    for name, got in _both_paths(lambda: dec.decode(mo, post, {0: ["c"] * logits.shape[1]}, use_nms=True)).items():
        _same_rows(got, ref, f"{tag} / {name}")


@pytest.mark.parametrize("n_cls,k,num_pre", [(26, 60000, 50000), (3, 60000, 50000), (3, 30000, 4000)])
def test_large_sweeps_stay_on_the_device_path(n_cls, k, num_pre):
    """Sweeps far beyond the old 16 384-candidate capacity (the decoder emits 212 992 candidates per sweep and early-training
    logits put most of them over ``min_confidence``): 4 sweeps x 60 000 candidates never leave ``rv_nms_sweeps`` -- per-candidate
    arrays for every candidate, class-relative pair masks, the 3-class case exceeds the default mask budget and is RESUMED over
    a buffer of the reported size (ordering stages not repeated); ``num_pre_nms`` = 4000 below the class sizes exercises the
    pre-NMS top-k cut on device.  Rows must equal the reference-shaped per-class loop over the FFI bit for bit."""
    from range_view_3d_detection_amd.math.ops import nms as hnms
    from test_gpu_model import _random_boxes

    cubs, scs, cats = [], [], []
    for b in range(4):
        cub, s = _random_boxes(k, 900 + b, 600.0)
        cubs.append(cub)
        scs.append(s)
        cats.append(torch.randint(0, n_cls, (k,), generator=torch.Generator().manual_seed(40 + b)))
    args = (torch.stack(cubs).to(DEV), torch.stack(scs).to(DEV), torch.stack(cats).to(DEV), num_pre, 300, 0.3, 0.1, "weighted")
    calls = []
    orig = hnms.nms_sweeps

    def spy(*a, **kw):
        out = orig(*a, **kw)
        calls.append(out[3])
        return out

    hnms.nms_sweeps = spy
    try:
        fast = hnms.batched_multiclass_nms(*args, n_classes=n_cls)
    finally:
        hnms.nms_sweeps = orig
    assert len(calls) == 1 and all(c >= 0 for c in calls[0]), calls  # every sweep was finished on device
    old = hnms.FUSED_CLASSES_MAX
    hnms.FUSED_CLASSES_MAX = 0
    try:
        loop = hnms.batched_multiclass_nms(*args, n_classes=n_cls)
    finally:
        hnms.FUSED_CLASSES_MAX = old
    assert fast[0].shape == loop[0].shape and fast[0].shape[0] == 4 * n_cls * 300
    for a, b_ in zip(fast, loop):
        assert torch.equal(a, b_)


def test_mask_budget_resume_equals_one_pass():
    """A mask word budget too small for the batch: the kernels report the words each sweep needs, the host resumes the mask
    stages over a buffer of that size -- same rows as a run whose budget sufficed."""
    from range_view_3d_detection_amd.math.ops import nms as hnms
    from test_gpu_model import _random_boxes

    cub, s = _random_boxes(5000, 77, 80.0)
    cat = torch.randint(0, 4, (5000,), generator=torch.Generator().manual_seed(3))
    args = (cub[None].to(DEV), s[None].to(DEV), cat[None].to(DEV), 50000, 100, 0.3, 0.1, "weighted")
    ref = hnms.batched_multiclass_nms(*args, n_classes=4)
    old = hnms.MASK_WORDS
    hnms.MASK_WORDS = 1000
    try:
        small = hnms.batched_multiclass_nms(*args, n_classes=4)
    finally:
        hnms.MASK_WORDS = old
    assert ref[0].shape[0] == 400
    for a, b_ in zip(small, ref):
        assert torch.equal(a, b_)
