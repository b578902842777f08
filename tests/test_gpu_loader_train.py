"""f3 -- the train-split loader chain on device against the reference's own ``__getitem__`` (``split_name == "train"``).

``tests/golden/loader_train_item.npz`` (``make_golden.py loader_train_item``): the reference augments the UNPADDED table
(``prototype/loader.py:598-603``), then ``features *= mask`` and the W padding (``:684-691, 792-815``).  Here:
``range_view_from_table(pad=False)`` -> ``augment_batch`` (``rv_augment``) -> ``pad_batch`` (``rv_pad_range_view``).
Bar: mask and every placement channel bit for bit, geometry 1e-6 of the channel maximum, boxes 1e-9; the seeded ``random``
draws equal the reference's.
"""

from __future__ import annotations

import random

import numpy as np
import pytest
import torch

from test_gpu_forward import DEV
from test_oracle_golden import TRAIN_TASKS, _train_item_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["av2", "waymo", "av2_dropout"])
def test_train_item_chain_matches_the_reference(golden, tag):
    from oracle import augment as oaug
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd.prototype import loader as ld

    g = golden("loader_train_item")
    names, table, ann_in, chain, roi, mode = _train_item_case(g, tag)
    cfg = {"feature_column_names": names, "filter_roi": roi, "height": 8, "width": 64}
    order = [str(n) for n in g.np(f"{tag}/augmentation_order")]
    aug_cfg = {"flip_azimuth": {"p": 1.0}, "random_rotation": {"low": -0.78539816, "high": 0.78539816, "p": 1.0} if tag == "av2" else {"low": 2.0, "high": 3.0, "p": 1.0},
               "random_global_scale": {"low": 0.95, "high": 1.05}, "random_global_translation": {"std_x": 0.5, "std_y": 0.5, "std_z": 0.2}}
    aug_cfg["point_dropout"] = {"p": 0.8}
    aug_cfg = {k: aug_cfg[k] for k in order}
    ann = ld.annotations_for_sweep(ann_in, 7, TRAIN_TASKS)
    assert ann[:, 10].tolist() == g.np(f"{tag}/ann_out/task_id").tolist() and ann[:, 11].tolist() == g.np(f"{tag}/ann_out/offset").tolist()
    random.seed(int(g.np(f"{tag}/seed")))
    np.random.seed(int(g.np(f"{tag}/seed")))  # point_dropout draws from numpy's global generator, as the reference does
    ds = "av2" if tag.startswith("av2") else "waymo"
    out = ld.train_batch_from_tables([table], ann, cfg, ds, aug_cfg, 1, mode, device=DEV)
    tr = out["transforms"][0]
    drawn = {op[0]: op[1] for op in tr.ops if len(op) > 1}
    if f"{tag}/theta" in g:
        assert drawn["rotate"] == float(g.np(f"{tag}/theta"))
    assert drawn["scale"] == float(g.np(f"{tag}/scale"))
    if f"{tag}/t" in g:
        assert list(drawn["translate"]) == g.np(f"{tag}/t").tolist()
    ref_f, ref_c, ref_m = g.np(f"{tag}/features"), g.np(f"{tag}/cart"), g.np(f"{tag}/mask")
    assert out["mask"].dtype == torch.bool and np.array_equal(out["mask"][0].cpu().numpy(), ref_m), tag
    got = out["features"][0].cpu().numpy()
    assert got.shape == ref_f.shape
    for i, n in enumerate(names):
        tol = 1e-6 * max(1.0, float(np.abs(ref_f[i]).max()))
        if n in ("x", "y", "z", "range") or (n == "intensity" and ds == "waymo"):
            assert np.max(np.abs(got[i] - ref_f[i])) <= tol, (tag, n)
        else:
            assert np.array_equal(got[i], ref_f[i]), (tag, n)
    assert np.max(np.abs(out["cart"][0].cpu().numpy() - ref_c)) <= 1e-6 * max(1.0, float(np.abs(ref_c).max())), tag
    cols = ("tx_m", "ty_m", "tz_m", "length_m", "width_m", "height_m", "qw", "qx", "qy", "qz")
    ref_a = np.stack([g.np(f"{tag}/ann_out/{c}") for c in cols])
    ga = out["annotations"].numpy()[:, :10].T
    assert np.max(np.abs(ga[:6] - ref_a[:6])) <= 1e-9 * max(1.0, np.max(np.abs(ref_a[:6]))), tag
    dyaw = oaug.yaw_of(ga[6:10]) - oaug.yaw_of(ref_a[6:10])
    assert np.max(np.abs(np.arctan2(np.sin(dyaw), np.cos(dyaw)))) < 1e-9, tag
    # augmenting an already padded batch is refused when the configured width is given
    padded = ld.range_view_from_table(table, cfg, ds, 1, mode, device=DEV)
    batch = {k: v[None] for k, v in padded.items()}
    with pytest.raises(L.RvError, match="BEFORE the W padding"):
        ld.augment_batch(batch, names, {k: v for k, v in aug_cfg.items() if k != "point_dropout"}, width=64)
