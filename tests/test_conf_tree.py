"""The boundary against the reference's OWN ``conf/`` tree (round-5 review, item 5).

``MetaDetector.__post_init__`` (/root/reference/src/torchbox3d/nn/meta/arch.py:41-46) does
``instantiate(self._backbone)``, ``instantiate(self._head)``, ``instantiate(self._decoder)`` on the composed Hydra config
(``_recursive_: false``: nested configs reach the constructors as configs).  These tests compose
``conf/model/range_view.yaml`` + ``conf/model/baseline.yaml`` + the ``rv-av2`` / ``rv-waymo`` experiment files the way Hydra
does (tests/tools/compose_conf.py: PyYAML only), swap ONLY the three ``_target_``s INTEGRATION.md section 2 names, and build this
package's classes with exactly the kwargs the reference's classes would receive.

* build container (``/root/reference`` present): the composition equals the committed ``tests/golden/conf_kwargs.json`` (data: the
  resolved kwargs, not YAML text) and the three objects construct from it;
* GPU box (no reference): the same dict -> objects -> one eval forward + ``decode`` with the composed ``post_processing_config`` and
  task table.
"""
from __future__ import annotations

import importlib
import math
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "tools"))
import compose_conf as cc  # noqa: E402

HAVE_REF = os.path.isdir("/root/reference/conf")
EXPERIMENTS = ("rv-av2", "rv-waymo")


def hydra_instantiate(cfg):
    """``hydra.utils.instantiate`` for a ``_recursive_: false`` node: reserved keys are consumed, everything else is a keyword
    argument, nested configs are passed through untouched."""
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    for reserved in ("_recursive_", "_convert_", "_partial_", "_args_"):
        cfg.pop(reserved, None)
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(**cfg)


def build_plugins(kwargs):
    kw = cc.swap_targets(kwargs)
    return hydra_instantiate(kw["_backbone"]), hydra_instantiate(kw["_head"]), hydra_instantiate(kw["_decoder"])


@pytest.mark.skipif(not HAVE_REF, reason="composes /root/reference/conf (build container only)")
@pytest.mark.parametrize("experiment", EXPERIMENTS)
def test_composed_conf_tree_equals_the_committed_kwargs(experiment):
    fresh = cc.plugin_kwargs(experiment)
    assert cc.from_wire(cc.to_wire(fresh)) == fresh  # (the wire form loses nothing: integer keys, .inf)
    assert fresh == cc.load_fixture()[experiment], "tests/golden/conf_kwargs.json is stale: python tests/tools/compose_conf.py"
    # the interpolations the constructors depend on, spelled out (range_view.yaml:71, :81-83, :94, :131)
    bb, hd, dec = fresh["_backbone"], fresh["_head"], fresh["_decoder"]
    assert bb["out_channels"] == bb["layers"][0] and bb["_net"]["layers"] == bb["layers"] and bb["_net"]["in_channels"] == bb["in_channels"]
    assert hd["task_in_channels"] == bb["out_channels"] and hd["tasks_cfg"] == fresh["tasks"] == hd["targets_config"]["tasks"]
    assert dec["enable_azimuth_invariant_targets"] is hd["targets_config"]["enable_azimuth_invariant_targets"] is True
    assert hd["fpn_kernel_sizes"] == {1: [3, 3]} and hd["targets_config"]["k"] == math.inf and dec["upper_bounds"][-1] == math.inf
    assert fresh["post_processing_config"]["nms_mode"] == "WEIGHTED" and fresh["trainer"]["precision"] == "bf16-mixed"
    assert set(cc.TARGET_SWAP) == {bb["_target_"], hd["_target_"], dec["_target_"]}


@pytest.mark.parametrize("experiment", EXPERIMENTS)
def test_plugins_construct_from_the_reference_kwargs(experiment):
    """Exactly the reference's kwargs (every key of the composed nodes, `dataset_name`, `_cls_loss`, `targets_config`, `tasks_cfg` ...
    included); a keyword the constructors rejected would be a TypeError here."""
    kwargs = cc.load_fixture()[experiment]
    backbone, head, decoder = build_plugins(kwargs)
    layers = kwargs["_backbone"]["layers"]
    n_cls = len(kwargs["tasks"][0])
    head_c = kwargs["_head"]["classification_head_channels"]
    assert type(backbone).__module__ == "range_view_3d_detection_amd.nn.backbones.dla" and type(backbone.net).__name__ == "RangeBackbone"
    assert type(backbone.stem).__name__ == "MetaKernel" and backbone.layers == layers
    sd_b, sd_h = backbone.state_dict(), head.state_dict()
    assert sd_b["net.res1.blocks.0.net.0.conv.weight"].shape == (layers[0], layers[0], 3, 3)
    # towers: fpn[1] -> head channels, 3x3 (fpn_kernel_sizes[1]); final 1x1 conv -> classes / 8 regressands (range_view.yaml:101-110)
    fpn_c = kwargs["_head"]["fpn"][1]
    n_blocks = kwargs["_head"]["num_classification_blocks"]
    assert sd_h["classification_head.1.0.blocks.0.0.weight"].shape == (head_c, fpn_c, 3, 3)
    assert sd_h[f"classification_head.1.0.blocks.{n_blocks}.0.weight"].shape == (n_cls, head_c, 1, 1)
    assert sd_h[f"regression_head.1.0.blocks.{kwargs['_head']['num_regression_blocks']}.0.weight"].shape == (8, head_c, 1, 1)
    assert type(head.cls_loss).__name__ == "VarifocalLoss" and head.cls_loss.alpha == 0.75 and head.cls_loss.gamma == 2
    assert decoder.subsampling_rates == [8, 2, 1] and decoder.enable_sample_by_range is True
    # the optimiser recipe MetaDetector.configure_optimizers builds from the same tree (arch.py:48-75; baseline.yaml:26-29)
    from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

    params = list(backbone.parameters()) + list(head.parameters())
    opt, sched = configure_optimizers(params, num_devices=8, batch_size=kwargs["batch_size"], total_steps=100, fused=False)
    assert math.isclose(sched.max_lrs[0] if hasattr(sched, "max_lrs") else opt.param_groups[0]["max_lr"],
                        0.00075 * math.sqrt(8 * kwargs["batch_size"]), rel_tol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("experiment", EXPERIMENTS)
def test_reference_kwargs_forward_and_decode_on_device(experiment):
    """The objects built from the reference's kwargs run: eval forward (fp16 autocast, as validation_step: detector.py:329-340) and
    ``decoder.decode(outputs, post_processing_config, tasks)`` as detector.py:352-362 calls it."""
    from bench import synthetic_batch

    kwargs = cc.load_fixture()[experiment]
    torch.manual_seed(0)
    backbone, head, decoder = build_plugins(kwargs)
    dev = torch.device("cuda:0")
    backbone, head = backbone.to(dev).eval(), head.to(dev).eval()
    n_feat = kwargs["_backbone"]["in_channels"]
    assert n_feat == len(kwargs["range_view_config"]["feature_column_names"])
    n_cls = len(kwargs["tasks"][0])
    W = 336 if experiment == "rv-waymo" else 256  # (a 64-row crop; 336 % 64 != 0 as 2656)
    batch = synthetic_batch(2, 64, W, seed=5, device=dev, n_feat=n_feat, n_cls=n_cls)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        feats = backbone(batch)
        assert sorted(feats) == [1, 2, 4, 16] and feats[1].shape == (2, kwargs["_head"]["fpn"][1], 64, W)
        outputs, losses = head(feats, batch, return_loss=False)
        assert losses == {} and outputs[1][0]["logits"].shape == (2, n_cls, 64, W) and outputs[1][0]["regressands"].shape == (2, 8, 64, W)
        outputs[1][0]["logits"] = outputs[1][0]["logits"] + 6.0 * (torch.rand(2, 1, 64, W, device=dev) < 0.02).float()
        params, scores, cats, bidx = decoder.decode(outputs, kwargs["post_processing_config"], kwargs["tasks"], use_nms=True)
    assert params.ndim == 2 and params.shape[1] == 10 and params.shape[0] == scores.shape[0] == cats.shape[0] == bidx.shape[0] > 0
    assert float(scores.min()) >= kwargs["post_processing_config"]["min_confidence"] and int(cats.max()) < n_cls
    assert torch.isfinite(params).all() and set(bidx.unique().tolist()) <= {0, 1}
    # ... and one training step through the same objects (return_loss=True writes the target dicts into `data`: detection_head.py:197-198)
    backbone.train(), head.train()
    feats = backbone(batch)
    _, losses = head(feats, batch, return_loss=True)
    losses["loss"].backward()
    assert torch.isfinite(losses["loss"]) and 1 in batch and 0 in batch[1]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in list(backbone.parameters()) + list(head.parameters()))
