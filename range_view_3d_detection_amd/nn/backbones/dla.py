"""Range-view backbone -- mirrors ``torchbox3d/nn/backbones/dla.py``.

``RangeBackbone`` (:15-131) and ``RangeNet`` (:134-208) keep the reference's constructor
keywords (so ``conf/model/range_view.yaml`` works with only ``_target_`` changed), attribute
names (state-dict keys) and the ``forward(dict) -> {1,2,4,16: Tensor}`` contract.  The
returned tensors are bf16 ``channels_last`` views of the engine's NHWC buffers (what the
reference produces under its bf16 autocast with cuDNN's preferred layout).
"""

from __future__ import annotations

import importlib
from typing import Any, Dict, Mapping

from torch import Tensor, nn

from ..blocks import AggregationBlock, BasicBlock, ResidualBlock
from ..stems import MetaKernel, RangePartition


class RangeBackbone(nn.Module):
    """Range View Net (based on DLA): 5 residual stages (W-stride only) + 4 transposed-conv aggregation blocks."""

    def __init__(self, in_channels: int, layers, out_channels: int) -> None:
        super().__init__()
        layers = list(layers)
        self.in_channels, self.layers, self.out_channels = in_channels, layers, out_channels
        self.res1 = ResidualBlock(layers[0], layers[0], stride=(1, 1), num_blocks=2)
        self.res2a = ResidualBlock(layers[0], layers[1], stride=(1, 2), num_blocks=3)
        self.res2 = ResidualBlock(layers[1], layers[2], stride=(1, 2), num_blocks=3)
        self.res3a = ResidualBlock(layers[2], layers[3], stride=(1, 2), num_blocks=5)
        self.res3 = ResidualBlock(layers[3], layers[4], stride=(1, 2), num_blocks=5)
        self.agg2 = AggregationBlock(layers[2], layers[4], layers[2], kernel_size=(3, 8), stride=(1, 4), padding=(1, 2), num_blocks=2)
        self.agg1 = AggregationBlock(layers[0], layers[2], layers[0], kernel_size=(3, 8), stride=(1, 4), padding=(1, 2), num_blocks=2)
        self.agg2a = AggregationBlock(layers[1], layers[2], layers[1], kernel_size=(3, 4), stride=(1, 2), padding=(1, 1), num_blocks=1)
        self.agg3 = AggregationBlock(layers[0], layers[1], layers[0], kernel_size=(3, 4), stride=(1, 2), padding=(1, 1), num_blocks=2)

    def forward(self, features: Tensor, cart: Tensor, mask: Tensor) -> Dict[int, Tensor]:
        from ... import program
        from ...engine import Act

        def build(t, x):
            a = Act.from_nchw(x)
            outs = program.range_backbone_program(t, self, a, None)
            return [a], [outs[1], outs[2], outs[4], outs[16]]

        o = program.run(build, self, [features])
        return {1: o[0], 2: o[1], 4: o[2], 16: o[3]}


def _instantiate(cfg: Mapping[str, Any]) -> Any:
    """Non-recursive ``hydra.utils.instantiate`` (the only form the reference uses): ``_target_`` + kwargs.

    A ``torchbox3d.`` target prefix is mapped onto this package so that the inner ``_net`` entry
    of ``conf/model/range_view.yaml:79-82`` needs no edit.
    """
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.pop("_recursive_", None)
    if target.startswith("torchbox3d."):
        target = "range_view_3d_detection_amd." + target[len("torchbox3d."):]
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(**cfg)


class RangeNet(nn.Module):
    """Stem dispatch (META / RANGE_PARTITION / BASIC, ``nn/backbones/dla.py:157-180``) + trunk."""

    def __init__(self, in_channels: int, layers, out_channels: int, projection_kernel_size: int, dataset_name: str,
                 num_neighbors: int, num_layers: int, stem_type: str, _net: Mapping[str, Any], compile: bool = False) -> None:
        super().__init__()
        self.in_channels, self.layers, self.out_channels = in_channels, list(layers), out_channels
        self.projection_kernel_size, self.dataset_name = projection_kernel_size, dataset_name
        self.num_neighbors, self.num_layers, self.stem_type = num_neighbors, num_layers, stem_type
        self._net, self.compile = _net, compile  # ``compile`` is accepted and ignored (no tracing compiler here)
        if stem_type == "META":
            self.stem = MetaKernel(in_channels=in_channels, out_channels=self.layers[0], num_neighbors=num_neighbors, num_layers=num_layers)
        elif stem_type == "RANGE_PARTITION":
            self.stem = RangePartition(in_channels=in_channels, out_channels=self.layers[0], num_neighbors=num_neighbors, num_layers=num_layers,
                                       projection_kernel_size=projection_kernel_size)
        elif stem_type == "BASIC":
            self.stem = BasicBlock(in_channels, self.layers[0], kernel_size=projection_kernel_size, project=True)
        else:
            raise NotImplementedError("This stem type is not implemented!")
        self.net = _instantiate(_net)

    def forward(self, x: Dict[str, Tensor]) -> Dict[int, Tensor]:
        from ... import program

        features, cart = x["features"], x["cart"]
        mask = x["mask"] if self.stem_type == "RANGE_PARTITION" else None  # (the other stems do not read it: dla.py:200-205)

        def build(t, f, c):
            outs = program.range_net_program(t, self, f, c, mask)
            return [None, None], [outs[1], outs[2], outs[4], outs[16]]

        o = program.run(build, self, [features, cart])
        return {1: o[0], 2: o[1], 4: o[2], 16: o[3]}
