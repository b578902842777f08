"""``DenseHead`` -- mirrors ``torchbox3d/nn/heads/dense_head.py:13-76``.

``num_blocks`` x [conv kxk "same" (no bias), BatchNorm2d, ReLU] + a final conv (bias, no
norm / activation); N(0, 0.01) weight init and the focal prior bias ``-log((1-p)/p)``
(``:62-72``).  ``forward`` returns fp32 logits / regressands as an NCHW view.
"""

from __future__ import annotations

import math
from typing import List, Optional

import torch
from torch import Tensor, nn

from ..stems import conv_norm_act


class DenseHead(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_cls: int, kernel_size, final_kernel_size: int,
                 num_blocks: int = 4, prior_prob: Optional[float] = None) -> None:
        super().__init__()
        self.in_channels, self.out_channels, self.num_cls = in_channels, out_channels, num_cls
        self.kernel_size, self.final_kernel_size = kernel_size, final_kernel_size
        self.num_blocks, self.prior_prob = num_blocks, prior_prob
        blocks: List[nn.Module] = [conv_norm_act(in_channels, out_channels, kernel_size, padding="same")]
        for _ in range(num_blocks - 1):
            blocks.append(conv_norm_act(out_channels, out_channels, kernel_size, padding="same"))
        blocks.append(conv_norm_act(out_channels, num_cls, final_kernel_size, norm=False, act=False, padding="same"))
        self.blocks = nn.Sequential(*blocks)
        for block in self.blocks:
            for layer in block:
                if isinstance(layer, nn.Conv2d):
                    torch.nn.init.normal_(layer.weight, std=0.01)
                    if layer.bias is not None:
                        torch.nn.init.zeros_(layer.bias)
        if prior_prob is not None:
            torch.nn.init.constant_(self.blocks[-1][0].bias, -(math.log((1 - prior_prob) / prior_prob)))

    def forward(self, x: Tensor, cart: Tensor = None, mask: Tensor = None) -> Tensor:
        from ... import program
        from ...engine import Act

        def build(t, xin):
            a = Act.from_nchw(xin)
            return [a], [program.dense_head_program(t, self, a)]

        return program.run(build, self, [x])[0]


def forward_pair(cls_head: DenseHead, reg_head: DenseHead, x: Tensor):
    """Classification and regression towers of one (stride, task) as ONE program on one tape: both read the same feature
    tensor, and in backward the second tower's input gradient accumulates into the first one's buffer inside its
    backward-data launch -- run as two autograd nodes, autograd adds the two 268 MB gradients in a pass of its own."""
    from ... import program
    from ...engine import Act

    def build(t, xin):
        a = Act.from_nchw(xin)
        return [a], list(program.dense_head_pair_program(t, cls_head, reg_head, a))

    params = [p for p in cls_head.parameters()] + [p for p in reg_head.parameters()]
    logits, regressands = program._ProgramFn.apply(build, cls_head.training, 1, x, *params)
    return logits, regressands
