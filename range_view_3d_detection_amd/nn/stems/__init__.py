"""Stems -- ``MetaKernel`` (``torchbox3d/nn/stems/__init__.py:12-85``) and ``RangePartition`` (``:88-135``) parameter holders.

``positional_kernel`` / ``fusion_kernel`` are sequences of torchvision-style
``Conv2dNormActivation`` triples; the sub-module names "0" (conv, no bias), "1" (BatchNorm2d),
"2" (ReLU) are part of the reference's state-dict keys and are reproduced here.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from ..blocks import BasicBlock


def conv_norm_act(in_channels: int, out_channels: int, kernel_size=1, norm: bool = True, act: bool = True, padding=None) -> nn.Sequential:
    """Layout of ``torchvision.ops.Conv2dNormActivation``: conv(bias = not norm) [, BatchNorm2d] [, ReLU]."""
    if not isinstance(kernel_size, int):
        kernel_size = tuple(kernel_size)
    k = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
    if padding is None or padding == "same":
        padding = ((k[0] - 1) // 2, (k[1] - 1) // 2)
    layers = [nn.Conv2d(in_channels, out_channels, kernel_size, 1, padding, bias=not norm)]
    if norm:
        layers.append(nn.BatchNorm2d(out_channels))
    if act:
        layers.append(nn.ReLU(inplace=True))
    seq = nn.Sequential(*layers)
    seq.out_channels = out_channels
    return seq


class MetaKernel(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_neighbors: int, num_layers: int = 2) -> None:
        super().__init__()
        if num_neighbors != 3:
            raise NotImplementedError("the HIP MetaKernel path implements the 3x3 neighbourhood every rv-* config uses")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_neighbors, self.num_layers = num_neighbors, num_layers
        self.projection = BasicBlock(in_channels, out_channels, kernel_size=1, project=True)
        self.positional_kernel = nn.Sequential(
            *[conv_norm_act(3 if i == 0 else out_channels, out_channels, 1) for i in range(num_layers)]
        )
        self.fusion_kernel = nn.Sequential(
            *[conv_norm_act(out_channels * num_neighbors**2 if i == 0 else out_channels, out_channels, 1) for i in range(num_layers)]
        )

    def forward(self, features: Tensor, cart: Tensor) -> Tensor:
        from ... import program

        return program.standalone(self, features, cart)


class RangePartition(nn.Module):
    """Six overlapping, closed range bands; every input channel once per band (zero outside it and where the pixel holds no return), then a
    projecting BasicBlock.  ``lower_bounds`` (int64) / ``upper_bounds`` (fp32) are frozen parameters as in the reference -- they are part of
    its ``state_dict``."""

    def __init__(self, in_channels: int, out_channels: int, num_neighbors: int, projection_kernel_size: int, num_layers: int = 2) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_neighbors, self.projection_kernel_size, self.num_layers = num_neighbors, projection_kernel_size, num_layers
        self.projection = BasicBlock(6 * in_channels, out_channels, kernel_size=projection_kernel_size, project=True)
        self.lower_bounds = nn.Parameter(torch.as_tensor((0, 10, 15, 20, 30, 45)).view(1, -1, 1, 1), requires_grad=False)
        self.upper_bounds = nn.Parameter(torch.as_tensor((15, 20, 30, 40, 60, torch.inf)).view(1, -1, 1, 1), requires_grad=False)

    def forward(self, features: Tensor, cart: Tensor, mask: Tensor) -> Tensor:
        from ... import program

        return program.standalone(self, features, cart, mask)
