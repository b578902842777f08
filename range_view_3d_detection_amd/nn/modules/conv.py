"""``Conv2dSame`` -- parameter holder with the reference's module path and state-dict keys.

Reference: ``torchbox3d/nn/modules/conv.py:25-80`` (zero "same" padding, total ``k-1`` with
the smaller half on the left/top, applied *before* a strided VALID conv).  The arithmetic
lives in the HIP tap-conv kernel (``csrc/tapconv.hip``); this module only owns
``conv.weight`` (OIHW fp32) and describes the geometry to the engine.
"""

from __future__ import annotations

from torch import Tensor, nn


def _pair(v):
    return tuple(v) if hasattr(v, "__iter__") else (v, v)


class Conv2dSame(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, dilation=1, **kwargs) -> None:
        super().__init__()
        self.conv = nn.Conv2d(
            in_channels, out_channels, kernel_size=_pair(kernel_size), stride=_pair(stride), dilation=_pair(dilation), **kwargs
        )
        kh, kw = _pair(kernel_size)
        # padding folded into the kernel's tap table (no F.pad copy): left/top = (k-1)//2 (conv.py:63-69)
        self.conv._rv_same_pad = ((kh - 1) // 2, (kw - 1) // 2)

    def forward(self, imgs: Tensor) -> Tensor:
        from ... import program

        return program.standalone(self, imgs)
