"""Functional interface -- mirrors ``torchbox3d/nn/functional/__init__.py:8-27``.

On the training path the varifocal term is evaluated INSIDE the fused detection-loss kernel (``csrc/loss.hip``,
``nn/heads/detection_head.py``); this stand-alone ``varifocal_loss`` exists for API parity only and is plain torch ops on
the caller's device (arbitrary shapes) -- it does not call the HIP library and is not on the hot path.
"""

from __future__ import annotations

import torch
from torch import Tensor


def varifocal_loss(input: Tensor, target: Tensor, alpha: float, gamma: float, reduction: str = "none") -> Tensor:
    """``[t>0] t bce + alpha [t==0] sigmoid(x)^gamma bce`` (element-wise, torch ops on the caller's device)."""
    bce = torch.nn.functional.binary_cross_entropy_with_logits(input, target, reduction="none")
    p = input.sigmoid()
    loss = (target > 0.0) * target * bce + alpha * (target == 0) * p.pow(gamma) * bce
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss
