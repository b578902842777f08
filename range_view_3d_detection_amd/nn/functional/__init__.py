"""Functional interface -- mirrors ``torchbox3d/nn/functional/__init__.py:8-27``.

``varifocal_loss`` is evaluated by the fused detection-loss kernel on the training path
(``csrc/loss.hip``); this stand-alone form exists for API parity and runs the same kernel on a
single-class, background-free problem so that it also works on arbitrary shapes.
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
