"""Loss configuration objects -- mirror of ``torchbox3d/nn/losses/classification.py:14-54``.

Instantiated from ``conf/model/range_view.yaml:95-99``; on the training path the parameters are
read by ``DetectionHead`` and handed to the fused HIP loss kernel.
"""

from __future__ import annotations

from dataclasses import dataclass

from torch import Tensor

from ..functional import varifocal_loss


@dataclass
class VarifocalLoss:
    alpha: float
    gamma: float
    reduction: str

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        return varifocal_loss(input=input, target=target, alpha=self.alpha, gamma=self.gamma, reduction=self.reduction)

    def __call__(self, input: Tensor, target: Tensor) -> Tensor:
        return self.forward(input, target)
