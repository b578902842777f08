"""Training-recipe glue -- mirrors ``torchbox3d/nn/meta/arch.py:48-75`` (``MetaDetector.configure_optimizers``).

The reference builds, through Hydra, ``torch.optim.AdamW(lr=1e-3)`` (``conf/model/range_view.yaml:52-55``) and a
``OneCycleLR`` stepped once per optimisation step (``"interval": "step"``) whose ``max_lr`` is scaled by
``sqrt(num_devices * batch_size)`` when ``use_linear_lr_scaling`` is set (``conf/model/baseline.yaml:25-28``:
``max_lr = 0.00075``, ``batch_size = 4`` per device) and whose ``total_steps`` is the trainer's
``estimated_stepping_batches``.  Lightning calls ``scheduler.step()`` after every ``optimizer.step()``; a plain training
loop (``bench.py``) does the same with the pair returned here.  ``fused=True`` swaps in ``range_view_3d_detection_amd.optim.AdamW``
(same hyper-parameters, state and arithmetic; optional ``max_grad_norm`` = Lightning's ``gradient_clip_val``).
"""

from __future__ import annotations

import math
from typing import Iterable, Tuple

import torch


def one_cycle_max_lr(max_lr: float, num_devices: int, batch_size: int, use_linear_lr_scaling: bool = True) -> float:
    """``max_lr * sqrt(num_devices * batch_size)`` (``arch.py:63-66``); ``batch_size`` is per device."""
    return max_lr * math.sqrt(num_devices * batch_size) if use_linear_lr_scaling else max_lr


def configure_optimizers(params: Iterable[torch.nn.Parameter], num_devices: int, batch_size: int, total_steps: int,
                         lr: float = 1e-3, max_lr: float = 0.00075, use_linear_lr_scaling: bool = True,
                         debug: bool = False, max_grad_norm: "float | None" = None, fused: bool = False) -> Tuple[torch.optim.Optimizer, "torch.optim.lr_scheduler.LRScheduler | None"]:
    """(AdamW, OneCycleLR stepped per optimisation step) as ``MetaDetector.configure_optimizers`` returns them; in ``debug``
    mode the reference attaches no scheduler (``arch.py:59``)."""
    if fused:  # the same AdamW (and, with max_grad_norm, the recipe's gradient clipping) in two HIP launches: ..optim.AdamW
        from ...optim import AdamW

        optimizer = AdamW(list(params), lr=lr, max_grad_norm=max_grad_norm)
    else:
        if max_grad_norm is not None:
            raise ValueError("max_grad_norm is folded into the fused optimiser only; call torch.nn.utils.clip_grad_norm_ with torch.optim.AdamW")
        optimizer = torch.optim.AdamW(list(params), lr=lr)
    if debug:
        return optimizer, None
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=one_cycle_max_lr(max_lr, num_devices, batch_size, use_linear_lr_scaling),
                                                    total_steps=int(total_steps))
    return optimizer, scheduler
