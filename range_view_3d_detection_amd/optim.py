"""``AdamW`` of the training recipe with the step of ALL parameters in two HIP launches (``rv_adamw_step``).

Drop-in for ``torch.optim.AdamW`` where the reference instantiates its optimiser (``conf/model/range_view.yaml:52-55`` ->
``nn/meta/arch.py:57``, ``_target_: torch.optim.AdamW``): same constructor keywords, same ``param_groups`` (so ``OneCycleLR``
drives ``lr`` and -- ``cycle_momentum`` -- ``betas`` exactly as it does torch's), same ``state`` keys (``step``, ``exp_avg``,
``exp_avg_sq``), same arithmetic operation by operation in fp32 (torch/optim/adamw.py, foreach path).  One extra keyword:
``max_grad_norm`` folds ``torch.nn.utils.clip_grad_norm_(params, max_grad_norm)`` (Lightning's ``gradient_clip_val: 35.0``,
``conf/trainer/train.yaml``) into the same two launches: the gradients are scaled on the fly and ``p.grad`` is left as the
backward pass wrote it; the total norm of the last step is ``optimizer.last_grad_norm`` (a device scalar, no host sync).

No CPU fallback: parameters must be fp32 CUDA tensors (``RvError`` otherwise).
"""

from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib as L


class AdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, amsgrad: bool = False, max_grad_norm: Optional[float] = None) -> None:
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by any shipped rv-* recipe")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm: Optional[Tensor] = None
        self._plans: Dict[tuple, dict] = {}

    def _init_state(self, p: Tensor) -> dict:
        st = self.state[p]
        if not st:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise L.RvError("range_view_3d_detection_amd.optim.AdamW: parameters must be contiguous fp32 CUDA tensors (no CPU fallback)")
            st["step"] = torch.tensor(0.0, dtype=torch.float32)  # host scalar, as torch keeps it on the non-capturable path
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _plan(self, ps: List[Tensor], dev) -> dict:
        """Static part of the device tables for this set of parameters (chunk list, p / m / v pointers, buffers)."""
        key = tuple((p.data_ptr(), p.numel(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
        plan = self._plans.get(key)
        if plan is None:
            ce = int(L.load().rv_optim_chunk_elems())
            tens = np.zeros((len(ps), 5), dtype=np.int64)  # (p, g, m, v, n) per tensor = struct OptTensor
            chunks = []
            for i, p in enumerate(ps):
                st = self.state[p]
                tens[i] = (p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                chunks += [(i, c) for c in range((p.numel() + ce - 1) // ce)]
            ch = torch.tensor(chunks, dtype=torch.int32).to(dev)
            self._plans.clear()
            plan = {"tens": tens, "host": torch.empty((len(ps), 5), dtype=torch.int64).pin_memory(), "chunks": ch, "n_chunks": len(chunks),
                    "partial": torch.empty(len(chunks), dtype=torch.float32, device=dev),
                    "table": torch.empty((len(ps), 5), dtype=torch.int64, device=dev),
                    "norm": torch.zeros(1, dtype=torch.float32, device=dev)}
            self._plans[key] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1 and self.max_grad_norm is not None:
            raise NotImplementedError("max_grad_norm is a norm over ALL parameters: use one parameter group")
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                self._init_state(p)
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32 or not g.is_cuda:
                    raise L.RvError("range_view_3d_detection_amd.optim.AdamW: gradients must be dense fp32 CUDA tensors")
                if not g.is_contiguous():
                    p.grad = g.contiguous()
            dev = ps[0].device
            plan = self._plan(ps, dev)
            tens = plan["tens"]
            tens[:, 1] = [p.grad.data_ptr() for p in ps]  # the only per-step part of the table
            if plan.get("copied") is not None:
                plan["copied"].synchronize()  # (the previous step's upload has long finished; this makes the reuse of the pinned buffer exact)
            plan["host"].copy_(torch.from_numpy(tens))
            plan["table"].copy_(plan["host"], non_blocking=True)
            plan["copied"] = torch.cuda.Event()
            plan["copied"].record()
            # torch.optim.AdamW bias-corrects every parameter with ITS OWN step count; one launch covers all parameters (and the
            # clipping norm is over all of them), so the counts must agree -- they do unless a parameter had no gradient on
            # earlier steps (an unused branch, frozen then unfrozen): refuse that rather than correct it with another's count
            steps = {float(self.state[p]["step"]) for p in ps}  # host scalars: no device sync
            if len(steps) != 1:
                raise NotImplementedError(
                    f"optim.AdamW: parameters with different step counts {sorted(steps)} in one step (some had no gradient earlier); "
                    "the fused launch applies one bias correction -- use torch.optim.AdamW for such a schedule")
            step = int(steps.pop()) + 1
            for p in ps:
                self.state[p]["step"] += 1
            beta1, beta2 = group["betas"]
            L.call("rv_adamw_step", L.ptr(plan["table"]), L.ptr(plan["chunks"]), L.i32(plan["n_chunks"]), L.ptr(plan["partial"]),
                   L.f64(float(group["lr"])), L.f64(float(beta1)), L.f64(float(beta2)), L.f64(float(group["eps"])),
                   L.f64(float(group["weight_decay"])), L.i64(step), L.f64(float(self.max_grad_norm) if self.max_grad_norm is not None else 0.0),
                   L.ptr(plan["norm"]), L.stream_ptr())
            # the kernel wrote the parameters and moments behind torch's back: tell autograd (and everything keyed on
            # Tensor._version, e.g. the packed bf16 weight images of engine.TapLayer) that they changed
            torch.autograd.graph.increment_version(ps)
            self.last_grad_norm = plan["norm"]
        return loss
