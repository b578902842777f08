"""Yardstick of the full-size GPU tests: what bf16 storage costs the CPU EMULATION of the model (oracle, ``Numerics.bf16``) against
the fp32 oracle, at the benchmarked widths and image sizes.  The crop tests (tests/test_gpu_realwidth.py) compute this yardstick
on the fly; at 64 x 2048 / 64 x 2656 one emulation pass costs 50-80 s of a 16-core host per test, so it is measured ONCE with this
script and recorded in tests/test_gpu_fullsize_train.py (``EMULATION``).  CPU only; run from the repo root:

    python tests/tools/emulation_yardstick.py [rv-av2|rv-waymo|first-step]
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def train_case(widths, n_feat, n_cls, W):
    from oracle import model as om
    from oracle import targets as otgt
    from test_gpu_realwidth import _prepare

    _, _, sd, batch = _prepare(widths, n_feat, n_cls, W, 3.0)

    def run(nm):
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
        _, logits, reg = om.detector_forward(batch["features"], batch["cart"], {**sd, **params}, nm=nm)
        tg = otgt.compute_targets(batch["cart"], batch["annotations"], n_cls)
        loss = otgt.detection_loss(logits, reg, batch["cart"], batch["mask"], tg, n_cls)["loss"]
        loss.backward()
        return logits.detach(), reg.detach(), float(loss.detach()), {k: p.grad for k, p in params.items()}

    lg32, rg32, loss32, g32 = run(om.Numerics(train=True))
    lg16, rg16, loss16, g16 = run(om.Numerics.bf16(train=True))
    cos = np.array([_cos(g16[k], g32[k]) for k in g32 if float(g32[k].norm()) >= 1e-9])
    return {"emu~fp32": rel_err(lg16, lg32), "reg emu~fp32": rel_err(rg16, rg32), "cos_logits": _cos(lg16, lg32), "cos_reg": _cos(rg16, rg32),
            "loss32": loss32, "loss16": loss16, "grad_cos_median": float(np.median(cos)), "grad_cos_q05": float(np.quantile(cos, 0.05)),
            "grad_cos_min": float(cos.min()), "threads": torch.get_num_threads()}


def first_step():
    import bench
    from oracle import model as om
    from oracle import targets as otgt

    torch.manual_seed(0)
    backbone, head = bench.build_model("rv-av2", bench.AV2_CLASSES)
    sd = {**{f"backbone.{k}": v for k, v in backbone.state_dict().items()}, **{f"head.{k}": v for k, v in head.state_dict().items()}}
    batch = bench.synthetic_batch(1, 64, 2048, seed=1234, device="cpu")
    tg = otgt.compute_targets(batch["cart"], batch["annotations"], bench.AV2_CLASSES)
    out = {}
    with torch.no_grad():
        for name, nm in (("loss32", om.Numerics(train=True)), ("loss16", om.Numerics.bf16(train=True))):
            _, logits, reg = om.detector_forward(batch["features"], batch["cart"], sd, nm=nm)
            out[name] = float(otgt.detection_loss(logits, reg, batch["cart"], batch["mask"], tg, bench.AV2_CLASSES)["loss"])
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["rv-av2", "rv-waymo", "first-step"]
    for w in which:
        r = first_step() if w == "first-step" else train_case(*{"rv-av2": ("rv-av2", 5, 26, 2048), "rv-waymo": ("rv-waymo", 6, 3, 2656)}[w])
        print(w, json.dumps(r), flush=True)
