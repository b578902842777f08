"""Full-size training-step parity (round-3 review, item 4): the composed model at the BENCHMARKED widths and image sizes,
B = 1, train mode, against the pinned oracle -- forward, targets (exact), loss (1e-2) and per-parameter gradient cosines with the
CPU bf16 emulation as the yardstick, exactly the checks tests/test_gpu_realwidth.py makes on 64 x 256 / 64 x 336 crops, but with
the library's own kernel selection at W = 2048 / 2656: every 3x3 layer on tapconv6, 32 tile columns per row, split-K rounds
of the weight gradient over the whole image, the ragged last column chunk of 2656 = 41 x 64 + 32.

Slow: two oracle passes (fp32, bf16 storage points) of a whole sweep each, ~60-80 s per pass for rv-av2 on 16 host cores.
"""

from __future__ import annotations

import pytest
import torch

from test_gpu_forward import DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("widths,n_feat,n_cls,W", [("rv-av2", 5, 26, 2048), ("rv-waymo", 6, 3, 2656)])
def test_full_size_train_step_vs_oracle(widths, n_feat, n_cls, W):
    from test_gpu_realwidth import _train_step_vs_oracle

    _train_step_vs_oracle(widths, n_feat, n_cls, W, 3.0, small_grids=False)


def test_bench_first_step_loss_matches_the_oracle():
    """``loss_first_step`` of the bench line -- the benchmark's own model (seed 0, natural initialisation), its own synthetic data
    (seed 1234), sweep 0, train mode, before any update -- against the fp32 oracle's loss of the same model on the same sweep:
    1e-2 relative (the bound of the crop tests), and within 2x the CPU bf16 emulation's own distance + 2e-3."""
    import bench
    from oracle import model as om
    from oracle import targets as otgt

    got = bench.first_step_loss(DEV)
    torch.manual_seed(0)
    backbone, head = bench.build_model("rv-av2", bench.AV2_CLASSES)
    sd = {**{f"backbone.{k}": v for k, v in backbone.state_dict().items()}, **{f"head.{k}": v for k, v in head.state_dict().items()}}
    batch = bench.synthetic_batch(1, 64, 2048, seed=1234, device="cpu")
    torch.set_num_threads(min(32, torch.get_num_threads()))
    tg = otgt.compute_targets(batch["cart"], batch["annotations"], bench.AV2_CLASSES)
    losses = {}
    with torch.no_grad():
        for name, nm in (("fp32", om.Numerics(train=True)), ("bf16", om.Numerics.bf16(train=True))):
            _, logits, reg = om.detector_forward(batch["features"], batch["cart"], sd, nm=nm)
            losses[name] = float(otgt.detection_loss(logits, reg, batch["cart"], batch["mask"], tg, bench.AV2_CLASSES)["loss"])
    print(f"loss_first_step: HIP {got:.6f}  oracle fp32 {losses['fp32']:.6f}  oracle bf16 emulation {losses['bf16']:.6f}")
    assert abs(got - losses["fp32"]) / abs(losses["fp32"]) < 1e-2, (got, losses)
    assert abs(got - losses["fp32"]) <= 2.0 * abs(losses["bf16"] - losses["fp32"]) + 2e-3 * abs(losses["fp32"]), (got, losses)
