"""The multi-stream training step (weight gradients free-running on a second stream, early release behind ragged backward-data
launches: engine.py RV3D_OVERLAP / RV3D_EARLY_WGRAD_FILL) against the same step with every kernel on ONE stream: the parameters
after a few optimizer steps must be bit-identical.  Every kernel on the path is deterministic, so a missing dependency between
the streams shows up as a difference (and so does a kernel that stops being deterministic)."""
import hashlib
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _run(widths: str, width: int, feats: int, classes: int, steps: int, overlap: bool):
    import bench
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

    torch.manual_seed(0)
    backbone, head = bench.build_model(widths, classes, feats)
    model = bench.Detector(backbone, head).to(DEV).train()
    params = list(model.parameters())
    opt, sched = configure_optimizers(params, num_devices=1, batch_size=2, total_steps=steps + 8, fused=True, max_grad_norm=35.0)
    batch = bench.synthetic_batch(2, 64, width, seed=7, device=DEV, n_feat=feats, n_cls=classes)
    saved = E.OVERLAP_WGRAD
    E.OVERLAP_WGRAD = saved and overlap
    try:
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = model(batch)
            loss.backward()
            opt.step()
            sched.step()
        torch.cuda.synchronize()
    finally:
        E.OVERLAP_WGRAD = saved
    h = hashlib.sha256()
    for t in list(params) + list(model.buffers()):
        h.update(t.detach().float().cpu().numpy().tobytes())
    return h.hexdigest(), float(loss.detach())


@pytest.mark.parametrize("widths,width,feats,classes", [("rv-av2", 2048, 5, 26),      # every persistent launch is whole rounds of tiles
                                                        ("rv-waymo", 2656, 6, 3)])    # ragged last rounds: the early-release path
def test_two_stream_step_is_bit_identical_to_the_one_stream_step(widths, width, feats, classes):
    from range_view_3d_detection_amd import engine as E

    assert E.OVERLAP_WGRAD, "the default overlap mode is expected to use the side stream"
    a = _run(widths, width, feats, classes, 3, True)
    b = _run(widths, width, feats, classes, 3, True)
    c = _run(widths, width, feats, classes, 3, False)
    assert a[0] == b[0], ("two runs of the two-stream step differ", a[1], b[1])
    assert a[0] == c[0], ("the two-stream step differs from the one-stream step", a[1], c[1])
