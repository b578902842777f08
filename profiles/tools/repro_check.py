"""Race check of the multi-stream step: N training steps of the bench configuration, twice from the same seed in one process, and
once more with every kernel on ONE stream (RV3D_OVERLAP=off semantics); the parameters after the last step must be bit-identical
in all three (every kernel on the path is deterministic; a missing stream dependency would show as a difference).

  python profiles/tools/repro_check.py [steps] [rv-av2|rv-waymo]
"""
import hashlib, os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from range_view_3d_detection_amd import engine as E
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
widths = sys.argv[2] if len(sys.argv) > 2 else "rv-av2"
dev = torch.device("cuda:0")
W, F, C = (2048, 5, bench.AV2_CLASSES) if widths == "rv-av2" else (2656, 6, 3)


def run(overlap: bool):
    torch.manual_seed(0)
    backbone, head = bench.build_model(widths, C, F)
    model = bench.Detector(backbone, head).to(dev).train()
    params = list(model.parameters())
    opt, sched = configure_optimizers(params, num_devices=1, batch_size=4, total_steps=steps + 8, fused=True, max_grad_norm=35.0)
    batch = bench.synthetic_batch(4, 64, W, seed=7, device=dev, n_feat=F, n_cls=C)
    saved, E.OVERLAP_WGRAD = E.OVERLAP_WGRAD, overlap and E.OVERLAP_WGRAD
    try:
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = model(batch)
            loss.backward()
            opt.step(); sched.step()
        torch.cuda.synchronize()
    finally:
        E.OVERLAP_WGRAD = saved
    h = hashlib.sha256()
    for p in params:
        h.update(p.detach().float().cpu().numpy().tobytes())
    for b in model.buffers():
        h.update(b.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()[:16], float(loss.detach())


a, b, c = run(True), run(True), run(False)
print(f"{widths}, {steps} steps: two streams {a}, again {b}, one stream {c}")
assert a[0] == b[0] == c[0], "parameters differ between runs: a stream dependency is missing (or a kernel is not deterministic)"
print("bit-identical")
