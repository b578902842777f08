"""Bisect of the rv-waymo fault of round 6 (profiles/r06_ab_notes.md section 4): free-running two-stream rv-waymo steps with the stem's 128-channel
operand write-out forced ON (engine.MATERIALIZE_POINTWISE_C += 128), optionally with that layer's weight gradient kept off the written-out tensor.

    python profiles/tools/diag_waymo_stem.py <variant> [steps]      variant: control | wgrad_ignores_plain | wgrad_on_main | guard | side_workspace
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from range_view_3d_detection_amd import engine as E, engine_bwd as EB
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

E.MATERIALIZE_POINTWISE_C = (256, 128)
variant = sys.argv[1] if len(sys.argv) > 1 else "control"
assert variant in ("control", "wgrad_ignores_plain", "wgrad_on_main", "guard", "side_workspace"), variant
EB.DIAG_SIDE_WORKSPACE = variant == "side_workspace"
EB.DIAG_WGRAD_IGNORES_PLAIN_1X1 = variant == "wgrad_ignores_plain"
EB.DIAG_PLAIN_1X1_WGRAD_ON_MAIN = variant == "wgrad_on_main"
E.DIAG_PLAIN_GUARD_PIXELS = 4096 if variant == "guard" else 0  # (1 MB of zeros either side of every written-out operand)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.manual_seed(0)
backbone, head = bench.build_model("rv-waymo", 3, 6)
batch = bench.synthetic_batch(4, 64, 2656, seed=4321, device=dev, n_feat=6, n_cls=3)
model = bench.Detector(backbone, head).to(dev).train()
opt, sched = configure_optimizers(list(model.parameters()), num_devices=1, batch_size=4, total_steps=10_000, fused=True, max_grad_norm=35.0)
t0 = time.perf_counter()
for i in range(steps):
    opt.zero_grad(set_to_none=True)
    loss = model(batch)
    loss.backward()
    opt.step()
    sched.step()
    if i % 50 == 49:
        torch.cuda.synchronize()
        print(f"step {i + 1}: loss {float(loss.detach()):.5f}, {1e3 * (time.perf_counter() - t0) / (i + 1):.2f} ms per step", flush=True)
torch.cuda.synchronize()
print("done: no fault in", steps, "steps; variant =", variant)
