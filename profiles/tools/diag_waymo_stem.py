"""Soak of the configuration that exposed round 6's wgrad3 fault (profiles/r06_ab_notes.md section 4): free-running two-stream rv-waymo steps with the
stem's 128-channel operand write-out forced ON (engine.MATERIALIZE_POINTWISE_C += 128), so that the stem conv's weight gradient is a wgrad3 launch on the
side stream beside the stem's backward.  Before the fix this died within 15-100 steps on every box (eight runs of eight); the variants of the bisect
(`wgrad_ignores_plain`, `wgrad_on_main`, `guard`, `side_workspace`) were switches of engine.py / engine_bwd.py at commit "wgrad3: tie the dangling LDS
prefetch registers ..." and went with the fix.

    python profiles/tools/diag_waymo_stem.py [steps]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from range_view_3d_detection_amd import engine as E
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

E.MATERIALIZE_POINTWISE_C = (256, 128)
steps = int(sys.argv[-1]) if len(sys.argv) > 1 and sys.argv[-1].isdigit() else 300
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
print("done: no fault in", steps, "steps")
