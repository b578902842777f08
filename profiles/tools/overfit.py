"""Overfit one synthetic batch (BASELINE configs[0] in spirit: debug-overfit): the rv-av2 model, 2 sweeps of 64 x 512, 150 steps of the
bench's training step (fwd + targets + loss + bwd + clip + AdamW + OneCycleLR); prints the loss every 10 steps.

  python profiles/tools/overfit.py
"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers
dev = torch.device("cuda:0")
torch.manual_seed(0)
backbone, head = bench.build_model("rv-av2", 26)
model = bench.Detector(backbone, head).to(dev).train()
params = list(model.parameters())
steps = 150
opt, sched = configure_optimizers(params, num_devices=1, batch_size=2, total_steps=steps + 8, fused=True, max_grad_norm=35.0)
batch = bench.synthetic_batch(2, 64, 512, seed=7, device=dev, boxes_per_sweep=12)
for i in range(steps):
    opt.zero_grad(set_to_none=True)
    loss = model(batch)
    loss.backward()
    opt.step(); sched.step()
    if i % 10 == 0 or i == steps - 1:
        print(f"step {i:4d}  loss {float(loss.detach()):.5f}  grad norm {float(opt.last_grad_norm):.3f}  lr {sched.get_last_lr()[0]:.2e}", flush=True)
assert torch.isfinite(loss)
