"""Soak: 300 training steps of the bench configuration (rv-av2, 4 x 64 x 2048 x 5) followed by 20 eval forward + decode + NMS
batches under fp16 autocast; prints loss, step time and the allocator's peak every 50 steps (no growth after warm-up expected)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers
dev = torch.device("cuda:0")
torch.manual_seed(0)
backbone, head = bench.build_model("rv-av2", bench.AV2_CLASSES)
model = bench.Detector(backbone, head).to(dev).train()
params = list(model.parameters())
steps = 300
opt, sched = configure_optimizers(params, num_devices=1, batch_size=4, total_steps=steps + 8, fused=True, max_grad_norm=35.0)
batch = bench.synthetic_batch(4, 64, 2048, seed=7, device=dev, n_feat=5, n_cls=bench.AV2_CLASSES)
t0 = time.perf_counter()
for i in range(steps):
    opt.zero_grad(set_to_none=True)
    loss = model(batch)
    loss.backward()
    opt.step(); sched.step()
    if (i + 1) % 50 == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"step {i + 1:4d}  loss {float(loss.detach()):.5f}  {1e3 * (t1 - t0) / 50:.1f} ms/step  peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB  "
              f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
        t0 = time.perf_counter()
assert torch.isfinite(loss)
