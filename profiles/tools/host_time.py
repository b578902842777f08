"""Host (Python + ctypes + allocator) time to ENQUEUE one training step against the GPU time of the step.

  python profiles/tools/host_time.py
"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers
dev = torch.device("cuda:0")
torch.manual_seed(0)
backbone, head = bench.build_model("rv-av2", 26)
model = bench.Detector(backbone, head).to(dev).train()
params = list(model.parameters())
opt, sched = configure_optimizers(params, num_devices=1, batch_size=4, total_steps=100, fused=True, max_grad_norm=35.0)
batch = bench.synthetic_batch(4, 64, 2048, seed=1234, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    t0 = time.perf_counter(); loss = model(batch); t1 = time.perf_counter()
    loss.backward(); t2 = time.perf_counter()
    opt.step(); sched.step(); t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for _ in range(3): step()
torch.cuda.synchronize()
for trial in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    f, b, o = step()
    th = time.perf_counter() - t
    torch.cuda.synchronize(); tg = time.perf_counter() - t
    print(f"host enqueue {1e3 * th:6.1f} ms (forward+loss {1e3 * f:5.1f}, backward {1e3 * b:5.1f}, optimizer {1e3 * o:4.1f}); step complete after {1e3 * tg:6.1f} ms", flush=True)
