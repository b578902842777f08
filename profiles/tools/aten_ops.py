"""Which ATen ops (fills, copies, element-wise) a training step of the rv-av2 bench model issues beside the C-ABI launches:
torch.profiler over one steady-state step, ops grouped by name and Python source line."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from range_view_3d_detection_amd import engine as E
Detector = bench.Detector
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers
dev = torch.device("cuda:0")
backbone, head = bench.build_model("rv-av2", bench.AV2_CLASSES, 5)
model = Detector(backbone, head).to(dev).train()
params = list(model.parameters())
opt, sched = configure_optimizers(params, num_devices=1, batch_size=4, total_steps=20, fused=True, max_grad_norm=35.0)
batch = bench.synthetic_batch(4, 64, 2048, seed=1, device=dev, n_feat=5, n_cls=bench.AV2_CLASSES)
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(batch); loss.backward(); opt.step(); sched.step()
for _ in range(3): step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
by = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::add", "aten::mul", "aten::sum", "aten::empty_like"):
        if ev.name in ("aten::zeros", "aten::empty_like"): continue
        key = (ev.name, str(ev.input_shapes)[:100])
        by[key] += 1
for (name, where), n in by.most_common(40):
    print(f"{n:5d} {name:14s} {where}")
