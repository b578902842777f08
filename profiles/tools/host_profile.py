"""cProfile of the HOST side of training steps (where the Python + ctypes + allocator time goes), local or synchronised path.

  python profiles/tools/host_profile.py            # local
  RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1 python profiles/tools/host_profile.py    # one-rank RCCL group, SyncBN + GradSync
"""
import cProfile, os, pstats, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from range_view_3d_detection_amd import engine as E
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers
dev = torch.device("cuda:0")
dist_on = os.environ.get("RV3D_FORCE_DIST") is not None
if dist_on:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29677")
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
torch.manual_seed(0)
backbone, head = bench.build_model("rv-av2", 26)
model = bench.Detector(backbone, head).to(dev).train()
params = list(model.parameters())
E.SYNC_BN = dist_on
if dist_on:
    E.GRAD_SYNC = E.GradSync(params, 1)
opt, sched = configure_optimizers(params, num_devices=1, batch_size=4, total_steps=100, fused=True, max_grad_norm=35.0)
batch = bench.synthetic_batch(4, 64, 2048, seed=1234, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(batch)
    loss.backward()
    if E.GRAD_SYNC is not None:
        E.GRAD_SYNC.finish()
    opt.step(); sched.step()
for _ in range(3): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): step()
th = time.perf_counter() - t
torch.cuda.synchronize(); tg = time.perf_counter() - t
print(f"5 steps: host enqueue {1e3 * th / 5:.1f} ms per step, complete after {1e3 * tg / 5:.1f} ms per step")
# (the backward of CUDA tensors runs on an autograd worker thread that cProfile does not follow: RV3D_HOST_PROFILE_BWD=1 keeps it on
#  the calling thread for this listing)
if os.environ.get("RV3D_HOST_PROFILE_BWD"):
    torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
