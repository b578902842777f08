"""Cost of one latency-bound SyncBN collective on the critical path, one-rank RCCL group on one GPU:
torch.distributed.all_reduce (c10d + ProcessGroupNCCL's stream) against ncclAllReduce enqueued on the compute stream.

  MASTER_ADDR=127.0.0.1 MASTER_PORT=29688 RANK=0 WORLD_SIZE=1 python profiles/tools/mb_collective.py
"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29688")
dist.init_process_group("nccl", device_id=torch.device("cuda:0"), rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)))
torch.cuda.set_device(0)
from range_view_3d_detection_amd import rccl
x = torch.randn(4096, 4096, device="cuda")
buf = torch.zeros(1025, device="cuda")
def chain(reduce, n=130):
    y = x
    for _ in range(n):
        y = y * 1.0001          # a kernel before (stands in for the conv)
        buf[0] = 1.0            # the count slot
        reduce(buf)
        y = y + buf[0] * 0.0    # a kernel that depends on the reduced buffer (the finalize)
    return y
for name, fn in (("no collective", lambda b: None), ("torch.distributed.all_reduce", dist.all_reduce), ("rccl.all_reduce_ (compute stream)", rccl.all_reduce_)):
    chain(fn, 10); torch.cuda.synchronize()
    t = time.perf_counter(); chain(fn); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{name:36s} {1e3 * dt:7.2f} ms for 130 links = {1e6 * dt / 130:6.1f} us per link", flush=True)
b2 = torch.arange(8, device="cuda", dtype=torch.float32); rccl.all_reduce_(b2); torch.cuda.synchronize(); print("one-rank sum:", b2.tolist())
rccl.shutdown(); dist.destroy_process_group()
