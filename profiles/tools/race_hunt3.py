"""Race hunt 3: the 25-step trajectory of the bench configuration, once on ONE stream (reference), then R times on the default
multi-stream schedule, free-running (no synchronisation inside a run).  Every step leaves a fingerprint on the device -- the norm
of each parameter's gradient (238 numbers, `torch._foreach_norm`) -- and the runs are compared with the reference afterwards: the
first step and the parameters whose gradients differ name the launch that went wrong.

  python profiles/tools/race_hunt3.py [reps] [steps] [rv-av2|rv-waymo]
"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from range_view_3d_detection_amd import engine as E
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
widths = sys.argv[3] if len(sys.argv) > 3 else "rv-av2"
dev = torch.device("cuda:0")
if os.environ.get("RV3D_FORCE_DIST") is not None:  # the synchronised path with one rank (where the wrong steps are ~10x as frequent)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29671")
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    E.SYNC_BN = True
W, F, C = (2048, 5, bench.AV2_CLASSES) if widths == "rv-av2" else (2656, 6, 3)
batch = bench.synthetic_batch(4, 64, W, seed=1234, device=dev, n_feat=F, n_cls=C)
names = None


def trajectory(overlap: bool):
    global names
    torch.manual_seed(0)
    backbone, head = bench.build_model(widths, C, F)
    model = bench.Detector(backbone, head).to(dev).train()
    names = [n for n, _ in model.named_parameters()]
    params = list(model.parameters())
    opt, sched = configure_optimizers(params, num_devices=1, batch_size=4, total_steps=steps + 8, fused=True, max_grad_norm=35.0)
    saved = E.OVERLAP_WGRAD
    E.OVERLAP_WGRAD = saved and overlap
    prints = []
    try:
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = model(batch)
            loss.backward()
            prints.append(torch.stack(torch._foreach_norm([p.grad for p in params])))
            opt.step(); sched.step()
        torch.cuda.synchronize()
    finally:
        E.OVERLAP_WGRAD = saved
    return torch.stack(prints).cpu()


ref = trajectory(False)
ref2 = trajectory(False)
assert torch.equal(ref, ref2), "the one-stream trajectory itself is not reproducible"
bad = 0
for r in range(reps):
    got = trajectory(True)
    if (r + 1) % 5 == 0:
        print(f"... {r + 1} of {reps} runs, {bad} with a wrong step so far", flush=True)
    if torch.equal(got, ref):
        continue
    bad += 1
    step = int((got != ref).any(dim=1).nonzero()[0])
    idx = (got[step] != ref[step]).nonzero().flatten().tolist()
    print(f"run {r}: first difference at step {step}: {len(idx)} of {len(names)} gradients", flush=True)
    for i in idx[:10]:
        print(f"    {names[i]}: norm {float(got[step, i])!r} against {float(ref[step, i])!r}", flush=True)
print(f"{widths}: {reps} runs of {steps} steps on the multi-stream schedule, {bad} with a wrong step")
