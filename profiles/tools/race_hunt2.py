"""Race hunt 2: R repetitions of N training steps from the same seed in one process (multi-stream default), checksum of the parameters
after each; optional per-step synchronisation (argument 3 = 1) and the bench's per-launch event profile (argument 4 = 1).

  python profiles/tools/race_hunt2.py [reps] [steps] [sync_each_step] [profile]
"""
import hashlib, os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from range_view_3d_detection_amd import engine as E
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
sync_each = len(sys.argv) > 3 and sys.argv[3] == "1"
profile = len(sys.argv) > 4 and sys.argv[4] in ("1", "2")
reset_each = len(sys.argv) > 4 and sys.argv[4] == "2"  # a fresh profile (its events destroyed) every step
dev = torch.device("cuda:0")
batch = bench.synthetic_batch(4, 64, 2048, seed=1234, device=dev)
sums = {}
for r in range(reps):
    torch.manual_seed(0)
    backbone, head = bench.build_model("rv-av2", bench.AV2_CLASSES)
    model = bench.Detector(backbone, head).to(dev).train()
    params = list(model.parameters())
    opt, sched = configure_optimizers(params, num_devices=1, batch_size=4, total_steps=steps + 8, fused=True, max_grad_norm=35.0)
    E.PROFILE = E.KernelProfile() if profile else None
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = model(batch)
        loss.backward()
        opt.step(); sched.step()
        if sync_each:
            torch.cuda.synchronize()
        if reset_each:
            E.PROFILE = E.KernelProfile()
    torch.cuda.synchronize()
    E.PROFILE = None
    h = hashlib.sha256()
    for p in params:
        h.update(p.detach().float().cpu().numpy().tobytes())
    k = h.hexdigest()[:12]
    sums[k] = sums.get(k, 0) + 1
    print(r, k, float(loss.detach()), flush=True)
print("distinct parameter checksums:", sums)
