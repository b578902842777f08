"""Race hunt: two copies of the model, same initialisation and batch; copy A steps with every kernel on one stream, copy B with the
default multi-stream schedule.  After each backward the gradients are compared parameter by parameter (they must be bit-identical);
the first difference is reported with the parameters it touches.

  python profiles/tools/race_hunt.py [steps] [rv-av2|rv-waymo]
"""
import os, sys, copy; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from range_view_3d_detection_amd import engine as E
from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
widths = sys.argv[2] if len(sys.argv) > 2 else "rv-av2"
dev = torch.device("cuda:0")
W, F, C = (2048, 5, bench.AV2_CLASSES) if widths == "rv-av2" else (2656, 6, 3)
torch.manual_seed(0)
backbone, head = bench.build_model(widths, C, F)
A = bench.Detector(backbone, head).to(dev).train()
B = copy.deepcopy(A)
names = [n for n, _ in A.named_parameters()]
pa, pb = list(A.parameters()), list(B.parameters())
oa, sa = configure_optimizers(pa, num_devices=1, batch_size=4, total_steps=steps + 8, fused=True, max_grad_norm=35.0)
ob, sb = configure_optimizers(pb, num_devices=1, batch_size=4, total_steps=steps + 8, fused=True, max_grad_norm=35.0)
batch = bench.synthetic_batch(4, 64, W, seed=7, device=dev, n_feat=F, n_cls=C)
overlap = E.OVERLAP_WGRAD
assert overlap
bad = 0
for i in range(steps):
    E.OVERLAP_WGRAD = False
    oa.zero_grad(set_to_none=True)
    la = A(batch); la.backward()
    E.OVERLAP_WGRAD = overlap
    ob.zero_grad(set_to_none=True)
    lb = B(batch); lb.backward()
    torch.cuda.synchronize()
    diff = [(n, float((x.grad.float() - y.grad.float()).abs().max()), float(x.grad.float().abs().max())) for n, x, y in zip(names, pa, pb)
            if not torch.equal(x.grad, y.grad)]
    if diff or float(la) != float(lb):
        bad += 1
        print(f"step {i}: loss {float(la)!r} vs {float(lb)!r}; {len(diff)} of {len(names)} gradients differ", flush=True)
        for n, d, m in diff[:12]:
            print(f"    {n}: max |diff| {d:.3e} (max |grad| {m:.3e})", flush=True)
        # re-synchronise B with A so that the hunt goes on from identical states
        for x, y in zip(pa, pb):
            y.grad.copy_(x.grad)
    oa.step(); sa.step(); ob.step(); sb.step()
    if (i + 1) % 50 == 0:
        print(f"step {i + 1}: {bad} divergent steps so far", flush=True)
same = all(torch.equal(x, y) for x, y in zip(pa, pb))
print(f"{widths}: {steps} steps, {bad} divergent; parameters equal at the end: {same}")
