"""Is tests/test_gpu_forward.py::test_meta_kernel_positional_pair_fused_matches_unfused[256] deterministic?  One-stream gradients as the
reference, 40 two-stream repetitions compared bit for bit (round 6: the test failed ONCE in a full-suite run with one outlier element in a
weight gradient).  python profiles/tools/flake_probe_meta.py"""
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import torch
import test_gpu_forward as T
from test_gpu_forward import rel_err, DEV
from range_view_3d_detection_amd import engine as E
from range_view_3d_detection_amd.nn.stems import MetaKernel
C = 256
gen = torch.Generator().manual_seed(21)
m = MetaKernel(5, C, 3, 2).to(DEV).train()
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
feats = torch.randn(1, 5, 16, 160, generator=gen).to(DEV)
cart = (torch.randn(1, 3, 16, 160, generator=gen) * 5).to(DEV)
probe = torch.randn(1, C, 16, 160, generator=gen).to(DEV)
def run(fused, overlap=True):
    E.POS_FUSE = fused
    E.OVERLAP_WGRAD = overlap
    m.load_state_dict(sd); m.zero_grad(set_to_none=True)
    out = m(feats, cart).float()
    (out * probe).sum().backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
base = {f: run(f, False) for f in (True, False)}
k = "fusion_kernel.1.0.weight"
print("one-stream fused vs unfused:", rel_err(base[True][k], base[False][k]))
for i in range(40):
    for f in (True, False):
        g = run(f, True)
        bad = [kk for kk in g if not torch.equal(g[kk], base[f][kk])]
        if bad:
            print(f"iter {i} fused={f}: two-stream differs from one-stream in", [(kk, rel_err(g[kk], base[f][kk])) for kk in bad])
print("done")
