"""In-process A/B of persistent tapconv4 launches on the 1x1 layers (rv_set_option "tapconv5_persist_blocks" covers both kernels)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import engine as E, _lib as L
dev = 'cuda:0'
def run(layer, x, iters=10):
    t = E.Tape(True, dev)
    for _ in range(2): E.ConvOp(t, layer, x, stats=True); t.ops.clear()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): E.ConvOp(t, layer, x, stats=True); t.ops.clear()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,256").split(',')]
lib = L.load()
for (cin, cout, W) in ((256, 256, 18432), (2304, 256, 2048), (256, 256, 2048), (512, 128, 1024)):
    m = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev)
    x = E.Act(torch.randn(4, 64, W, cin, device=dev).to(torch.bfloat16))
    layer = E.tap_layer(m)
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            lib.rv_set_option(b"tapconv5_persist_blocks", L.i32(v))
            res[v].append(run(layer, x))
    gb = 4 * 64 * W * (cin + cout) * 2 / 1e9
    for v in variants:
        r = sorted(res[v]); med = r[len(r) // 2]
        print(f"1x1 {cin}->{cout} W{W} persist {v:4d} median {med:8.1f} us  min {r[0]:8.1f} us  {gb / med * 1e3:6.2f} TB/s", flush=True)
lib.rv_set_option(b"tapconv5_persist_blocks", L.i32(256))
