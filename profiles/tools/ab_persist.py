"""In-process A/B of the persistent tapconv5 launch (rv_set_option "tapconv5_persist_blocks": 0 = one workgroup per tile)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import engine as E, _lib as L
dev = 'cuda:0'
def setup(cin, cout, k, N, H, W):
    m = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False).to(dev)
    x = E.Act(torch.randn(N, H, W, cin, device=dev).to(torch.bfloat16))
    return E.tap_layer(m), x
def run(layer, x, iters=10):
    t = E.Tape(True, dev)
    for _ in range(2): E.ConvOp(t, layer, x, stats=True); t.ops.clear()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): E.ConvOp(t, layer, x, stats=True); t.ops.clear()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,256,512").split(',')]
lib = L.load()
for (cin, cout, W) in ((512, 512, 2048), (256, 256, 2048), (128, 128, 1024), (128, 128, 512)):
    layer, x = setup(cin, cout, 3, 4, 64, W)
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            lib.rv_set_option(b"tapconv5_persist_blocks", L.i32(v))
            res[v].append(run(layer, x))
    fl = 2.0 * 4 * 64 * W * 9 * cin * cout
    for v in variants:
        r = sorted(res[v]); med = r[len(r) // 2]
        print(f"{cin}->{cout} W{W} persist {v:4d} median {med:8.1f} us  min {r[0]:8.1f} us  {fl / med / 1e6:7.1f} TFLOP/s", flush=True)
lib.rv_set_option(b"tapconv5_persist_blocks", L.i32(256))
