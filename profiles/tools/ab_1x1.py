"""In-process A/B on the pointwise (1x1) conv shapes of the model."""
import os, sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import engine as E
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
variants = sys.argv[1].split(',')
for (cin, cout, W) in ((256, 256, 2048), (256, 256, 18432), (2304, 256, 2048), (256, 2304, 2048), (512, 512, 2048)):
    m = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev)
    x = E.Act(torch.randn(4, 64, W, cin, device=dev).to(torch.bfloat16))
    layer = E.tap_layer(m)
    res = {v: [] for v in variants}
    for rnd in range(3):
        for v in variants:
            k, val = v.split('=')
            os.environ[k] = val
            res[v].append(run(layer, x))
            del os.environ[k]
    gb = 4 * 64 * W * (cin + cout) * 2 / 1e9
    for v in variants:
        r = sorted(res[v]); med = r[len(r) // 2]
        print(f"1x1 {cin}->{cout} W{W} {v:16s} median {med:8.1f} us  {2.0*4*64*W*cin*cout/med/1e6:7.1f} TFLOP/s  {gb/med*1e6/1e3:6.2f} TB/s", flush=True)
