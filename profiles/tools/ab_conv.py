"""In-process A/B of tap-conv kernel variants (interleaved rounds on one device; median / min).

  python profiles/tools/ab_conv.py RV3D_X=0,RV3D_NO_TAPCONV4=1      # each item: an environment switch read at launch time
"""
import os, sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import engine as E
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
variants = sys.argv[1].split(',')   # e.g. "RV3D_TC4_VARIANT=0,RV3D_TC4_VARIANT=1,RV3D_NO_TAPCONV4=1"
shapes = ((512, 512, 2048), (256, 256, 2048), (128, 128, 1024), (128, 128, 512), (256, 128, 2048))
if len(sys.argv) > 2: shapes = [shapes[int(i)] for i in sys.argv[2].split(',')]   # optional: indices of the shapes to run
for (cin, cout, W) in shapes:
    layer, x = setup(cin, cout, 3, 4, 64, W)
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            k, val = v.split('=')
            os.environ[k] = val
            res[v].append(run(layer, x))
            del os.environ[k]
    fl = 2.0 * 4 * 64 * W * 9 * cin * cout
    for v in variants:
        r = sorted(res[v]); med = r[len(r) // 2]
        print(f"{cin}->{cout} W{W} {v:24s} median {med:8.1f} us  min {r[0]:8.1f} us  {fl / med / 1e6:7.1f} TFLOP/s", flush=True)
