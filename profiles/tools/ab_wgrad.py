"""In-process A/B of weight-gradient kernel variants; also checks that the variants agree bit for bit.

  python profiles/tools/ab_wgrad.py RV3D_X=0,RV3D_NO_WGRAD3=1
"""
import os, sys, ctypes; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import engine as E
from range_view_3d_detection_amd import _lib as L
dev = 'cuda:0'
def setup(cin, cout, k, N, H, W):
    m = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False).to(dev)
    layer = E.tap_layer(m); g = layer.geom
    x = E.Act(torch.randn(N, H, W, cin, device=dev).to(torch.bfloat16))
    dy = E.Act(torch.randn(N, H, W, cout, device=dev).to(torch.bfloat16))
    wshape = L.TapShape(N, H, W, W, 0, 0, 0)
    ws = torch.empty(L.load().rv_tap_wgrad_workspace_bytes(ctypes.byref(g), ctypes.byref(wshape)), dtype=torch.uint8, device=dev)
    packed = torch.empty((k * k, E.pad32(cout), E.pad32(cin)), dtype=torch.float32, device=dev)
    def run(): L.call("rv_tap_wgrad", ctypes.byref(g), ctypes.byref(wshape), dy.ptr(), L.i32(dy.ld), x.ptr(), L.i32(x.ld), None, None, L.i32(1), L.ptr(packed), L.ptr(ws), L.stream_ptr())
    return run, packed
def time(run, iters=10):
    for _ in range(2): run()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): run()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
variants = sys.argv[1].split(',')
SH = ((512, 512, 2048), (256, 256, 2048), (128, 128, 1024))
if len(sys.argv) > 2: SH = [SH[int(i)] for i in sys.argv[2].split(',')]
for (cin, cout, W) in SH:
    run, packed = setup(cin, cout, 3, 4, 64, W)
    res = {v: [] for v in variants}; outs = {}
    for rnd in range(5):
        for v in variants:
            k, val = v.split('=')
            os.environ[k] = val
            res[v].append(time(run)); outs[v] = packed.clone()
            del os.environ[k]
    fl = 2.0 * 4 * 64 * W * 9 * cin * cout
    ref = outs[variants[-1]]
    for v in variants:
        r = sorted(res[v]); med = r[len(r) // 2]
        print(f"wgrad {cin}->{cout} {v:22s} median {med:8.1f} us  min {r[0]:8.1f}  {fl / med / 1e6:7.1f} TFLOP/s  maxdiff vs last {float((outs[v]-ref).abs().max()):.3e} (max {float(ref.abs().max()):.1f})", flush=True)
