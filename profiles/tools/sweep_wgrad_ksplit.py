"""Split-K factor of wgrad3 against time, per shape (RV3D_WGRAD_KSPLIT forces the factor; default = the library's choice).

  python profiles/tools/sweep_wgrad_ksplit.py
"""
import os, sys, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    info = (ctypes.c_int32 * 4)()
    L.call("rv_tap_wgrad_info", ctypes.byref(g), ctypes.byref(wshape), info)
    packed = torch.empty((k * k, E.pad32(cout), E.pad32(cin)), dtype=torch.float32, device=dev)
    def run(): L.call("rv_tap_wgrad", ctypes.byref(g), ctypes.byref(wshape), dy.ptr(), L.i32(dy.ld), x.ptr(), L.i32(x.ld), None, None, L.i32(1), L.ptr(packed), L.ptr(ws), L.stream_ptr())
    return run, list(info)
def time(run, iters=10):
    for _ in range(2): run()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): run()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
SH = ((128, 128, 3, 2656), (128, 128, 3, 1328), (128, 128, 3, 1024), (128, 128, 3, 664), (128, 128, 3, 512), (128, 128, 3, 256), (128, 128, 3, 128),
      (256, 256, 3, 2656), (256, 256, 3, 2048), (512, 512, 3, 2048), (256, 256, 1, 2048), (128, 128, 1, 1024))
for (cin, cout, k, W) in SH:
    fl = 2.0 * 4 * 64 * W * k * k * cin * cout
    os.environ.pop("RV3D_WGRAD_KSPLIT", None)
    run, info = setup(cin, cout, k, 4, 64, W)
    base = min(time(run) for _ in range(3))
    line = [f"default ks={info[1]} grid={info[2]}: {base:7.1f} us {fl / base / 1e6:6.0f} TF/s |"]
    groups = k * ((k + 2) // 3) * max(1, cin // 128) * max(1, cout // 128)
    cands = sorted({max(1, r * 256 // groups) for r in (1, 2, 3, 4)} | {max(1, (r * 256 + groups - 1) // groups) for r in (1, 2)} | {max(1, 128 // groups), max(1, 192 // groups)})
    for ks in cands:
        os.environ["RV3D_WGRAD_KSPLIT"] = str(ks)
        run, info = setup(cin, cout, k, 4, 64, W)
        t = min(time(run) for _ in range(3))
        line.append(f"ks={info[1]}/g{info[2]}: {t:6.1f}")
    os.environ.pop("RV3D_WGRAD_KSPLIT", None)
    print(f"wgrad {cin:3d}<->{cout:3d} k{k} W{W:5d}  " + "  ".join(line), flush=True)
