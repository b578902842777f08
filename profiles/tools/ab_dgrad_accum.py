"""Backward-data launches (scatter form of a 3x3 conv) with and without RV_OUT_ACCUM, per kernel generation (in-process A/B).

  python profiles/tools/ab_dgrad_accum.py
"""
import os, sys, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import engine as E
from range_view_3d_detection_amd import _lib as L
dev = 'cuda:0'
def setup(cin, cout, N, H, W, accum):
    m = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False).to(dev)
    layer = E.tap_layer(m)
    dy = E.Act(torch.randn(N, H, W, cout, device=dev).to(torch.bfloat16))
    dst = E.Act(torch.randn(N, H, W, cin, device=dev).to(torch.bfloat16))
    shape = L.TapShape(N, H, W, W, dy.ld, dst.ld, L.OUT_ACCUM if accum else 0)
    w = layer.packed("scatter")
    def run(): L.call("rv_tap_scatter", ctypes.byref(layer.geom), ctypes.byref(shape), dy.ptr(), None, None, L.ptr(w), None, dst.ptr(), None, L.stream_ptr())
    return run
def time(run, iters=10):
    for _ in range(2): run()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): run()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (cin, cout, W) in ((128, 128, 1024), (256, 128, 2048), (256, 256, 2048), (512, 512, 2048)):
    fl = 2.0 * 4 * 64 * W * 9 * cin * cout
    row = []
    for env in ({}, {"RV3D_NO_TAPCONV6": "1"}):
        for accum in (False, True):
            os.environ.update(env)
            run = setup(cin, cout, 4, 64, W, accum)
            t = min(time(run) for _ in range(3))
            for k in env: del os.environ[k]
            row.append(f"{'t5' if env else 't6'}{'+acc' if accum else '    '} {t:7.1f} us {fl / t / 1e6:6.0f}")
    print(f"dgrad {cout}->{cin} W{W}: " + " | ".join(row), flush=True)
