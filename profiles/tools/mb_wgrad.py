"""wgrad3 alone on random bf16 operands (3x3 512 <-> 512 and 256 <-> 256 at 4 x 64 x 2048): us per launch (+ reduce) and TFLOP/s of whatever
library RV3D_LIB selects.  argv[1] = a label for the output line (profiles/tools/diag_wgrad.sh)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import _lib as L

label = sys.argv[1] if len(sys.argv) > 1 else "product"
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
lib = L.load()
for C in (512, 256):
    N, H, W = 4, 64, 2048
    g = L.TapGeom(3, 3, 1, 1, 1, C, C)
    s = L.TapShape(N, H, W, W, 0, 0, L.WGRAD_TORCH_LAYOUT)
    ws = torch.empty(lib.rv_tap_wgrad_workspace_bytes(ctypes.byref(g), ctypes.byref(s)), dtype=torch.uint8, device=dev)
    grad = torch.empty((C, C, 3, 3), dtype=torch.float32, device=dev)
    u = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    v = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    call = lambda: L.call("rv_tap_wgrad", ctypes.byref(g), ctypes.byref(s), L.ptr(u), L.i32(C), L.ptr(v), L.i32(C), None, None, L.i32(1), L.ptr(grad), L.ptr(ws), L.stream_ptr())
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100 if C == 512 else 300
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print(f"wgrad3 {label:14s} {C}<->{C} 3x3 4x64x2048: {us:8.1f} us per launch (+ reduce)  {2.0 * N * H * W * 9 * C * C / us / 1e6:7.1f} TFLOP/s", flush=True)
