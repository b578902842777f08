"""ew_combine (block output / materialise passes): achieved HBM bandwidth on the bench shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from range_view_3d_detection_amd import _lib as L
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (px, c, two) in [(4*64*2048, 512, False), (4*64*2048, 256, False), (4*64*2048, 256, True), (4*64*1024, 128, True), (4*64*512, 128, True)]:
    a = torch.randn(px, c, device=dev).bfloat16(); b = torch.randn(px, c, device=dev).bfloat16() if two else None
    out = torch.empty_like(a)
    sc = torch.rand(c, device=dev) + 0.5; sh = torch.randn(c, device=dev) * 0.1
    t = bench(lambda: L.call("rv_ew_combine", L.i64(px), L.i32(c), L.ptr(a), L.i32(c), L.ptr(sc), L.ptr(sh), L.ptr(b) if two else None, L.i32(c), L.ptr(sc) if two else None,
                              L.ptr(sh) if two else None, L.ptr(out), L.i32(c), L.i32(L.EW_RELU_A | (L.EW_RELU_OUT if two else 0)), L.stream_ptr()))
    nb = (3 if two else 2) * px * c * 2 / 1e9
    print(f"px {px} c {c} operands {2 if two else 1}: {t*1e3:7.1f} us {nb/t:6.2f} TB/s")
