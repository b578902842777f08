"""MetaKernel gather kernels at the bench shape (4 x 64 x 2048, C = 256): time and algorithmic TB/s."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from range_view_3d_detection_amd import _lib as L
dev = torch.device("cuda:0")
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
N, H, W, C = 4, 64, 2048, 256
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
pos, dgeo, feat = bf(N, H, W, 9, C), bf(N, H, W, 9, C), bf(N, H, W, C)
geo, dy, dfeat = torch.empty_like(pos), torch.empty_like(pos), torch.empty_like(feat)
sc, sh, mu, isd = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
coef = torch.rand(3, C, device=dev)
rows = L.load().rv_meta_bwd_rows(L.i32(N), L.i32(H), L.i32(W))
partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, C), dtype=torch.float32, device=dev)
T = pos.numel() * 2 / 1e9
t = bench(lambda: L.call("rv_meta_modulate", L.ptr(pos), L.ptr(sc), L.ptr(sh), L.ptr(feat), L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(geo), L.stream_ptr()))
print(f"modulate fwd   {t*1e3:7.1f} us  {(2*T + T/9)/t:5.2f} TB/s")
t = bench(lambda: L.call("rv_meta_modulate_bwd_sums", L.ptr(dgeo), L.ptr(pos), L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(isd), L.ptr(feat), L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(dfeat), L.i32(C), L.ptr(partial), L.stream_ptr()))
print(f"bwd sums       {t*1e3:7.1f} us  {(2*T + 2*T/9)/t:5.2f} TB/s")
t = bench(lambda: L.call("rv_meta_modulate_bwd_apply", L.ptr(dgeo), L.ptr(pos), L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(isd), L.ptr(coef), L.ptr(feat), L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(dy), L.stream_ptr()))
print(f"bwd apply      {t*1e3:7.1f} us  {(3*T + T/9)/t:5.2f} TB/s")
