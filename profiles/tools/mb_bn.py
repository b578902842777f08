"""BN-backward kernels: achieved HBM bandwidth on the bench shapes."""
import sys, torch, ctypes
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
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
for (px, c, with_out, with_res) in [(4*64*2048, 512, False, False), (4*64*2048, 256, False, False), (4*64*2048, 256, True, True), (4*64*18432, 256, False, False), (4*64*1024, 128, False, False)]:
    dout = torch.randn(px, c, device=dev).bfloat16(); y = torch.randn(px, c, device=dev).bfloat16()
    out = torch.randn(px, c, device=dev).bfloat16() if with_out else None
    sc = torch.rand(c, device=dev) + 0.5; sh = torch.randn(c, device=dev) * 0.1; mu = torch.randn(c, device=dev) * 0.1; isd = torch.rand(c, device=dev) + 0.5
    rows = (px + 511) // 512
    partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, c), dtype=torch.float32, device=dev)
    coef = torch.rand(3, c, device=dev)
    dy = torch.empty_like(dout); dres = torch.zeros_like(dout) if with_res else None
    common = [L.i64(px), L.i32(c), L.ptr(dout), L.i32(c), L.ptr(out) if out is not None else None, L.i32(c), L.ptr(y), L.i32(c), L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(isd)]
    for extra, label in ((0, ""),):
        flags = L.BNB_RELU_Z | extra
        t_r = bench(lambda: L.call("rv_bn_bwd_reduce", *common, L.i32(flags), L.ptr(partial), L.stream_ptr()))
        t_a = bench(lambda: L.call("rv_bn_bwd_apply", *common, L.ptr(coef), L.i32(flags | (L.BNB_RES_ACCUM if with_res else 0)), L.ptr(dy), L.i32(c), L.ptr(dres) if with_res else None, L.i32(c), L.stream_ptr()))
        tb = px * c * 2 / 1e9
        n_r = 2 + (1 if with_out else 0); n_a = n_r + 1 + (2 if with_res else 0)
        print(f"{label:9s} px {px} c {c} out {with_out} res {with_res}: reduce {t_r*1e3:7.1f} us {n_r*tb/t_r:6.2f} TB/s | apply {t_a*1e3:7.1f} us {n_a*tb/t_a:6.2f} TB/s")
