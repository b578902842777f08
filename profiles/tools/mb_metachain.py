"""Microbenchmark of the chained MetaKernel backward (csrc/metachain.hip) at the rv-av2 stem shape against the unchained passes it
replaces (rv_tap_scatter GEMM + rv_meta_modulate_bwd_sums / _apply), each alone on the GPU:

    python profiles/tools/mb_metachain.py [path/to/librv3d_hip.so ...]   (extra libraries: ablation builds of the same entry points)
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import _lib as L

dev = "cuda:0"
N, H, W, C, K = (4, 64, 2048, 256, 256) if os.environ.get("MB_WAYMO") is None else (4, 64, 2656, 128, 128)
g = torch.Generator().manual_seed(0)
bf = lambda *s, k=1.0: (k * torch.randn(*s, generator=g)).to(torch.bfloat16).to(dev)
dz, y, feat, w = bf(N, H, W, K), bf(N, H, W, 9, C), bf(N, H, W, C), bf(9 * C, K, k=K ** -0.5)
scale, shift, mean, invstd = (torch.rand(C, generator=g).add(0.5).to(dev), torch.randn(C, generator=g).mul(0.3).to(dev),
                              torch.randn(C, generator=g).mul(0.2).to(dev), torch.rand(C, generator=g).add(0.5).to(dev))
coef = torch.rand(3, C, generator=g).to(dev)
dfeat, dy, dgeo = torch.empty_like(feat), torch.empty_like(y), torch.empty_like(y)
st = L.stream_ptr()


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def bench_lib(path):
    lib = ctypes.CDLL(path) if path else L.load()
    rows = lib.rv_meta_chain_rows(L.i32(N), L.i32(H), L.i32(W))
    partial = torch.empty((rows + 8, 2, C), dtype=torch.float32, device=dev)
    def sums():
        assert lib.rv_meta_chain_bwd_sums(L.ptr(dz), L.i32(K), L.i32(K), L.ptr(w), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(feat),
                                          L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(dfeat), L.i32(C), L.ptr(partial), st) == 0
    def apply():
        assert lib.rv_meta_chain_bwd_apply(L.ptr(dz), L.i32(K), L.i32(K), L.ptr(w), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(coef),
                                           L.ptr(feat), L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(dy), st) == 0
    print(f"{os.path.basename(path) if path else 'library'}: chain sums {timed(sums):8.1f} us   chain apply {timed(apply):8.1f} us", flush=True)


bench_lib(None)
for p in sys.argv[1:]:
    bench_lib(p)
# the unchained passes
lib = L.load()
rows = lib.rv_meta_bwd_rows(L.i32(N), L.i32(H), L.i32(W))
partial = torch.empty((rows + 8, 2, C), dtype=torch.float32, device=dev)
def usums():
    L.call("rv_meta_modulate_bwd_sums", L.ptr(dgeo), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(feat), L.i32(C),
           L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(dfeat), L.i32(C), L.ptr(partial), st)
def uapply():
    L.call("rv_meta_modulate_bwd_apply", L.ptr(dgeo), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(coef), L.ptr(feat),
           L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(dy), st)
geom = L.TapGeom(1, 1, 1, 0, 0, K, 9 * C)
shape = L.TapShape(N, H, W, W, K, 9 * C, 0)
def gemm():
    L.call("rv_tap_scatter", ctypes.byref(geom), ctypes.byref(shape), L.ptr(dz), None, None, L.ptr(w), None, L.ptr(dgeo), None, st)
print(f"unchained: backward-data GEMM {timed(gemm):8.1f} us   sums {timed(usums):8.1f} us   apply {timed(uapply):8.1f} us")
