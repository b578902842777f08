"""Inference stem at the bench shape (4 x 64 x 2048, C = 256 / 128): rv_pos_modulate_forward (positional pair + modulation, only
`geo` stored) against rv_pos_forward + rv_meta_modulate (h1, y2, geo stored; y2 read back)."""
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
N, H, W = 4, 64, 2048
for tag, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    for C in (256, 128):
        P = 9 * N * H * W
        rel = torch.zeros(P, 32, dtype=dt, device=dev); rel[:, :3] = torch.randn(P, 3, device=dev).to(dt)
        w1 = torch.zeros(C, 32, dtype=dt, device=dev); w1[:, :3] = torch.randn(C, 3, device=dev).to(dt)
        w2 = (torch.randn(C, C, device=dev) / 16).to(dt)
        s1, t1 = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
        s2, t2 = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
        feat = torch.randn(N * H * W, C, device=dev).to(dt)
        h1, y2 = torch.empty(P, C, dtype=dt, device=dev), torch.empty(P, C, dtype=dt, device=dev)
        geo = torch.empty(N * H * W, 9 * C, dtype=dt, device=dev)
        T = P * C * 2 / 1e9
        with L.operand(tag):
            ta = bench(lambda: L.call("rv_pos_forward", L.ptr(rel), L.i32(32), L.i32(3), L.i64(P), L.ptr(w1), L.i32(32), L.ptr(s1), L.ptr(t1), L.ptr(w2), L.i32(C),
                                      L.ptr(h1), L.ptr(y2), None, L.stream_ptr()))
            tb = bench(lambda: L.call("rv_meta_modulate", L.ptr(y2), L.ptr(s2), L.ptr(t2), L.ptr(feat), L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.i32(C), L.ptr(geo),
                                      L.stream_ptr()))
            tc = bench(lambda: L.call("rv_pos_modulate_forward", L.ptr(rel), L.i32(32), L.i32(3), L.ptr(w1), L.i32(32), L.ptr(s1), L.ptr(t1), L.ptr(w2), L.i32(C),
                                      L.ptr(s2), L.ptr(t2), L.ptr(feat), L.i32(C), L.i32(N), L.i32(H), L.i32(W), L.ptr(geo), L.stream_ptr()))
        print(f"{tag} C={C}: pos_forward {ta*1e3:7.1f} us + modulate {tb*1e3:7.1f} us = {(ta+tb)*1e3:7.1f}   fused {tc*1e3:7.1f} us ({T/tc:4.2f} TB/s written, "
              f"{2.0*P*C*C/tc/1e9:5.0f} TFLOP/s)", flush=True)
