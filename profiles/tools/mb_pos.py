"""Positional pair of the MetaKernel stem at the bench shape (P = 9 x 4 x 64 x 2048, C = 256): rv_pos_forward against the
rv_smallk_forward apply pass + the 1x1 tap-conv it replaces."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from range_view_3d_detection_amd import _lib as L, engine as E
dev = torch.device("cuda:0")
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
P, C = 9 * 4 * 64 * 2048, 256
rel = torch.zeros(P, 32, dtype=torch.bfloat16, device=dev); rel[:, :3] = torch.randn(P, 3, device=dev).bfloat16()
w1 = torch.zeros(C, 32, dtype=torch.bfloat16, device=dev); w1[:, :3] = torch.randn(C, 3, device=dev).bfloat16()
w2 = (torch.randn(C, C, device=dev) / 16).bfloat16()
s1, t1 = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
h1, y2 = torch.empty(P, C, dtype=torch.bfloat16, device=dev), torch.empty(P, C, dtype=torch.bfloat16, device=dev)
rows = L.load().rv_pos_forward_rows(L.i64(P))
partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, C), dtype=torch.float32, device=dev)
T = P * C * 2 / 1e9
t = bench(lambda: L.call("rv_pos_forward", L.ptr(rel), L.i32(32), L.i32(3), L.i64(P), L.ptr(w1), L.i32(32), L.ptr(s1), L.ptr(t1), L.ptr(w2), L.i32(C), L.ptr(h1), L.ptr(y2), L.ptr(partial), L.stream_ptr()))
print(f"rv_pos_forward            {t*1e3:7.1f} us   {2*T/t:5.2f} TB/s written   {2.0*P*C*C/t/1e9:6.0f} TFLOP/s")
t1_ = bench(lambda: L.call("rv_smallk_forward", L.ptr(rel), L.i32(32), L.i64(P), L.i32(3), L.ptr(w1), L.i32(32), L.i32(C), None, L.i64(P), None, None, L.f32(1e-5), L.f32(0.1), None, None, L.ptr(s1), L.ptr(t1), None, None, L.i32(1), L.ptr(h1), L.i32(C), L.stream_ptr()))
m = torch.nn.Conv2d(C, C, 1, bias=False).to(dev)
x = E.Act(h1.view(4, 64, 2048 * 9, C))
lay = E.tap_layer(m)
tp = E.Tape(True, dev)
def conv():
    E.ConvOp(tp, lay, x, stats=True); tp.ops.clear()
t2_ = bench(conv)
print(f"smallk apply + 1x1 conv   {t1_*1e3:7.1f} + {t2_*1e3:7.1f} = {(t1_+t2_)*1e3:7.1f} us")
# ---- backward: rv_pos_backward_sums against the 1x1 backward-data conv + rv_bn_bwd_smallk_sums it replaces
dy2 = torch.randn(P, C, device=dev).bfloat16()
mu, isd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
ws = torch.empty(L.load().rv_bn_bwd_smallk_workspace_bytes(L.i64(P), L.i32(C), L.i32(3)), dtype=torch.uint8, device=dev)
sums = torch.zeros(6 * C, dtype=torch.float64, device=dev); moms = torch.zeros(20, dtype=torch.float64, device=dev)
tb = bench(lambda: L.call("rv_pos_backward_sums", L.i64(P), L.i32(C), L.ptr(dy2), L.ptr(w2), L.ptr(rel), L.i32(32), L.i32(3), L.ptr(w1), L.i32(32), L.ptr(s1), L.ptr(t1),
                          L.ptr(mu), L.ptr(isd), L.ptr(sums), L.ptr(moms), L.ptr(ws), L.stream_ptr()))
print(f"rv_pos_backward_sums      {tb*1e3:7.1f} us   {T/tb:5.2f} TB/s read")
dh1 = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
g = lay.geom
shape = L.TapShape(4, 64, 2048 * 9, 2048 * 9, C, C, 0)
t3 = bench(lambda: L.call("rv_tap_scatter", ctypes.byref(g), ctypes.byref(shape), L.ptr(dy2), None, None, L.ptr(lay.packed("scatter")), None, L.ptr(dh1), None, L.stream_ptr()))
t4 = bench(lambda: L.call("rv_bn_bwd_smallk_sums", L.i64(P), L.i32(C), L.ptr(dh1), L.i32(C), None, L.i32(0), None, L.i32(0), L.ptr(s1), L.ptr(t1), L.ptr(mu), L.ptr(isd),
                          L.i32(L.BNB_RELU_Z | L.BNB_Y_FROM_INPUT), L.ptr(rel), L.i32(32), L.i32(3), L.ptr(w1), L.i32(32), L.ptr(sums), L.ptr(moms), L.ptr(ws), L.stream_ptr()))
print(f"1x1 dgrad + smallk sums   {t3*1e3:7.1f} + {t4*1e3:7.1f} = {(t3+t4)*1e3:7.1f} us")
