"""Weight-gradient launches on a second stream while the compute stream is busy: the split-K slabs of a SHORT launch (64 workgroups:
the 1x1 128 <-> 128 projection layer at 4 x 64 x 512) must be complete when the reduction that follows on the same stream sums
them.  Round 4 found one wrong weight gradient of exactly this layer in ~2500 training steps of the two-stream schedule (the
reduction had read a slab region that still held the previous tenant of the workspace memory: profiles/r04_ab_notes.md, "A wrong
step"); since then the weight-gradient kernels end with an explicit agent-scope release of their slabs (csrc/wgrad.hip).  This test
recreates the neighbourhood: a big weight gradient, then the small one, workspaces recycled through the caching allocator, on a
high-priority side stream, beside a tap-conv loop on the compute stream -- every result must equal the quiet one, bit for bit.
(A guard, not a reproduction: 150 rounds pass without the fence too; the failure needed ~2500 whole training steps,
profiles/tools/race_hunt3.py.)  Round 6 found what is most likely the real cause of that wrong gradient -- an address race in wgrad3's
epilogue -- as a FAULT of rv-waymo's stem; the second test below is its neighbourhood."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _wgrad_setup(cin, cout, k, N, H, W, seed):
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E

    g = torch.Generator().manual_seed(seed)
    m = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False).to(DEV)
    layer = E.tap_layer(m)
    geom = layer.geom
    x = E.Act(torch.randn(N, H, W, cin, generator=g).to(DEV).to(torch.bfloat16))
    dy = E.Act(torch.randn(N, H, W, cout, generator=g).to(DEV).to(torch.bfloat16))
    shape = L.TapShape(N, H, W, W, 0, 0, L.WGRAD_TORCH_LAYOUT)
    nbytes = L.load().rv_tap_wgrad_workspace_bytes(ctypes.byref(geom), ctypes.byref(shape))

    def run():
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)  # (allocated and dropped per launch, as the engine does)
        out = torch.empty((cout, cin, k, k), dtype=torch.float32, device=DEV)
        L.call("rv_tap_wgrad", ctypes.byref(geom), ctypes.byref(shape), dy.ptr(), L.i32(dy.ld), x.ptr(), L.i32(x.ld), None, None, L.i32(1),
               L.ptr(out), L.ptr(ws), L.stream_ptr())
        return out

    return run, (m, x, dy)


def test_short_weight_gradient_beside_a_busy_compute_stream_is_exact():
    from range_view_3d_detection_amd import engine as E

    big, keep_big = _wgrad_setup(256, 256, 3, 4, 64, 1024, 1)
    small, keep_small = _wgrad_setup(128, 128, 1, 4, 64, 512, 2)
    ref = small().clone()
    torch.cuda.synchronize()
    # the compute stream's load: a 256 -> 256 3x3 tap-conv, over and over
    conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(DEV)
    layer = E.tap_layer(conv)
    xin = E.Act(torch.randn(4, 64, 2048, 256, device=DEV).to(torch.bfloat16))
    tape = E.Tape(True, DEV)
    side = torch.cuda.Stream(device=DEV, priority=-1)
    outs = []
    rounds = 150
    for r in range(rounds):
        for _ in range(2):
            E.ConvOp(tape, layer, xin, stats=True)
            tape.ops.clear()
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            big()
            outs.append(small())
    torch.cuda.synchronize()
    wrong = [i for i, o in enumerate(outs) if not torch.equal(o, ref)]
    assert not wrong, f"{len(wrong)} of {rounds} weight gradients differ from the quiet result (first at round {wrong[0]})"


def test_stem_weight_gradient_beside_the_batchnorm_backward_passes_is_exact():
    """Round 6's neighbourhood: the one-tap instance of wgrad3 (1x1 128 <-> 128 at 4 x 64 x 2656: one tile, 256 K slices) on the high-priority side
    stream WHILE the compute stream runs the BatchNorm-backward reduce + apply passes over a tensor of the same size -- what rv-waymo's stem does
    when the second fusion conv's operand is written out.  The kernel's last LDS prefetch used to be in flight when the epilogue formed its first
    slab address in the prefetch's destination registers; with the LDS kept busy by the neighbour the read came back late and the stores went to
    address 0 (a fault within 100 training steps; with the other registers of that arithmetic: misplaced slab lines, round 4's wrong gradient).
    Fixed in csrc/wgrad.hip (the registers are operands of the wait; tests/test_host_cpu.py checks the machine code); this is the guard on the
    behaviour: every result equals the quiet one, bit for bit."""
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E

    N, H, W, C = 4, 64, 2656, 128
    wgrad, keep = _wgrad_setup(C, C, 1, N, H, W, 3)
    ref = wgrad().clone()
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(4)
    dout = E.Act(torch.randn(N, H, W, C, generator=g).to(DEV).to(torch.bfloat16))
    raw = E.Act(torch.randn(N, H, W, C, generator=g).to(DEV).to(torch.bfloat16))
    dy = raw.like()
    scale, shift, mean = (torch.randn(C, generator=g).to(DEV) for _ in range(3))
    invstd = torch.rand(C, generator=g).to(DEV) + 0.5
    coef = torch.randn(3, C, generator=g).to(DEV)
    pixels = raw.pixels
    rows = L.load().rv_bn_bwd_rows(L.i64(pixels))
    partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, C), dtype=torch.float32, device=DEV)
    common = (L.i64(pixels), L.i32(C), dout.ptr(), L.i32(dout.ld), None, L.i32(0), raw.ptr(), L.i32(raw.ld), L.ptr(scale), L.ptr(shift), L.ptr(mean),
              L.ptr(invstd))
    side = torch.cuda.Stream(device=DEV, priority=-1)
    outs = []
    rounds = 300
    for r in range(rounds):
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            outs.append(wgrad())
        L.call("rv_bn_bwd_reduce", *common, L.i32(L.BNB_RELU_Z), L.ptr(partial), L.stream_ptr())
        L.call("rv_bn_bwd_apply", *common, L.ptr(coef), L.i32(L.BNB_RELU_Z), dy.ptr(), L.i32(dy.ld), None, L.i32(0), L.stream_ptr())
    torch.cuda.synchronize()
    wrong = [i for i, o in enumerate(outs) if not torch.equal(o, ref)]
    assert not wrong, f"{len(wrong)} of {rounds} weight gradients differ from the quiet result (first at round {wrong[0]})"
