"""Weighted NMS of sweeps beyond the old 16 384-candidate capacity: device-resident path vs the per-class loop over the FFI.

  python profiles/tools/nms_large_time.py
"""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd.math.ops import nms as hnms
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))), "tests"))
from test_gpu_model import _random_boxes
dev = "cuda:0"
for n_cls, k, spread in ((26, 6000, 200.0), (26, 60000, 600.0), (3, 60000, 600.0), (26, 212992, 1200.0)):
    cubs, scs, cats = [], [], []
    for b in range(4):
        cub, s = _random_boxes(k, 900 + b, spread)
        cubs.append(cub); scs.append(s)
        cats.append(torch.randint(0, n_cls, (k,), generator=torch.Generator().manual_seed(40 + b)))
    args = (torch.stack(cubs).to(dev), torch.stack(scs).to(dev), torch.stack(cats).to(dev), 50000, 1000, 0.3, 0.1, "weighted")
    def timed(fn, iters):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(iters): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / iters * 1e3, r
    ms, r = timed(lambda: hnms.batched_multiclass_nms(*args, n_classes=n_cls), 5)
    line = f"B=4 x {k} candidates, {n_cls} classes: rv_nms_sweeps {ms:.2f} ms per batch ({r[0].shape[0]} rows out)"
    if k <= 60000:
        old = hnms.FUSED_CLASSES_MAX; hnms.FUSED_CLASSES_MAX = 0
        try: ms2, r2 = timed(lambda: hnms.batched_multiclass_nms(*args, n_classes=n_cls), 2)
        finally: hnms.FUSED_CLASSES_MAX = old
        line += f"; per-class loop over the FFI {ms2:.1f} ms (rows identical: {all(torch.equal(a, b) for a, b in zip(r, r2))})"
    print(line, flush=True)
