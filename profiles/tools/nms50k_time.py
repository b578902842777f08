"""Weighted NMS of ONE class at num_pre_nms = 50 000 (conf/model/range_view.yaml:44) -- wall time of `weighted_nms`
(sort + rv_wnms: sin/cos, pair masks, blocked scan, merge; one device->host read) for a sparse and a crowded scene.

  python profiles/tools/nms50k_time.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from range_view_3d_detection_amd.math.ops import nms as hnms
from test_gpu_model import _random_boxes

for n, spread in ((6000, 120.0), (50_000, 400.0), (50_000, 100.0)):
    cub, scores = _random_boxes(n, 2, spread)
    half = cub[:, 3:5] / 2
    rect = torch.cat([cub[:, :2] - half, cub[:, :2] + half, cub[:, 6:7]], dim=-1).cuda()
    data = torch.cat([cub[:, :6], cub[:, 6:7].sin(), cub[:, 6:7].cos()], dim=1).cuda()
    s = scores.cuda()
    for _ in range(2):
        k, o, c = hnms.weighted_nms(rect, data, s, 0.3, 0.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        k, o, c = hnms.weighted_nms(rect, data, s, 0.3, 0.5)
    torch.cuda.synchronize()
    print(f"weighted_nms n={n} spread={spread:.0f} m: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call, {k.numel()} kept, largest cluster {int(c.max())}", flush=True)
