"""Inference path (tools/benchmark.py analogue): eval forward + decode + weighted NMS, per-batch latency."""
import math, sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch
from bench import build_model, synthetic_batch
from range_view_3d_detection_amd.nn.decoders.range_decoder import RangeDecoder
dev = torch.device('cuda:0')
torch.manual_seed(0)
backbone, head = build_model("rv-av2", 26)
backbone.to(dev).eval(); head.to(dev).eval()
# classification bias -4.6 (init) => few candidates above 0.1; raise some to exercise NMS with ~thousands of boxes
dec = RangeDecoder(True, True, [0, 15, 30], [15, 30, math.inf], [8, 2, 1])
post = {"num_pre_nms": 50000, "num_post_nms": 1000, "nms_threshold": 0.3, "min_confidence": 0.1, "nms_mode": "WEIGHTED"}
tasks = {0: [f"C{i}" for i in range(26)]}
for B in (1, 4):
    batch = synthetic_batch(B, 64, 2048, seed=1, device=dev)
    def fwd():
        with torch.no_grad():
            feats = backbone(batch)
            out, _ = head(feats, batch, return_loss=False)
        return out
    def full():
        out = fwd()
        with torch.no_grad():
            o = out[1][0]
            g = torch.Generator(device=dev).manual_seed(0)
            bump = (torch.rand(o["logits"].shape[0], 1, *o["logits"].shape[2:], device=dev, generator=g) < 0.03).float()
            o["logits"] = o["logits"] + 3.0 * bump  # random-init model: lift 3 % of the pixels over the confidence threshold
            return dec.decode(out, post, tasks, use_nms=True)
    from range_view_3d_detection_amd.math.ops import nms as hnms
    def full_loop():
        old = hnms.FUSED_CLASSES_MAX; hnms.FUSED_CLASSES_MAX = 0
        try: return full()
        finally: hnms.FUSED_CLASSES_MAX = old
    for name, fn in (("forward", fwd), ("forward+decode+NMS (device-resident batch path)", full), ("forward+decode+NMS (per-class loop)", full_loop)):
        for _ in range(3): r = fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): r = fn()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 10 * 1e3
        extra = f" ({r[0].shape[0]} boxes out)" if name != "forward" else ""
        print(f"B={B} {name}: {ms:.1f} ms/batch, {B / ms * 1e3:.1f} sweeps/s{extra}")

# B = 1 is launch-bound (~250 C-ABI launches per forward): the same forward captured once into a HIP graph and replayed
try:
    batch = synthetic_batch(1, 64, 2048, seed=1, device=dev)
    def fwd1():
        with torch.no_grad():
            feats = backbone(batch)
            out, _ = head(feats, batch, return_loss=False)
        return out[1][0]["logits"], out[1][0]["regressands"]
    for _ in range(3): fwd1()
    torch.cuda.synchronize()
    ref = [t.clone() for t in fwd1()]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fwd1()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        outs = fwd1()
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(outs, ref))
    t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 20 * 1e3
    print(f"B=1 forward as a HIP graph replay: {ms:.2f} ms/batch, {1 / ms * 1e3:.1f} sweeps/s (outputs identical to the eager run: {same})")
except Exception as e:  # noqa: BLE001
    print("HIP graph capture of the eval forward failed:", type(e).__name__, str(e)[:300])
