"""Eval forward of the rv-av2 model, 4 sweeps: bf16 operands against fp16 operands (same process, alternating rounds).

  python profiles/tools/ab_eval_operand.py
"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import build_model, synthetic_batch
from range_view_3d_detection_amd import program
dev = torch.device("cuda:0")
torch.manual_seed(0)
backbone, head = build_model("rv-av2", 26)
backbone.to(dev).eval(); head.to(dev).eval()
batch = synthetic_batch(4, 64, 2048, seed=1, device=dev)
def fwd():
    with torch.no_grad():
        out, _ = head(backbone(batch), batch, return_loss=False)
    return out
res = {"bf16": [], "f16": []}
for rnd in range(5):
    for tag in ("bf16", "f16"):
        program.EVAL_OPERAND = tag
        for _ in range(3): fwd()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fwd()
        torch.cuda.synchronize(); res[tag].append((time.perf_counter() - t) / 10 * 1e3)
for tag, v in res.items():
    v = sorted(v); print(f"{tag}: median {v[len(v) // 2]:.2f} ms per batch of 4 (min {v[0]:.2f}, max {v[-1]:.2f})")
