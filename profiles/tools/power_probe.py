"""Is the dominant kernel pipeline-bound or POWER-bound?  tapconv5 (3x3 512 -> 512, 4 x 64 x 2048) in a loop while rocm-smi is
sampled: random bf16 data on 256 / 128 / 64 CUs (rv_set_option "tapconv5_persist_blocks") and all-zero data on 256 CUs.

  python profiles/tools/power_probe.py
"""
import sys, os, ctypes, subprocess, threading, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import engine as E, _lib as L
dev = 'cuda:0'
m = torch.nn.Conv2d(512, 512, 3, padding=1, bias=False).to(dev)
x = E.Act(torch.randn(4, 64, 2048, 512, device=dev).to(torch.bfloat16))
xz = E.Act(torch.zeros(4, 64, 2048, 512, device=dev).to(torch.bfloat16))
layer = E.tap_layer(m)
fl = 2.0 * 4 * 64 * 2048 * 9 * 512 * 512
def loop(xx, secs):
    t = E.Tape(True, dev); t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20): E.ConvOp(t, layer, xx, stats=True); t.ops.clear()
        torch.cuda.synchronize(); n += 20
    return (time.time() - t0) / n * 1e6
def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    g = lambda key: next((l.split(":")[-1].strip() for l in out.splitlines() if key in l), "?")
    return g("sclk clock level"), g("Current Socket Graphics Package Power"), g("Max Graphics Package Power")
print("| data | CUs | us per launch | TFLOP/s | TFLOP/s scaled to 256 CUs | sclk | package power (W) | cap (W) |\n|---|---|---|---|---|---|---|---|")
for tag, xx, p in (("random", x, 256), ("random", x, 128), ("random", x, 64), ("zeros", xz, 256)):
    L.load().rv_set_option(b"tapconv5_persist_blocks", ctypes.c_int32(p))
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("us", loop(xx, 6.0)))
    th.start(); time.sleep(3.0); clk, pw, cap = smi(); th.join()
    tf = fl / res["us"] / 1e6
    print(f"| {tag} | {p} | {res['us']:.0f} | {tf:.0f} | {tf * 256 / p:.0f} | {clk} | {pw} | {cap} |", flush=True)
L.load().rv_set_option(b"tapconv5_persist_blocks", ctypes.c_int32(256))
