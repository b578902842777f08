"""Is wgrad3 pipeline-bound or POWER-bound?  (round-4 review, item 6: what r03_power_cap.md did for the tap-conv, for the weight
gradient.)  wgrad3 on the 3x3 512 <-> 512 layer at 4 x 64 x 2048 (2.47 TFLOP per launch, 256 workgroups: the balanced split) in a
loop while rocm-smi is sampled: random bf16 operands and all-zero operands on all 256 CUs.  Fewer CUs: run the same script under
`HSA_CU_MASK=0:0-127` (a process-wide mask of the ROCr runtime; a per-stream mask through hipExtStreamCreateWithCUMask did not
restrict anything on this stack -- round 5, the 128 / 64 rows of the first run took the 256-CU time).

  python profiles/tools/power_probe_wgrad.py
  HSA_CU_MASK=0:0-127 python profiles/tools/power_probe_wgrad.py 128
"""
import ctypes, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from range_view_3d_detection_amd import _lib as L

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
N, H, W, C = 4, 64, 2048, 512
g = L.TapGeom(3, 3, 1, 1, 1, C, C)
s = L.TapShape(N, H, W, W, 0, 0, L.WGRAD_TORCH_LAYOUT)
lib = L.load()
ws = torch.empty(lib.rv_tap_wgrad_workspace_bytes(ctypes.byref(g), ctypes.byref(s)), dtype=torch.uint8, device=dev)
grad = torch.empty((C, C, 3, 3), dtype=torch.float32, device=dev)
info = (ctypes.c_int32 * 4)()
lib.rv_tap_wgrad_info(ctypes.byref(g), ctypes.byref(s), info)
print(f"kernel generation {info[0]}, split-K slabs {info[1]}, workgroups {info[2]}")
fl = 2.0 * N * H * W * 9 * C * C


def loop(u, v, stream, secs):
    t0 = time.time(); n = 0
    with torch.cuda.stream(stream):
        while time.time() - t0 < secs:
            for _ in range(10):
                L.call("rv_tap_wgrad", ctypes.byref(g), ctypes.byref(s), L.ptr(u), L.i32(C), L.ptr(v), L.i32(C), None, None, L.i32(1), L.ptr(grad), L.ptr(ws),
                       L.stream_ptr())
            stream.synchronize(); n += 10
    return (time.time() - t0) / n * 1e6


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    get = lambda key: next((l.split(":")[-1].strip() for l in out.splitlines() if key in l), "?")
    return get("sclk clock level"), get("Current Socket Graphics Package Power"), get("Max Graphics Package Power")


u = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
v = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
uz, vz = torch.zeros_like(u), torch.zeros_like(v)
print("| data | CUs | us per launch (+ reduce) | TFLOP/s | TFLOP/s scaled to 256 CUs | sclk | package power (W) | cap (W) |\n|---|---|---|---|---|---|---|---|")
cus = int(sys.argv[1]) if len(sys.argv) > 1 else 256  # (what HSA_CU_MASK leaves: for the "scaled" column only)
print(f"HSA_CU_MASK={os.environ.get('HSA_CU_MASK')}")
for tag, a, b in (("random", u, v), ("zeros", uz, vz)):
    st = torch.cuda.Stream()
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("us", loop(a, b, st, 6.0)))
    th.start(); time.sleep(3.0); clk, pw, cap = smi(); th.join()
    tf = fl / res["us"] / 1e6
    print(f"| {tag} | {cus} | {res['us']:.0f} | {tf:.0f} | {tf * 256 / cus:.0f} | {clk} | {pw} | {cap} |", flush=True)
