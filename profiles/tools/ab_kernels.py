"""Per-kernel view of an in-process A/B over a module attribute: for every value, three training steps with the side stream OFF and events
around each tap-conv / weight-gradient launch (engine.KernelProfile; RV3D_PROFILE_SHAPES=1 for per-shape names) plus the HBM-group hook
of bench.py for the bandwidth-bound launches.  Prints ms per step per kernel for each value, side by side.

    RV3D_PROFILE_SHAPES=1 python profiles/tools/ab_kernels.py _lib.SELECT=0,16777216 [--widths rv-av2|rv-waymo] [--min-ms 0.05]
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("spec")
    ap.add_argument("--widths", default="rv-av2")
    ap.add_argument("--min-ms", type=float, default=0.05)
    args = ap.parse_args()
    target, values = args.spec.split("=")
    modname, attr = target.rsplit(".", 1)
    mod = importlib.import_module("range_view_3d_detection_amd." + modname)
    values = [eval(v) for v in values.split(",")]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from range_view_3d_detection_amd import _lib as L
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

    torch.manual_seed(0)
    if args.widths == "rv-av2":
        backbone, head = bench.build_model("rv-av2", 26, 5)
        batch = bench.synthetic_batch(4, 64, 2048, seed=1234, device=dev)
    else:
        backbone, head = bench.build_model("rv-waymo", 3, 6)
        batch = bench.synthetic_batch(4, 64, 2656, seed=4321, device=dev, n_feat=6, n_cls=3)
    model = bench.Detector(backbone, head).to(dev).train()
    opt, sched = configure_optimizers(list(model.parameters()), num_devices=1, batch_size=4, total_steps=10_000, fused=True, max_grad_norm=35.0)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model(batch)
        loss.backward()
        opt.step()
        sched.step()
        return loss

    E.OVERLAP_WGRAD = False
    tables = []
    for v in values:
        setattr(mod, attr, v)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        E.PROFILE = prof = E.KernelProfile()
        hbm = {}

        def hook(name, nbytes, launch, hbm=hbm):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            hbm.setdefault(name, []).append((e0, e1, nbytes))

        L.HBM_HOOK = hook
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        L.HBM_HOOK = None
        E.PROFILE = None
        tab = {k: (d["launches"] / 3, d["ms"] / 3, d["tflops"]) for k, d in prof.summary().items()}
        for name, evs in hbm.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in evs)
            gb = sum(n for _, _, n in evs) / 1e9
            tab["[hbm] " + name] = (len(evs) / 3, ms / 3, gb / ms if ms else 0.0)  # (third column: TB/s)
        tables.append(tab)
    keys = sorted(set().union(*tables), key=lambda k: -max(t.get(k, (0, 0, 0))[1] for t in tables))
    print(f"{'kernel':100s} " + " ".join(f"{target.split('.')[-1]}={v!r:>12}" for v in values))
    tot = [0.0] * len(values)
    for k in keys:
        row = [t.get(k, (0, 0.0, 0.0)) for t in tables]
        for i, r in enumerate(row):
            tot[i] += r[1]
        if max(r[1] for r in row) < args.min_ms:
            continue
        print(f"{k[:100]:100s} " + " ".join(f"{r[0]:5.1f}x {r[1]:7.3f} ms" for r in row))
    print(f"{'sum of timed launches':100s} " + " ".join(f"      {x:8.3f} ms" for x in tot))


if __name__ == "__main__":
    main()
