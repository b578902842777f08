"""In-process A/B of a training step over module attributes (the library has no environment switches for its fusions: tests and
A/B tools flip the module attribute).  One model, one batch, the variants interleaved (A B A B ...) so that box-to-box and
thermal drift cancel:

    python profiles/tools/ab_attr.py engine_bwd.HEAD_FINAL_FUSE=True,False [--widths rv-av2|rv-waymo] [--steps 10] [--rounds 3]
                                     [--set module.ATTR=value ...]   (attributes held fixed for every variant)
"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("spec", help="module.ATTR=v1,v2[,v3] (module relative to range_view_3d_detection_amd)")
    ap.add_argument("--widths", default="rv-av2")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--set", action="append", default=[])
    args = ap.parse_args()
    for fixed in args.set:
        ftarget, fvalue = fixed.split("=")
        fmod, fattr = ftarget.rsplit(".", 1)
        setattr(importlib.import_module("range_view_3d_detection_amd." + fmod), fattr, eval(fvalue))
    target, values = args.spec.split("=")
    modname, attr = target.rsplit(".", 1)
    mod = importlib.import_module("range_view_3d_detection_amd." + modname)
    values = [eval(v) for v in values.split(";" if ";" in values else ",")]  # (";" between values that contain commas)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

    torch.manual_seed(0)
    if args.widths == "rv-av2":
        backbone, head = bench.build_model("rv-av2", 26, 5)
        batch = bench.synthetic_batch(4, 64, 2048, seed=1234, device=dev)
    else:
        backbone, head = bench.build_model("rv-waymo", 3, 6)
        batch = bench.synthetic_batch(4, 64, 2656, seed=4321, device=dev, n_feat=6, n_cls=3)
    model = bench.Detector(backbone, head).to(dev).train()
    params = list(model.parameters())
    opt, sched = configure_optimizers(params, num_devices=1, batch_size=4, total_steps=10_000, fused=True, max_grad_norm=35.0)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model(batch)
        loss.backward()
        opt.step()
        sched.step()
        return loss

    for v in values:
        setattr(mod, attr, v)
        for _ in range(3):
            step()
    torch.cuda.synchronize()
    res = {repr(v): [] for v in values}
    for r in range(args.rounds):
        for v in values:
            setattr(mod, attr, v)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = step()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / args.steps
            res[repr(v)].append(ms)
            print(f"round {r} {target}={v!r}: {ms:.2f} ms per step (loss {float(loss):.5f})", flush=True)
    for k, v in res.items():
        print(f"{target}={k}: " + " / ".join(f"{x:.2f}" for x in v) + f"   mean {sum(v) / len(v):.2f} ms")


if __name__ == "__main__":
    main()
