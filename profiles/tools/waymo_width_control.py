"""rv-waymo control experiment (round-5 review, item 2): the same model at W = 2560 (1280 tap-conv tiles = exactly 5.0 rounds of
256 persistent workgroups) and at W = 2656 (1328 tiles = 5.19 rounds), interleaved in ONE process, so that what the ragged last round
costs is measured instead of estimated.  Prints ms per step, ms per step per 1000 columns, and the per-kernel tables (one stream).

    python profiles/tools/waymo_width_control.py [--widths 2560,2656,2688,2816] [--steps 10] [--rounds 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--widths", default="2560,2656")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    widths = [int(w) for w in args.widths.split(",")]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

    torch.manual_seed(0)
    backbone, head = bench.build_model("rv-waymo", 3, 6)
    model = bench.Detector(backbone, head).to(dev).train()
    opt, sched = configure_optimizers(list(model.parameters()), num_devices=1, batch_size=4, total_steps=100_000, fused=True, max_grad_norm=35.0)
    batches = {w: bench.synthetic_batch(4, 64, w, seed=4321, device=dev, n_feat=6, n_cls=3) for w in widths}

    def step(w):
        opt.zero_grad(set_to_none=True)
        loss = model(batches[w])
        loss.backward()
        opt.step()
        sched.step()
        return loss

    for w in widths:
        for _ in range(3):
            step(w)
    torch.cuda.synchronize()
    res = {w: [] for w in widths}
    for r in range(args.rounds):
        for w in widths:
            step(w)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(w)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / args.steps
            res[w].append(ms)
            print(f"round {r} W={w}: {ms:.2f} ms per step", flush=True)
    base = None
    for w in widths:
        m = sum(res[w]) / len(res[w])
        per_col = m / w * 1000
        base = base or per_col
        print(f"W={w}: " + " / ".join(f"{x:.2f}" for x in res[w]) + f"   mean {m:.2f} ms = {4e3 / m:.2f} sweeps/s; {per_col:.3f} ms per 1000 columns "
              f"({100 * (per_col / base - 1):+.1f} % against W={widths[0]}); tiles of a 64 x W layer: {4 * 4 * ((w + 31) // 32)} = {4 * 4 * ((w + 31) // 32) / 256:.2f} rounds")
    # per-kernel, one stream, both widths
    overlap, E.OVERLAP_WGRAD = E.OVERLAP_WGRAD, False
    for w in widths:
        step(w)
        E.PROFILE = prof = E.KernelProfile()
        for _ in range(3):
            step(w)
        torch.cuda.synchronize()
        E.PROFILE = None
        print(f"--- W={w}, one stream, per kernel (ms per step)")
        for k, v in sorted(prof.summary().items(), key=lambda kv: -kv[1]["ms"]):
            print(f"  {k[:90]:90s} n {v['launches'] / 3:5.1f}  {v['ms'] / 3:7.3f} ms  {v['tflops']:7.1f} TF/s")
    E.OVERLAP_WGRAD = overlap


if __name__ == "__main__":
    main()
