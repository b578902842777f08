#!/usr/bin/env python3
"""Headline benchmark: training-step throughput (sweeps/s, fwd+bwd) of the rv-av2 range-view detector.

    python bench.py --gpus N --steps 10 --warmup 3        (N > 1: this process starts the N ranks itself, `self_launch`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W          (already a rank: WORLD_SIZE must equal --gpus)

One "step" = one pass of the hot path over one batch of synthetic sweeps per GPU: MetaKernel stem +
DLA backbone + cls/reg towers (forward), device target assignment + varifocal/L1 loss, backward
through every layer, gradient clipping (35.0) and the AdamW update -- the work of one
``Detector.training_step`` (nn/arch/detector.py:238-247) with ``conf/experiment/rv-av2.yaml``.
Inputs are resident in HBM before the timed region.  N > 1: one process per GPU, sweeps sharded
across ranks (weak scaling, 4 sweeps per GPU), gradient all-reduce over RCCL per finished program node
(engine.GradSync: the towers' gradients travel under the backbone's backward; RV3D_DDP=1: torch DDP),
BatchNorm statistics all-reduced (SyncBN as in conf/trainer/train.yaml:15).

Rank 0 prints ONE JSON line (contract in the task statement) including
  "roofline":     the dominant kernel (tapconv6: 512-pixel x 128-channel-tile bf16 MFMA tap-conv with the input halo resident in LDS)
                  against the dense bf16 MFMA peak.  `achieved` / `frac` / `avg_launch_us`: HIP events around each of its launches INSIDE
                  the timed region, on the stream it runs on -- the configuration that produced ms_per_step, where a backward-data launch
                  is often queued behind a weight gradient of the second stream; `isolated`: the same events in two extra steps with the
                  side stream off (the kernel's own duration); `traffic` = its HBM-side bytes per launch from this round's PMC passes
                  (profiles/), `algorithmic_bytes` = operands once in + result once out per launch;
  "roofline_hbm": the HBM-bound kernel group (BatchNorm backward, element-wise block passes, MetaKernel stem, head-final passes) against
                  8 TB/s: events around each of its launches in the timed configuration (`isolated`: side stream off), algorithmic bytes
                  from the C-ABI arguments, HBM-side traffic from the same PMC passes;
  "whole_step":   3 x the model's forward FLOPs over the step time, against the same peak; `traffic_ratio` =
                  HBM-side bytes of ALL kernels per step (PMC passes) over the algorithmic bytes of the step;
  "loss_first_step": the loss of the seed-0 model on sweep 0 of the seed-1234 batch before any update (one reproducible number;
                  tests/test_gpu_fullsize_train.py compares it with the fp32 oracle);
  "kernels" / "kernels_isolated": per-kernel event timings inside the timed region / with nothing else on the GPU;
  "cpu_baseline": the oracle (CPU restatement of the reference, ``oracle/``) timed on this box's host
                  cores on a bounded sample (N == 1 only).
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

AV2_CLASSES = 26


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; N > 1 without WORLD_SIZE in the environment: this process "
                                                         "starts the N ranks itself (torch.distributed.run as a child process)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="sweeps per GPU")
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--widths", default="rv-av2", help="rv-av2 (the metric's configuration), rv-waymo, or c<int> debug widths")
    ap.add_argument("--features", type=int, default=5, help="input channels: 5 (AV2) or 6 (Waymo)")
    ap.add_argument("--classes", type=int, default=AV2_CLASSES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["auto", "sample", "full"], default="auto",
                    help="auto: the whole 64x2048 sweep when the host is fast enough (one or two timed iterations), else the cropped sample; "
                         "sample: a 64xW crop (~45 s); full: 1 warm-up + 2 timed iterations of the whole sweep (minutes)")
    ap.add_argument("--no-extra", action="store_true", help="skip the forward_only and rv_waymo legs (extra keys of the JSON line)")
    ap.add_argument("--no-sync-bn", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="warm-up + timed steps and nothing else (no isolated / HBM-group passes, no extra legs, "
                                                              "no CPU baseline): the command the profiler passes of profiles/tools/collect_round.sh run")
    return ap.parse_args(argv)


def self_launch(args, argv) -> int | None:
    """``python bench.py --gpus N`` from ONE command, as the reference starts its N ranks from one (`scripts/train.sh:16-21`,
    Lightning's `devices` + DDP strategy, `conf/trainer/train.yaml:39-44`).  With N > 1 and no ``WORLD_SIZE`` in the environment
    this process is only the launcher: it starts ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py ...`` as a
    CHILD process (never exec: a process that has initialised the GPU must not be replaced, and this one has not even imported
    torch yet), passes rank 0's single JSON line through on stdout, the ranks' stderr on stderr, and returns the child's exit
    code.  Returns None when this process is itself a rank (or N == 1).  A ``--gpus`` that disagrees with ``WORLD_SIZE`` is an
    error: a record labelled N GPUs must have been produced by N ranks."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus and os.environ.get("RV3D_FORCE_DIST") is None:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with --nproc-per-node {args.gpus} "
                  f"(or drop WORLD_SIZE and let bench.py start the ranks)", file=sys.stderr)
            return 2
        return None
    if args.gpus <= 1:
        return None
    import socket
    import subprocess

    assert "torch" not in sys.modules, "the launcher must start the ranks before anything can initialise HIP"
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench launcher] {args.gpus} ranks: {' '.join(cmd)} (torch imported in the launcher: {'torch' in sys.modules})", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for ln in child.stdout:  # rank 0's JSON line (anything else a rank prints on stdout goes to stderr)
        if ln.startswith("{"):
            lines.append(ln)
        else:
            sys.stderr.write(ln)
    rc = child.wait()
    for ln in lines:
        sys.stdout.write(ln)
    sys.stdout.flush()
    return rc


if __name__ == "__main__":
    _ARGS = parse_args()
    _rc = self_launch(_ARGS, sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

if os.environ.get("RV3D_DIRECT_RCCL") is None:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime starts (see range_view_3d_detection_amd/__init__.py)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PMC_TRAFFIC = "profiles/r06_pmc_traffic.json"  # HBM bytes per launch per kernel, collected over this same command this round
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (MI355X_MICROARCH.md: ~2.5 PF dense; 2:1-sparsity figures are not used)
HBM_PEAK_GBS = 8000.0

# forward FLOPs per sweep (forward hooks on the reference's own modules); fwd + bwd = 3x
FWD_TFLOP_PER_SWEEP = {("rv-av2", 2048): 7.736, ("rv-waymo", 2656): 2.905}
# algorithmic activation bytes per sweep, forward, bf16 (each conv reads its input once and writes its output once)
FWD_ALG_GB_PER_SWEEP = {("rv-av2", 2048): 8.40, ("rv-waymo", 2656): 5.97}


def synthetic_batch(B: int, H: int, W: int, seed: int, device, n_feat: int = 5, boxes_per_sweep: int = 16, n_cls: int = AV2_CLASSES):
    """Synthetic sweeps as SURVEY.md §8d defines them (seeded; built on CPU, then moved to HBM)."""
    g = torch.Generator().manual_seed(seed)
    inc = torch.linspace(0.2, -0.4, H).view(1, 1, H, 1)
    az = torch.linspace(math.pi, -math.pi, W).view(1, 1, 1, W)
    # piecewise-smooth ranges (so that boxes contain several pixels) + noise, 10 % dropped returns
    r = 20.0 + 15.0 * torch.sin(3 * az + 1.3 * torch.rand(B, 1, 1, 1, generator=g)) + 10.0 * torch.cos(7 * inc) + torch.rand(B, 1, H, W, generator=g)
    r = r.clamp(1.5, 80.0)
    mask = torch.rand(B, 1, H, W, generator=g) >= 0.1
    cart = torch.cat([r * inc.cos() * az.cos(), r * inc.cos() * az.sin(), r * inc.sin().expand(B, 1, H, W)], dim=1) * mask
    intensity = torch.rand(B, 1, H, W, generator=g)
    feats = [intensity, r, cart[:, 0:1], cart[:, 1:2], cart[:, 2:3]]
    if n_feat == 6:
        feats = [torch.rand(B, 1, H, W, generator=g)] + feats
    features = (torch.cat(feats, dim=1) * mask).float()
    rows = []
    for b in range(B):
        valid = mask[b, 0].nonzero()
        pick = valid[torch.randperm(valid.shape[0], generator=g)[:boxes_per_sweep]]
        for h, w in pick.tolist():
            ctr = cart[b, :, h, w].double().tolist()
            lwh = (torch.tensor([1.0, 1.0, 1.0]) + torch.rand(3, generator=g) * torch.tensor([5.0, 2.0, 2.0])).tolist()
            yaw = (torch.rand(1, generator=g).item() * 2 - 1) * math.pi
            cat = int(torch.randint(0, n_cls, (1,), generator=g).item())
            rows.append(ctr + lwh + [math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)] + [0.0, float(cat), float(b)])
    ann = torch.tensor(rows, dtype=torch.float64)
    return {"features": features.to(device), "cart": cart.float().to(device), "mask": mask.to(device), "annotations": ann}


def build_model(widths: str, n_cls: int, in_channels: int = 5):
    from range_view_3d_detection_amd.nn.backbones.dla import RangeNet
    from range_view_3d_detection_amd.nn.heads.detection_head import DetectionHead

    if widths == "rv-av2":
        layers, head_c = [256, 128, 128, 128, 128], 512
    elif widths == "rv-waymo":
        layers, head_c = [128] * 5, 256
    else:  # debug widths "c<int>"
        c = int(widths[1:])
        layers, head_c = [c] * 5, 2 * c
    backbone = RangeNet(in_channels=in_channels, layers=layers, out_channels=layers[0], projection_kernel_size=1, dataset_name="av2",
                        num_neighbors=3, num_layers=2, stem_type="META",
                        _net={"_target_": "torchbox3d.nn.backbones.dla.RangeBackbone", "in_channels": in_channels, "layers": layers,
                              "out_channels": layers[0]})
    tasks = {0: [f"C{i}" for i in range(n_cls)]}
    tcfg = {"dataset_name": "av2", "tasks": tasks, "enable_azimuth_invariant_targets": True, "range_partitions": {1: [0.0, math.inf]},
            "fpn_assignment_method": None, "k": math.inf, "affinity_fn": "GAUSSIAN", "normalize_affinities": False, "sigma": 0.75}
    head = DetectionHead(fpn={1: 2 * layers[0]}, fpn_kernel_sizes={1: [3, 3]}, targets_config=tcfg, num_classification_blocks=4,
                         num_regression_blocks=4, final_kernel_size=1, tasks_cfg=tasks, task_in_channels=layers[0],
                         classification_weight=1.0, regression_weight=1.0, coding_weights=[1.0] * 8,
                         classification_head_channels=head_c, regression_head_channels=head_c,
                         classification_normalization_method="FOREGROUND",
                         _cls_loss={"_target_": "torchbox3d.nn.losses.classification.VarifocalLoss", "alpha": 0.75, "gamma": 2.0, "reduction": "none"},
                         _regression_loss={"_target_": "torch.nn.L1Loss", "reduction": "none"})
    return backbone, head


class Detector(torch.nn.Module):
    """``Detector.forward`` (nn/arch/detector.py:196-210): backbone -> head(return_loss) -> loss."""

    def __init__(self, backbone, head) -> None:
        super().__init__()
        self.backbone, self.head = backbone, head

    def forward(self, data):
        feats = self.backbone(data)
        _, losses = self.head(feats, data, return_loss=True)
        return losses["loss"]


def _progress(msg: str) -> None:
    """Progress lines on stderr (stdout carries the one JSON line): a long silent run looks hung to whoever launched it."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(seconds_budget: float = 45.0, full: bool = False, auto_full_limit: float = 0.0):
    """Oracle (``oracle/`` = CPU restatement of the reference, fp32 PyTorch-CPU) fwd+loss+bwd, B=1, rv-av2 widths
    (1 warm-up + 2 timed iterations, thread count and CPU stated).

    ``full``: the whole 64 x 2048 x 5 sweep (minutes of CPU time: ``--cpu-baseline full``).  ``sample``: the widest 64 x W crop
    of the same sweep whose warm-up + two timed iterations fit ``seconds_budget`` on the cores this box grants, scaled to
    sweeps/s by W / 2048 (the path is convolutional: cost is linear in W).  ``auto_full_limit`` > 0 (the default mode): the
    WHOLE sweep, warmed up on a 64 x 256 crop, one timed iteration and a second one if the first took
    less than half the limit -- unless the 64 x 32 probe predicts more than ``auto_full_limit`` seconds per iteration, in
    which case the cropped sample is taken and the ``sample`` string says so.
    """
    from oracle import model as om
    from oracle import targets as otgt

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:  # a cgroup CPU quota below the affinity mask (a shared GPU box): more threads than the quota only get throttled
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    H, W_full = 64, 2048
    backbone, head = build_model("rv-av2", AV2_CLASSES)
    sd = {**{f"backbone.{k}": v for k, v in backbone.state_dict().items()}, **{f"head.{k}": v for k, v in head.state_dict().items()}}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    weights = {**sd, **params}

    def run(W: int) -> float:
        batch = synthetic_batch(1, H, W, seed=0, device="cpu", boxes_per_sweep=2)
        t0 = time.perf_counter()
        feats, logits, reg = om.detector_forward(batch["features"], batch["cart"], weights, nm=om.Numerics(train=True))
        tg = otgt.compute_targets(batch["cart"], batch["annotations"], AV2_CLASSES)
        otgt.detection_loss(logits, reg, batch["cart"], batch["mask"], tg, AV2_CLASSES)["loss"].backward()
        for v in params.values():
            v.grad = None
        return time.perf_counter() - t0

    def config1() -> dict:
        """BASELINE configs[0] (debug-overfit: one synthetic 64 x 512 x 5 sweep, the nearest reference-valid tiny backbone
        ``layers=[16]*5`` -- RangeBackbone is hard-wired to five entries), CPU only: the oracle's fwd + loss + bwd."""
        tb, th = build_model("c16", 5)
        tsd = {**{f"backbone.{k}": v for k, v in tb.state_dict().items()}, **{f"head.{k}": v for k, v in th.state_dict().items()}}
        tparams = {k: v.clone().requires_grad_(True) for k, v in tsd.items() if v.dtype.is_floating_point and "running_" not in k}
        tw = {**tsd, **tparams}
        tbatch = synthetic_batch(1, H, 512, seed=0, device="cpu", boxes_per_sweep=4, n_cls=5)
        ts = []
        for _ in range(3):  # (the first one is the warm-up)
            t0 = time.perf_counter()
            _, lg, rg = om.detector_forward(tbatch["features"], tbatch["cart"], tw, nm=om.Numerics(train=True))
            ttg = otgt.compute_targets(tbatch["cart"], tbatch["annotations"], 5)
            otgt.detection_loss(lg, rg, tbatch["cart"], tbatch["mask"], ttg, 5)["loss"].backward()
            for v in tparams.values():
                v.grad = None
            ts.append(time.perf_counter() - t0)
        dt1 = sum(ts[1:]) / 2
        return {"value": 1.0 / dt1, "unit": "sweeps/s", "seconds_per_iteration": round(dt1, 3),
                "sample": "BASELINE configs[0]: oracle fwd+loss+bwd, layers=[16]*5 (towers 32, 5 classes), fp32, one synthetic 64x512x5 sweep, 1 warm-up + 2 timed iterations"}

    # thread count: all granted cores, unless a short probe shows the PyTorch-CPU convolutions are faster on 32
    probes = {}
    for th in sorted({cores, min(cores, 32)}):
        torch.set_num_threads(th)
        run(32)  # warm-up (allocator, thread pool), discarded
        probes[th] = run(32)
        _progress(f"cpu_baseline probe: 64x32 crop on {th} threads {probes[th]:.2f} s")
    threads = min(probes, key=probes.get)
    torch.set_num_threads(threads)
    W = W_full
    note = ""
    if auto_full_limit > 0 and not full:
        # measured on the pool's hosts: the whole sweep takes ~2.2x the 64x32 probe scaled by the width (cache footprint)
        est = 2.2 * probes[threads] * (W_full / 32)
        if est <= auto_full_limit:
            _progress(f"cpu_baseline: the whole 64x{W_full} sweep on {threads} threads (probe estimate <= {est:.0f} s per iteration); warm-up on a 64x256 crop")
            run(256)
            times = [run(W_full)]
            _progress(f"cpu_baseline full-sweep iteration 0: {times[0]:.1f} s")
            if times[0] < 0.5 * auto_full_limit:
                times.append(run(W_full))
                _progress(f"cpu_baseline full-sweep iteration 1: {times[1]:.1f} s")
            dt = sum(times) / len(times)
            return {"value": 1.0 / dt, "unit": "sweeps/s", "cores": threads, "kind": "port",
                    "sample": f"oracle fwd+loss+bwd, rv-av2 widths, fp32, B=1, 64x{W_full}x5 (the WHOLE sweep), warm-up on a 64x256 crop + "
                              f"{len(times)} timed iteration(s) of {' / '.join(f'{t:.1f}' for t in times)} s on {threads} threads ({cores} cores visible, {_cpu_model()})",
                    "config1": config1()}
        note = f"; the whole sweep was estimated at {est:.0f} s per iteration on this host: cropped sample instead"
    if not full:
        W = 32
        while W < W_full and probes[threads] * (2 * W / 32) * 3 < seconds_budget:  # warm-up + two timed iterations within the budget
            W *= 2
    _progress(f"cpu_baseline: 64x{W} crop on {threads} threads, 1 warm-up + 2 timed iterations")
    times = []
    for i in range(3):  # the first one is the warm-up at the timed size
        times.append(run(W))
        _progress(f"cpu_baseline iteration {i}: {times[-1]:.1f} s")
    times = times[1:]
    dt = sum(times) / len(times)
    return {
        "value": (W / W_full) / dt, "unit": "sweeps/s", "cores": threads, "kind": "port",
        "sample": f"oracle fwd+loss+bwd, rv-av2 widths, fp32, B=1, 64x{W}x5 ({'the full sweep' if W == W_full else f'{W}/{W_full} of a sweep'}), "
                  f"1 warm-up + 2 timed iterations of {times[0]:.1f} / {times[1]:.1f} s on {threads} threads ({cores} cores visible, {_cpu_model()}){note}",
        "config1": config1(),
    }


def forward_only_leg(model, batch, n_cls: int, dev, warmup: int = 5, iters: int = 20) -> dict:
    """BASELINE configs[1] / the reference's latency harness (tools/benchmark.py:91-122, 231-238): eval forward of the rv-av2
    model on the bench batch (4 sweeps), ``RangeDecoder.decode(use_nms=True)`` behind it; every stage bracketed by
    ``torch.cuda.synchronize()`` as ``bench()`` there does, 5 warm-up + 20 timed iterations.  The model is at random init
    (a fresh seeded model, classification bias -4.6: nothing reaches ``min_confidence``), so 3 % of the pixels are lifted over the threshold to
    give the NMS ~1.8 k boxes per sweep to work on (done inside the decoder stage's timing).  Also the un-synchronised
    pipeline rate, and the dominant forward kernel against the MFMA peak from events around its launches."""
    from range_view_3d_detection_amd import engine as E
    from range_view_3d_detection_amd.nn.decoders.range_decoder import RangeDecoder

    del model  # a FRESH random-init model (seeded): the headline's model has taken optimizer steps on one synthetic batch
    torch.manual_seed(0)
    backbone, head = build_model("rv-av2", n_cls)
    backbone, head = backbone.to(dev).eval(), head.to(dev).eval()
    dec = RangeDecoder(True, True, [0, 15, 30], [15, 30, math.inf], [8, 2, 1])
    post = {"num_pre_nms": 50000, "num_post_nms": 1000, "nms_threshold": 0.3, "min_confidence": 0.1, "nms_mode": "WEIGHTED"}
    tasks = {0: [f"C{i}" for i in range(n_cls)]}
    B = batch["features"].shape[0]
    g = torch.Generator(device=dev).manual_seed(0)
    bump = 3.0 * (torch.rand(B, 1, *batch["features"].shape[2:], device=dev, generator=g) < 0.03).float()

    def stage(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, 1e3 * (time.perf_counter() - t0)

    def decode(out):
        out[1][0]["logits"] = out[1][0]["logits"] + bump
        return dec.decode(out, post, tasks, use_nms=True)

    times = {"backbone": [], "head": [], "decoder": []}
    boxes = 0
    # fp16 autocast: what the reference's harness and validation_step run under (tools/benchmark.py:84-88, detector.py:329-333):
    # the eval programs then run on the fp16-operand build of the library (librv3d_hip_f16.so)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for i in range(warmup + iters):
            feats, tb = stage(lambda: backbone(batch))
            (out, _), th = stage(lambda: head(feats, batch, return_loss=False))
            res, td = stage(lambda: decode(out))
            if i >= warmup:
                times["backbone"].append(tb)
                times["head"].append(th)
                times["decoder"].append(td)
                boxes = int(res[0].shape[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):  # the same three calls back to back, one synchronisation at the end
            feats = backbone(batch)
            out, _ = head(feats, batch, return_loss=False)
            res = decode(out)
        torch.cuda.synchronize()
        pipelined = 1e3 * (time.perf_counter() - t0) / iters
        E.PROFILE = prof = E.KernelProfile()
        for _ in range(3):
            feats = backbone(batch)
            out, _ = head(feats, batch, return_loss=False)
        torch.cuda.synchronize()
        E.PROFILE = None
    mean = {k: sum(v) / len(v) for k, v in times.items()}
    total = sum(mean.values())
    roof = prof.roofline(MFMA_BF16_PEAK_TFLOPS)
    return {"workload": f"rv-av2 eval forward + decode + weighted NMS, {B} synthetic 64x2048x5 sweeps (BASELINE configs[1]); stages synchronised as tools/benchmark.py:231-238",
            "batch": B, "ms_per_batch": {k: round(v, 3) for k, v in mean.items()}, "ms_per_batch_total": round(total, 3),
            "ms_per_sweep": round(total / B, 3), "sweeps_per_s": round(1e3 * B / total, 2), "ms_per_batch_pipelined": round(pipelined, 3),
            "sweeps_per_s_pipelined": round(1e3 * B / pipelined, 2), "boxes_out_per_batch": boxes, "dtype": "f16",
            "dominant_kernel": {k: roof.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us")}}


def rv_waymo_leg(dev, batch_size: int = 4, warmup: int = 3, steps: int = 10) -> dict:
    """The single-GPU shard of BASELINE configs[4]: rv-waymo widths ([128]*5, towers 256, 3 classes), 6 input features,
    64 x 2656 sweeps (2650 padded by [3, 3]), the same training step as the headline (fwd + targets + loss + bwd + clip + AdamW)."""
    from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

    torch.manual_seed(0)
    backbone, head = build_model("rv-waymo", 3, 6)
    model = Detector(backbone, head).to(dev).train()
    params = [p for p in model.parameters()]
    opt, sched = configure_optimizers(params, num_devices=1, batch_size=batch_size, total_steps=warmup + steps + 8, fused=True, max_grad_norm=35.0)
    batch = synthetic_batch(batch_size, 64, 2656, seed=4321, device=dev, n_feat=6, n_cls=3)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model(batch)
        loss.backward()
        opt.step()
        sched.step()
        return loss

    from range_view_3d_detection_amd import engine as E

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    E.PROFILE = prof = E.KernelProfile()  # events around each tap-conv / wgrad launch, as in the headline's timed region
    sampler = GpuSampler(dev.index or 0).start()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_conditions = sampler.stop()
    if gpu_conditions["sclk_mhz_median"]:
        gpu_conditions["ms_per_step_at_reference_sclk"] = round(1e3 * dt / steps * gpu_conditions["sclk_mhz_median"] / SCLK_REFERENCE_MHZ, 3)
    E.PROFILE = iso = E.KernelProfile()  # ... and with the side stream off (the kernels' own durations; see roofline() below)
    overlap, E.OVERLAP_WGRAD = E.OVERLAP_WGRAD, False
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    E.OVERLAP_WGRAD = overlap
    E.PROFILE = None
    summ, isum = prof.summary(), iso.summary()
    roof = roofline(prof, iso, pmc_key="rv_waymo")  # in-run figures; `isolated` beside them; traffic from the waymo PMC passes
    ws = whole_step("rv-waymo", 2656, batch_size, dt / steps)
    tr = traffic_ratio("rv_waymo", "rv-waymo", 2656, batch_size)
    if tr is not None:
        ws["traffic_ratio"] = tr
    hbm = measure_hbm_group(step, pmc_key="rv_waymo")
    return {"workload": f"rv-waymo full model, fwd+bwd+AdamW, {batch_size} synthetic 64x2656x6 sweeps (single-GPU shard of BASELINE configs[4])",
            "sweeps_per_s": round(batch_size * steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": warmup,
            "loss": float(loss.detach().item()), "dtype": "bf16", "gpu": gpu_conditions,
            "roofline": roof, "roofline_hbm": hbm, "whole_step": ws,
            "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "tflops": round(v["tflops"], 1)} for k, v in summ.items()},
            "kernels_isolated": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "tflops": round(v["tflops"], 1)} for k, v in isum.items()}}


def whole_step(widths: str, width: int, sweeps: int, seconds: float) -> dict:
    """The whole training step against the dense bf16 MFMA peak: 3 x the model's forward FLOPs per sweep."""
    tflop = 3.0 * FWD_TFLOP_PER_SWEEP[(widths, width)] * sweeps
    return {"tflop": round(tflop, 2), "achieved": round(tflop / seconds, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflop / seconds / MFMA_BF16_PEAK_TFLOPS, 4)}


def first_step_loss(dev, widths: str = "rv-av2", n_cls: int = AV2_CLASSES, n_feat: int = 5, width: int = 2048) -> float:
    """ONE reproducible number of the benchmarked model: the loss of the freshly initialised (seed 0) model in train mode on the
    first sweep of the benchmark's own synthetic data (seed 1234, B = 1), before any update.  tests/test_gpu_fullsize_train.py
    compares it with the oracle's fp32 value of the same quantity (1e-2)."""
    torch.manual_seed(0)
    backbone, head = build_model(widths, n_cls, n_feat)
    model = Detector(backbone, head).to(dev).train()
    batch = synthetic_batch(1, 64, width, seed=1234, device=dev, n_feat=n_feat, n_cls=n_cls)
    with torch.no_grad():
        return float(model(batch).item())


def _pmc_rows() -> dict:
    try:
        with open(os.path.join(ROOT, PMC_TRAFFIC)) as f:
            return json.load(f)
    except OSError:
        return {}


def _pmc_section(key: str) -> dict:
    """The headline's rows (key "kernels": the top level of the file) or a named section of the same shape (``rv_waymo``)."""
    pmc = _pmc_rows()
    return pmc if key == "kernels" else pmc.get(key, {})


def _pmc_traffic_of(kernel: str, pmc: dict):
    """HBM-side bytes per launch of ``kernel`` from the committed PMC passes (launch-weighted over its template instances)."""
    rows = pmc.get("kernels", {})
    name = kernel.replace("(+reduce)", "")
    stem = name[:-1] if name.endswith(">") else name
    hits = [v for k, v in rows.items() if k == name or k.startswith(stem + ",") or k.startswith(stem + ">")] or [v for k, v in rows.items() if k == name.split("<")[0]]
    if not hits and name.startswith("tapconv6_kernel"):  # (its template arguments are the epilogue kind, not the tile width: all instances)
        hits = [v for k, v in rows.items() if k.startswith("tapconv6_kernel<")]
    if not hits:
        return None
    n = sum(v["launches"] for v in hits)
    return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / n


def roofline(prof, iso, pmc_key: str = "kernels") -> dict:
    """Dominant kernel against the dense bf16 MFMA peak.  ``achieved`` / ``frac`` / ``avg_launch_us`` are the IN-RUN figures: HIP
    events around every launch of the kernel inside the timed region, on the stream it runs on, in the configuration that produced
    ``ms_per_step`` (weight gradients free-running on the side stream: a backward-data launch queued behind a resident wgrad3
    workgroup has that wait in its event-to-event time).  ``isolated`` = the same events in two extra steps with the side stream
    off (the kernel's own duration; what ``rocprofv3 --kernel-trace`` of the one-stream run reports).  The dominant kernel is the
    one with the most time of its own (isolated), so that a wait does not choose it.  ``traffic`` = HBM-side bytes per launch from
    the committed PMC passes over this same command (``PMC_TRAFFIC``, profiles/pmc_traffic.py: FETCH_SIZE x 2 on gfx950 +
    WRITE_SIZE, separate passes; null when that file has no row for the kernel)."""
    iso_sum, live_sum = iso.summary(), prof.summary()
    if not iso_sum or not live_sum:
        return {}
    name = max((k for k in iso_sum if k in live_sum), key=lambda k: iso_sum[k]["ms"])
    r = prof.roofline(MFMA_BF16_PEAK_TFLOPS, name)
    r["measured"] = "HIP events around every launch of the kernel INSIDE the timed region (the configuration of ms_per_step: weight gradients on the side stream)"
    i = iso_sum[name]
    r["isolated"] = {"achieved": i["tflops"], "frac": i["tflops"] / MFMA_BF16_PEAK_TFLOPS, "avg_launch_us": i["avg_us"], "launches": i["launches"],
                     "measured": "the same events in 2 extra training steps with the side stream off (the kernel alone on the GPU)"}
    pmc = _pmc_section(pmc_key)
    t = _pmc_traffic_of(r["kernel"], pmc) if pmc else None
    r["traffic"] = t
    if t is not None:
        r["traffic_source"] = PMC_TRAFFIC + (f" [{pmc_key}]" if pmc_key != "kernels" else "") + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 gfx950 correction)"
    return r


# entry point of the C ABI -> the kernels it launches (profiler names), for the HBM-bound group of the step
HBM_GROUP_KERNELS = {
    "rv_ew_combine": ("ew_combine",), "rv_bn_bwd_reduce": ("bn_bwd_reduce_",), "rv_bn_bwd_apply": ("bn_bwd_apply_",),
    "rv_bn_bwd_reduce_pair": ("bn_bwd_reduce2",), "rv_bn_bwd_apply_pair": ("bn_bwd_apply2",), "rv_meta_modulate": ("meta_modulate_kernel",),
    "rv_meta_modulate_bwd_sums": ("meta_bwd_sums",), "rv_meta_modulate_bwd_apply": ("meta_bwd_apply",), "rv_pos_forward": ("pos_fwd_kernel",),
    "rv_pos_backward_sums": ("pos_bwd_kernel",), "rv_head_final_bwd_sums": ("head_final_bwd_kernel<false",),
    "rv_head_final_bwd_apply": ("head_final_bwd_kernel<true",),
}


def measure_hbm_group(step, steps: int = 2, pmc_key: str = "kernels") -> dict:
    """In-run figures (`_hbm_group_pass` in the timed configuration) with the one-stream figures beside them (`isolated`: with the
    weight gradients free-running at high priority on the side stream a bandwidth-bound pass shares the CUs and the memory system
    with a resident wgrad3 workgroup, and its event-to-event time includes that)."""
    from range_view_3d_detection_amd import engine as E

    r = _hbm_group_pass(step, steps, pmc_key)
    if r and E.OVERLAP_WGRAD:
        overlap, E.OVERLAP_WGRAD = E.OVERLAP_WGRAD, False
        try:
            i = _hbm_group_pass(step, steps, pmc_key)
        finally:
            E.OVERLAP_WGRAD = overlap
        r["isolated"] = {k: i[k] for k in ("achieved", "frac", "ms_per_step")}
        r["isolated"]["measured"] = "the same events with the side stream off (the passes alone on the GPU)"
    return r


def _hbm_group_pass(step, steps: int, pmc_key: str) -> dict:
    """The HBM-bound kernels of the step (BatchNorm backward reduce / apply, the element-wise block passes, the MetaKernel stem's
    gather / modulation and positional-pair kernels) against the HBM roofline: HIP events around each of their launches in
    ``steps`` extra training steps in the SAME configuration as the timed region (two streams), algorithmic bytes per launch from
    the entry points' own arguments (range_view_3d_detection_amd/_lib.py::HBM_BYTES: operands once in, results once out)."""
    from range_view_3d_detection_amd import _lib as L

    recs = []

    def hook(name, nbytes, launch):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        recs.append((name, nbytes, a, b))

    L.HBM_HOOK = hook
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    finally:
        L.HBM_HOOK = None
    agg = {}
    for name, nbytes, a, b in recs:
        d = agg.setdefault(name, {"launches": 0, "ms": 0.0, "gbyte": 0.0})
        d["launches"] += 1
        d["ms"] += a.elapsed_time(b)
        d["gbyte"] += nbytes / 1e9
    pmc = _pmc_section(pmc_key).get("kernels", {})
    out_k, tot_ms, tot_gb, tot_traffic, traffic_ok = {}, 0.0, 0.0, 0.0, bool(pmc)
    for name, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        per_step = {"launches": d["launches"] / steps, "ms": round(d["ms"] / steps, 3), "algorithmic_gb": round(d["gbyte"] / steps, 3),
                    "achieved_gbs": round(d["gbyte"] / max(d["ms"] * 1e-3, 1e-12), 1)}
        hits = [v for k, v in pmc.items() if any(k.startswith(pre) for pre in HBM_GROUP_KERNELS[name])]
        if hits:  # HBM-side bytes per step of this entry point's kernels: bytes per launch (PMC) x this run's launches per step
            n = sum(v["launches"] for v in hits)
            per_step["traffic_gb"] = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / n * per_step["launches"] / 1e9, 3)
            tot_traffic += per_step["traffic_gb"]
        else:
            traffic_ok = False
        out_k[name] = per_step
        tot_ms += d["ms"] / steps
        tot_gb += d["gbyte"] / steps
    if not agg:
        return {}
    ach = tot_gb / max(tot_ms * 1e-3, 1e-12)
    return {"bound": "hbm", "kernels": "BatchNorm backward reduce/apply, element-wise block passes, MetaKernel stem gather/modulation + positional pair",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "ms_per_step": round(tot_ms, 3), "algorithmic_gb_per_step": round(tot_gb, 3),
            "traffic": round(1e9 * tot_traffic, 1) if traffic_ok else None, "traffic_gb_per_step": round(tot_traffic, 3) if traffic_ok else None,
            "measured": f"HIP events around each launch of the group in {steps} extra training steps, same two-stream configuration as the timed region; "
                        "algorithmic bytes from the C-ABI arguments; traffic from " + PMC_TRAFFIC,
            "per_entry_point": out_k}


def traffic_ratio(pmc_key: str, widths: str, width: int, sweeps: int):
    """Whole-step HBM-side bytes (all kernels, PMC passes) over the algorithmic bytes of the step (activations once in /
    once out per conv with BatchNorm / ReLU / add fused, bf16: forward figure x 3 for fwd + bwd)."""
    sec = _pmc_section(pmc_key)
    if not sec or "total_gb_all_kernels" not in sec or not sec.get("steps_profiled"):
        return None
    alg = 3.0 * FWD_ALG_GB_PER_SWEEP[(widths, width)] * sweeps
    per_step = sec["total_gb_all_kernels"] / sec["steps_profiled"]
    return {"hbm_gb_per_step": round(per_step, 1), "algorithmic_gb_per_step": round(alg, 1), "ratio": round(per_step / alg, 3), "source": PMC_TRAFFIC}


SCLK_REFERENCE_MHZ = 2100.0  # ms_per_step_at_reference_sclk = ms_per_step * sclk_mhz_median / this (comparable across boxes of the pool)


class GpuSampler:
    """Shader clock and package power of this rank's GPU, sampled from sysfs in a host thread while the timed region runs (no extra
    process, nothing that touches the GPU: the hwmon / pp_dpm files of the device's PCI node).  One binary differs by up to 4.5 % between
    boxes of the pool -- this MFMA-heavy step runs at the package power cap, where the clock a given chip holds decides -- so a bench
    record without the conditions it ran under cannot tell a slower box from a regression (round-5 review, item 6; the reference's
    harness states its conditions the same way, tools/benchmark.py:231-238)."""

    def __init__(self, device_index: int, period_s: float = 0.05) -> None:
        import glob
        import threading

        self.period = period_s
        self.clk, self.pw = [], []
        self.cap_w = None
        self.source = None
        self._stop = threading.Event()
        self._thread = None
        node = None
        try:
            props = torch.cuda.get_device_properties(device_index)
            bdf = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
            if os.path.isdir(f"/sys/bus/pci/devices/{bdf}"):
                node = f"/sys/bus/pci/devices/{bdf}"
        except Exception:  # noqa: BLE001 (an older torch without the PCI fields)
            node = None
        if node is None:  # one visible card: take the only amdgpu node that has an hwmon directory
            cands = [os.path.dirname(os.path.dirname(h)) for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")]
            node = cands[device_index] if device_index < len(cands) else (cands[0] if cands else None)
        self.f_clk = self.f_pw = self.f_dpm = None
        if node is not None:
            hw = sorted(glob.glob(os.path.join(node, "hwmon", "hwmon*")))
            if hw:
                for name in ("power1_average", "power1_input"):
                    if os.path.exists(os.path.join(hw[0], name)):
                        self.f_pw = os.path.join(hw[0], name)
                        break
                if os.path.exists(os.path.join(hw[0], "freq1_input")):
                    self.f_clk = os.path.join(hw[0], "freq1_input")
                cap = self._read(os.path.join(hw[0], "power1_cap"))
                self.cap_w = round(cap / 1e6, 1) if cap else None
            if os.path.exists(os.path.join(node, "pp_dpm_sclk")):
                self.f_dpm = os.path.join(node, "pp_dpm_sclk")
            self.source = node

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except Exception:  # noqa: BLE001
            return None

    def _dpm_mhz(self):
        try:
            with open(self.f_dpm) as f:
                for line in f:
                    if line.rstrip().endswith("*"):
                        return float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        except Exception:  # noqa: BLE001
            pass
        return None

    def _run(self) -> None:
        while not self._stop.is_set():
            c = self._read(self.f_clk) if self.f_clk else None
            c = c / 1e6 if c else (self._dpm_mhz() if self.f_dpm else None)
            if c:
                self.clk.append(c)
            w = self._read(self.f_pw) if self.f_pw else None
            if w:
                self.pw.append(w / 1e6)
            self._stop.wait(self.period)

    def start(self) -> "GpuSampler":
        import threading

        if self.f_clk or self.f_pw or self.f_dpm:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def stop(self) -> dict:
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
        med = lambda v: round(sorted(v)[len(v) // 2], 1) if v else None
        return {"sclk_mhz_median": med(self.clk), "power_w_median": med(self.pw), "power_cap_w": self.cap_w,
                "sclk_mhz_reference": SCLK_REFERENCE_MHZ,
                "sclk_mhz_min": round(min(self.clk), 1) if self.clk else None, "power_w_max": round(max(self.pw), 1) if self.pw else None,
                "samples": max(len(self.clk), len(self.pw)), "period_ms": round(1e3 * self.period),
                "source": (f"sysfs {os.path.basename(self.f_clk or self.f_dpm or '-')} / {os.path.basename(self.f_pw or '-')} of {self.source}") if self.source else None,
                "note": "host-side sysfs view sampled in a thread during the timed region; the in-kernel clock of an MFMA-dense loop reads up to ~10 % "
                        "below it (MI355X_MICROARCH.md, DVFS give-back 6) -- for comparing records across boxes, not a kernel measurement"}


def main(args=None) -> None:
    args = args if args is not None else parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("RV3D_FORCE_DIST") is not None  # (forced: the RCCL path with one rank, tests/test_gpu_ddp.py)
    if torch.cuda.device_count() == 0:  # (counting devices does not initialise HIP)
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        backend = os.environ.get("RV3D_DIST_BACKEND", "nccl")  # "gloo": the 2-ranks-on-one-GPU test of this script
        local_rank %= max(torch.cuda.device_count(), 1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from range_view_3d_detection_amd import engine as E

    torch.manual_seed(0)
    backbone, head = build_model(args.widths, args.classes, args.features)
    model = Detector(backbone, head).to(dev).train()
    E.SYNC_BN = dist_on and not args.no_sync_bn  # explicit: sync every BatchNorm (what Lightning's sync_batchnorm: true does) / local
    step_model = model
    params = [p for p in model.parameters()]
    use_ddp = os.environ.get("RV3D_DDP") is not None  # A/B: torch's DistributedDataParallel instead of engine.GradSync
    if dist_on and use_ddp:
        step_model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], gradient_as_bucket_view=True, static_graph=True)
    elif dist_on:
        # the model's backward is three autograd nodes: one flat gradient buffer, one multi-tensor copy + one asynchronous RCCL
        # all-reduce per finished node, instead of DDP's per-parameter hooks / bucket copies / divisions (engine.GradSync)
        E.GRAD_SYNC = E.GradSync(params, world)
        if world > 1:
            E.GRAD_SYNC.broadcast_parameters(model)
    # the reference's recipe (nn/meta/arch.py:48-75): AdamW(1e-3) + OneCycleLR(max_lr = 0.00075 * sqrt(devices * batch)), per step
    from range_view_3d_detection_amd.nn.meta.arch import configure_optimizers

    fused_opt = True  # (False: torch.optim.AdamW + clip_grad_norm_, ~10 foreach launches -- tests/test_gpu_model.py compares the two)
    opt, sched = configure_optimizers(params, num_devices=world, batch_size=args.batch, total_steps=args.warmup + args.steps + 8,
                                      fused=fused_opt, max_grad_norm=35.0 if fused_opt else None)
    batch = synthetic_batch(args.batch, args.height, args.width, seed=1234 + rank, device=dev, n_feat=args.features, n_cls=args.classes)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = step_model(batch)
        loss.backward()
        if E.GRAD_SYNC is not None:
            E.GRAD_SYNC.finish()  # gradients averaged over the ranks, p.grad = views of the flat buffer
        if not fused_opt:
            torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()  # fused: gradient clipping at 35.0 + AdamW in two launches (rv_adamw_step)
        sched.step()
        return loss

    _progress(f"model on {dev}, {args.batch} sweeps of {args.height}x{args.width}x{args.features}; warm-up")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    _progress("warm-up done; timed region")
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # events around each tap-conv / wgrad launch inside the timed region
    prof = E.KernelProfile()
    E.COLLECTIVES.reset()
    sampler = GpuSampler(local_rank).start() if rank == 0 else None
    t0 = time.perf_counter()
    E.PROFILE = prof
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gpu_conditions = sampler.stop() if sampler is not None else None
    if gpu_conditions is not None and gpu_conditions["sclk_mhz_median"]:
        # the step is MFMA-dominated and power-capped: to first order its time scales with 1 / (the clock this chip holds under it)
        gpu_conditions["ms_per_step_at_reference_sclk"] = round(1e3 * elapsed / args.steps * gpu_conditions["sclk_mhz_median"] / SCLK_REFERENCE_MHZ, 3)
    _progress(f"timed region done: {1e3 * elapsed / args.steps:.1f} ms per step")
    E.PROFILE = None
    sync_calls, sync_bytes = E.COLLECTIVES.calls / args.steps, E.COLLECTIVES.bytes / args.steps
    # the same per-kernel events once more, outside the timed region, with the weight-gradient side stream off: with
    # it on, kernels of the two streams share the CUs and each one's event-to-event time includes its neighbour's
    iso = E.KernelProfile()
    hbm_group = {}
    if not args.timed_only:
        E.PROFILE, overlap = iso, E.OVERLAP_WGRAD
        E.OVERLAP_WGRAD = False
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        E.OVERLAP_WGRAD = overlap
        E.PROFILE = None
        # ... and the HBM-bound group (BatchNorm backward, element-wise, stem) in the timed configuration again, with events around ITS launches
        hbm_group = measure_hbm_group(step)  # (every rank: the steps carry collectives)
    ranks_seen, devices_seen = 1, 1
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        # proof of how many ranks the BACKEND saw (not what WORLD_SIZE says): a sum of ones over the real process group, and the number of
        # distinct (host, PCI device) pairs among them -- two ranks sharing one GPU (tests/test_gpu_ddp.py, gloo) count as one device
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(ones)
        ranks_seen = int(round(float(ones.item())))
        props = torch.cuda.get_device_properties(dev)
        ident = f"{os.uname().nodename}/{getattr(props, 'pci_domain_id', 0)}:{getattr(props, 'pci_bus_id', local_rank)}:{getattr(props, 'pci_device_id', 0)}"
        idents = [None] * world
        torch.distributed.all_gather_object(idents, ident)
        devices_seen = len(set(idents))

    if rank == 0:
        sweeps = args.batch * world * args.steps
        out = {
            "metric": "sweeps/sec (fwd+bwd, 64x2048x5 range image)", "value": sweeps / elapsed, "unit": "sweeps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "ranks_seen": ranks_seen, "devices_seen": devices_seen,
            "gpu": gpu_conditions,
            "config": {"workload": f"{args.widths} full model (MetaKernel stem + DLA backbone/FPN + cls/reg towers + targets + loss), "
                                   f"fwd+bwd+AdamW, {args.batch} synthetic {args.height}x{args.width}x{args.features} sweeps per GPU"
                                   + (" (BASELINE configs[2])" if (args.widths, args.width, args.height, args.batch) == ("rv-av2", 2048, 64, 4) else ""),
                       "global_batch": args.batch * world, "sweep": [args.height, args.width, args.features], "parallelism": f"dp{world}",
                       "sync_bn": bool(E.SYNC_BN), "loss": float(loss.detach().item()),
                       "collectives": {"per_step": {"sync_bn_all_reduce": {"calls": sync_calls, "bytes": sync_bytes},
                                                    "gradient_all_reduce_bytes": 4 * sum(p.numel() for p in params) if dist_on else 0},
                                       "gradient_sync": ("DistributedDataParallel" if use_ddp else "engine.GradSync (flat buffer, one all-reduce per autograd node)") if dist_on else None,
                                       "note": "SyncBN: (2C+1) fp32 per BatchNorm layer and direction, layers that become available together share one all-reduce"},
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2)},
            "roofline": roofline(prof, iso),
            "roofline_hbm": hbm_group,
            "kernels": prof.summary(),
            # the same launches with nothing else on the GPU (side stream off, two steps outside the timed region): with the small
            # layers' weight gradients on the side stream the live event-to-event times of overlapping kernels include their neighbour
            "kernels_isolated": iso.summary(),
        }
        if (args.widths, args.width) in FWD_TFLOP_PER_SWEEP and args.height == 64:
            out["whole_step"] = whole_step(args.widths, args.width, args.batch * world, elapsed / args.steps)
            tr = traffic_ratio("kernels", args.widths, args.width, args.batch) if (args.widths, args.batch) == ("rv-av2", 4) else None
            if tr is not None:  # (per GPU: the PMC passes are one rank's)
                out["whole_step"]["traffic_ratio"] = tr
        headline = (args.widths, args.width, args.height, args.features) == ("rv-av2", 2048, 64, 5)
        if world == 1 and headline and not args.timed_only:
            # (the model of the timed region has taken optimizer steps; this is a fresh one: 0.3 s)
            out["loss_first_step"] = {"value": first_step_loss(dev), "what": "train-mode loss of the seed-0 model on sweep 0 of the seed-1234 batch, B = 1, "
                                      "before any update; the fp32 oracle's value of the same quantity is asserted in tests/test_gpu_fullsize_train.py (1e-2)"}
        if world == 1 and not args.no_extra and headline and not args.timed_only:
            # extra keys, outside the timed region: BASELINE configs[1] (forward only + decode + NMS) and the one-GPU shard of configs[4]
            _progress("forward_only leg (eval forward + decode + weighted NMS)")
            out["forward_only"] = forward_only_leg(model, batch, args.classes, dev)
            del opt, sched, step_model, loss
            model.zero_grad(set_to_none=True)
            torch.cuda.empty_cache()
            _progress("rv_waymo leg (64x2656x6, training step)")
            out["rv_waymo"] = rv_waymo_leg(dev)
        if world == 1 and not args.no_cpu_baseline and not args.timed_only:
            out["cpu_baseline"] = cpu_baseline(full=args.cpu_baseline == "full", auto_full_limit=160.0 if args.cpu_baseline == "auto" else 0.0)
        print(json.dumps(out))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(_ARGS)
