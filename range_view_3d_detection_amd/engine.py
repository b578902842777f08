"""Host-side execution engine: NHWC bf16 device tensors + a tape of fused HIP ops.

The reference runs the detector as ~600 separate ATen/cuDNN calls stitched together by
autograd.  Here a forward pass records a short *tape* of fused operations (tap-conv with
folded BatchNorm prologue and statistics epilogue, BatchNorm finalisation, element-wise
combine, MetaKernel gather/modulate); the backward pass replays the tape in reverse with
hand-derived gradients.  PyTorch only sees two autograd nodes (backbone, head), owns the
memory (caching allocator) and the stream.  Every op calls the C ABI in ``include/rv3d.h``
through ``_lib`` -- there is no PyTorch/CPU fallback for any of them.
"""

from __future__ import annotations

import ctypes
import os
import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor, nn

from . import _lib as L

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# Cross-rank BatchNorm statistics (the reference trains with ``sync_batchnorm: true``, conf/trainer/train.yaml:15: Lightning
# converts every BatchNorm2d into nn.SyncBatchNorm).  When a layer is "sync", the per-channel (sum, sum of squares, count) /
# (sum g, sum g*xhat) reductions are all-reduced over the default process group (RCCL) before they are finalised.
#   SYNC_BN = None  (default) decide per layer from the parameter holder: nn.SyncBatchNorm -> sync; a plain BatchNorm2d in
#                   training mode under an initialised process group with world_size > 1 RAISES (silently using per-rank
#                   statistics would diverge from the reference) -- pick one of the two explicit settings instead;
#   SYNC_BN = True  sync every BatchNorm regardless of the holder's class (what bench.py sets for N > 1);
#   SYNC_BN = False per-rank statistics (torch DDP semantics with plain BatchNorm2d).
SYNC_BN: Optional[bool] = None
# Test hook: with SYNC_BN = True, take the synchronised code path (device reduce -> all-reduce -> finalize from the totals)
# even in a process group of ONE rank -- the only way to run the RCCL collectives of this path on a one-GPU box.
SYNC_WORLD1 = os.environ.get("RV3D_SYNC_WORLD1") is not None


def _dist_world() -> int:
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_world_size()
    return 1


def bn_sync_world(bn: nn.Module, training: bool) -> int:
    """World size the statistics of ``bn`` are reduced over (1 = local)."""
    world = _dist_world()
    if SYNC_WORLD1 and SYNC_BN is True and training and torch.distributed.is_initialized():
        return max(world, 2)  # (callers only test `> 1`; the counts travel through the all-reduce)
    if world == 1 or not training or SYNC_BN is False:
        return 1
    if SYNC_BN is True or isinstance(bn, nn.SyncBatchNorm):
        return world
    raise L.RvError(
        f"{type(bn).__name__} in training mode under a process group of {world} ranks: the reference trains with "
        "sync_batchnorm (conf/trainer/train.yaml:15).  Convert the modules (torch.nn.SyncBatchNorm.convert_sync_batchnorm / "
        "Lightning sync_batchnorm: true), or set range_view_3d_detection_amd.engine.SYNC_BN = True (sync) / False (per-rank "
        "statistics) explicitly.")


class KernelProfile:
    """Event pairs around individual kernel launches (bench.py: live per-kernel timing inside the timed region)."""

    def __init__(self) -> None:
        self.records: List[tuple] = []  # (name, flops, start event, end event, algorithmic bytes)

    def launch(self, name: str, flops: float, fn, nbytes: float = 0.0) -> None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.records.append((name, flops, a, b, nbytes))

    def summary(self) -> Dict[str, Dict[str, float]]:
        torch.cuda.synchronize()
        agg: Dict[str, Dict[str, float]] = {}
        for name, flops, a, b, nbytes in self.records:
            d = agg.setdefault(name, {"launches": 0, "ms": 0.0, "tflop": 0.0, "gbyte": 0.0})
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
            d["tflop"] += flops / 1e12
            d["gbyte"] += nbytes / 1e9
        for d in agg.values():
            d["avg_us"] = 1e3 * d["ms"] / max(d["launches"], 1)
            d["tflops"] = d["tflop"] / max(d["ms"] * 1e-3, 1e-12)
        return agg

    def roofline(self, peak_tflops: float, name: Optional[str] = None) -> Dict[str, object]:
        agg = self.summary()
        if not agg:
            return {}
        if name is None or name not in agg:
            name = max(agg, key=lambda k: agg[k]["ms"])
        d = agg[name]
        return {"kernel": name, "bound": "mfma", "achieved": d["tflops"], "peak": peak_tflops, "unit": "TFLOP/s",
                "frac": d["tflops"] / peak_tflops, "traffic": None, "launches": d["launches"], "avg_launch_us": d["avg_us"],
                "flops_per_launch": 1e12 * d["tflop"] / max(d["launches"], 1),
                # every operand tensor once in, the result once out, bf16 (fp32 for a weight gradient's result)
                "algorithmic_bytes": 1e9 * d["gbyte"] / max(d["launches"], 1)}


PROFILE: Optional[KernelProfile] = None

# Weight gradients can run on a second HIP stream: wgrad(L) is independent of the main backward chain (dgrad(L) -> BatchNorm
# backward(L-1) -> dgrad(L-1) ...).
#   RV3D_OVERLAP=free        (DEFAULT since the end of round 4, see below) every weight gradient on the side stream.  Round-3 A/B on the final kernels
#                            (profiles/r03_overlap_ab.md): 3.6 % SLOWER than one stream -- two persistent one-workgroup-per-CU
#                            kernels with 150 KB of LDS each cannot share a CU, the big layers only take turns;
#   RV3D_OVERLAP=small[:T]   (T = 0.1; the round-3 default) only the weight gradients of layers below T TFLOP (the 128-channel DLA stages at
#                            W <= 1024 and the 1x1 layers, whose kernels have fewer tiles than the chip has CUs) -- they run
#                            beside the equally small backward-data / BatchNorm kernels of the main chain on CUs those leave
#                            idle: -2.0 ms per step (103.1 / 103.3 -> 101.1 at T = 0.1, 101.2 / 101.7 at 0.35, 101.9 at 0.7);
#   RV3D_OVERLAP=chain[:T]   (-0.5 .. -0.9 ms per rv-av2 step against small, profiles/r04_ab_notes.md) the small layers as above, and the BIG weight gradients on the side stream too, but chained: the next
#                            backward-data launch of the main stream waits for the weight gradient issued before it, so the two
#                            MFMA families still take turns on the CUs (dgrad(L), wgrad(L), dgrad(L-1), ...) -- what runs BESIDE
#                            wgrad(L) is the bandwidth-bound BatchNorm backward of layer L-1 on the main stream, whose kernels are
#                            sized (<= 96 VGPRs, <= 28 KB of LDS) to fit on a CU next to a resident wgrad3 workgroup;
#   RV3D_OVERLAP=off         one stream.
# Round 4, after the lean BatchNorm-backward kernels, the balanced wgrad3 split and GPU_MAX_HW_QUEUES=8: `free` is the fastest again
# (93.55 / 93.77 against 95.03 / 95.02 chain, 95.59 / 95.46 small, 96.77 off on one box; 94.08 / 95.50 against 96.64 / 96.35 on
# another; rv-waymo 53.7 against 53.9) -- the chained mode pays an event round trip per layer, and with the lean passes resident
# beside the weight gradients (the octet forms: +1 ms in this mode) there is now something to run beside them.
_OVERLAP = os.environ.get("RV3D_OVERLAP", "free")
OVERLAP_WGRAD = _OVERLAP == "free" or _OVERLAP.startswith("small") or _OVERLAP.startswith("chain")
OVERLAP_CHAIN = _OVERLAP.startswith("chain")
# Synchronised BatchNorm over real ranks: weight gradients are held until the next BatchNorm-backward all-reduce is under way
# (engine_bwd.conv_backward); RV3D_HOLD_WGRAD=0: launch them at once, as in the local case
HOLD_WGRAD_FOR_SYNC_BN = os.environ.get("RV3D_HOLD_WGRAD", "1") != "0"


def sync_bn_active() -> bool:
    """True when BatchNorm statistics are all-reduced in this process (SyncBN over > 1 ranks, or the one-rank test of that path)."""
    if SYNC_BN is False or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return False
    return _dist_world() > 1 or (SYNC_WORLD1 and SYNC_BN is True)


# chain mode: a big weight gradient is released before (not after) the backward-data launch of its layer when that launch's last round of
# persistent tiles fills less than this fraction of the CUs (rv-waymo: W = 2656 gives 1328 tiles = 5.19 rounds); 0 = never
EARLY_WGRAD_FILL = float(os.environ.get("RV3D_EARLY_WGRAD_FILL", "0.9"))
OVERLAP_MAX_TFLOP: Optional[float] = ((float(_OVERLAP.split(":")[1]) if ":" in _OVERLAP else 0.1)
                                      if (_OVERLAP.startswith("small") or _OVERLAP.startswith("chain")) else None)
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


_CU_COUNT: Dict[int, int] = {}


def cu_count(device) -> int:
    """Compute units of ``device`` (256 on an MI355X) -- from the device properties, not a literal."""
    idx = torch.device(device).index or 0
    if idx not in _CU_COUNT:
        _CU_COUNT[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    return _CU_COUNT[idx]


def side_stream(device) -> "torch.cuda.Stream":
    idx = torch.device(device).index or 0
    if idx not in _SIDE_STREAMS:
        # high priority (-1): a weight gradient that is ready takes the CUs before the backward-data launch behind it on the main stream --
        # the order the dependencies give anyway (priority 0: 94.08 / 95.50 against 93.99 / 94.69 ms, profiles/r04_ab_notes.md)
        prio = -1
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=device, priority=prio)
    return _SIDE_STREAMS[idx]


# (Measured and dropped: the forward pass of a head's two towers on two streams.  Round 4, free-running: 0.6 ms per rv-av2 step slower
#  (two persistent tapconv6 launches take the CUs from each other).  Round 5, CHAINED by events so that the convs take strict turns
#  and only the other tower's bandwidth-bound write-out pass runs beside a conv: 98.53 against 97.75 ms (rv-av2), 54.48 against 54.19
#  (rv-waymo), three interleaved rounds on one box -- the write-out beside a power-capped conv costs more than it hides.)


def _launch(name: str, flops: float, fn, nbytes: float = 0.0) -> None:
    if PROFILE is not None:
        PROFILE.launch(name, flops, fn, nbytes)
    else:
        fn()


def tap_kernel_name(geom, shape, scatter: bool) -> str:
    info = (ctypes.c_int32 * 4)()
    L.call("rv_tap_launch_info", ctypes.byref(geom), ctypes.byref(shape), L.i32(1 if scatter else 0), info)
    if info[0] in (4, 5, 6):
        name = f"tapconv{info[0]}_kernel<{info[1]}>"
    elif info[0] in (2, 3):
        name = f"tapconv{info[0]}_kernel<{info[1]}>"
    elif info[0] == 7:
        name = f"pointwise_kernel<{info[1]}>"
    else:
        name = f"tapconv_kernel<{info[1] // 16},{info[1] % 16}>"
    if os.environ.get("RV3D_PROFILE_SHAPES"):
        name += f" {'S' if scatter else 'G'} k{geom.kh}x{geom.kw}s{geom.stride_w} {geom.cu}<->{geom.cv} {shape.N}x{shape.H}x{shape.Wu} f{shape.flags}"
    return name


def tap_flops(geom, shape) -> float:
    return 2.0 * shape.N * shape.H * shape.Wu * geom.kh * geom.kw * geom.cu * geom.cv


def tap_bytes(geom, shape, wgrad: bool = False) -> float:
    """Algorithmic bytes of one tap-conv launch: the coarse and the fine tensor once each (bf16, padded channels);
    a weight gradient reads both and writes the fp32 parameter gradient."""
    px = float(shape.N) * shape.H
    act = 2.0 * px * (shape.Wu * pad32(geom.cu) + shape.Wv * pad32(geom.cv))
    return act + (4.0 * geom.kh * geom.kw * geom.cu * geom.cv if wgrad else 2.0 * geom.kh * geom.kw * pad32(geom.cu) * pad32(geom.cv))


def all_reduce_(t: Tensor) -> None:
    """Sum ``t`` over the ranks of the default process group, in place: ``torch.distributed.all_reduce`` (RCCL on the GPUs, gloo
    in the CPU tests).  ``RV3D_DIRECT_RCCL=1`` opts in to the direct binding (``rccl.py``: ``ncclAllReduce`` enqueued on the
    current compute stream, no second stream and no event round trip -- these are latency-bound collectives on the critical
    path); opt-in because only a one-rank communicator of it could ever be tested on this one-GPU build."""
    from . import rccl

    if t.is_cuda and rccl.available():
        rccl.all_reduce_(t)
    else:
        torch.distributed.all_reduce(t)


def _totals_into(partial: Tensor, rows: int, count: int, region: Tensor, local_copy: Optional[Tensor] = None) -> None:
    """region[: 2C] = column sums of ``rows`` partial rows, region[2C] = count (one launch: ``rv_reduce_rows_count``)."""
    if count >= 1 << 24:
        raise L.RvError("SyncBN: more than 2^24 elements per channel on one rank (the count travels as fp32)")
    n = region.numel() - 1
    if partial.is_cuda:
        L.call("rv_reduce_rows_count", L.ptr(partial), L.i32(rows), L.i32(n), L.f32(float(count)), L.ptr(region), L.ptr(local_copy), L.stream_ptr())
    else:  # the gloo tests of the host logic run this function on CPU tensors
        region[:n] = partial[:rows].double().sum(dim=0).float().view(-1)
        region[n] = float(count)
        if local_copy is not None:
            local_copy.view(-1)[:n] = region[:n]


def allreduce_partial_rows(partial: Tensor, rows: int, count: int, local_copy: Optional[Tensor] = None) -> Tensor:
    """Column-sum ``rows`` partial-statistics rows on the device (fp64 accumulation), append this rank's element count, and
    all-reduce the (2*C + 1) totals in ONE collective over the default process group (RCCL on the GPUs; gloo in the CPU
    tests).  Returns a flat (2*C + 1) buffer, ready for ``rv_bn_finalize`` / ``rv_bn_bwd_finalize`` with ``rows = 1,
    count = -1``: the global count sits in the slot behind the totals and is read on the device (ranks may hold different
    numbers of pixels: an uneven last batch).  ``local_copy`` (2, C): receives this rank's own totals."""
    c2 = partial.shape[1] * partial.shape[2]
    flat = torch.empty(c2 + 1, dtype=torch.float32, device=partial.device)
    _totals_into(partial, rows, count, flat, local_copy)
    COLLECTIVES.add(flat)
    all_reduce_(flat)
    return flat


def allreduce_partial_rows_many(items: Sequence[Tuple[Tensor, int, int, Optional[Tensor]]]) -> List[Tensor]:
    """``allreduce_partial_rows`` for several BatchNorm layers at once: item = (partial rows, rows, this rank's count, local
    copy or None).  The (2*C_i + 1) totals of every layer are packed back to back into ONE fp32 buffer and travel in ONE
    collective -- layers whose statistics become available together (the two convs a BasicBlock applies to the same input,
    layer i of the classification and the regression tower) share a latency-bound all-reduce instead of paying one each.
    Returns one flat (2*C_i + 1) view per item."""
    sizes = [p.shape[1] * p.shape[2] + 1 for p, _, _, _ in items]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 3) // 4 * 4  # 16-byte aligned regions
    flat = torch.empty(total, dtype=torch.float32, device=items[0][0].device)  # (the pad floats between regions are never read)
    views = []
    for (partial, rows, count, copy), o, n in zip(items, offs, sizes):
        v = flat[o : o + n]
        _totals_into(partial, rows, count, v, copy)
        views.append(v)
    COLLECTIVES.add(flat)
    all_reduce_(flat)
    return views


class GradSync:
    """Data-parallel gradient averaging for the fused programs, without ``DistributedDataParallel``.

    The model's backward is THREE autograd nodes (backbone + stem, the tower pair, the loss): per-parameter hooks, bucket copies
    and per-bucket divisions -- DDP's machinery for graphs of thousands of nodes -- cost ~380 tiny kernels per step here
    (+3.4 ms on one rank, profiles/r03_syncbn_collectives.md).  Instead: ONE flat fp32 buffer holds every parameter's gradient;
    when a program node's backward has finished (``program._ProgramFn.backward``) its gradients are copied into their views of
    that buffer by one multi-tensor copy and the node's contiguous range is all-reduced asynchronously (RCCL through
    ``torch.distributed``: the towers' 75 MB travel while the backbone's backward runs, the backbone's 63 MB are exposed at the
    end, as under DDP); ``finish()`` waits, divides by the world size and points ``p.grad`` at the views.  No gradient
    accumulation over several backward passes (the flat buffer is overwritten per step) -- the recipe has none."""

    def __init__(self, params: Sequence[nn.Parameter], world: Optional[int] = None) -> None:
        self.params = [p for p in params if p.requires_grad]
        self.world = world if world is not None else _dist_world()
        self.offsets: Dict[int, Tuple[int, int]] = {}
        total = 0
        for p in self.params:
            self.offsets[id(p)] = (total, p.numel())
            total += (p.numel() + 63) // 64 * 64  # 256-byte aligned views
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._views: Dict[int, Tensor] = {}
        self.works: List[tuple] = []
        self.pending: List[nn.Parameter] = []  # parameters whose gradient sits in the flat buffer since the last finish()
        self.bytes = 4 * sum(p.numel() for p in self.params)

    def view(self, p: nn.Parameter) -> Tensor:
        """This parameter's region of the flat buffer, shaped like it.  ONE tensor object per parameter for the life of the
        GradSync (autograd never sees these: ``finish()`` installs them): making 238 views per step was 2-3 ms of host time with
        the GPU idle at the end of the backward pass (profiles/r04_syncbn_collectives.md)."""
        v = self._views.get(id(p))
        if v is None:
            o, n = self.offsets[id(p)]
            v = self._views[id(p)] = self.flat[o : o + n].view_as(p)
        return v

    def broadcast_parameters(self, module: nn.Module) -> None:
        """Rank 0's parameters and buffers to every rank, once (what DDP does at construction)."""
        for t in list(module.parameters()) + list(module.buffers()):
            torch.distributed.broadcast(t.data, src=0)

    def reduce_node(self, params: Sequence[nn.Parameter], grads: Sequence[Optional[Tensor]]) -> List[Optional[Tensor]]:
        """Gradients of the parameters of ONE finished autograd node -> their flat views, all-reduce started.  Returns the list
        with ``None`` in place of the gradients it took over: ``finish()`` installs the views as ``p.grad`` (handing the views to
        autograd here would make AccumulateGrad clone every one of them -- they are referenced from this object -- or, with a
        view already installed from the previous step, add the buffer to itself).
        The node's parameters need not be contiguous in ``model.parameters()`` order (a DetectionHead with several strides or
        tasks registers all classification towers before the regression towers): the flat range is cut into maximal runs of
        parameters that all belong to THIS node and each run travels in its own all-reduce, so no region is reduced while it
        still holds another node's stale values, and every region is scaled exactly once."""
        out = list(grads)
        idx = [i for i, (p, g) in enumerate(zip(params, grads)) if g is not None and id(p) in self.offsets]
        if not idx:
            return out
        pending = {id(p) for p in self.pending}
        for i in idx:
            if id(params[i]) in pending:
                raise L.RvError("GradSync: a parameter was reduced by two program nodes in one step -- its region of the flat buffer may still "
                                "be travelling in the first node's all-reduce (share the parameter inside ONE program, or call finish() between)")
            if params[i].grad is not None and params[i].grad is self._views.get(id(params[i])):
                raise L.RvError("GradSync: p.grad still is last step's view of the flat buffer -- call zero_grad(set_to_none=True) "
                                "before backward (gradient accumulation over several backward passes is not supported)")
        dst = [self.view(params[i]) for i in idx]
        torch._foreach_copy_(dst, [grads[i].to(torch.float32) for i in idx])
        spans = sorted((self.offsets[id(params[i])][0], (self.offsets[id(params[i])][1] + 63) // 64 * 64) for i in idx)
        runs: List[List[int]] = []
        for o, n in spans:  # (views are 256-byte aligned and laid out back to back: adjacent parameters touch exactly)
            if runs and runs[-1][1] == o:
                runs[-1][1] = o + n
            else:
                runs.append([o, o + n])
        live = torch.distributed.is_available() and torch.distributed.is_initialized()  # (also with ONE rank: the one-GPU test of the path)
        for lo, hi in runs:
            seg = self.flat[lo:hi]
            self.works.append((torch.distributed.all_reduce(seg, async_op=True) if live else None, seg))
        for i in idx:
            self.pending.append(params[i])
            out[i] = None
        return out

    def finish(self) -> None:
        """Before the optimizer step: wait for the collectives, average, and make ``p.grad`` of every parameter reduced this
        step its view of the flat buffer.  Until then ``p.grad`` of those parameters is None: anything that reads gradients
        (``clip_grad_norm_``, gradient-norm logging, Lightning's ``on_after_backward``) must run AFTER ``finish()``."""
        for work, seg in self.works:
            if work is not None:
                work.wait()
            if self.world > 1:
                seg.mul_(1.0 / self.world)
        self.works.clear()
        for p in self.pending:
            p.grad = self.view(p)
        self.pending.clear()


GRAD_SYNC: Optional[GradSync] = None


class _CollectiveLog:
    """Per-step census of the collectives the engine itself issues (bench.py reports it in ``config.collectives``)."""

    def __init__(self) -> None:
        self.calls, self.bytes = 0, 0

    def add(self, t: Tensor) -> None:
        self.calls += 1
        self.bytes += t.numel() * t.element_size()

    def reset(self) -> None:
        self.calls, self.bytes = 0, 0


COLLECTIVES = _CollectiveLog()


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def _require_cuda(t: Tensor, what: str) -> None:
    if not t.is_cuda:
        raise L.RvError(
            f"{what} is on {t.device}: the range_view_3d_detection_amd modules only run on an MI355X "
            "(HIP kernels, no CPU fallback)."
        )


# ---------------------------------------------------------------------------------------------
# device tensors
# ---------------------------------------------------------------------------------------------
class Act:
    """NHWC bf16 activation (or activation gradient): ``data`` is (N,H,W,C) with strides (H*W*ld, W*ld, ld, 1)."""

    __slots__ = ("data", "c", "parent", "c0", "_rv_owned")

    def __init__(self, data: Tensor, c: Optional[int] = None, parent: Optional["Act"] = None, c0: int = 0) -> None:
        assert data.dtype in (torch.bfloat16, torch.float16) and data.dim() == 4 and data.stride(3) == 1
        self.data = data
        self.c = data.shape[3] if c is None else c  # logical channels (<= stored, stored is a multiple of 32)
        self.parent = parent
        self.c0 = c0
        self._rv_owned = False  # True: a gradient buffer private to the tape (safe to accumulate into)

    @staticmethod
    def empty(n: int, h: int, w: int, c: int, device, zero: bool = False) -> "Act":
        cp = pad32(c)
        data = (torch.zeros if zero else torch.empty)((n, h, w, cp), dtype=L.act_dtype(), device=device)
        return Act(data, c)

    @property
    def N(self) -> int:
        return self.data.shape[0]

    @property
    def H(self) -> int:
        return self.data.shape[1]

    @property
    def W(self) -> int:
        return self.data.shape[2]

    @property
    def cp(self) -> int:
        return self.data.shape[3]

    @property
    def ld(self) -> int:
        return self.data.stride(2)

    @property
    def pixels(self) -> int:
        return self.N * self.H * self.W

    def ptr(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(self.data.data_ptr())

    def slice(self, c0: int, c1: int) -> "Act":
        assert c0 % 32 == 0 and c1 % 32 == 0
        return Act(self.data[..., c0:c1], c1 - c0, parent=self, c0=c0)

    def like(self, zero: bool = False) -> "Act":
        return Act.empty(self.N, self.H, self.W, self.c, self.data.device, zero)

    def nchw(self) -> Tensor:
        """(N,C,H,W) channels_last *view* of the logical channels -- what user code sees."""
        return self.data[..., : self.c].permute(0, 3, 1, 2)

    @staticmethod
    def from_nchw(x: Tensor) -> "Act":
        """Any (N,C,H,W) CUDA tensor -> Act (zero-copy when it already is channels_last with C % 32 == 0 in the operand type of
        the program being built: bf16, or fp16 for an eval program under autocast(float16))."""
        n, c, h, w = x.shape
        nhwc = x.permute(0, 2, 3, 1)
        if x.dtype == L.act_dtype() and c % 32 == 0 and nhwc.is_contiguous():
            return Act(nhwc, c)
        out = Act.empty(n, h, w, c, x.device, zero=(c % 32 != 0))
        out.data[..., :c].copy_(nhwc)
        return out


@dataclass(eq=False)
class BnState:
    """Folded BatchNorm of one layer for the current step."""

    module: nn.BatchNorm2d
    scale: Tensor
    shift: Tensor
    mean: Optional[Tensor] = None
    invstd: Optional[Tensor] = None
    count: int = 0


@dataclass(eq=False)
class Lazy:
    """``relu?(scale * raw + shift)`` that is never written to HBM: consumers fold it into their operand load."""

    raw: Act
    bn: BnState
    relu: bool = True
    plain: Optional[Act] = None  # written out once when a consumer cannot fold it (see ConvOp)

    def materialized(self) -> Act:
        if self.plain is None:
            out = self.raw.like()
            L.call("rv_ew_combine", L.i64(self.raw.pixels), L.i32(self.raw.cp), self.raw.ptr(), L.i32(self.raw.ld), L.ptr(self.bn.scale),
                   L.ptr(self.bn.shift), None, L.i32(0), None, None, out.ptr(), L.i32(out.ld), L.i32(L.EW_RELU_A if self.relu else 0),
                   L.stream_ptr())
            self.plain = out
        return self.plain


Operand = Union[Act, Lazy]

# The LDS-DMA tap-conv (tapconv4) streams its operands global -> LDS without passing through registers, so it cannot
# apply a folded BatchNorm(+ReLU) on the way in.  On the layers it is eligible for it is enough faster than the
# register-staged kernels (3x3 512 -> 512: 1330 vs 980 TFLOP/s, one box) to pay for writing the operand out once
# (one HBM-bound pass); forward conv, and the weight gradient in backward, then both read the plain tensor.
MATERIALIZE_FOR_DMA = True  # (module attribute: tests and A/B tools flip it in-process; no environment switch)
# (Measured and dropped in round 4: the split-K sums of all weight gradients in ONE batched launch at the end of a program's backward --
#  bit-identical, 76 launches fewer, 1.0-1.6 ms per step SLOWER: the immediate reduction reads slabs that are still in the Infinity Cache.)
# conv -> BatchNorm(+ReLU) -> conv: the second conv's backward-data launch also forms the BatchNorm-backward sums (rv_tap_data_grad_bnb)
BNB_FUSE = True
# (Measured and dropped, rounds 3-4: the LAST writer of a block output's gradient forming those sums over the complete gradient in a
#  masked epilogue -- three 16-byte prefetches per pass, time-neutral to slightly slower; profiles/r04_ab_notes.md.)


def _dma_generation(geom, n: int, h: int, wu: int, wv: int, ld_src: int, ld_dst: int, scatter: bool) -> int:
    """Kernel generation the library would pick for this launch on a PLAIN bf16 operand (rv_tap_launch_info)."""
    info = (ctypes.c_int32 * 4)()
    shape = L.TapShape(n, h, wu, wv, ld_src, ld_dst, 0)
    return info[0] if L.load().rv_tap_launch_info(ctypes.byref(geom), ctypes.byref(shape), 1 if scatter else 0, info) == 0 else 0


def _dma_eligible(geom, n: int, h: int, wu: int, wv: int, ld_src: int, ld_dst: int, scatter: bool) -> bool:
    return _dma_generation(geom, n, h, wu, wv, ld_src, ld_dst, scatter) in (4, 5, 6)


# 1x1 C -> C layers fed by a folded BatchNorm+ReLU (the stem's second fusion conv): written out once as well when the plain launch runs on the
# pointwise streaming GEMM (generation 7, round 6) -- write-out + streaming GEMM + wgrad3 on the plain operand against the register-staged
# tapconv2 + wgrad2 (profiles/r06_ab_notes.md).  Other 1x1 shapes keep the folded operand (the tiled kernel saves what the pass costs).
MATERIALIZE_FOR_POINTWISE = True
MATERIALIZE_POINTWISE_C = (256,)  # input widths that take it (128: time-neutral on rv-waymo, +0.27 ms on rv-av2 -- profiles/r06_ab_notes.md section 4; profiles/tools/diag_waymo_stem.py adds it back)


# ---------------------------------------------------------------------------------------------
# conv layers: geometry + packed weights
# ---------------------------------------------------------------------------------------------
_LAYERS: "weakref.WeakSet[TapLayer]" = weakref.WeakSet()
_PACK_TABLES: Dict[tuple, Tensor] = {}
BATCH_PACK = True
# Strided layers (stride-2 convs, the ConvTranspose2d up-samplers): run their strided gather and their weight gradient on the
# stride-1 FOLDED view (fine tensor read as (N, H, W/s, s*C)) so that the LDS-DMA kernels take them.
FOLD_STRIDED = True


def prepack_stale() -> None:
    """Re-pack every layer whose parameter changed since its images were made (= all of them after an optimiser step) in ONE
    launch (``rv_pack_batch``) instead of one ``rv_pack_weight`` launch per layer when the layer is next used (~160 launches
    per training step on the rv-* models).  Called at the start of a training-mode program; layers seen for the first time,
    layers with a permuted weight (``in_perm``) and non-fp32 parameters stay on the lazy per-layer path.  Both images of a
    layer are written (a layer whose input needs no gradient never reads its scatter image: a few small ones)."""
    if not BATCH_PACK:  # (no grad-mode test here: inside an autograd.Function's forward grad mode is off)
        return
    stale = []
    for l in _LAYERS:
        w = l.weight
        if (l._version is None or l.in_perm is not None or not w.requires_grad or not w.is_cuda or w.dtype != torch.float32
                or not w.is_contiguous() or not l._packed):
            continue
        if (w._version, w.data_ptr()) != l._version:
            stale.append(l)
    if len(stale) < 2:
        return
    stale.sort(key=id)
    dev = stale[0].weight.device
    for l in stale:
        if not l._images:
            n = L.load().rv_packed_weight_bytes(ctypes.byref(l.geom)) // 2
            l._images = {f: torch.empty(n, dtype=torch.bfloat16, device=dev) for f in ("gather", "scatter")}
    folded = [l for l in stale if l._fold_image is not None]  # (layers whose folded image has been used at least once)
    key = tuple((id(l), l.weight.data_ptr(), l._images["gather"].data_ptr(), l._images["scatter"].data_ptr(),
                 l._fold_image.data_ptr() if l._fold_image is not None else 0) for l in stale)
    n_entries = 2 * len(stale) + len(folded)
    table = _PACK_TABLES.get(key)
    if table is None:
        eb = L.load().rv_pack_batch_entry_bytes()
        host = (ctypes.c_uint8 * (eb * n_entries))()
        for i, l in enumerate(stale):
            L.call("rv_pack_batch_fill", ctypes.byref(l.geom), L.ptr(l.weight), L.ptr(l._images["gather"]), L.ptr(l._images["scatter"]),
                   ctypes.byref(host, 2 * eb * i))
        for i, l in enumerate(folded):
            L.call("rv_pack_batch_fill_folded", ctypes.byref(l.geom), L.ptr(l.weight), L.ptr(l._fold_image), ctypes.byref(host, eb * (2 * len(stale) + i)))
        _PACK_TABLES.clear()  # one live table (a second model in the process rebuilds it: cheap)
        table = torch.frombuffer(host, dtype=torch.uint8).clone().to(dev)
        _PACK_TABLES[key] = table
    L.call("rv_pack_batch", L.ptr(table), L.i32(n_entries), L.stream_ptr())
    for l in stale:
        l._packed = dict(l._images)
        l._version = (l.weight._version, l.weight.data_ptr())
    for l in folded:
        l._fold_version = l._version


class TapLayer:
    """Geometry and packed bf16 weight images of one nn.Conv2d / nn.ConvTranspose2d parameter."""

    def __init__(self, weight: nn.Parameter, stride_w: int, pad: Tuple[int, int], transposed: bool,
                 bias: Optional[nn.Parameter] = None, in_perm: Optional[Tuple[int, int]] = None) -> None:
        self.weight = weight
        self.bias = bias
        self.transposed = transposed
        self.in_perm = in_perm  # (C, taps): MetaKernel fusion conv, reference channel order c*taps+k -> k*Cpad+c
        cu, cv, kh, kw = weight.shape
        if in_perm is not None:
            c, taps = in_perm
            cv = taps * pad32(c)
        self.geom = L.TapGeom(kh, kw, stride_w, pad[0], pad[1], cu, cv)
        self.cu, self.cv = cu, cv
        self._packed: Dict[str, Tensor] = {}
        self._version = None
        self._images: Dict[str, Tensor] = {}  # persistent buffers of the batched re-pack (prepack_stale)
        self._fold_geom = None
        self._fold_image: Optional[Tensor] = None
        self._fold_version = None
        self._bias_pad: Optional[Tuple[tuple, Tensor]] = None
        _LAYERS.add(self)

    # forward direction of the torch module: conv = gather, conv-transpose = scatter
    @property
    def fwd_form(self) -> str:
        return "scatter" if self.transposed else "gather"

    @property
    def c_in(self) -> int:
        return self.cu if self.transposed else self.cv

    @property
    def c_out(self) -> int:
        return self.cv if self.transposed else self.cu

    def _torch_weight(self, out_scale: Optional[Tensor] = None) -> Tensor:
        """fp32 weight in the layout the packers take; ``out_scale`` (c_out,) multiplies every OUTPUT channel first (an eval-mode
        BatchNorm folded into the weights)."""
        w = self.weight.detach()
        if out_scale is not None:
            w = w.float() * (out_scale.view(1, -1, 1, 1) if self.transposed else out_scale.view(-1, 1, 1, 1))
        if self.in_perm is not None:
            c, taps = self.in_perm
            cu = w.shape[0]
            w = w.reshape(cu, c, taps).permute(0, 2, 1)  # [cu][k][c]
            w = torch.nn.functional.pad(w, (0, pad32(c) - c)).reshape(cu, taps * pad32(c), 1, 1)
        return w.contiguous().float()

    def unpermute_grad(self, g: Tensor) -> Tensor:
        """Packed-order gradient [cu][cv_eff][kh][kw] -> the parameter's own layout."""
        if self.in_perm is None:
            return g
        c, taps = self.in_perm
        cu = g.shape[0]
        return g.reshape(cu, taps, pad32(c))[..., :c].permute(0, 2, 1).reshape(self.weight.shape).contiguous()

    # ---- folded (stride-1) form of a strided layer: csrc/misc.hip rv_fold_geom --------------------------------------
    def fold_geom(self) -> Optional["L.TapGeom"]:
        """Geometry of the stride-1 view of this layer's strided gather / weight gradient (None for stride-1 layers)."""
        if self.geom.stride_w == 1 or not FOLD_STRIDED or self.in_perm is not None:
            return None
        if self._fold_geom is None:
            gf = L.TapGeom()
            L.call("rv_fold_geom", ctypes.byref(self.geom), ctypes.byref(gf))
            self._fold_geom = gf
        return self._fold_geom

    def packed_folded(self) -> Tensor:
        """bf16 gather image of the folded form; re-packed with the other images when the parameter changed."""
        if L.operand_tag() != "bf16":  # (eval under autocast(float16): its own image, re-made when the parameter changed)
            ver = (self.weight._version, self.weight.data_ptr())
            hit = self.__dict__.get("_fold_f16")
            if hit is None or hit[0] != ver:
                n = L.load().rv_packed_weight_bytes(ctypes.byref(self.fold_geom())) // 2
                img = torch.empty(n, dtype=L.act_dtype(), device=self.weight.device)
                L.call("rv_pack_weight_folded", ctypes.byref(self.geom), L.ptr(self.weight.detach().contiguous().float()), L.ptr(img), L.stream_ptr())
                self.__dict__["_fold_f16"] = hit = (ver, img)
            return hit[1]
        ver = (self.weight._version, self.weight.data_ptr())
        if ver != self._fold_version or self._fold_image is None:
            gf = self.fold_geom()
            n = L.load().rv_packed_weight_bytes(ctypes.byref(gf)) // 2
            if self._fold_image is None:
                self._fold_image = torch.empty(n, dtype=torch.bfloat16, device=self.weight.device)
            L.call("rv_pack_weight_folded", ctypes.byref(self.geom), L.ptr(self.weight.detach().contiguous().float()), L.ptr(self._fold_image), L.stream_ptr())
            self._fold_version = ver
        return self._fold_image

    def packed_eval(self, form: str, bn: nn.Module) -> Tuple[Tensor, Tensor]:
        """Inference image of conv -> eval-mode BatchNorm: (weight image with gamma / sqrt(var + eps) folded into every output
        channel, padded bias beta - mean * scale), cached until the conv weight or any BatchNorm tensor changes.  The fold is
        a handful of small fp32 torch ops once per checkpoint, not per forward."""
        key = (self.weight._version, self.weight.data_ptr(), bn.weight._version, bn.bias._version,
               bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr())
        cache = self.__dict__.setdefault("_eval_fold", {})
        hit = cache.get((form, L.operand_tag()))
        if hit is None or hit[0] != key:
            c, cp = self.c_out, pad32(self.c_out)
            scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
            shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
            w = self._torch_weight(out_scale=scale.float())
            geom = self.fold_geom() if form == "folded" else self.geom
            nbytes = L.load().rv_packed_weight_bytes(ctypes.byref(geom))
            img = torch.empty(nbytes // 2, dtype=L.act_dtype(), device=self.weight.device)
            if form == "folded":
                L.call("rv_pack_weight_folded", ctypes.byref(self.geom), L.ptr(w.contiguous()), L.ptr(img), L.stream_ptr())
            else:
                L.call("rv_pack_weight", ctypes.byref(self.geom), L.ptr(w), L.ptr(img if form == "gather" else None),
                       L.ptr(img if form == "scatter" else None), L.stream_ptr())
            bias = torch.zeros(cp, dtype=torch.float32, device=self.weight.device)
            bias[:c] = shift.float()
            cache[(form, L.operand_tag())] = hit = (key, img, bias)
        return hit[1], hit[2]

    def padded_bias(self) -> Tensor:
        """fp32 bias padded to the 32-channel slab, cached until the parameter changes (two launches per step and biased layer otherwise)."""
        b = self.bias
        key = (b._version, b.data_ptr())
        if self._bias_pad is None or self._bias_pad[0] != key:
            self._bias_pad = (key, torch.nn.functional.pad(b.detach().float(), (0, pad32(self.c_out) - self.c_out)))
        return self._bias_pad[1]

    def invalidate(self) -> None:
        """Drop the packed bf16 images.  They are re-packed automatically when ``weight._version`` or its storage changes
        (optimizer steps, ``load_state_dict``, ``copy_``); writes that bypass the version counter (``p.data.add_(...)``,
        weight-averaging swaps through ``.data``) do not show there -- call this (or ``engine.invalidate_packed_weights``)
        after such surgery."""
        self._packed.clear()
        self._version = None
        self._fold_version = None

    def packed(self, form: str) -> Tensor:
        """bf16 weight image for ``form``; re-packed when the parameter changed (optimizer step / load_state_dict)."""
        ver = (self.weight._version, self.weight.data_ptr())
        if ver != self._version:
            self._packed.clear()
            self._version = ver
        if L.operand_tag() != "bf16":  # fp16 images (inference under autocast(float16)): their own cache entries, packed by that build
            key = form + ":" + L.operand_tag()
            if key not in self._packed:
                nbytes = L.load().rv_packed_weight_bytes(ctypes.byref(self.geom))
                buf = torch.empty(nbytes // 2, dtype=L.act_dtype(), device=self.weight.device)
                L.call("rv_pack_weight", ctypes.byref(self.geom), L.ptr(self._torch_weight()), L.ptr(buf if form == "gather" else None),
                       L.ptr(buf if form == "scatter" else None), L.stream_ptr())
                self._packed[key] = buf
            return self._packed[key]
        if form not in self._packed:
            nbytes = L.load().rv_packed_weight_bytes(ctypes.byref(self.geom))
            w = self._torch_weight()
            # a training step needs both images (forward in one form, backward-data in the other): one launch writes both
            forms = ("gather", "scatter") if (torch.is_grad_enabled() and self.weight.requires_grad and "gather" not in self._packed and "scatter" not in self._packed) else (form,)
            bufs = {f: torch.empty(nbytes // 2, dtype=torch.bfloat16, device=self.weight.device) for f in forms}
            L.call("rv_pack_weight", ctypes.byref(self.geom), L.ptr(w), L.ptr(bufs.get("gather")), L.ptr(bufs.get("scatter")), L.stream_ptr())
            self._packed.update(bufs)
        return self._packed[form]


# ---------------------------------------------------------------------------------------------
# tape
# ---------------------------------------------------------------------------------------------
class Tape:
    """Ops of one forward pass + gradient bookkeeping of the matching backward pass."""

    def __init__(self, training: bool, device) -> None:
        self.training = training
        self.device = device
        # eval-mode programs take the inference forms (BatchNorm folded into the weights, block sums and the stem's modulation in
        # conv epilogues).  They are INFERENCE-ONLY: BatchNorm backward is built on batch statistics, so a backward pass through
        # an eval-mode program raises RvError (fine-tuning with frozen statistics is not on the reference's path: its trainer
        # always runs training_step in train mode, nn/arch/detector.py:238-247).  Not switched on "does anything require grad":
        # an eval forward outside torch.no_grad() would silently lose the fused forms.
        self.inference = not training
        self.ops: List["Op"] = []
        # backward state
        self.grads: Dict[int, Act] = {}          # id(root Act) -> gradient Act (same padded shape)
        self.written: set = set()                # ids of gradient Acts (roots) already holding a value
        self.lazy_in: Dict[int, Tuple[Act, Optional[Act]]] = {}  # id(Lazy) -> (dOut, OUT mask source)
        self.lazy_sums: Dict[int, Tuple[Tensor, int]] = {}  # id(Lazy) -> BatchNorm-backward partial sums its ONE writer formed, rows
        self.meta_in: Dict[int, tuple] = {}      # id(Lazy) -> (dgeo, feat, partial sums, rows): MetaKernel modulation fused into the BatchNorm backward
        self.head_final: Dict[int, tuple] = {}   # id(Lazy) -> (call head, BatchNorm-backward partial sums, rows, dY of the tower's final conv, its scatter image)
        self.producers: Dict[int, "Op"] = {}     # id(block-output Act) -> the CombineOp that made it
        self.bn_of: Dict[int, "Op"] = {}         # id(Lazy) -> its BnOp
        self.raw_grad: Dict[int, Act] = {}       # id(raw Act) -> gradient w.r.t. the raw conv output
        self._param_grads: Dict[int, Tensor] = {}  # id(param) -> fp32 gradient (read through `param_grads`)
        self.params: Dict[int, nn.Parameter] = {}
        self.used_side_stream = False
        self.chained_wgrad = None  # (RV3D_OVERLAP=chain) event behind the last big weight gradient on the side stream
        self.held_wgrads: List = []  # weight-gradient launches held back for a SyncBN all-reduce (engine_bwd._release_held_wgrads)
        self.bn_counters: List[Tensor] = []

    # ---- gradient buffers (views follow their parents) ----
    def grad_buffer(self, a: Act) -> Tuple[Act, bool]:
        """Gradient Act for ``a`` and whether it already holds a value (=> accumulate)."""
        root, chain = a, []
        while root.parent is not None:
            chain.append(root)
            root = root.parent
        key = id(root)
        if key not in self.grads:
            self.grads[key] = root.like(zero=bool(chain))  # partial (view) writers need a defined background
            if chain:
                self.written.add(key)
        g = self.grads[key]
        have = key in self.written
        for v in reversed(chain):
            g = g.slice(v.c0, v.c0 + v.cp)
        return g, have

    def mark_written(self, a: Act) -> None:
        root = a
        while root.parent is not None:
            root = root.parent
        self.written.add(id(root))

    def set_grad(self, a: Act, g: Act) -> None:
        assert a.parent is None
        self.grads[id(a)] = g
        self.written.add(id(a))

    # ---- gradients of Lazy operands (unmaterialised relu(bn(conv)) outputs) ----
    def lazy_grad_target(self, lazy: "Lazy") -> Tuple[Act, bool]:
        """Buffer a consumer's backward-data kernel writes the gradient w.r.t. ``lazy``'s (activated) value into, and whether
        it must ACCUMULATE: a Lazy may feed several consumers (conv + projection conv of a BasicBlock, say), whose
        contributions add up -- a second writer folds the pending entry into one plain buffer first."""
        key = id(lazy)
        prev = self.lazy_in.get(key)
        self.lazy_sums.pop(key, None)  # (a first writer re-registers its sums after its launch; a second one voids them)
        if prev is None:
            dst = lazy.raw.like()
            dst._rv_owned = True
            self.lazy_in[key] = (dst, None, None)
            return dst, False
        self._flatten_lazy_grad(key)
        return self.lazy_in[key][0], True

    def add_lazy_grad(self, lazy: "Lazy", dout: Act, mask: Optional[Act], res=None) -> None:
        """Register ``dOut * [mask > 0]`` (mask None: ``dOut``) as (a share of) the gradient w.r.t. ``lazy``'s value; ``res`` =
        (buffer, accumulate) asks the BatchNorm-backward apply pass to also emit that masked gradient for a residual
        branch -- only possible for a single consumer, otherwise it is written here."""
        key = id(lazy)
        self.lazy_sums.pop(key, None)
        if key not in self.lazy_in:
            self.lazy_in[key] = (dout, mask, res)
            return
        self._flatten_lazy_grad(key)
        buf = self.lazy_in[key][0]
        self._masked_into(dout, mask, buf, True)
        if res is not None:
            self._masked_into(dout, mask, res[0], res[1])

    def _masked_into(self, dout: Act, mask: Optional[Act], dst: Act, accumulate: bool) -> None:
        L.call("rv_ew_mask_grad", L.i64(dout.pixels), L.i32(dout.cp), dout.ptr(), L.i32(dout.ld), mask.ptr() if mask is not None else None,
               L.i32(mask.ld if mask is not None else 0), dst.ptr(), L.i32(dst.ld), L.i32(1 if accumulate else 0), L.stream_ptr())

    def _flatten_lazy_grad(self, key: int) -> None:
        dout, mask, res = (self.lazy_in[key] + (None,))[:3]
        if mask is None and res is None and getattr(dout, "_rv_owned", False):
            return  # already a private plain buffer
        buf = dout.like()
        buf._rv_owned = True
        self._masked_into(dout, mask, buf, False)
        if res is not None:
            self._masked_into(dout, mask, res[0], res[1])
        self.lazy_in[key] = (buf, None, None)

    @property
    def param_grads(self) -> Dict[int, Tensor]:
        return self._param_grads

    def add_param_grad(self, p: nn.Parameter, g: Tensor) -> None:
        k = id(p)
        self.params[k] = p
        self._param_grads[k] = self._param_grads[k] + g if k in self._param_grads else g  # (a parameter used by two layers: the sum)

    def backward(self) -> None:
        from . import engine_bwd

        # Consecutive BatchNorm backward ops (layers whose output gradients are all available: conv_bn_many puts their BnOps
        # next to each other) are independent of one another; under SyncBN their (sum g, sum g*xhat) all-reduces travel as ONE
        # collective: every op of the run forms its local sums first, then one all-reduce, then the finalize + apply passes.
        pending: List[tuple] = []
        for op in reversed(self.ops):
            if GROUP_SYNC_BN and isinstance(op, BnOp) and op.sync_world > 1:
                rec = engine_bwd.bn_backward_begin(op, self)
                if rec is not None:
                    pending.append(rec)
                continue
            if pending:
                engine_bwd.bn_backward_finish(pending, self)
                pending = []
            op.backward(self)
        if pending:
            engine_bwd.bn_backward_finish(pending, self)
        engine_bwd._release_held_wgrads(self)
        if self.used_side_stream:  # parameter gradients (and the buffers the side stream read) are final after this
            torch.cuda.current_stream().wait_stream(side_stream(self.device))


class Op:
    def backward(self, t: Tape) -> None:  # pragma: no cover - interface
        raise NotImplementedError


def _operand_parts(x: Operand) -> Tuple[Act, Optional[Tensor], Optional[Tensor], int]:
    if isinstance(x, Lazy):
        return x.raw, x.bn.scale, x.bn.shift, L.IN_AFFINE | (L.IN_RELU if x.relu else 0)
    return x, None, None, 0


# ---------------------------------------------------------------------------------------------
# conv op
# ---------------------------------------------------------------------------------------------
class ConvOp(Op):
    """``out = layer(x)``: tap-conv forward; optional batch statistics, bias, fp32 output."""

    def __init__(self, t: Tape, layer: TapLayer, x: Operand, stats: bool = False, out_f32: bool = False,
                 out: Optional[Act] = None, need_input_grad: bool = True, precomputed: Optional[Tuple[Optional[Tensor], int]] = None,
                 eval_bn: Optional[nn.Module] = None, relu_out: bool = False, residual: Optional[Act] = None,
                 res_relu: bool = False) -> None:
        """``precomputed`` = (partial statistics rows, rows): ``out`` already holds the layer's output (a fused kernel wrote it);
        the op only records what backward needs.  ``eval_bn`` (inference only): the eval-mode BatchNorm behind the conv is folded
        into the weight image and the bias, ``relu_out`` applies the ReLU in the epilogue -- the output is the activation;
        ``residual`` is added to that activation in the same epilogue (``res_relu``: ReLU after the sum) -- rv_tap_residual."""
        self.layer, self.x, self.need_input_grad = layer, x, need_input_grad
        self.eval_bn = eval_bn
        assert residual is None or (eval_bn is not None and not out_f32 and not stats)
        self.pos_first: Optional["SmallKOp"] = None
        src, sc, sh, flags = _operand_parts(x)
        form = layer.fwd_form
        g = layer.geom
        if form == "gather":
            wv, wu = src.W, src.W // g.stride_w
            w_out = wu
        else:
            wu, wv = src.W, src.W * g.stride_w
            w_out = wv
        assert src.cp == pad32(layer.c_in), (src.cp, layer.c_in)
        self.x_plain = None
        # (not for 1x1 layers in general: there the extra pass costs what the faster kernel saves)
        if (isinstance(x, Lazy) and MATERIALIZE_FOR_DMA and not out_f32 and g.kh * g.kw > 1
                and _dma_eligible(g, src.N, src.H, wu, wv, src.ld, pad32(layer.c_out), form == "scatter")):
            self.x_plain = src = x.materialized()
            sc = sh = None
            flags = 0
        elif (isinstance(x, Lazy) and MATERIALIZE_FOR_DMA and MATERIALIZE_FOR_POINTWISE and not out_f32 and g.kh * g.kw == 1 and layer.bias is None
              and eval_bn is None and pad32(layer.c_in) in MATERIALIZE_POINTWISE_C
              and _dma_generation(g, src.N, src.H, wu, wv, pad32(layer.c_in), out.ld if out is not None else pad32(layer.c_out), form == "scatter") == 7):
            # (256 input channels only -- rv-av2's stem.  For 128 input channels (rv-waymo's stem conv, the 1/2 .. 1/8-resolution layers of both models) the
            #  write-out is time-neutral on rv-waymo and +0.27 ms per rv-av2 step: off on that measurement, profiles/r06_ab_notes.md section 4 -- the fault
            #  first met on this route was wgrad3's, fixed there.  Asked with the strides the launch will really have.)
            self.x_plain = src = x.materialized()
            sc = sh = None
            flags = 0
        elif (isinstance(x, Lazy) and MATERIALIZE_FOR_DMA and not out_f32 and form == "gather" and g.stride_w > 1 and g.kh * g.kw > 1
              and layer.fold_geom() is not None and src.ld == src.cp and src.W == g.stride_w * wu
              and _dma_eligible(layer.fold_geom(), src.N, src.H, wu, wu, g.stride_w * src.ld, pad32(layer.c_out), False)):
            # strided multi-tap conv fed by a folded BatchNorm+ReLU: written out once, the FOLDED stride-1 form (forward and
            # weight gradient) then runs on the LDS-DMA kernels instead of the generic strided ones
            self.x_plain = src = x.materialized()
            sc = sh = None
            flags = 0
        self.out_f32 = out_f32
        if out_f32:
            self.out_t = torch.empty((src.N, src.H, w_out, pad32(layer.c_out)), dtype=torch.float32, device=t.device)
            dst_ptr, ld_dst = L.ptr(self.out_t), self.out_t.stride(2)
            flags |= L.OUT_F32
            self.out = None
        else:
            self.out = out if out is not None else Act.empty(src.N, src.H, w_out, layer.c_out, t.device)
            dst_ptr, ld_dst = self.out.ptr(), self.out.ld
        bias = layer.bias
        if eval_bn is not None:
            assert bias is None and not stats and not t.training
            flags |= L.OUT_BIAS | (L.OUT_RELU if relu_out else 0) | (L.OUT_RES_RELU if residual is not None and res_relu else 0)
            bias_p = None  # (filled below, with the folded weight image)
        elif bias is not None:
            flags |= L.OUT_BIAS
            bias_p = layer.padded_bias()
        else:
            bias_p = None
        self.shape = L.TapShape(src.N, src.H, wu, wv, src.ld, ld_dst, flags | (L.OUT_STATS if stats else 0))
        # a strided gather (stride-2 conv forward) runs on the stride-1 FOLDED view when the LDS-DMA kernels take that
        lg, lshape, wp = g, self.shape, None
        gf = layer.fold_geom() if (form == "gather" and g.stride_w > 1 and sc is None and flags & (L.IN_AFFINE | L.IN_RELU) == 0) else None
        # (the folded view reads row pitch wu * s * ld: only when the fine width is exactly s * wu -- an odd width keeps the generic kernel)
        if gf is not None and src.ld == src.cp and src.W == g.stride_w * wu and _dma_eligible(gf, src.N, src.H, wu, wu, g.stride_w * src.ld, ld_dst, False):
            lg = gf
            lshape = L.TapShape(src.N, src.H, wu, wu, g.stride_w * src.ld, ld_dst, self.shape.flags)
            wp = layer.packed_folded() if eval_bn is None else "folded"
        self.partial = None
        self.rows = 0
        if stats:
            self.rows = L.load().rv_tap_stats_rows(ctypes.byref(lg), ctypes.byref(lshape), 1 if form == "scatter" else 0)
            if self.rows < 0:
                raise L.RvError("rv_tap_stats_rows: " + L.load().rv_last_error().decode())
            self.partial = torch.empty((self.rows + L.STATS_SCRATCH_ROWS, 2, pad32(layer.c_out)), dtype=torch.float32,
                                       device=t.device)
        if eval_bn is not None:
            wp, bias_p = layer.packed_eval("folded" if isinstance(wp, str) else form, eval_bn)
        elif wp is None:
            wp = layer.packed(form)
        call = lambda: L.call("rv_tap_" + form, ctypes.byref(lg), ctypes.byref(lshape), src.ptr(), L.ptr(sc), L.ptr(sh),
                              L.ptr(wp), L.ptr(bias_p), dst_ptr, L.ptr(self.partial), L.stream_ptr())
        if residual is not None:
            assert sc is None and flags & (L.IN_AFFINE | L.IN_RELU) == 0 and residual.cp == self.out.cp and residual.pixels == self.out.pixels
            call = lambda: L.call("rv_tap_residual", ctypes.byref(lg), ctypes.byref(lshape), L.i32(1 if form == "scatter" else 0), src.ptr(),
                                  L.ptr(wp), L.ptr(bias_p), residual.ptr(), L.i32(residual.ld), dst_ptr, L.stream_ptr())
        if precomputed is not None:
            assert out is not None and not out_f32 and bias is None
            self.partial, self.rows = precomputed
        elif PROFILE is not None:
            _launch(tap_kernel_name(lg, lshape, form == "scatter"), tap_flops(g, self.shape), call, tap_bytes(g, self.shape))
        else:
            call()
        self.count = src.N * src.H * w_out
        t.ops.append(self)

    def backward(self, t: Tape) -> None:
        from . import engine_bwd  # local import: backward kernels are a separate module

        if self.eval_bn is not None:
            raise L.RvError("this conv ran with an eval-mode BatchNorm folded into its weights (inference): there is no backward")
        engine_bwd.conv_backward(self, t)


# ---------------------------------------------------------------------------------------------
# batch norm
# ---------------------------------------------------------------------------------------------
class BnOp(Op):
    """Finalise batch statistics (train) or fold running statistics (eval) into scale/shift."""

    def __init__(self, t: Tape, conv: ConvOp, bn: nn.BatchNorm2d, relu: bool = True, reduced: Optional[Tensor] = None) -> None:
        """``reduced``: this layer's all-reduced (2*C + 1) totals when the caller grouped its collective with other layers'
        (``conv_bn_many``); None: the op issues its own."""
        self.conv = conv
        self.sync_world = 1
        c = bn.num_features
        cp = pad32(c)
        dev = t.device
        gamma, beta = _padded(bn.weight, cp), _padded(bn.bias, cp)
        scale = torch.empty(cp, dtype=torch.float32, device=dev)
        shift = torch.empty(cp, dtype=torch.float32, device=dev)
        self.gamma_p = gamma
        if t.training:
            mean = torch.empty(cp, dtype=torch.float32, device=dev)
            invstd = torch.empty(cp, dtype=torch.float32, device=dev)
            rm, rv = _padded(bn.running_mean, cp), _padded(bn.running_var, cp, 1.0)
            self.sync_world = bn_sync_world(bn, True)
            count_arg = conv.count
            if self.sync_world > 1:  # SyncBN: (sum, sum of squares, count) in one all-reduce; the kernel reads the global count
                conv.partial = reduced if reduced is not None else allreduce_partial_rows(conv.partial, conv.rows, conv.count)
                conv.rows, count_arg = 1, -1
            L.call("rv_bn_finalize", L.ptr(conv.partial), L.i32(conv.rows), L.i32(cp), L.i64(count_arg), L.ptr(gamma),
                   L.ptr(beta), L.f32(bn.eps), L.f32(bn.momentum if bn.momentum is not None else 0.1), L.ptr(rm),
                   L.ptr(rv), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.stream_ptr())
            if bn.running_mean.shape[0] != cp:  # padded copies: write the logical channels back
                bn.running_mean.copy_(rm[:c])
                bn.running_var.copy_(rv[:c])
            t.bn_counters.append(bn.num_batches_tracked)  # incremented together at the end of the forward pass
            conv.partial = None
            self.state = BnState(bn, scale, shift, mean, invstd, conv.count)
        else:
            rm, rv = _padded(bn.running_mean, cp), _padded(bn.running_var, cp, 1.0)
            L.call("rv_bn_fold_eval", L.i32(cp), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.f32(bn.eps),
                   L.ptr(scale), L.ptr(shift), L.stream_ptr())
            self.state = BnState(bn, scale, shift)
        self.lazy = Lazy(conv.out, self.state, relu)
        t.bn_of[id(self.lazy)] = self  # (backward: the BatchNorm op behind a Lazy operand -- on the tape, not on the Lazy: no reference cycle)
        t.ops.append(self)

    def backward(self, t: Tape) -> None:
        from . import engine_bwd

        engine_bwd.bn_backward(self, t)


def _padded(p: Tensor, cp: int, fill: float = 0.0) -> Tensor:
    p = p.detach()
    if p.shape[0] == cp:
        return p if p.dtype == torch.float32 else p.float()
    out = torch.full((cp,), fill, dtype=torch.float32, device=p.device)
    out[: p.shape[0]] = p
    return out


SMALLK_FORWARD = True


class SmallKOp(Op):
    """``h = relu(BatchNorm(conv1x1(x)))`` for a conv with at most 8 input channels whose input needs no gradient (the
    stem's 3 -> C positional conv on the 9x unfolded grid, the 5/6 -> C feature projection): one element-wise pass
    writes the ACTIVATED output, the batch statistics come in closed form from the moments of ``x``
    (``rv_smallk_forward``), and backward is ``rv_bn_bwd_smallk`` (BatchNorm backward + weight gradient in one pass).
    Returns a plain ``Act`` -- the consumer conv then runs on the LDS-DMA kernels."""

    def __init__(self, t: Tape, layer: TapLayer, x: Act, bn: nn.BatchNorm2d, apply: bool = True) -> None:
        """``apply=False``: only the statistics / folded scale and shift are formed; the caller's fused kernel writes ``out``
        (``PosPair``: rv_pos_forward generates it in the second layer's operand staging)."""
        self.layer, self.x, self.bn = layer, x, bn
        self.sync_world = 1
        self.grads_done = False  # set when the consumer's backward already produced this layer's parameter gradients (pos pair)
        c, cin = bn.num_features, layer.c_in
        cp = pad32(c)
        dev = t.device
        self.gamma_p, self.beta_p = _padded(bn.weight, cp), _padded(bn.bias, cp)
        scale = torch.empty(cp, dtype=torch.float32, device=dev)
        shift = torch.empty(cp, dtype=torch.float32, device=dev)
        self.scale, self.shift = scale, shift  # (backward recomputes y = W x and gates / normalises it with these)
        self.out = Act.empty(x.N, x.H, x.W, c, dev)
        wp = layer.packed("gather")
        rm, rv = _padded(bn.running_mean, cp), _padded(bn.running_var, cp, 1.0)
        self.count = x.pixels
        if t.training:
            self.mean = torch.empty(cp, dtype=torch.float32, device=dev)
            self.invstd = torch.empty(cp, dtype=torch.float32, device=dev)
            ws = torch.empty(L.load().rv_smallk_forward_workspace_bytes(L.i32(cin)), dtype=torch.uint8, device=dev)
            moments = torch.empty(73, dtype=torch.float64, device=dev)
            L.call("rv_smallk_moments", x.ptr(), L.i32(x.ld), L.i64(x.pixels), L.i32(cin), L.ptr(moments), L.ptr(ws), L.stream_ptr())
            self.sync_world = bn_sync_world(bn, True)
            if self.sync_world > 1:
                # SyncBN: the moments are sums over pixels -> all-reduce them together with this rank's pixel count (the slot
                # behind them); the closed form then reads the GLOBAL count on the device (count = -1): no host round trip
                cin_pad = 4 if cin <= 4 else 8
                moments[cin_pad + cin_pad * cin_pad : cin_pad + cin_pad * cin_pad + 1].fill_(float(x.pixels))  # (fill: a kernel argument, not a blocking host-to-device copy)
                COLLECTIVES.add(moments)
                all_reduce_(moments)
                self.count = -1
            L.call("rv_smallk_forward", x.ptr(), L.i32(x.ld), L.i64(x.pixels), L.i32(cin), L.ptr(wp), L.i32(pad32(cin)), L.i32(cp),
                   L.ptr(moments), L.i64(self.count), L.ptr(self.gamma_p), L.ptr(self.beta_p), L.f32(bn.eps),
                   L.f32(bn.momentum if bn.momentum is not None else 0.1), L.ptr(rm), L.ptr(rv), L.ptr(scale), L.ptr(shift),
                   L.ptr(self.mean), L.ptr(self.invstd), L.i32(1), self.out.ptr() if apply else None, L.i32(self.out.ld), L.stream_ptr())
            if bn.running_mean.shape[0] != cp:
                bn.running_mean.copy_(rm[:c])
                bn.running_var.copy_(rv[:c])
            t.bn_counters.append(bn.num_batches_tracked)
        else:
            L.call("rv_bn_fold_eval", L.i32(cp), L.ptr(self.gamma_p), L.ptr(self.beta_p), L.ptr(rm), L.ptr(rv), L.f32(bn.eps),
                   L.ptr(scale), L.ptr(shift), L.stream_ptr())
            if apply:
                L.call("rv_smallk_forward", x.ptr(), L.i32(x.ld), L.i64(x.pixels), L.i32(cin), L.ptr(wp), L.i32(pad32(cin)), L.i32(cp),
                       None, L.i64(self.count), None, None, L.f32(bn.eps), L.f32(0.1), None, None, L.ptr(scale), L.ptr(shift), None, None,
                       L.i32(1), self.out.ptr(), L.i32(self.out.ld), L.stream_ptr())
        t.ops.append(self)

    def backward(self, t: Tape) -> None:
        from . import engine_bwd

        engine_bwd.smallk_backward(self, t)


def _smallk_eligible(layer: TapLayer, x: Operand, relu: bool, need_input_grad: bool) -> bool:
    """(The backward of this path recomputes the raw output y = W x from the <= 8 input channels -- RV_BNB_Y_FROM_INPUT -- so
    the ReLU gate and xhat come from fp32 values with the layer's own batch statistics, whatever gamma is.)"""
    g = layer.geom
    return (SMALLK_FORWARD and relu and not need_input_grad and isinstance(x, Act) and layer.fwd_form == "gather" and g.kh == 1
            and g.kw == 1 and g.stride_w == 1 and layer.c_in <= 8 and layer.in_perm is None and layer.bias is None)


EVAL_FOLD = True


def conv_bn(t: Tape, layer: TapLayer, x: Operand, bn: nn.BatchNorm2d, relu: bool = True,
            need_input_grad: bool = True, smallk: bool = True, fold_eval: bool = True, out: Optional[Act] = None) -> Operand:
    """conv -> BatchNorm (-> ReLU).  ``smallk=False`` keeps a small-K layer on the generic path (a Lazy result), for consumers
    that fold the BatchNorm themselves (MetaModulateOp; ``fold_eval=False`` likewise keeps the Lazy form in eval mode)."""
    if smallk and _smallk_eligible(layer, x, relu, need_input_grad):
        return SmallKOp(t, layer, x, bn).out
    if t.inference and EVAL_FOLD and fold_eval and layer.bias is None:
        # inference: BatchNorm folded into the weight image and the bias, ReLU in the epilogue -- the conv writes the activation
        # itself (no folded operand for the consumer to apply, no write-out pass for the LDS-DMA kernels)
        # (``out``: where the activation goes -- honoured on this path only, the caller checks ``result is out``)
        return ConvOp(t, layer, x, need_input_grad=need_input_grad, eval_bn=bn, relu_out=relu, out=out).out
    conv = ConvOp(t, layer, x, stats=t.training, need_input_grad=need_input_grad)
    return BnOp(t, conv, bn, relu).lazy


EVAL_RES_FUSE = True


def conv_bn_residual(t: Tape, layer: TapLayer, x: Operand, bn: nn.BatchNorm2d, res: Operand, relu_conv: bool, relu_out: bool,
                     out: Optional[Act] = None) -> Optional[Act]:
    """Inference: ``relu_out?( relu_conv?(bn(conv(x))) + res )`` in the conv's own launch (BatchNorm folded, residual added in the
    epilogue: rv_tap_residual).  None when the fusion does not apply (training; folded operands) -- the caller then takes
    ``conv_bn`` + ``CombineOp``, whose result is the same bit for bit."""
    if (not t.inference or not (EVAL_FOLD and EVAL_RES_FUSE) or layer.bias is not None or not isinstance(x, Act) or not isinstance(res, Act)
            or res.cp != pad32(layer.c_out)):
        return None
    return ConvOp(t, layer, x, eval_bn=bn, relu_out=relu_conv, residual=res, res_relu=relu_out, out=out).out


GROUP_SYNC_BN = os.environ.get("RV3D_NO_GROUP_SYNC_BN") is None


def conv_bn_many(t: Tape, specs: Sequence[Tuple[TapLayer, Operand, nn.BatchNorm2d, bool, bool]]) -> List[Operand]:
    """Several independent conv -> BatchNorm (-> ReLU) layers whose inputs are all available: every conv is launched first,
    then every BatchNorm is finalised -- under SyncBN with ONE all-reduce for the whole group (``allreduce_partial_rows_many``)
    instead of one per layer; the BnOps sit next to each other on the tape, so the backward pass groups their collectives
    too (``Tape.backward``).  spec = (layer, input, bn, relu, need_input_grad).  Results in spec order."""
    if t.inference and EVAL_FOLD and all(layer.bias is None for layer, *_ in specs):
        return [ConvOp(t, layer, x, need_input_grad=nig, eval_bn=bn, relu_out=relu).out for layer, x, bn, relu, nig in specs]
    convs = [ConvOp(t, layer, x, stats=t.training, need_input_grad=nig) for layer, x, _, _, nig in specs]
    reduced: List[Optional[Tensor]] = [None] * len(specs)
    if t.training and GROUP_SYNC_BN and len(specs) > 1:
        sync = [i for i, (_, _, bn, _, _) in enumerate(specs) if bn_sync_world(bn, True) > 1]
        if len(sync) > 1:
            views = allreduce_partial_rows_many([(convs[i].partial, convs[i].rows, convs[i].count, None) for i in sync])
            for i, v in zip(sync, views):
                reduced[i] = v
    return [BnOp(t, conv, bn, relu, reduced=r).lazy for conv, (_, _, bn, relu, _), r in zip(convs, specs, reduced)]


POS_FUSE = True


def pos_pair_eligible(l0: TapLayer, l1: TapLayer, x: Operand) -> bool:
    """3 -> C -> C positional pair of the MetaKernel stem (C = 256: rv-av2, C = 128: rv-waymo) on a plain 9x-grid input: the pair runs
    as rv_pos_forward."""
    g0, g1 = l0.geom, l1.geom
    return (POS_FUSE and SMALLK_FORWARD and isinstance(x, Act) and _smallk_eligible(l0, x, True, False) and l0.c_in <= 3 and l0.c_out in (256, 128)
            and l1.fwd_form == "gather" and g1.kh == 1 and g1.kw == 1 and g1.stride_w == 1 and l1.c_in == l0.c_out and l1.c_out == l0.c_out
            and l1.bias is None and l1.in_perm is None and x.ld >= 4)


def pos_pair(t: Tape, l0: TapLayer, bn0: nn.BatchNorm2d, l1: TapLayer, bn1: nn.BatchNorm2d, x: Act) -> "Lazy":
    """``relu(bn1(conv1(relu(bn0(conv0(x))))))`` of the two positional layers: closed-form statistics of the first layer, then ONE
    persistent kernel that generates its activated output in the second layer's operand staging (rv_pos_forward).  The ops
    recorded on the tape are the ordinary SmallKOp / ConvOp / BnOp, so backward is unchanged."""
    sk = SmallKOp(t, l0, x, bn0, apply=False)
    h1 = sk.out
    y2 = Act.empty(x.N, x.H, x.W, l1.c_out, t.device)
    rows = L.load().rv_pos_forward_rows(L.i64(x.pixels))
    partial = torch.empty((rows + L.STATS_SCRATCH_ROWS, 2, y2.cp), dtype=torch.float32, device=t.device) if t.training else None
    call = lambda: L.call("rv_pos_forward", x.ptr(), L.i32(x.ld), L.i32(l0.c_in), L.i64(x.pixels), L.ptr(l0.packed("gather")), L.i32(pad32(l0.c_in)),
                          L.ptr(sk.scale), L.ptr(sk.shift), L.ptr(l1.packed("gather")), L.i32(y2.cp), h1.ptr(), y2.ptr(), L.ptr(partial), L.stream_ptr())
    if PROFILE is not None:
        _launch("pos_fwd_kernel", 2.0 * x.pixels * l1.c_in * l1.c_out, call)
    else:
        call()
    conv = ConvOp(t, l1, h1, stats=t.training, out=y2, precomputed=(partial, rows))
    conv.pos_first = sk  # backward: this conv's input gradient is consumed by `sk`'s BatchNorm backward inside one kernel
    return BnOp(t, conv, bn1, True).lazy


POS_MOD_FUSE = True


def _eval_scale_shift(bn: nn.BatchNorm2d, cp: int, dev) -> Tuple[Tensor, Tensor]:
    """Eval-mode BatchNorm as (scale, shift) over the padded channels."""
    scale = torch.empty(cp, dtype=torch.float32, device=dev)
    shift = torch.empty(cp, dtype=torch.float32, device=dev)
    L.call("rv_bn_fold_eval", L.i32(cp), L.ptr(_padded(bn.weight, cp)), L.ptr(_padded(bn.bias, cp)), L.ptr(_padded(bn.running_mean, cp)),
           L.ptr(_padded(bn.running_var, cp, 1.0)), L.f32(bn.eps), L.ptr(scale), L.ptr(shift), L.stream_ptr())
    return scale, shift


def pos_modulate_eligible(t: Tape, l0: TapLayer, l1: TapLayer, x: Operand, feat: Operand) -> bool:
    """Inference: the positional pair AND the modulation in one kernel (rv_pos_modulate_forward)."""
    return (t.inference and POS_MOD_FUSE and pos_pair_eligible(l0, l1, x) and isinstance(feat, Act) and feat.cp == pad32(l1.c_out)
            and feat.cp == l1.c_out and feat.W >= 32 and x.pixels == 9 * feat.pixels and x.pixels < 2**31 - 512)


class PosModulateOp(Op):
    """Inference only: ``geo = relu(bn1(conv1(relu(bn0(conv0(rel)))))) * unfold(feat)`` of MetaKernel.forward
    (nn/stems/__init__.py:76-83) from ONE persistent kernel -- the two positional tensors of the 9x grid never reach memory."""

    def __init__(self, t: Tape, l0: TapLayer, bn0: nn.BatchNorm2d, l1: TapLayer, bn1: nn.BatchNorm2d, x: Act, feat: Act) -> None:
        assert not t.training
        cp = feat.cp
        s1, t1 = _eval_scale_shift(bn0, cp, t.device)
        s2, t2 = _eval_scale_shift(bn1, cp, t.device)
        self.out = Act.empty(feat.N, feat.H, feat.W, 9 * cp, t.device)
        call = lambda: L.call("rv_pos_modulate_forward", x.ptr(), L.i32(x.ld), L.i32(l0.c_in), L.ptr(l0.packed("gather")), L.i32(pad32(l0.c_in)),
                              L.ptr(s1), L.ptr(t1), L.ptr(l1.packed("gather")), L.i32(cp), L.ptr(s2), L.ptr(t2), feat.ptr(), L.i32(feat.ld),
                              L.i32(feat.N), L.i32(feat.H), L.i32(feat.W), self.out.ptr(), L.stream_ptr())
        if PROFILE is not None:
            _launch("pos_fwd_kernel<eval>", 2.0 * x.pixels * l1.c_in * l1.c_out, call)
        else:
            call()
        t.ops.append(self)

    def backward(self, t: Tape) -> None:
        raise L.RvError("the fused positional pair + modulation is an inference kernel: there is no backward")


# ---------------------------------------------------------------------------------------------
# element-wise combine
# ---------------------------------------------------------------------------------------------
COMBINE_LAUNCHES = 0  # element-wise combine passes issued so far (tests: the inference epilogues remove them)


class CombineOp(Op):
    """``out = relu?( fa(a) + fb(b) )`` with Lazy operands folded in; materialises a block output."""

    def __init__(self, t: Tape, a: Operand, b: Optional[Operand], relu_out: bool, out: Optional[Act] = None) -> None:
        global COMBINE_LAUNCHES
        COMBINE_LAUNCHES += 1
        self.a, self.b, self.relu_out = a, b, relu_out
        ra, sa, ta, fa = _operand_parts(a)
        flags = (L.EW_RELU_A if fa & L.IN_RELU else 0) | (L.EW_RELU_OUT if relu_out else 0)
        rb = sb = tb = None
        if b is not None:
            rb, sb, tb, fb = _operand_parts(b)
            flags |= L.EW_RELU_B if fb & L.IN_RELU else 0
            assert rb.cp == ra.cp and rb.pixels == ra.pixels
        self.out = out if out is not None else ra.like()
        L.call("rv_ew_combine", L.i64(ra.pixels), L.i32(ra.cp), ra.ptr(), L.i32(ra.ld), L.ptr(sa), L.ptr(ta),
               rb.ptr() if rb is not None else None, L.i32(rb.ld if rb is not None else 0), L.ptr(sb), L.ptr(tb),
               self.out.ptr(), L.i32(self.out.ld), L.i32(flags), L.stream_ptr())
        t.producers[id(self.out)] = self
        t.ops.append(self)

    def backward(self, t: Tape) -> None:
        from . import engine_bwd

        engine_bwd.combine_backward(self, t)


# ---------------------------------------------------------------------------------------------
# MetaKernel pieces
# ---------------------------------------------------------------------------------------------
class MetaRelativeOp(Op):
    """cart (B,3,H,W) fp32 NCHW -> relative neighbour coordinates as an (N,H,9W,32) bf16 image (no gradient)."""

    def __init__(self, t: Tape, cart: Tensor) -> None:
        n, _, h, w = cart.shape
        cart = cart.contiguous().float()
        self.out = Act.empty(n, h, w * 9, 3, t.device)
        L.call("rv_meta_relative", L.ptr(cart), L.i32(n), L.i32(h), L.i32(w), self.out.ptr(), L.stream_ptr())
        t.ops.append(self)

    def backward(self, t: Tape) -> None:
        return None


class MetaModulateOp(Op):
    """geo[n,h,w,k*C+c] = relu(bn(pos))[n,h,w*9+k,c] * feat[n,h+dy_k,w+dx_k,c]."""

    def __init__(self, t: Tape, pos: Lazy, feat: Act) -> None:
        self.pos, self.feat = pos, feat
        n, h, w, cp = feat.N, feat.H, feat.W, feat.cp
        assert pos.raw.W == 9 * w and pos.raw.cp == cp and pos.raw.ld == cp
        self.out = Act.empty(n, h, w, 9 * cp, t.device)
        L.call("rv_meta_modulate", pos.raw.ptr(), L.ptr(pos.bn.scale), L.ptr(pos.bn.shift), feat.ptr(), L.i32(feat.ld),
               L.i32(n), L.i32(h), L.i32(w), L.i32(cp), self.out.ptr(), L.stream_ptr())
        t.ops.append(self)

    def backward(self, t: Tape) -> None:
        from . import engine_bwd

        engine_bwd.modulate_backward(self, t)


class ConcatOp(Op):
    """Channel concat through a copy (only when the parts are not multiples of 32 channels: tiny test models)."""

    def __init__(self, t: Tape, parts: Sequence[Act]) -> None:
        self.parts = list(parts)
        c = sum(p.c for p in parts)
        ref = parts[0]
        self.out = Act.empty(ref.N, ref.H, ref.W, c, t.device, zero=True)
        o = 0
        for p in parts:
            self.out.data[..., o : o + p.c].copy_(p.data[..., : p.c])
            o += p.c
        t.ops.append(self)

    def backward(self, t: Tape) -> None:
        g, have = t.grad_buffer(self.out)
        if not have:
            return
        o = 0
        for p in self.parts:
            gp, have_p = t.grad_buffer(p)
            if have_p:
                gp.data[..., : p.c].add_(g.data[..., o : o + p.c])
            else:
                gp.data.zero_()
                gp.data[..., : p.c].copy_(g.data[..., o : o + p.c])
                t.mark_written(p)
            o += p.c


def invalidate_packed_weights(model: nn.Module) -> None:
    """``TapLayer.invalidate()`` for every conv of ``model`` (after in-place edits through ``.data``)."""
    for m in model.modules():
        layer = m.__dict__.get("_rv_layer")
        if layer is not None:
            layer.invalidate()


def tap_layer(module: nn.Module, **kw) -> TapLayer:
    """The (cached) TapLayer of an ``nn.Conv2d`` / ``nn.ConvTranspose2d`` parameter holder."""
    layer = module.__dict__.get("_rv_layer")
    if layer is None or layer.weight is not module.weight:
        kh, kw_ = module.kernel_size
        sh, sw = module.stride
        if sh != 1:
            raise NotImplementedError("vertical stride is 1 everywhere on the range-view path (dla.py:37-63)")
        if tuple(module.dilation) != (1, 1):
            raise NotImplementedError("dilation != 1 is not used on the range-view path")
        if isinstance(module, nn.ConvTranspose2d):
            pad = tuple(module.padding)
            layer = TapLayer(module.weight, sw, pad, transposed=True, bias=module.bias, **kw)
        else:
            pad = getattr(module, "_rv_same_pad", None)
            if pad is None:
                pad = module.padding
                pad = ((kh - 1) // 2, (kw_ - 1) // 2) if pad == "same" else tuple(pad)
            layer = TapLayer(module.weight, sw, pad, transposed=False, bias=module.bias, **kw)
        module.__dict__["_rv_layer"] = layer
    return layer
