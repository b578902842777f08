"""RCCL, bound directly (ctypes) for the latency-bound collectives of the hot path.

SyncBN moves ``(2C + 1)`` floats per BatchNorm layer and direction -- ~130 collectives of a few hundred bytes to a few KB per
training step, each one on the critical path (conv -> statistics -> all-reduce -> finalize -> next conv).  Through
``torch.distributed`` every one of them pays c10d's host path and a round trip between the compute stream and
ProcessGroupNCCL's own stream (two event waits); measured on one MI355X with a process group of one rank: ~190 us per
collective, 25 ms of a 105 ms step (profiles/r03_syncbn_collectives.md).  Here ``ncclAllReduce`` is enqueued on the CURRENT
compute stream of the calling thread: no second stream, no events, one C call.

The communicator is bootstrapped over the existing ``torch.distributed`` process group (rank 0 creates the ``ncclUniqueId``
and broadcasts it), uses the RCCL library PyTorch itself ships and has already loaded, and lives for the life of the
process group.  Backends other than ``nccl`` (the gloo tests on CPU) keep using ``torch.distributed``.

OPT-IN (``RV3D_DIRECT_RCCL=1``).  This build has only ever had ONE GPU: the binding is tested with a one-rank communicator
(``tests/test_gpu_ddp.py``, ``profiles/tools/mb_collective.py``), its multi-rank bootstrap never ran.  The default therefore
stays ``torch.distributed.all_reduce`` -- the path every multi-process test of this repository exercises -- and the direct
binding is the A/B for a node (``profiles/r03_syncbn_collectives.md``: on one rank the two differ by ~1.5 % of a step).
"""

from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_NCCL_DTYPE = {torch.float32: 7, torch.float64: 8, torch.float16: 6, torch.bfloat16: 9, torch.int32: 2, torch.int64: 4}
_NCCL_SUM = 0


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


class RcclError(RuntimeError):
    pass


_lib: Optional[ctypes.CDLL] = None
_comm: Optional[ctypes.c_void_p] = None
_comm_key = None
DISABLED = os.environ.get("RV3D_DIRECT_RCCL") is None


def _load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            path = "librccl.so"
        lib = ctypes.CDLL(path)
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib = lib
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RcclError(f"{what}: {_load().ncclGetErrorString(rc).decode()}")


def available() -> bool:
    """True when the default process group runs over RCCL and the direct binding is not switched off."""
    d = torch.distributed
    return (not DISABLED) and d.is_available() and d.is_initialized() and d.get_backend() == "nccl"


def communicator() -> ctypes.c_void_p:
    """The process-wide communicator over the ranks of the default group (created on first use: a collective call)."""
    global _comm, _comm_key
    d = torch.distributed
    key = (d.get_rank(), d.get_world_size(), id(d.group.WORLD))
    if _comm is not None and _comm_key == key:
        return _comm
    lib = _load()
    uid = _UniqueId()
    if d.get_rank() == 0:
        _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
    box = [ctypes.string_at(ctypes.byref(uid), 128) if d.get_rank() == 0 else None]  # (all 128 bytes: the id holds NULs)
    d.broadcast_object_list(box, src=0)
    ctypes.memmove(ctypes.byref(uid), box[0], 128)
    comm = ctypes.c_void_p()
    _check(lib.ncclCommInitRank(ctypes.byref(comm), d.get_world_size(), uid, d.get_rank()), "ncclCommInitRank")
    _comm, _comm_key = comm, key
    return comm


def all_reduce_(t: torch.Tensor) -> None:
    """In-place sum over the ranks, enqueued on the current stream of ``t``'s device (asynchronous with respect to the host)."""
    if not (t.is_cuda and t.is_contiguous()):
        raise RcclError("rccl.all_reduce_ needs a contiguous device tensor")
    comm = communicator()
    stream = ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    p = ctypes.c_void_p(t.data_ptr())
    _check(_load().ncclAllReduce(p, p, t.numel(), _NCCL_DTYPE[t.dtype], _NCCL_SUM, comm, stream), "ncclAllReduce")


def shutdown() -> None:
    global _comm, _comm_key
    if _comm is not None:
        _load().ncclCommDestroy(_comm)
        _comm, _comm_key = None, None
