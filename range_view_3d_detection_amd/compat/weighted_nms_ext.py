"""``weighted_nms_ext`` -- the op-level FFI the reference binds for weighted NMS, on ``librv3d_hip.so``.

The reference calls (``/root/reference/src/torchbox3d/math/ops/nms.py:161-170``)::

    num_out = weighted_nms_ext.wnms_gpu(boxes, data2merge_score, output, keep, count,
                                        nms_threshold, merge_thresh, boxes.device.index)

with the TorchEx contract: inputs sorted by score descending; ``boxes`` f32 (N,5) [x1,y1,x2,y2,ry] on the
device; ``data2merge_score`` f32 (N,C+1) on the device, last column = score; ``output`` f32 (N,C+1) zero-initialised by the
caller, rows ``[0, num_out)`` filled in place; ``keep`` **int64 on the HOST** (N,), entries ``[0, num_out)`` filled with the
indices (into the sorted order) of the kept boxes; ``count`` int64 (N,) on the device, members merged into each kept box;
returns ``num_out`` as a Python int (synchronous); no stream argument -- the current stream of ``device_index``.
Post-conditions the caller asserts (``nms.py:173-174``): ``output[num_out:] == 0`` and ``count[:num_out] > 0``.

The arithmetic is ``rv_wnms`` (csrc/nms.hip; declared semantics in oracle/nms.py -- the TorchEx source is not part of the
reference tree, parity unpinned).  No CPU fallback: host tensors for boxes/data raise.
"""

from __future__ import annotations

import ctypes

import torch
from torch import Tensor

from range_view_3d_detection_amd import _lib as L


def wnms_gpu(boxes: Tensor, data2merge_score: Tensor, output: Tensor, keep: Tensor, count: Tensor, nms_thresh: float,
             merge_thresh: float, device_index: int) -> int:
    for name, t in (("boxes", boxes), ("data2merge_score", data2merge_score), ("output", output), ("count", count)):
        if not t.is_cuda:
            raise L.RvError(f"wnms_gpu: {name} must live on the GPU (no CPU fallback)")
        if not t.is_contiguous():
            raise L.RvError(f"wnms_gpu: {name} must be contiguous")
    if boxes.dtype != torch.float32 or data2merge_score.dtype != torch.float32 or output.dtype != torch.float32:
        raise L.RvError("wnms_gpu: boxes / data2merge_score / output must be float32")
    if count.dtype != torch.int64 or keep.dtype != torch.int64 or keep.is_cuda:
        raise L.RvError("wnms_gpu: keep must be a host int64 tensor, count a device int64 tensor")
    n, d = data2merge_score.shape
    if boxes.shape != (n, 5) or output.shape != (n, d) or keep.numel() < n or count.numel() < n:
        raise L.RvError("wnms_gpu: shape mismatch")
    if device_index is not None and boxes.device.index != device_index:
        raise L.RvError(f"wnms_gpu: tensors are on cuda:{boxes.device.index}, device_index says {device_index}")
    if n == 0:
        return 0
    with torch.cuda.device(boxes.device):
        keep_dev = torch.empty(n, dtype=torch.int64, device=boxes.device)
        ws = torch.empty(L.load().rv_wnms_workspace_bytes(L.i64(n)), dtype=torch.uint8, device=boxes.device)
        num_out = ctypes.c_int64(0)
        L.call("rv_wnms", L.ptr(boxes), L.ptr(data2merge_score), L.i64(n), L.i32(d), L.f32(nms_thresh), L.f32(merge_thresh),
               L.ptr(output), L.ptr(keep_dev), L.ptr(count), L.ptr(ws), ctypes.byref(num_out), L.stream_ptr())
        k = int(num_out.value)
        keep[:k].copy_(keep_dev[:k])  # device -> host, synchronous
    return k
