"""Box decoding -- mirrors ``torchbox3d/math/ops/coding.py:79-144``."""

from __future__ import annotations

import torch
from torch import Tensor

from ... import _lib as L
from ...engine import _require_cuda


def decode_range_view(regressands: Tensor, cart: Tensor, enable_azimuth_invariant_targets: bool) -> Tensor:
    """(B,8,H,W) regressands + (B,3,H,W) points -> (B,7,H,W) [x,y,z,l,w,h,yaw].

    fp64 arithmetic on device, rounded once to the input dtype (``coding.py:126-128,144``).
    """
    _require_cuda(regressands, "regressands")
    dtype = regressands.dtype
    B, _, H, W = regressands.shape
    reg = regressands.detach().float().contiguous()
    c = cart.detach().float().contiguous()
    out = torch.empty((B, 7, H, W), dtype=torch.float32, device=reg.device)
    L.call("rv_decode_range_view", L.ptr(reg), L.ptr(c), L.i32(B), L.i32(H), L.i32(W),
           L.i32(1 if enable_azimuth_invariant_targets else 0), L.ptr(out), L.stream_ptr())
    return out.to(dtype)


# ---------------------------------------------------------------------------------------------
# Detections wire / on-disk format: ``build_dataframe`` (``coding.py:11-76``) and the per-sweep
# feather files ``Detector.validation_step`` writes (``nn/arch/detector.py:366-380``).  The reference builds polars
# frames; polars is not in this image, so the same table is an Arrow table (``pyarrow``), which is what
# ``DataFrame.write_ipc`` puts on disk anyway (Arrow IPC file format == feather v2).  Pinned by
# ``tests/golden/detections_frame.npz``: the reference's own ``build_dataframe`` run in the build container over a stand-in for
# polars (column order, dtypes, rows; ``tests/test_host_cpu.py::test_detections_table_matches_the_reference_build_dataframe``).
# ---------------------------------------------------------------------------------------------
DETECTION_COLUMNS = ("tx_m", "ty_m", "tz_m", "length_m", "width_m", "height_m", "qw", "qx", "qy", "qz")


def _column(frame, name: str):
    """A named column of a dict-of-sequences / pandas / pyarrow / polars-like frame, as a python list."""
    if isinstance(frame, dict):
        col = frame[name]
    elif hasattr(frame, "column") and hasattr(frame, "schema"):  # pyarrow.Table
        col = frame.column(name).to_pylist()
    elif hasattr(frame, "get_column"):  # polars
        col = frame.get_column(name).to_list()
    else:  # pandas
        col = frame[name].tolist()
    return list(col.tolist() if hasattr(col, "tolist") else col)


def build_dataframe(params: Tensor, scores: Tensor, categories: Tensor, batch_index: Tensor, uuids, idx_to_category):
    """Detections -> Arrow table with the reference's detection schema.

    ``params`` (N,10) = [tx_m, ty_m, tz_m, length_m, width_m, height_m, qw, qx, qy, qz]; ``categories`` / ``batch_index``
    come back from the decoder as floats (``nms.py:113,242``) and are truncated to int32 as the reference does.  ``uuids``
    maps ``batch_index`` -> (``log_id``, ``timestamp_ns``); ``idx_to_category`` lists the category names in class-index order
    (rows of the reference's task frame; its ``task_id`` / ``offset`` columns are dropped there too).  Inner joins on
    ``batch_index`` and ``category_index`` in detection order; ``category_index`` is dropped, ``batch_index`` stays.
    """
    import pyarrow as pa

    p = params.detach().float().cpu().reshape(-1, 10).numpy()
    s = scores.detach().float().cpu().reshape(-1).numpy()
    c = categories.detach().cpu().reshape(-1).int().numpy()
    b = batch_index.detach().cpu().reshape(-1).int().numpy()
    names = _column(idx_to_category, "category") if not isinstance(idx_to_category, (list, tuple)) else list(idx_to_category)
    key = {int(bi): (str(l), int(t)) for bi, l, t in zip(_column(uuids, "batch_index"), _column(uuids, "log_id"), _column(uuids, "timestamp_ns"))}
    keep = [i for i in range(len(s)) if int(b[i]) in key and 0 <= int(c[i]) < len(names)]  # inner joins
    cols = {name: pa.array(p[keep, j], type=pa.float32()) for j, name in enumerate(DETECTION_COLUMNS)}
    cols["score"] = pa.array(s[keep], type=pa.float32())
    cols["batch_index"] = pa.array(b[keep], type=pa.int32())
    cols["log_id"] = pa.array([key[int(b[i])][0] for i in keep], type=pa.string())
    cols["timestamp_ns"] = pa.array([key[int(b[i])][1] for i in keep], type=pa.int64())
    cols["category"] = pa.array([names[int(c[i])] for i in keep], type=pa.string())
    return pa.table(cols)


def write_detections(table, dst_dir: str, run_uuid: str):
    """One feather (Arrow IPC) file per (log_id, timestamp_ns), rows in table order:
    ``<dst_dir>/predictions/<run_uuid>/<log_id>/<timestamp_ns>.feather`` (``detector.py:366-380``).  Returns the paths."""
    import os

    import pyarrow as pa

    logs, stamps = table.column("log_id").to_pylist(), table.column("timestamp_ns").to_pylist()
    groups = {}
    for i, k in enumerate(zip(logs, stamps)):
        groups.setdefault(k, []).append(i)  # first-appearance order, as group_by(maintain_order=True)
    paths = []
    for (log_id, ts), rows in groups.items():
        dst = os.path.join(dst_dir, "predictions", run_uuid, log_id, f"{ts}.feather")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        part = table.take(pa.array(rows, type=pa.int64()))
        with pa.OSFile(dst, "wb") as sink, pa.ipc.new_file(sink, part.schema) as w:  # Arrow IPC file == feather v2, uncompressed
            w.write_table(part)
        paths.append(dst)
    return paths
