"""What of ``torchbox3d/prototype/loader.py`` shapes the hot path's input, on device: the table -> image step of
``DataLoader.__getitem__`` (``loader.py:568-705``: :func:`range_view_from_table`), the W padding that makes the range-image
width divisible by 16 (``subsample_range_view``, ``loader.py:792-815``) and the augmentations (``loader.py:825-990``)."""

from __future__ import annotations

import ctypes
from typing import Any, Dict, List, Mapping, Tuple

import torch
from torch import Tensor

from .. import _lib as L
from ..engine import _require_cuda

_PAD = {("waymo", 1): 3, ("waymo", 4): 19, ("av2", 1): 4, ("av2", 4): 28}


def _pad(x: Tensor, mask, pad: int, circular: bool) -> Tensor:
    c, h, w = x.shape
    src = x.float().contiguous()
    m = mask.float().reshape(h, w).contiguous() if mask is not None else None
    out = torch.empty((c, h, w + 2 * pad), dtype=torch.float32, device=x.device)
    L.call("rv_pad_range_view", L.ptr(src), L.ptr(m), L.i32(c), L.i32(h), L.i32(w), L.i32(pad), L.i32(1 if circular else 0), L.ptr(out),
           L.stream_ptr())
    return out


def subsample_range_view(range_view: Tensor, mask: Tensor, cart: Tensor, dataset_name: str, x_stride: int, mode: str) -> Tuple[Tensor, Tensor, Tensor]:
    """``range_view *= mask``; pad W by [4,4] (AV2) / [3,3] (Waymo) (``constant`` or ``circular``); ``[:, :, ::x_stride]``."""
    _require_cuda(range_view, "range_view")
    pad = _PAD[(dataset_name, 4 if x_stride == 4 else 1)]
    circ = mode == "circular"
    rv = _pad(range_view, mask, pad, circ)[:, :, ::x_stride]
    m = _pad(mask.float(), None, pad, circ)[:, :, ::x_stride]
    c = _pad(cart, None, pad, circ)[:, :, ::x_stride]
    return rv, m, c


CART_COLUMNS = ("x", "y", "z")


def read_sweep_table(path) -> Dict[str, "np.ndarray"]:
    """Range-view feather file (``pl.scan_ipc(self.lidar_path(...))``, ``loader.py:595``) -> name -> column array (host I/O)."""
    import pyarrow as pa

    with pa.memory_map(str(path), "r") as src:
        t = pa.ipc.open_file(src).read_all()
    return {name: t.column(name).to_numpy(zero_copy_only=False) for name in t.column_names}


def range_view_from_table(table: Mapping[str, Any], range_view_config: Mapping[str, Any], dataset_name: str, x_stride: int = 1,
                          padding_mode: str = "constant", device="cuda", pad: bool = True, keep=None) -> Dict[str, Tensor]:
    """``DataLoader.__getitem__`` from the sweep table on (``loader.py:594-690``): ``table`` maps column names to H*W-row
    arrays (numpy or tensors); returns ``features`` (F,H,W'), ``mask`` (1,H,W') bool, ``cart`` (3,H,W') on ``device`` with W'
    the padded width.  ``pad=False`` stops BEFORE ``subsample_range_view`` (``loader.py:684-691``): width = the configured width,
    ``features`` not yet multiplied by the mask -- the state the reference's augmentations see (``loader.py:598-603``); finish with
    :func:`pad_batch`.  ``keep``: H*W booleans of a ``point_dropout`` (``loader.py:506-512``: every column times the keep mask) --
    it multiplies the table exactly where the ROI flag does, so it rides on that column of the one kernel.  The needed columns cross PCIe once, as ONE (n_cols, H*W) fp32 block; ROI filter, tanh(intensity)
    (Waymo), the 1e-9 of ``timedelta_ns``, the (F,H,W) layout and the mask are one kernel (``rv_table_to_range_view``).
    The ``view`` feature (``loader.py:605-624``: laser rows through the reverse ``ROW_MAPPING_64``, then 2 for rows <= 32 and 1
    for the rest, both times ``range > 0``) is formed on the host from the ``laser_number`` and ``range`` columns before the
    upload (two small integer maps over H*W rows); the dataset's row table is an argument, ``range_view_config["row_mapping_64"]``,
    like the other id tables of this package (the reference imports it from ``datasets/argoverse/constants.py``)."""
    import numpy as np

    names = list(range_view_config["feature_column_names"])
    h, w = int(range_view_config["height"]), int(range_view_config["width"])
    roi = bool(range_view_config.get("filter_roi", False))
    if "view" in names:
        if range_view_config.get("row_mapping_64") is None:
            raise L.RvError("the 'view' feature needs range_view_config['row_mapping_64'] (ROW_MAPPING_64 of datasets/argoverse/constants.py)")
        as_np = lambda c: c.detach().cpu().numpy() if isinstance(c, Tensor) else np.asarray(c)  # noqa: E731
        flag = (as_np(table["is_within_roi"]) != 0).astype(np.float32) if roi else np.float32(1.0)  # the ROI filter comes first (loader.py:599-601)
        ln = as_np(table["laser_number"]).astype(np.float32) * flag
        pos = ((as_np(table["range"]).astype(np.float32) * flag) > 0).astype(np.float32)
        rev = {int(v): i for i, v in enumerate(np.asarray(range_view_config["row_mapping_64"]).tolist())}
        ln2 = ln.copy()
        for k, v in rev.items():
            ln2[ln == k] = v
        ln2 = ln2 * pos
        table = dict(table)
        table["laser_number"] = ln2
        table["view"] = (2.0 * (ln2 <= 32).astype(np.float32) + (ln2 > 32).astype(np.float32)) * pos
    need: List[str] = []
    for n in names + list(CART_COLUMNS) + ["range"] + (["is_within_roi"] if roi else []):
        if n not in need:
            need.append(n)
    cols = []
    for n in need:
        c = table[n]
        c = c.detach().cpu().numpy() if isinstance(c, Tensor) else np.asarray(c)
        if c.shape != (h * w,):
            raise L.RvError(f"column {n!r} has shape {c.shape}, expected ({h * w},) = height * width rows")
        cols.append(c.astype(np.float32, copy=False))
    if keep is not None:
        k = np.asarray(keep).reshape(-1)
        if k.shape != (h * w,):
            raise L.RvError(f"keep mask has {k.size} entries, expected {h * w}")
        if roi:
            cols[need.index("is_within_roi")] = (cols[need.index("is_within_roi")] != 0).astype(np.float32) * k.astype(np.float32)
        else:
            need.append("is_within_roi")
            cols.append(k.astype(np.float32))
            roi = True
    host = torch.from_numpy(np.stack(cols))
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.RvError("range_view_from_table needs a CUDA (ROCm) device: the hot path has no CPU fallback")
    block = host.pin_memory().to(dev, non_blocking=True)
    feat_col = (ctypes.c_int32 * len(names))(*[need.index(n) for n in names])
    feat_op = (ctypes.c_int32 * len(names))(*[1 if (n == "intensity" and dataset_name == "waymo") else (2 if n == "timedelta_ns" else 0) for n in names])
    cart_col = (ctypes.c_int32 * 3)(*[need.index(n) for n in CART_COLUMNS])
    features = torch.empty((len(names), h, w), dtype=torch.float32, device=dev)
    cart = torch.empty((3, h, w), dtype=torch.float32, device=dev)
    mask = torch.empty((1, h, w), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.call("rv_table_to_range_view", L.ptr(block), L.i32(len(need)), L.i64(h * w), L.i32(len(names)), feat_col, feat_op, cart_col,
               L.i32(need.index("range")), L.i32(need.index("is_within_roi") if roi else -1), L.ptr(features), L.ptr(cart), L.ptr(mask),
               L.stream_ptr())
        if not pad:
            return {"features": features, "mask": mask.bool(), "cart": cart}
        features, m, cart = subsample_range_view(features, mask.bool(), cart, dataset_name, x_stride, padding_mode)
    return {"features": features, "mask": m > 0.5 if m.dtype != torch.bool else m, "cart": cart}


def pad_batch(batch: Mapping[str, Any], dataset_name: str, x_stride: int = 1, padding_mode: str = "constant") -> Dict[str, Any]:
    """``subsample_range_view`` (``loader.py:792-815``) over an UNPADDED batch dict: ``features`` (B,F,H,W) ``*= mask``, then
    ``features`` / ``mask`` / ``cart`` padded in W.  Other keys pass through."""
    out = dict(batch)
    fs, ms, cs = [], [], []
    for b in range(batch["features"].shape[0]):
        f, m, c = subsample_range_view(batch["features"][b], batch["mask"][b], batch["cart"][b], dataset_name, x_stride, padding_mode)
        fs.append(f)
        ms.append(m > 0.5)
        cs.append(c)
    out["features"], out["mask"], out["cart"] = torch.stack(fs), torch.stack(ms), torch.stack(cs)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Augmentations (``prototype/loader.py:514-549, 825-990``) on device tensors
# ---------------------------------------------------------------------------------------------------------------------
# The reference augments ONE sweep at a time inside ``Dataset.__getitem__`` (a polars table on a DataLoader worker), drawing
# from Python's ``random`` in the order of ``augmentations_config``.  Here a whole batch that already sits in HBM is
# augmented by ``rv_augment`` (csrc/augment.hip); the random draws are made on the host, sweep by sweep, in the SAME order
# with the same ``random`` calls, so that a seeded run picks the same parameters as the reference would.
import math  # noqa: E402
import random as _random  # noqa: E402
from typing import Any, Dict, List, Mapping, Optional, Sequence  # noqa: E402

COLS = ("tx_m", "ty_m", "tz_m", "length_m", "width_m", "height_m", "qw", "qx", "qy", "qz", "task_id", "offset", "batch_index")


class SweepTransform:
    """Composite of a chain of augmentations for one sweep: column map w_src = (a*w + b) mod W, xyz' = A xyz + t, and the
    map (Ar, tr) whose norm becomes the ``range`` channel (only after a random_global_scale)."""

    def __init__(self, width: int) -> None:
        self.W = width
        self.a, self.b = 1, 0
        self.A = torch.eye(3, dtype=torch.float64)
        self.t = torch.zeros(3, dtype=torch.float64)
        self.Ar, self.tr, self.use_range = torch.eye(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64), False
        self.ops: List[Any] = []

    def _then_cols(self, a2: int, b2: int) -> None:
        # new_out[w] = old_out[(a2*w + b2) mod W] = in[(a*(a2*w + b2) + b) mod W]
        self.a, self.b = self.a * a2, (self.a * b2 + self.b) % self.W

    def _then_affine(self, M: Tensor, d: Tensor) -> None:
        self.A, self.t = M @ self.A, M @ self.t + d

    def flip(self) -> None:  # loader.py:948-990
        self._then_cols(-1, self.W - 1)
        self._then_affine(torch.diag(torch.tensor([1.0, -1.0, 1.0], dtype=torch.float64)), torch.zeros(3, dtype=torch.float64))
        self.ops.append(("flip",))

    def rotate(self, theta: float) -> None:  # loader.py:825-882: roll by floor(theta / tau * W), xyz by rot.T = Rz(-theta)
        shift = math.floor(theta / math.tau * self.W)
        self._then_cols(1, -shift)
        c, s = math.cos(theta), math.sin(theta)
        self._then_affine(torch.tensor([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64), torch.zeros(3, dtype=torch.float64))
        self.ops.append(("rotate", theta))

    def scale(self, s: float) -> None:  # loader.py:885-915: range := ||xyz|| at this point of the chain
        self._then_affine(s * torch.eye(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64))
        self.Ar, self.tr, self.use_range = self.A.clone(), self.t.clone(), True
        self.ops.append(("scale", s))

    def translate(self, t: Sequence[float]) -> None:  # loader.py:918-945 (the range column is left as it is)
        self._then_affine(torch.eye(3, dtype=torch.float64), torch.tensor(list(t), dtype=torch.float64))
        self.ops.append(("translate", tuple(t)))

    def packed(self) -> Tensor:
        return torch.cat([torch.tensor([float(self.a), float(self.b)], dtype=torch.float64), self.A.flatten(), self.t, self.Ar.flatten(), self.tr,
                          torch.tensor([1.0 if self.use_range else 0.0] + [0.0] * 5, dtype=torch.float64)])

    # ---- annotations (host, fp64; a handful of rows) ----
    def apply_to_annotations(self, ann: Tensor) -> Tensor:
        """(M, >= 10) rows in ``COLS`` order [tx ty tz l w h qw qx qy qz ...] -> transformed copy."""
        ann = ann.clone().double()
        if ann.shape[0] == 0:
            return ann
        for op in self.ops:
            if op[0] == "flip":
                ann[:, 1] = -ann[:, 1]
                w, x, y, z = ann[:, 6], ann[:, 7], ann[:, 8], ann[:, 9]
                yaw = -torch.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))  # the reference keeps the (negated) yaw only
                ann[:, 6], ann[:, 7], ann[:, 8], ann[:, 9] = torch.cos(yaw / 2), 0.0, 0.0, torch.sin(yaw / 2)
            elif op[0] == "rotate":
                c, s = math.cos(op[1]), math.sin(op[1])
                tx, ty = ann[:, 0].clone(), ann[:, 1].clone()
                ann[:, 0], ann[:, 1] = c * tx + s * ty, -s * tx + c * ty
                cw, cz = math.cos(-op[1] / 2), math.sin(-op[1] / 2)  # q' = q * q_z(-theta)  (mat = R_q @ rot.T)
                w, x, y, z = ann[:, 6].clone(), ann[:, 7].clone(), ann[:, 8].clone(), ann[:, 9].clone()
                ann[:, 6], ann[:, 7], ann[:, 8], ann[:, 9] = w * cw - z * cz, x * cw + y * cz, y * cw - x * cz, w * cz + z * cw
            elif op[0] == "scale":
                ann[:, :6] = op[1] * ann[:, :6]
            elif op[0] == "translate":
                ann[:, :3] = ann[:, :3] + torch.tensor(op[1], dtype=torch.float64)
        return ann


def draw_sweep_transform(width: int, augmentations_config: Mapping[str, Mapping[str, float]], rng=_random) -> SweepTransform:
    """The draws of ``Dataset.apply_augmentations`` (``loader.py:514-549``) for ONE sweep, in config order, with the
    reference's own ``random`` calls (flip: ``random() > p`` skips; rotation: ``random() > p`` skips, then ``uniform``;
    scale: ``uniform``; translation: three ``normalvariate``)."""
    tr = SweepTransform(width)
    for k, v in augmentations_config.items():
        if k == "flip_azimuth":
            if rng.random() > v["p"]:
                continue
            tr.flip()
        elif k == "random_rotation":
            if rng.random() > v["p"]:
                continue
            tr.rotate(rng.uniform(v["low"], v["high"]))
        elif k == "random_global_scale":
            tr.scale(rng.uniform(v["low"], v["high"]))
        elif k == "random_global_translation":
            tr.translate([rng.normalvariate(0, v["std_x"]), rng.normalvariate(0, v["std_y"]), rng.normalvariate(0, v["std_z"])])
        elif k == "point_dropout":
            raise NotImplementedError("point_dropout inside a chain: supported at the HEAD of augmentations_config only (train_batch_from_tables folds its "
                                      "keep mask into the table -> image kernel); no shipped rv-* recipe enables it")
        else:
            raise KeyError(f"unknown augmentation {k!r}")
    return tr


def apply_sweep_transforms(x: Tensor, transforms: Sequence[SweepTransform], xyz_channels: Optional[Sequence[int]] = None,
                           range_channel: int = -1) -> Tensor:
    """(B, C, H, W) tensor -> augmented copy (``rv_augment``).  ``xyz_channels`` = positions of x, y, z among the channels
    (None: apply the column map only -- the mask); bool tensors go through as fp32 0 / 1."""
    _require_cuda(x, "tensor")
    was_bool = x.dtype == torch.bool
    src = x.float().contiguous()
    b, c, h, w = src.shape
    assert len(transforms) == b and all(t.W == w for t in transforms)
    params = torch.stack([t.packed() for t in transforms]).to(src.device)
    out = torch.empty_like(src)
    ix, iy, iz = (-1, -1, -1) if xyz_channels is None else xyz_channels
    L.call("rv_augment", L.ptr(src), L.ptr(out), L.i32(b), L.i32(c), L.i32(h), L.i32(w), L.i32(ix), L.i32(iy), L.i32(iz), L.i32(range_channel),
           L.ptr(params), L.stream_ptr())
    return out > 0.5 if was_bool else out


def augment_batch(batch: Dict[str, Any], feature_column_names: Sequence[str], augmentations_config: Mapping[str, Mapping[str, float]],
                  rng=_random, width: Optional[int] = None) -> Dict[str, Any]:
    """Batch-dict contract of the reference's loader (``loader.py:568-705, 245-248``): ``features`` (B, F, H, W) fp32 with the
    channels named by ``feature_column_names`` (``conf/model/range_view.yaml:141-146``: x, y, z and range among them),
    ``cart`` (B, 3, H, W) fp32, ``mask`` (B, 1, H, W) bool (= range > 0), ``annotations`` (M, 13) fp64 rows in ``COLS`` order,
    sorted by sweep (``batch_index`` last).  Returns a new dict with all four augmented consistently, sweep by sweep.

    The batch must be UNPADDED (W = ``range_view_config["width"]``: 1800 / 2650, not 1808 / 2656): the reference augments the
    table before ``subsample_range_view`` pads it (``loader.py:598-603`` then ``:684-691``), and the roll of ``random_rotation``
    is modulo the configured width.  Pass ``width`` (the configured width) to have that checked; :func:`train_batch_from_tables`
    is the whole chain.  The mask is recomputed as the reference does (``range > 0`` on the augmented table): it follows the
    column map, except that a ``random_global_scale`` re-derives ``range`` from the coordinates at that point of the chain."""
    names = list(feature_column_names)
    feats, cart, mask = batch["features"], batch["cart"], batch["mask"]
    if width is not None and feats.shape[-1] != int(width):
        raise L.RvError(f"augment_batch on a batch of width {feats.shape[-1]}, configured width {width}: augment BEFORE the W padding "
                        "(range_view_from_table(..., pad=False) -> augment_batch -> pad_batch)")
    trs = [draw_sweep_transform(feats.shape[-1], augmentations_config, rng) for _ in range(feats.shape[0])]
    xyz = [names.index(n) for n in ("x", "y", "z")] if all(n in names for n in ("x", "y", "z")) else None
    out = dict(batch)
    out["features"] = apply_sweep_transforms(feats, trs, xyz, names.index("range") if "range" in names and xyz is not None else -1)
    out["cart"] = apply_sweep_transforms(cart, trs, (0, 1, 2))
    # mask' = (augmented range > 0): [x, y, z, valid] through the same kernel with `valid` in the range slot
    aux = apply_sweep_transforms(torch.cat([cart.float(), mask.float()], dim=1), trs, (0, 1, 2), 3)
    out["mask"] = aux[:, 3:4] > 0
    ann = batch.get("annotations")
    if ann is not None and ann.shape[0] > 0:
        ann = torch.as_tensor(ann).double()
        parts = [trs[b].apply_to_annotations(ann[ann[:, -1] == b]) for b in range(feats.shape[0])]
        out["annotations"] = torch.cat(parts) if parts else ann
    out["transforms"] = trs
    return out


def annotations_for_sweep(table: Mapping[str, Any], timestamp_ns: int, tasks: Mapping[int, Sequence[str]], batch_index: int = 0) -> Tensor:
    """The annotation rows ``__getitem__`` keeps for one sweep (``loader.py:583-589, 699-704``): ``timestamp_ns`` equal,
    ``num_interior_pts > 0``, category among the configured tasks; joined with the task frame (``task_id``, ``offset`` = index in
    the SORTED category list of the task, ``loader.py:553-565``) and stably sorted by (task_id, offset).  Returns (M, 13) fp64
    rows in ``COLS`` order (host work on a handful of rows)."""
    import numpy as np

    frame = {}
    for k, cats in tasks.items():
        for offset, c in enumerate(sorted(cats)):
            frame[c] = (int(k), offset)
    ts = np.asarray(table["timestamp_ns"])
    npts = np.asarray(table["num_interior_pts"])
    cat = [str(c) for c in np.asarray(table["category"]).tolist()]
    rows = [i for i in range(len(cat)) if int(ts[i]) == int(timestamp_ns) and int(npts[i]) > 0 and cat[i] in frame]
    rows.sort(key=lambda i: frame[cat[i]])  # Python's sort is stable, as polars' ``sort``
    out = torch.zeros((len(rows), 13), dtype=torch.float64)
    for j, name in enumerate(COLS[:10]):
        col = np.asarray(table[name], dtype=np.float64)
        out[:, j] = torch.from_numpy(col[rows]) if rows else out[:, j]
    for r, i in enumerate(rows):
        out[r, 10], out[r, 11], out[r, 12] = frame[cat[i]][0], frame[cat[i]][1], batch_index
    return out


def train_batch_from_tables(tables: Sequence[Mapping[str, Any]], annotations: Optional[Tensor], range_view_config: Mapping[str, Any],
                            dataset_name: str, augmentations_config: Optional[Mapping[str, Mapping[str, float]]], x_stride: int = 1,
                            padding_mode: str = "constant", rng=_random, device="cuda") -> Dict[str, Any]:
    """The train-split item chain of ``DataLoader.__getitem__`` (``loader.py:594-705``) for a batch of sweep tables, in the
    reference's order: ROI filter + table -> image (unpadded) -> augmentations -> ``features *= mask`` + W padding.
    ``annotations``: (M, 13) fp64 rows in ``COLS`` order (``batch_index`` = position in ``tables``) or None."""
    import numpy as np

    aug = dict(augmentations_config) if augmentations_config else {}
    keeps = [None] * len(tables)
    if aug and next(iter(aug)) == "point_dropout":
        # ``_point_dropout`` (loader.py:506-512): ``np.random.rand(rows, 1) <= p`` from numpy's GLOBAL generator, one draw per
        # sweep, every column of the table times the mask -- i.e. dropped pixels become empty pixels before anything else
        p = float(aug.pop("point_dropout")["p"])
        hw = int(range_view_config["height"]) * int(range_view_config["width"])
        keeps = [(np.random.rand(hw, 1) <= p).reshape(-1) for _ in tables]
    items = [range_view_from_table(t, range_view_config, dataset_name, x_stride, padding_mode, device, pad=False, keep=k) for t, k in zip(tables, keeps)]
    batch: Dict[str, Any] = {k: torch.stack([it[k] for it in items]) for k in ("features", "mask", "cart")}
    if annotations is not None:
        batch["annotations"] = annotations
    if aug:
        batch = augment_batch(batch, range_view_config["feature_column_names"], aug, rng, width=int(range_view_config["width"]))
    return pad_batch(batch, dataset_name, x_stride, padding_mode)
