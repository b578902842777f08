"""Oracle: the loader's per-sweep contract, table -> (features, cart, mask).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  numpy restatement of ``DataLoader.__getitem__``,
``/root/reference/src/torchbox3d/prototype/loader.py``:

* ``:594-601``  the sweep table; ``filter_roi``: EVERY column times the 0/1 ``is_within_roi`` column
* ``:625-634``  feature columns in ``feature_column_names`` order; Waymo: tanh(intensity); ``timedelta_ns`` times 1e-9
* ``:636-652``  ``_npy_to_tch`` (``:818-822``): column f of the (H*W, F) table is image f, row-major (H, W); mask = range > 0
* ``:690-697``  ``subsample_range_view`` (``:792-815``): features *= mask, W padded by [4,4] (av2) / [3,3] (waymo), constant or
                circular, every ``x_stride``-th column

Pinned by ``tests/golden/loader_item.npz`` (made by ``make_golden.py loader_item`` running the reference's own ``__getitem__``).
* ``:605-624``  the ``view`` feature (``row_mapping_64`` = the dataset's ROW_MAPPING_64, an argument).
"""

from __future__ import annotations

from typing import Dict, Mapping, Sequence

import numpy as np

_PAD = {("waymo", 1): 3, ("waymo", 4): 19, ("av2", 1): 4, ("av2", 4): 28}


def _pad_w(x: np.ndarray, pad: int, mode: str) -> np.ndarray:
    return np.pad(x, ((0, 0), (0, 0), (pad, pad)), mode="wrap" if mode == "circular" else "constant")


def range_view_from_table(table: Mapping[str, np.ndarray], feature_column_names: Sequence[str], height: int, width: int, dataset_name: str,
                          filter_roi: bool, x_stride: int = 1, padding_mode: str = "constant", row_mapping_64=None) -> Dict[str, np.ndarray]:
    names = list(feature_column_names)
    roi = np.asarray(table["is_within_roi"]).astype(np.float32) if filter_roi else np.float32(1.0)
    if "view" in names:  # loader.py:605-624 (after the ROI filter): reverse ROW_MAPPING_64 on the laser rows, then the view id
        assert row_mapping_64 is not None
        rev = {int(v): i for i, v in enumerate(np.asarray(row_mapping_64).tolist())}
        ln = np.asarray(table["laser_number"], dtype=np.float32) * roi
        pos = ((np.asarray(table["range"], dtype=np.float32) * roi) > 0).astype(np.float32)
        ln2 = ln.copy()
        for k, v in rev.items():
            ln2[ln == k] = v
        ln2 = ln2 * pos
        table = dict(table, laser_number=ln2, view=(2.0 * (ln2 <= 32).astype(np.float32) + (ln2 > 32).astype(np.float32)) * pos)
    col = lambda n: np.asarray(table[n], dtype=np.float32) * roi  # noqa: E731
    feats = []
    for n in names:
        v = col(n)
        if n == "intensity" and dataset_name == "waymo":
            v = np.tanh(v)
        if n == "timedelta_ns":
            v = (v.astype(np.float64) * 1e-9).astype(np.float32)
        feats.append(v)
    features = np.stack(feats).reshape(len(names), height, width)
    cart = np.stack([col(n) for n in ("x", "y", "z")]).reshape(3, height, width)
    mask = (col("range").reshape(1, height, width) > 0.0)
    pad = _PAD[(dataset_name, 4 if x_stride == 4 else 1)]
    features = features * mask
    return {"features": _pad_w(features, pad, padding_mode)[:, :, ::x_stride], "mask": _pad_w(mask, pad, padding_mode)[:, :, ::x_stride],
            "cart": _pad_w(cart, pad, padding_mode)[:, :, ::x_stride]}


def train_item_from_table(table: Mapping[str, np.ndarray], feature_column_names: Sequence[str], height: int, width: int, dataset_name: str,
                          filter_roi: bool, augmentations: Sequence[tuple], x_stride: int = 1, padding_mode: str = "constant") -> Dict[str, np.ndarray]:
    """``__getitem__`` with ``split_name == "train"`` (``loader.py:594-697``): the ROI filter, then the augmentations on the
    UNPADDED table (``:598-603``; every column travels through flips / rolls, ``x y z range`` are rewritten), then the feature /
    cart / mask images (mask = AUGMENTED ``range`` > 0) and ``subsample_range_view``.  ``augmentations``: the chain with its
    draws, e.g. ``[("dropout", keep), ("flip",), ("rotate", theta), ("scale", s), ("translate", (tx, ty, tz))]``.  Pinned by
    ``tests/golden/loader_train_item.npz``."""
    from . import augment as oaug

    cols = [n for n in table if n != "is_within_roi"]
    roi = np.asarray(table["is_within_roi"]).astype(np.float32) if filter_roi else np.float32(1.0)
    sweep = np.stack([np.asarray(table[n], dtype=np.float32) * roi for n in cols]).reshape(len(cols), height, width).astype(np.float64)
    ann = np.zeros((10, 0))
    for op in augmentations:
        if op[0] == "dropout":  # loader.py:506-512: every column times the keep mask
            sweep = sweep * np.asarray(op[1], dtype=np.float64).reshape(1, height, width)
        elif op[0] == "flip":
            sweep, _ = oaug.flip(sweep, cols, ann)
        elif op[0] == "rotate":
            sweep, _ = oaug.rotate(sweep, cols, ann, op[1])
        elif op[0] == "scale":
            sweep, _ = oaug.scale(sweep, cols, ann, op[1])
        elif op[0] == "translate":
            sweep, _ = oaug.translate(sweep, cols, ann, op[1])
        else:
            raise KeyError(op[0])
    aug = {n: sweep[i].reshape(-1).astype(np.float32) for i, n in enumerate(cols)}
    return range_view_from_table(aug, feature_column_names, height, width, dataset_name, False, x_stride, padding_mode)


def annotations_for_sweep(table: Mapping[str, np.ndarray], timestamp_ns: int, tasks: Mapping[int, Sequence[str]]):
    """``loader.py:583-589, 553-565, 699-704``: rows with this timestamp, ``num_interior_pts > 0`` and a configured category, joined
    with the task frame (offset = index in the task's SORTED category list), stably sorted by (task_id, offset).  Returns the
    kept row indices in output order and their (task_id, offset)."""
    frame = {c: (int(k), o) for k, cats in tasks.items() for o, c in enumerate(sorted(cats))}
    cat = [str(c) for c in np.asarray(table["category"]).tolist()]
    rows = [i for i in range(len(cat)) if int(table["timestamp_ns"][i]) == int(timestamp_ns) and int(table["num_interior_pts"][i]) > 0 and cat[i] in frame]
    rows.sort(key=lambda i: frame[cat[i]])
    return rows, [frame[cat[i]] for i in rows]
