"""Stand-in modules that let the *reference's own Python* import in this container.

Used ONLY by ``tests/golden/make_golden.py`` (the committed script that generated the
``tests/golden/*.npz`` fixtures by running the reference on CPU).  Nothing in the
product, the oracle, or the test-suite imports this file at run time: the reference does
not exist on the GPU box.

The reference (``/root/reference/src/torchbox3d``) depends on third-party packages that
are absent here.  Each stub below restates only the *interface* the
hot path touches; behaviour that matters numerically is restated faithfully:

* ``torchvision.ops.Conv2dNormActivation`` -- Sequential[Conv2d(bias = norm is None),
  norm, activation(inplace)], default padding ``(k-1)//2*dilation``, accepts
  ``padding="same"``; sub-module names "0","1","2" are part of the state-dict keys.
* ``kornia.geometry.conversions`` -- yaw-only quaternion helpers.
* ``omegaconf.DictConfig/ListConfig`` -- hashable attr-dict / list.
* ``hydra.utils.instantiate`` -- non-recursive ``_target_`` instantiation.
* ``polars`` -- dtype names + a small columnar frame (``select`` / ``with_columns`` / ``to_numpy`` / ``from_numpy`` /
  ``pl.col`` scalar arithmetic with polars' dtype rules): what the loader's augmentations use.
* ``numba.njit`` -- identity decorator (the z-buffer loop runs as plain Python).
"""

from __future__ import annotations

import enum
import importlib
import math
import sys
import types
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import nn


def _module(name: str) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__path__ = []  # behave like a package
    sys.modules[name] = mod
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], child, mod)
    return mod


# --------------------------------------------------------------------------------------
# torchvision.ops
# --------------------------------------------------------------------------------------
class Conv2dNormActivation(nn.Sequential):
    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        kernel_size=3,
        stride=1,
        padding=None,
        groups: int = 1,
        norm_layer=nn.BatchNorm2d,
        activation_layer=nn.ReLU,
        dilation=1,
        inplace: Optional[bool] = True,
        bias: Optional[bool] = None,
    ) -> None:
        if padding is None:
            if isinstance(kernel_size, int) and isinstance(dilation, int):
                padding = (kernel_size - 1) // 2 * dilation
            else:
                ks = tuple(kernel_size) if not isinstance(kernel_size, int) else (kernel_size,) * 2
                dl = tuple(dilation) if not isinstance(dilation, int) else (dilation,) * 2
                padding = tuple((k - 1) // 2 * d for k, d in zip(ks, dl))
        if bias is None:
            bias = norm_layer is None
        if not isinstance(kernel_size, int):
            kernel_size = tuple(kernel_size)
        layers: List[nn.Module] = [
            nn.Conv2d(
                in_channels,
                out_channels,
                kernel_size,
                stride,
                padding,
                dilation=dilation,
                groups=groups,
                bias=bias,
            )
        ]
        if norm_layer is not None:
            layers.append(norm_layer(out_channels))
        if activation_layer is not None:
            params = {} if inplace is None else {"inplace": inplace}
            layers.append(activation_layer(**params))
        super().__init__(*layers)
        self.out_channels = out_channels


def _sigmoid_focal_loss(*args: Any, **kwargs: Any):  # never called on the configured path
    raise NotImplementedError


# --------------------------------------------------------------------------------------
# omegaconf / hydra
# --------------------------------------------------------------------------------------
class DictConfig(dict):
    """Attribute dict; hashable by identity (nn.Module machinery hashes dataclass fields)."""

    def __getattr__(self, key: str) -> Any:
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key: str, value: Any) -> None:
        self[key] = value

    def __hash__(self) -> int:  # type: ignore[override]
        return id(self)

    def __eq__(self, other: Any) -> bool:
        return self is other


class ListConfig(list):
    def __hash__(self) -> int:  # type: ignore[override]
        return id(self)

    def __eq__(self, other: Any) -> bool:
        return self is other


def _instantiate(cfg: Dict[str, Any], *args: Any, **kwargs: Any) -> Any:
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.pop("_recursive_", None)
    module_name, _, attr = target.rpartition(".")
    cls = getattr(importlib.import_module(module_name), attr)
    cfg.update(kwargs)
    return cls(*args, **cfg)


# --------------------------------------------------------------------------------------
# kornia.geometry.conversions (yaw-only use on the hot path)
# --------------------------------------------------------------------------------------
def quaternion_from_euler(roll, pitch, yaw):
    cy, sy = torch.cos(yaw * 0.5), torch.sin(yaw * 0.5)
    cp, sp = torch.cos(pitch * 0.5), torch.sin(pitch * 0.5)
    cr, sr = torch.cos(roll * 0.5), torch.sin(roll * 0.5)
    qw = cr * cp * cy + sr * sp * sy
    qx = sr * cp * cy - cr * sp * sy
    qy = cr * sp * cy + sr * cp * sy
    qz = cr * cp * sy - sr * sp * cy
    return qw, qx, qy, qz


def euler_from_quaternion(w, x, y, z):
    sinr_cosp = 2.0 * (w * x + y * z)
    cosr_cosp = 1.0 - 2.0 * (x * x + y * y)
    roll = torch.atan2(sinr_cosp, cosr_cosp)
    sinp = (2.0 * (w * y - z * x)).clamp(-1.0, 1.0)
    pitch = torch.asin(sinp)
    siny_cosp = 2.0 * (w * z + x * y)
    cosy_cosp = 1.0 - 2.0 * (y * y + z * z)
    yaw = torch.atan2(siny_cosp, cosy_cosp)
    return roll, pitch, yaw


def quaternion_to_rotation_matrix(q, order=None):
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    m = torch.stack(
        [
            1 - 2 * (y * y + z * z),
            2 * (x * y - w * z),
            2 * (x * z + w * y),
            2 * (x * y + w * z),
            1 - 2 * (x * x + z * z),
            2 * (y * z - w * x),
            2 * (x * z - w * y),
            2 * (y * z + w * x),
            1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return m.reshape(q.shape[:-1] + (3, 3))


class QuaternionCoeffOrder(enum.Enum):
    XYZW = "xyzw"
    WXYZ = "wxyz"


# --------------------------------------------------------------------------------------
# polars (just enough for utils/polars.py and the dtype names in math/ops/coding.py)
# --------------------------------------------------------------------------------------
class _PlExpr:
    """``pl.col(...)`` / ``pl.lit(...)`` and the arithmetic / comparisons the loader and the converter use on them.  Polars keeps
    the COLUMN's dtype when the other operand is a Python scalar (``scale * pl.col("x")`` on a Float32 column stays Float32)
    -- reproduced with a same-dtype numpy scalar."""

    def __init__(self, fn, names=None) -> None:
        self.fn = fn
        self.names = names  # column selector (``pl.col("a", "b")`` / ``pl.col(["a", "b"])``) when not None

    def _bin(self, other, op, keep_dtype=True):
        def run(frame):
            a = self.fn(frame)
            if isinstance(other, _PlExpr):
                b = other.fn(frame)
            elif isinstance(other, _PlSeries):
                b = other.values
            else:
                b = np.asarray(other, dtype=a.dtype) if keep_dtype and np.issubdtype(np.asarray(a).dtype, np.floating) else other
            return op(a, b)

        return _PlExpr(run)

    def __mul__(self, o):
        return self._bin(o, lambda a, b: a * b)

    __rmul__ = __mul__

    def __add__(self, o):
        return self._bin(o, lambda a, b: a + b)

    __radd__ = __add__

    def __eq__(self, o):  # noqa: D105
        return self._bin(o, lambda a, b: a == b, keep_dtype=False)

    def eq(self, o):
        return self == o

    def __gt__(self, o):
        return self._bin(o, lambda a, b: a > b, keep_dtype=False)

    def __lt__(self, o):
        return self._bin(o, lambda a, b: a < b, keep_dtype=False)

    def __le__(self, o):
        return self._bin(o, lambda a, b: a <= b, keep_dtype=False)

    def replace(self, mapping):
        """``Expr.replace({old: new})``: values equal to a key become its value, everything else stays; the column keeps its dtype."""
        def run(frame):
            a = np.asarray(self.fn(frame))
            out = a.copy()
            for k, v in mapping.items():
                out[a == k] = v
            return out

        return _PlExpr(run)

    def __and__(self, o):
        return self._bin(o, lambda a, b: a & b, keep_dtype=False)

    def gt(self, o):
        return self > o

    def is_in(self, values):
        vals = list(values)
        return _PlExpr(lambda frame: np.isin(self.fn(frame), vals))

    def tanh(self):  # polars keeps the column's float dtype
        return _PlExpr(lambda frame: np.tanh(self.fn(frame)))

    def cast(self, dtype):
        npdt = {"Float32": np.float32, "Float64": np.float64, "Int64": np.int64}.get(dtype, dtype)
        return _PlExpr(lambda frame: np.asarray(self.fn(frame)).astype(npdt))

    def __rmul__(self, o):
        return self._bin(o, lambda a, b: a * b)

    __hash__ = None


class _PlSeries:
    def __init__(self, name: str, values) -> None:
        self.name, self.values = name, np.asarray(values)

    def min(self):
        return self.values.min()

    def max(self):
        return self.values.max()

    def to_numpy(self, writable: bool = False):
        return self.values.copy()

    def cast(self, dtype):
        return _PlSeries(self.name, self.values.astype({"Int64": np.int64, "Float32": np.float32, "Float64": np.float64}.get(dtype, dtype)))

    def search_sorted(self, other, side: str = "left"):
        other = other.values if isinstance(other, _PlSeries) else np.asarray(other)
        return np.searchsorted(self.values, other, side=side)

    def __add__(self, o):
        return _PlSeries(self.name, self.values + (o.values if isinstance(o, _PlSeries) else o))

    __radd__ = __add__

    def __len__(self):
        return len(self.values)

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    @property
    def dtype(self):
        return self.values.dtype


def _names_of(columns):
    if isinstance(columns, _PlExpr):
        assert columns.names is not None
        return list(columns.names)
    if isinstance(columns, str):
        return [columns]
    out = []
    for c in columns:
        out += _names_of(c) if isinstance(c, (_PlExpr, list, tuple)) else [c]
    return out


_NPDT = {"Float32": np.float32, "Float64": np.float64, "UInt8": np.uint8, "UInt32": np.uint32, "Boolean": np.bool_, "Int64": np.int64,
         "Int32": np.int32}


class _PlFrame:
    """Columnar frame: name -> 1-D numpy array (dtype kept per column, as polars does).  Eager and "lazy" are the same object."""

    def __init__(self, data: Dict[str, Sequence[Any]], schema=None, schema_overrides=None, **_: Any) -> None:
        schema = schema if schema is not None else schema_overrides  # (overrides: only the columns present are cast)
        self._data = {}
        for k, v in data.items():
            v = v.values if isinstance(v, _PlSeries) else np.asarray(v)
            if schema is not None and k in schema and schema[k] != "Utf8":
                v = v.astype(_NPDT.get(schema[k], schema[k]))
            self._data[k] = v

    def select(self, *columns) -> "_PlFrame":
        names = _names_of(columns[0] if len(columns) == 1 else list(columns))
        return _PlFrame({c: self._data[c] for c in names})

    def to_numpy(self, writable: bool = False) -> np.ndarray:
        cols = list(self._data.values())
        if not cols:
            return np.zeros((0, 0))
        if all(c.dtype.kind in "US" for c in cols):
            return np.stack(cols, axis=1)
        dt = np.result_type(*[c.dtype for c in cols])
        if not np.issubdtype(dt, np.floating):
            dt = np.float64 if dt.kind not in "iu" else dt
        return np.stack([np.asarray(c, dtype=dt) for c in cols], axis=1)  # mixed Float32 / Float64 columns upcast

    def with_columns(self, *exprs: Any, **named: Any) -> "_PlFrame":
        out = dict(self._data)
        for e in exprs:
            if isinstance(e, _PlSeries):
                out[e.name] = e.values
        for k, v in named.items():
            if isinstance(v, _PlExpr):
                v = v.fn(self._data)
            elif isinstance(v, _PlSeries):
                v = v.values
            v = np.asarray(v)
            n = self.shape[0]
            out[k] = np.broadcast_to(v, (n,)).copy() if v.ndim == 0 else v
        return _PlFrame(out)

    def filter(self, mask) -> "_PlFrame":
        m = mask.fn(self._data) if isinstance(mask, _PlExpr) else np.asarray(mask)
        return _PlFrame({k: v[m] for k, v in self._data.items()})

    def with_row_count(self, name: str = "row_nr") -> "_PlFrame":
        return _PlFrame({name: np.arange(self.shape[0], dtype=np.uint32), **self._data})

    def cast(self, dtypes: Dict[str, Any]) -> "_PlFrame":
        return _PlFrame({k: (v.astype(_NPDT.get(dtypes[k], dtypes[k])) if k in dtypes else v) for k, v in self._data.items()})

    def drop(self, columns) -> "_PlFrame":
        gone = set(_names_of(columns))
        return _PlFrame({k: v for k, v in self._data.items() if k not in gone})

    def join(self, other: "_PlFrame", on: str, how: str = "inner") -> "_PlFrame":
        """Inner join on one key: rows in LEFT order (each left row once per matching right row, right rows in their order);
        columns = the left frame's, then the right frame's non-key columns -- polars' inner-join output."""
        assert how == "inner"
        lk, rk = self._data[on], other._data[on]
        li, ri = [], []
        for i, key in enumerate(lk.tolist()):
            for j in np.nonzero(rk == key)[0].tolist():
                li.append(i)
                ri.append(j)
        li, ri = np.asarray(li, dtype=np.int64), np.asarray(ri, dtype=np.int64)
        out = {k: v[li] for k, v in self._data.items()}
        for k, v in other._data.items():
            if k != on:
                out[k if k not in out else k + "_right"] = v[ri]
        return _PlFrame(out)

    def sort(self, by) -> "_PlFrame":
        keys = [self._data[k] for k in reversed(_names_of(by))]
        order = np.lexsort(keys) if self.shape[0] else np.zeros(0, dtype=np.int64)  # lexsort is stable, last key primary
        return _PlFrame({k: v[order] for k, v in self._data.items()})

    def row(self, index: int):
        return tuple(v[index].item() if hasattr(v[index], "item") else v[index] for v in self._data.values())

    def __mul__(self, other):
        """``frame * series``: every column times the series, element-wise (a Boolean series acts as 0 / 1; the column keeps its
        dtype when it is a float, as in polars' arithmetic supertype rules for Float32 x Boolean)."""
        o = other.values if isinstance(other, _PlSeries) else np.asarray(other)
        out = {}
        for k, v in self._data.items():
            w = o.astype(v.dtype) if o.dtype == np.bool_ and v.dtype.kind in "fiu" else o
            out[k] = (v.astype(np.uint8) * w.astype(np.uint8)).astype(np.bool_) if v.dtype == np.bool_ else v * w
        return _PlFrame(out)

    def collect(self) -> "_PlFrame":
        return self

    def lazy(self) -> "_PlFrame":
        return self

    @property
    def schema(self) -> Dict[str, Any]:
        return {k: v.dtype for k, v in self._data.items()}

    @property
    def columns(self):
        return list(self._data)

    @property
    def shape(self):
        n = len(next(iter(self._data.values()))) if self._data else 0
        return (n, len(self._data))

    def __getitem__(self, key):
        if isinstance(key, str):
            return _PlSeries(key, self._data[key])
        idx = np.asarray(key)
        return _PlFrame({k: v[idx] for k, v in self._data.items()})  # rows by integer array


def _pl_col(*names):
    if len(names) == 1 and isinstance(names[0], (list, tuple)):
        names = tuple(names[0])
    if len(names) == 1:
        n = names[0]
        return _PlExpr(lambda frame: np.asarray(frame[n]), names=[n])
    return _PlExpr(None, names=list(names))


def _pl_scan_ipc(path, **_: Any) -> _PlFrame:
    import pyarrow.feather as feather

    t = feather.read_table(str(path))
    return _PlFrame({name: t.column(name).to_numpy(zero_copy_only=False) for name in t.column_names})


def _pl_from_numpy(data: np.ndarray, schema: Dict[str, Any]) -> _PlFrame:
    return _PlFrame({name: np.asarray(data[:, i], dtype=dt) for i, (name, dt) in enumerate(schema.items())})


def install() -> None:
    """Register every stub in ``sys.modules`` and put the reference on ``sys.path``."""
    import os

    os.environ.setdefault("PYTORCH_JIT", "0")
    ref_src = "/root/reference/src"
    if ref_src not in sys.path:
        sys.path.insert(0, ref_src)

    tv = _module("torchvision")
    ops = _module("torchvision.ops")
    ops.Conv2dNormActivation = Conv2dNormActivation
    ops.sigmoid_focal_loss = _sigmoid_focal_loss
    tv.ops = ops

    oc = _module("omegaconf")
    oc.DictConfig = DictConfig
    oc.ListConfig = ListConfig
    oc.MISSING = "???"

    class _OmegaConf:
        @staticmethod
        def register_new_resolver(*_: Any, **__: Any) -> None:
            return None

    oc.OmegaConf = _OmegaConf

    _module("hydra")
    hu = _module("hydra.utils")
    hu.instantiate = _instantiate

    plm = _module("pytorch_lightning")
    plm.LightningModule = nn.Module
    plm.LightningDataModule = object
    _module("pytorch_lightning.core")
    core_mod = _module("pytorch_lightning.core.module")
    core_mod.LightningModule = nn.Module

    pl = _module("polars")
    for name in ("Float32", "Float64", "UInt8", "UInt16", "UInt32", "Int32", "Int64", "Utf8", "Boolean"):
        setattr(pl, name, name)
    pl.DataFrame = _PlFrame
    pl.LazyFrame = _PlFrame
    pl.col = _pl_col
    pl.lit = lambda v: _PlExpr(lambda frame: np.asarray(v))
    pl.Series = _PlSeries
    pl.from_numpy = _pl_from_numpy

    class _PlConfig:
        @staticmethod
        def set_tbl_rows(*_: Any) -> None:
            return None

    pl.Config = _PlConfig
    pl.__dict__["__getattr__"] = lambda name: type(name, (), {})  # type names used in annotations only

    _module("kornia")
    _module("kornia.geometry")
    kc = _module("kornia.geometry.conversions")
    kc.quaternion_from_euler = quaternion_from_euler
    kc.euler_from_quaternion = euler_from_quaternion
    kc.quaternion_to_rotation_matrix = quaternion_to_rotation_matrix
    kc.QuaternionCoeffOrder = QuaternionCoeffOrder

    nbm = _module("numba")

    def njit(*args: Any, **kwargs: Any):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    nbm.njit = njit

    _module("detectron2")
    _module("detectron2.layers")
    d2 = _module("detectron2.layers.nms")
    d2.nms_rotated = None
    wn = _module("weighted_nms_ext")
    wn.wnms_gpu = None  # binary absent: weighted-NMS arithmetic cannot be run here
    _module("mmcv")
    mo = _module("mmcv.ops")
    mo.box_iou_rotated = None
    mo.boxes_iou3d = None
    mob = _module("mmcv.ops.box_iou_rotated")
    mob.box_iou_rotated = None

    # av2 (only for converters/av2/utils.py -> datasets/argoverse/constants.py)
    _module("av2")
    _module("av2.geometry")
    se3 = _module("av2.geometry.se3")

    class SE3:  # av2.geometry.se3.SE3 as the converter uses it: inverse() and transform_point_cloud() (row-vector points)
        def __init__(self, rotation, translation) -> None:
            self.rotation, self.translation = np.asarray(rotation, dtype=np.float64), np.asarray(translation, dtype=np.float64)

        def inverse(self):
            return SE3(self.rotation.T, self.rotation.T.dot(-self.translation))

        def transform_point_cloud(self, pts):
            return np.asarray(pts) @ self.rotation.T + self.translation

    se3.SE3 = SE3
    _module("av2.datasets")
    _module("av2.datasets.sensor")
    av2c = _module("av2.datasets.sensor.constants")

    class AnnotationCategories(str, enum.Enum):
        REGULAR_VEHICLE = "REGULAR_VEHICLE"

    av2c.AnnotationCategories = AnnotationCategories
    ev = _module("av2.evaluation")
    ev.SensorCompetitionCategories = AnnotationCategories
    _module("av2.evaluation.detection")
    eve = _module("av2.evaluation.detection.eval")
    eve.DetectionCfg = object


def load_converter_utils():
    """Import ``/root/reference/converters/av2/utils.py`` as a module (it is not a package)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "ref_converters_av2_utils", "/root/reference/converters/av2/utils.py"
    )
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
