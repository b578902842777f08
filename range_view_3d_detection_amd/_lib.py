"""ctypes binding of ``librv3d_hip.so`` (the C ABI declared in ``include/rv3d.h``).

There is deliberately no CPU fallback: if the library is missing or a call fails the host
code raises.  ``load()`` only dlopens the library (works without a GPU, which is what the
CPU test-suite checks); every compute entry point needs a device.
"""

from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RV3D_LIB") or os.path.join(_HERE, "librv3d_hip.so")  # (RV3D_LIB: A/B of two builds in one gpurun call)
# the same sources built with fp16 operands (csrc/common.h, RV_OPERAND_F16): inference under torch.autocast(dtype=float16)
LIB_PATH_F16 = os.environ.get("RV3D_LIB_F16") or os.path.join(_HERE, "librv3d_hip_f16.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "rv3d.h")

# flags (mirror include/rv3d.h)
IN_AFFINE, IN_RELU, OUT_F32, OUT_BIAS, OUT_STATS, OUT_ACCUM, OUT_RELU = 1, 2, 4, 8, 16, 32, 64
OUT_RES_RELU = 256
WGRAD_TORCH_LAYOUT = 128  # rv_tap_wgrad: result in dT[cu][cv][kh][kw] (no unpack pass)
# kernel-selection hints (rvTapShape.flags, per call: the library keeps no mutable state).  SELECT is OR-ed into every TapShape
# built while a `select(...)` block is active -- the parity tests' way of running the production kernels on crops / pinning a generation.
SEL_SMALL_GRIDS, SEL_SMALL_GRIDS6, SEL_NO_GEN6, SEL_NO_GEN5, SEL_NO_POINTWISE, SEL_NO_POINTWISE_BWD = 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 25
SELECT = 0
EW_RELU_A, EW_RELU_B, EW_RELU_OUT = 1, 2, 4
BNB_RELU_Z, BNB_RES_ACCUM, BNB_Y_FROM_INPUT = 1, 2, 4
STATS_SCRATCH_ROWS = 128


class TapGeom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("kh", "kw", "stride_w", "pad_h", "pad_w", "cu", "cv")]


class TapShape(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("N", "H", "Wu", "Wv", "ld_src", "ld_dst", "flags")]

    def __init__(self, *args, **kw) -> None:
        super().__init__(*args, **kw)
        self.flags |= SELECT


class select:
    """``with select(SEL_SMALL_GRIDS | ...):`` -- every tap-conv / weight-gradient call issued inside carries these RV_SEL_* hints."""

    def __init__(self, flags: int) -> None:
        self.flags = flags

    def __enter__(self):
        global SELECT
        self.old, SELECT = SELECT, SELECT | self.flags
        return self

    def __exit__(self, *exc):
        global SELECT
        SELECT = self.old


class BnbEpilogue(ctypes.Structure):
    """``rvBnbEpilogue`` of include/rv3d.h (rv_tap_data_grad_bnb)."""

    _fields_ = [("y", ctypes.c_void_p), ("ld_y", ctypes.c_int32), ("flags", ctypes.c_int32), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
                ("mean", ctypes.c_void_p), ("invstd", ctypes.c_void_p), ("partial", ctypes.c_void_p)]


class RvError(RuntimeError):
    pass


_lib: Optional[ctypes.CDLL] = None
_lib_f16: Optional[ctypes.CDLL] = None
_OPERAND = "bf16"  # the operand type of the calls being issued: "bf16" (librv3d_hip.so) or "f16" (librv3d_hip_f16.so)


class operand:
    """``with operand("f16"):`` -- every ``call`` / ``load`` inside goes to the fp16-operand build of the library, and
    ``act_dtype()`` is ``torch.float16``.  Set by ``program._ProgramFn`` for an eval-mode program under
    ``torch.autocast(dtype=torch.float16)`` (the reference's ``eval_precision: 16``)."""

    def __init__(self, tag: str) -> None:
        if tag not in ("bf16", "f16"):
            raise RvError(f"unknown operand type {tag!r}")
        self.tag = tag

    def __enter__(self):
        global _OPERAND
        self.old, _OPERAND = _OPERAND, self.tag
        return self

    def __exit__(self, *exc):
        global _OPERAND
        _OPERAND = self.old


def operand_tag() -> str:
    return _OPERAND


def act_dtype() -> "torch.dtype":
    """torch dtype of the 16-bit activation tensors of the current operand type."""
    return torch.float16 if _OPERAND == "f16" else torch.bfloat16


def declared_symbols() -> List[str]:
    """Every function name declared in include/rv3d.h (used by the export test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rv_[a-z0-9_]+)\s*\(", text)))


def load(tag: Optional[str] = None) -> ctypes.CDLL:
    global _lib, _lib_f16
    tag = tag or _OPERAND
    if tag == "f16":
        if _lib_f16 is None:
            _lib_f16 = _dlopen(LIB_PATH_F16)
        return _lib_f16
    if _lib is None:
        _lib = _dlopen(LIB_PATH)
    return _lib


def _dlopen(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise RvError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    # Load order matters on a GPU box: the library registers its code objects with the HIP runtime when it is loaded, and
    # it must find the runtime PyTorch already initialised (loaded the other way round -- e.g. build() followed by
    # smoke() in one process -- its launches failed with "no ROCm-capable device is detected").  device_count() does not
    # touch the GPU, so CPU-only boxes (the build check) pass through.
    if torch.cuda.device_count() > 0 and torch.cuda.is_available():
        torch.cuda.init()
    lib = ctypes.CDLL(path)
    lib.rv_last_error.restype = ctypes.c_char_p
    for name in ("rv_packed_weight_bytes", "rv_decode_num_candidates", "rv_wnms_workspace_bytes", "rv_tap_wgrad_workspace_bytes", "rv_bn_bwd_smallk_workspace_bytes", "rv_smallk_forward_workspace_bytes", "rv_nms_sweeps_workspace_bytes", "rv_pack_batch_entry_bytes"):
        if hasattr(lib, name):
            getattr(lib, name).restype = ctypes.c_int64
    return lib


def _as_arg(v):
    if v is None:
        return ctypes.c_void_p(0)
    if isinstance(v, (ctypes.Structure,)):
        return ctypes.byref(v)
    if isinstance(v, float):
        return ctypes.c_float(v)
    if isinstance(v, int):
        return ctypes.c_int64(v) if abs(v) > 0x7FFFFFFF else ctypes.c_int32(v)
    return v


_DEBUG = bool(int(os.environ.get("RV3D_DEBUG_SYNC", "0")))


def _v(a):
    return getattr(a, "value", a)


def _px_c(a):
    return float(_v(a[0])) * float(_v(a[1]))


# Algorithmic bytes (every operand tensor once in, every result once out, 16-bit activations) of the HBM-bound entry
# points, from their own arguments -- bench.py's `roofline_hbm` prices the group against the HBM roofline.  Argument positions as
# declared in include/rv3d.h.
HBM_BYTES = {
    "rv_ew_combine": lambda a: 2.0 * _px_c(a) * (2 + (_v(a[6]) is not None)),
    "rv_bn_bwd_reduce": lambda a: 2.0 * _px_c(a) * (2 + (_v(a[4]) is not None)),
    "rv_bn_bwd_apply": lambda a: 2.0 * _px_c(a) * (3 + (_v(a[4]) is not None) + 2 * (_v(a[16]) is not None)),
    "rv_bn_bwd_reduce_pair": lambda a: 2.0 * _px_c(a) * 4,
    "rv_bn_bwd_apply_pair": lambda a: 2.0 * _px_c(a) * 6,
    # MetaKernel stem: pixels = N H W, the 9x-grid tensors hold 9 C values per pixel
    "rv_meta_modulate": lambda a: 2.0 * _v(a[5]) * _v(a[6]) * _v(a[7]) * _v(a[8]) * 19,
    "rv_meta_modulate_bwd_sums": lambda a: 2.0 * _v(a[8]) * _v(a[9]) * _v(a[10]) * _v(a[11]) * 20,
    "rv_meta_modulate_bwd_apply": lambda a: 2.0 * _v(a[9]) * _v(a[10]) * _v(a[11]) * _v(a[12]) * 28,
    "rv_pos_forward": lambda a: float(_v(a[3])) * (16 + 4.0 * _v(a[9])),
    "rv_pos_backward_sums": lambda a: float(_v(a[0])) * (16 + 2.0 * _v(a[1])),
    # final conv of a tower fused with the BatchNorm backward in front of it: y (+ the 32-channel dY) in; _apply also writes dy
    "rv_head_final_bwd_sums": lambda a: 2.0 * _v(a[0]) * (_v(a[1]) + 32),
    "rv_head_final_bwd_apply": lambda a: 2.0 * _v(a[0]) * (2 * _v(a[1]) + 32),
}
HBM_HOOK = None  # callable(name, algorithmic bytes, launch) or None: set by bench.py around its HBM-group measurement


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point; raise with rv_last_error() on failure.

    ``RV3D_DEBUG_SYNC=1`` prints every call and synchronises after it (locates a faulting launch).
    """
    if HBM_HOOK is not None and name in HBM_BYTES:
        hook, nbytes = HBM_HOOK, HBM_BYTES[name](args)
        return hook(name, nbytes, lambda: _call(name, *args))
    return _call(name, *args)


def _call(name: str, *args) -> None:
    fn = getattr(load(), name)
    if _DEBUG:
        import sys

        import torch

        print(f"[rv3d] {name}", file=sys.stderr, flush=True)
    rc = fn(*args)
    if rc != 0:
        raise RvError(f"{name} failed: {load().rv_last_error().decode()}")
    if _DEBUG:
        torch.cuda.synchronize()


def ptr(t) -> ctypes.c_void_p:
    """Device (or host) pointer of a torch tensor / None."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr() -> ctypes.c_void_p:
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


i32 = ctypes.c_int32
i64 = ctypes.c_int64
f32 = ctypes.c_float
f64 = ctypes.c_double
