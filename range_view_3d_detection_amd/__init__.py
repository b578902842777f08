"""range_view_3d_detection_amd -- MI355X-native hot path of the range-view LiDAR detector.

Drop-in for the data-parallel forward/backward path of ``torchbox3d``
(benjaminrwilson/range-view-3d-detection): the module tree under ``nn/`` and ``math/``
mirrors the reference's import paths, constructor signatures, ``forward``/``decode``
contracts and ``state_dict`` keys, so swapping the Hydra ``_target_`` prefix
``torchbox3d.`` -> ``range_view_3d_detection_amd.`` in ``conf/model/range_view.yaml`` is the
whole integration.  All compute runs in hand-written HIP kernels for
gfx950 behind the C ABI of ``include/rv3d.h``; there is no CPU fallback.
"""

__version__ = "0.1"


import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A training step here uses
# the compute stream, the weight-gradient side stream and -- data parallel -- ProcessGroupNCCL's stream; with four queues the side
# stream ended up sharing a queue with another one as soon as a process group existed, and the step lost its overlap: +2.6..3 ms per
# rv-av2 step from merely initialising RCCL, +5.1 with the collectives, against +1.6..3.2 (and -0.5 ms with no process group) at 8
# (profiles/r04_hw_queues.md).  Read by the runtime when it initialises, i.e. at the first HIP call after this import; an explicit
# setting wins.  The opt-in direct RCCL binding (rccl.py) is the exception: 124 ms per step at 8 queues against 101 at the default.
if _os.environ.get("RV3D_DIRECT_RCCL") is None:
    if "GPU_MAX_HW_QUEUES" not in _os.environ:
        import torch as _torch

        if _torch.cuda.is_initialized():  # too late for this process: say so instead of silently running with four queues
            import warnings as _warnings

            _warnings.warn("range_view_3d_detection_amd: HIP was initialised before this import, so GPU_MAX_HW_QUEUES=8 cannot take "
                           "effect; the weight-gradient side stream will share a hardware queue once an RCCL process group exists "
                           "(+2.6..5 ms per rv-av2 step).  Import the package (or export GPU_MAX_HW_QUEUES=8) before the first CUDA/HIP call.")
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
