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
