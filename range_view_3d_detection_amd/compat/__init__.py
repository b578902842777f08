"""Drop-in stand-ins for the third-party extension modules the reference imports by name.

Put this directory on ``sys.path`` (or ``PYTHONPATH``) and the reference's own ``torchbox3d/math/ops/nms.py`` runs unchanged:
its ``import weighted_nms_ext`` resolves to ``compat/weighted_nms_ext.py``, which forwards to ``librv3d_hip.so``.
"""
