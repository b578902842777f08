"""CPU oracle for the range-view hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a plain restatement (PyTorch-CPU / numpy / C, fp32 + fp64 exactly where
the reference uses them) of the algorithms on the hot path of
``benjaminrwilson/range-view-3d-detection`` (``torchbox3d``).  Every function cites the
reference ``file:line`` it follows (paths relative to ``/root/reference/``).

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` -- as the *checker* / reported CPU baseline only.  The shipped package
``range_view_3d_detection_amd`` never imports ``oracle`` and has no CPU fallback.

Pinning status:

* Everything except weighted NMS is pinned against outputs of the reference itself,
  run in the build container and committed as ``tests/golden/*.npz`` (generator:
  ``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks the oracle
  against those fixtures.
* ``oracle.nms`` wrapper logic (class loop, top-k cuts, merged-score ranking, float categories / batch index) is pinned by
  ``tests/golden/nms_wrapper.npz`` (the reference's own wrapper functions run over a stand-in for the absent extension).
* ``oracle.nms.weighted_nms``'s inner kernel -- **parity unpinned**: the arithmetic lives in the
  un-vendored, un-pinned third-party CUDA extension ``weighted_nms_ext`` (TorchEx),
  whose source and binary are absent.  The oracle states its own semantics (see
  ``oracle/nms.py``) and is checked against the reference's in-tree post-conditions
  (``math/ops/nms.py:173-174``) and call-site wrapper logic only.
"""
