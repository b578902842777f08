"""pytest configuration: registers the ``gpu`` marker and shared fixtures."""

from __future__ import annotations

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_global_rng_per_test(request):
    """Every test starts from a global torch / numpy RNG state that depends on ITS OWN name only: modules built without an explicit generator
    (``MetaKernel(...)``, ``Conv2d(...)``) used to draw from whatever state the tests before them had left, so a bound on the distance between two
    bf16 realisations could pass or fail with the composition of the suite (round 6: adding a test file moved one such bound from 0.025 to 0.057)."""
    import zlib

    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF
    torch.manual_seed(seed)
    np.random.seed(seed % (2**32 - 1))
    yield


class Golden:
    """Lazy view of a ``tests/golden/<name>.npz`` fixture generated from the reference."""

    def __init__(self, name: str) -> None:
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.keys = list(self._z.keys())

    def __getitem__(self, key: str) -> torch.Tensor:
        return torch.from_numpy(np.asarray(self._z[key]))

    def np(self, key: str) -> np.ndarray:
        return np.asarray(self._z[key])

    def __contains__(self, key: str) -> bool:
        return key in self._z

    def sub(self, prefix: str) -> dict:
        """All arrays under ``prefix/`` as {stripped key: tensor}."""
        p = prefix.rstrip("/") + "/"
        return {k[len(p):]: self[k] for k in self.keys if k.startswith(p)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name: str) -> Golden:
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load
