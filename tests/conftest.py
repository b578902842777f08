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
