"""The N > 1 path of bench.py end to end on ONE GPU: two ranks (one process each) share cuda:0 and talk over gloo, so the
DistributedDataParallel wrapper, the SyncBN statistic all-reduces, the side-stream weight gradients and the rank-0 JSON
line are exercised exactly as under ``torch.distributed.run`` on a node (there the backend is nccl = RCCL)."""

from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n: int, extra=()):
    env = dict(os.environ, RV3D_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--widths", "c32", "--width", "512", "--height", "16",
           "--batch", "2", "--classes", "5", "--no-cpu-baseline", *extra]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_ranks_sync_bn():
    j = _run(2)
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["config"]["sync_bn"] is True
    assert j["config"]["global_batch"] == 4 and j["value"] > 0 and j["scaling"] == "weak"
    assert 0.0 < j["config"]["loss"] < 100.0
    assert "cpu_baseline" not in j  # rank 0 at N = 1 only


def test_two_ranks_local_bn():
    j = _run(2, ["--no-sync-bn"])
    assert j["config"]["sync_bn"] is False and 0.0 < j["config"]["loss"] < 100.0
