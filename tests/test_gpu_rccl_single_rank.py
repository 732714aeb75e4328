"""The RCCL backend itself in the GPU suite: a one-rank "nccl" process group carries the packed tile buffers of the
pipelined frame loop (tests/rccl_worker.py), and the re-assembled frames equal the single-GPU frame."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_pipelined_gather_through_rccl_with_one_rank():
    run = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py")], capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "nccl single-rank pipelined path ok" in run.stdout
