"""The sharded frame loop end to end with more than one rank: ranks share the one GPU of the box and exchange their
packed tiles over gloo; every rank must end up with the frame a single rank renders."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,W,H", [(2, 320, 200), (3, 200, 136), (8, 256, 256)])
def test_ranks_on_one_gpu_assemble_the_single_rank_frame(world, W, H, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res = str(tmp_path / "res")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(r), str(world), str(port),
                               str(W), str(H), res]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    for r in range(world):
        assert open(f"{res}.{r}").read() == "ok"
