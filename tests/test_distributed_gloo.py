"""N > 1 host path on CPU: two gloo ranks shard a frame into interleaved tiles, all-gather the
packed buffers and re-assemble it.  (Rendering itself needs the GPU; the tile arithmetic, buffer
sizes and gather order do not.)"""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("W,H,tile", [(200, 120, 16), (97, 45, 8), (64, 64, 32)])
def test_two_ranks_gather_and_untile(W, H, tile, tmp_path):
    port, world = _free_port(), 2
    res = str(tmp_path / "res")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), str(port),
                               str(W), str(H), str(tile), res]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    for r in range(world):
        assert open(f"{res}.{r}").read() == "ok"


def test_tile_ownership_partition():
    from scenes_util import owned_tiles

    W, H, t = 3840, 2160, 32
    all_tiles = sorted(k for r in range(8) for k in owned_tiles(W, H, t, t, r, 8))
    assert all_tiles == list(range(120 * 68))
    counts = [len(owned_tiles(W, H, t, t, r, 8)) for r in range(8)]
    assert max(counts) - min(counts) <= 1
