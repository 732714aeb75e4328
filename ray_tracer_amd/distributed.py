"""Image-space sharding of one frame over the GPUs of a node: one process per GPU.

The reference parallelises a frame over 8 host threads that own contiguous column strips and
write one shared canvas (src/main.cpp:34-52).  Here every rank holds the whole scene in its own
HBM and renders the screen tiles k with k % world == rank into a packed buffer
([local tile][tile_h][tile_w][3]); the only exchange is one all-gather of those equally sized
buffers (RCCL over xGMI when the process group is "nccl"), after which a small kernel writes the
row-major frame.  Cost per pixel varies a lot across the image, so tiles are small and interleaved
rather than contiguous strips.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import LAYOUT_PACKED_TILES, LAYOUT_ROW_MAJOR, RenderParams, Scene, lib, make_params, untile


def shard_params(width: int, height: int, rank: int, world: int, *, tile: int = 32, **kw) -> RenderParams:
    """Render params of `rank`: its interleaved tiles, packed layout when there is more than one rank."""
    layout = LAYOUT_PACKED_TILES if world > 1 else LAYOUT_ROW_MAJOR
    return make_params(width, height, tile=(tile, tile), tile_rank=rank, tile_count=world, layout=layout, **kw)


def output_pixels(params: RenderParams) -> int:
    return int(lib().rtx_output_pixels(C.byref(params)))


def gather_packed(local: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """All-gathers the ranks' packed tile buffers into one [world * n] tensor (rank-major)."""
    if world == 1:
        return local
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(out, local, group=group)
    except (RuntimeError, NotImplementedError):
        parts = list(out.chunk(world))
        dist.all_gather(parts, local, group=group)
    return out


class ShardedRenderer:
    """Keeps the device buffers of one rank and renders whole frames cooperatively."""

    def __init__(self, scene: Scene, width: int, height: int, rank: int, world: int, device: torch.device, *,
                 tile: int = 32, group=None, **render_kw):
        self.scene, self.rank, self.world, self.group = scene, rank, world, group
        self.width, self.height = width, height
        self.params = shard_params(width, height, rank, world, tile=tile, **render_kw)
        n = output_pixels(self.params) * 3
        self.frame = torch.zeros(height * width * 3, dtype=torch.uint8, device=device)
        self.local = self.frame if world == 1 else torch.zeros(n, dtype=torch.uint8, device=device)
        self.gathered = None if world == 1 else torch.zeros(world * n, dtype=torch.uint8, device=device)

    def render(self) -> torch.Tensor:
        """One frame; returns the row-major (H*W*3,) uint8 tensor (valid on every rank)."""
        stream = torch.cuda.current_stream().cuda_stream
        self.scene.render_device(self.params, rgb_u8=self.local.data_ptr(), stream=stream)
        if self.world > 1:
            dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
            untile(self.params, self.world, self.gathered.data_ptr(), self.frame.data_ptr(), 3, stream)
        return self.frame
