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
    """Keeps the device buffers of one rank and renders whole frames cooperatively.

    `inflight` frames may be in progress at once: frame k is rendered, gathered and re-assembled on
    stream k % inflight with its own buffers, so the end of a frame (lanes running dry), its
    all-gather and its re-assembly overlap the next frame's render.  4 = one stream per hardware
    queue of the process.
    """

    def __init__(self, scene: Scene, width: int, height: int, rank: int, world: int, device: torch.device, *,
                 tile: int = 32, group=None, inflight: int = 1, **render_kw):
        self.scene, self.rank, self.world, self.group = scene, rank, world, group
        self.width, self.height = width, height
        self.params = shard_params(width, height, rank, world, tile=tile, **render_kw)
        self.inflight = max(1, min(8, int(inflight)))
        n = output_pixels(self.params) * 3
        self.frames = [torch.zeros(height * width * 3, dtype=torch.uint8, device=device) for _ in range(self.inflight)]
        self.locals = self.frames if world == 1 else [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(self.inflight)]
        self.gathers = [None if world == 1 else torch.zeros(world * n, dtype=torch.uint8, device=device)
                        for _ in range(self.inflight)]
        with torch.cuda.device(device):
            self.streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=device) for _ in range(self.inflight - 1)]
        # a process group without device collectives (gloo) gathers through host memory: rehearsals only
        self._host_gather = world > 1 and dist.get_backend(group) != "nccl"
        self._next = 0

    def submit(self, *, events=None, ticks: int = 0) -> int:
        """Queues the next frame and returns its slot; `frames[slot]` is complete once `streams[slot]` has
        drained.  events: an optional (start, end) pair of torch events recorded around the render kernel;
        ticks: device pointer for rtx_render_device_timed's two clock values."""
        slot = self._next % self.inflight
        self._next += 1
        stream = self.streams[slot]
        with torch.cuda.stream(stream):
            if events is not None:
                events[0].record()
            self.scene.render_device(self.params, rgb_u8=self.locals[slot].data_ptr(), ticks=ticks, stream=stream.cuda_stream)
            if events is not None:
                events[1].record()
            if self.world > 1:
                if self._host_gather:
                    parts = [torch.empty(self.locals[slot].numel(), dtype=torch.uint8) for _ in range(self.world)]
                    dist.all_gather(parts, self.locals[slot].cpu(), group=self.group)
                    self.gathers[slot].copy_(torch.cat(parts))
                else:
                    dist.all_gather_into_tensor(self.gathers[slot], self.locals[slot], group=self.group)
                untile(self.params, self.world, self.gathers[slot].data_ptr(), self.frames[slot].data_ptr(), 3, stream.cuda_stream)
        return slot

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()

    def render(self) -> torch.Tensor:
        """One frame, finished; returns the row-major (H*W*3,) uint8 tensor (valid on every rank)."""
        slot = self.submit()
        self.streams[slot].synchronize()
        return self.frames[slot]
