"""Kernel time of one rank's share (tiles k = rank mod world) of a 4K frame on one GPU, with 1 and 3 frames in flight:
what strong scaling over `world` GPUs can reach before the gather."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import ctypes as C
import torch
import ray_tracer_amd as rt
FL = [int(x) for x in os.environ.get("FLIST", "1,4").split(",")]

host = rt.HostScene.standin("dragon", seed=0, no_bvh=True)
dev = rt.Scene(host)
W, H = 3840, 2160
TILE = int(os.environ.get("TILE", "32"))
for world in (1, 2, 4, 8):
    p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, tile=(TILE, TILE), tile_rank=0, tile_count=world,
                       layout=rt.LAYOUT_PACKED_TILES if world > 1 else rt.LAYOUT_ROW_MAJOR)
    px = int(rt.lib().rtx_output_pixels(C.byref(p)))
    out = {}
    for F in FL:
        streams = [torch.cuda.Stream() for _ in range(F)]
        bufs = [torch.zeros(px * 3, dtype=torch.uint8, device="cuda") for _ in range(F)]
        def run(n):
            for k in range(n):
                dev.render_device(p, rgb_u8=bufs[k % F].data_ptr(), stream=streams[k % F].cuda_stream)
            torch.cuda.synchronize()
        run(6)
        t0 = time.perf_counter(); run(60); out[F] = (time.perf_counter() - t0) / 60 * 1e3
    print(f"world {world}: tile {TILE} rank-0 share per frame, {FL} in flight: " + " ".join(f"{out[F]:.3f}" for F in FL) + " ms", flush=True)
