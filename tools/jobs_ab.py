"""Lighting jobs on and off (RTX_LIGHTING_JOBS, read per launch) in one process on one GPU: a 4K frame alone, four in
flight, and one rank's share of the frame for 8 ranks with 1 / 4 / 8 in flight.  usage: jobs_ab.py [blob|knot] [scene]"""
import os, sys, time, zlib
sys.path.insert(0, os.getcwd())
import ctypes as C
import torch
import ray_tracer_amd as rt

family = sys.argv[1] if len(sys.argv) > 1 else "blob"
name = sys.argv[2] if len(sys.argv) > 2 else "dragon"
host = rt.HostScene.standin(name, seed=0, no_bvh=True, family=family)
dev = rt.Scene(host)
W, H = 3840, 2160

def timed(p, px, F, n):
    streams = [torch.cuda.Stream() for _ in range(F)]
    bufs = [torch.zeros(px * 3, dtype=torch.uint8, device="cuda") for _ in range(F)]
    def run(k):
        for i in range(k):
            dev.render_device(p, rgb_u8=bufs[i % F].data_ptr(), stream=streams[i % F].cuda_stream)
            if F == 1:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    run(2 * F + 2)
    t0 = time.perf_counter(); run(n); ms = (time.perf_counter() - t0) / n * 1e3
    return ms, zlib.crc32(bufs[0].cpu().numpy().tobytes())

for rep in range(2):
    for jobs in os.environ.get("JOBS", "0,1").split(","):
        os.environ["RTX_LIGHTING_JOBS"] = jobs
        line = [f"{family} {name} jobs={jobs}"]
        for world, flist in ((1, (1, 4)), (8, (1, 4, 8))):
            p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, tile=(32, 32), tile_rank=0, tile_count=world,
                               layout=rt.LAYOUT_PACKED_TILES if world > 1 else rt.LAYOUT_ROW_MAJOR)
            px = int(rt.lib().rtx_output_pixels(C.byref(p)))
            for F in flist:
                ms, crc = timed(p, px, F, 40)
                line.append(f"world {world} x{F}: {ms:.3f} ms ({crc:08x})")
        print("  ".join(line), flush=True)
