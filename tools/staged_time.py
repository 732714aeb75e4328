#!/usr/bin/env python3
"""The fused launch (rtx_render_device) against the staged pipeline (rtx_render_device_staged) in one process: a 4K frame
alone (each waited for), four frames in flight, and rank 0's share of the frame for eight ranks (packed tiles) alone and
with four / eight launches in flight.  usage: tools/staged_time.py <scene> <family> [rounds]"""
import ctypes as C, os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import ray_tracer_amd as rt

scene, family = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot", no_bvh=True) if scene == "teapot" else rt.HostScene.standin(scene, seed=0, no_bvh=True, family=family)
dev = rt.Scene(host)
W, H = 3840, 2160


def measure(call, p, F, n):
    px = int(rt.lib().rtx_output_pixels(C.byref(p)))
    streams = [torch.cuda.Stream() for _ in range(F)]
    bufs = [torch.zeros(px * 3, dtype=torch.uint8, device="cuda") for _ in range(F)]
    def run(k):
        for i in range(k):
            call(p, rgb_u8=bufs[i % F].data_ptr(), stream=streams[i % F].cuda_stream)
            if F == 1:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    run(2 * F + 2)
    t0 = time.perf_counter(); run(n); ms = (time.perf_counter() - t0) / n * 1e3
    return ms, zlib.crc32(bufs[0].cpu().numpy().tobytes())


for r in range(rounds):
    for name, call in (("fused ", dev.render_device), ("staged", dev.render_device_staged)):
        whole = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
        share = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, tile_rank=0, tile_count=8, layout=rt.LAYOUT_PACKED_TILES)
        a, crc = measure(call, whole, 1, 20)
        b, _ = measure(call, whole, 4, 40)
        c, crc8 = measure(call, share, 1, 40)
        d, _ = measure(call, share, 4, 80)
        e, _ = measure(call, share, 8, 80)
        print(f"{scene} {family} {name}  world 1 x1: {a:7.3f} ms  x4: {b:7.3f} ms   world 8 x1: {c:6.3f} ms  x4: {d:6.3f} ms  x8: {e:6.3f} ms   crc {crc:08x} {crc8:08x}", flush=True)
