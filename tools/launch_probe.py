#!/usr/bin/env python3
"""Why do several launches in flight render faster than one launch with the chip to itself?  Times, per 4K frame's worth
of rays: one frame at a time; 2/3/4/8 frames in flight; one launch over four frames' worth of pixels (7680x4320) and
over four samples per pixel; and four launches in flight that each keep only a quarter of the chip (RTX_BLOCKS_PER_CU=1).
usage: tools/launch_probe.py [scene] [family]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt

scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
family = sys.argv[2] if len(sys.argv) > 2 else "blob"
dev = rt.Scene(rt.HostScene.standin(scene, no_bvh=True, family=family))
W, H = 3840, 2160


def per_frame(p, F, n, px):
    streams = [torch.cuda.Stream() for _ in range(F)]
    bufs = [torch.zeros(px * 3, dtype=torch.uint8, device="cuda") for _ in range(F)]
    def run(k):
        for i in range(k):
            dev.render_device(p, rgb_u8=bufs[i % F].data_ptr(), stream=streams[i % F].cuda_stream)
        torch.cuda.synchronize()
    run(2 * F)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run(n); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


p4k = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
for F in (1, 2, 3, 4, 6, 8):
    print(f"4K, {F} in flight: {per_frame(p4k, F, 8 * F, W * H):7.3f} ms per frame", flush=True)
p8k = rt.make_params(2 * W, 2 * H, traversal=rt.TRAVERSAL_PRUNED)
print(f"8K (4 x the pixels), 1 in flight: {per_frame(p8k, 1, 6, 4 * W * H) / 4:7.3f} ms per 4K frame's worth", flush=True)
print(f"8K (4 x the pixels), 2 in flight: {per_frame(p8k, 2, 8, 4 * W * H) / 4:7.3f} ms per 4K frame's worth", flush=True)
ps4 = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, spp=4)
print(f"4K 4 spp, 1 in flight: {per_frame(ps4, 1, 6, W * H) / 4:7.3f} ms per 4K frame's worth", flush=True)
for bpc in (1, 2):
    os.environ["RTX_BLOCKS_PER_CU"] = str(bpc)
    for F in (1, 2, 4, 8):
        print(f"4K, launches of {bpc} block(s) per CU, {F} in flight: {per_frame(p4k, F, 8 * F, W * H):7.3f} ms per frame", flush=True)
del os.environ["RTX_BLOCKS_PER_CU"]
for order in ("0", "1"):
    os.environ["RTX_CENTRE_FIRST"] = order
    print(f"4K centre_first={order}: 1 in flight {per_frame(p4k, 1, 8, W * H):7.3f}, 4 in flight {per_frame(p4k, 4, 32, W * H):7.3f} ms per frame", flush=True)
