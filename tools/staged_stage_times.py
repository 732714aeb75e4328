#!/usr/bin/env python3
"""Per-launch durations of one staged frame (events between the launches are not possible from outside, so each stage is
timed by running the frame under RTX_STAGED_ONLY=<n>, which stops after launch n, and differencing).
usage: tools/staged_stage_times.py <scene> <family> [tile_count]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import ray_tracer_amd as rt
scene, family = sys.argv[1], sys.argv[2]
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1
host = rt.HostScene.standin(scene, seed=0, no_bvh=True, family=family)
dev = rt.Scene(host)
W, H = 3840, 2160
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, tile_rank=0, tile_count=count, layout=rt.LAYOUT_PACKED_TILES if count > 1 else rt.LAYOUT_ROW_MAJOR)
px = int(rt.lib().rtx_output_pixels(C.byref(p)))
buf = torch.zeros(px * 3, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for run in os.environ.get("RUNS", "1,8,16,32").split(","):
    os.environ["RTX_STAGED_REFILL"] = run
    prev = 0.0
    out = []
    for last in range(6):
        os.environ["RTX_STAGED_ONLY"] = str(last)
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.render_device_staged(p, rgb_u8=buf.data_ptr(), stream=st)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        t = min(ts[1:])
        out.append(t - prev)
        prev = t
    print(f"{scene} {family} 1/{count} of the tiles, refill at {run} lanes: launches 0..5 take " + " ".join(f"{x:.3f}" for x in out) + f" ms, sum {prev:.3f}", flush=True)
