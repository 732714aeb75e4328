#!/usr/bin/env python3
"""Renders one frame with RTX_WAVE_LOG set and prints how many waves were resident over time.
usage: tools/wave_timeline.py [scene] [mode] [chunk]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ray_tracer_amd as rt
scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
mode = rt.TRAVERSAL_PRUNED if (len(sys.argv) < 3 or sys.argv[2] == "pruned") else rt.TRAVERSAL_REFERENCE
os.environ["RTX_CHUNK"] = sys.argv[3] if len(sys.argv) > 3 else "64"
W, H = 3840, 2160
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot") if scene == "teapot" else rt.HostScene.standin(scene)
dev = rt.Scene(host)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
p = rt.make_params(W, H, traversal=mode)
dev.render_device(p, rgb_u8=buf.data_ptr()); torch.cuda.synchronize()
os.environ["RTX_WAVE_LOG"] = "/tmp/wave_log.bin"
dev.render_device(p, rgb_u8=buf.data_ptr()); torch.cuda.synchronize()
del os.environ["RTX_WAVE_LOG"]
a = np.fromfile("/tmp/wave_log.bin", dtype=np.uint64).reshape(-1, 2)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min(); s = (a[:, 0] - t0) / 100.0; e = (a[:, 1] - t0) / 100.0   # microseconds
dur = e - s
print(f"waves {len(a)}  span {e.max():.0f} us  sum of wave durations {dur.sum()/1e3:.1f} ms  -> mean resident waves {dur.sum()/e.max():.0f} (chip holds 2048 at 2/SIMD)")
print("duration percentiles us:", {q: round(float(np.percentile(dur, q)), 1) for q in (50, 90, 99, 99.9, 100)})
bins = np.linspace(0, e.max(), 21)
for i in range(20):
    lo, hi = bins[i], bins[i + 1]
    resident = (np.minimum(e, hi) - np.maximum(s, lo)).clip(min=0).sum() / (hi - lo)
    started = ((s >= lo) & (s < hi)).sum()
    print(f"  {lo:8.0f}-{hi:8.0f} us  resident {resident:7.0f}  started {started}")
# the slowest waves and where they are on screen (chunk = 64: wave -> tile, 8x8 block)
if os.environ["RTX_CHUNK"] == "64":
    order = np.argsort(-dur)[:12]
    idx = np.nonzero(np.fromfile("/tmp/wave_log.bin", dtype=np.uint64).reshape(-1, 2)[:, 1] > 0)[0]
    for j in order:
        gw = int(idx[j]); tile, blk = divmod(gw, 16); ty, tx = divmod(tile, 120); by, bx = divmod(blk, 4)
        print(f"  wave {gw}: {dur[j]:.0f} us, start {s[j]:.0f} us, pixels x {tx*32+bx*8}..+8, y {ty*32+by*8}..+8")
    # per-pixel cost map: run stats per small window around the slowest wave
    j = order[0]; gw = int(idx[j]); tile, blk = divmod(gw, 16); ty, tx = divmod(tile, 120); by, bx = divmod(blk, 4)
    x0, y0 = tx*32+bx*8, ty*32+by*8
    for (wx0, wx1, wy0, wy1, name) in ((x0, x0+8, y0, y0+8, "slowest 8x8"), (1900, 1908, 1000, 1008, "centre 8x8")):
        f = dev.render(W, H, traversal=mode, window=(wx0, wx1, wy0, wy1), want_stats=True, want_hit=False)
        print(name, {k: v for k, v in f.stats.as_dict().items() if k != "kernel_ms"})
