#!/usr/bin/env python3
"""Wave residency over time of ONE launch of a staged frame (RTX_STAGED_ONLY=<n> makes it the last launch, whose waves
leave their clock values in RTX_WAVE_LOG).  usage: tools/staged_timeline.py <scene> <family> <launch>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import ray_tracer_amd as rt
scene, family, launch = sys.argv[1], sys.argv[2], sys.argv[3]
W, H = 3840, 2160
dev = rt.Scene(rt.HostScene.standin(scene, seed=0, no_bvh=True, family=family))
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
os.environ["RTX_STAGED_ONLY"] = launch
dev.render_device_staged(p, rgb_u8=buf.data_ptr()); torch.cuda.synchronize()
os.environ["RTX_WAVE_LOG"] = "/tmp/wave_log.bin"
dev.render_device_staged(p, rgb_u8=buf.data_ptr()); torch.cuda.synchronize()
a = np.fromfile("/tmp/wave_log.bin", dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min(); s = (a[:, 0] - t0) / 100.0; e = (a[:, 1] - t0) / 100.0
dur = e - s
print(f"{scene} {family} launch {launch}: waves {len(a)}  span {e.max():.0f} us  mean resident waves {dur.sum()/e.max():.0f}")
print(" wave start percentiles us:", {q: round(float(np.percentile(s, q)), 1) for q in (1, 50, 99, 100)})
print(" wave lifetime percentiles us:", {q: round(float(np.percentile(dur, q)), 1) for q in (1, 50, 99, 100)})
last = (a[:, 2].astype(np.int64) - int(t0)) / 100.0
print(" last block drawn percentiles us:", {q: round(float(np.percentile(last, q)), 1) for q in (1, 50, 99, 100)})
print(" from last block to exit percentiles us:", {q: round(float(np.percentile(e - last, q)), 1) for q in (1, 50, 90, 99, 100)})
bins = np.linspace(0, e.max(), 11)
for i in range(10):
    lo, hi = bins[i], bins[i + 1]
    print(f"  {lo:8.0f}-{hi:8.0f} us  resident {(np.minimum(e, hi) - np.maximum(s, lo)).clip(min=0).sum() / (hi - lo):7.0f}")
