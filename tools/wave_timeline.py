#!/usr/bin/env python3
"""Renders one frame with RTX_WAVE_LOG set (every wave records its start/end tick) and prints how
many waves were resident over time -- ramp-up, steady state and tail of the persistent launch.
usage: tools/wave_timeline.py [scene] [pruned|reference]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ray_tracer_amd as rt
scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
mode = rt.TRAVERSAL_PRUNED if (len(sys.argv) < 3 or sys.argv[2] == "pruned") else rt.TRAVERSAL_REFERENCE
W, H = 3840, 2160
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot") if scene == "teapot" else rt.HostScene.standin(scene)
dev = rt.Scene(host)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
p = rt.make_params(W, H, traversal=mode)
dev.render_device(p, rgb_u8=buf.data_ptr()); torch.cuda.synchronize()
os.environ["RTX_WAVE_LOG"] = "/tmp/wave_log.bin"
dev.render_device(p, rgb_u8=buf.data_ptr()); torch.cuda.synchronize()
del os.environ["RTX_WAVE_LOG"]
a = np.fromfile("/tmp/wave_log.bin", dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min(); s = (a[:, 0] - t0) / 100.0; e = (a[:, 1] - t0) / 100.0   # microseconds (100 MHz ticks)
dur = e - s
print(f"waves {len(a)}  span {e.max():.0f} us  sum of wave lifetimes {dur.sum()/1e3:.1f} ms  -> mean resident waves {dur.sum()/e.max():.0f}")
print("lifetime percentiles us:", {q: round(float(np.percentile(dur, q)), 1) for q in (1, 50, 99, 100)})
bins = np.linspace(0, e.max(), 21)
for i in range(20):
    lo, hi = bins[i], bins[i + 1]
    resident = (np.minimum(e, hi) - np.maximum(s, lo)).clip(min=0).sum() / (hi - lo)
    print(f"  {lo:8.0f}-{hi:8.0f} us  resident {resident:7.0f}")

# the end of the launch: when did the waves that leave last draw their last block, and where in the frame was it?
last = (a[:, 2] - t0) / 100.0
blk = a[:, 3]
bx, by, bq = (blk & 0xFFFF).astype(int), ((blk >> 16) & 0xFFFF).astype(int), (blk >> 32).astype(int)
print(f"last block drawn: percentiles us {({q: round(float(np.percentile(last, q)), 1) for q in (1, 50, 99, 100)})}")
print(f"time from a wave's last block to its exit: percentiles us {({q: round(float(np.percentile(e - last, q)), 1) for q in (1, 50, 90, 99, 100)})}")
late = np.argsort(e)[-200:]
print("the 200 waves that leave last: exit %.0f..%.0f us, last block drawn %.0f..%.0f us (median %.0f), after it %.0f..%.0f us (median %.0f)" % (
    e[late].min(), e[late].max(), last[late].min(), last[late].max(), np.median(last[late]), (e - last)[late].min(), (e - last)[late].max(), np.median((e - last)[late])))
print("  their last blocks: x %d..%d (median %d), y %d..%d (median %d) of %dx%d; queues %s" % (bx[late].min(), bx[late].max(), np.median(bx[late]), by[late].min(), by[late].max(), np.median(by[late]), W, H, np.bincount(bq[late], minlength=8).tolist()))
for lo, hi in ((0, 1000), (1000, 2000), (2000, 2500), (2500, 3000), (3000, 9999)):
    sel = (last >= lo) & (last < hi)
    if sel.any():
        print(f"  waves whose last block was drawn in {lo}-{hi} us: {int(sel.sum())}, they run {np.median((e - last)[sel]):.0f} us more (median), max {(e - last)[sel].max():.0f}")
