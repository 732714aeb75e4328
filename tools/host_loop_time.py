#!/usr/bin/env python3
"""Time per frame DELIVERED TO HOST MEMORY through the C ABI, dragon stand-in at 3840x2160: rtx_render() call by call
against the rtx_render_begin / rtx_render_end loop (two slots).  usage: tools/host_loop_time.py [scene] [frames]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import ray_tracer_amd as rt

scene_name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W, H = 3840, 2160
scene = rt.Scene(rt.HostScene.standin(scene_name, no_bvh=True))
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
rgb = np.zeros((H, W, 3), dtype=np.uint8)
lib = rt.lib()
import ctypes as C
for _ in range(3):
    lib.rtx_render(scene._h, C.byref(p), rgb.ctypes.data, None, None)
t0 = time.perf_counter()
for _ in range(frames):
    lib.rtx_render(scene._h, C.byref(p), rgb.ctypes.data, None, None)
t1 = time.perf_counter()
print(f"rtx_render call by call:          {(t1 - t0) * 1e3 / frames:7.3f} ms per frame")
for slots in (2, 3, 4):
    t0 = time.perf_counter()
    for k in range(slots - 1):
        scene.render_begin(p, k)
    for n in range(frames):
        if n + slots - 1 < frames:
            scene.render_begin(p, (n + slots - 1) % slots)
        scene.render_end(n % slots, rgb)
    t1 = time.perf_counter()
    print(f"begin/end loop, {slots} slots in flight: {(t1 - t0) * 1e3 / frames:7.3f} ms per frame")
