"""Times rtx_render() (host buffers in and out: what a drop-in caller sees) against the kernel alone."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import ctypes as C
import numpy as np
import ray_tracer_amd as rt

scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
W, H = 3840, 2160
dev = rt.Scene(rt.HostScene.standin(scene, seed=0, no_bvh=True))
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
rgb = np.zeros((H, W, 3), dtype=np.uint8)
st = rt.Stats()
for want_stats in (False,):
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        rc = rt.lib().rtx_render(dev._h, C.byref(p), rgb.ctypes.data, None, None)
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    print(f"rtx_render {scene} {W}x{H}: best {min(ts[1:])*1e3:.2f} ms, median {sorted(ts[1:])[2]*1e3:.2f} ms")
