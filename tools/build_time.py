"""Times the host and device BVH builders on the stand-in meshes (one line per scene)."""
import sys
import time

import numpy as np

import ray_tracer_amd as rt

for name in sys.argv[1:] or ["teapot_like", "bunny", "erato", "dragon", "aurelius"]:
    if name == "teapot_like":
        hs = rt.HostScene.standin("teapot", 6330, seed=1)
    else:
        hs = rt.HostScene.standin(name, seed=1)
    a = hs.arrays()
    v = a["vertices"]
    rt.build_bvh(v[:1000], rt.BUILDER_DEVICE)           # warm the runtime
    t0 = time.time()
    host = rt.build_bvh(v, rt.BUILDER_HOST)
    t_host = time.time() - t0
    best = None
    for _ in range(3):
        t0 = time.time()
        dev = rt.build_bvh(v, rt.BUILDER_DEVICE)
        wall = time.time() - t0
        if best is None or wall < best[0]:
            best = (wall, dev["seconds"])
    same = dev["nodes"].tobytes() == host["nodes"].tobytes() and np.array_equal(dev["indices"], host["indices"])
    print(f"{name}: triangles={v.shape[0]} nodes={host['nodes'].shape[0]} depth={host['depth']} "
          f"host_build_s={host['seconds']:.3f} (call {t_host:.3f}) device_kernels_ms={best[1]*1e3:.2f} "
          f"device_call_ms={best[0]*1e3:.1f} identical={same}", flush=True)
