"""Time from triangles in host memory to a scene ready to render, per builder."""
import sys
import time

import numpy as np

import ray_tracer_amd as rt

for name in sys.argv[1:] or ["bunny", "erato", "dragon", "aurelius"]:
    hs = rt.HostScene.standin(name, seed=1)
    a = hs.arrays()
    kw = dict(camera_position=a["camera_position"], camera_rotation_deg=a["camera_rotation_deg"],
              directional_lights=[(0.5, (0.0, 1.0, -1.0))], ambient=0.2)
    args = (a["vertices"], a["colors"], a["specularity"], a["reflectivity"])
    rt.Scene.from_arrays(*[x[:2000] for x in args], **kw)          # warm the runtime
    out = {}
    for label, builder in (("device", rt.BUILDER_DEVICE), ("host", rt.BUILDER_HOST)):
        best = None
        for _ in range(3 if label == "device" else 1):
            t0 = time.time()
            sc = rt.Scene.from_arrays(*args, builder=builder, **kw)
            wall = time.time() - t0
            if best is None or wall < best[0]:
                best = (wall, sc.build_seconds)
            f = sc.render(64, 36, traversal=rt.TRAVERSAL_PRUNED)
            out[label] = (best, f.rgb.copy(), f.hit.copy())
            sc.close()
    same = np.array_equal(out["device"][1], out["host"][1]) and np.array_equal(out["device"][2], out["host"][2])
    d, h = out["device"][0], out["host"][0]
    print(f"{name}: triangles={a['vertices'].shape[0]} create_ms device={d[0]*1e3:.1f} (build kernels {d[1]*1e3:.2f}) "
          f"host={h[0]*1e3:.1f} (build {h[1]*1e3:.1f}) speedup={h[0]/d[0]:.1f}x same_frame={same}", flush=True)
