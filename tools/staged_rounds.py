#!/usr/bin/env python3
"""Wave rounds of each launch of a staged frame (counting kernel, RTX_STAGED_ONLY=<n> differences) beside the fused
launch's: where the frame's rounds are.  usage: tools/staged_rounds.py <scene> <family>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import ray_tracer_amd as rt
scene, family = sys.argv[1], sys.argv[2]
W, H = 3840, 2160
dev = rt.Scene(rt.HostScene.standin(scene, seed=0, no_bvh=True, family=family))
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
def counters(call):
    cnt = torch.zeros(15, dtype=torch.int64, device="cuda")
    call(p, rgb_u8=buf.data_ptr(), stats=cnt.data_ptr()); torch.cuda.synchronize()
    return cnt.cpu().numpy().astype(np.int64)
f = counters(dev.render_device)
print(f"{scene} {family} fused : rays {f[0]+f[1]+f[2]:>9d}  rounds pair/triangle/shading {f[9]:>9d} {f[10]:>8d} {f[11]:>8d}  lane use {f[12]/64/f[9]:.3f} {f[13]/64/f[10]:.3f} {f[14]/64/f[11]:.3f}")
prev = np.zeros(15, dtype=np.int64)
for last in range(5):
    os.environ["RTX_STAGED_ONLY"] = str(last)
    c = counters(dev.render_device_staged)
    d = c - prev; prev = c
    print(f"  staged launch {last}: rays {d[0]+d[1]+d[2]:>9d}  rounds pair/triangle/shading {d[9]:>9d} {d[10]:>8d} {d[11]:>8d}  lane use {d[12]/64/max(1,d[9]):.3f} {d[13]/64/max(1,d[10]):.3f} {d[14]/64/max(1,d[11]):.3f}  pair steps per ray {d[12]/max(1,d[0]+d[1]+d[2]):.1f}")
del os.environ["RTX_STAGED_ONLY"]
c = counters(dev.render_device_staged)
print(f"  staged frame : rays {c[0]+c[1]+c[2]:>9d}  rounds pair/triangle/shading {c[9]:>9d} {c[10]:>8d} {c[11]:>8d}  lane use {c[12]/64/c[9]:.3f} {c[13]/64/c[10]:.3f} {c[14]/64/c[11]:.3f}")
