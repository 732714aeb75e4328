#!/usr/bin/env python3
"""Histogram of node-pair steps per query of one 4K frame, by query kind (needs a library built with
-DRTX_EXPERIMENT_STEPHIST, named by RTX_LIB: make -C ray_tracer_amd/csrc ../librtx_stephist.so).  usage: tools/step_histogram.py <scene> <family>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import ray_tracer_amd as rt
scene, family = sys.argv[1], sys.argv[2]
dev = rt.Scene(rt.HostScene.standin(scene, seed=0, no_bvh=True, family=family))
W, H = 3840, 2160
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
cnt = torch.zeros(64, dtype=torch.int64, device="cuda")
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
dev.render_device(p, rgb_u8=buf.data_ptr(), stats=cnt.data_ptr()); torch.cuda.synchronize()
v = cnt.cpu().numpy()
names = ["<8", "8-15", "16-31", "32-63", "64-127", "128-255", "256-511", "512-1023", "1024+"]
for kind, label in ((0, "closest-hit"), (1, "shadow")):
    q, s = v[16 + 12 * kind:16 + 12 * kind + 9], v[40 + 12 * kind:40 + 12 * kind + 9]
    print(f"{scene} {family} {label} queries {q.sum()}  pair steps {s.sum()}  mean {s.sum()/max(1,q.sum()):.1f}")
    for b in range(9):
        if q[b]:
            print(f"   {names[b]:>9s} steps: {100*q[b]/q.sum():5.1f} % of the queries, {100*s[b]/s.sum():5.1f} % of the pair steps")
