#!/usr/bin/env python3
"""One scene, a few staged and fused 4K frames one at a time (for a kernel trace), then the scheduler statistics of both.
usage: tools/staged_profile.py <scene> <family>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import ray_tracer_amd as rt
scene, family = sys.argv[1], sys.argv[2]
host = rt.HostScene.standin(scene, seed=0, no_bvh=True, family=family)
dev = rt.Scene(host)
W, H = 3840, 2160
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for call in (dev.render_device, dev.render_device_staged):
    for _ in range(4):
        call(p, rgb_u8=buf.data_ptr(), stream=st)
        torch.cuda.synchronize()
for name, call in (("fused", dev.render_device), ("staged", dev.render_device_staged)):
    cnt = torch.zeros(15, dtype=torch.int64, device="cuda")
    call(p, rgb_u8=buf.data_ptr(), stats=cnt.data_ptr(), stream=st)
    torch.cuda.synchronize()
    v = cnt.cpu().numpy()
    print(name, "rounds", v[9:12], "lane util", [round(v[12 + i] / (64.0 * v[9 + i]), 3) for i in range(3)], "lane steps", v[12:15])
