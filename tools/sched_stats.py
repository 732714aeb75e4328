#!/usr/bin/env python3
"""Prints the wave scheduler's round counts and lane utilisation for one frame.  usage: tools/sched_stats.py [scene] [mode] [W] [H]"""
import os, sys
sys.path.insert(0, os.getcwd())
import ray_tracer_amd as rt
scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
mode = rt.TRAVERSAL_PRUNED if (len(sys.argv) < 3 or sys.argv[2] == "pruned") else rt.TRAVERSAL_REFERENCE
W = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
H = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot") if scene == "teapot" else rt.HostScene.standin(scene)
dev = rt.Scene(host)
f = dev.render(W, H, traversal=mode, want_stats=True, want_hit=False)
st = f.stats
d = st.as_dict()
print(d)
print(st.scheduling())
rays = d["rays"]
tot_rounds = sum(st.rounds)
print(f"rays {rays}  nodes/ray {d['node_visits']/rays:.1f}  tris/ray {d['triangle_tests']/rays:.2f}  wave-rounds {tot_rounds}  wave-rounds per 64 rays {tot_rounds*64/rays:.1f}  lane-steps/ray {sum(st.lane_steps)/rays:.1f}")
import torch
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
p = rt.make_params(W, H, traversal=mode)
ts = []
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); dev.render_device(p, rgb_u8=buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = min(ts[1:])
print(f"kernel {ms:.3f} ms  {rays/ms/1e3:.0f} Mrays/s  ns per wave-round (1024 SIMDs busy) {ms*1e6*1024/tot_rounds:.0f}  scene MB {dev.device_bytes/1e6:.0f}")
