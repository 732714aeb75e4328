#!/usr/bin/env python3
"""Frame time against the size of the top-of-tree table kept in LDS (RTX_TOP_PAIRS caps it at scene creation).
usage: tools/top_pairs_probe.py [scene] [W] [H]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt

scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot") if scene == "teapot" else rt.HostScene.standin(scene, no_bvh=True)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for cap in (0, 8, 16, 32, 64, 128, 512):
    os.environ["RTX_TOP_PAIRS"] = str(cap)
    dev = rt.Scene(host)
    for mode, name in ((rt.TRAVERSAL_PRUNED, "pruned"), (rt.TRAVERSAL_REFERENCE, "reference")):
        p = rt.make_params(W, H, traversal=mode)
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); dev.render_device(p, rgb_u8=buf.data_ptr(), stream=st); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print(f"{scene} {name:9s} top pairs <= {cap:4d}: {min(ts[1:]):8.3f} ms", flush=True)
    dev.close()
