#!/usr/bin/env python3
"""Frame time (one frame at a time) for values of one environment knob the library reads per launch.
usage: tools/knob_probe.py KNOB v1,v2,... [scene] [family] [W] [H]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt

knob, values = sys.argv[1], sys.argv[2].split(",")
scene = sys.argv[3] if len(sys.argv) > 3 else "dragon"
family = sys.argv[4] if len(sys.argv) > 4 else "blob"
W = int(sys.argv[5]) if len(sys.argv) > 5 else 3840
H = int(sys.argv[6]) if len(sys.argv) > 6 else 2160
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot", no_bvh=True) if scene == "teapot" else rt.HostScene.standin(scene, no_bvh=True, family=family)
dev = rt.Scene(host)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for v in values:
        os.environ[knob] = v
        for mode, name in ((rt.TRAVERSAL_PRUNED, "pruned"), (rt.TRAVERSAL_REFERENCE, "reference")):
            p = rt.make_params(W, H, traversal=mode)
            ts = []
            for _ in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); dev.render_device(p, rgb_u8=buf.data_ptr(), stream=st); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            print(f"{scene}/{family} {W}x{H} {name:9s} {knob}={v}: {min(ts[1:]):8.3f} ms", flush=True)
