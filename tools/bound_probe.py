#!/usr/bin/env python3
"""Is the render kernel bound by latency or by the throughput of the memory path?  Times a long frame
(several samples per pixel, so the end-of-frame tail is a small share) with 1..4 resident blocks per CU:
time that scales with 1/blocks says latency, time that stops falling says a saturated unit.
usage: tools/bound_probe.py [scene] [spp]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt

scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = 3840, 2160
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot") if scene == "teapot" else rt.HostScene.standin(scene)
dev = rt.Scene(host)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mode, name in ((rt.TRAVERSAL_PRUNED, "pruned"), (rt.TRAVERSAL_REFERENCE, "reference")):
    p = rt.make_params(W, H, traversal=mode, spp=spp)
    for bpc in (1, 2, 3, 4):
        os.environ["RTX_BLOCKS_PER_CU"] = str(bpc)
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); dev.render_device(p, rgb_u8=buf.data_ptr(), stream=st); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print(f"{scene} {name} spp={spp} blocks/CU={bpc}: {min(ts[1:]):8.3f} ms", flush=True)
