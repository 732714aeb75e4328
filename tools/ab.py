#!/usr/bin/env python3
"""A/B timing of library variants in ONE process is not possible (one librtx per process), so this
runs each variant in a child process on the same GPU, interleaved, several rounds, and prints the
median / min kernel time.  usage: tools/ab.py <scene> <W> <H> <mode> <rounds> lib1.so lib2.so ..."""
import json, os, subprocess, sys, statistics

CHILD = r'''
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt
scene, W, H, mode, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot") if scene == "teapot" else rt.HostScene.standin(scene)
dev = rt.Scene(host)
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED if mode == "pruned" else rt.TRAVERSAL_REFERENCE)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): dev.render_device(p, rgb_u8=buf.data_ptr(), stream=st)
torch.cuda.synchronize()
ts = []
for _ in range(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); dev.render_device(p, rgb_u8=buf.data_ptr(), stream=st); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
import zlib
print(json.dumps({"ms": ts, "crc": zlib.crc32(buf.cpu().numpy().tobytes())}))
'''

def main():
    scene, W, H, mode, rounds = sys.argv[1:6]
    libs = sys.argv[6:]
    res = {l: [] for l in libs}
    crc = {}
    for r in range(int(rounds)):
        for l in libs:
            env = dict(os.environ, RTX_LIB=os.path.abspath(l))
            out = subprocess.run([sys.executable, "-c", CHILD, scene, W, H, mode, "5"], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(l, "FAILED", out.stderr[-500:]); continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[l] += d["ms"]; crc[l] = d["crc"]
    for l in libs:
        if res[l]:
            print(f"{os.path.basename(l):28s} median {statistics.median(res[l]):8.3f} ms  min {min(res[l]):8.3f} ms  crc {crc[l]:08x}")

main()
