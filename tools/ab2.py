#!/usr/bin/env python3
"""A/B of library builds (one librtx per process, so each build runs in child processes, interleaved, several rounds):
a 4K frame alone (each waited for) and four frames in flight.  usage: tools/ab2.py <scene> <family> <rounds> lib1.so lib2.so ..."""
import json, os, subprocess, sys, statistics

CHILD = r'''
import os, sys, json, time, zlib
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt
scene, family = sys.argv[1], sys.argv[2]
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot", no_bvh=True) if scene == "teapot" else rt.HostScene.standin(scene, seed=0, no_bvh=True, family=family)
dev = rt.Scene(host)
W, H = 3840, 2160
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
res = {}
for F in (1, 4):
    streams = [torch.cuda.Stream() for _ in range(F)]
    bufs = [torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda") for _ in range(F)]
    def run(n):
        for i in range(n):
            dev.render_device(p, rgb_u8=bufs[i % F].data_ptr(), stream=streams[i % F].cuda_stream)
            if F == 1: torch.cuda.synchronize()
        torch.cuda.synchronize()
    run(2 * F + 2)
    t0 = time.perf_counter(); run(40); res[F] = (time.perf_counter() - t0) / 40 * 1e3
print(json.dumps({"alone": res[1], "x4": res[4], "crc": zlib.crc32(bufs[0].cpu().numpy().tobytes())}))
'''

def main():
    scene, family, rounds = sys.argv[1:4]
    libs = sys.argv[4:]
    res = {l: {"alone": [], "x4": []} for l in libs}
    crc = {}
    for r in range(int(rounds)):
        for l in libs:
            env = dict(os.environ, RTX_LIB=os.path.abspath(l))
            out = subprocess.run([sys.executable, "-c", CHILD, scene, family], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(l, "FAILED", out.stderr[-500:], flush=True); continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[l]["alone"].append(d["alone"]); res[l]["x4"].append(d["x4"]); crc[l] = d["crc"]
    for l in libs:
        if res[l]["alone"]:
            print(f"{scene} {family} {os.path.basename(l):28s} alone median {statistics.median(res[l]['alone']):7.3f} min {min(res[l]['alone']):7.3f}   x4 median {statistics.median(res[l]['x4']):7.3f} min {min(res[l]['x4']):7.3f} ms  crc {crc[l]:08x}", flush=True)

main()
