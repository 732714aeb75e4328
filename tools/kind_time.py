#!/usr/bin/env python3
"""Where a wave's time goes, by kind of step: renders one 4K frame with a library built with -DRTX_EXPERIMENT_KINDTIME
(RTX_LIB=<that library>), whose waves add up the shader-clock cycles they spend inside node-pair steps, triangle steps
and shading / refill steps and in all rounds together, and leave the sums in the wave log.  The rounds of each kind come
from a second launch with the work counters on.
build: make -C ray_tracer_amd/csrc ../librtx_kindtime.so
usage: RTX_LIB=ray_tracer_amd/librtx_kindtime.so tools/kind_time.py [scene] [family]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import ray_tracer_amd as rt

scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
family = sys.argv[2] if len(sys.argv) > 2 else "knot"
W, H = 3840, 2160
host = rt.HostScene.standin(scene, seed=0, no_bvh=True, family=family)
dev = rt.Scene(host)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
dev.render_device(p, rgb_u8=buf.data_ptr()); torch.cuda.synchronize()
os.environ["RTX_WAVE_LOG"] = "/tmp/kind_log.bin"
dev.render_device(p, rgb_u8=buf.data_ptr()); torch.cuda.synchronize()
del os.environ["RTX_WAVE_LOG"]
a = np.fromfile("/tmp/kind_log.bin", dtype=np.uint64).reshape(-1, 4)
sections = a[-2:].reshape(-1).astype(np.float64)      # cycles inside the shading step, summed over the waves
a = a[:-2]
a = a[a[:, 1] > 0]
life_us = (a[:, 1] - a[:, 0]) / 100.0
m32 = np.uint64(0xFFFFFFFF)
pair, tri = (a[:, 2] & m32).astype(np.float64), (a[:, 2] >> np.uint64(32)).astype(np.float64)
shade, rounds = (a[:, 3] & m32).astype(np.float64), (a[:, 3] >> np.uint64(32)).astype(np.float64)
total = rounds.sum()
cnt = torch.zeros(15, dtype=torch.int64, device="cuda")
dev.render_device(p, rgb_u8=buf.data_ptr(), stats=cnt.data_ptr()); torch.cuda.synchronize()
c = cnt.cpu().numpy()
n_round = {"node-pair": int(c[9]), "triangle": int(c[10]), "shading / refill": int(c[11])}
print(f"{scene} {family}: {len(a)} waves, mean lifetime {life_us.mean():.0f} us, cycles in rounds per wave {rounds.mean() / 1e6:.2f} M "
      f"({rounds.mean() / life_us.mean():.0f} per us of lifetime)")
for name, x in (("node-pair", pair), ("triangle", tri), ("shading / refill", shade)):
    print(f"  {name + ' steps':24s} {100 * x.sum() / total:5.1f} % of the waves' time   {n_round[name]:9d} rounds   {x.sum() / n_round[name]:7.0f} cycles per round")
other = total - pair.sum() - tri.sum() - shade.sum()
print(f"  {'choosing the step':24s} {100 * other / total:5.1f} % of the waves' time   {other / sum(n_round.values()):7.0f} cycles per step run")
names = ["the trip to memory (normal, material, scratch)", "closest-hit bookkeeping / lighting term", "lights loop, shadow query starts",
         "local colour, reflection query start", "blend", None, None, "pixel write and what is left"]
whole = sections[6]
print(f"  inside the shading step ({100 * whole / shade.sum():.0f} % of the shading / refill time; the rest is handing out pixels and starting camera rays):")
for k, name in enumerate(names):
    if name: print(f"    {name:52s} {100 * sections[k] / whole:5.1f} %   {sections[k] / n_round['shading / refill']:7.0f} cycles per round")
rest = shade.sum() - whole
print(f"  outside it: drawing 8x8 blocks from the queues {sections[5] / n_round['shading / refill']:7.0f} cycles per round, "
      f"pixels to lanes, camera rays, query starts {(rest - sections[5]) / n_round['shading / refill']:7.0f}")
