"""Identity of the device-built tree with the host builder's on one large random soup.  usage: tools/big_build.py [n]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import ray_tracer_amd as rt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(3)
c = rng.uniform(-100, 100, size=(n, 1, 3))
v = (c + rng.normal(scale=0.2, size=(n, 3, 3))).reshape(n, 9)
t0 = time.time(); d = rt.build_bvh(v, builder=rt.BUILDER_DEVICE); t_dev = time.time() - t0
t0 = time.time(); h = rt.build_bvh(v, builder=rt.BUILDER_HOST); t_host = time.time() - t0
same = d["nodes"].tobytes() == h["nodes"].tobytes() and np.array_equal(d["indices"], h["indices"])
print(f"n={n} nodes={len(h['nodes'])} depth={h['depth']} host {t_host:.2f} s (build {h['seconds']:.2f}) device call {t_dev:.3f} s (kernels {d['seconds']*1e3:.1f} ms) identical={same}", flush=True)
assert same
