"""Scheduler knobs (environment variables the library reads per launch) timed with four frames in flight, the bench's mode:
there the chip is never empty, so fuller rounds can matter more than short waits.  usage: tools/explore_inflight.py [group...]"""
import os, sys, time, itertools
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt

W, H, F = 3840, 2160, 4
dev = rt.Scene(rt.HostScene.standin("dragon", seed=0, no_bvh=True))
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
streams = [torch.cuda.Stream() for _ in range(F)]
bufs = [torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda") for _ in range(F)]
DEFAULTS = dict(RTX_W_INNER=3, RTX_W_LEAF=4, RTX_W_SHADE=4, RTX_TAIL_LANES=16, RTX_ALSO_LANES=24, RTX_ALSO_LEAF=16, RTX_ALSO_SHADE=24)


def run(n):
    for k in range(n):
        dev.render_device(p, rgb_u8=bufs[k % F].data_ptr(), stream=streams[k % F].cuda_stream)
    torch.cuda.synchronize()


def timed(label, **env):
    for k, v in {**DEFAULTS, **env}.items():
        os.environ[k] = str(v)
    run(8)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run(40); best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    print(f"{label:60s} {best:7.3f} ms", flush=True)


groups = sys.argv[1:] or ["base", "also", "weights", "tail"]
if "base" in groups:
    for _ in range(3): timed("defaults")
if "also" in groups:
    for a in (16, 24, 32, 40, 65):
        for l in (12, 16, 24, 32, 65):
            timed(f"also_lanes={a} also_leaf={l} also_shade={a}", RTX_ALSO_LANES=a, RTX_ALSO_LEAF=l, RTX_ALSO_SHADE=a)
if "weights" in groups:
    for wi, wl, ws in ((1, 1, 1), (1, 2, 2), (2, 3, 3), (1, 3, 3), (1, 2, 3), (1, 3, 2), (2, 3, 4), (1, 4, 4), (3, 4, 4)):
        timed(f"weights inner/leaf/shade={wi}/{wl}/{ws}", RTX_W_INNER=wi, RTX_W_LEAF=wl, RTX_W_SHADE=ws)
if "weights2" in groups:
    for wi, wl, ws in ((1, 1, 1), (1, 2, 2), (3, 4, 4), (4, 5, 5), (5, 6, 6), (2, 2, 3), (3, 3, 4), (3, 4, 3), (4, 4, 5), (4, 5, 4), (1, 1, 2), (2, 1, 2), (3, 2, 3), (1, 1, 1), (1, 2, 2)):
        timed(f"weights inner/leaf/shade={wi}/{wl}/{ws}", RTX_W_INNER=wi, RTX_W_LEAF=wl, RTX_W_SHADE=ws)
if "tail" in groups:
    for t in (0, 8, 16, 32, 64):
        timed(f"tail_lanes={t}", RTX_TAIL_LANES=t)
if "queues" in groups:
    timed("single queue", RTX_XCD_QUEUES=0)
    for m in (1, 2, 4, 8, 16, 32):
        timed(f"8 XCD queues, macro={m} tiles", RTX_XCD_QUEUES=1, RTX_MACRO=m)
    os.environ.update(RTX_XCD_QUEUES="1", RTX_MACRO="8")
if "tile" in groups:
    for t in (8, 16, 32, 64):
        p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, tile=(t, t))
        timed(f"tile={t}")
    p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
