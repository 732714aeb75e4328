#!/usr/bin/env python3
"""Timing experiments on one GPU: tuning knobs (environment variables the library reads per launch).
usage: tools/explore.py [scene] [pruned|reference] [experiments...]   experiments: weights bpc queues tile"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt

scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
mode = rt.TRAVERSAL_PRUNED if (len(sys.argv) < 3 or sys.argv[2] == "pruned") else rt.TRAVERSAL_REFERENCE
W, H = 3840, 2160
host = rt.HostScene.load_obj("tests/golden/teapot.obj", "teapot") if scene == "teapot" else rt.HostScene.standin(scene)
dev = rt.Scene(host)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def timeit(p, n=3):
    dev.render_device(p, rgb_u8=buf.data_ptr(), stream=st); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); dev.render_device(p, rgb_u8=buf.data_ptr(), stream=st); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


def run(label, env, **kw):
    for k, v in env.items():
        os.environ[k] = str(v)
    p = rt.make_params(W, H, traversal=mode, **kw)
    print(f"{label:50s} {timeit(p):8.3f} ms", flush=True)


exps = sys.argv[3:] or ["weights", "bpc", "queues", "tile"]
if "weights" in exps:
    for wi, wl, ws in ((1, 1, 1), (1, 2, 2), (2, 3, 3), (1, 3, 3), (1, 2, 4), (1, 3, 2), (2, 3, 4), (3, 4, 4), (1, 4, 4), (1, 2, 3)):
        run(f"weights inner/leaf/shade={wi}/{wl}/{ws}", {"RTX_W_INNER": wi, "RTX_W_LEAF": wl, "RTX_W_SHADE": ws})
    os.environ.update(RTX_W_INNER="3", RTX_W_LEAF="4", RTX_W_SHADE="4")
if "bpc" in exps:
    for b in (2, 3, 4, 5, 6, 8):
        run(f"blocks per CU={b}", {"RTX_BLOCKS_PER_CU": b})
    del os.environ["RTX_BLOCKS_PER_CU"]
if "queues" in exps:
    run("single queue", {"RTX_XCD_QUEUES": 0})
    for m in (1, 2, 4, 8, 16, 32):
        run(f"8 XCD queues, macro={m} tiles", {"RTX_XCD_QUEUES": 1, "RTX_MACRO": m})
    os.environ.update(RTX_XCD_QUEUES="1", RTX_MACRO="8")
if "tail" in exps:
    for t in (0, 4, 8, 16, 24, 32, 48, 64):
        run(f"tail_lanes={t}", {"RTX_TAIL_LANES": t})
    os.environ["RTX_TAIL_LANES"] = "16"
if "tile" in exps:
    for t in (8, 16, 32, 64):
        run(f"tile={t}", {}, tile=(t, t))
if "also" in exps:
    for t in (65, 48, 40, 32, 24, 16, 8):
        run(f"also_lanes={t}", {"RTX_ALSO_LANES": t})
    os.environ["RTX_ALSO_LANES"] = "65"
if "also3" in exps:
    for ai, al, ash in ((24, 24, 24), (24, 16, 24), (24, 12, 24), (24, 8, 24), (24, 24, 16), (24, 24, 12), (24, 16, 16), (32, 12, 16),
                        (24, 12, 12), (32, 16, 24), (16, 12, 16), (24, 20, 32), (24, 16, 40)):
        run(f"also inner/leaf/shade={ai}/{al}/{ash}", {"RTX_ALSO_LANES": ai, "RTX_ALSO_LEAF": al, "RTX_ALSO_SHADE": ash})
    for k in ("RTX_ALSO_LANES", "RTX_ALSO_LEAF", "RTX_ALSO_SHADE"):
        os.environ.pop(k, None)
