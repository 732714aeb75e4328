"""Frames per launch against frames in flight, one GPU: ms per 4K frame of the dragon stand-ins for 1 / 2 / 4 / 8 frames
in ONE launch on one stream, and for single-frame launches on 1 / 4 streams.  usage: batch_time.py [blob|knot]"""
import os, sys, time, zlib
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt
family = sys.argv[1] if len(sys.argv) > 1 else "blob"
host = rt.HostScene.standin("dragon", seed=0, no_bvh=True, family=family)
dev = rt.Scene(host)
W, H = 3840, 2160
p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
buf = torch.zeros(8 * W * H * 3, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def crc(k):
    return "%08x" % zlib.crc32(buf[k * W * H * 3:(k + 1) * W * H * 3].cpu().numpy().tobytes())
for rep in range(2):
    for n in (1, 2, 4, 8):
        def run(k):
            for _ in range(k):
                dev.render_device_batch(p, n, rgb_u8=buf.data_ptr(), stream=st)
            torch.cuda.synchronize()
        run(2)
        k = max(4, 48 // n)
        t0 = time.perf_counter(); run(k); ms = (time.perf_counter() - t0) / (k * n) * 1e3
        print(f"{family} dragon, {n} frame(s) per launch, one stream: {ms:.3f} ms per frame  crc first {crc(0)} last {crc(n - 1)}", flush=True)
    for F in (1, 4):
        streams = [torch.cuda.Stream() for _ in range(F)]
        def run(k):
            for i in range(k):
                dev.render_device(p, rgb_u8=buf[(i % F) * W * H * 3:].data_ptr(), stream=streams[i % F].cuda_stream)
            torch.cuda.synchronize()
        run(2 * F)
        t0 = time.perf_counter(); run(48); ms = (time.perf_counter() - t0) / 48 * 1e3
        print(f"{family} dragon, single-frame launches on {F} stream(s): {ms:.3f} ms per frame  crc {crc(0)}", flush=True)
