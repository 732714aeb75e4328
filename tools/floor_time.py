"""Per-frame time of small frames with 1 and 4 in flight: the fixed cost of a launch (work-counter reset, kernel start,
the longest pixel's dependent rounds) that a rank's share of a sharded frame cannot go below."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import ray_tracer_amd as rt
FL = [int(x) for x in os.environ.get("FLIST", "1,4").split(",")]

host = rt.HostScene.standin("dragon", seed=0, no_bvh=True)
dev = rt.Scene(host)
for W, H in ((64, 64), (256, 256), (960, 540), (1920, 1080), (3840, 2160)):
    p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
    out = {}
    for F in FL:
        streams = [torch.cuda.Stream() for _ in range(F)]
        bufs = [torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda") for _ in range(F)]
        def run(n):
            for k in range(n):
                dev.render_device(p, rgb_u8=bufs[k % F].data_ptr(), stream=streams[k % F].cuda_stream)
            torch.cuda.synchronize()
        run(8)
        t0 = time.perf_counter(); run(200); out[F] = (time.perf_counter() - t0) / 200 * 1e3
    print(f"{W}x{H}: per frame, {FL} in flight: " + " ".join(f"{out[F]:.3f}" for F in FL) + " ms", flush=True)
