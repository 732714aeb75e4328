import sys, os
sys.path.insert(0, os.getcwd())
import torch, ctypes as C
import ray_tracer_amd as rt
host = rt.HostScene.standin("dragon", seed=0, no_bvh=True)
dev = rt.Scene(host)
W, H = 3840, 2160
stream = torch.cuda.current_stream().cuda_stream
for world in (1, 2, 4, 8):
    for tile in (32, 16, 8):
        p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, tile=(tile, tile), tile_rank=0, tile_count=world,
                           layout=rt.LAYOUT_PACKED_TILES if world > 1 else rt.LAYOUT_ROW_MAJOR)
        px = int(rt.lib().rtx_output_pixels(C.byref(p)))
        buf = torch.zeros(px * 3, dtype=torch.uint8, device="cuda")
        ts = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); dev.render_device(p, rgb_u8=buf.data_ptr(), stream=stream); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print(f"world {world} tile {tile}: rank-0 kernel {min(ts[1:]):.3f} ms  (ideal {4.43/world:.3f})", flush=True)
