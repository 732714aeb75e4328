"""Worker of tests/test_gpu_rccl_single_rank.py: the pipelined multi-GPU frame loop (render own tiles -> RCCL all-gather ->
re-assembly, three frames in flight) with a process group of backend "nccl" (= RCCL) and ONE rank, which is all a one-GPU
box can hold.  The process plays both ranks of a two-way tile split; the all-gather moves the packed buffer through RCCL."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 3000))
import torch, torch.distributed as dist, ctypes as C, zlib
import ray_tracer_amd as rt
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev,
                        pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
hs = rt.HostScene.load_obj(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "teapot.obj"), "teapot", no_bvh=True)
scene = rt.Scene(hs)
W, H = 1920, 1080
world = 1
params = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, tile=(32, 32), tile_rank=0, tile_count=2, layout=rt.LAYOUT_PACKED_TILES)
params1 = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, tile=(32, 32), tile_rank=1, tile_count=2, layout=rt.LAYOUT_PACKED_TILES)
px = int(rt.lib().rtx_output_pixels(C.byref(params)))
F = 3
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(F - 1)]
bufs = [torch.zeros(2 * px * 3, dtype=torch.uint8, device=dev) for _ in range(F)]
gath = [torch.zeros(2 * px * 3, dtype=torch.uint8, device=dev) for _ in range(F)]
frames = [torch.zeros(H * W * 3, dtype=torch.uint8, device=dev) for _ in range(F)]
for k in range(9):
    s = streams[k % F]
    with torch.cuda.stream(s):
        # one process plays both ranks' renders, the 1-rank all-gather moves the packed buffer
        scene.render_device(params, rgb_u8=bufs[k % F].data_ptr(), stream=s.cuda_stream)
        scene.render_device(params1, rgb_u8=bufs[k % F][px * 3:].data_ptr(), stream=s.cuda_stream)
        dist.all_gather_into_tensor(gath[k % F], bufs[k % F])
        rt.untile(params, 2, gath[k % F].data_ptr(), frames[k % F].data_ptr(), 3, s.cuda_stream)
torch.cuda.synchronize()
dist.barrier()
ref = scene.render(W, H, traversal=rt.TRAVERSAL_PRUNED).rgb
for f in frames:
    assert (f.cpu().numpy().reshape(H, W, 3) == ref).all()
print("nccl single-rank pipelined path ok")
dist.destroy_process_group()
