"""Worker of the GPU test of the sharded frame loop: `world` ranks on ONE GPU, gloo for the exchange (RCCL wants one GPU
per rank), the library's own ShardedRenderer otherwise: render own tiles -> all-gather -> re-assembly, frames in flight."""
import os
import sys


def main(rank, world, port, W, H, result_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch
    import torch.distributed as dist

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import ray_tracer_amd as rt
    from ray_tracer_amd.distributed import ShardedRenderer

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    scene = rt.Scene(rt.HostScene.load_obj(os.path.join(here, "golden", "teapot.obj"), "teapot", no_bvh=True), device=0)
    want = scene.render(W, H, traversal=rt.TRAVERSAL_PRUNED).rgb
    renderer = ShardedRenderer(scene, W, H, rank, world, dev, tile=16, inflight=2, traversal=rt.TRAVERSAL_PRUNED)
    ok = True
    slots = [renderer.submit() for _ in range(4)]
    renderer.synchronize()
    for slot in set(slots):
        ok = ok and np.array_equal(renderer.frames[slot].cpu().numpy().reshape(H, W, 3), want)
    ok = ok and np.array_equal(renderer.render().cpu().numpy().reshape(H, W, 3), want)
    dist.barrier()
    dist.destroy_process_group()
    with open(f"{result_path}.{rank}", "w") as f:
        f.write("ok" if ok else "mismatch")


if __name__ == "__main__":
    a = sys.argv
    main(int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), a[6])
