"""Worker of the world_size-2 gloo test (CPU): the host logic of the sharded frame --
tile ownership, equal packed sizes, gather order, re-assembly."""
import os
import sys

import numpy as np


def main(rank, world, port, W, H, tile, result_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    from ray_tracer_amd import distributed as rd
    from scenes_util import pack_reference, untile_reference

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)                       # same "frame" on every rank
    frame = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    p = rd.shard_params(W, H, rank, world, tile=tile)
    n = rd.output_pixels(p)
    packed = pack_reference(frame, tile, tile, rank, world)
    assert packed.size == n * 3, (packed.size, n)
    gathered = rd.gather_packed(torch.from_numpy(packed.reshape(-1)), world)
    whole = untile_reference(gathered.numpy(), W, H, tile, tile, world)
    ok = bool(np.array_equal(whole, frame))
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n]))
    ok = ok and len({int(s) for s in sizes}) == 1          # every rank's buffer has the same size
    dist.barrier()
    dist.destroy_process_group()
    with open(f"{result_path}.{rank}", "w") as f:
        f.write("ok" if ok else "mismatch")


if __name__ == "__main__":
    a = sys.argv
    main(int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), a[7])
