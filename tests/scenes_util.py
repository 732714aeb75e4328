"""Seeded synthetic scenes for the parity tests: the same arrays go to the oracle and to the HIP path."""
import numpy as np


def soup(seed, n=300, *, extent=4.0, tri_size=1.2, point_lights=True, exotic_spec=False):
    """A triangle soup in front of the camera with mixed materials."""
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-extent, extent, size=(n, 1, 3)) + np.array([0, 0, 10.0])
    verts = centers + rng.uniform(-tri_size, tri_size, size=(n, 3, 3))
    # a few axis-aligned and shared-edge triangles: equal-t ties and rays lying in box faces
    verts[0] = [[-6, -6, 14], [6, -6, 14], [6, 6, 14]]
    verts[1] = [[6, 6, 14], [-6, 6, 14], [-6, -6, 14]]
    verts[2] = verts[0]                      # exact duplicate: tie broken by leaf order
    colors = rng.integers(0, 256, size=(n, 3)).astype(np.int32)
    spec = rng.choice(np.array([-1, 1, 1, 1], dtype=np.int32), size=n)
    if exotic_spec:
        spec = rng.choice(np.array([-1, 1, 2, 10, 50], dtype=np.int32), size=n)
    refl = rng.choice(np.array([-1.0, 0.0, 0.2, 0.4, 0.9]), size=n)
    kw = dict(
        camera_position=(0.3, -0.2, -3.0),
        camera_rotation_deg=(float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5)), float(rng.uniform(-10, 10))),
        viewport=(1.0, 1.0, 1.0),
        background=(200, 220, 255),
        ambient=0.15,
        point_lights=[(0.6, (2.0, 5.0, 0.0)), (0.3, (-4.0, 1.0, 6.0))] if point_lights else [],
        directional_lights=[(0.4, (0.0, 1.0, -1.0)), (0.2, (1.0, 4.0, 4.0))],
    )
    return verts.reshape(n, 9), colors, spec, refl, kw


def untile_reference(gathered, W, H, tile_w, tile_h, n_ranks, bpp=3):
    """numpy restatement of the tile re-assembly: gathered is [rank][local tile][tile_h][tile_w][bpp]."""
    tiles_x = (W + tile_w - 1) // tile_w
    tiles_y = (H + tile_h - 1) // tile_h
    per_rank = (tiles_x * tiles_y + n_ranks - 1) // n_ranks
    g = np.asarray(gathered).reshape(n_ranks, per_rank, tile_h, tile_w, bpp)
    out = np.zeros((H, W, bpp), dtype=g.dtype)
    for k in range(tiles_x * tiles_y):
        ty, tx = divmod(k, tiles_x)
        r, lt = k % n_ranks, k // n_ranks
        y0, x0 = ty * tile_h, tx * tile_w
        h, w = min(tile_h, H - y0), min(tile_w, W - x0)
        out[y0:y0 + h, x0:x0 + w] = g[r, lt, :h, :w]
    return out


def owned_tiles(W, H, tile_w, tile_h, rank, n_ranks):
    tiles_x = (W + tile_w - 1) // tile_w
    tiles_y = (H + tile_h - 1) // tile_h
    return [k for k in range(tiles_x * tiles_y) if k % n_ranks == rank]


def pack_reference(frame, tile_w, tile_h, rank, n_ranks):
    """numpy restatement of the packed layout one rank renders: [local tile][tile_h][tile_w][bpp]."""
    H, W, bpp = frame.shape
    tiles_x = (W + tile_w - 1) // tile_w
    tiles_y = (H + tile_h - 1) // tile_h
    per_rank = (tiles_x * tiles_y + n_ranks - 1) // n_ranks
    out = np.zeros((per_rank, tile_h, tile_w, bpp), dtype=frame.dtype)
    for k in owned_tiles(W, H, tile_w, tile_h, rank, n_ranks):
        ty, tx = divmod(k, tiles_x)
        y0, x0 = ty * tile_h, tx * tile_w
        h, w = min(tile_h, H - y0), min(tile_w, W - x0)
        out[k // n_ranks, :h, :w] = frame[y0:y0 + h, x0:x0 + w]
    return out
