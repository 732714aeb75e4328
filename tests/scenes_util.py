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


def renumber_with_gap(nodes):
    """The same tree with one unused record after the root, as builders that start their node counter at 2 number it:
    every left child then sits at an even index (the reference's builder puts them at odd ones, bvh.hpp:148-149)."""
    out = np.zeros(len(nodes) + 1, dtype=nodes.dtype)
    out[0] = nodes[0]
    out[2:] = nodes[1:]
    inner = out["count"] == 0
    inner[1] = False
    out["first"][inner] += 1
    return out


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


# ---------------------------------------------------------------- seeded odd scenes for the campaigns
def fuzz_geometry(rng):
    family = int(rng.integers(0, 7))
    if family == 0:      # plain soup
        n = int(rng.integers(4, 3000))
        # the last three: products of coordinates go subnormal, or overflow
        scale = float(rng.choice([1e-3, 0.1, 1.0, 1.0, 30.0, 1e4, 1.0, 1.0, 1e-3, 30.0, 1e-155, 1e-162, 1e140]))
        c = rng.uniform(-4, 4, size=(n, 1, 3)) * scale
        v = c + rng.uniform(-1.5, 1.5, size=(n, 3, 3)) * scale * float(rng.choice([0.05, 0.3, 1.0]))
        return "soup", v, scale
    if family == 1:      # axis-aligned lattice of quads on integer coordinates: ties everywhere
        k = int(rng.integers(2, 14))
        tris = []
        for i in range(k):
            for j in range(k):
                z = float(rng.integers(0, 3))
                a, b, c, d = (i, j, z), (i + 1, j, z), (i + 1, j + 1, z), (i, j + 1, z)
                tris += [(a, b, c), (c, d, a)]
        v = np.array(tris, dtype=np.float64) - np.array([k / 2, k / 2, 0.0])
        return "lattice", v, float(k)
    if family == 2:      # stacks of identical and nearly identical triangles
        base = rng.uniform(-2, 2, size=(int(rng.integers(1, 6)), 3, 3))
        reps = int(rng.integers(2, 40))
        v = np.repeat(base, reps, axis=0)
        v[::3] += rng.uniform(-1e-12, 1e-12, size=v[::3].shape)
        return "stacks", v, 2.0
    if family == 3:      # slivers and zero-area triangles among ordinary ones
        n = int(rng.integers(20, 600))
        v = rng.uniform(-3, 3, size=(n, 3, 3))
        v[::5, 2] = v[::5, 1]                                  # two equal corners
        v[1::7, 2] = (v[1::7, 0] + v[1::7, 1]) / 2             # three corners on a line
        v[2::9, 1] = v[2::9, 0] + rng.uniform(-1e-9, 1e-9, size=v[2::9, 0].shape)
        return "slivers", v, 3.0
    if family == 4:      # two far clusters of very different size
        n = int(rng.integers(50, 1500))
        v = rng.uniform(-1, 1, size=(n, 3, 3)) * 0.01
        v[: n // 2] = v[: n // 2] * 3000 + np.array([500.0, 0, 0])
        return "clusters", v, 50.0
    if family == 5:      # closed box around the origin (camera may sit inside), plus clutter
        s = 5.0
        p = [(-s, -s, -s), (s, -s, -s), (s, s, -s), (-s, s, -s), (-s, -s, s), (s, -s, s), (s, s, s), (-s, s, s)]
        quads = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 3, 7, 4)]
        tris = []
        for a, b, c, d in quads:
            tris += [(p[a], p[b], p[c]), (p[c], p[d], p[a])]
        clutter = rng.uniform(-4, 4, size=(int(rng.integers(0, 300)), 1, 3)) + rng.uniform(-0.5, 0.5, size=(1, 3, 3))
        v = np.concatenate([np.array(tris, dtype=np.float64), clutter.reshape(-1, 3, 3)]) if clutter.size else np.array(tris, dtype=np.float64)
        return "room", v, 5.0
    # one long row: a deep, lopsided tree
    n = int(rng.integers(30, 400))
    x = np.cumsum(rng.uniform(0.01, 1.0, size=n) * np.geomspace(1, float(rng.choice([1, 100, 1e6])), n))
    v = np.zeros((n, 3, 3))
    v[:, :, 0] = x[:, None] + rng.uniform(0, 0.3, size=(n, 3))
    v[:, :, 1] = rng.uniform(-1, 1, size=(n, 3))
    v[:, :, 2] = rng.uniform(-1, 1, size=(n, 3))
    v -= np.array([x[n // 3], 0, 0])
    return "row", v, float(x[n // 2] - x[n // 3] + 2.0)


def fuzz_case(seed, exotic_spec=False):
    rng = np.random.default_rng(1000 + seed)
    family, v, size = fuzz_geometry(rng)
    v = np.ascontiguousarray(v.reshape(-1, 9))
    n = v.shape[0]
    colors = rng.integers(0, 256, size=(n, 3)).astype(np.int32)
    spec = rng.choice(np.array([-1, 1, 1], dtype=np.int32), size=n)           # pow(x, 1) is exact on both sides
    if exotic_spec:                                                            # same libm on both sides only
        spec = np.where(rng.random(n) < 0.4, rng.choice(np.array([2, 3, 10, 50, 1000], dtype=np.int32), size=n), spec).astype(np.int32)
    refl = rng.choice(np.array([-1.0, 0.0, 0.2, 0.5, 1.0]), size=n)
    where = int(rng.integers(0, 3))
    if where == 0:
        cam = (float(rng.uniform(-0.3, 0.3)) * size, float(rng.uniform(-0.3, 0.3)) * size, -3.0 * size)
    elif where == 1:
        cam = tuple(float(x) for x in rng.uniform(-0.5, 0.5, size=3) * size)          # inside the geometry
    else:
        cam = (0.0, 0.0, -2.0 * size)                                                 # on the axes: zero ray components
    rot_menu = [(0.0, 0.0, 0.0), (0.0, 0.1, 0.0), (0.0, 90.0, 0.0), (0.0, 180.0, 0.0),
                tuple(float(x) for x in rng.uniform(-30, 30, size=3))]
    rot = rot_menu[int(rng.integers(0, len(rot_menu)))]
    n_point, n_dir = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    kw = dict(
        camera_position=cam, camera_rotation_deg=rot, viewport=(1.0, float(rng.choice([1.0, 0.5625])), 1.0),
        background=tuple(int(x) for x in rng.integers(0, 256, size=3)), ambient=float(rng.choice([0.0, 0.2, 0.7])),
        point_lights=[(float(rng.uniform(0.1, 0.8)), tuple(float(x) for x in rng.uniform(-2, 2, size=3) * size)) for _ in range(n_point)],
        directional_lights=[(float(rng.uniform(0.1, 0.8)), tuple(float(x) for x in rng.choice([0.0, 1.0, -1.0, 0.3], size=3)))
                            for _ in range(n_dir)],
    )
    if rng.random() < 0.08:
        # lights bright enough to push colour * lighting past the int range (color.hpp:47-53)
        boost = float(rng.choice([1e8, 1e12, 1e300]))
        kw["point_lights"] = [(i * boost, p) for i, p in kw["point_lights"]]
        kw["directional_lights"] = [(i * boost, p) for i, p in kw["directional_lights"]]
    opts = dict(
        W=int(rng.choice([8, 33, 64, 96, 130])), H=int(rng.choice([8, 31, 64, 72])), depth=int(rng.integers(0, 6)),
        spp=int(rng.choice([1, 1, 1, 4])), seed=int(rng.integers(0, 1 << 30)),
        mode=int(rng.integers(0, 2)),        # 0: the reference's traversal, 1: the pruned one
        builder=int(rng.integers(0, 2)),     # 0: host builder, 1: device builder
    )
    return family, (v, colors, spec, refl, kw), opts
