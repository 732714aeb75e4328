"""Seeded campaign of odd scenes: the HIP path through the C ABI against the CPU oracle, bit for bit.

Every case draws a geometry family, a camera (sometimes inside the geometry, sometimes with rotations
that give rays exact zero components), lights, materials, recursion depth, supersampling, the BVH
builder (host or device) and the traversal mode from its seed.  Compared: primary hit ids, primary t,
unclamped RGB ints, 8-bit RGB, and for the reference traversal the census of rays, node visits and
triangle tests.  RTX_FUZZ_CASES sets how many cases run (default: a handful, seconds); RTX_FUZZ_FIRST
the first seed.
"""
import os

import numpy as np
import pytest
import torch

import ray_tracer_amd as rt
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("RTX_FUZZ_CASES", "10"))
FIRST = int(os.environ.get("RTX_FUZZ_FIRST", "0"))


def _geometry(rng):
    family = int(rng.integers(0, 7))
    if family == 0:      # plain soup
        n = int(rng.integers(4, 3000))
        scale = float(rng.choice([1e-3, 0.1, 1.0, 1.0, 30.0, 1e4]))
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


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    family, v, size = _geometry(rng)
    v = np.ascontiguousarray(v.reshape(-1, 9))
    n = v.shape[0]
    colors = rng.integers(0, 256, size=(n, 3)).astype(np.int32)
    spec = rng.choice(np.array([-1, 1, 1], dtype=np.int32), size=n)           # pow(x, 1) is exact on both sides
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
    opts = dict(
        W=int(rng.choice([8, 33, 64, 96, 130])), H=int(rng.choice([8, 31, 64, 72])), depth=int(rng.integers(0, 6)),
        spp=int(rng.choice([1, 1, 1, 4])), seed=int(rng.integers(0, 1 << 30)),
        mode=int(rng.choice([rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_PRUNED])),
        builder=int(rng.choice([rt.BUILDER_HOST, rt.BUILDER_DEVICE])),
    )
    return family, (v, colors, spec, refl, kw), opts


@pytest.mark.parametrize("seed", range(FIRST, FIRST + CASES))
def test_random_case_is_bit_exact(seed):
    family, (v, colors, spec, refl, kw), o = _case(seed)
    tag = f"seed {seed} {family} n={v.shape[0]} {o}"
    ora = OracleScene.from_arrays(v, colors, spec, refl, **kw)
    dev = rt.Scene.from_arrays(v, colors, spec, refl, builder=o["builder"], **kw)
    try:
        assert dev.bvh_nodes == ora.n_nodes, tag
        W, H = o["W"], o["H"]
        want = ora.render(W, H, depth=o["depth"], spp=o["spp"], seed=o["seed"], want_t=True)
        cuda = torch.device("cuda")
        hit = torch.full((H, W), -7, dtype=torch.int32, device=cuda)
        t = torch.zeros((H, W), dtype=torch.float64, device=cuda)
        rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device=cuda)
        rgb_i32 = torch.zeros((H, W, 3), dtype=torch.int32, device=cuda)
        stats = torch.zeros(16, dtype=torch.int64, device=cuda)
        p = rt.make_params(W, H, depth=o["depth"], spp=o["spp"], seed=o["seed"], traversal=o["mode"])
        dev.render_device(p, rgb_u8=rgb.data_ptr(), hit=hit.data_ptr(), rgb_i32=rgb_i32.data_ptr(), t=t.data_ptr(),
                          stats=stats.data_ptr() if o["mode"] == rt.TRAVERSAL_REFERENCE else 0)
        torch.cuda.synchronize()
        assert np.array_equal(hit.cpu().numpy(), want["hit"]), tag
        assert t.cpu().numpy().tobytes() == want["t"].tobytes(), tag
        if o["spp"] == 1:
            assert np.array_equal(rgb_i32.cpu().numpy(), want["rgb_i32"]), tag
        assert np.array_equal(rgb.cpu().numpy(), want["rgb_u8"]), tag
        if o["mode"] == rt.TRAVERSAL_REFERENCE:
            got = [int(x) for x in stats.cpu().numpy()[:9]]
            cen = want["census"]
            assert got == [getattr(cen, name) for name, _ in cen._fields_], tag
    finally:
        dev.close()
        ora.close()
