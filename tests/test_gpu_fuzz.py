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
from scenes_util import fuzz_case

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("RTX_FUZZ_CASES", "10"))
FIRST = int(os.environ.get("RTX_FUZZ_FIRST", "0"))


@pytest.mark.parametrize("seed", range(FIRST, FIRST + CASES))
def test_random_case_is_bit_exact(seed):
    family, (v, colors, spec, refl, kw), o = fuzz_case(seed)
    o["mode"] = [rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_PRUNED][o["mode"]]
    o["builder"] = [rt.BUILDER_HOST, rt.BUILDER_DEVICE][o["builder"]]
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
            assert got == [getattr(cen, name) for name, _ in cen._fields_[:9]], tag
    finally:
        dev.close()
        ora.close()


@pytest.mark.parametrize("seed", range(FIRST, FIRST + max(1, CASES // 2)))
def test_random_case_with_specular_exponents(seed):
    """The same campaign with exponents 2..1000 on part of the materials: ids and t still bit for bit; colours may
    differ by one level where the host's pow is not correctly rounded (see rtx_selftest_pow)."""
    family, (v, colors, spec, refl, kw), o = fuzz_case(seed, exotic_spec=True)
    mode = [rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_PRUNED][o["mode"]]
    tag = f"seed {seed} {family} n={v.shape[0]} {o}"
    ora = OracleScene.from_arrays(v, colors, spec, refl, **kw)
    dev = rt.Scene.from_arrays(v, colors, spec, refl, builder=[rt.BUILDER_HOST, rt.BUILDER_DEVICE][o["builder"]], **kw)
    try:
        W, H = o["W"], o["H"]
        want = ora.render(W, H, depth=o["depth"], spp=o["spp"], seed=o["seed"], want_t=True)
        cuda = torch.device("cuda")
        hit = torch.full((H, W), -7, dtype=torch.int32, device=cuda)
        t = torch.zeros((H, W), dtype=torch.float64, device=cuda)
        rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device=cuda)
        p = rt.make_params(W, H, depth=o["depth"], spp=o["spp"], seed=o["seed"], traversal=mode)
        dev.render_device(p, rgb_u8=rgb.data_ptr(), hit=hit.data_ptr(), t=t.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(hit.cpu().numpy(), want["hit"]), tag
        assert t.cpu().numpy().tobytes() == want["t"].tobytes(), tag
        diff = np.abs(rgb.cpu().numpy().astype(int) - want["rgb_u8"].astype(int))
        assert diff.max() <= 1 and (diff != 0).sum() <= max(2, diff.size // 2000), tag
    finally:
        dev.close()
        ora.close()
