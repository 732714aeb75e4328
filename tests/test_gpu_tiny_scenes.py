"""Scenes of one, two and three triangles (the root is a leaf; the smallest trees with an inner node): both builders,
both traversal modes, against the oracle."""
import numpy as np
import pytest

import ray_tracer_amd as rt
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 3, 5])
@pytest.mark.parametrize("builder", [rt.BUILDER_HOST, rt.BUILDER_DEVICE])
def test_tiny_scene(n, builder):
    rng = np.random.default_rng(40 + n)
    v = (rng.uniform(-2, 2, size=(n, 1, 3)) + rng.uniform(-1.5, 1.5, size=(n, 3, 3)) + np.array([0, 0, 6.0])).reshape(n, 9)
    colors = rng.integers(0, 256, size=(n, 3)).astype(np.int32)
    spec = np.ones(n, dtype=np.int32)
    refl = np.full(n, 0.3)
    kw = dict(camera_position=(0.0, 0.0, -2.0), ambient=0.3, directional_lights=[(0.6, (0.0, 1.0, -1.0))],
              point_lights=[(0.4, (1.0, 2.0, 0.0))])
    ora = OracleScene.from_arrays(v, colors, spec, refl, **kw)
    dev = rt.Scene.from_arrays(v, colors, spec, refl, builder=builder, **kw)
    assert dev.bvh_nodes == ora.n_nodes
    want = ora.render(96, 64, depth=3)
    assert (want["hit"] >= 0).any()
    for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_PRUNED):
        got = dev.render(96, 64, depth=3, traversal=mode)
        assert np.array_equal(got.hit, want["hit"]) and np.array_equal(got.rgb, want["rgb_u8"])
    dev.close()
