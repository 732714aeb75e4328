"""GPU parity: the HIP path through the C ABI against the CPU oracle on the same scenes.

Bar: primary hit ids, primary t, unclamped RGB ints and 8-bit RGB all bit-exact
(the north-star tolerance is >= 99.99 % ids and +-1 LSB; the tests hold the path to 100 % / 0).
"""
import numpy as np
import pytest

import ray_tracer_amd as rt
from oracle_binding import OracleScene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def teapot(teapot_obj):
    host = rt.HostScene.load_obj(teapot_obj, "teapot")
    dev = rt.Scene(host)
    ora = OracleScene.load_obj(teapot_obj, "teapot")
    yield host, dev, ora
    dev.close()


@pytest.mark.parametrize("mode", [rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_PRUNED])
@pytest.mark.parametrize("size", [(256, 256), (320, 180), (64, 200)])
def test_teapot_ids_and_rgb_bit_exact(teapot, mode, size):
    _, dev, ora = teapot
    W, H = size
    want = ora.render(W, H, depth=3)
    got = dev.render(W, H, depth=3, traversal=mode)
    assert np.array_equal(got.hit, want["hit"]), f"{(got.hit != want['hit']).sum()} ids differ"
    assert np.array_equal(got.rgb, want["rgb_u8"])


def test_teapot_census_matches_reference_traversal(teapot):
    _, dev, ora = teapot
    want = ora.render(200, 200, depth=3)["census"].as_dict()
    got = dev.render(200, 200, depth=3, traversal=rt.TRAVERSAL_REFERENCE, want_stats=True).stats.as_dict()
    got.pop("kernel_ms")
    assert got == want


@pytest.mark.parametrize("depth", [0, 1, 2, 5])
def test_teapot_depths(teapot, depth):
    _, dev, ora = teapot
    want = ora.render(128, 128, depth=depth)
    for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_PRUNED):
        got = dev.render(128, 128, depth=depth, traversal=mode)
        assert np.array_equal(got.hit, want["hit"])
        assert np.array_equal(got.rgb, want["rgb_u8"])
