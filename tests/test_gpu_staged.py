"""rtx_render_device_staged(): the frame as one launch per level of the reflection chain, rays regrouped between
launches.  It must return what rtx_render_device() returns -- 8-bit and unclamped colours, primary ids, primary t and
the work counters, bit for bit -- and what the oracle computes, for every depth, light set, window, tile layout and
traversal mode."""
import ctypes as C

import numpy as np
import pytest

import ray_tracer_amd as rt
from oracle_binding import OracleScene
from scenes_util import soup

pytestmark = pytest.mark.gpu


def _both(scene, W, H, *, stats=True, **kw):
    """The same frame through both pipelines: dicts of rgb, ints, hit, t, counters."""
    import torch
    kw.setdefault("traversal", rt.TRAVERSAL_PRUNED)
    p = rt.make_params(W, H, **kw)
    px = int(rt.lib().rtx_output_pixels(C.byref(p)))
    row_major = p.layout == rt.LAYOUT_ROW_MAJOR
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for call in (scene.render_device, scene.render_device_staged):
        rgb = torch.full((px * 3,), 7, dtype=torch.uint8, device="cuda")
        ints = torch.full((px * 3,), -5, dtype=torch.int32, device="cuda")
        hit = torch.full((px,), -9, dtype=torch.int32, device="cuda")
        t = torch.full((px,), -1.0, dtype=torch.float64, device="cuda")
        cnt = torch.zeros(15, dtype=torch.int64, device="cuda")
        call(p, rgb_u8=rgb.data_ptr(), rgb_i32=ints.data_ptr(), hit=hit.data_ptr() if row_major else 0,
             t=t.data_ptr() if row_major else 0, stats=cnt.data_ptr() if stats else 0, stream=st)
        torch.cuda.synchronize()
        out.append(dict(rgb=rgb.cpu().numpy(), ints=ints.cpu().numpy(), hit=hit.cpu().numpy(), t=t.cpu().numpy(),
                        counters=cnt.cpu().numpy()[:9]))
    return out


def _same(a, b):
    for k in ("rgb", "ints", "hit", "counters"):
        assert np.array_equal(a[k], b[k]), k
    assert a["t"].tobytes() == b["t"].tobytes()


@pytest.fixture(scope="module")
def teapot(teapot_obj):
    host = rt.HostScene.load_obj(teapot_obj, "teapot", no_bvh=True)
    dev = rt.Scene(host)
    yield dev, OracleScene.load_obj(teapot_obj, "teapot")
    dev.close()


@pytest.mark.parametrize("mode", [rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_PRUNED])
@pytest.mark.parametrize("depth", [0, 1, 3, 5])
def test_teapot_staged_equals_fused_and_oracle(teapot, mode, depth):
    dev, ora = teapot
    W, H = 264, 200
    fused, staged = _both(dev, W, H, depth=depth, traversal=mode)
    _same(fused, staged)
    want = ora.render(W, H, depth=depth)
    assert np.array_equal(staged["hit"].reshape(H, W), want["hit"])
    assert np.array_equal(staged["rgb"].reshape(H, W, 3), want["rgb_u8"])
    if mode == rt.TRAVERSAL_REFERENCE:
        census = want["census"].as_dict()
        names = ["primary_rays", "reflection_rays", "shadow_rays", "shadow_occluded", "closest_hits", "node_visits", "box_hits",
                 "leaf_visits", "triangle_tests"]
        assert [int(x) for x in staged["counters"]] == [census[n] for n in names]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_soups_with_point_lights_and_odd_materials(seed):
    """Several lights of both kinds, specular exponents other than 1, triangles that do not reflect, turned cameras."""
    v, col, spec, refl, cam = soup(seed, exotic_spec=True)
    host = rt.HostScene.from_arrays(v, col, spec, refl, **cam)
    dev = rt.Scene(host)
    ora = OracleScene.from_arrays(v, col, spec, refl, **cam)
    W, H = 160, 120
    for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_PRUNED):
        for depth in (2, 4):
            fused, staged = _both(dev, W, H, depth=depth, traversal=mode)
            _same(fused, staged)
    want = ora.render(W, H, depth=4)
    assert np.array_equal(staged["hit"].reshape(H, W), want["hit"])
    got, ref = staged["rgb"].reshape(H, W, 3).astype(int), want["rgb_u8"].astype(int)
    assert np.abs(got - ref).max() <= 1 and (got != ref).mean() < 1e-4      # pow(x, n >= 2): glibc's last bit, as in the fused tests
    dev.close()


def test_no_lights_and_dark_scene():
    """No light at all: the lighting pass finishes every hit without a query."""
    v, col, spec, refl, cam = soup(7, point_lights=False)
    cam = dict(cam, directional_lights=[])
    dev = rt.Scene(rt.HostScene.from_arrays(v, col, spec, refl, **cam))
    fused, staged = _both(dev, 120, 96, depth=3, traversal=rt.TRAVERSAL_PRUNED)
    _same(fused, staged)
    assert int(staged["counters"][2]) == 0
    dev.close()


def test_window_and_packed_tiles(teapot):
    dev, _ = teapot
    W, H = 200, 136
    fused, staged = _both(dev, W, H, window=(37, 150, 11, 97), depth=3)
    _same(fused, staged)
    assert (staged["rgb"].reshape(H, W, 3)[:11] == 7).all()           # outside the window nothing is written
    for rank in (0, 2):
        fused, staged = _both(dev, W, H, tile=(32, 16), tile_rank=rank, tile_count=3, layout=rt.LAYOUT_PACKED_TILES, depth=2)
        for k in ("rgb", "ints", "counters"):
            assert np.array_equal(fused[k], staged[k]), (rank, k)


def test_frames_in_a_row_and_in_flight(teapot):
    """The record arrays and work counters of a scratch slot are reused by the next frame of another size; frames on
    several streams each have their own."""
    import torch
    dev, _ = teapot
    for W, H in ((320, 200), (96, 64), (328, 264)):
        fused, staged = _both(dev, W, H, depth=3)
        _same(fused, staged)
    W, H = 256, 160
    p = rt.make_params(W, H, depth=3)
    want = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
    dev.render_device(p, rgb_u8=want.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    bufs = [torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda") for _ in range(12)]
    for i, b in enumerate(bufs):
        dev.render_device_staged(p, rgb_u8=b.data_ptr(), stream=streams[i % 4].cuda_stream)
    torch.cuda.synchronize()
    for b in bufs:
        assert torch.equal(b, want)


def test_several_samples_are_refused(teapot):
    dev, _ = teapot
    p = rt.make_params(64, 64, spp=4)
    with pytest.raises(rt.RtxError):
        dev.render_device_staged(p, rgb_u8=1)


def test_full_size_standin_frame_equals_fused():
    """BASELINE's dragon frame (3840x2160, 871 316 triangles, knotted-tube stand-in): every pixel, id, t and counter."""
    host = rt.HostScene.standin("dragon", seed=0, no_bvh=True, family="knot")
    dev = rt.Scene(host)
    fused, staged = _both(dev, 3840, 2160, depth=3, traversal=rt.TRAVERSAL_PRUNED)
    _same(fused, staged)
    dev.close()
