"""rtx_render_device_batch(): several frames in one launch.  Every frame of the launch must be the frame a launch of
its own renders: same camera for all of them, a camera per frame, several samples per pixel, a rank's packed tiles."""
import ctypes as C
import os

import numpy as np
import pytest

import ray_tracer_amd as rt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

CAMERAS = [(-0.015, 4.0, -11.99, 0.0, 0.1, 0.0),          # the teapot preset's own (scene.hpp:259-260)
           (1.5, 3.0, -10.0, 3.0, -8.0, 1.0),
           (-2.0, 5.5, -9.0, 10.0, 12.0, -29.266614452137418),
           (0.0, 2.0, -14.0, 0.0, 0.0, 90.0)]


@pytest.fixture(scope="module")
def host():
    return rt.HostScene.load_obj(os.path.join(HERE, "golden", "teapot.obj"), "teapot", no_bvh=True)


def _alone(host, cam, W, H, **kw):
    host.set_camera(cam[:3], cam[3:])
    scene = rt.Scene(host)
    rgb = scene.render(W, H, **kw).rgb
    scene.close()
    return rgb


@pytest.mark.parametrize("kw", [dict(traversal=rt.TRAVERSAL_PRUNED), dict(traversal=rt.TRAVERSAL_REFERENCE, depth=1),
                                dict(traversal=rt.TRAVERSAL_PRUNED, spp=4, seed=9)])
def test_frames_of_one_launch_equal_frames_launched_alone(host, kw):
    import torch
    W, H = 328, 200
    want = [_alone(host, cam, W, H, **kw) for cam in CAMERAS]
    host.set_camera(CAMERAS[0][:3], CAMERAS[0][3:])
    scene = rt.Scene(host)
    p = rt.make_params(W, H, **kw)
    n = len(CAMERAS)
    out = torch.zeros(n * W * H * 3, dtype=torch.uint8, device="cuda")
    ints = torch.zeros(n * W * H * 3, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    scene.render_device_batch(p, n, cameras=np.array(CAMERAS), rgb_u8=out.data_ptr(), rgb_i32=ints.data_ptr(), stream=st)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(n, H, W, 3)
    for f in range(n):
        assert np.array_equal(got[f], want[f]), f
    # no cameras: the scene's own, n times; and a second launch on the same scratch slot sees clean work counters
    for rep in range(2):
        out.zero_()
        scene.render_device_batch(p, 3, rgb_u8=out.data_ptr(), stream=st)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(n, H, W, 3)
        for f in range(3):
            assert np.array_equal(got[f], want[0]), (rep, f)
        assert not got[3].any()
    scene.close()


def test_a_ranks_packed_tiles_over_several_frames(host):
    import torch
    W, H = 200, 136
    host.set_camera(CAMERAS[0][:3], CAMERAS[0][3:])
    scene = rt.Scene(host)
    p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED, tile=(32, 16), tile_rank=1, tile_count=3, layout=rt.LAYOUT_PACKED_TILES)
    px = int(rt.lib().rtx_output_pixels(C.byref(p)))
    one = torch.zeros(px * 3, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    scene.render_device(p, rgb_u8=one.data_ptr(), stream=st)
    many = torch.zeros(2 * px * 3, dtype=torch.uint8, device="cuda")
    scene.render_device_batch(p, 2, rgb_u8=many.data_ptr(), stream=st)
    torch.cuda.synchronize()
    a, b = one.cpu().numpy(), many.cpu().numpy().reshape(2, -1)
    assert np.array_equal(b[0], a) and np.array_equal(b[1], a)
    for bad in (0, 9):
        with pytest.raises(rt.RtxError):
            scene.render_device_batch(p, bad, rgb_u8=many.data_ptr(), stream=st)
    scene.close()
