"""rtx_scene_group_*: one frame over the GPUs of a node from the C ABI alone (scene replicas, tiles dealt round-robin,
RCCL all-gather from ncclCommInitAll communicators, re-assembly on the first device).  A one-GPU box runs it with one
member: every step, the all-gather included, still goes through RCCL.  Members beyond the first are covered by the same
code with N = hipGetDeviceCount() on a multi-GPU node (test_every_gpu_of_the_node)."""
import ctypes as C
import hashlib
import json
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import ray_tracer_amd as rt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def teapot():
    return rt.HostScene.load_obj(os.path.join(HERE, "golden", "teapot.obj"), "teapot", no_bvh=True)


def test_one_member_group_equals_rtx_render(teapot):
    scene = rt.Scene(teapot)
    group = rt.SceneGroup(teapot, devices=[0])
    assert group.size == 1 and group.devices() == [0]
    for (W, H, kw) in [(640, 360, dict(traversal=rt.TRAVERSAL_PRUNED)), (200, 104, dict(traversal=rt.TRAVERSAL_REFERENCE, depth=1)),
                       (330, 170, dict(traversal=rt.TRAVERSAL_PRUNED, window=(10, 300, 8, 160), tile=(16, 8))),
                       (96, 64, dict(traversal=rt.TRAVERSAL_PRUNED, spp=4, seed=5))]:
        want = scene.render(W, H, **kw).rgb
        assert np.array_equal(group.render(W, H, **kw), want), (W, H, kw)
    group.close()


@pytest.mark.parametrize("gather", [rt.GATHER_ALL, rt.GATHER_ROOT])
def test_both_exchanges_with_one_member_and_frames_in_flight(teapot, gather):
    """The all-gather and the send/receive gather to the root (with one member: the root's own copy), ten frames over
    the four slots whose exchanges share the member's one exchange stream: every frame equals rtx_render's."""
    scene = rt.Scene(teapot)
    group = rt.SceneGroup(teapot, devices=[0])
    group.set_gather(gather)
    want = scene.render(400, 232, traversal=rt.TRAVERSAL_PRUNED).rgb
    assert np.array_equal(group.render(400, 232, traversal=rt.TRAVERSAL_PRUNED), want)
    p = rt.make_params(400, 232, traversal=rt.TRAVERSAL_PRUNED)
    for k in range(10):
        group.submit(p, k % 4)
    group.sync()
    assert np.array_equal(group.render(400, 232, traversal=rt.TRAVERSAL_PRUNED), want)
    with pytest.raises(rt.RtxError):
        group.set_gather(5)
    with pytest.raises(rt.RtxError):
        group.submit(rt.make_params(-4, 16), 0)              # refused before any buffer is touched
    group.close()


def test_frames_in_flight_on_the_four_slots(teapot):
    import torch
    scene = rt.Scene(teapot)
    group = rt.SceneGroup(teapot, n_devices=1)
    W, H = 512, 288
    want = scene.render(W, H, traversal=rt.TRAVERSAL_PRUNED).rgb
    p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
    ptrs = [group.submit(p, k % 4) for k in range(10)]        # more frames than slots
    group.sync()
    assert len(set(ptrs)) == 4
    # the frames live in buffers the group owns: rtx_untile with ONE tile as large as the frame is a device-to-device copy
    whole = rt.make_params(W, H, tile=(W, H), tile_count=1, layout=rt.LAYOUT_PACKED_TILES)
    for ptr in set(ptrs):
        mine = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
        rt.untile(whole, 1, ptr, mine.data_ptr(), 3, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(mine.cpu().numpy().reshape(H, W, 3), want)
    with pytest.raises(rt.RtxError):
        group.submit(p, 4)
    group.close()


def test_bad_device_lists_are_refused(teapot):
    with pytest.raises(rt.RtxError):
        rt.SceneGroup(teapot, devices=[0, 0])
    with pytest.raises(rt.RtxError):
        rt.SceneGroup(teapot, devices=[99])
    with pytest.raises(rt.RtxError):
        rt.SceneGroup(teapot, n_devices=99)


def test_every_gpu_of_the_node(teapot):
    """N = hipGetDeviceCount(): on a one-GPU box this is the one-member case again; on an 8-GPU node it is the real thing."""
    import torch
    n = torch.cuda.device_count()
    scene = rt.Scene(teapot)
    group = rt.SceneGroup(teapot, n_devices=0)
    assert group.size == n
    want = scene.render(800, 450, traversal=rt.TRAVERSAL_PRUNED).rgb
    assert np.array_equal(group.render(800, 450, traversal=rt.TRAVERSAL_PRUNED), want)
    group.close()


def test_reference_shaped_driver_over_a_gpu_group(teapot_obj, golden_dir, tmp_path):
    """main_like_reference --gpus 0 (every GPU of the node): the reference's main() shape, frames through rtx_render_sharded (RCCL)."""
    os.makedirs(tmp_path / "scenes")
    os.makedirs(tmp_path / "out")
    shutil.copyfile(teapot_obj, tmp_path / "scenes" / "teapot.obj")
    exe = os.path.join(ROOT, "ray_tracer_amd", "main_like_reference")
    for args in (["--whole-frame", "--gpus", "0"], ["--gpus", "0"]):
        run = subprocess.run([exe] + args, cwd=tmp_path, capture_output=True, text=True, timeout=200)
        assert run.returncode == 0, run.stderr
        lines = run.stdout.splitlines()
        assert lines[0] == "[ TRI ] 6330 triangles." and lines[1].startswith("[ BVH ] ") and len(lines) == 3
        assert re.fullmatch(r"(\d+\.\d{3} ){3}", lines[2])
        want = json.load(open(os.path.join(golden_dir, "golden.json")))["teapot_1920"]["ppm_md5"]
        assert hashlib.md5(open(tmp_path / "out" / "canvas.ppm", "rb").read()).hexdigest() == want
