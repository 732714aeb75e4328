"""Several frames of one scene in flight on different streams: each must equal the frame rendered alone."""
import os

import numpy as np
import pytest

import ray_tracer_amd as rt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_overlapping_frames_equal_frames_rendered_alone():
    import torch
    hs = rt.HostScene.load_obj(os.path.join(HERE, "golden", "teapot.obj"), "teapot", no_bvh=True)
    scene = rt.Scene(hs)
    W, H = 640, 400
    jobs = [dict(traversal=rt.TRAVERSAL_PRUNED), dict(traversal=rt.TRAVERSAL_REFERENCE), dict(traversal=rt.TRAVERSAL_PRUNED, depth=1),
            dict(traversal=rt.TRAVERSAL_PRUNED, window=(100, 500, 50, 300)), dict(traversal=rt.TRAVERSAL_PRUNED, spp=2, seed=7),
            dict(traversal=rt.TRAVERSAL_REFERENCE, depth=0), dict(traversal=rt.TRAVERSAL_PRUNED)]
    alone = []
    for kw in jobs:
        buf = torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda")
        scene.render_device(rt.make_params(W, H, **kw), rgb_u8=buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        alone.append(buf.cpu().numpy())
    streams = [torch.cuda.Stream() for _ in range(3)]
    bufs = [torch.zeros(H * W * 3, dtype=torch.uint8, device="cuda") for _ in jobs]
    ticks = torch.zeros(2 * len(jobs), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):                       # more launches than scratch slots
        for k, kw in enumerate(jobs):
            s = streams[k % 3]
            with torch.cuda.stream(s):
                bufs[k].zero_()
                scene.render_device(rt.make_params(W, H, **kw), rgb_u8=bufs[k].data_ptr(), ticks=ticks[2 * k:].data_ptr(),
                                    stream=s.cuda_stream)
        torch.cuda.synchronize()
        for k in range(len(jobs)):
            assert np.array_equal(bufs[k].cpu().numpy(), alone[k]), (rep, k)
    t = ticks.cpu().numpy().astype(np.uint64).reshape(-1, 2)
    # every launch left its start (as its complement) and its end on the device clock
    assert (t[:, 0] != 0).all() and (t[:, 1] > ~t[:, 0]).all()


def test_sharded_renderer_single_rank_with_frames_in_flight():
    import torch
    from ray_tracer_amd.distributed import ShardedRenderer
    hs = rt.HostScene.load_obj(os.path.join(HERE, "golden", "teapot.obj"), "teapot", no_bvh=True)
    scene = rt.Scene(hs)
    want = scene.render(320, 200, traversal=rt.TRAVERSAL_PRUNED).rgb
    r = ShardedRenderer(scene, 320, 200, 0, 1, torch.device("cuda", torch.cuda.current_device()), inflight=4,
                        traversal=rt.TRAVERSAL_PRUNED)
    slots = [r.submit() for _ in range(9)]
    r.synchronize()
    assert slots == [0, 1, 2, 3, 0, 1, 2, 3, 0]
    for f in r.frames:
        assert np.array_equal(f.cpu().numpy().reshape(200, 320, 3), want)
    assert np.array_equal(r.render().cpu().numpy().reshape(200, 320, 3), want)


def test_render_begin_end_loop_delivers_the_same_frames():
    """rtx_render_begin / rtx_render_end: frame n + 1 queued before frame n is collected; every frame equals rtx_render's."""
    hs = rt.HostScene.load_obj(os.path.join(HERE, "golden", "teapot.obj"), "teapot", no_bvh=True)
    scene = rt.Scene(hs)
    W, H = 480, 272
    want = scene.render(W, H, traversal=rt.TRAVERSAL_PRUNED)
    p = rt.make_params(W, H, traversal=rt.TRAVERSAL_PRUNED)
    frames = 7
    scene.render_begin(p, 0, want_hit=True)
    for n in range(frames):
        if n + 1 < frames:
            scene.render_begin(p, (n + 1) % 2, want_hit=True)
        rgb = np.zeros((H, W, 3), dtype=np.uint8)
        hit = np.zeros((H, W), dtype=np.int32)
        scene.render_end(n % 2, rgb, hit)
        assert np.array_equal(rgb, want.rgb) and np.array_equal(hit, want.hit), n
    # a window: pixels outside it read 0 / -1, as with rtx_render
    kw = dict(traversal=rt.TRAVERSAL_PRUNED, window=(40, 400, 16, 200))
    want = scene.render(W, H, **kw)
    scene.render_begin(rt.make_params(W, H, **kw), 3, want_hit=True)
    rgb = np.full((H, W, 3), 9, dtype=np.uint8)
    hit = np.zeros((H, W), dtype=np.int32)
    scene.render_end(3, rgb, hit)
    assert np.array_equal(rgb, want.rgb) and np.array_equal(hit, want.hit)
    # slots hold one frame at a time
    with pytest.raises(rt.RtxError):
        scene.render_end(1, rgb)
    scene.render_begin(p, 1)
    with pytest.raises(rt.RtxError):
        scene.render_begin(p, 1)
    with pytest.raises(rt.RtxError):
        scene.render_end(1, rgb, hit)             # ids were not asked for
    scene.render_end(1, rgb)
    with pytest.raises(rt.RtxError):
        scene.render_begin(p, 4)


def test_default_params_select_the_pruned_traversal():
    import ctypes as C
    p = rt.RenderParams()
    rt.lib().rtx_render_params_default(C.byref(p), 64, 48)
    assert (p.traversal, p.depth, p.spp, p.tile_w, p.tile_h) == (rt.TRAVERSAL_PRUNED, 3, 1, 32, 32)


def test_lighting_jobs_change_nothing_but_the_schedule(monkeypatch):
    """At the end of a launch lanes without work take over the lighting of other lanes' surface hits
    (RTX_LIGHTING_JOBS, on by default).  Frames, ids, t and the census are those of the serial schedule; small frames
    are almost all tail, which is where the jobs run."""
    hs = rt.HostScene.load_obj(os.path.join(HERE, "golden", "teapot.obj"), "teapot", no_bvh=True)
    scene = rt.Scene(hs)
    for (W, H, kw) in [(64, 64, dict(traversal=rt.TRAVERSAL_PRUNED)), (200, 120, dict(traversal=rt.TRAVERSAL_REFERENCE)),
                       (96, 96, dict(traversal=rt.TRAVERSAL_PRUNED, depth=5)), (40, 24, dict(traversal=rt.TRAVERSAL_PRUNED, depth=0))]:
        frames = []
        for jobs in ("0", "1"):
            monkeypatch.setenv("RTX_LIGHTING_JOBS", jobs)
            f = scene.render(W, H, want_stats=True, **kw)
            st = f.stats.as_dict()
            st.pop("kernel_ms")
            for k in ("rounds", "lane_steps"):
                st.pop(k, None)
            frames.append((f.rgb, f.hit, st))
        assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1])
        assert frames[0][2] == frames[1][2], (frames[0][2], frames[1][2])
    scene.close()
