"""The GPU BVH builder against the host restatement of the reference builder
(include/bvh.hpp:84-162,236-305), which the CPU tests pin to the reference's own arrays:
node records and `indices` must be byte-identical."""
import os

import numpy as np
import pytest

import ray_tracer_amd as rt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def same_tree(vertices):
    host = rt.build_bvh(vertices, rt.BUILDER_HOST)
    dev = rt.build_bvh(vertices, rt.BUILDER_DEVICE)
    assert dev["nodes"].shape == host["nodes"].shape
    assert np.array_equal(dev["indices"], host["indices"])
    assert dev["nodes"].tobytes() == host["nodes"].tobytes()
    assert dev["depth"] == host["depth"]
    return host, dev


def test_teapot_tree_is_identical():
    hs = rt.HostScene.load_obj(os.path.join(HERE, "golden", "teapot.obj"), "teapot")
    host, _ = same_tree(hs.arrays()["vertices"])
    assert host["nodes"].shape[0] == 12541


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (3, 3), (17, 4), (64, 5), (65, 6), (1000, 7), (4097, 8), (50000, 9)])
def test_random_soups(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.uniform(-4, 4, size=(n, 1, 3)) + rng.uniform(-1.2, 1.2, size=(n, 3, 3))
    same_tree(v.reshape(n, 9))


def test_lattice_with_ties_zeros_and_duplicates():
    """Coordinates on a coarse integer lattice around the origin: many equal centroids, many
    zero coordinates of both signs, duplicate triangles, empty bins and one-sided partitions."""
    rng = np.random.default_rng(11)
    v = rng.integers(-2, 3, size=(3000, 9)).astype(np.float64)
    v[rng.random(v.shape) < 0.2] = -0.0
    v[1000:1500] = v[:500]
    same_tree(v)


def test_clustered_and_degenerate_triangles():
    rng = np.random.default_rng(12)
    centers = rng.normal(size=(20, 1, 3)) * 50
    v = (centers + rng.normal(size=(20, 400, 3)) * 0.01).reshape(-1, 1, 3) + rng.normal(size=(8000, 3, 3)) * 1e-3
    v = v.reshape(-1, 9)
    v[::7, 3:6] = v[::7, 0:3]          # zero-area triangles
    v[::11] = np.tile(v[::11, :3], 3)  # points
    same_tree(v)


def test_sorted_slivers_make_a_deep_tree():
    """Geometrically growing spacing: every split peels off a few triangles, so the tree is a long chain."""
    x = np.cumsum(1.5 ** np.arange(60))
    v = np.zeros((60, 9))
    v[:, 0], v[:, 3], v[:, 6] = x, x + 0.1, x
    v[:, 4], v[:, 8] = 0.1, 0.1
    host, _ = same_tree(v)
    assert host["depth"] > 12


def test_more_level_nodes_than_estimated_builds_again_with_the_true_bound(monkeypatch):
    """The level passes size their per-node arrays by an estimate (n / 8 + 1024 nodes of more than 64 triangles and
    their children); a mesh that needs more must not write past them and is built again with room for 2n.  The
    estimate is forced down here (a chain that really exceeds it needs coordinates beyond the double range)."""
    x = np.cumsum(1.2 ** np.arange(2000))
    v = np.zeros((2000, 9))
    v[:, 0], v[:, 3], v[:, 6] = x, x * (1 + 1e-3), x
    v[:, 4], v[:, 8] = 0.1, 0.1
    host = rt.build_bvh(v, rt.BUILDER_HOST)
    assert host["depth"] > 100
    for estimate in ("4", "50", "300"):
        monkeypatch.setenv("RTX_BVH_LEVEL_NODES", estimate)
        dev = rt.build_bvh(v, rt.BUILDER_DEVICE)
        assert dev["nodes"].tobytes() == host["nodes"].tobytes() and np.array_equal(dev["indices"], host["indices"])
    hs = rt.HostScene.standin("bunny", 20010, seed=5)
    want = hs.arrays()
    monkeypatch.setenv("RTX_BVH_LEVEL_NODES", "16")
    dev = rt.build_bvh(want["vertices"], rt.BUILDER_DEVICE)
    assert dev["nodes"].tobytes() == want["nodes"].tobytes() and np.array_equal(dev["indices"], want["indices"])


def test_standin_bunny_tree_and_render():
    hs = rt.HostScene.standin("bunny", 20010, seed=3)
    before = hs.arrays()
    hs.rebuild_bvh(rt.BUILDER_DEVICE)
    after = hs.arrays()
    assert after["nodes"].tobytes() == before["nodes"].tobytes()
    assert np.array_equal(after["indices"], before["indices"])
    frame = rt.Scene(hs).render(96, 54, traversal=rt.TRAVERSAL_PRUNED)
    assert (frame.hit >= 0).any()


def test_full_size_dragon_tree():
    hs = rt.HostScene.standin("dragon", seed=1)
    a = hs.arrays()
    dev = rt.build_bvh(a["vertices"], rt.BUILDER_DEVICE)
    assert dev["nodes"].tobytes() == a["nodes"].tobytes()
    assert np.array_equal(dev["indices"], a["indices"])


def _render_both(scene, w, h):
    out = []
    for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_PRUNED):
        f = scene.render(w, h, traversal=mode, want_stats=True)
        out.append((f.rgb.copy(), f.hit.copy(), f.stats.as_dict()))
    return out


def test_scene_built_on_device_equals_scene_from_host_tree():
    """rtx_scene_create (tree built and flattened on the device, normals computed there) against
    rtx_scene_create_from_host (host tree, host normals): same pixels, ids and work counters."""
    from scenes_util import soup
    verts, colors, spec, refl, kw = soup(21, 5000)
    host_scene = rt.Scene(rt.HostScene.from_arrays(verts, colors, spec, refl, **kw))
    dev_scene = rt.Scene.from_arrays(verts, colors, spec, refl, **kw)
    assert dev_scene.build_seconds > 0
    assert dev_scene.device_bytes == host_scene.device_bytes
    for (rgb_a, hit_a, st_a), (rgb_b, hit_b, st_b) in zip(_render_both(host_scene, 160, 120), _render_both(dev_scene, 160, 120)):
        assert np.array_equal(hit_a, hit_b)
        assert np.array_equal(rgb_a, rgb_b)
        for key in ("primary_rays", "reflection_rays", "shadow_rays", "node_visits", "triangle_tests", "leaf_visits"):
            assert st_a[key] == st_b[key], key


def test_scene_built_on_device_with_given_normals_and_host_builder_choice():
    hs = rt.HostScene.load_obj(os.path.join(HERE, "golden", "teapot.obj"), "teapot")
    a = hs.arrays()
    kw = dict(camera_position=a["camera_position"], camera_rotation_deg=a["camera_rotation_deg"],
              directional_lights=[(0.5, (0.0, 1.0, -1.0))], ambient=0.2)
    ref = rt.Scene(hs).render(128, 128, traversal=rt.TRAVERSAL_REFERENCE)
    for builder, normals in ((rt.BUILDER_DEVICE, a["normals"]), (rt.BUILDER_DEVICE, None), (rt.BUILDER_HOST, None)):
        sc = rt.Scene.from_arrays(a["vertices"], a["colors"], a["specularity"], a["reflectivity"], builder=builder,
                                  normals=normals, **kw)
        f = sc.render(128, 128, traversal=rt.TRAVERSAL_REFERENCE)
        assert np.array_equal(f.hit, ref.hit)
        assert np.array_equal(f.rgb, ref.rgb)


def test_host_scene_without_a_tree_is_built_on_the_device_at_create():
    with_tree = rt.HostScene.standin("bunny", 20010, seed=3)
    without = rt.HostScene.standin("bunny", 20010, seed=3, no_bvh=True)
    assert without.n_nodes == 0 and with_tree.n_nodes > 0
    a, b = rt.Scene(with_tree), rt.Scene(without)
    assert b.bvh_nodes == with_tree.n_nodes and b.bvh_depth == with_tree.bvh_depth
    assert a.build_seconds == 0 and b.build_seconds > 0
    fa = a.render(192, 108, traversal=rt.TRAVERSAL_REFERENCE, want_stats=True)
    fb = b.render(192, 108, traversal=rt.TRAVERSAL_REFERENCE, want_stats=True)
    assert np.array_equal(fa.hit, fb.hit) and np.array_equal(fa.rgb, fb.rgb)
    assert fa.stats.as_dict()["node_visits"] == fb.stats.as_dict()["node_visits"]


def test_loader_with_device_bvh_returns_the_host_builders_arrays():
    path = os.path.join(HERE, "golden", "teapot.obj")
    a = rt.HostScene.load_obj(path, "teapot").arrays()
    b = rt.HostScene.load_obj(path, "teapot", device_bvh=True).arrays()
    assert a["nodes"].tobytes() == b["nodes"].tobytes() and np.array_equal(a["indices"], b["indices"])


@pytest.mark.parametrize("n", [40, 300])
def test_partition_that_leaves_one_side_empty_still_permutes_the_leaf(n):
    """Centroids one ulp apart: the chosen plane rounds back onto the smaller centroid, `centroid < plane` holds for
    no triangle, the reference's partition loop rotates the range by one and the node stays a leaf (bvh.hpp:128-141)."""
    rng = np.random.default_rng(n)
    v = np.zeros((n, 3, 3))
    v[:, :, 0] = 1.0
    v[:, 0, 1], v[:, 1, 1], v[:, 2, 1] = 0.0, 1.0, 0.0
    v[:, 0, 2], v[:, 1, 2], v[:, 2, 2] = 0.0, 0.0, 1.0
    odd = rng.random(n) < 0.5
    v[odd, 2, 0] = 1.0 + 2.0 ** -51          # centroid x = 1 + 2^-52
    host, dev = same_tree(v.reshape(n, 9))
    assert host["nodes"].shape[0] == 1 and not np.array_equal(host["indices"], np.arange(n))


def test_many_identical_triangles_make_a_large_leaf():
    rng = np.random.default_rng(3)
    v = rng.uniform(-1, 1, size=(500, 9))
    v[100:400] = v[100]                      # 300 copies: no axis separates them
    host, _ = same_tree(v)
    assert host["nodes"]["count"].max() >= 300
