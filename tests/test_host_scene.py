"""Host side of the product (OBJ loader, scene presets, BVH builder, stand-in generator) against the
reference-derived fixtures.  No GPU: librtx.so's host entry points only."""
import hashlib
import json
import os

import numpy as np
import pytest

import ray_tracer_amd as rt
from oracle_binding import OracleScene, fnv1a


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "golden.json")))


def _geometry(a, centroids=None):
    return a["vertices"], a["normals"]


def test_teapot_loader_and_builder_match_reference(teapot_obj, golden):
    hs = rt.HostScene.load_obj(teapot_obj, "teapot")
    a = hs.arrays()
    want = golden["teapot_128"]
    assert hs.n_triangles == 6330 and hs.n_nodes == 12541 and hs.bvh_depth == 21
    assert fnv1a(a["nodes"]) == want["nodes_fnv"]
    assert fnv1a(a["indices"]) == want["indices_fnv"]
    ora = OracleScene.load_obj(teapot_obj, "teapot")
    g = ora.geometry()
    assert np.array_equal(g[:, :9], a["vertices"]) and np.array_equal(g[:, 9:12], a["normals"])
    col, spec, refl = ora.materials()
    assert np.array_equal(col, a["colors"]) and np.array_equal(spec, a["specularity"]) and np.array_equal(refl, a["reflectivity"])
    # the factory's material ranges (include/scene.hpp:273-289)
    assert (a["reflectivity"][:-10] == 0.2).all() and (a["reflectivity"][-10:] == 0.4).all()
    assert (a["colors"][-4:-2] == [0, 255, 0]).all() and (a["colors"][-2:] == [255, 0, 0]).all()
    assert list(a["camera_position"]) == [-0.015, 4, -11.99] and list(a["camera_rotation_deg"]) == [0, 0.1, 0]


def test_trailing_newline_quirk(teapot_obj, golden, tmp_path):
    p = tmp_path / "nl.obj"
    p.write_bytes(open(teapot_obj, "rb").read() + b"\n")
    hs = rt.HostScene.load_obj(str(p), "teapot")
    want = golden["teapot_trailing_newline_64"]
    assert hs.n_triangles == 6331
    a = hs.arrays()
    assert fnv1a(a["nodes"]) == want["nodes_fnv"] and fnv1a(a["indices"]) == want["indices_fnv"]
    assert np.array_equal(a["vertices"][-1], a["vertices"][-2])


def test_loader_errors(tmp_path):
    with pytest.raises(rt.RtxError) as e:
        rt.HostScene.load_obj(str(tmp_path / "missing.obj"), "teapot")
    assert "Failed to open file." in str(e.value) and e.value.code == -2
    bad = tmp_path / "slashes.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1/1 2/2/2 3/3/3")
    with pytest.raises(rt.RtxError):
        rt.HostScene.load_obj(str(bad), "teapot")
    few = tmp_path / "few.obj"
    few.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3")
    with pytest.raises(rt.RtxError):
        rt.HostScene.load_obj(str(few), "teapot")      # presets address the last 10 triangles
    oob = tmp_path / "oob.obj"
    oob.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9")
    with pytest.raises(rt.RtxError):
        rt.HostScene.load_obj(str(oob), "teapot")
    with pytest.raises(rt.RtxError):
        rt.HostScene.load_obj(str(few), "no-such-preset")


@pytest.mark.parametrize("scene,ntri,seed,key", [
    ("bunny", 20010, 7, "standin_bunny_20010_seed7_192x108"),
    ("bunny", 144056, 0, "standin_bunny_144056_seed0_480x270"),
    ("dragon", 871316, 0, "standin_dragon_871316_seed0_960x540"),
])
def test_standin_is_deterministic_and_builder_matches_reference(scene, ntri, seed, key, golden, tmp_path):
    obj = str(tmp_path / f"{scene}.obj")
    hs = rt.HostScene.standin(scene, ntri, seed=seed, obj_path_out=obj)
    want = golden[key]
    assert hs.n_triangles == ntri == int(want["triangles"])
    assert hashlib.md5(open(obj, "rb").read()).hexdigest() == want["obj_md5"]
    a = hs.arrays()
    assert not np.isnan(a["normals"]).any()
    assert hs.n_nodes == int(want["nodes_used"])
    assert fnv1a(a["nodes"]) == want["nodes_fnv"]          # the reference built this tree from the same .obj
    assert fnv1a(a["indices"]) == want["indices_fnv"]
    # in-memory stand-in == the same stand-in re-loaded from its file
    again = rt.HostScene.load_obj(obj, scene)
    assert np.array_equal(again.arrays()["vertices"], a["vertices"])


def test_random_soups_builder_matches_oracle():
    from scenes_util import soup

    for seed in range(6):
        v, col, spec, refl, cam = soup(seed, n=50 + 40 * seed)
        hs = rt.HostScene.from_arrays(v, col, spec, refl, **cam)
        ora = OracleScene.from_arrays(v, col, spec, refl, **cam)
        a = hs.arrays()
        assert a["nodes"].tobytes() == ora.nodes().tobytes()
        assert np.array_equal(a["indices"], ora.indices())
        assert hs.bvh_depth == ora.max_depth


def test_caller_supplied_bvh_is_validated():
    from scenes_util import soup

    v, col, spec, refl, cam = soup(3, n=40)
    built = rt.HostScene.from_arrays(v, col, spec, refl, **cam).arrays()
    ok = rt.HostScene.from_arrays(v, col, spec, refl, nodes=built["nodes"], indices=built["indices"], **cam)
    assert ok.arrays()["nodes"].tobytes() == built["nodes"].tobytes()
    broken = built["nodes"].copy()
    inner = np.nonzero(broken["count"] == 0)[0][0]
    broken["first"][inner] = len(broken) + 5            # child out of range
    with pytest.raises(rt.RtxError):
        rt.HostScene.from_arrays(v, col, spec, refl, nodes=broken, indices=built["indices"], **cam)
    loop = built["nodes"].copy()
    deep = np.nonzero(loop["count"] == 0)[0][-1]
    assert deep != 0
    loop["first"][deep] = 1                              # points back up the tree
    with pytest.raises(rt.RtxError):
        rt.HostScene.from_arrays(v, col, spec, refl, nodes=loop, indices=built["indices"], **cam)
    bad_idx = built["indices"].copy()
    bad_idx[0] = 10 ** 6
    with pytest.raises(rt.RtxError):
        rt.HostScene.from_arrays(v, col, spec, refl, nodes=built["nodes"], indices=bad_idx, **cam)


def test_caller_supplied_bvh_with_the_references_uninitialised_tail():
    """The reference allocates 2n-1 nodes and leaves those past nodesUsed uninitialised (bvh.hpp:91): a caller that
    passes the allocated length must get the same scene, whatever the tail holds; leaf ranges that overlap, leaves that
    leave a triangle out and a child index of 2^64-1 are refused."""
    from scenes_util import soup

    v, col, spec, refl, cam = soup(5, n=60)
    built = rt.HostScene.from_arrays(v, col, spec, refl, **cam).arrays()
    nodes, indices = built["nodes"], built["indices"]
    assert len(nodes) < 2 * 60 - 1
    padded = np.zeros(2 * 60 - 1, dtype=nodes.dtype)
    padded[: len(nodes)] = nodes
    padded["first"][len(nodes):] = 0xDEADBEEFDEADBEEF                  # garbage where the reference has garbage
    padded["count"][len(nodes):] = 0xFEEDFACE
    padded["bmin"][len(nodes):] = np.nan
    hs = rt.HostScene.from_arrays(v, col, spec, refl, nodes=padded, indices=indices, **cam)
    got = hs.arrays()
    assert got["nodes"].tobytes() == nodes.tobytes() and np.array_equal(got["indices"], indices)
    leaves = np.nonzero(nodes["count"] > 0)[0]
    overlap = nodes.copy()
    a, b = leaves[0], leaves[1]
    overlap["first"][b] = overlap["first"][a]                          # two leaves over the same positions
    overlap["count"][b] = overlap["count"][a]
    with pytest.raises(rt.RtxError) as e:
        rt.HostScene.from_arrays(v, col, spec, refl, nodes=overlap, indices=indices, **cam)
    assert "overlap" in str(e.value) or "cover" in str(e.value)
    wide = nodes.copy()
    big = leaves[np.argmax(nodes["count"][leaves])]
    if wide["count"][big] > 1:
        wide["count"][big] -= 1                                        # one position in no leaf
        with pytest.raises(rt.RtxError):
            rt.HostScene.from_arrays(v, col, spec, refl, nodes=wide, indices=indices, **cam)
    wrap = nodes.copy()
    inner = np.nonzero(wrap["count"] == 0)[0][0]
    wrap["first"][inner] = 0xFFFFFFFFFFFFFFFF                          # first + 1 wraps to 0
    with pytest.raises(rt.RtxError):
        rt.HostScene.from_arrays(v, col, spec, refl, nodes=wrap, indices=indices, **cam)


def test_caller_supplied_bvh_numbered_from_two_and_leaf_order():
    """A tree whose left children sit at even indices is accepted as it is; one whose leaf ranges do not ascend in
    depth-first order is refused, because equal-t ties between leaves are settled by that order (bvh.hpp:204-208)."""
    from scenes_util import renumber_with_gap, soup

    v, col, spec, refl, cam = soup(7, n=80)
    built = rt.HostScene.from_arrays(v, col, spec, refl, **cam).arrays()
    nodes, indices = built["nodes"], built["indices"]
    moved = renumber_with_gap(nodes)
    hs = rt.HostScene.from_arrays(v, col, spec, refl, nodes=moved, indices=indices, **cam)
    got = hs.arrays()["nodes"]
    assert len(got) == len(nodes) + 1 and got[2:].tobytes() == moved[2:].tobytes()
    # swap the two children of the root: the same set of leaves, the right subtree's ranges now come first
    root = nodes[0]
    assert root["count"] == 0
    swapped = nodes.copy()
    l = int(root["first"])
    swapped[[l, l + 1]] = swapped[[l + 1, l]]
    with pytest.raises(rt.RtxError) as e:
        rt.HostScene.from_arrays(v, col, spec, refl, nodes=swapped, indices=indices, **cam)
    assert "depth-first" in str(e.value)


def test_empty_scene_is_an_error():
    with pytest.raises(rt.RtxError):
        rt.HostScene.from_arrays(np.zeros((0, 9)), np.zeros((0, 3)), np.zeros(0), np.zeros(0), camera_position=(0, 0, 0))


def test_save_ppm_format(tmp_path):
    rgb = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    p = tmp_path / "c.ppm"
    rt.save_ppm(str(p), rgb)
    assert p.read_bytes() == b"P6\n7 5 255\n" + rgb.tobytes()       # src/main.cpp:103


def test_tolerant_loader_reads_what_denormalize_plus_the_reference_loader_read(tmp_path):
    """A general OBJ (vt/vn groups, negative indices, comments, CRLF, a polygon, trailing newlines)
    against the same mesh rewritten the way tools/denormalize.py:1-36 does and read by the strict loader."""
    rng = np.random.default_rng(5)
    nv, nf = 40, 60
    v = rng.normal(size=(nv, 3)).round(4)
    faces = rng.integers(1, nv + 1, size=(nf, 3))
    lines = ["# a comment", "mtllib x.mtl", "o thing"]
    lines += [f"v {a:.4f} {b:.4f} {c:.4f}" for a, b, c in v]
    lines += ["vn 0 0 1", "vt 0.5 0.5", "s off", "usemtl m"]
    plain = []
    for k, (a, b, c) in enumerate(faces):
        style = k % 5
        if style == 0: lines.append(f"f {a} {b} {c}")
        elif style == 1: lines.append(f"f {a}/1 {b}/1 {c}/1")
        elif style == 2: lines.append(f"f {a}//1 {b}//1 {c}//1")
        elif style == 3: lines.append(f"f {a}/1/1 {b}/1/1 {c}/1/1 {a}/1/1")           # polygon: first three count
        else: lines.append(f"f {a - nv - 9} {b - nv - 9} {c - nv - 9}")    # negative = from the end of ALL nv + 8 vertices
        plain.append(f"f {a} {b} {c}")
    box = [f"v {x} {y} {z}" for x in (-9, 9) for y in (-9, 9) for z in (-9, 9)]
    lines += box
    boxf = [(nv + 1, nv + 2, nv + 3)] * 10
    lines += [f"f {a} {b} {c}" for a, b, c in boxf]
    general = tmp_path / "general.obj"
    general.write_bytes(("\r\n".join(lines) + "\r\n\r\n").encode())
    strict = tmp_path / "strict.obj"
    strict.write_text("\n".join([f"v {a:.4f} {b:.4f} {c:.4f}" for a, b, c in v] + box + plain +
                                [f"f {a} {b} {c}" for a, b, c in boxf]))          # no trailing newline
    a = rt.HostScene.load_obj(str(general), "teapot", tolerant=True).arrays()
    b = rt.HostScene.load_obj(str(strict), "teapot").arrays()
    assert a["vertices"].shape == (nf + 10, 9)
    assert a["vertices"].tobytes() == b["vertices"].tobytes()
    assert a["nodes"].tobytes() == b["nodes"].tobytes()


def test_tolerant_loader_on_the_teapot_equals_the_strict_one():
    path = os.path.join(os.path.dirname(__file__), "golden", "teapot.obj")
    a = rt.HostScene.load_obj(path, "teapot", tolerant=True).arrays()
    b = rt.HostScene.load_obj(path, "teapot").arrays()
    assert a["vertices"].shape[0] == 6330
    assert a["vertices"].tobytes() == b["vertices"].tobytes() and np.array_equal(a["indices"], b["indices"])


def test_tolerant_loader_errors(tmp_path):
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(rt.RtxError, match="does not exist"):
        rt.HostScene.load_obj(str(bad), "teapot", tolerant=True)
    with pytest.raises(rt.RtxError, match="Failed to open file"):
        rt.HostScene.load_obj(str(tmp_path / "missing.obj"), "teapot", tolerant=True)


def test_number_literals_read_like_strtod(tmp_path):
    """Scene::parseOBJ reads coordinates with `fs >> double` (scene.hpp:229-231): correctly rounded, as
    Python's float().  Short literals take the loaders' own conversion, long or extreme ones strtod."""
    import random
    rnd = random.Random(5)

    def literal():
        k = rnd.random()
        if k < 0.30: return "%.6f" % rnd.uniform(-100, 100)
        if k < 0.40: return repr(rnd.uniform(-1, 1))
        if k < 0.50: return "%d" % rnd.randint(-1000, 1000)
        if k < 0.60: return "%.3e" % rnd.uniform(-1e5, 1e5)
        if k < 0.65: return "%.17g" % rnd.uniform(-1e-3, 1e-3)
        if k < 0.70: return "%dE%+d" % (rnd.randint(0, 10 ** 15), rnd.randint(-30, 30))
        if k < 0.75: return "." + str(rnd.randint(0, 10 ** 9))
        if k < 0.80: return str(rnd.randint(0, 999)) + "."
        if k < 0.85: return "%.25f" % rnd.uniform(-3, 3)
        if k < 0.90: return "-0.0"
        if k < 0.95: return "+%d.%020d" % (rnd.randint(0, 99), rnd.randint(0, 10 ** 20 - 1))
        return "0.%s" % ("0" * rnd.randint(0, 30) + str(rnd.randint(1, 10 ** 6)))

    n = 6000
    lits = [[literal() for _ in range(3)] for _ in range(n)]
    path = tmp_path / "literals.obj"
    path.write_text("".join("v %s %s %s\n" % tuple(a) for a in lits) +
                    "\n".join("f %d %d %d" % (i + 1, i + 2, i + 3) for i in range(0, n - 2, 3)))
    want = np.array([float(x) for i in range(0, n - 2, 3) for k in range(3) for x in lits[i + k]])
    for tolerant in (False, True):
        got = rt.HostScene.load_obj(str(path), "teapot", tolerant=tolerant, no_bvh=True).arrays()["vertices"]
        assert got.size == want.size
        assert got.tobytes() == want.tobytes()        # bit for bit, signed zeros included


def test_file_sizes_around_a_page_boundary_load_alike(tmp_path):
    """The loaders map the file when its last page has room for a terminator and read it otherwise."""
    body = "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\n" + "\n".join("f 1 2 3" for _ in range(9)) + "\nf 2 3 4"
    want = None
    for size in (4095, 4096, 4097, 8192):
        path = tmp_path / f"padded_{size}.obj"
        path.write_text("#" * (size - len(body) - 1) + "\n" + body)
        assert os.path.getsize(path) == size
        for tolerant in (False, True):
            got = rt.HostScene.load_obj(str(path), "teapot", tolerant=tolerant, no_bvh=True).arrays()["vertices"]
            want = got if want is None else want
            assert got.shape == (10, 9) and got.tobytes() == want.tobytes()


DENORMALIZE = "/root/reference/tools/denormalize.py"


@pytest.mark.skipif(not os.path.exists(DENORMALIZE), reason="the reference's tools/ only exist in the build container")
def test_tolerant_loader_equals_the_reference_script_followed_by_the_strict_loader(tmp_path):
    """tools/denormalize.py itself (it reads scenes/bunny-o.obj and writes scenes/bunny.obj under the current directory)
    on a mesh with vt/vn index groups and negative indices, then the strict loader on its output, against the tolerant
    loader on the original.  The script's output ends in a newline, so the strict loader -- like the reference's -- reads
    its last face twice; the tolerant loader reads every face once."""
    import subprocess
    import sys
    rng = np.random.default_rng(11)
    nv, nf = 50, 80
    v = rng.normal(size=(nv, 3))
    lines = ["# mesh", "o thing"] + ["v %r %r %r" % tuple(float(x) for x in p) for p in v] + ["vn 0 0 1", "vt 0.5 0.5"]
    for k in range(nf):
        a, b, c = (int(x) for x in rng.integers(1, nv + 1, size=3))
        if k % 3 == 0: lines.append(f"f {a}/1 {b}/1 {c}/1")
        elif k % 3 == 1: lines.append(f"f {a}//1 {b}//1 {c}//1")
        else: lines.append(f"f {a - nv - 1}/1/1 {b - nv - 1}/1/1 {c - nv - 1}/1/1")
    os.makedirs(tmp_path / "scenes")
    (tmp_path / "scenes" / "bunny-o.obj").write_text("\n".join(lines) + "\n")
    subprocess.run([sys.executable, DENORMALIZE], cwd=tmp_path, check=True)
    strict = rt.HostScene.load_obj(str(tmp_path / "scenes" / "bunny.obj"), "teapot", no_bvh=True).arrays()["vertices"]
    tolerant = rt.HostScene.load_obj(str(tmp_path / "scenes" / "bunny-o.obj"), "teapot", tolerant=True, no_bvh=True).arrays()["vertices"]
    assert tolerant.shape == (nf, 9) and strict.shape == (nf + 1, 9)
    assert strict[:nf].tobytes() == tolerant.tobytes() and strict[nf].tobytes() == strict[nf - 1].tobytes()
    # RTX_LOAD_LAST_FACE_TWICE: the tolerant loader on the original yields the pipeline's scene, materials included (the
    # presets address the last ten triangles from the END of the list, so one triangle more shifts them)
    a = rt.HostScene.load_obj(str(tmp_path / "scenes" / "bunny.obj"), "bunny").arrays()
    b = rt.HostScene.load_obj(str(tmp_path / "scenes" / "bunny-o.obj"), "bunny", tolerant=True, last_face_twice=True).arrays()
    for key in ("vertices", "normals", "colors", "specularity", "reflectivity", "nodes", "indices"):
        assert a[key].tobytes() == b[key].tobytes(), key
    c = rt.HostScene.load_obj(str(tmp_path / "scenes" / "bunny-o.obj"), "bunny", tolerant=True).arrays()
    assert not np.array_equal(c["reflectivity"], a["reflectivity"][:nf])          # without the flag the overrides sit one triangle earlier


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (300, 217), (1920, 37)])
def test_png_writer_round_trips(tmp_path, w, h):
    """Decoded with zlib and struct only: signature, IHDR, one IDAT of stored blocks, IEND, every CRC."""
    import struct, zlib
    rng = np.random.default_rng(w * 1000 + h)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    path = tmp_path / "x.png"
    rt.save_png(str(path), img)
    data = path.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(kind + body)
        chunks.append((kind, body))
        pos += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    assert np.array_equal(raw[:, 1:].reshape(h, w, 3), img)
