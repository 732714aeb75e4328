"""Pins the oracle against the reference ITSELF on seeded odd scenes.

oracle/_ref/ref_bvh is the reference's own headers behind a small harness (oracle/ref_harness.cpp).  For every
case a scene is written in the reference's OBJ dialect where its teapot factory looks for it, the camera, lights
and materials go in through the harness' REF_SCENE_FILE directives (public members of Scene), and what the
reference built and rendered -- BVH nodes, indices, primary ids, primary t, unclamped RGB -- must equal the
oracle's arrays bit for bit.  Unlike the stock factories this covers point lights, several directional lights,
specular exponents other than 1 (same glibc pow on both sides), cameras turned about every axis and inside the
geometry, recursion depths 0..5 and degenerate geometry.

RTX_REF_FUZZ_CASES sets how many cases run (default 24); RTX_REF_FUZZ_FIRST the first seed.
"""
import os
import subprocess

import numpy as np
import pytest

from oracle_binding import NODE_DT, ORACLE_DIR, OracleScene
from scenes_util import fuzz_case

REF_BVH = os.path.join(ORACLE_DIR, "_ref", "ref_bvh")
CASES = int(os.environ.get("RTX_REF_FUZZ_CASES", "24"))
FIRST = int(os.environ.get("RTX_REF_FUZZ_FIRST", "0"))

pytestmark = pytest.mark.skipif(not os.path.exists(REF_BVH), reason="oracle/_ref is only built where /root/reference exists")


def _write_scene(root, v, colors, spec, refl, kw):
    os.makedirs(os.path.join(root, "scenes"), exist_ok=True)
    n = v.shape[0]
    lines = []
    for tri in v:
        lines += ["v %r %r %r" % tuple(float(x) for x in tri[3 * k:3 * k + 3]) for k in range(3)]
    lines += ["f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3) for i in range(n)]
    with open(os.path.join(root, "scenes", "teapot.obj"), "w") as f:
        f.write("\n".join(lines))                       # no trailing newline: it would duplicate the last face
    rec = np.zeros(n, dtype=np.dtype([("rgb", "<i4", 3), ("spec", "<i4"), ("refl", "<f8")]))
    rec["rgb"], rec["spec"], rec["refl"] = colors, spec, refl
    rec.tofile(os.path.join(root, "materials.bin"))
    d = ["camera " + " ".join(repr(float(x)) for x in (*kw["camera_position"], *kw["camera_rotation_deg"])),
         "viewport " + " ".join(repr(float(x)) for x in kw["viewport"]),
         "background %d %d %d" % tuple(kw["background"]),
         "ambient %r" % float(kw["ambient"]), "nolights"]
    d += ["point %r %r %r %r" % (float(i), *[float(x) for x in p]) for i, p in kw["point_lights"]]
    d += ["directional %r %r %r %r" % (float(i), *[float(x) for x in p]) for i, p in kw["directional_lights"]]
    d += ["materials " + os.path.join(root, "materials.bin")]
    with open(os.path.join(root, "scene.txt"), "w") as f:
        f.write("\n".join(d) + "\n")


@pytest.mark.parametrize("seed", range(FIRST, FIRST + CASES))
def test_oracle_equals_reference(seed, tmp_path):
    family, (v, colors, spec, refl, kw), o = fuzz_case(seed, exotic_spec=True)
    if v.shape[0] < 10:                                  # the factories address the last ten triangles (scene.hpp:273-289)
        reps = -(-10 // v.shape[0])
        v, colors, spec, refl = (np.concatenate([a] * reps) for a in (v, colors, spec, refl))
    W, H, depth = o["W"], o["H"], o["depth"]
    tag = f"seed {seed} {family} n={v.shape[0]} {W}x{H} depth {depth}"
    root = str(tmp_path)
    _write_scene(root, v, colors, spec, refl, kw)
    prefix = os.path.join(root, "out")
    env = dict(os.environ, REF_SCENE_FILE=os.path.join(root, "scene.txt"))
    subprocess.run([REF_BVH, "teapot", root, str(W), str(H), str(depth), prefix, "2"], check=True, capture_output=True, env=env)
    ora = OracleScene.from_arrays(v, colors, spec, refl, **kw)
    try:
        assert np.fromfile(prefix + ".nodes.bin", dtype=NODE_DT).tobytes() == ora.nodes().tobytes(), tag
        assert np.array_equal(np.fromfile(prefix + ".indices.u64", dtype=np.uint64), ora.indices()), tag
        got = ora.render(W, H, depth=depth, want_t=True)
        assert np.array_equal(np.fromfile(prefix + ".hit.i32", dtype=np.int32).reshape(H, W), got["hit"]), tag
        assert np.fromfile(prefix + ".t.f64", dtype=np.float64).tobytes() == got["t"].tobytes(), tag
        assert np.array_equal(np.fromfile(prefix + ".rgb.i32", dtype=np.int32).reshape(H, W, 3), got["rgb_i32"]), tag
        assert np.array_equal(np.fromfile(prefix + ".rgb.u8", dtype=np.uint8).reshape(H, W, 3), got["rgb_u8"]), tag
    finally:
        ora.close()
