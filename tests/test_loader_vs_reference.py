"""The strict OBJ loader against the reference's own parseOBJ (include/scene.hpp:213-253) on seeded odd texts.

oracle/_ref/ref_bvh loads `scenes/teapot.obj` under a given directory through the reference's teapot factory
and prints the triangle count and hashes of the geometry the Triangle constructor left (vertices, normal,
centroid) and of the BVH it built.  The texts stay inside what the reference can read at all (a record whose
numbers do not parse leaves its stream failed and its loop spinning): records may be split over lines, use any
white space, signs, exponents and leading dots, be glued to their letter, be separated by junk characters that
are neither `v` nor `f`, and the file may end in white space (which makes the reference push the last record
twice).  RTX_LOADER_FUZZ_CASES sets how many cases run (default 30).
"""
import os
import random
import subprocess

import numpy as np
import pytest

import ray_tracer_amd as rt
from oracle_binding import ORACLE_DIR, fnv1a

REF_BVH = os.path.join(ORACLE_DIR, "_ref", "ref_bvh")
CASES = int(os.environ.get("RTX_LOADER_FUZZ_CASES", "30"))
FIRST = int(os.environ.get("RTX_LOADER_FUZZ_FIRST", "0"))

pytestmark = pytest.mark.skipif(not os.path.exists(REF_BVH), reason="oracle/_ref is only built where /root/reference exists")


def _number(rnd, x):
    style = rnd.randrange(8)
    if style == 0: return repr(x)
    if style == 1: return "%.6f" % x
    if style == 2: return "%.3e" % x
    if style == 3: return ("+" if x >= 0 else "") + "%.4f" % x
    if style == 4: return "%.12E" % x
    if style == 5: return "%d" % round(x)
    if style == 6: return ("%.5f" % x).replace("0.", ".", 1) if 0 <= x < 1 else "%.5f" % x
    return "%.20f" % x


def _text(seed):
    rnd = random.Random(seed)
    n_v = rnd.randrange(12, 200)
    n_f = rnd.randrange(10, 300)
    gap = lambda: rnd.choice([" ", " ", "  ", "\t", " \t ", "\n", "\r\n", " \n  "])
    junk = lambda: rnd.choice(["", "", "", "#", "# 12 ;;", "g", "o", "s 1", "usemtl", "%%", "3", "-", "."])
    out, defined = [], 0
    pending_faces = n_f
    while defined < n_v or pending_faces > 0:
        # a face may only use vertices that are already defined (the reference indexes them unchecked)
        if defined >= 3 and pending_faces > 0 and (defined >= n_v or rnd.random() < 0.4):
            ids = [rnd.randrange(1, defined + 1) for _ in range(3)]
            letter = "f" + rnd.choice([" ", "", "\t", "\n"])
            out.append(letter + gap().join(("+%d" % i) if rnd.random() < 0.1 else str(i) for i in ids))
            pending_faces -= 1
        else:
            xyz = [rnd.uniform(-5, 5) * rnd.choice([1, 1, 1e-3, 100]) for _ in range(3)]
            letter = "v" + rnd.choice([" ", "", "\t", "\n"])
            nums = [_number(rnd, x) for x in xyz]
            if letter == "v" and nums[0][0] == ".":
                letter = "v "
            out.append(letter + gap().join(nums))
            defined += 1
        out.append(rnd.choice(["\n", "\n", "\r\n", " ", "\n\n"]) + junk() + rnd.choice(["\n", " ", ""]))
    text = "".join(out).rstrip()
    return text + rnd.choice(["", "", "\n", " \n\n", "\t"])            # trailing white space: last record twice


@pytest.mark.parametrize("seed", range(FIRST, FIRST + CASES))
def test_strict_loader_equals_parseobj(seed, tmp_path):
    os.makedirs(tmp_path / "scenes")
    path = tmp_path / "scenes" / "teapot.obj"
    path.write_bytes(_text(seed).encode())
    out = subprocess.run([REF_BVH, "teapot", str(tmp_path), "0", "0", "3", "-", "1"], check=True, capture_output=True,
                         text=True, timeout=60).stdout
    want = dict(line.split(None, 1) for line in out.splitlines() if not line.startswith("["))
    a = rt.HostScene.load_obj(str(path), "teapot").arrays()
    v = a["vertices"]
    assert v.shape[0] == int(want["triangles"]), seed
    centroid = ((v[:, 0:3] + v[:, 3:6]) + v[:, 6:9]) / 3            # triangle.hpp:24
    assert fnv1a(np.concatenate([v, a["normals"], centroid], axis=1)) == want["geometry_fnv"].strip(), seed
    assert a["nodes"].shape[0] == int(want["nodes_used"]), seed
    assert fnv1a(a["nodes"]) == want["nodes_fnv"].strip(), seed
    assert fnv1a(a["indices"]) == want["indices_fnv"].strip(), seed
