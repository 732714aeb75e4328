"""The library's host-only code under AddressSanitizer and UndefinedBehaviorSanitizer.

`make -C ray_tracer_amd/csrc asan` builds ray_tracer_amd/host_check_asan: host_scene.cpp and scenegen.cpp (OBJ loaders that scan
mmapped text with hand-written pointer code, presets, stand-in generators, the scalar BVH builder, BVH validation) behind a
small command-line checker, instrumented with -fsanitize=address,undefined -fno-sanitize-recover=all.  Every run below must end
without a sanitizer report, and what it prints must equal what the shipped (uninstrumented) library computes.
"""
import os
import subprocess

import numpy as np
import pytest

import ray_tracer_amd as rt
from oracle_binding import fnv1a
from test_loader_vs_reference import _text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "ray_tracer_amd", "host_check_asan")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ray_tracer_amd", "csrc"), "asan"], stdout=subprocess.DEVNULL)


def run(*args, expect=0):
    r = subprocess.run([CHECK, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == expect, (r.returncode, r.stderr[-500:])
    return dict(line.split(None, 1) for line in r.stdout.splitlines())


def same_as_library(out, host):
    a = host.arrays()
    assert int(out["triangles"]) == a["vertices"].shape[0]
    assert out["vertices_fnv"] == fnv1a(a["vertices"]) and out["normals_fnv"] == fnv1a(a["normals"])
    assert int(out["nodes"]) == len(a["nodes"]) and out["nodes_fnv"] == fnv1a(a["nodes"]) and out["indices_fnv"] == fnv1a(a["indices"])
    assert int(out["depth"]) == host.bvh_depth


def test_teapot_both_loaders(teapot_obj):
    same_as_library(run("parse", "strict", teapot_obj, "teapot"), rt.HostScene.load_obj(teapot_obj, "teapot"))
    same_as_library(run("parse", "tolerant", teapot_obj, "teapot"), rt.HostScene.load_obj(teapot_obj, "teapot", tolerant=True))


@pytest.mark.parametrize("seed", range(12))
def test_odd_texts_in_the_references_dialect(seed, tmp_path):
    path = tmp_path / "odd.obj"
    path.write_bytes(_text(seed).encode())
    same_as_library(run("parse", "strict", path, "teapot"), rt.HostScene.load_obj(str(path), "teapot"))


def test_general_obj_text(tmp_path):
    path = tmp_path / "general.obj"
    lines = ["# a comment", "o thing", "vn 0 0 1", "vt 0.5 0.5"]
    rnd = np.random.default_rng(3)
    pts = rnd.uniform(-3, 3, size=(40, 3))
    lines += ["v %r %r %r" % tuple(float(c) for c in p) for p in pts]
    for k in range(60):
        i, j, l = rnd.choice(40, size=3, replace=False) + 1
        form = k % 4
        lines.append(["f %d %d %d" % (i, j, l), "f %d/1 %d/1 %d/1" % (i, j, l), "f %d//1 %d//1 %d//1" % (i, j, l),
                      "f %d/1/1 %d/1/1 %d/1/1" % (i - 41, j - 41, l - 41)][form])        # the last form: negative indices
    lines += ["s off", "usemtl none"]
    path.write_text("\n".join(lines) + "\n")
    same_as_library(run("parse", "tolerant", path, "teapot"), rt.HostScene.load_obj(str(path), "teapot", tolerant=True))


@pytest.mark.parametrize("text", [b"", b"v", b"v 1 2", b"f 1 2 3", b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9", b"v 0 0 0\nf 1 2",
                                  b"f -1 -2 -3", b"v 1e999 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3", b"\x00\xff\xfe garbage \x80", b"v " * 5000,
                                  b"f 1/2/3/4/5 6 7", b"v 0 0 0 v 1 0 0 v 0 1 0 f 1 2 3 f 3 2"])
def test_broken_files_end_in_an_error_or_a_scene_but_never_in_a_report(text, tmp_path):
    path = tmp_path / "broken.obj"
    path.write_bytes(text)
    for loader in ("strict", "tolerant"):
        r = subprocess.run([CHECK, "parse", loader, str(path), "teapot"], capture_output=True, text=True, timeout=60)
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        assert r.returncode in (0, 2, 3, 4), (loader, r.returncode, r.stderr[-300:])


@pytest.mark.parametrize("family", ["blob", "knot"])
def test_standin_generators(family):
    out = run("standin", "bunny", 20010, 7, family)
    same_as_library(out, rt.HostScene.standin("bunny", 20010, seed=7, family=family))
    run("standin", "dragon", 30, 0, family, expect=2 if family == "knot" else 0)        # the smallest sizes (knot: 38)
    run("standin", "erato", 38, 1, family)


def test_bvh_validation_on_garbage(teapot_obj):
    """The reference's 2n-1 allocation with an uninitialised tail is accepted and cut back; 800 broken trees (child
    indices of 2^64-1, past the end, pointing at the root or at themselves; leaf ranges past the end or overlapping) are
    all refused, none read out of bounds."""
    out = run("poisoned", teapot_obj, "teapot")
    assert out["padded"].split() == ["0", "nodes", "12541", "same", "1"]
    cases, refused = int(out["broken"].split()[0]), int(out["broken"].split()[2])
    assert cases >= 800 and refused == cases
