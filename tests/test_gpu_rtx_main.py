"""ray_tracer_amd/rtx_main, the reference's main() (src/main.cpp:24-73) on the GPU: console lines, output file, errors."""
import hashlib
import json
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "ray_tracer_amd", "rtx_main")


@pytest.mark.parametrize("args", [[], ["--mode", "pruned"], ["--builder", "host"], ["--overlap"]])
def test_stock_frame_is_the_references_file(teapot_obj, golden_dir, tmp_path, args):
    """Run the way the reference is run -- from a directory holding scenes/teapot.obj, writing out/canvas.ppm, no
    arguments needed: 1920x1920, depth 3 -- the file has the md5 of the reference's own canvas.ppm."""
    os.makedirs(tmp_path / "scenes")
    os.makedirs(tmp_path / "out")
    shutil.copyfile(teapot_obj, tmp_path / "scenes" / "teapot.obj")
    run = subprocess.run([MAIN] + args, cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    lines = run.stdout.splitlines()
    assert lines[0] == "[ TRI ] 6330 triangles."                                   # scene.hpp:248
    assert re.fullmatch(r"\[ BVH \] [0-9.e-]+ seconds\.", lines[1])                # scene.hpp:49-51
    assert re.fullmatch(r"(\d+\.\d{3,4} ){3}", lines[2])                           # src/main.cpp:40,57,61: three frames
    want = json.load(open(os.path.join(golden_dir, "golden.json")))["teapot_1920"]["ppm_md5"]
    assert hashlib.md5(open(tmp_path / "out" / "canvas.ppm", "rb").read()).hexdigest() == want


def test_missing_scene_file_ends_like_the_reference(tmp_path):
    run = subprocess.run([MAIN], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert run.returncode == 255 and run.stderr == "Failed to open file.\n"        # scene.hpp:216-218, src/main.cpp:27-32


def test_unwritable_output_ends_like_the_reference(teapot_obj, tmp_path):
    os.makedirs(tmp_path / "scenes")
    shutil.copyfile(teapot_obj, tmp_path / "scenes" / "teapot.obj")                # no out/ directory: src/main.cpp:98-101
    run = subprocess.run([MAIN, "--width", "64", "--height", "64"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert run.returncode == 255 and run.stderr == "Failed to open file.\n"


LIKE = os.path.join(ROOT, "ray_tracer_amd", "main_like_reference")


@pytest.mark.parametrize("args", [[], ["--whole-frame"], ["--threads"]])
def test_driver_written_against_the_reference_shaped_header(teapot_obj, golden_dir, tmp_path, args):
    """main_like_reference.cpp uses include/rtx_reference_api.hpp -- Scene::teapot(), render(scene, xMin, xMax) over
    eight column strips (or one call for the whole width), saveCanvas() -- and must leave the reference's canvas.ppm."""
    os.makedirs(tmp_path / "scenes")
    os.makedirs(tmp_path / "out")
    shutil.copyfile(teapot_obj, tmp_path / "scenes" / "teapot.obj")
    run = subprocess.run([LIKE] + args, cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    lines = run.stdout.splitlines()
    assert lines[0] == "[ TRI ] 6330 triangles." and lines[1].startswith("[ BVH ] ")
    assert re.fullmatch(r"(\d+\.\d{3} ){3}", lines[2])
    want = json.load(open(os.path.join(golden_dir, "golden.json")))["teapot_1920"]["ppm_md5"]
    assert hashlib.md5(open(tmp_path / "out" / "canvas.ppm", "rb").read()).hexdigest() == want


def test_reference_shaped_driver_fails_like_the_reference(tmp_path):
    run = subprocess.run([LIKE], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert run.returncode == 255 and run.stderr == "Failed to open file.\n"
