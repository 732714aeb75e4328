"""The C ABI: librtx.so loads without a GPU and exports every function include/rtx.h declares."""
import ctypes
import os
import re

import ray_tracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(rt.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"librtx.so does not export {n}"
    assert sorted(rt.EXPORTS) == names


def test_struct_sizes_match_the_header():
    assert ctypes.sizeof(rt.BvhNode) == 64          # the reference's BVHNode
    assert ctypes.sizeof(rt.Light) == 32
    assert ctypes.sizeof(rt.RenderParams) == 64
    assert ctypes.sizeof(rt.Stats) == 128


def test_host_only_calls_work_without_gpu():
    lib = rt.lib()
    assert lib.rtx_version().startswith(b"rtx-mi355x")
    p = rt.make_params(100, 60, tile=(16, 8), tile_rank=1, tile_count=4, layout=rt.LAYOUT_PACKED_TILES)
    # 7 x 8 = 56 tiles over 4 ranks -> 14 each
    assert lib.rtx_output_pixels(ctypes.byref(p)) == 14 * 16 * 8
    p = rt.make_params(100, 60)
    assert lib.rtx_output_pixels(ctypes.byref(p)) == 6000
    p.depth = 99
    assert lib.rtx_output_pixels(ctypes.byref(p)) == 0
    assert b"depth" in lib.rtx_last_error()


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback stands behind the binding: without librtx.so every use raises."""
    import subprocess
    import sys
    code = "import ray_tracer_amd as rt\ntry:\n    rt.lib()\nexcept rt.RtxError as e:\n    print('raised', e)\n"
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RTX_LIB=str(tmp_path / "absent.so"), PYTHONPATH=ROOT),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "raised" in out.stdout and "is missing" in out.stdout


def test_render_without_a_gpu_is_an_error_not_a_fallback(teapot_obj):
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("this machine has a GPU")
    host = rt.HostScene.load_obj(teapot_obj, "teapot")          # loading and the host builder need no GPU
    with pytest.raises(rt.RtxError):
        rt.Scene(host)                                          # the device scene does
    with pytest.raises(rt.RtxError):
        rt.build_bvh(host.arrays()["vertices"], builder=rt.BUILDER_DEVICE)


def test_scene_group_without_a_gpu_is_an_error(teapot_obj):
    """The multi-GPU entry points fail with a status and a message where there is no GPU (or no RCCL): nothing crashes."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("this machine has a GPU")
    host = rt.HostScene.load_obj(teapot_obj, "teapot")
    with pytest.raises(rt.RtxError) as e:
        rt.SceneGroup(host, n_devices=0)
    assert e.value.code in (-3, -4) and str(e.value)
    lib = rt.lib()
    assert lib.rtx_scene_group_size(None) == 0 and lib.rtx_scene_group_device(None, 0) == -1
    lib.rtx_scene_group_destroy(None)


def test_header_is_plain_c(tmp_path):
    """include/rtx.h is what a C, cgo or FFI binding includes: it must compile as C99 on its own."""
    import subprocess
    src = tmp_path / "use_rtx.c"
    src.write_text('#include "rtx.h"\nint main(void) { rtx_render_params p; rtx_render_params_default(&p, 4, 4); return p.depth != 3; }\n')
    exe = tmp_path / "use_rtx"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", os.path.dirname(rt.LIB_PATH), "-lrtx", "-Wl,-rpath," + os.path.dirname(rt.LIB_PATH),
                    "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
