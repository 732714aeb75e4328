"""ctypes binding of librtx.so (the C ABI in include/rtx.h).

Mirrors what the reference's main() does around its render threads
(src/main.cpp:24-73): load a scene (Scene::teapot() etc., include/scene.hpp:255-448),
render frames, save a PPM.  There is no CPU fallback: if librtx.so is missing or a
call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTX_LIB selects another build of the library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("RTX_LIB") or os.path.join(_HERE, "librtx.so")

TRAVERSAL_REFERENCE = 0
TRAVERSAL_PRUNED = 1
LAYOUT_ROW_MAJOR = 0
LAYOUT_PACKED_TILES = 1

# face counts of the reference's scenes (include/scene.hpp:255,294,333,372,411)
SCENE_TRIANGLES = {"teapot": 6330, "bunny": 144056, "erato": 412508, "dragon": 871316, "aurelius": 1704768}


class RtxError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rtx error {code}: {message}")
        self.code = code


class BvhNode(C.Structure):
    _fields_ = [("bmin", C.c_double * 3), ("bmax", C.c_double * 3), ("first", C.c_uint64), ("count", C.c_uint64)]


class Light(C.Structure):
    _fields_ = [("intensity", C.c_double), ("v", C.c_double * 3)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_triangles", C.c_uint64),
        ("vertices", C.POINTER(C.c_double)),
        ("normals", C.POINTER(C.c_double)),
        ("colors", C.POINTER(C.c_int32)),
        ("specularity", C.POINTER(C.c_int32)),
        ("reflectivity", C.POINTER(C.c_double)),
        ("n_nodes", C.c_uint64),
        ("nodes", C.POINTER(BvhNode)),
        ("indices", C.POINTER(C.c_uint64)),
        ("viewport", C.c_double * 3),
        ("camera_position", C.c_double * 3),
        ("camera_rotation_deg", C.c_double * 3),
        ("background", C.c_int32 * 3),
        ("ambient", C.c_double),
        ("n_point_lights", C.c_uint32),
        ("point_lights", C.POINTER(Light)),
        ("n_directional_lights", C.c_uint32),
        ("directional_lights", C.POINTER(Light)),
        ("device", C.c_int32),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("x0", C.c_int32), ("x1", C.c_int32), ("y0", C.c_int32), ("y1", C.c_int32),
        ("depth", C.c_int32), ("spp", C.c_int32), ("seed", C.c_uint64),
        ("traversal", C.c_int32),
        ("tile_w", C.c_int32), ("tile_h", C.c_int32), ("tile_rank", C.c_int32), ("tile_count", C.c_int32),
        ("layout", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "primary_rays", "reflection_rays", "shadow_rays", "shadow_occluded", "closest_hits",
        "node_visits", "box_hits", "leaf_visits", "triangle_tests")] + [
        ("rounds", C.c_uint64 * 3), ("lane_steps", C.c_uint64 * 3), ("kernel_ms", C.c_double)]

    @property
    def rays(self) -> int:
        return self.primary_rays + self.reflection_rays + self.shadow_rays

    def as_dict(self) -> dict:
        d = {n: getattr(self, n) for n, _ in self._fields_ if n not in ("rounds", "lane_steps")}
        d["rays"] = self.rays
        return d

    def scheduling(self) -> dict:
        """Rounds per step kind and the lane utilisation of each."""
        names = ("inner", "leaf", "shade")
        return {n: {"rounds": int(self.rounds[i]), "lane_util": (self.lane_steps[i] / (64.0 * self.rounds[i])) if self.rounds[i] else 0.0}
                for i, n in enumerate(names)}


# every symbol include/rtx.h declares
EXPORTS = (
    "rtx_host_scene_load_obj", "rtx_host_scene_standin", "rtx_host_scene_from_desc", "rtx_host_scene_desc",
    "rtx_host_scene_bvh_depth", "rtx_host_scene_set_camera", "rtx_host_scene_set_lights", "rtx_host_scene_free",
    "rtx_scene_create", "rtx_scene_create_from_host", "rtx_scene_destroy", "rtx_scene_device_bytes",
    "rtx_render_params_default", "rtx_output_pixels", "rtx_render", "rtx_render_device", "rtx_render_device_timed", "rtx_render_device_staged",
    "rtx_untile",
    "rtx_save_ppm", "rtx_save_png", "rtx_selftest_division", "rtx_selftest_pow", "rtx_selftest_sqrt", "rtx_last_error", "rtx_version",
    "rtx_host_scene_load_obj_ex", "rtx_host_scene_standin_ex",
    "rtx_bvh_build", "rtx_host_scene_rebuild_bvh", "rtx_scene_create_with_builder", "rtx_scene_build_seconds",
    "rtx_scene_bvh_nodes", "rtx_scene_bvh_depth", "rtx_render_device_noalloc", "rtx_render_begin", "rtx_render_end",
    "rtx_scene_group_create", "rtx_scene_group_create_from_host", "rtx_scene_group_destroy", "rtx_scene_group_size",
    "rtx_scene_group_device", "rtx_scene_group_scene", "rtx_render_sharded", "rtx_render_sharded_device", "rtx_scene_group_sync",
    "rtx_scene_group_set_gather", "rtx_gather_plan", "rtx_render_device_batch",
)

_lib = None


def lib() -> C.CDLL:
    """Loads librtx.so (built in-tree by ray_tracer_amd/csrc/Makefile); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtxError(-3, f"{LIB_PATH} is missing: build it with `make -C ray_tracer_amd/csrc`")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.  If it is loaded
    # AFTER librtx.so has pulled in /opt/rocm's copy, the process holds two runtimes and the
    # second one finds no GPU.  Importing torch first makes librtx.so bind to torch's copy
    # (same soname), so device pointers and streams can be shared with torch tensors.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rtx_last_error.restype = C.c_char_p
    L.rtx_version.restype = C.c_char_p
    L.rtx_host_scene_load_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rtx_host_scene_standin.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rtx_host_scene_load_obj_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_int32, C.POINTER(vp)]
    L.rtx_host_scene_standin_ex.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint32, C.c_int32, C.POINTER(vp)]
    L.rtx_host_scene_from_desc.argtypes = [C.POINTER(SceneDesc), C.POINTER(vp)]
    L.rtx_host_scene_desc.argtypes = [vp, C.POINTER(SceneDesc)]
    L.rtx_host_scene_bvh_depth.argtypes = [vp]
    L.rtx_host_scene_bvh_depth.restype = C.c_int32
    L.rtx_host_scene_set_camera.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rtx_host_scene_set_lights.argtypes = [vp, C.c_double, C.c_uint32, C.POINTER(Light), C.c_uint32, C.POINTER(Light)]
    L.rtx_host_scene_free.argtypes = [vp]
    L.rtx_host_scene_free.restype = None
    L.rtx_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(vp)]
    L.rtx_scene_create_with_builder.argtypes = [C.POINTER(SceneDesc), C.c_int32, C.POINTER(vp)]
    L.rtx_scene_build_seconds.argtypes = [vp]
    L.rtx_scene_build_seconds.restype = C.c_double
    L.rtx_scene_bvh_nodes.argtypes = [vp]
    L.rtx_scene_bvh_nodes.restype = C.c_uint64
    L.rtx_scene_bvh_depth.argtypes = [vp]
    L.rtx_scene_bvh_depth.restype = C.c_int32
    L.rtx_scene_create_from_host.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.rtx_scene_destroy.argtypes = [vp]
    L.rtx_scene_destroy.restype = None
    L.rtx_scene_device_bytes.argtypes = [vp]
    L.rtx_scene_device_bytes.restype = C.c_uint64
    L.rtx_render_params_default.argtypes = [C.POINTER(RenderParams), C.c_int32, C.c_int32]
    L.rtx_render_params_default.restype = None
    L.rtx_output_pixels.argtypes = [C.POINTER(RenderParams)]
    L.rtx_output_pixels.restype = C.c_uint64
    L.rtx_render.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    L.rtx_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, vp, vp, vp]
    L.rtx_render_device_timed.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, vp, vp, vp, vp]
    L.rtx_render_device_staged.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, vp, vp, vp, vp]
    L.rtx_render_device_noalloc.argtypes = [vp, C.POINTER(RenderParams), vp, vp, vp, vp, vp, vp]
    L.rtx_render_begin.argtypes = [vp, C.POINTER(RenderParams), C.c_int32, C.c_int32]
    L.rtx_render_end.argtypes = [vp, C.c_int32, vp, vp]
    L.rtx_scene_group_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int32), C.c_int32, C.POINTER(vp)]
    L.rtx_scene_group_create_from_host.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32, C.POINTER(vp)]
    L.rtx_scene_group_destroy.argtypes = [vp]
    L.rtx_scene_group_destroy.restype = None
    L.rtx_scene_group_size.argtypes = [vp]
    L.rtx_scene_group_size.restype = C.c_int32
    L.rtx_scene_group_device.argtypes = [vp, C.c_int32]
    L.rtx_scene_group_device.restype = C.c_int32
    L.rtx_scene_group_scene.argtypes = [vp, C.c_int32]
    L.rtx_scene_group_scene.restype = vp
    L.rtx_render_sharded.argtypes = [vp, C.POINTER(RenderParams), vp]
    L.rtx_render_sharded_device.argtypes = [vp, C.POINTER(RenderParams), C.c_int32, C.POINTER(vp)]
    L.rtx_scene_group_sync.argtypes = [vp, C.c_int32]
    L.rtx_scene_group_set_gather.argtypes = [vp, C.c_int32]
    L.rtx_untile.argtypes = [C.POINTER(RenderParams), C.c_int32, vp, vp, C.c_int32, vp]
    L.rtx_save_ppm.argtypes = [C.c_char_p, C.c_int32, C.c_int32, vp]
    L.rtx_save_png.argtypes = [C.c_char_p, C.c_int32, C.c_int32, vp]
    L.rtx_selftest_division.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.rtx_selftest_pow.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rtx_selftest_sqrt.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.rtx_bvh_build.argtypes = [C.POINTER(C.c_double), C.c_uint64, C.c_int32, C.c_int32, vp, C.POINTER(C.c_uint64),
                                C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.rtx_host_scene_rebuild_bvh.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    _lib = L
    return L


def version() -> str:
    """rtx_version(): library name, target and a hash of the render kernels' sources."""
    return lib().rtx_version().decode()


def kernel_build() -> str:
    """The hash of the render kernels' sources this library was built from (last word of rtx_version())."""
    return version().split()[-1]


def _check(rc: int) -> None:
    if rc != 0:
        raise RtxError(rc, lib().rtx_last_error().decode(errors="replace"))


def _dptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


NODE_DTYPE = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("first", "<u8"), ("count", "<u8")])
BUILDER_HOST, BUILDER_DEVICE = 0, 1
LOAD_VERBOSE, LOAD_DEVICE_BVH, LOAD_TOLERANT, LOAD_NO_BVH, STANDIN_KNOT, LOAD_LAST_FACE_TWICE = 1, 2, 4, 8, 16, 32


def build_bvh(vertices, builder: int = BUILDER_DEVICE, device: int = -1) -> dict:
    """BVH::BVH (include/bvh.hpp:84-98) over (n, 9) vertices: the reference's `nodes` and `indices`."""
    v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 9)
    n = v.shape[0]
    nodes = np.zeros(max(1, 2 * n - 1), dtype=NODE_DTYPE)
    indices = np.zeros(n, dtype=np.uint64)
    used, depth, seconds = C.c_uint64(), C.c_int32(), C.c_double()
    _check(lib().rtx_bvh_build(_dptr(v, C.c_double), n, builder, device, nodes.ctypes.data, _dptr(indices, C.c_uint64),
                               C.byref(used), C.byref(depth), C.byref(seconds)))
    return {"nodes": nodes[: used.value].copy(), "indices": indices, "depth": depth.value, "seconds": seconds.value}


def selftest_pow(n: int = 1 << 20, seed: int = 1):
    """(results differing from the host's pow, largest difference in ulps) of the device's pow on the specular domain."""
    d, m = C.c_uint64(), C.c_uint64()
    _check(lib().rtx_selftest_pow(n, seed, C.byref(d), C.byref(m)))
    return d.value, m.value


def selftest_sqrt(n: int = 1 << 22, seed: int = 1) -> int:
    """Results of the device's f64 sqrt that differ from the host's correctly rounded one (must be 0)."""
    d = C.c_uint64()
    _check(lib().rtx_selftest_sqrt(n, seed, C.byref(d)))
    return d.value


def make_params(width: int, height: int, *, depth: int = 3, spp: int = 1, seed: int = 0,
                traversal: int = TRAVERSAL_REFERENCE, window=None, tile=(32, 32), tile_rank: int = 0,
                tile_count: int = 1, layout: int = LAYOUT_ROW_MAJOR) -> RenderParams:
    p = RenderParams()
    lib().rtx_render_params_default(C.byref(p), width, height)
    p.depth, p.spp, p.seed, p.traversal = depth, spp, seed, traversal
    if window is not None:
        p.x0, p.x1, p.y0, p.y1 = window
    p.tile_w, p.tile_h = tile
    p.tile_rank, p.tile_count, p.layout = tile_rank, tile_count, layout
    return p


def desc_from_arrays(vertices, colors, specularity, reflectivity, *, camera_position, camera_rotation_deg=(0, 0, 0),
                     viewport=(1, 1, 1), background=(255, 255, 255), ambient=0.2, point_lights=(), directional_lights=(),
                     nodes=None, indices=None, normals=None, device=-1):
    """An rtx_scene_desc over numpy arrays; returns (desc, objects that must stay alive while it is used)."""
    v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 9)
    n = v.shape[0]
    col = np.ascontiguousarray(colors, dtype=np.int32).reshape(n, 3)
    spec = np.ascontiguousarray(specularity, dtype=np.int32).reshape(n)
    refl = np.ascontiguousarray(reflectivity, dtype=np.float64).reshape(n)
    d = SceneDesc()
    d.n_triangles = n
    d.vertices, d.colors = _dptr(v, C.c_double), _dptr(col, C.c_int32)
    d.specularity, d.reflectivity = _dptr(spec, C.c_int32), _dptr(refl, C.c_double)
    keep = [v, col, spec, refl]
    if normals is not None:
        nr = np.ascontiguousarray(normals, dtype=np.float64).reshape(n, 3)
        d.normals = _dptr(nr, C.c_double)
        keep.append(nr)
    if nodes is not None:
        nd = np.ascontiguousarray(nodes)
        ix = np.ascontiguousarray(indices, dtype=np.uint64)
        d.n_nodes = nd.shape[0]
        d.nodes = C.cast(nd.ctypes.data, C.POINTER(BvhNode))
        d.indices = _dptr(ix, C.c_uint64)
        keep += [nd, ix]
    d.viewport[:] = viewport
    d.camera_position[:] = camera_position
    d.camera_rotation_deg[:] = camera_rotation_deg
    d.background[:] = background
    d.ambient = ambient
    pl = (Light * max(1, len(point_lights)))()
    for i, (inten, vec) in enumerate(point_lights):
        pl[i].intensity = inten
        pl[i].v[:] = vec
    dl = (Light * max(1, len(directional_lights)))()
    for i, (inten, vec) in enumerate(directional_lights):
        dl[i].intensity = inten
        dl[i].v[:] = vec
    d.n_point_lights, d.point_lights = len(point_lights), pl
    d.n_directional_lights, d.directional_lights = len(directional_lights), dl
    d.device = device
    keep += [pl, dl]
    return d, keep


class HostScene:
    """What the reference holds in `Scene` + `BVH` (include/scene.hpp:15-23, include/bvh.hpp:32-37)."""

    def __init__(self, handle):
        self._h = handle
        self._keep = None

    @classmethod
    def load_obj(cls, path: str, preset: str, verbose: bool = False, *, device_bvh: bool = False, tolerant: bool = False,
                 no_bvh: bool = False, device: int = -1, last_face_twice: bool = False) -> "HostScene":
        h = C.c_void_p()
        flags = ((LOAD_VERBOSE if verbose else 0) | (LOAD_DEVICE_BVH if device_bvh else 0) | (LOAD_TOLERANT if tolerant else 0) |
                 (LOAD_NO_BVH if no_bvh else 0) | (LOAD_LAST_FACE_TWICE if last_face_twice else 0))
        _check(lib().rtx_host_scene_load_obj_ex(os.fsencode(path), preset.encode(), flags, device, C.byref(h)))
        return cls(h)

    @classmethod
    def standin(cls, preset: str, n_triangles: int | None = None, seed: int = 0, obj_path_out: str | None = None,
                verbose: bool = False, *, device_bvh: bool = False, no_bvh: bool = False, device: int = -1,
                family: str = "blob") -> "HostScene":
        """family: "blob" (a bumpy star-shaped surface) or "knot" (a knotted tube: concave, self-occluding)."""
        h = C.c_void_p()
        n = SCENE_TRIANGLES[preset] if n_triangles is None else n_triangles
        out = os.fsencode(obj_path_out) if obj_path_out else None
        flags = (LOAD_VERBOSE if verbose else 0) | (LOAD_DEVICE_BVH if device_bvh else 0) | (LOAD_NO_BVH if no_bvh else 0)
        if family not in ("blob", "knot"):
            raise ValueError("family must be 'blob' or 'knot'")
        flags |= STANDIN_KNOT if family == "knot" else 0
        _check(lib().rtx_host_scene_standin_ex(preset.encode(), n, seed, out, flags, device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, vertices, colors, specularity, reflectivity, **kw) -> "HostScene":
        d, keep = desc_from_arrays(vertices, colors, specularity, reflectivity, **kw)
        h = C.c_void_p()
        _check(lib().rtx_host_scene_from_desc(C.byref(d), C.byref(h)))
        del keep
        return cls(h)

    def desc(self) -> SceneDesc:
        d = SceneDesc()
        _check(lib().rtx_host_scene_desc(self._h, C.byref(d)))
        return d

    @property
    def n_triangles(self) -> int:
        return int(self.desc().n_triangles)

    @property
    def n_nodes(self) -> int:
        return int(self.desc().n_nodes)

    @property
    def bvh_depth(self) -> int:
        return int(lib().rtx_host_scene_bvh_depth(self._h))

    def arrays(self) -> dict:
        """Copies of the scene arrays (numpy)."""
        d = self.desc()
        n, m = int(d.n_triangles), int(d.n_nodes)
        node_dt = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("first", "<u8"), ("count", "<u8")])
        return {
            "vertices": np.ctypeslib.as_array(d.vertices, (n, 9)).copy(),
            "normals": np.ctypeslib.as_array(d.normals, (n, 3)).copy(),
            "colors": np.ctypeslib.as_array(d.colors, (n, 3)).copy(),
            "specularity": np.ctypeslib.as_array(d.specularity, (n,)).copy(),
            "reflectivity": np.ctypeslib.as_array(d.reflectivity, (n,)).copy(),
            # a scene loaded without a tree has neither
            "nodes": np.frombuffer(C.string_at(d.nodes, m * 64), dtype=node_dt).copy() if m else np.zeros(0, node_dt),
            "indices": np.ctypeslib.as_array(d.indices, (n,)).copy() if m else np.zeros(0, np.uint64),
            "camera_position": np.array(d.camera_position[:]),
            "camera_rotation_deg": np.array(d.camera_rotation_deg[:]),
        }

    def rebuild_bvh(self, builder: int = BUILDER_DEVICE, device: int = -1) -> float:
        """Replaces the scene's BVH by one from `builder`; returns the build time in seconds."""
        seconds = C.c_double()
        _check(lib().rtx_host_scene_rebuild_bvh(self._h, builder, device, C.byref(seconds)))
        return seconds.value

    def set_camera(self, position, rotation_deg) -> None:
        pos = (C.c_double * 3)(*position)
        rot = (C.c_double * 3)(*rotation_deg)
        _check(lib().rtx_host_scene_set_camera(self._h, pos, rot))

    def close(self) -> None:
        if self._h:
            lib().rtx_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class Frame:
    rgb: np.ndarray            # (H, W, 3) uint8
    hit: np.ndarray | None     # (H, W) int32 primary triangle ids, -1 = background
    stats: Stats | None


class Scene:
    """A scene resident in one GPU's HBM."""

    def __init__(self, host: HostScene | None, device: int = -1, *, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        _check(lib().rtx_scene_create_from_host(host._h, device, C.byref(h)))
        self._h = h

    @classmethod
    def from_arrays(cls, vertices, colors, specularity, reflectivity, *, builder: int = BUILDER_DEVICE, **kw) -> "Scene":
        """rtx_scene_create_with_builder(): triangles in, device scene out (BVH built by `builder`
        unless nodes/indices are passed)."""
        d, keep = desc_from_arrays(vertices, colors, specularity, reflectivity, **kw)
        h = C.c_void_p()
        _check(lib().rtx_scene_create_with_builder(C.byref(d), builder, C.byref(h)))
        del keep
        return cls(None, _handle=h)

    @property
    def build_seconds(self) -> float:
        return float(lib().rtx_scene_build_seconds(self._h))

    @property
    def bvh_nodes(self) -> int:
        return int(lib().rtx_scene_bvh_nodes(self._h))

    @property
    def bvh_depth(self) -> int:
        return int(lib().rtx_scene_bvh_depth(self._h))

    @property
    def device_bytes(self) -> int:
        return int(lib().rtx_scene_device_bytes(self._h))

    def render(self, width: int, height: int, *, want_hit: bool = True, want_stats: bool = False, **kw) -> Frame:
        """One frame into host arrays through rtx_render()."""
        p = make_params(width, height, **kw)
        px = int(lib().rtx_output_pixels(C.byref(p)))
        packed = p.layout == LAYOUT_PACKED_TILES
        rgb = np.zeros((px, 3), dtype=np.uint8)
        hit = np.full(px, -1, dtype=np.int32) if (want_hit and not packed) else None
        st = Stats() if want_stats else None
        _check(lib().rtx_render(self._h, C.byref(p), rgb.ctypes.data, hit.ctypes.data if hit is not None else None,
                                C.byref(st) if st is not None else None))
        if not packed:
            rgb = rgb.reshape(height, width, 3)
            if hit is not None:
                hit = hit.reshape(height, width)
        return Frame(rgb, hit, st)

    def render_device(self, params: RenderParams, *, rgb_u8=0, hit=0, rgb_i32=0, t=0, stats=0, ticks=0, stream=0) -> None:
        """Asynchronous frame into device pointers (ints, e.g. torch.Tensor.data_ptr()).  ticks: two uint64 the
        caller zeroed; afterwards [0] is the complement of the kernel's first-block start and [1] its last-block end on the
        device's 100 MHz clock (duration = [1] - ~[0])."""
        _check(lib().rtx_render_device_timed(self._h, C.byref(params), rgb_u8 or None, hit or None, rgb_i32 or None,
                                             t or None, stats or None, ticks or None, stream or None))

    def render_device_staged(self, params: RenderParams, *, rgb_u8=0, hit=0, rgb_i32=0, t=0, stats=0, ticks=0, stream=0) -> None:
        """render_device() through the staged pipeline (one launch per level of the reflection chain; one sample per pixel)."""
        _check(lib().rtx_render_device_staged(self._h, C.byref(params), rgb_u8 or None, hit or None, rgb_i32 or None,
                                              t or None, stats or None, ticks or None, stream or None))

    def render_device_batch(self, params: RenderParams, n_frames: int, *, cameras=None, rgb_u8=0, rgb_i32=0, stats=0, ticks=0, stream=0) -> None:
        """rtx_render_device_batch(): n_frames frames in one launch, consecutive in the output buffers; cameras: optional
        (n_frames, 6) array of position + rotation in degrees, one camera per frame."""
        cam = None
        if cameras is not None:
            cam = np.ascontiguousarray(cameras, dtype=np.float64).reshape(n_frames, 6)
        L = lib()
        L.rtx_render_device_batch.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        _check(L.rtx_render_device_batch(self._h, C.byref(params), n_frames, cam.ctypes.data if cam is not None else None,
                                         rgb_u8 or None, rgb_i32 or None, stats or None, ticks or None, stream or None))

    def render_begin(self, params: RenderParams, slot: int, want_hit: bool = False) -> None:
        """rtx_render_begin(): queues a frame on `slot` (0..3) on its way to host memory."""
        _check(lib().rtx_render_begin(self._h, C.byref(params), slot, 1 if want_hit else 0))

    def render_end(self, slot: int, rgb: np.ndarray | None = None, hit: np.ndarray | None = None) -> None:
        """rtx_render_end(): waits for the slot's frame and copies it into the arrays given."""
        _check(lib().rtx_render_end(self._h, slot, rgb.ctypes.data if rgb is not None else None,
                                    hit.ctypes.data if hit is not None else None))

    def render_device_noalloc(self, params: RenderParams, *, rgb_u8=0, hit=0, rgb_i32=0, t=0, stats=0, stream=0) -> None:
        """render_device() through the entry point that allocates nothing (scratch in the library's fixed device array)."""
        _check(lib().rtx_render_device_noalloc(self._h, C.byref(params), rgb_u8 or None, hit or None, rgb_i32 or None,
                                               t or None, stats or None, stream or None))

    def close(self) -> None:
        if self._h:
            lib().rtx_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SceneGroup:
    """A scene replicated on several GPUs of the node and rendered cooperatively from one process
    (rtx_scene_group_*): screen tiles round-robin over the devices, one RCCL all-gather, frame on the first device."""

    def __init__(self, host: HostScene, devices=None, n_devices: int = 0):
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*devices)
            _check(lib().rtx_scene_group_create_from_host(host._h, arr, len(devices), C.byref(h)))
        else:
            _check(lib().rtx_scene_group_create_from_host(host._h, None, n_devices, C.byref(h)))
        self._h = h

    @property
    def size(self) -> int:
        return int(lib().rtx_scene_group_size(self._h))

    def devices(self) -> list:
        return [int(lib().rtx_scene_group_device(self._h, i)) for i in range(self.size)]

    def set_gather(self, mode: int) -> None:
        """GATHER_ALL (one all-gather) or GATHER_ROOT (sends to the root, which alone ends with the frame's tiles)."""
        _check(lib().rtx_scene_group_set_gather(self._h, mode))

    def render(self, width: int, height: int, **kw) -> np.ndarray:
        """One frame through rtx_render_sharded(): (H, W, 3) uint8."""
        p = make_params(width, height, **kw)
        rgb = np.zeros((height, width, 3), dtype=np.uint8)
        _check(lib().rtx_render_sharded(self._h, C.byref(p), rgb.ctypes.data))
        return rgb

    def submit(self, params: RenderParams, slot: int) -> int:
        """Queues a frame on `slot`; returns the root-device pointer of the slot's frame (valid after sync())."""
        ptr = C.c_void_p()
        _check(lib().rtx_render_sharded_device(self._h, C.byref(params), slot, C.byref(ptr)))
        return int(ptr.value or 0)

    def sync(self, slot: int = -1) -> None:
        _check(lib().rtx_scene_group_sync(self._h, slot))

    def close(self) -> None:
        if self._h:
            lib().rtx_scene_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


GATHER_ALL, GATHER_ROOT = 0, 1
GATHER_OP_ALLGATHER, GATHER_OP_SEND, GATHER_OP_RECV, GATHER_OP_COPY = 0, 1, 2, 3


class GatherOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("member", C.c_int32), ("peer", C.c_int32), ("reserved", C.c_int32),
                ("src_offset", C.c_uint64), ("dst_offset", C.c_uint64), ("bytes", C.c_uint64)]


def gather_plan(n_members: int, local_bytes: int, mode: int) -> list:
    """rtx_gather_plan(): the operations a scene group runs to bring its members' packed tile buffers together."""
    ops = (GatherOp * (2 * max(1, n_members) + 2))()
    L = lib()
    L.rtx_gather_plan.restype = C.c_int32
    L.rtx_gather_plan.argtypes = [C.c_int32, C.c_uint64, C.c_int32, C.POINTER(GatherOp), C.c_int32]
    n = L.rtx_gather_plan(n_members, local_bytes, mode, ops, len(ops))
    _check(min(n, 0))
    return [ops[i] for i in range(n)]


def untile(params: RenderParams, n_ranks: int, gathered_ptr: int, out_ptr: int, bytes_per_pixel: int = 3, stream: int = 0):
    _check(lib().rtx_untile(C.byref(params), n_ranks, gathered_ptr, out_ptr, bytes_per_pixel, stream or None))


def selftest_division(n: int, seed: int = 0) -> int:
    """Mismatches between the slab test's fast division and the plain one over n seeded pairs (must be 0)."""
    bad = C.c_uint64(0)
    _check(lib().rtx_selftest_division(n, seed, C.byref(bad)))
    return int(bad.value)


def save_ppm(path: str, rgb: np.ndarray) -> None:
    """saveCanvas (src/main.cpp:96-116)."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = a.shape[:2]
    _check(lib().rtx_save_ppm(os.fsencode(path), w, h, a.ctypes.data))


def save_png(path: str, rgb: np.ndarray) -> None:
    """The canvas as an 8-bit RGB PNG."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = a.shape[:2]
    _check(lib().rtx_save_png(os.fsencode(path), w, h, a.ctypes.data))
