"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference render path (oracle/rt_oracle.cpp).
Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


class Census(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "primary_rays", "reflection_rays", "shadow_rays", "shadow_occluded", "closest_hits",
        "node_visits", "box_hits", "leaf_visits", "triangle_tests", "shadow_idle", "shadow_idle_occluded")]

    @property
    def rays(self):
        return self.primary_rays + self.reflection_rays + self.shadow_rays

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("shadow_idle")}
        d["rays"] = self.rays
        return d


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(LIB)
        vp = C.c_void_p
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.ora_scene_create.restype = vp
        L.ora_scene_create.argtypes = [C.c_int64, dp, ip, ip, dp, dp, dp, dp, ip, C.c_double, C.c_int32, dp, C.c_int32, dp]
        L.ora_scene_load_obj.restype = vp
        L.ora_scene_load_obj.argtypes = [C.c_char_p, C.c_char_p]
        L.ora_scene_free.argtypes = [vp]
        L.ora_last_error.restype = C.c_char_p
        L.ora_num_triangles.restype = C.c_int64
        L.ora_num_triangles.argtypes = [vp]
        L.ora_num_nodes.restype = C.c_int64
        L.ora_num_nodes.argtypes = [vp]
        L.ora_max_depth.argtypes = [vp]
        L.ora_get_nodes.argtypes = [vp, vp]
        L.ora_get_indices.argtypes = [vp, vp]
        L.ora_get_geometry.argtypes = [vp, vp]
        L.ora_get_materials.argtypes = [vp, vp, vp, vp]
        L.ora_get_camera.argtypes = [vp, vp, vp]
        L.ora_render.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32,
                                 vp, vp, vp, vp, C.POINTER(Census)]
        L.ora_jitter.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, dp, dp]
        L.ora_closest_hit.argtypes = [vp, dp, dp, C.c_double, C.c_double, C.c_int32, dp, ip]
        L.ora_fnv1a.restype = C.c_uint64
        L.ora_fnv1a.argtypes = [vp, C.c_uint64]
        _lib = L
    return _lib


NODE_DT = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("first", "<u8"), ("count", "<u8")])


def fnv1a(a: np.ndarray) -> str:
    a = np.ascontiguousarray(a)
    return "%016x" % lib().ora_fnv1a(a.ctypes.data, a.nbytes)


class OracleScene:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: " + lib().ora_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def load_obj(cls, path, preset):
        return cls(lib().ora_scene_load_obj(os.fsencode(path), preset.encode()))

    @classmethod
    def from_arrays(cls, vertices, colors, specularity, reflectivity, *, camera_position, camera_rotation_deg=(0, 0, 0),
                    viewport=(1, 1, 1), background=(255, 255, 255), ambient=0.2, point_lights=(), directional_lights=()):
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 9)
        n = v.shape[0]
        col = np.ascontiguousarray(colors, dtype=np.int32).reshape(n, 3)
        spec = np.ascontiguousarray(specularity, dtype=np.int32).reshape(n)
        refl = np.ascontiguousarray(reflectivity, dtype=np.float64).reshape(n)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        f3 = lambda x: (C.c_double * 3)(*x)
        pl = np.array([[i, *vec] for i, vec in point_lights], dtype=np.float64).reshape(-1, 4)
        dl = np.array([[i, *vec] for i, vec in directional_lights], dtype=np.float64).reshape(-1, 4)
        h = lib().ora_scene_create(n, v.ctypes.data_as(dp), col.ctypes.data_as(ip), spec.ctypes.data_as(ip),
                                   refl.ctypes.data_as(dp), f3(viewport), f3(camera_position), f3(camera_rotation_deg),
                                   (C.c_int32 * 3)(*background), ambient, len(pl), pl.ctypes.data_as(dp), len(dl),
                                   dl.ctypes.data_as(dp))
        return cls(h)

    @property
    def n_triangles(self):
        return lib().ora_num_triangles(self._h)

    @property
    def n_nodes(self):
        return lib().ora_num_nodes(self._h)

    @property
    def max_depth(self):
        return lib().ora_max_depth(self._h)

    def nodes(self):
        a = np.zeros(self.n_nodes, dtype=NODE_DT)
        lib().ora_get_nodes(self._h, a.ctypes.data)
        return a

    def indices(self):
        a = np.zeros(self.n_triangles, dtype=np.uint64)
        lib().ora_get_indices(self._h, a.ctypes.data)
        return a

    def geometry(self):
        a = np.zeros((self.n_triangles, 15), dtype=np.float64)
        lib().ora_get_geometry(self._h, a.ctypes.data)
        return a

    def materials(self):
        n = self.n_triangles
        col, spec, refl = np.zeros((n, 3), np.int32), np.zeros(n, np.int32), np.zeros(n, np.float64)
        lib().ora_get_materials(self._h, col.ctypes.data, spec.ctypes.data, refl.ctypes.data)
        return col, spec, refl

    def render(self, W, H, depth=3, spp=1, seed=0, threads=8, use_bvh=True, want_t=False):
        hit = np.full((H, W), -2, np.int32)
        rgb_i32 = np.zeros((H, W, 3), np.int32)
        rgb_u8 = np.zeros((H, W, 3), np.uint8)
        t = np.zeros((H, W), np.float64)
        cen = Census()
        rc = lib().ora_render(self._h, W, H, depth, spp, seed, threads, int(use_bvh), hit.ctypes.data, rgb_i32.ctypes.data,
                              rgb_u8.ctypes.data, t.ctypes.data, C.byref(cen))
        if rc != 0:
            raise RuntimeError("oracle: " + lib().ora_last_error().decode())
        out = {"hit": hit, "rgb_i32": rgb_i32, "rgb_u8": rgb_u8, "census": cen}
        if want_t:
            out["t"] = t
        return out

    def closest_hit(self, o, r, tmin, tmax, use_bvh=True):
        t, i = C.c_double(), C.c_int32()
        any_ = lib().ora_closest_hit(self._h, (C.c_double * 3)(*o), (C.c_double * 3)(*r), tmin, tmax, int(use_bvh),
                                     C.byref(t), C.byref(i))
        return bool(any_), t.value, i.value

    def close(self):
        if self._h:
            lib().ora_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def jitter(seed, k, s):
    jx, jy = C.c_double(), C.c_double()
    lib().ora_jitter(seed, k, s, C.byref(jx), C.byref(jy))
    return jx.value, jy.value
