#!/usr/bin/env python3
"""Generates the committed parity fixtures from the REFERENCE ITSELF.

Runs only in the build container: it needs oracle/_ref/ref_bvh and ref_brute (the
reference's own headers compiled from /root/reference, see oracle/ref_harness.cpp) and
/root/reference/scenes/teapot.obj.  Outputs (all data, no reference source):

  tests/golden/teapot.obj            the reference's scene file (input data)
  tests/golden/golden.json           hashes / counts printed by the reference harness
  tests/golden/teapot_128.npz        raw primary hit ids, unclamped RGB, primary t (128x128)
  tests/golden/bunny20010_192x108.npz  same for a stand-in mesh (the reference parses the
                                       .obj this repository's generator wrote)
  tests/golden/teapot_turned_96x64.npz raw arrays for the teapot seen from TURNED_CAMERAS (the stock
                                       factories only ever turn the camera by 0.1 degrees)

`make_golden.py --turned-only` regenerates just the turned-camera fixtures and keeps the rest of golden.json;
`--full-size-only` adds the hashes of the two benchmarked 3840x2160 frames as the reference renders them (minutes);
`--knot-only` adds the fixtures of the knotted-tube stand-in family (small raw case, small and full-size hashes).
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_BVH = os.path.join(ROOT, "oracle", "_ref", "ref_bvh")
REF_BRUTE = os.path.join(ROOT, "oracle", "_ref", "ref_brute")
REFERENCE = "/root/reference"


# BASELINE's single-frame configurations at their full sizes
FULL_SIZE = [("bunny", 144056, 1920, 1080), ("erato", 412508, 3840, 2160), ("dragon", 871316, 3840, 2160),
             ("aurelius", 1704768, 3840, 2160)]

# position, rotation in degrees: all three angles at once, a quarter turn, nearly half a turn
TURNED_CAMERAS = [
    ((3.5, 6.0, -9.0), (-29.266614452137418, -28.274912085355048, -4.294279981910943)),
    ((-6.0, 5.0, -8.0), (20.0, 33.3, 7.0)),
    ((0.0, 12.0, 0.5), (90.0, 0.0, 0.0)),
    ((2.0, 3.0, 8.0), (10.0, 170.0, -15.0)),
]


def run_ref(binary, scene, scene_root, W, H, depth, prefix="-", threads=8, camera=None):
    env = dict(os.environ)
    if camera is not None:
        env["REF_CAMERA"] = ",".join(repr(float(x)) for x in (*camera[0], *camera[1]))
    out = subprocess.run([binary, scene, scene_root, str(W), str(H), str(depth), prefix, str(threads)],
                         check=True, capture_output=True, text=True, env=env).stdout
    d = {}
    for line in out.splitlines():
        if line.startswith("["):
            continue
        k, *v = line.split()
        d[k] = v[0] if len(v) == 1 else " ".join(v)
    d.pop("frame_seconds", None)
    return d


def raw(prefix, W, H):
    return dict(
        hit=np.fromfile(prefix + ".hit.i32", dtype=np.int32).reshape(H, W),
        rgb_i32=np.fromfile(prefix + ".rgb.i32", dtype=np.int32).reshape(H, W, 3),
        t=np.fromfile(prefix + ".t.f64", dtype=np.float64).reshape(H, W),
    )


def turned(g, tmp):
    arrays = {}
    for k, cam in enumerate(TURNED_CAMERAS):
        p = os.path.join(tmp, f"turned{k}")
        g[f"teapot_turned{k}_96x64"] = run_ref(REF_BVH, "teapot", REFERENCE, 96, 64, 3, p, camera=cam)
        g[f"teapot_turned{k}_96x64"]["camera"] = [list(cam[0]), list(cam[1])]
        for name, a in raw(p, 96, 64).items():
            arrays[f"{name}{k}"] = a
        g[f"teapot_turned{k}_640x360"] = run_ref(REF_BVH, "teapot", REFERENCE, 640, 360, 3, camera=cam)
    np.savez_compressed(os.path.join(HERE, "teapot_turned_96x64.npz"), **arrays)


def main():
    import ray_tracer_amd as rt

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if "--full-size-only" in sys.argv:
        # BASELINE's frames themselves (bunny 1920x1080; erato, dragon, aurelius 3840x2160), rendered by the reference:
        # minutes of CPU each, hashes only
        path = os.path.join(HERE, "golden.json")
        g = json.load(open(path))
        with tempfile.TemporaryDirectory() as tmp:
            for scene, ntri, W, H in FULL_SIZE:
                if f"standin_{scene}_{ntri}_seed0_{W}x{H}" in g and "--force" not in sys.argv:
                    continue
                root = os.path.join(tmp, scene)
                os.makedirs(os.path.join(root, "scenes"))
                obj = os.path.join(root, "scenes", f"{scene}.obj")
                rt.HostScene.standin(scene, ntri, seed=0, obj_path_out=obj, no_bvh=True).close()
                key = f"standin_{scene}_{ntri}_seed0_{W}x{H}"
                g[key] = run_ref(REF_BVH, scene, root, W, H, 3)
                g[key]["obj_md5"] = hashlib.md5(open(obj, "rb").read()).hexdigest()
                shutil.rmtree(root)
                print(key, g[key]["hit_fnv"], flush=True)
                with open(path, "w") as f:
                    json.dump(g, f, indent=1, sort_keys=True)
        return
    if "--knot-only" in sys.argv:
        # the second stand-in family (a knotted tube: concave, self-occluding) as the reference renders it: a small
        # case with raw arrays, every scene at a small size and BASELINE's frames at their full sizes (hashes)
        path = os.path.join(HERE, "golden.json")
        g = json.load(open(path))
        cases = [("bunny", 20010, 7, 192, 108, True)]
        for scene, ntri, W, H in FULL_SIZE:
            cases += [(scene, ntri, 0, 480, 270, False), (scene, ntri, 0, W, H, False)]
        with tempfile.TemporaryDirectory() as tmp:
            for scene, ntri, seed, W, H, keep_raw in cases:
                key = f"knot_{scene}_{ntri}_seed{seed}_{W}x{H}"
                if key in g and "--force" not in sys.argv:
                    continue
                root = os.path.join(tmp, f"{scene}{ntri}")
                os.makedirs(os.path.join(root, "scenes"))
                obj = os.path.join(root, "scenes", f"{scene}.obj")
                rt.HostScene.standin(scene, ntri, seed=seed, obj_path_out=obj, no_bvh=True, family="knot").close()
                p = os.path.join(tmp, key)
                g[key] = run_ref(REF_BVH, scene, root, W, H, 3, p if keep_raw else "-")
                g[key]["obj_md5"] = hashlib.md5(open(obj, "rb").read()).hexdigest()
                if keep_raw:
                    np.savez_compressed(os.path.join(HERE, f"knot_{scene}{ntri}_{W}x{H}.npz"), **raw(p, W, H))
                shutil.rmtree(root)
                print(key, g[key]["hit_fnv"], g[key].get("rays"), flush=True)
                with open(path, "w") as f:
                    json.dump(g, f, indent=1, sort_keys=True)
        return
    if "--showcase-only" in sys.argv:
        # the library's "showcase" preset (point lights, specular exponents 10 and 50) rendered by the reference: its
        # lights and materials go into the reference's Scene through the harness directives
        path = os.path.join(HERE, "golden.json")
        g = json.load(open(path))
        hs = rt.HostScene.load_obj(os.path.join(HERE, "teapot.obj"), "showcase", no_bvh=True)
        a, d = hs.arrays(), hs.desc()
        with tempfile.TemporaryDirectory() as tmp:
            rec = np.zeros(a["colors"].shape[0], dtype=np.dtype([("rgb", "<i4", 3), ("spec", "<i4"), ("refl", "<f8")]))
            rec["rgb"], rec["spec"], rec["refl"] = a["colors"], a["specularity"], a["reflectivity"]
            rec.tofile(os.path.join(tmp, "materials.bin"))
            lines = ["ambient %r" % float(d.ambient), "nolights"]
            lines += ["point %r %r %r %r" % (d.point_lights[i].intensity, *d.point_lights[i].v) for i in range(d.n_point_lights)]
            lines += ["directional %r %r %r %r" % (d.directional_lights[i].intensity, *d.directional_lights[i].v)
                      for i in range(d.n_directional_lights)]
            lines += ["materials " + os.path.join(tmp, "materials.bin")]
            open(os.path.join(tmp, "scene.txt"), "w").write("\n".join(lines) + "\n")
            os.environ["REF_SCENE_FILE"] = os.path.join(tmp, "scene.txt")
            p = os.path.join(tmp, "show")
            g["teapot_showcase_160x120"] = run_ref(REF_BVH, "teapot", REFERENCE, 160, 120, 3, p)
            np.savez_compressed(os.path.join(HERE, "teapot_showcase_160x120.npz"), **raw(p, 160, 120),
                                rgb_u8=np.fromfile(p + ".rgb.u8", dtype=np.uint8).reshape(120, 160, 3))
            g["teapot_showcase_1920x1080"] = run_ref(REF_BVH, "teapot", REFERENCE, 1920, 1080, 3, p)
            np.savez_compressed(os.path.join(HERE, "teapot_showcase_1920x1080_rgb.npz"),
                                rgb_u8=np.fromfile(p + ".rgb.u8", dtype=np.uint8).reshape(1080, 1920, 3))
            del os.environ["REF_SCENE_FILE"]
        with open(path, "w") as f:
            json.dump(g, f, indent=1, sort_keys=True)
        print("updated", path)
        return
    if "--turned-only" in sys.argv:
        path = os.path.join(HERE, "golden.json")
        g = json.load(open(path))
        with tempfile.TemporaryDirectory() as tmp:
            turned(g, tmp)
        with open(path, "w") as f:
            json.dump(g, f, indent=1, sort_keys=True)
        print("updated", path)
        return
    shutil.copyfile(os.path.join(REFERENCE, "scenes", "teapot.obj"), os.path.join(HERE, "teapot.obj"))
    g = {"_generator": "tests/golden/make_golden.py", "_source": "oracle/_ref (reference headers compiled with g++ -O3)"}
    with tempfile.TemporaryDirectory() as tmp:
        # ---- teapot, the one scene the reference ships
        g["teapot_200_bvh"] = run_ref(REF_BVH, "teapot", REFERENCE, 200, 200, 3)
        g["teapot_200_brute"] = run_ref(REF_BRUTE, "teapot", REFERENCE, 200, 200, 3)
        p = os.path.join(tmp, "t128")
        g["teapot_128"] = run_ref(REF_BVH, "teapot", REFERENCE, 128, 128, 3, p)
        np.savez_compressed(os.path.join(HERE, "teapot_128.npz"), **raw(p, 128, 128))
        p = os.path.join(tmp, "t1920")
        g["teapot_1920"] = run_ref(REF_BVH, "teapot", REFERENCE, 1920, 1920, 3, p)
        ppm = b"P6\n1920 1920 255\n" + open(p + ".rgb.u8", "rb").read()
        g["teapot_1920"]["ppm_md5"] = hashlib.md5(ppm).hexdigest()
        g["teapot_1920x1080"] = run_ref(REF_BVH, "teapot", REFERENCE, 1920, 1080, 3)
        for depth in (0, 1):
            g[f"teapot_160x96_depth{depth}"] = run_ref(REF_BVH, "teapot", REFERENCE, 160, 96, depth)
        # the loader quirk: a trailing newline duplicates the last face
        os.makedirs(os.path.join(tmp, "nl", "scenes"))
        with open(os.path.join(tmp, "nl", "scenes", "teapot.obj"), "wb") as f:
            f.write(open(os.path.join(REFERENCE, "scenes", "teapot.obj"), "rb").read() + b"\n")
        g["teapot_trailing_newline_64"] = run_ref(REF_BVH, "teapot", os.path.join(tmp, "nl"), 64, 64, 3)
        turned(g, tmp)
        # ---- stand-in meshes: this repository's generator writes the .obj, the reference renders it
        cases = [("bunny", 20010, 7, 192, 108, True), ("bunny", 144056, 0, 480, 270, False),
                 ("erato", 412508, 0, 480, 270, False), ("dragon", 871316, 0, 960, 540, False),
                 ("aurelius", 1704768, 0, 480, 270, False)]
        for scene, ntri, seed, W, H, keep_raw in cases:
            root = os.path.join(tmp, f"{scene}{ntri}")
            os.makedirs(os.path.join(root, "scenes"))
            obj = os.path.join(root, "scenes", f"{scene}.obj")
            rt.HostScene.standin(scene, ntri, seed=seed, obj_path_out=obj).close()
            key = f"standin_{scene}_{ntri}_seed{seed}_{W}x{H}"
            p = os.path.join(tmp, key)
            g[key] = run_ref(REF_BVH, scene, root, W, H, 3, p if keep_raw else "-")
            g[key]["obj_md5"] = hashlib.md5(open(obj, "rb").read()).hexdigest()
            if keep_raw:
                np.savez_compressed(os.path.join(HERE, f"{scene}{ntri}_{W}x{H}.npz"), **raw(p, W, H))
            shutil.rmtree(root)
            print(key, g[key]["hit_fnv"], flush=True)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.json"))


if __name__ == "__main__":
    main()
