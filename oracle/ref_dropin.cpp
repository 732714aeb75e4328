// TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// The drop-in, carried out: the reference's own Scene factory, loader and BVH constructor (its unmodified
// headers, included from /root/reference where they lie), the binding a maintainer would add -- upload() below --
// and one rtx_render() call in place of the thread fan-out of src/main.cpp:45-52.
// The same process then renders the frame the reference's way (toViewport + rotate + traceRay per pixel,
// src/main.cpp:75-88, on host threads) and counts the pixels whose primary triangle id or colour differ.
// Built in the container by `make ref` into oracle/_ref/ref_dropin and linked against ray_tracer_amd/librtx.so.
#include "common.hpp"
#undef USE_BVH
#define USE_BVH 1
#include "scene.hpp"

#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "rtx.h"

// ---- the binding a maintainer would add: the reference's Scene handed to rtx_scene_create ---------------------
static rtx_scene *upload(Scene &scene) {
    std::vector<double> v, nrm, refl; std::vector<int32_t> col, spec;
    for (Triangle &t : scene.triangles) {                       // include/triangle.hpp:8-15
        for (Vector3 p : {t.v0, t.v1, t.v2}) { v.push_back(p.x); v.push_back(p.y); v.push_back(p.z); }
        nrm.insert(nrm.end(), {t.normal.x, t.normal.y, t.normal.z});
        col.insert(col.end(), {t.color.r, t.color.g, t.color.b});
        spec.push_back(t.specularity); refl.push_back(t.reflectivity);
    }
    rtx_scene_desc d = {};
    d.n_triangles = scene.triangles.size();
    d.vertices = v.data(); d.normals = nrm.data(); d.colors = col.data();
    d.specularity = spec.data(); d.reflectivity = refl.data();
    static_assert(sizeof(BVHNode) == sizeof(rtx_bvh_node), "same 64-byte record");   // include/bvh.hpp:25-30
    d.n_nodes = scene.bvh.nodesUsed;                            // the tree the reference built
    d.nodes = reinterpret_cast<const rtx_bvh_node *>(scene.bvh.nodes);
    d.indices = reinterpret_cast<const uint64_t *>(scene.bvh.indices.data());
    const Vector3 *src[3] = {&scene.viewport, &scene.cameraPosition, &scene.cameraRotation};
    double *dst[3] = {d.viewport, d.camera_position, d.camera_rotation_deg};
    for (int k = 0; k < 3; k++) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    d.background[0] = scene.backgroundColor.r; d.background[1] = scene.backgroundColor.g; d.background[2] = scene.backgroundColor.b;
    d.ambient = scene.ambientLight;
    std::vector<rtx_light> pl, dl;                              // include/light.hpp:8-16
    for (Point &p : scene.pointLights) pl.push_back({p.intensity, {p.position.x, p.position.y, p.position.z}});
    for (Directional &p : scene.directionalLights) dl.push_back({p.intensity, {p.direction.x, p.direction.y, p.direction.z}});
    d.n_point_lights = pl.size(); d.point_lights = pl.data();
    d.n_directional_lights = dl.size(); d.directional_lights = dl.data();
    d.device = -1;
    rtx_scene *s = nullptr;
    if (rtx_scene_create(&d, &s) != RTX_OK) throw rtx_last_error();      // main() prints it and returns -1
    return s;
}
// ----------------------------------------------------------------------------------------------------------------

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <dir holding scenes/teapot.obj> <W> <H> <pruned|reference> [let-the-library-build-the-tree]\n", argv[0]);
        return 2;
    }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    const bool pruned = std::string(argv[4]) == "pruned";
    const bool own_tree = argc > 5 && atoi(argv[5]) != 0;
    if (chdir(argv[1]) != 0) { fprintf(stderr, "cannot chdir to %s\n", argv[1]); return 2; }
    Scene scene;
    rtx_scene *gpu_scene = nullptr;
    try {
        scene = Scene::teapot();
        if (own_tree) {
            // the other route: n_nodes = 0, the library builds the tree (on the GPU)
            Scene bare = scene;
            bare.bvh.nodesUsed = 0;
            bare.bvh.nodes = nullptr;
            decltype(bare.bvh.indices)().swap(bare.bvh.indices);      // no storage left: data() is null
            gpu_scene = upload(bare);
        } else {
            gpu_scene = upload(scene);
        }
    } catch (const char *e) {
        std::cerr << e;
        return -1;
    }

    // the third route: the reference keeps its data structures and only the build moves to the GPU
    {
        std::vector<double> v;
        for (Triangle &t : scene.triangles)
            for (Vector3 q : {t.v0, t.v1, t.v2}) { v.push_back(q.x); v.push_back(q.y); v.push_back(q.z); }
        const size_t n = scene.triangles.size();
        std::vector<rtx_bvh_node> nodes(2 * n - 1);
        std::vector<uint64_t> indices(n);
        uint64_t used = 0;
        if (rtx_bvh_build(v.data(), n, RTX_BUILDER_DEVICE, -1, nodes.data(), indices.data(), &used, nullptr, nullptr) != RTX_OK) {
            std::cerr << rtx_last_error();
            return -1;
        }
        const bool same = used == scene.bvh.nodesUsed && memcmp(nodes.data(), scene.bvh.nodes, used * sizeof(BVHNode)) == 0 &&
                          memcmp(indices.data(), scene.bvh.indices.data(), n * sizeof(uint64_t)) == 0;
        printf("rtx_bvh_build on the device: %llu nodes, %s BVH::nodes / BVH::indices\n", static_cast<unsigned long long>(used),
               same ? "identical to" : "DIFFERENT from");
        if (!same) return 1;
    }

    // per frame, instead of the 8 threads
    rtx_render_params p;
    rtx_render_params_default(&p, W, H);                          // depth = 3 as src/main.cpp:83
    if (pruned) p.traversal = RTX_TRAVERSAL_PRUNED;
    std::vector<uint8_t> rgb(size_t(W) * H * 3);
    std::vector<int32_t> ids(size_t(W) * H);
    if (rtx_render(gpu_scene, &p, rgb.data(), ids.data(), nullptr) != RTX_OK) {
        std::cerr << rtx_last_error();
        return -1;
    }

    // the reference's own frame, src/main.cpp:75-94
    std::vector<uint8_t> want(rgb.size());
    std::vector<int32_t> want_ids(ids.size());
    const int threads = 8;
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; i++)
        pool.emplace_back([&, i]() {
            Scene mine = scene;                                      // passed by value, src/main.cpp:46
            for (int x = -W / 2 + i; x < W - W / 2; x += threads)
                for (int y = -(H - H / 2); y < H / 2; y++) {
                    Vector3 ray = mine.toViewport(x, y, W, H)
                                      .rotateX(mine.cameraRotation.x).rotateY(mine.cameraRotation.y).rotateZ(mine.cameraRotation.z);
                    Color c = mine.traceRay(mine.cameraPosition, ray, 1.0, INFINITY, 3).normalize();
                    const size_t k = size_t(-y + H / 2 - 1) * W + size_t(x + W / 2);
                    want[3 * k] = static_cast<uint8_t>(c.r); want[3 * k + 1] = static_cast<uint8_t>(c.g); want[3 * k + 2] = static_cast<uint8_t>(c.b);
                    want_ids[k] = std::get<2>(mine.bvh.intersectRay(mine.cameraPosition, ray, 1.0, INFINITY, 0));
                }
        });
    for (std::thread &t : pool) t.join();
    size_t bad_rgb = 0, bad_id = 0;
    for (size_t k = 0; k < ids.size(); k++) {
        bad_id += ids[k] != want_ids[k];
        bad_rgb += rgb[3 * k] != want[3 * k] || rgb[3 * k + 1] != want[3 * k + 1] || rgb[3 * k + 2] != want[3 * k + 2];
    }
    printf("pixels %zu id_mismatches %zu rgb_mismatches %zu\n", ids.size(), bad_id, bad_rgb);
    rtx_scene_destroy(gpu_scene);
    return (bad_id || bad_rgb) ? 1 : 0;
}
