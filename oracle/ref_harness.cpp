// TEST INFRASTRUCTURE ONLY. Not part of the product; nothing outside tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
// under oracle/.
//
// Harness around the UNMODIFIED reference headers, compiled from where they lie
// (-I/root/reference/include); the binary lands in oracle/_ref/ (git-ignored).
// It drives the reference's own Scene factories, BVH builder and traceRay and
// dumps what the parity fixtures need.  No reference source is copied: the only
// things taken from the reference are its headers, by #include.
//
// The reference selects brute force vs BVH with an unconditional
// `#define USE_BVH 0` (include/common.hpp:14).  common.hpp is include-guarded,
// so including it first and re-defining the macro before scene.hpp makes the
// `#if USE_BVH` blocks in include/scene.hpp:41,92,154 see the value chosen by
// -DREF_USE_BVH=0|1 on the command line of this harness.
//
// The per-pixel loop restates src/main.cpp:75-94 (render/putPixel) with the
// canvas size and recursion depth as run-time arguments; src/main.cpp itself
// cannot be used because the canvas is a compile-time sized global.
#include "common.hpp"
#undef USE_BVH
#define USE_BVH REF_USE_BVH
#include "scene.hpp"

#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

uint64_t fnv1a(const void *data, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

void dump(const std::string &path, const void *data, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(data, 1, n, f) != n) {
        fprintf(stderr, "ref_harness: cannot write %s\n", path.c_str());
        exit(2);
    }
    fclose(f);
}

struct Out {
    int W, H;
    std::vector<int32_t> hit;   // H*W, row i = -y + H/2 - 1, col j = x + W/2
    std::vector<int32_t> rgb;   // H*W*3 unclamped Color ints
    std::vector<double> t;      // H*W primary closestT (INF on miss)
};

// One column strip, as the reference's render(scene, xMin, xMax) does it.
void strip(Scene scene, int xMin, int xMax, int depth, Out *out, bool with_ids) {
    const int W = out->W, H = out->H;
    for (int x = xMin; x < xMax; x++) {
        // y spans [-H/2, H/2) for even H (src/main.cpp:77); an odd H gets its bottom row too.
        for (int y = -(H - H / 2); y < H / 2; y++) {
            Vector3 ray = scene.toViewport(x, y, W, H)
                              .rotateX(scene.cameraRotation.x)
                              .rotateY(scene.cameraRotation.y)
                              .rotateZ(scene.cameraRotation.z);
            // the id/t dump costs one extra primary query per pixel; timing runs skip it
            tuple<bool, double, int> prim = {false, 0.0, -2};
            if (with_ids) {
#if USE_BVH
                prim = scene.bvh.intersectRay(scene.cameraPosition, ray, 1.0, INFINITY, 0);
#else
                prim = scene.closestIntersection(scene.cameraPosition, ray, 1.0, INFINITY);
#endif
            }
            Color c = scene.traceRay(scene.cameraPosition, ray, 1.0, INFINITY, depth);
            size_t i = static_cast<size_t>(-y + H / 2 - 1);
            size_t j = static_cast<size_t>(x + W / 2);
            size_t k = i * static_cast<size_t>(W) + j;
            out->hit[k] = std::get<2>(prim);
            out->t[k] = std::get<1>(prim);
            out->rgb[3 * k + 0] = c.r;
            out->rgb[3 * k + 1] = c.g;
            out->rgb[3 * k + 2] = c.b;
        }
    }
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr,
                "usage: %s <teapot|bunny|erato|dragon|aurelius> <dir holding scenes/NAME.obj> "
                "<W> <H> <depth> <out_prefix|-> [threads] [timing-only repeats]\n",
                argv[0]);
        return 2;
    }
    std::string name = argv[1];
    std::string root = argv[2];
    Out out;
    out.W = atoi(argv[3]);
    out.H = atoi(argv[4]);
    int depth = atoi(argv[5]);
    std::string prefix = argv[6];
    int threads = argc > 7 ? atoi(argv[7]) : 8;
    // > 0: render that many frames exactly as main() times them (no id query), print the seconds, dump nothing
    int timing_repeats = argc > 8 ? atoi(argv[8]) : 0;

    // out_prefix is resolved against the original cwd, the scene factories
    // against `root` (they hard-code "scenes/NAME.obj", include/scene.hpp:262).
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) return 2;
    if (prefix != "-" && prefix[0] != '/') prefix = std::string(cwd) + "/" + prefix;
    if (chdir(root.c_str()) != 0) {
        fprintf(stderr, "ref_harness: cannot chdir to %s\n", root.c_str());
        return 2;
    }

    Scene scene;
    try {
        if (name == "teapot") scene = Scene::teapot();
        else if (name == "bunny") scene = Scene::bunny();
        else if (name == "erato") scene = Scene::erato();
        else if (name == "dragon") scene = Scene::dragon();
        else if (name == "aurelius") scene = Scene::aurelius();
        else { fprintf(stderr, "unknown scene %s\n", name.c_str()); return 2; }
    } catch (const char *e) {
        std::cerr << e;
        return -1;
    }

    // REF_CAMERA="px,py,pz,rx,ry,rz" replaces the factory's camera (public members of Scene,
    // include/scene.hpp:17-18): fixtures with a turned camera, which the stock factories never set.
    if (const char *cam = getenv("REF_CAMERA")) {
        double v[6];
        if (sscanf(cam, "%lf,%lf,%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) != 6) {
            fprintf(stderr, "ref_harness: REF_CAMERA needs six numbers\n");
            return 2;
        }
        scene.cameraPosition = {v[0], v[1], v[2]};
        scene.cameraRotation = {v[3], v[4], v[5]};
    }

    // REF_SCENE_FILE=<text file> overrides more of the factory's scene through Scene's public members
    // (include/scene.hpp:15-23), one directive per line:
    //   camera px py pz rx ry rz | viewport x y z | background r g b | ambient a | nolights |
    //   point i x y z | directional i x y z | materials <file of {int32 r,g,b,spec; double refl} per triangle>
    // Lights given here are appended after `nolights` cleared the factory's.  Materials only touch
    // Scene::triangles, where traceRay reads them (scene.hpp:105-113); the BVH keeps its own copy of the geometry.
    if (const char *path = getenv("REF_SCENE_FILE")) {
        FILE *f = fopen(path, "r");
        if (!f) { fprintf(stderr, "ref_harness: cannot read %s\n", path); return 2; }
        char key[64];
        while (fscanf(f, "%63s", key) == 1) {
            std::string k = key;
            double v[6];
            bool ok = true;
            if (k == "camera") {
                ok = fscanf(f, "%lf %lf %lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6;
                scene.cameraPosition = {v[0], v[1], v[2]};
                scene.cameraRotation = {v[3], v[4], v[5]};
            } else if (k == "viewport") {
                ok = fscanf(f, "%lf %lf %lf", &v[0], &v[1], &v[2]) == 3;
                scene.viewport = {v[0], v[1], v[2]};
            } else if (k == "background") {
                int c[3];
                ok = fscanf(f, "%d %d %d", &c[0], &c[1], &c[2]) == 3;
                scene.backgroundColor = Color(c[0], c[1], c[2]);
            } else if (k == "ambient") {
                ok = fscanf(f, "%lf", &v[0]) == 1;
                scene.ambientLight = v[0];
            } else if (k == "nolights") {
                scene.pointLights.clear();
                scene.directionalLights.clear();
            } else if (k == "point") {
                ok = fscanf(f, "%lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3]) == 4;
                scene.pointLights.push_back({v[0], {v[1], v[2], v[3]}});
            } else if (k == "directional") {
                ok = fscanf(f, "%lf %lf %lf %lf", &v[0], &v[1], &v[2], &v[3]) == 4;
                scene.directionalLights.push_back({v[0], {v[1], v[2], v[3]}});
            } else if (k == "materials") {
                char mpath[4096];
                ok = fscanf(f, "%4095s", mpath) == 1;
                FILE *m = ok ? fopen(mpath, "rb") : nullptr;
                ok = m != nullptr;
                for (size_t i = 0; ok && i < scene.triangles.size(); i++) {
                    struct { int32_t r, g, b, spec; double refl; } rec;
                    ok = fread(&rec, sizeof rec, 1, m) == 1;
                    scene.triangles[i].color = Color(rec.r, rec.g, rec.b);
                    scene.triangles[i].specularity = rec.spec;
                    scene.triangles[i].reflectivity = rec.refl;
                }
                if (m) fclose(m);
            } else {
                ok = false;
            }
            if (!ok) { fprintf(stderr, "ref_harness: bad directive '%s' in %s\n", key, path); return 2; }
        }
        fclose(f);
    }

    size_t npx = static_cast<size_t>(out.W) * static_cast<size_t>(out.H);
    out.hit.assign(npx, -2);
    out.rgb.assign(npx * 3, 0);
    out.t.assign(npx, 0.0);

    const bool with_ids = timing_repeats == 0;
    for (int rep = 1; rep < timing_repeats; rep++) {
        auto a = std::chrono::high_resolution_clock::now();
        std::vector<std::thread> pool;
        int start = -out.W / 2, step = out.W / threads;
        for (int i = 0; i < threads; i++) {
            int x0 = start + i * step;
            int x1 = (i == threads - 1) ? (out.W - out.W / 2) : x0 + step;
            pool.emplace_back(strip, scene, x0, x1, depth, &out, false);
        }
        for (auto &th : pool) th.join();
        auto b = std::chrono::high_resolution_clock::now();
        printf("frame_seconds %.3f threads %d\n", std::chrono::duration<double>(b - a).count(), threads);
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    if (out.W > 0 && out.H > 0) {
        std::vector<std::thread> pool;
        int start = -out.W / 2;
        // W/threads columns per strip like src/main.cpp:38; the last strip
        // takes the remainder so that any W renders completely.
        int step = out.W / threads;
        for (int i = 0; i < threads; i++) {
            int x0 = start + i * step;
            int x1 = (i == threads - 1) ? (out.W - out.W / 2) : x0 + step;
            pool.emplace_back(strip, scene, x0, x1, depth, &out, with_ids);
        }
        for (auto &th : pool) th.join();
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    double secs = std::chrono::duration<double>(t1 - t0).count();

    std::vector<unsigned char> rgb8(npx * 3);
    for (size_t k = 0; k < npx; k++) {
        Color c = Color(out.rgb[3 * k], out.rgb[3 * k + 1], out.rgb[3 * k + 2]).normalize();
        rgb8[3 * k + 0] = static_cast<unsigned char>(c.r);
        rgb8[3 * k + 1] = static_cast<unsigned char>(c.g);
        rgb8[3 * k + 2] = static_cast<unsigned char>(c.b);
    }

    printf("mode %s\n", USE_BVH ? "bvh" : "brute");
    printf("triangles %zu\n", scene.triangles.size());
    printf("frame_seconds %.3f threads %d\n", secs, threads);
    printf("hit_fnv %016llx\n", static_cast<unsigned long long>(fnv1a(out.hit.data(), npx * 4)));
    printf("rgb_i32_fnv %016llx\n", static_cast<unsigned long long>(fnv1a(out.rgb.data(), npx * 12)));
    printf("rgb_u8_fnv %016llx\n", static_cast<unsigned long long>(fnv1a(rgb8.data(), npx * 3)));
    printf("t_fnv %016llx\n", static_cast<unsigned long long>(fnv1a(out.t.data(), npx * 8)));
#if USE_BVH
    printf("nodes_used %zu\n", scene.bvh.nodesUsed);
    static_assert(sizeof(BVHNode) == 64, "BVHNode layout");
    printf("nodes_fnv %016llx\n", static_cast<unsigned long long>(
                                      fnv1a(scene.bvh.nodes, scene.bvh.nodesUsed * sizeof(BVHNode))));
    printf("indices_fnv %016llx\n",
           static_cast<unsigned long long>(fnv1a(scene.bvh.indices.data(), scene.bvh.indices.size() * 8)));
#endif
    {
        // geometry as the Triangle ctor left it: v0 v1 v2 normal centroid (15 doubles)
        uint64_t h = 1469598103934665603ull;
        for (Triangle &tr : scene.triangles) {
            double g[15] = {tr.v0.x, tr.v0.y, tr.v0.z, tr.v1.x, tr.v1.y, tr.v1.z, tr.v2.x, tr.v2.y,
                            tr.v2.z, tr.normal.x, tr.normal.y, tr.normal.z, tr.centroid.x,
                            tr.centroid.y, tr.centroid.z};
            h = fnv1a(g, sizeof g, h);
        }
        printf("geometry_fnv %016llx\n", static_cast<unsigned long long>(h));
    }

    if (prefix != "-" && with_ids) {
        dump(prefix + ".hit.i32", out.hit.data(), npx * 4);
        dump(prefix + ".rgb.i32", out.rgb.data(), npx * 12);
        dump(prefix + ".rgb.u8", rgb8.data(), npx * 3);
        dump(prefix + ".t.f64", out.t.data(), npx * 8);
#if USE_BVH
        dump(prefix + ".nodes.bin", scene.bvh.nodes, scene.bvh.nodesUsed * sizeof(BVHNode));
        dump(prefix + ".indices.u64", scene.bvh.indices.data(), scene.bvh.indices.size() * 8);
#endif
    }
    return 0;
}
