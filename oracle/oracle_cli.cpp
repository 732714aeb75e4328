// TEST INFRASTRUCTURE ONLY -- command-line front end of the oracle (rt_oracle.h).
// Prints the same "key value" lines as oracle/ref_harness.cpp so the two can be
// diffed, plus the work census; also times the render for bench.py's cpu_baseline.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rt_oracle.h"

static void dump(const std::string &path, const void *data, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(data, 1, n, f) != n) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s <preset> <file.obj> <W> <H> <depth> <out_prefix|-> [threads] [bvh=1] [spp=1] [repeats=1]\n", argv[0]);
        return 2;
    }
    std::string preset = argv[1], obj = argv[2], prefix = argv[6];
    int W = atoi(argv[3]), H = atoi(argv[4]), depth = atoi(argv[5]);
    int threads = argc > 7 ? atoi(argv[7]) : 8;
    int use_bvh = argc > 8 ? atoi(argv[8]) : 1;
    int spp = argc > 9 ? atoi(argv[9]) : 1;
    int repeats = argc > 10 ? atoi(argv[10]) : 1;
    auto b0 = std::chrono::high_resolution_clock::now();
    ora_scene *s = ora_scene_load_obj(obj.c_str(), preset.c_str());
    if (!s) { fprintf(stderr, "%s", ora_last_error()); return -1; }
    auto b1 = std::chrono::high_resolution_clock::now();
    size_t npx = (size_t)W * H;
    std::vector<int32_t> hit(npx), rgb(npx * 3);
    std::vector<uint8_t> rgb8(npx * 3);
    std::vector<double> t(npx);
    ora_census c;
    printf("mode %s\n", use_bvh ? "bvh" : "brute");
    printf("triangles %lld\n", (long long)ora_num_triangles(s));
    printf("load_build_seconds %.3f\n", std::chrono::duration<double>(b1 - b0).count());
    for (int r = 0; r < repeats; r++) {
        auto t0 = std::chrono::high_resolution_clock::now();
        if (ora_render(s, W, H, depth, spp, 0, threads, use_bvh, hit.data(), rgb.data(), rgb8.data(), t.data(), &c)) {
            fprintf(stderr, "%s\n", ora_last_error());
            return -1;
        }
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("frame_seconds %.3f threads %d\n", std::chrono::duration<double>(t1 - t0).count(), threads);
    }
    printf("hit_fnv %016llx\n", (unsigned long long)ora_fnv1a(hit.data(), npx * 4));
    printf("rgb_i32_fnv %016llx\n", (unsigned long long)ora_fnv1a(rgb.data(), npx * 12));
    printf("rgb_u8_fnv %016llx\n", (unsigned long long)ora_fnv1a(rgb8.data(), npx * 3));
    printf("t_fnv %016llx\n", (unsigned long long)ora_fnv1a(t.data(), npx * 8));
    size_t M = ora_num_nodes(s), N = ora_num_triangles(s);
    std::vector<char> nodes(M * 64);
    std::vector<uint64_t> idx(N);
    std::vector<double> geo(N * 15);
    ora_get_nodes(s, nodes.data());
    ora_get_indices(s, idx.data());
    ora_get_geometry(s, geo.data());
    printf("nodes_used %zu\n", M);
    printf("nodes_fnv %016llx\n", (unsigned long long)ora_fnv1a(nodes.data(), nodes.size()));
    printf("indices_fnv %016llx\n", (unsigned long long)ora_fnv1a(idx.data(), N * 8));
    printf("geometry_fnv %016llx\n", (unsigned long long)ora_fnv1a(geo.data(), geo.size() * 8));
    printf("max_depth %d\n", ora_max_depth(s));
    unsigned long long rays = c.primary_rays + c.reflection_rays + c.shadow_rays;
    printf("census primary %llu reflection %llu shadow %llu occluded %llu closest_hits %llu rays %llu\n",
           (unsigned long long)c.primary_rays, (unsigned long long)c.reflection_rays,
           (unsigned long long)c.shadow_rays, (unsigned long long)c.shadow_occluded,
           (unsigned long long)c.closest_hits, rays);
    printf("census node_visits %llu box_hits %llu leaf_visits %llu triangle_tests %llu\n",
           (unsigned long long)c.node_visits, (unsigned long long)c.box_hits,
           (unsigned long long)c.leaf_visits, (unsigned long long)c.triangle_tests);
    if (rays)
        printf("per_ray node_visits %.3f triangle_tests %.3f bytes %.1f\n", (double)c.node_visits / rays,
               (double)c.triangle_tests / rays, ((double)c.node_visits * 64 + (double)c.triangle_tests * 96) / rays);
    if (prefix != "-") {
        dump(prefix + ".hit.i32", hit.data(), npx * 4);
        dump(prefix + ".rgb.i32", rgb.data(), npx * 12);
        dump(prefix + ".rgb.u8", rgb8.data(), npx * 3);
        dump(prefix + ".t.f64", t.data(), npx * 8);
        dump(prefix + ".nodes.bin", nodes.data(), nodes.size());
        dump(prefix + ".indices.u64", idx.data(), N * 8);
    }
    ora_scene_free(s);
    return 0;
}
