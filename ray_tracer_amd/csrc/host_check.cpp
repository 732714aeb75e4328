// Command-line checker over the host-only part of the library (OBJ loaders, scene presets, stand-in
// generators, the scalar BVH builder, BVH validation): no HIP, so it builds with plain g++ -- and with
// -fsanitize=address,undefined (`make asan`), which is what it is for: the loaders scan mmapped text with
// hand-written pointer code.  Prints counts and FNV-1a hashes the tests compare with the shipped library's.
//   host_check parse strict|tolerant FILE PRESET
//   host_check standin PRESET N SEED blob|knot
//   host_check poisoned FILE PRESET        (the file's own tree padded to 2n-1 records of garbage, then broken ones)
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <string>

#include "host_scene.hpp"

static uint64_t fnv1a(const void *data, size_t bytes, uint64_t h = 1469598103934665603ull) {   // the basis the test helpers use
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

static int finish(rtx::HostScene &s, const std::string &preset) {
    std::string err;
    if (rtx::apply_preset(&s, preset, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 3; }
    rtx::setup_triangles(&s, false);
    if (rtx::build_bvh(&s, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 4; }
    printf("triangles %zu\n", (size_t)s.n_triangles());
    printf("vertices_fnv %016" PRIx64 "\n", fnv1a(s.vertices.data(), s.vertices.size() * 8));
    printf("normals_fnv %016" PRIx64 "\n", fnv1a(s.normals.data(), s.normals.size() * 8));
    printf("materials_fnv %016" PRIx64 "\n", fnv1a(s.reflectivity.data(), s.reflectivity.size() * 8, fnv1a(s.colors.data(), s.colors.size() * 4)));
    printf("nodes %zu\n", s.nodes.size());
    printf("nodes_fnv %016" PRIx64 "\n", fnv1a(s.nodes.data(), s.nodes.size() * sizeof(rtx_bvh_node)));
    printf("indices_fnv %016" PRIx64 "\n", fnv1a(s.indices.data(), s.indices.size() * 8));
    printf("depth %d\n", s.bvh_depth);
    return 0;
}

int main(int argc, char **argv) {
    const std::string cmd = argc > 1 ? argv[1] : "";
    std::string err;
    rtx::HostScene s;
    if (cmd == "parse" && argc == 5) {
        const bool tolerant = !strcmp(argv[2], "tolerant");
        const int rc = tolerant ? rtx::parse_obj_tolerant(argv[3], &s.vertices, &err) : rtx::parse_obj(argv[3], &s.vertices, &err);
        if (rc) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
        return finish(s, argv[4]);
    }
    if (cmd == "standin" && argc == 6) {
        std::vector<double> verts;
        std::vector<uint32_t> faces;
        if (rtx::generate_standin(argv[2], strtoull(argv[3], nullptr, 10), strtoull(argv[4], nullptr, 10), !strcmp(argv[5], "knot"), &verts, &faces, &err)) {
            fprintf(stderr, "%s\n", err.c_str());
            return 2;
        }
        for (uint32_t v : faces) s.vertices.insert(s.vertices.end(), &verts[3 * (size_t)v], &verts[3 * (size_t)v] + 3);
        return finish(s, argv[2]);
    }
    if (cmd == "poisoned" && argc == 4) {
        if (rtx::parse_obj(argv[2], &s.vertices, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
        if (int rc = finish(s, argv[3])) return rc;
        const std::vector<rtx_bvh_node> good = s.nodes;
        const size_t n = s.n_triangles();
        // the reference's allocation: 2n-1 records, the ones past nodesUsed never written (bvh.hpp:91)
        s.nodes.resize(2 * n - 1);
        for (size_t k = good.size(); k < s.nodes.size(); k++) memset(&s.nodes[k], 0xA5, sizeof(rtx_bvh_node));
        int rc = rtx::validate_bvh(&s, &err);
        printf("padded %d nodes %zu same %d\n", rc, s.nodes.size(), (int)(s.nodes.size() == good.size() && !memcmp(s.nodes.data(), good.data(), good.size() * sizeof(rtx_bvh_node))));
        int refused = 0, cases = 0;
        for (size_t k = 0; k < good.size(); k++) {
            if (good[k].count != 0) continue;
            for (uint64_t bad : {UINT64_MAX, (uint64_t)good.size() - 1, (uint64_t)0, (uint64_t)k}) {
                s.nodes = good;
                s.nodes[k].first = bad;
                cases++;
                refused += rtx::validate_bvh(&s, &err) != 0;
            }
            if (cases >= 400) break;
        }
        for (size_t k = 0; k < good.size() && cases < 800; k++) {
            if (good[k].count == 0) continue;
            for (uint64_t bad_first : {(uint64_t)n, (uint64_t)n - 1, UINT64_MAX}) {
                s.nodes = good;
                s.nodes[k].first = bad_first;
                s.nodes[k].count = 2;
                cases++;
                refused += rtx::validate_bvh(&s, &err) != 0;
            }
        }
        printf("broken %d refused %d\n", cases, refused);
        return 0;
    }
    fprintf(stderr, "usage: host_check parse strict|tolerant FILE PRESET | standin PRESET N SEED blob|knot | poisoned FILE PRESET\n");
    return 1;
}
