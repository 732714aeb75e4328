// Host side of the render path: what the reference keeps in `Scene` and `BVH`
// (include/scene.hpp:15-23, include/bvh.hpp:32-37) as flat arrays, the OBJ loader,
// the five scene factories and the binned-SAH BVH builder.  Compiled with
// -ffp-contract=off: the builder must reproduce the reference's tree bit for bit,
// because tree topology decides which triangle wins an equal-t tie.
#ifndef RTX_HOST_SCENE_HPP
#define RTX_HOST_SCENE_HPP

#include <cstdint>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "rtx.h"

namespace rtx {

// Allocator whose resize() leaves new elements unwritten: the big per-triangle arrays are filled
// right after by worker threads, which then also take the first-touch page faults in parallel.
template <typename T>
struct DefaultInit : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInit<U>; };
    DefaultInit() = default;
    template <typename U> DefaultInit(const DefaultInit<U> &) {}
    template <typename U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
using Doubles = std::vector<double, DefaultInit<double>>;
using Ints = std::vector<int32_t, DefaultInit<int32_t>>;

struct HostScene {
    // per triangle, index = the reference's index into Scene::triangles
    Doubles vertices;       // 9 per triangle
    Doubles normals;        // 3
    Doubles centroids;      // 3; filled by build_bvh when it needs them
    Ints colors;            // 3
    Ints specularity;
    Doubles reflectivity;
    // BVH
    std::vector<rtx_bvh_node> nodes; // nodesUsed entries
    std::vector<uint64_t> indices;
    int32_t bvh_depth = 0;
    double bvh_seconds = 0.0;
    // camera & lights
    double viewport[3] = {1, 1, 1};
    double camera_position[3] = {0, 0, 0};
    double camera_rotation_deg[3] = {0, 0, 0};
    int32_t background[3] = {255, 255, 255};
    double ambient = 0.0;
    std::vector<rtx_light> point_lights, directional_lights;

    uint64_t n_triangles() const { return specularity.size(); }
};

// Each returns an rtx_status and leaves the message in `err`.
int parse_obj(const std::string &path, Doubles *vertices9, std::string *err);
// General OBJ text: what tools/denormalize.py followed by Scene::parseOBJ accepts, each face once.
int parse_obj_tolerant(const std::string &path, Doubles *vertices9, std::string *err);
int apply_preset(HostScene *hs, const std::string &preset, std::string *err);
// Triangle ctor work (include/triangle.hpp:21-24) for triangles lacking it.
void setup_triangles(HostScene *hs, bool have_normals);
int build_bvh(HostScene *hs, std::string *err);
int32_t measure_bvh_depth(const std::vector<rtx_bvh_node> &nodes);
int validate_bvh(HostScene *hs, std::string *err);

// memcpy spread over the worker threads (a 4K frame is 25 MB: one thread takes longer than the GPU takes to render it).
void parallel_copy(void *dst, const void *src, size_t bytes);

// Stand-in mesh generator (scenegen.cpp): vertices9 of n_triangles faces,
// the last 10 being the open box.
// family 0: a bumpy star-shaped blob in a box around z = 0; family 1: a knotted tube in a box that fills the view.
int generate_standin(const std::string &preset, uint64_t n_triangles, uint64_t seed, int family,
                     std::vector<double> *vertex_list, std::vector<uint32_t> *faces, std::string *err);
int write_obj(const std::string &path, const std::vector<double> &vertex_list,
              const std::vector<uint32_t> &faces, std::string *err);

}  // namespace rtx
#endif
