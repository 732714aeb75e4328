// Host side of the render path: what the reference keeps in `Scene` and `BVH`
// (include/scene.hpp:15-23, include/bvh.hpp:32-37) as flat arrays, the OBJ loader,
// the five scene factories and the binned-SAH BVH builder.  Compiled with
// -ffp-contract=off: the builder must reproduce the reference's tree bit for bit,
// because tree topology decides which triangle wins an equal-t tie.
#ifndef RTX_HOST_SCENE_HPP
#define RTX_HOST_SCENE_HPP

#include <cstdint>
#include <string>
#include <vector>

#include "rtx.h"

namespace rtx {

struct HostScene {
    // per triangle, index = the reference's index into Scene::triangles
    std::vector<double> vertices;    // 9 per triangle
    std::vector<double> normals;     // 3
    std::vector<double> centroids;   // 3
    std::vector<int32_t> colors;     // 3
    std::vector<int32_t> specularity;
    std::vector<double> reflectivity;
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
int parse_obj(const std::string &path, std::vector<double> *vertices9, std::string *err);
// General OBJ text: what tools/denormalize.py followed by Scene::parseOBJ accepts, each face once.
int parse_obj_tolerant(const std::string &path, std::vector<double> *vertices9, std::string *err);
int apply_preset(HostScene *hs, const std::string &preset, std::string *err);
// Triangle ctor work (include/triangle.hpp:21-24) for triangles lacking it.
void setup_triangles(HostScene *hs, bool have_normals);
int build_bvh(HostScene *hs, std::string *err);
int32_t measure_bvh_depth(const std::vector<rtx_bvh_node> &nodes);
int validate_bvh(const HostScene &hs, std::string *err);

// Stand-in mesh generator (scenegen.cpp): vertices9 of n_triangles faces,
// the last 10 being the open box.
int generate_standin(const std::string &preset, uint64_t n_triangles, uint64_t seed,
                     std::vector<double> *vertex_list, std::vector<uint32_t> *faces, std::string *err);
int write_obj(const std::string &path, const std::vector<double> &vertex_list,
              const std::vector<uint32_t> &faces, std::string *err);

}  // namespace rtx
#endif
