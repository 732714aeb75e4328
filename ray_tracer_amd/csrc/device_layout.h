// Scene layout in HBM and the kernel argument blocks, shared by the host code that
// flattens a scene (rtx_api.cpp) and the kernels (rtx_kernels.hip).
//
// Derived from the reference's BVHNode (include/bvh.hpp:25-30), Triangle
// (include/triangle.hpp:8-15) and BVH::indices (include/bvh.hpp:34):
//
//  * nodes: 64-byte records.  Device slot = reference node index + 1, so the root
//    is slot 1 and every sibling pair (the reference allocates children as
//    nodesUsed++, nodesUsed++; bvh.hpp:148-149) sits at slots (2k, 2k+1), i.e. in one
//    128-byte line.  Slot 0 is unused.
//  * triangles: 128-byte records stored in BVH::indices order, so a leaf's triangles
//    are contiguous and `first` of a leaf addresses this array directly.  Each record
//    carries the original triangle index (what the reference reports as the hit id)
//    and its leaf's `first`, which decides equal-t ties between leaves.
//  * materials: 32-byte records in the same order.
#ifndef RTX_DEVICE_LAYOUT_H
#define RTX_DEVICE_LAYOUT_H

#include <stdint.h>

namespace rtx {

// How the walk refers to a node it may enter next, in one word -- the form its traversal stack holds, so that entering
// a child and entering a node that comes off the stack are the same few instructions:
//   bit 31 = 0: an inner node, bits 30..0 = slot of its left child (or kTopFlag | pair index, see below)
//   bit 31 = 1: a leaf, bits 30..28 = count - 1 (1..7 triangles), bits 27..0 = first triangle position; the value 7 in
//               bits 30..28 marks a bigger leaf, then bits 27..0 hold the node's own slot and its header is read.
// (Triangle positions and node slots stay below 2^28; checked when the scene is created.)
constexpr uint32_t node_entry(uint32_t slot, uint32_t first, uint32_t count) {
    return count == 0u ? first : (count <= 7u ? (0x80000000u | ((count - 1u) << 28) | first) : (0xF0000000u | slot));
}

struct alignas(64) DevNode {
    double bmin[3];
    double bmax[3];
    uint32_t first;   // inner: slot of the left child (right = first + 1); leaf: first triangle position
    uint32_t count;   // 0 = inner
    // the left node of a sibling pair (wherever an inner node's `first` points): node_entry() of itself and of its
    // sibling, so that a pair test reads 112 contiguous bytes (this record and the sibling's box) and finds both
    // children's references ready-made; other nodes: unused
    uint32_t pad[2];
};
static_assert(sizeof(DevNode) == 64, "DevNode");

// bit 30 of a node reference: a pair of the top table (bits 15..0 = its index), not a slot of `nodes`
constexpr uint32_t kTopFlag = 0x40000000u;
// pairs the table may hold at most (its queue lives in the build kernel's LDS)
constexpr uint32_t kTopPairsMax = 512;

// Members in the order the intersection test uses them: the plane (n, d) first, so that the distance along the ray is
// being computed while the vertices are still on their way.
struct alignas(128) DevTri {
    double n[3];       // Triangle::normal, computed on the host (triangle.hpp:21-23)
    double d;          // -(n . v0), the per-call constant of triangle.hpp:39
    double v0[3], v1[3], v2[3];
    uint32_t orig_id;  // index into the reference's Scene::triangles
    uint32_t leaf_first;
    double pad[2];
};
static_assert(sizeof(DevTri) == 128, "DevTri");

struct alignas(32) DevMat {
    int32_t color[3];
    int32_t specularity;
    double reflectivity;
    double pad;
};
static_assert(sizeof(DevMat) == 32, "DevMat");

struct alignas(32) DevLight {
    double intensity;
    double v[3];
};

struct DevScene {
    const DevNode *nodes;
    const DevTri *tris;
    const DevMat *mats;
    const DevLight *lights;   // point lights first, then directional
    // The top of the tree once more, for the render kernel's LDS: `n_top_pairs` sibling pairs
    // in breadth-first order (pair 0 = slots 0 and 1), 128 bytes each.  Inside it an inner
    // child whose own pair is in the table carries kTopFlag | pair index as `first` (and as
    // its reference in the left record's padding).
    const DevNode *top;
    uint32_t n_top_pairs, top_pad;
    uint32_t n_nodes, n_tris, n_point, n_dir;
    double cam[3];
    double viewport[3];
    // cos/sin of the three camera rotations, evaluated on the host with libm exactly
    // as Vector3::rotateX/Y/Z do per pixel (include/vector3.hpp:54-82)
    double cx, sx, cy, sy, cz, sz;
    double ambient;
    double scene_scale;       // largest |coordinate| of the root box
    int32_t bg[3];
    int32_t bvh_depth;
    // 1 when every box coordinate, the camera and the light positions are 0 or within
    // 2^-300..2^300 and every box has min <= max: the slab test may divide through RayDiv
    int32_t fast_div_ok;
};

enum : int32_t { SP_PRIMARY = 0, SP_REFLECT = 1, SP_LIGHT = 2, SP_RESOLVE = 3 };

struct DevRender {
    int32_t W, H, x0, x1, y0, y1;
    int32_t depth, spp;
    uint64_t seed;
    int32_t tile_w, tile_h, tile_rank, tile_count;
    int32_t tiles_x, tiles_y, owned_tiles, layout;
    double px, py;            // viewport.x / W, viewport.y / H  (scene.hpp:83-84)
    uint8_t *rgb;
    int32_t *hit;
    int32_t *rgb_i32;
    double *t;
    unsigned long long *stats; // 15 counters, rtx_stats order
    uint32_t stack_stride;     // LDS stack entries per level (= block size)
    // per-lane scratch (see Scratch in rtx_kernels.hip): field-major, `scratch_lanes` lanes per field
    double *scratch_f64;
    int32_t *scratch_i32;
    uint64_t scratch_lanes;
    // work queues: queue q owns the local tiles tile_order[9 + tile_order[q] .. 9 + tile_order[q+1]);
    // work[32*q] counts the 8x8 blocks handed out from it (zeroed before every launch)
    uint32_t *work;
    const uint32_t *tile_order;
    uint32_t n_queues;
    uint32_t total_blocks;     // 8x8 blocks of this launch
    int32_t w_inner, w_leaf, w_shade;   // weights of the wave's step choice
    int32_t tail_lanes;                 // at most this many live lanes: run every kind of step each round
    int32_t also_lanes;                 // a kind with at least this many waiting lanes runs in every round
    int32_t also_leaf, also_shade;      // (inner node steps: also_lanes; triangle steps: also_leaf; shading: also_shade)
    // several frames in one launch (rtx_render_device_batch): queue q hands out its blocks frame after frame; frame f
    // writes its pixels f * frame_pixels further on, through camera cams[f] = {position[3], cos x, sin x, cos y, sin y,
    // cos z, sin z} when has_cams is set (else every frame has the scene's camera)
    int32_t n_frames, has_cams;
    uint32_t frame_pixels, frames_pad;
    double cams[8][9];
    int32_t lighting_jobs;              // 1: lanes without work take over the lighting of other lanes' surface hits at the end of a launch
    // The staged pipeline (rtx_render_device_staged): a frame is a sequence of launches of the same kernel, each of which
    // serves the queues' blocks once per pass; a lane's unit of work is then ONE query chain of ONE pixel, not the pixel:
    //   SP_PRIMARY  the camera ray's closest-hit query                                    (launch 0)
    //   SP_REFLECT  the closest-hit query of level `stage`, for pixels whose level stage-1 hit reflects
    //   SP_LIGHT    the shadow queries and lighting sum of the level stage-1 hit
    //   SP_RESOLVE  the blend of a pixel's levels, innermost first, and putPixel           (last launch)
    // Between launches a pixel lives in the record arrays below, indexed by the pixel's place in the output buffers.
    int32_t stage, n_passes;
    int32_t pass_kind[2];
    uint64_t rec_pixels;                // P: entries per record array (= rtx_output_pixels of the launch)
    int32_t *rec_pos;                   // [depth + 1][P]  triangle position of the level's hit (written for hits only)
    // [depth + 1][total_blocks]: bit (row & 7) * 8 + (col & 7) of word b of level l = that pixel of 8x8 block b (local tile x
    // blocks per tile + block in tile) has a hit at level l.  All zero before a frame; the last launch leaves it so.
    unsigned long long *hitmask;
    uint32_t draw_run;                  // blocks a wave draws from a queue with one atomic
    int32_t refill_min;                 // lanes without work are given new items when this many of them wait
    double *rec_pt, *rec_ray;           // [2][3][P]       hit point and incoming ray of a level, levels alternate between the halves
    int32_t *lvl_local;                 // [depth + 1][3][P]  localColor of a level (colour x lighting, truncated: scene.hpp:113)
    double *lvl_refl;                   // [depth + 1][P]  reflectivity of the level's triangle
    unsigned long long *wave_log;       // diagnostics, or NULL: per wave {start, end, latest block drawn} in 100 MHz ticks and that block (x0 | y0 << 16 | queue << 32)
    unsigned long long *ticks;          // {earliest block start, latest block end} in 100 MHz ticks, or NULL
};

}  // namespace rtx
#endif
