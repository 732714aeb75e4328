// GPU builder for the reference's binned-SAH BVH (include/bvh.hpp:84-162,236-305) and the
// device-side conversion of its result to the render kernel's layout.
#ifndef RTX_BVH_BUILD_H
#define RTX_BVH_BUILD_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <string>

#include "device_layout.h"
#include "rtx.h"

namespace rtx {

// A tree that stays on the device it was built on; owns the builder's buffers.
struct DeviceBvh {
    struct Impl;
    const rtx_bvh_node *nodes = nullptr;   // [n_nodes], the reference's records and numbering
    const uint32_t *indices = nullptr;     // [n_triangles], BVH::indices
    const uint64_t *indices64 = nullptr;   // the same, widened
    const double *vertices = nullptr;      // [9 n_triangles], the uploaded copy
    uint64_t n_triangles = 0, n_nodes = 0;
    int32_t depth = 0;
    double device_ms = 0;                  // from the end of the upload to the end of the last kernel
    Impl *impl = nullptr;
    DeviceBvh();
    ~DeviceBvh();
    DeviceBvh(const DeviceBvh &) = delete;
    DeviceBvh &operator=(const DeviceBvh &) = delete;
};

// Ordinary (pageable) host memory -> device memory of the current device at the rate of pinned memory: the bytes pass
// through a small ring of pinned buffers kept by the process, the copy into the ring overlapping the previous piece's DMA
// (the runtime's own path for pageable memory reaches half the link's rate).  Ordered on `stream`; returns when the
// last piece has landed.  Callers take turns: a second upload waits for the first.
hipError_t upload_pageable(void *dst, const void *src, size_t bytes, hipStream_t stream);

// Builds on the current HIP device from 9 doubles per triangle (host memory).
// *vertices_up (optional) is set to 1 once the vertices are on the device -- or the build has failed before --
// so that a caller's other uploads can queue behind them.
int build_bvh_resident(const double *vertices, uint64_t n_triangles, DeviceBvh *out, std::string *err,
                       std::atomic<int> *vertices_up = nullptr);

// The same, copied to host arrays: nodes_out holds 2n-1 records, indices_out n.
int build_bvh_device(const double *vertices, uint64_t n_triangles, rtx_bvh_node *nodes_out, uint64_t *indices_out,
                     uint64_t *n_nodes_out, int32_t *depth_out, double *device_ms_out, std::string *err);

// Writes DevNode[n_nodes + 1], DevTri[n] and DevMat[n] (device memory of the caller) from
// a resident tree and per-triangle arrays already on the device (d_normals may be NULL:
// the Triangle ctor's normal is computed).  tame_out: 1 when every box admits the fast
// slab division; root_box: bmin, bmax of the root.
int flatten_scene_device(const DeviceBvh &bvh, const double *d_normals, const int32_t *d_colors, const int32_t *d_spec,
                         const double *d_refl, DevNode *d_nodes, DevTri *d_tris, DevMat *d_mats, int32_t *tame_out,
                         double root_box[6], std::string *err);

}  // namespace rtx
#endif
