// GPU builder for the reference's binned-SAH BVH (include/bvh.hpp:84-162,236-305).
#ifndef RTX_BVH_BUILD_H
#define RTX_BVH_BUILD_H

#include <cstdint>
#include <string>

#include "rtx.h"

namespace rtx {

// Builds on the current HIP device from 9 doubles per triangle (host memory).
// nodes_out holds 2n-1 records, indices_out n.  device_ms_out: time between the end of the
// upload and the end of the last kernel.
int build_bvh_device(const double *vertices, uint64_t n_triangles, rtx_bvh_node *nodes_out, uint64_t *indices_out,
                     uint64_t *n_nodes_out, int32_t *depth_out, double *device_ms_out, std::string *err);

}  // namespace rtx
#endif
