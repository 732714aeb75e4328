// Launch wrappers of rtx_kernels.hip, called by the C ABI (rtx_api.cpp).
#ifndef RTX_KERNELS_H
#define RTX_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_layout.h"

namespace rtx {

// must equal RTX_LAYOUT_PACKED_TILES in include/rtx.h (checked in rtx_api.cpp)
constexpr int RTX_LAYOUT_PACKED_TILES_ = 1;

// LDS bytes a render launch needs per block: one u32 traversal-stack slot per tree level per lane.
inline size_t render_lds_bytes(int32_t bvh_depth, uint32_t block_threads) {
    return (size_t)(bvh_depth + 1) * block_threads * sizeof(uint32_t);
}
// waves of a render launch: one per 8x8-pixel block of every owned tile
inline uint64_t render_waves(const DevRender &rp) {
    return (uint64_t)(rp.tile_w / 8) * (uint64_t)(rp.tile_h / 8) * (uint64_t)rp.owned_tiles;
}
// scratch a render launch needs: per resident lane (4 + depth) doubles and (1 + 3 * depth) ints,
// after 1 KiB for the work counters
inline uint64_t render_scratch_bytes(uint64_t lanes, int32_t depth) {
    return 1024 + lanes * ((4 + (uint64_t)depth) * 8 + (1 + 3 * (uint64_t)depth) * 4);
}

// 256-thread blocks of the render kernel one CU keeps resident (= its waves per SIMD)
int render_blocks_per_cu();
hipError_t launch_render(const DevScene &sc, const DevRender &rp, bool prune, bool stats, uint32_t block_threads,
                         uint32_t grid_blocks, hipStream_t st);
hipError_t launch_divtest(uint64_t seed, uint64_t n, unsigned long long *d_mismatches, hipStream_t st);
hipError_t launch_sqrttest(const double *d_x, double *d_out, uint64_t n, hipStream_t st);
hipError_t launch_powtest(const double *d_x, const int32_t *d_e, double *d_out, uint64_t n, hipStream_t st);
hipError_t launch_untile(const uint8_t *src, uint8_t *dst, int W, int H, int tile_w, int tile_h, int tiles_x, int n_ranks,
                         uint64_t rank_stride, int bpp, hipStream_t st);

}  // namespace rtx
#endif
