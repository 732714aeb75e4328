// Launch wrappers of rtx_kernels.hip, called by the C ABI (rtx_api.cpp).
#ifndef RTX_KERNELS_H
#define RTX_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "device_layout.h"

namespace rtx {

// must equal RTX_LAYOUT_PACKED_TILES in include/rtx.h (checked in rtx_api.cpp)
constexpr int RTX_LAYOUT_PACKED_TILES_ = 1;

// LDS bytes of a render block in front of its copy of the top of the tree: the traversal stacks, one u32 slot per tree
// level per lane; a row in which a lane counts the parties still at work on its pixel (see the lighting jobs in
// render_kernel); a row that holds the triangle whose lighting the lane is working out; 256 bytes of lights.
inline size_t render_lds_bytes(int32_t bvh_depth, uint32_t block_threads) {
    return (size_t)(bvh_depth + 3) * block_threads * sizeof(uint32_t) + 256;
}
// Sibling pairs of the top of the tree a render block keeps in LDS behind its stacks: what is left of
// the block's share of a CU's 160 KiB (render_blocks_per_cu() blocks of 256 threads, or their
// equivalent in smaller blocks, are resident), 128 bytes a pair.
inline uint32_t render_top_pairs(int32_t bvh_depth, uint32_t block_threads, int blocks_per_cu) {
    const size_t share = (size_t)163840 / (size_t)blocks_per_cu * block_threads / 256;
    const size_t stack = render_lds_bytes(bvh_depth, block_threads);
    if (stack + 128 > share) return 0;
    const size_t pairs = (share - stack) / 128;
    return (uint32_t)(pairs < kTopPairsMax ? pairs : kTopPairsMax);
}
// Builds the table (DevScene::top) from the flattened nodes; *d_count receives the pairs written (<= max_pairs).
hipError_t launch_top_table(const DevNode *d_nodes, DevNode *d_top, uint32_t max_pairs, uint32_t *d_count, hipStream_t st);
// waves of a render launch: one per 8x8-pixel block of every owned tile
inline uint64_t render_waves(const DevRender &rp) {
    return (uint64_t)(rp.tile_w / 8) * (uint64_t)(rp.tile_h / 8) * (uint64_t)rp.owned_tiles;
}

// The scratch buffer of a render launch: 1 KiB of work counters, the work-queue table
// (9 + owned tiles + 2 x macro tiles words, see tile_order_kernel), then the per-lane arrays.
constexpr uint64_t kScratchHeaderBytes = 1024;
// Word of that header in which the waves of a launch count themselves out (the queue counters are words 0, 32, ..., 224);
// the last one zeroes the counters and this word again, so the header must be zero only before a slot's FIRST launch.
constexpr uint32_t kWavesOutWord = 16;
inline uint64_t render_order_words(const DevRender &rp, int macro) {
    const uint64_t macros = (uint64_t)((rp.tiles_x + macro - 1) / macro) * (uint64_t)((rp.tiles_y + macro - 1) / macro);
    return 9 + (uint64_t)rp.owned_tiles + 2 * macros;
}
inline uint64_t render_order_bytes(const DevRender &rp, int macro) { return (render_order_words(rp, macro) * 4 + 255) / 256 * 256; }
// per-lane scratch (struct Scratch in rtx_kernels.hip): doubles, then 32-bit words, each field-major over the resident lanes
inline uint64_t render_lane_f64s(int32_t depth) { return 5 + 2 * (uint64_t)depth; }
inline uint64_t render_lane_i32s(int32_t depth) { return 10 + 3 * (uint64_t)depth; }
inline uint64_t render_lane_bytes(int32_t depth) { return render_lane_f64s(depth) * 8 + render_lane_i32s(depth) * 4; }

// Size of the file-scope device array behind rtx_render_device_noalloc, per GPU.
constexpr uint64_t kNoallocScratchBytes = 48ull << 20;
// Device address of that array on `device`, which must be current (looked up once per device).
hipError_t noalloc_scratch(int device, void **ptr);

// Builds the work-queue table of `rp` at d_order (render_order_words() words) on the stream.
// clear_counters: the kernel also zeroes the launch's work counters (rp.work, kScratchHeaderBytes), in place of a memset.
hipError_t launch_tile_order(uint32_t *d_order, const DevRender &rp, int macro, int n_queues, int centre_first, bool clear_counters,
                             hipStream_t st);

// 256-thread blocks of the render kernel one CU keeps resident (= its waves per SIMD)
int render_blocks_per_cu();
// staged: one launch of the staged pipeline (rp.stage, rp.n_passes, rp.pass_kind and the record arrays set)
hipError_t launch_render(const DevScene &sc, const DevRender &rp, bool prune, bool stats, uint32_t block_threads,
                         uint32_t grid_blocks, hipStream_t st, bool staged = false);
// bytes of a staged frame's record arrays for P output pixels (DevRender::rec_pos ... lvl_refl)
inline uint64_t staged_record_bytes(uint64_t P, int32_t depth) {
    const uint64_t levels = (uint64_t)depth + 1;
    return P * (levels * 4 + 2 * 2 * 3 * 8 + levels * 3 * 4 + levels * 8);
}
// and of its per-block hit masks (DevRender::hitmask), which sit in front of the records
inline uint64_t staged_mask_bytes(uint64_t total_blocks, int32_t depth) { return ((uint64_t)depth + 1) * total_blocks * 8; }
hipError_t launch_divtest(uint64_t seed, uint64_t n, unsigned long long *d_mismatches, hipStream_t st);
hipError_t launch_sqrttest(const double *d_x, double *d_out, uint64_t n, hipStream_t st);
hipError_t launch_powtest(const double *d_x, const int32_t *d_e, double *d_out, uint64_t n, hipStream_t st);
hipError_t launch_untile(const uint8_t *src, uint8_t *dst, int W, int H, int tile_w, int tile_h, int tiles_x, int n_ranks,
                         uint64_t rank_stride, int bpp, hipStream_t st);

}  // namespace rtx
#endif
