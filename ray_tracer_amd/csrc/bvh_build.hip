// Binned-SAH BVH build on the GPU that produces the arrays of the reference's builder
// (BVH::BVH / updateNodeBounds / subdivide / findBestSplitPlane,
// include/bvh.hpp:84-162,236-305) bit for bit: same node numbering, same bounds, same
// permutation of `indices`.
//
// Why that is possible in parallel:
//  * every reduction in the reference builder is a min, a max or an integer count, so
//    the order of accumulation does not change a value (only the sign of a zero, which is
//    tracked separately for the stored node bounds, see `zero_sign`);
//  * the in-place partition of bvh.hpp:128-140 is a fixed function of the left/right flags:
//    with nL = number of "left" elements, A = [0,nL), B = [nL,m), h_k = the k-th "right"
//    element of A from the front and l_k = the k-th "left" element of B from the back
//    (l_0 = m), the loop moves
//        left in A            -> stays
//        k-th right in A      -> l_{k-1} - 1
//        k-th left in B       -> h_k
//        right at position nL -> l_{k*} - 1      (k* = number of rights in A)
//        other rights in B    -> one position down
//    (derived from: elements are examined from the front after a left and from the back
//    after a right; lefts stay where they are examined, the t-th right examined lands at
//    m-1-t).  Prefix counts of the flags give every rank, two scatter passes apply it;
//  * the reference numbers nodes in depth-first order, children in pairs; an inner node's
//    pair index follows from the number of inner nodes before it in pre-order, which is
//    (inner nodes with a smaller `first`) + (ancestors with the same `first`).
//
// Nodes with more than kSmall triangles are built breadth-first, one level per iteration
// (bounds, bins, plane choice, flags + prefix counts, children, two scatter passes), in
// temporary node records; every node with at most kSmall triangles is handed to one wave
// that builds its whole subtree in registers (subtree_kernel).  The records are renumbered
// into the reference's order at the end.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bvh_build.h"
#include "device_layout.h"

namespace rtx {
namespace {

constexpr uint32_t kInactive = 0xffffffffu;
constexpr int kBins = 8;
constexpr int kBlock = 256;
constexpr int kFat = 1024;      // block size of the kernels that reduce into per-node records
constexpr int kScanPerThread = 4;
constexpr int kScanChunk = kBlock * kScanPerThread;   // positions per scan block
// A node with at most this many triangles leaves the level-by-level passes; one wave
// builds its whole subtree (subtree_kernel).
constexpr uint32_t kSmall = 64;
constexpr uint32_t kSerialMembers = 8;   // subtree nodes up to this size are reduced lane by lane

// Order-preserving map double -> u64 (with -0 < +0), so that min/max run as integer atomics.
__device__ __forceinline__ uint64_t key_of(double x) {
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k ^ 0x8000000000000000ull) : ~k;
    return __longlong_as_double((long long)b);
}
constexpr uint64_t kKeyMinInit = ~0ull;   // above every key
constexpr uint64_t kKeyMaxInit = 0ull;    // below every key

// Temporary node records, structure of arrays, indexed by the breadth-first id.
struct Nodes {
    uint32_t *first, *count, *parent;
    uint32_t *link;     // bit 0: is the right child; bits 1..: ancestors with the same `first`
    uint32_t *child;    // id of the left child (right = +1); 0 = leaf
    int32_t *axis;      // split axis chosen for this level, -1 = no split
    double *pos;        // split position
    uint32_t *nleft;    // "left" elements of the partition
    uint32_t *kstar;    // "right" elements among the first nleft
    uint64_t *box;      // [6] keys of bmin xyz, bmax xyz over the vertices
    uint64_t *cbox;     // [6] keys of the centroid bounds
    uint64_t *zero;     // [3] per axis: ((sequence position of the last zero coordinate + 1) << 1) | sign
};

__device__ __forceinline__ void init_node(const Nodes &nd, uint32_t t, uint32_t first, uint32_t count, uint32_t parent,
                                          uint32_t link) {
    nd.first[t] = first;
    nd.count[t] = count;
    nd.parent[t] = parent;
    nd.link[t] = link;
    nd.child[t] = 0;
    nd.axis[t] = -1;
    for (int k = 0; k < 3; k++) {
        nd.box[6 * (size_t)t + k] = kKeyMinInit;
        nd.box[6 * (size_t)t + 3 + k] = kKeyMaxInit;
        nd.cbox[6 * (size_t)t + k] = kKeyMinInit;
        nd.cbox[6 * (size_t)t + 3 + k] = kKeyMaxInit;
        nd.zero[3 * (size_t)t + k] = 0;
    }
}

// What every pass needs of a triangle, computed once: the keys of its box, its centroid and, per
// axis, which of its vertices is the last one with a zero coordinate (3 bits per axis:
// (vertex + 1) << 1 | 1 for +0, 0 = no zero) for the sign bookkeeping of the stored bounds.
struct alignas(16) TriRec {
    uint64_t lo[3], hi[3];
    double c[3];
    uint32_t zeros, pad;
};
static_assert(sizeof(TriRec) == 80, "TriRec");

__device__ __forceinline__ TriRec load_tri(const TriRec *__restrict__ T, uint32_t t) {
    // five 16-byte loads
    const uint4 *q = reinterpret_cast<const uint4 *>(T + t);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    TriRec r;
    auto u64 = [](uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; };
    r.lo[0] = u64(a.x, a.y); r.lo[1] = u64(a.z, a.w); r.lo[2] = u64(b.x, b.y);
    r.hi[0] = u64(b.z, b.w); r.hi[1] = u64(c.x, c.y); r.hi[2] = u64(c.z, c.w);
    r.c[0] = __longlong_as_double((long long)u64(d.x, d.y)); r.c[1] = __longlong_as_double((long long)u64(d.z, d.w));
    r.c[2] = __longlong_as_double((long long)u64(e.x, e.y));
    r.zeros = e.z; r.pad = 0;
    return r;
}
// Tracker value of a triangle at sequence position p for one axis (see Nodes::zero).
__device__ __forceinline__ uint64_t zero_key(uint32_t zeros, int axis, uint64_t p) {
    const uint32_t zi = (zeros >> (3 * axis)) & 7u;
    return zi ? ((p * 3 + (zi >> 1)) << 1) | (zi & 1u) : 0ull;
}

// Triangle::centroid as the Triangle ctor computes it (include/triangle.hpp:24), and the record above.
__global__ void centroid_kernel(const double *__restrict__ V, TriRec *__restrict__ T, uint32_t *__restrict__ idx,
                                uint32_t *__restrict__ node_of, uint32_t n, Nodes nd) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p == 0) init_node(nd, 0, 0, n, 0, 0);
    if (p >= n) return;
    const double *v = V + 9 * (size_t)p;
    TriRec r;
    r.zeros = 0; r.pad = 0;
    for (int k = 0; k < 3; k++) {
        const double a = v[k], b = v[3 + k], c = v[6 + k];
        const uint64_t ka = key_of(a), kb = key_of(b), kc = key_of(c);
        uint64_t m = ka < kb ? ka : kb;
        r.lo[k] = m < kc ? m : kc;
        m = ka > kb ? ka : kb;
        r.hi[k] = m > kc ? m : kc;
        r.c[k] = ((a + b) + c) / 3;
        // the last coordinate of this triangle that is a zero, in the order v0, v1, v2
        uint32_t zi = 0;
        if (a == 0.0) zi = (1u << 1) | (uint32_t)(ka >> 63);
        if (b == 0.0) zi = (2u << 1) | (uint32_t)(kb >> 63);
        if (c == 0.0) zi = (3u << 1) | (uint32_t)(kc >> 63);
        r.zeros |= zi << (3 * k);
    }
    T[p] = r;
    idx[p] = p;
    node_of[p] = 0;
}

__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int d) {
    const uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ double shfl_f64(double v, int src) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = __shfl((uint32_t)b, src), hi = __shfl((uint32_t)(b >> 32), src);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// Reduction over runs of equal `node` among consecutive lanes; the last lane of a run
// ends up with the run's result.
template <bool IS_MIN>
__device__ __forceinline__ uint64_t run_reduce(uint64_t v, uint32_t node, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = shfl_up64(v, d);
        const uint32_t on = __shfl_up(node, d);
        if (lane >= d && on == node) v = IS_MIN ? (o < v ? o : v) : (o > v ? o : v);
    }
    return v;
}

// Wave reductions without trips through memory: four DPP exchanges inside rows of 16 lanes
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), one ds_swizzle
// across the rows of a half wave (lane ^ 16) and two readlanes across the halves.  The
// result is uniform.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v) {
    return ((uint64_t)dpp32<CTRL>((uint32_t)(v >> 32)) << 32) | dpp32<CTRL>((uint32_t)v);
}
__device__ __forceinline__ uint32_t swz16_32(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }
__device__ __forceinline__ uint64_t swz16_64(uint64_t v) {
    return ((uint64_t)swz16_32((uint32_t)(v >> 32)) << 32) | swz16_32((uint32_t)v);
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
template <bool IS_MIN>
__device__ __forceinline__ uint64_t pick64(uint64_t a, uint64_t b) { return IS_MIN ? (b < a ? b : a) : (b > a ? b : a); }
template <bool IS_MIN>
__device__ __forceinline__ uint64_t wave_reduce64(uint64_t v) {
    v = pick64<IS_MIN>(v, dpp64<0xB1>(v));
    v = pick64<IS_MIN>(v, dpp64<0x4E>(v));
    v = pick64<IS_MIN>(v, dpp64<0x141>(v));
    v = pick64<IS_MIN>(v, dpp64<0x140>(v));
    v = pick64<IS_MIN>(v, swz16_64(v));
    return pick64<IS_MIN>(readlane64(v, 0), readlane64(v, 32));
}
__device__ __forceinline__ uint32_t wave_max32(uint32_t v) {
    uint32_t o;
    o = dpp32<0xB1>(v); v = o > v ? o : v;
    o = dpp32<0x4E>(v); v = o > v ? o : v;
    o = dpp32<0x141>(v); v = o > v ? o : v;
    o = dpp32<0x140>(v); v = o > v ? o : v;
    o = swz16_32(v); v = o > v ? o : v;
    const uint32_t x = __builtin_amdgcn_readlane(v, 0), y = __builtin_amdgcn_readlane(v, 32);
    return x > y ? x : y;
}
// LDS traffic of one wave is executed in order; this keeps the compiler from moving
// accesses across the point where other lanes' writes are consumed.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// updateNodeBounds (bvh.hpp:100-115) for every node of the new level at once, plus the
// centroid bounds findBestSplitPlane starts with (bvh.hpp:242-250).
// A block walks `tiles_per_block` consecutive tiles of kFat positions.  Tiles that lie
// inside one node are accumulated in LDS across tiles, so that a large node receives one
// set of atomics per block; other tiles send one set per run of lanes that share a node.
__global__ __launch_bounds__(kFat, 8) void bounds_kernel(const TriRec *__restrict__ T, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ node_of,
                                                      uint32_t n, Nodes nd, uint32_t tiles_per_block) {
    // 0-2 bmin, 3-5 bmax, 6-8 centroid min, 9-11 centroid max, 12-14 zero trackers
    __shared__ unsigned long long s_acc[15];
    const int lane = threadIdx.x & 63;
    const bool acc_is_min = threadIdx.x < 3 || (threadIdx.x >= 6 && threadIdx.x < 9);
    if (threadIdx.x < 15) s_acc[threadIdx.x] = acc_is_min ? kKeyMinInit : kKeyMaxInit;
    uint32_t acc_node = kInactive;                   // block-uniform
    auto flush = [&]() {
        __syncthreads();
        if (acc_node != kInactive && threadIdx.x < 15) {
            const unsigned long long v = s_acc[threadIdx.x];
            unsigned long long *dst = threadIdx.x < 6    ? (unsigned long long *)nd.box + 6 * (size_t)acc_node + threadIdx.x
                                      : threadIdx.x < 12 ? (unsigned long long *)nd.cbox + 6 * (size_t)acc_node + (threadIdx.x - 6)
                                                         : (unsigned long long *)nd.zero + 3 * (size_t)acc_node + (threadIdx.x - 12);
            if (acc_is_min) atomicMin(dst, v); else if (v) atomicMax(dst, v);
            s_acc[threadIdx.x] = acc_is_min ? kKeyMinInit : kKeyMaxInit;
        }
        __syncthreads();
    };
    __syncthreads();
    for (uint32_t tile = 0; tile < tiles_per_block; tile++) {
        const uint64_t start = ((uint64_t)blockIdx.x * tiles_per_block + tile) * kFat;
        if (start >= n) break;
        const uint32_t p = (uint32_t)start + threadIdx.x;
        const uint32_t node = p < n ? node_of[p] : kInactive;
        if (!__syncthreads_or(node != kInactive)) continue;
        const uint32_t tile_node = node_of[start];
        const bool uniform = __syncthreads_and(p >= n || node == tile_node) && tile_node != kInactive;
        if (!uniform || acc_node != tile_node) {
            flush();
            acc_node = uniform ? tile_node : kInactive;
        }
        const bool act = node != kInactive;
        TriRec r;
        if (act) r = load_tri(T, idx[p]);
        // One axis at a time, reduced and sent before the next one is touched: fewer values live at once, so the kernel
        // fits 64 registers and two blocks share a CU (the pass waits for its gathers, not for arithmetic).
        const uint32_t next = __shfl_down(node, 1);
        const bool tail = uniform ? (lane == 0 && __ballot(node != kInactive) != 0)
                                  : ((lane == 63 || next != node) && node != kInactive);
        unsigned long long *box = uniform ? s_acc : (unsigned long long *)nd.box + 6 * (size_t)(tail ? node : 0);
        unsigned long long *cbox = uniform ? s_acc + 6 : (unsigned long long *)nd.cbox + 6 * (size_t)(tail ? node : 0);
        unsigned long long *zero = uniform ? s_acc + 12 : (unsigned long long *)nd.zero + 3 * (size_t)(tail ? node : 0);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // lanes without a triangle hold the neutral values
            uint64_t l = act ? r.lo[k] : kKeyMinInit, h = act ? r.hi[k] : kKeyMaxInit;
            uint64_t cl = act ? key_of(r.c[k]) : kKeyMinInit, ch = act ? key_of(r.c[k]) : kKeyMaxInit;
            uint64_t z = act ? zero_key(r.zeros, k, p) : 0ull;
            if (uniform) {
                // every lane of the tile (bar the ones past the end, which hold the neutral values) belongs to one node:
                // plain wave reductions, no segment bookkeeping and no trips through the LDS crossbar
                l = wave_reduce64<true>(l);
                h = wave_reduce64<false>(h);
                cl = wave_reduce64<true>(cl);
                ch = wave_reduce64<false>(ch);
                z = wave_reduce64<false>(z);
            } else {
                l = run_reduce<true>(l, node, lane);
                h = run_reduce<false>(h, node, lane);
                cl = run_reduce<true>(cl, node, lane);
                ch = run_reduce<false>(ch, node, lane);
                z = run_reduce<false>(z, node, lane);
            }
            if (tail) {
                atomicMin(box + k, (unsigned long long)l);
                atomicMax(box + 3 + k, (unsigned long long)h);
                atomicMin(cbox + k, (unsigned long long)cl);
                atomicMax(cbox + 3 + k, (unsigned long long)ch);
                if (z) atomicMax(zero + k, (unsigned long long)z);
            }
        }
    }
    flush();
}

// Per-level bin storage for the nodes [level_begin, level_end).
struct Bins {
    uint64_t *lo;     // [width][3 axes][kBins][3]
    uint64_t *hi;
    uint32_t *cnt;    // [width][3 axes][kBins]
};
constexpr int kBinSlots = 3 * kBins;

// Empty bins for the `width` nodes of a level (one launch in place of three fills).
__global__ void bins_clear_kernel(Bins bins, uint32_t width) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x, slots = (uint64_t)width * kBinSlots;
    if (i < slots * 3) { bins.lo[i] = kKeyMinInit; bins.hi[i] = kKeyMaxInit; }
    if (i < slots) bins.cnt[i] = 0;
}

// The binning loop of findBestSplitPlane (bvh.hpp:259-266) for all three axes of every
// node of the level.  Same tiling as bounds_kernel: tiles inside one node fill one set of
// bins in LDS per block; in other tiles each wave accumulates every run of lanes that
// share a node in its own LDS bins and adds the non-empty ones to the node's bins.
// The CU's rate of 64-bit LDS atomics is what this pass waits for.
__global__ __launch_bounds__(kFat, 8) void bin_kernel(const TriRec *__restrict__ T, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ node_of,
                                                   uint32_t n, Nodes nd, Bins bins, uint32_t level_begin,
                                                   uint32_t tiles_per_block) {
    constexpr int kWaves = kFat / 64;
    // Same-address LDS atomics of one instruction run one after the other, so every set of bins exists in several
    // copies and lane l adds to copy l mod (number of copies); the copies are merged when the set is sent.
    // Sets [0, kWaves * kWaveCopies): kWaveCopies per wave, for tiles that span several nodes.
    // Sets [kMine, kMine + kCopies): the block's, for tiles inside one node.
    constexpr int kCopies = 8, kWaveCopies = 2, kMine = kWaves * kWaveCopies;
    __shared__ unsigned long long s_lo[kMine + kCopies][kBinSlots][3], s_hi[kMine + kCopies][kBinSlots][3];
    __shared__ uint32_t s_cnt[kMine + kCopies][kBinSlots];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto reset = [&](int which, int slot) {
        s_cnt[which][slot] = 0;
        for (int k = 0; k < 3; k++) { s_lo[which][slot][k] = kKeyMinInit; s_hi[which][slot][k] = kKeyMaxInit; }
    };
    // merges the `copies` sets that start at `first` into the first of them, adds it to the node's bins and clears them
    auto send = [&](int first, int copies, int slot, uint32_t node) {
        for (int c = 1; c < copies; c++) {
            s_cnt[first][slot] += s_cnt[first + c][slot];
            for (int k = 0; k < 3; k++) {
                const unsigned long long l = s_lo[first + c][slot][k], h = s_hi[first + c][slot][k];
                if (l < s_lo[first][slot][k]) s_lo[first][slot][k] = l;
                if (h > s_hi[first][slot][k]) s_hi[first][slot][k] = h;
            }
        }
        if (s_cnt[first][slot]) {
            const size_t w = (size_t)(node - level_begin) * kBinSlots + slot;
            atomicAdd(bins.cnt + w, s_cnt[first][slot]);
            for (int k = 0; k < 3; k++) {
                atomicMin((unsigned long long *)bins.lo + 3 * w + k, s_lo[first][slot][k]);
                atomicMax((unsigned long long *)bins.hi + 3 * w + k, s_hi[first][slot][k]);
            }
        }
        for (int c = 0; c < copies; c++) reset(first + c, slot);
    };
    uint32_t acc_node = kInactive;                   // block-uniform
    auto flush = [&]() {
        __syncthreads();
        if (acc_node != kInactive && threadIdx.x < kBinSlots) send(kMine, kCopies, threadIdx.x, acc_node);
        __syncthreads();
    };
    if (threadIdx.x < kBinSlots)
        for (int c = 0; c < kMine + kCopies; c++) reset(c, threadIdx.x);
    __syncthreads();
    for (uint32_t tile = 0; tile < tiles_per_block; tile++) {
        const uint64_t start = ((uint64_t)blockIdx.x * tiles_per_block + tile) * kFat;
        if (start >= n) break;
        const uint32_t p = (uint32_t)start + threadIdx.x;
        const uint32_t node = p < n ? node_of[p] : kInactive;
        if (!__syncthreads_or(node != kInactive)) continue;
        const uint32_t tile_node = node_of[start];
        const bool uniform = __syncthreads_and(p >= n || node == tile_node) && tile_node != kInactive;
        if (!uniform || acc_node != tile_node) {
            flush();
            acc_node = uniform ? tile_node : kInactive;
        }
        int bin[3] = {-1, -1, -1};
        uint64_t lo[3], hi[3];
        if (node != kInactive) {
            const TriRec r = load_tri(T, idx[p]);
#pragma unroll
            for (int k = 0; k < 3; k++) { lo[k] = r.lo[k]; hi[k] = r.hi[k]; }
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const double cmin = value_of(nd.cbox[6 * (size_t)node + a]), cmax = value_of(nd.cbox[6 * (size_t)node + 3 + a]);
                if (cmin == cmax) continue;                                   // bvh.hpp:252
                const double scale = kBins / (cmax - cmin);
                size_t b = static_cast<size_t>((r.c[a] - cmin) * scale);
                if (b > kBins - 1) b = kBins - 1;
                bin[a] = (int)b;
            }
        }
        auto add_to = [&](int which) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                if (bin[a] < 0) continue;
                const int s = a * kBins + bin[a];
                atomicAdd(&s_cnt[which][s], 1u);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    atomicMin(&s_lo[which][s][k], (unsigned long long)lo[k]);
                    atomicMax(&s_hi[which][s][k], (unsigned long long)hi[k]);
                }
            }
        };
        if (uniform) {
            if (node != kInactive) add_to(kMine + (lane & (kCopies - 1)));
            continue;
        }
        // the runs of this wave, one after the other; the wave's sets are its own, so it only waits for itself
        uint64_t todo = __ballot(node != kInactive);
        while (todo) {
            const uint32_t cur = __shfl(node, __ffsll((unsigned long long)todo) - 1);
            const bool mine = node == cur;
            if (mine) add_to(wave * kWaveCopies + (lane & (kWaveCopies - 1)));
            wave_sync();
            if (lane < kBinSlots) send(wave * kWaveCopies, kWaveCopies, lane, cur);
            wave_sync();
            todo &= ~__ballot(mine);
        }
    }
    flush();
}

// AABB::area (bvh.hpp:74-78)
__device__ __forceinline__ double half_area(const double lo[3], const double hi[3]) {
    const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return ex * ey + ey * ez + ez * ex;
}

// findBestSplitPlane's plane sweep (bvh.hpp:268-302) and subdivide's cost test
// (bvh.hpp:117-126): four lanes per node of the level, three of them sweep one axis each, the first
// one then takes the axes' results in the reference's order (a later axis wins only with a smaller cost).
__global__ void plane_kernel(Nodes nd, Bins bins, uint32_t level_begin, uint32_t level_end) {
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t t = level_begin + (g >> 2);
    const int a = (int)(g & 3u);
    if (t >= level_end || nd.count[t] <= kSmall) return;          // the four lanes of a node leave together
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double best = inf, best_pos = 0;
    int best_axis = -1;
    if (a < 3) {
        const double cmin = value_of(nd.cbox[6 * (size_t)t + a]), cmax = value_of(nd.cbox[6 * (size_t)t + 3 + a]);
        if (cmin != cmax) {
        const size_t w = ((size_t)(t - level_begin) * 3 + a) * kBins;
        double blo[kBins][3], bhi[kBins][3];
        int bcount[kBins];
        for (int b = 0; b < kBins; b++) {
            bcount[b] = (int)bins.cnt[w + b];
            for (int k = 0; k < 3; k++) {
                blo[b][k] = value_of(bins.lo[3 * (w + b) + k]);
                bhi[b][k] = value_of(bins.hi[3 * (w + b) + k]);
            }
        }
        double larea[kBins - 1], rarea[kBins - 1];
        int lcount[kBins - 1], rcount[kBins - 1];
        double llo[3] = {inf, inf, inf}, lhi[3] = {-inf, -inf, -inf};
        double rlo[3] = {inf, inf, inf}, rhi[3] = {-inf, -inf, -inf};
        int lsum = 0, rsum = 0;
        for (int i = 0; i < kBins - 1; i++) {
            lsum += bcount[i];
            lcount[i] = lsum;
            if (bcount[i]) for (int k = 0; k < 3; k++) {                  // AABB::grow(AABB), bvh.hpp:67-72
                llo[k] = llo[k] < blo[i][k] ? llo[k] : blo[i][k];
                lhi[k] = lhi[k] > bhi[i][k] ? lhi[k] : bhi[i][k];
            }
            larea[i] = half_area(llo, lhi);
            const int rb = kBins - 1 - i;
            rsum += bcount[rb];
            rcount[kBins - 2 - i] = rsum;
            if (bcount[rb]) for (int k = 0; k < 3; k++) {
                rlo[k] = rlo[k] < blo[rb][k] ? rlo[k] : blo[rb][k];
                rhi[k] = rhi[k] > bhi[rb][k] ? rhi[k] : bhi[rb][k];
            }
            rarea[kBins - 2 - i] = half_area(rlo, rhi);
        }
        const double width = (cmax - cmin) / kBins;
        for (int i = 0; i < kBins - 1; i++) {
            const double cost = lcount[i] * larea[i] + rcount[i] * rarea[i];
            if (cost < best) {
                best = cost;
                best_axis = a;
                best_pos = cmin + width * static_cast<double>(i + 1);
            }
        }
        }
    }
    // axes 1 and 2 hand their best to the node's first lane
    for (int from = 1; from < 3; from++) {
        const int src = (int)(threadIdx.x & 63u) - a + from;
        const double c = shfl_f64(best, src), ps = shfl_f64(best_pos, src);
        if (a == 0 && c < best) { best = c; best_pos = ps; best_axis = from; }
    }
    if (a != 0) return;
    double lo[3], hi[3];
    for (int k = 0; k < 3; k++) { lo[k] = value_of(nd.box[6 * (size_t)t + k]); hi[k] = value_of(nd.box[6 * (size_t)t + 3 + k]); }
    const double parent_cost = static_cast<double>(nd.count[t]) * half_area(lo, hi);   // BVHNode::cost, bvh.hpp:80-82
    if (best >= parent_cost) best_axis = -1;
    nd.axis[t] = best_axis;
    nd.pos[t] = best_pos;
}

// "left" flags of the partition (bvh.hpp:131) with their exclusive prefix count inside
// each chunk of kScanChunk positions; chunk totals go to `sums`.
__global__ void flag_kernel(const TriRec *__restrict__ T, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ node_of,
                            uint32_t n, Nodes nd, uint8_t *__restrict__ flag, uint32_t *__restrict__ local,
                            uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    uint32_t f[kScanPerThread], mine = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) {
        const uint32_t p = base + k;
        f[k] = 0;
        if (p < n) {
            const uint32_t node = node_of[p];
            if (node != kInactive) {
                const int axis = nd.axis[node];
                if (axis >= 0) f[k] = T[idx[p]].c[axis] < nd.pos[node] ? 1u : 0u;
            }
        }
        mine += f[k];
    }
    // exclusive scan of `mine` over the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < kBlock / 64; w++) {
        if (w < wave) before += s_wave[w];
        total += s_wave[w];
    }
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) {
        const uint32_t p = base + k;
        if (p < n) { flag[p] = (uint8_t)f[k]; local[p] = run; }
        run += f[k];
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// Same local scan for an array of counts (in place).
__global__ void local_scan_kernel(uint32_t *__restrict__ data, uint32_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    uint32_t f[kScanPerThread], mine = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) { f[k] = base + k < n ? data[base + k] : 0; mine += f[k]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < kBlock / 64; w++) {
        if (w < wave) before += s_wave[w];
        total += s_wave[w];
    }
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < kScanPerThread; k++) {
        if (base + k < n) data[base + k] = run;
        run += f[k];
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// Exclusive scan of the chunk totals by one block; sums[count] receives the grand total.
__global__ void scan_sums_kernel(uint32_t *__restrict__ sums, uint32_t count) {
    __shared__ uint32_t s_wave[kBlock / 64];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t start = 0; start < count; start += kBlock) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t v = i < count ? sums[i] : 0;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry, total = 0;
        for (int w = 0; w < kBlock / 64; w++) {
            if (w < wave) before += s_wave[w];
            total += s_wave[w];
        }
        if (i < count) sums[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[count] = s_carry;
}

struct Counters {
    uint32_t next_id;   // temporary node ids handed out so far
    uint32_t small;     // entries of the small-subtree list
    uint32_t big;       // children created so far that stay in the level passes
    uint32_t deepest;   // deepest level reached inside the small subtrees
};

struct Prefix {
    const uint32_t *local, *sums;
    __device__ __forceinline__ uint32_t operator()(uint32_t p) const { return sums[p / kScanChunk] + local[p]; }
};

// Children of every node that splits (bvh.hpp:141-158); a partition that leaves one side
// empty makes the node a leaf, but its permutation is still applied (bvh.hpp:139-141).
__global__ void split_kernel(Nodes nd, Prefix pre, const uint8_t *__restrict__ flag, uint32_t level_begin, uint32_t level_end,
                             Counters *__restrict__ counters, uint2 *__restrict__ small_list, uint32_t level, uint32_t level_cap) {
    const uint32_t t = level_begin + blockIdx.x * kBlock + threadIdx.x;
    if (t >= level_end || nd.count[t] <= kSmall || nd.axis[t] < 0) return;
    const uint32_t f0 = nd.first[t], m = nd.count[t];
    const uint32_t e0 = pre(f0);
    const uint32_t nl = pre(f0 + m - 1) + flag[f0 + m - 1] - e0;
    uint32_t ks = 0;
    if (nl != 0 && nl != m) ks = nl - (pre(f0 + nl) - e0);
    nd.nleft[t] = nl;
    nd.kstar[t] = ks;
    if (nl == 0 || nl == m) return;
    const uint32_t c = atomicAdd(&counters->next_id, 2u);
    // the per-level-node arrays hold `level_cap` records: past them nothing is written, the node stays without
    // children and the host, which sees next_id beyond the bound after this level, starts over with larger arrays
    if (c + 1 >= level_cap) return;
    nd.child[t] = c;
    init_node(nd, c, f0, nl, t, ((nd.link[t] >> 1) + 1) << 1);
    init_node(nd, c + 1, f0 + nl, m - nl, t, 1u);
    const uint32_t sizes[2] = {nl, m - nl};
    uint32_t big = 0;
    for (int k = 0; k < 2; k++) {
        if (sizes[k] <= kSmall) small_list[atomicAdd(&counters->small, 1u)] = make_uint2(c + k, level + 1);
        else big++;
    }
    if (big) atomicAdd(&counters->big, big);
}

// Positions (relative to the node) of the k-th right element of A and of the k-th left
// element of B from the back.
__global__ void holes_kernel(const uint32_t *__restrict__ node_of, uint32_t n, Nodes nd, Prefix pre,
                             const uint8_t *__restrict__ flag, uint32_t *__restrict__ holes, uint32_t *__restrict__ fills) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t t = node_of[p];
    if (t == kInactive || nd.axis[t] < 0) return;
    const uint32_t f0 = nd.first[t], nl = nd.nleft[t], rel = p - f0, f = flag[p];
    const uint32_t lefts_to_here = pre(p) + f - pre(f0);
    if (rel < nl && !f) holes[f0 + (rel + 1 - lefts_to_here) - 1] = rel;
    if (rel >= nl && f) fills[f0 + (nl - lefts_to_here + 1) - 1] = rel;
}

// Applies the partition's permutation and hands every position to its child node.
__global__ void permute_kernel(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ node_of, uint32_t n, Nodes nd,
                               Prefix pre, const uint8_t *__restrict__ flag, const uint32_t *__restrict__ holes,
                               const uint32_t *__restrict__ fills, uint32_t *__restrict__ idx_out,
                               uint32_t *__restrict__ node_out) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t t = node_of[p];
    if (t == kInactive || nd.axis[t] < 0) {
        idx_out[p] = idx[p];
        node_out[p] = kInactive;
        return;
    }
    const uint32_t f0 = nd.first[t], m = nd.count[t], nl = nd.nleft[t], ks = nd.kstar[t], rel = p - f0, f = flag[p];
    const uint32_t lefts_to_here = pre(p) + f - pre(f0);
    uint32_t dest;
    if (rel < nl) {
        if (f) dest = rel;
        else {
            const uint32_t k = rel + 1 - lefts_to_here;
            dest = (k == 1 ? m : fills[f0 + k - 2]) - 1;
        }
    } else {
        if (f) dest = holes[f0 + (nl - lefts_to_here + 1) - 1];
        else if (rel == nl) dest = (ks == 0 ? m : fills[f0 + ks - 1]) - 1;
        else dest = rel - 1;
    }
    idx_out[f0 + dest] = idx[p];
    const uint32_t c = nd.child[t];
    uint32_t to = kInactive;
    if (c) {
        to = dest < nl ? c : c + 1;
        if ((dest < nl ? nl : m - nl) <= kSmall) to = kInactive;
    }
    node_out[f0 + dest] = to;
}


// ---- one wave builds a whole small subtree -------------------------------------------

// Smallest cost over the wave; on equal costs the smallest tag.
__device__ __forceinline__ void wave_argmin(double &cost, int &tag) {
    auto step = [&](uint64_t oc_bits, uint32_t ot) {
        const double oc = __longlong_as_double((long long)oc_bits);
        if (oc < cost || (oc == cost && (int)ot < tag)) { cost = oc; tag = (int)ot; }
    };
    step(dpp64<0xB1>((uint64_t)__double_as_longlong(cost)), dpp32<0xB1>((uint32_t)tag));
    step(dpp64<0x4E>((uint64_t)__double_as_longlong(cost)), dpp32<0x4E>((uint32_t)tag));
    step(dpp64<0x141>((uint64_t)__double_as_longlong(cost)), dpp32<0x141>((uint32_t)tag));
    step(dpp64<0x140>((uint64_t)__double_as_longlong(cost)), dpp32<0x140>((uint32_t)tag));
    step(swz16_64((uint64_t)__double_as_longlong(cost)), swz16_32((uint32_t)tag));
    const uint64_t c0 = readlane64((uint64_t)__double_as_longlong(cost), 0), c1 = readlane64((uint64_t)__double_as_longlong(cost), 32);
    const uint32_t t0 = __builtin_amdgcn_readlane((uint32_t)tag, 0), t1 = __builtin_amdgcn_readlane((uint32_t)tag, 32);
    cost = __longlong_as_double((long long)c0);
    tag = (int)t0;
    step(c1, t1);
}
__device__ __forceinline__ uint64_t low_mask(uint32_t k) { return k >= 64 ? ~0ull : ((1ull << k) - 1); }
// Position of the k-th set bit (k >= 1, counted from bit 0) of a mask that has one.
__device__ __forceinline__ uint32_t select_bit(uint64_t mask, uint32_t k) {
    uint32_t pos = 0;
#pragma unroll
    for (int step = 32; step; step >>= 1)
        if ((uint32_t)__popcll(mask & low_mask(pos + step)) < k) pos += step;
    return pos;
}
// subdivide (bvh.hpp:117-162) for every node below a small root, by one wave: lane = one
// triangle of the root, which keeps its data in registers and only changes its position
// `pos` inside the root's range when a partition moves it.  Pending nodes are kept in a
// stack spread over the lanes' registers.  Node ids come from a block of 2m-2 ids reserved
// for the root; ids left over are marked unused.
__global__ __launch_bounds__(kBlock, 5) void subtree_kernel(const TriRec *__restrict__ T,
                                                         uint32_t *__restrict__ idx, const uint2 *__restrict__ small_list,
                                                         const uint32_t *__restrict__ need_local,
                                                         const uint32_t *__restrict__ need_sums, uint32_t n_small,
                                                         uint32_t id_base, Nodes nd, Counters *__restrict__ counters) {
    __shared__ unsigned long long s_lo[kBlock / 64][kBinSlots][3], s_hi[kBlock / 64][kBinSlots][3];
    __shared__ uint32_t s_cnt[kBlock / 64][kBinSlots];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w = blockIdx.x * (kBlock / 64) + wave;
    if (w >= n_small) return;                       // whole waves leave; no block barrier below
    const uint2 entry = small_list[w];
    const uint32_t root = entry.x;
    const uint32_t f0 = nd.first[root], m = nd.count[root];
    uint32_t next_free = id_base + need_sums[w / kScanChunk] + need_local[w];
    const uint32_t id_end = next_free + 2 * m - 2;
    const bool active = (uint32_t)lane < m;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);

    uint32_t tri = 0, pos = lane;
    double c[3] = {0, 0, 0};
    uint64_t lo[3], hi[3];
    uint32_t zinfo[3] = {0, 0, 0};     // ((index of the last zero vertex + 1) << 1) | sign, 0 = none
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = kKeyMinInit; hi[k] = kKeyMaxInit; }
    if (active) {
        tri = idx[f0 + lane];
        const TriRec r = load_tri(T, tri);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = r.lo[k];
            hi[k] = r.hi[k];
            zinfo[k] = (r.zeros >> (3 * k)) & 7u;
            c[k] = r.c[k];
        }
    }

    // half the surface of the triangle's own box (AABB::area, bvh.hpp:74-78), what it adds to a split's cost when it is alone on its side
    double own_area = 0;
    if (active) {
        double l3[3], h3[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { l3[k] = value_of(lo[k]); h3[k] = value_of(hi[k]); }
        own_area = half_area(l3, h3);
    }

    // pending nodes: entry e lives in lane e
    uint32_t st_range = 0, st_id = 0, st_run = 0;
    int sp = 0;
    if (lane == 0) { st_range = 0u | (m << 8) | (entry.y << 16); st_id = root; st_run = nd.link[root] >> 1; }
    sp = 1;
    uint32_t deepest = entry.y;

    while (sp > 0) {
        sp = __builtin_amdgcn_readfirstlane(sp - 1);
        const uint32_t range = __builtin_amdgcn_readlane(st_range, sp);
        const uint32_t id = __builtin_amdgcn_readlane(st_id, sp);
        const uint32_t run = __builtin_amdgcn_readlane(st_run, sp);
        const uint32_t a = range & 0xff, b = (range >> 8) & 0xff, lev = range >> 16;
        const uint32_t cnt = b - a;
        deepest = lev > deepest ? lev : deepest;
        const bool member = active && pos >= a && pos < b;

        // updateNodeBounds (bvh.hpp:100-115) and the centroid bounds of bvh.hpp:242-250.
        // Few members: read them lane by lane into uniform values; many: butterfly reductions.
        uint64_t box[6], zero[3], cbox_k[6];
        if (cnt == 1) {
            // a single triangle: its own box; it cannot be split (bvh.hpp:252 skips every axis)
            const int src = __ffsll((unsigned long long)__ballot(member)) - 1;
            const uint32_t p_src = __builtin_amdgcn_readlane(pos, src);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                box[k] = readlane64(lo[k], src);
                box[3 + k] = readlane64(hi[k], src);
                cbox_k[k] = cbox_k[3 + k] = 0;
                const uint32_t zi = __builtin_amdgcn_readlane(zinfo[k], src);
                zero[k] = zi ? ((p_src * 3 + (zi >> 1)) << 1) | (zi & 1) : 0u;
            }
        } else if (cnt <= kSerialMembers) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                box[k] = cbox_k[k] = kKeyMinInit;
                box[3 + k] = cbox_k[3 + k] = kKeyMaxInit;
                zero[k] = 0;
            }
            uint64_t todo = __ballot(member);
            while (todo) {
                const int src = __ffsll((unsigned long long)todo) - 1;
                todo &= todo - 1;
                const uint32_t p_src = __builtin_amdgcn_readlane(pos, src);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint64_t l = readlane64(lo[k], src), h = readlane64(hi[k], src);
                    const uint64_t kc = key_of(__longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(c[k]), src)));
                    const uint32_t zi = __builtin_amdgcn_readlane(zinfo[k], src);
                    box[k] = l < box[k] ? l : box[k];
                    box[3 + k] = h > box[3 + k] ? h : box[3 + k];
                    cbox_k[k] = kc < cbox_k[k] ? kc : cbox_k[k];
                    cbox_k[3 + k] = kc > cbox_k[3 + k] ? kc : cbox_k[3 + k];
                    const uint64_t z = zi ? ((p_src * 3 + (zi >> 1)) << 1) | (zi & 1) : 0u;
                    zero[k] = z > zero[k] ? z : zero[k];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                box[k] = wave_reduce64<true>(member ? lo[k] : kKeyMinInit);
                box[3 + k] = wave_reduce64<false>(member ? hi[k] : kKeyMaxInit);
                const uint32_t z = (member && zinfo[k]) ? ((pos * 3 + (zinfo[k] >> 1)) << 1) | (zinfo[k] & 1) : 0u;
                zero[k] = wave_max32(z);
                const uint64_t kc = key_of(c[k]);
                cbox_k[k] = wave_reduce64<true>(member ? kc : kKeyMinInit);
                cbox_k[3 + k] = wave_reduce64<false>(member ? kc : kKeyMaxInit);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) nd.box[6 * (size_t)id + k] = box[k];
#pragma unroll
            for (int k = 0; k < 3; k++) nd.zero[3 * (size_t)id + k] = zero[k];
        }

        // findBestSplitPlane (bvh.hpp:236-305)
        double best = inf, best_pos = 0;
        int best_axis = -1;
        bool pair_done = false;
        if (cnt == 2) {
            // Two triangles, the commonest inner node by far.  On an axis where the centroids differ the one with the
            // smaller centroid falls in bin 0 ((c - cmin) * scale = 0) and the other in bin 7 (d * (8 / d) is within an
            // ulp of 8, clamped), so all seven planes of every such axis cut between them at the same cost,
            // area(one) + area(other) (1 * x = x, and a sum does not depend on the order of its terms): the scan of
            // bvh.hpp:268-302 keeps the first, plane 0 of the lowest axis.  Centroid differences so small or large that 8 / d
            // overflows or underflows (or that are not numbers) take the general path.
            double cmin[3], cmax[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { cmin[k] = value_of(cbox_k[k]); cmax[k] = value_of(cbox_k[3 + k]); }
            const uint64_t both = __ballot(member);
            const int la = __ffsll((unsigned long long)both) - 1, lb = 63 - __clzll((unsigned long long)both);
            const double area_sum = __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(own_area), la)) +
                                    __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(own_area), lb));
            int ax = -1;
            bool tame = true;
#pragma unroll
            for (int k = 2; k >= 0; k--)
                if (cmin[k] != cmax[k]) {
                    ax = k;
                    tame = tame && (cmax[k] - cmin[k]) >= 0x1p-1000 && (cmax[k] - cmin[k]) <= 0x1p1000;
                }
            if (tame) {
                pair_done = true;
                if (ax >= 0 && area_sum < inf) {
                    const double lo_c = ax == 0 ? cmin[0] : ax == 1 ? cmin[1] : cmin[2];
                    const double hi_c = ax == 0 ? cmax[0] : ax == 1 ? cmax[1] : cmax[2];
                    const double width = (hi_c - lo_c) / kBins;
                    best = area_sum;
                    best_axis = ax;
                    best_pos = lo_c + width * static_cast<double>(1);
                }
            }
        }
        if (cnt >= 2 && !pair_done) {
            double cmin[3], cmax[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { cmin[k] = value_of(cbox_k[k]); cmax[k] = value_of(cbox_k[3 + k]); }
            // lanes 0..20: one candidate plane each (axis-major, as the reference scans them)
            const int ax = lane / (kBins - 1), plane = lane % (kBins - 1);
            const double lo_c = ax == 0 ? cmin[0] : ax == 1 ? cmin[1] : cmin[2];
            const double hi_c = ax == 0 ? cmax[0] : ax == 1 ? cmax[1] : cmax[2];
            const bool candidate = lane < 3 * (kBins - 1) && lo_c != hi_c;
            double llo[3] = {inf, inf, inf}, lhi[3] = {-inf, -inf, -inf};
            double rlo[3] = {inf, inf, inf}, rhi[3] = {-inf, -inf, -inf};
            int lcount = 0, rcount = 0;
            if (cnt <= kSerialMembers) {
                // few members: every candidate lane sorts each member to its side directly
                const double scale = kBins / (hi_c - lo_c);
                uint64_t todo = __ballot(member);
                while (todo) {
                    const int src = __ffsll((unsigned long long)todo) - 1;
                    todo &= todo - 1;
                    const double c0 = __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(c[0]), src));
                    const double c1 = __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(c[1]), src));
                    const double c2 = __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(c[2]), src));
                    const double cs = ax == 0 ? c0 : ax == 1 ? c1 : c2;
                    size_t bb = candidate ? static_cast<size_t>((cs - lo_c) * scale) : 0;
                    if (bb > kBins - 1) bb = kBins - 1;
                    const bool left = (int)bb <= plane;
                    if (left) lcount++; else rcount++;
                    // (min/max by value: the sign of a zero cannot change an area)
                    if (left) {
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            llo[k] = __builtin_fmin(llo[k], value_of(readlane64(lo[k], src)));
                            lhi[k] = __builtin_fmax(lhi[k], value_of(readlane64(hi[k], src)));
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            rlo[k] = __builtin_fmin(rlo[k], value_of(readlane64(lo[k], src)));
                            rhi[k] = __builtin_fmax(rhi[k], value_of(readlane64(hi[k], src)));
                        }
                    }
                }
            } else {
                // bins in LDS (bvh.hpp:259-266), then the sweep of bvh.hpp:268-290 per candidate
                if (lane < kBinSlots) {
                    s_cnt[wave][lane] = 0;
#pragma unroll
                    for (int k = 0; k < 3; k++) { s_lo[wave][lane][k] = kKeyMinInit; s_hi[wave][lane][k] = kKeyMaxInit; }
                }
                wave_sync();
                if (member) {
#pragma unroll
                    for (int x = 0; x < 3; x++) {
                        if (cmin[x] == cmax[x]) continue;
                        const double scale = kBins / (cmax[x] - cmin[x]);
                        size_t bb = static_cast<size_t>((c[x] - cmin[x]) * scale);
                        if (bb > kBins - 1) bb = kBins - 1;
                        const int slot = x * kBins + (int)bb;
                        atomicAdd(&s_cnt[wave][slot], 1u);
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            atomicMin(&s_lo[wave][slot][k], (unsigned long long)lo[k]);
                            atomicMax(&s_hi[wave][slot][k], (unsigned long long)hi[k]);
                        }
                    }
                }
                wave_sync();
                if (candidate) {
                    for (int bb = 0; bb < kBins; bb++) {
                        const int slot = ax * kBins + bb;
                        const int bc = (int)s_cnt[wave][slot];
                        if (!bc) continue;
                        const bool left = bb <= plane;
                        if (left) lcount += bc; else rcount += bc;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const double bl = value_of(s_lo[wave][slot][k]), bh = value_of(s_hi[wave][slot][k]);
                            if (left) { llo[k] = llo[k] < bl ? llo[k] : bl; lhi[k] = lhi[k] > bh ? lhi[k] : bh; }
                            else { rlo[k] = rlo[k] < bl ? rlo[k] : bl; rhi[k] = rhi[k] > bh ? rhi[k] : bh; }
                        }
                    }
                }
                wave_sync();
            }
            double cost = inf, cand_pos = 0;
            if (candidate) {
                const double v = lcount * half_area(llo, lhi) + rcount * half_area(rlo, rhi);
                if (v < inf) cost = v;                   // NaN and +inf never win (bvh.hpp:297)
                const double width = (hi_c - lo_c) / kBins;
                cand_pos = lo_c + width * static_cast<double>(plane + 1);
            }
            // smallest cost; on a tie the candidate the sequential scan meets first
            double bc = cost;
            int bl = lane;
            wave_argmin(bc, bl);
            if (bc < inf) {
                best = bc;
                best_axis = bl / (kBins - 1);
                best_pos = shfl_f64(cand_pos, bl);
            }
        }
        double blo[3], bhi[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { blo[k] = value_of(box[k]); bhi[k] = value_of(box[3 + k]); }
        const double parent_cost = static_cast<double>(cnt) * half_area(blo, bhi);      // bvh.hpp:80-82
        if (best >= parent_cost) best_axis = -1;

        uint32_t child = 0;
        if (best_axis >= 0) {
            // partition (bvh.hpp:128-140) as a permutation of positions
            const double mine = best_axis == 0 ? c[0] : best_axis == 1 ? c[1] : c[2];
            const bool f = member && mine < best_pos;
            const uint32_t rel = pos - a;
            // lane L receives the flag of the triangle at position L
            const uint64_t F = __ballot(__builtin_amdgcn_ds_permute((int)(pos << 2), f ? 1 : 0) != 0) >> a;
            const uint32_t nl = (uint32_t)__popcll(F);
            const uint64_t RA = ~F & low_mask(nl), LB = F & ~low_mask(nl);
            const uint32_t ks = (uint32_t)__popcll(RA);
            if (member) {
                uint32_t dest;
                if (rel < nl) {
                    if (f) dest = rel;
                    else {
                        const uint32_t k = (uint32_t)__popcll(RA & low_mask(rel)) + 1;
                        dest = (k == 1 ? cnt : select_bit(LB, ks - (k - 1) + 1)) - 1;   // (k-1)-th left of B from the back
                    }
                } else {
                    if (f) dest = select_bit(RA, (uint32_t)__popcll(LB >> rel));
                    else if (rel == nl) dest = (ks == 0 ? cnt : select_bit(LB, 1)) - 1;   // ks-th from the back = lowest
                    else dest = rel - 1;
                }
                pos = a + dest;
            }
            if (nl != 0 && nl != cnt) {
                child = next_free;
                next_free += 2;
                if (lane == 0) {
                    nd.first[child] = f0 + a; nd.count[child] = nl; nd.parent[child] = id; nd.link[child] = (run + 1) << 1;
                    nd.first[child + 1] = f0 + a + nl; nd.count[child + 1] = cnt - nl; nd.parent[child + 1] = id; nd.link[child + 1] = 1u;
                }
                // A child of one triangle -- half of all nodes -- is finished here by the lane that holds the triangle:
                // its box is the triangle's, it cannot be split (bvh.hpp:252 skips every axis), so it never goes on the stack.
                const bool left_single = nl == 1, right_single = cnt - nl == 1;
                const bool is_left = left_single && pos == a, is_right = right_single && pos == a + nl;
                if (member && (is_left || is_right)) {
                    const uint32_t cid = is_left ? child : child + 1;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        nd.box[6 * (size_t)cid + k] = lo[k];
                        nd.box[6 * (size_t)cid + 3 + k] = hi[k];
                        nd.zero[3 * (size_t)cid + k] = zinfo[k] ? ((pos * 3 + (zinfo[k] >> 1)) << 1) | (zinfo[k] & 1) : 0u;
                    }
                    nd.child[cid] = 0;
                }
                if (left_single || right_single) deepest = lev + 1 > deepest ? lev + 1 : deepest;
                if (!right_single) {
                    if (lane == sp) { st_range = (a + nl) | (b << 8) | ((lev + 1) << 16); st_id = child + 1; st_run = 0; }
                    sp += 1;
                }
                if (!left_single) {
                    if (lane == sp) { st_range = a | ((a + nl) << 8) | ((lev + 1) << 16); st_id = child; st_run = run + 1; }
                    sp += 1;
                }
            }
        }
        if (lane == 0) nd.child[id] = child;
    }
    if (active) idx[f0 + pos] = tri;
    for (uint32_t t = next_free + lane; t < id_end; t += 64) nd.parent[t] = kInactive;
    if (lane == 0) atomicMax(&counters->deepest, deepest);
}

// Ids a small root may need: 2m-2.
__global__ void subtree_need_kernel(const uint2 *__restrict__ small_list, uint32_t n_small, Nodes nd, uint32_t *__restrict__ need) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_small) need[i] = 2 * nd.count[small_list[i].x] - 2;
}

// Number of inner nodes that start at each position.
__global__ void inner_count_kernel(Nodes nd, uint32_t n_nodes, uint32_t *__restrict__ starts) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t < n_nodes && nd.parent[t] != kInactive && nd.child[t]) atomicAdd(starts + nd.first[t], 1u);
}

// Stored bound of a node: the reduced value, with the sign of a zero taken from the
// last zero coordinate in the reference's accumulation order (its min/max keep the
// later operand on a tie, include/vector3.hpp:92-106).
__device__ __forceinline__ double zero_sign(double v, uint64_t tracker) {
    if (v != 0.0) return v;
    return (tracker & 1) ? 0.0 : -0.0;
}

// Writes the reference's node array: pre-order pair numbering.
__global__ void emit_kernel(Nodes nd, uint32_t n_nodes, Prefix pre, rtx_bvh_node *__restrict__ out) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_nodes || nd.parent[t] == kInactive) return;
    uint64_t where = 0;
    if (t != 0) {
        const uint32_t par = nd.parent[t];
        const uint64_t rank = (uint64_t)pre(nd.first[par]) + (nd.link[par] >> 1);
        where = 1 + 2 * rank + (nd.link[t] & 1);
    }
    rtx_bvh_node o;
    for (int k = 0; k < 3; k++) {
        o.bmin[k] = zero_sign(value_of(nd.box[6 * (size_t)t + k]), nd.zero[3 * (size_t)t + k]);
        o.bmax[k] = zero_sign(value_of(nd.box[6 * (size_t)t + 3 + k]), nd.zero[3 * (size_t)t + k]);
    }
    if (nd.child[t]) {
        o.first = 1 + 2 * ((uint64_t)pre(nd.first[t]) + (nd.link[t] >> 1));
        o.count = 0;
    } else {
        o.first = nd.first[t];
        o.count = nd.count[t];
    }
    out[where] = o;
}

__global__ void widen_kernel(const uint32_t *__restrict__ idx, uint64_t *__restrict__ out, uint32_t n) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p < n) out[p] = idx[p];
}

// The builder's device memory: one arena sized from the triangle count (one allocation and
// one release instead of three dozen, each of which stalls the device), with separate
// allocations only if the estimate falls short.
struct DeviceBuffers {
    std::vector<void *> all;
    char *arena = nullptr;
    size_t arena_bytes = 0, used = 0;
    ~DeviceBuffers() { for (void *p : all) (void)hipFree(p); }
    hipError_t reserve(size_t bytes) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return e;
        all.push_back(p);
        arena = static_cast<char *>(p);
        arena_bytes = bytes;
        used = 0;
        return hipSuccess;
    }
    template <typename T>
    hipError_t get(T **out, size_t count) {
        const size_t bytes = ((count ? count : 1) * sizeof(T) + 255) & ~size_t(255);
        if (used + bytes <= arena_bytes) {
            *out = reinterpret_cast<T *>(arena + used);
            used += bytes;
            return hipSuccess;
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) all.push_back(p);
        *out = static_cast<T *>(p);
        return e;
    }
};

inline uint32_t blocks_for(uint64_t items, uint32_t per_block = kBlock) { return (uint32_t)((items + per_block - 1) / per_block); }

}  // namespace

#define HIP_TRY_B(expr)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) { *err = std::string(#expr) + ": " + hipGetErrorString(e_); return RTX_ERR_DEVICE; } \
    } while (0)

struct DeviceBvh::Impl {
    DeviceBuffers buf;
};
DeviceBvh::DeviceBvh() = default;
DeviceBvh::~DeviceBvh() { delete impl; }

hipError_t upload_pageable(void *dst, const void *src, size_t bytes, hipStream_t stream) {
    constexpr size_t kPiece = 4u << 20;
    constexpr int kThreads = 4, kRing = 2;
    // what was queued on `stream` before comes first, as with a plain copy
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (bytes < 8 * kPiece) {
        e = bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream) : hipSuccess;
        return e == hipSuccess ? hipStreamSynchronize(stream) : e;
    }
    // One host thread fills pinned memory slower than the link drains it: kThreads of them take the pieces in turn, each
    // with two pinned buffers, so that filling one buffer overlaps the DMA of the other.
    static std::mutex turn;
    static void *ring[kThreads][kRing] = {};        // pinned, visible to every device; kept until the process ends
    std::lock_guard<std::mutex> hold(turn);
    for (int t = 0; t < kThreads; t++)
        for (int k = 0; k < kRing; k++)
            if (!ring[t][k]) {
                e = hipHostMalloc(&ring[t][k], kPiece, hipHostMallocPortable);
                if (e != hipSuccess) {
                    // no pinned memory to be had: the runtime's own path for pageable memory, at half the rate
                    ring[t][k] = nullptr;
                    (void)hipGetLastError();
                    e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
                    return e == hipSuccess ? hipStreamSynchronize(stream) : e;
                }
            }
    int device = 0;
    e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    // Every piece goes down the caller's stream (one stream already runs the link at its rate; streams of the library's
    // own would take hardware queues away from the caller's).  The events are made once per device and kept.
    struct Lanes { hipEvent_t landed[kThreads][kRing] = {}; };
    static std::map<int, Lanes> per_device;
    Lanes &lanes = per_device[device];
    for (int t = 0; t < kThreads; t++)
        for (int k = 0; k < kRing; k++)
            if (!lanes.landed[t][k]) {
                e = hipEventCreateWithFlags(&lanes.landed[t][k], hipEventDisableTiming);
                if (e != hipSuccess) { lanes.landed[t][k] = nullptr; return e; }
            }
    const size_t n_pieces = (bytes + kPiece - 1) / kPiece;
    hipError_t result[kThreads];
    auto work = [&](int t) {
        hipError_t r = hipSetDevice(device);
        size_t round = 0;
        for (size_t i = (size_t)t; i < n_pieces && r == hipSuccess; i += kThreads, round++) {
            const int k = (int)(round % kRing);
            const size_t off = i * kPiece, len = std::min(kPiece, bytes - off);
            if (round >= (size_t)kRing) r = hipEventSynchronize(lanes.landed[t][k]);     // the buffer's previous piece has left it
            if (r != hipSuccess) break;
            memcpy(ring[t][k], static_cast<const char *>(src) + off, len);
            r = hipMemcpyAsync(static_cast<char *>(dst) + off, ring[t][k], len, hipMemcpyHostToDevice, stream);
            // recorded after this thread's copy whatever the other threads queue in between
            if (r == hipSuccess) r = hipEventRecord(lanes.landed[t][k], stream);
        }
        result[t] = r;
    };
    std::thread helpers[kThreads - 1];
    for (int t = 1; t < kThreads; t++) helpers[t - 1] = std::thread(work, t);
    work(0);
    for (std::thread &h : helpers) h.join();
    // the buffers are the next caller's: every piece must have left them before the turn ends
    const hipError_t drained = hipStreamSynchronize(stream);
    for (int t = 0; t < kThreads; t++) if (result[t] != hipSuccess) return result[t];
    return drained;
}

// One attempt with room for `top_cap` nodes in the level passes (nodes of more than kSmall triangles and their
// children).  *overflow: the mesh needs more of them; nothing was written out of bounds.
static int build_resident_attempt(const double *vertices, uint32_t n, uint32_t top_cap, DeviceBvh *out, std::string *err, bool *overflow,
                                  std::atomic<int> *vertices_up);

int build_bvh_resident(const double *vertices, uint64_t n64, DeviceBvh *out, std::string *err, std::atomic<int> *vertices_up) {
    struct Release { std::atomic<int> *f; ~Release() { if (f) f->store(1); } } release{vertices_up};   // whatever happens below
    if (n64 == 0) { *err = "cannot build a BVH over zero triangles"; return RTX_ERR_INVALID; }
    if (n64 >= (1ull << 28)) { *err = "more than 2^28 triangles"; return RTX_ERR_UNSUPPORTED; }
    const uint32_t n = (uint32_t)n64;
    // A node of the level passes has more than kSmall triangles or is the child of one.  Meshes as scanners and
    // modelling tools produce them stay far below n / 8 such nodes; the true bound is 2n (a chain of splits that each
    // peel one triangle off a large node), and a mesh that gets past the first estimate is built again with that.
    bool overflow = false;
    uint32_t estimate = n / 8 + 1024;
    if (const char *v = getenv("RTX_BVH_LEVEL_NODES")) estimate = (uint32_t)std::max(4, atoi(v));   // tests: force the second attempt
    int rc = build_resident_attempt(vertices, n, estimate, out, err, &overflow, vertices_up);
    if (overflow) rc = build_resident_attempt(vertices, n, 2 * n + 2, out, err, &overflow, vertices_up);
    if (overflow) { *err = "BVH build: more level nodes than their bound"; return RTX_ERR_DEVICE; }
    return rc;
}

static int build_resident_attempt(const double *vertices, uint32_t n, uint32_t top_cap, DeviceBvh *out, std::string *err, bool *overflow,
                                  std::atomic<int> *vertices_up) {
    *overflow = false;
    const uint32_t cap = top_cap + 2 * n;            // + 2m-2 reserved ids per small root
    const uint32_t chunks = blocks_for(n, kScanChunk);
    delete out->impl;
    out->impl = new DeviceBvh::Impl;
    DeviceBuffers &buf = out->impl->buf;
    // per triangle: 152 B of vertices and per-triangle records, ~40 B of index arrays, 2.1 node records of 92 B,
    // 136 B of output; per level node (n / 8 of them at most) 68 B of split data and 1248 B of bins
    const auto t_reserve = std::chrono::steady_clock::now();
    HIP_TRY_B(buf.reserve((size_t)n * 700 + (size_t)(top_cap > n / 8 + 1024 ? top_cap - n / 8 : 0) * 1400 + (16u << 20)));
    if (getenv("RTX_TIMING"))
        fprintf(stderr, "[rtx timing] %-22s %8.3f ms\n", "arena",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_reserve).count());
    double *V;
    TriRec *T;
    uint32_t *idx[2], *node_of[2], *local, *sums, *holes, *fills, *starts;
    uint8_t *flag;
    uint2 *small_list;
    Counters *counters;
    Nodes nd;
    HIP_TRY_B(buf.get(&V, 9 * (size_t)n));
    HIP_TRY_B(buf.get(&T, n));
    for (int k = 0; k < 2; k++) { HIP_TRY_B(buf.get(&idx[k], n)); HIP_TRY_B(buf.get(&node_of[k], n)); }
    HIP_TRY_B(buf.get(&local, n));
    HIP_TRY_B(buf.get(&sums, (size_t)chunks + 1));
    HIP_TRY_B(buf.get(&holes, n));
    HIP_TRY_B(buf.get(&fills, n));
    HIP_TRY_B(buf.get(&flag, n));
    HIP_TRY_B(buf.get(&counters, 1));
    HIP_TRY_B(buf.get(&small_list, n));
    HIP_TRY_B(buf.get(&starts, n));
    HIP_TRY_B(buf.get(&nd.first, cap)); HIP_TRY_B(buf.get(&nd.count, cap)); HIP_TRY_B(buf.get(&nd.parent, cap));
    HIP_TRY_B(buf.get(&nd.link, cap)); HIP_TRY_B(buf.get(&nd.child, cap));
    HIP_TRY_B(buf.get(&nd.box, 6 * (size_t)cap)); HIP_TRY_B(buf.get(&nd.zero, 3 * (size_t)cap));
    HIP_TRY_B(buf.get(&nd.axis, top_cap)); HIP_TRY_B(buf.get(&nd.pos, top_cap)); HIP_TRY_B(buf.get(&nd.nleft, top_cap));
    HIP_TRY_B(buf.get(&nd.kstar, top_cap)); HIP_TRY_B(buf.get(&nd.cbox, 6 * (size_t)top_cap));
    rtx_bvh_node *d_out;
    uint64_t *d_indices;
    HIP_TRY_B(buf.get(&d_out, 2 * (size_t)n - 1));
    HIP_TRY_B(buf.get(&d_indices, n));

    hipStream_t stream = nullptr;
    hipEvent_t ev0, ev1;
    HIP_TRY_B(hipEventCreate(&ev0));
    HIP_TRY_B(hipEventCreate(&ev1));
    struct EventGuard { hipEvent_t a, b; ~EventGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } guard{ev0, ev1};
    const bool timing = getenv("RTX_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    HIP_TRY_B(upload_pageable(V, vertices, 9 * (size_t)n * sizeof(double), stream));
    if (vertices_up) vertices_up->store(1);
    if (timing)
        fprintf(stderr, "[rtx timing] %-22s %8.3f ms\n", "vertices up",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    HIP_TRY_B(hipEventRecord(ev0, stream));

    const uint32_t pos_blocks = blocks_for(n);
    const uint32_t tiles = blocks_for(n, kFat);
    Counters host_counters{1, 0, 0, 0};
    const bool root_is_small = n <= kSmall;
    if (root_is_small) {
        host_counters.small = 1;
        const uint2 root_entry = make_uint2(0, 0);
        HIP_TRY_B(hipMemcpyAsync(small_list, &root_entry, sizeof root_entry, hipMemcpyHostToDevice, stream));
    }
    HIP_TRY_B(hipMemcpyAsync(counters, &host_counters, sizeof host_counters, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(centroid_kernel, dim3(pos_blocks), dim3(kBlock), 0, stream, V, T, idx[0], node_of[0], n, nd);

    Bins bins{nullptr, nullptr, nullptr};
    uint64_t bins_width = 0;
    const Prefix pre{local, sums};
    uint32_t level_begin = 0, level_end = 1;
    int cur = 0, depth = 0;
    for (uint32_t level = 0; !root_is_small; level++) {
        depth = (int)level;
        // While a level has only a few nodes every block's atomics meet at the same few
        // addresses: then a block walks two tiles and sends one set per node (more tiles per
        // block leave CUs with uneven numbers of blocks, and the LDS atomics of the bin pass
        // are what the level waits for).  Later one tile per block keeps more loads in flight.
        const uint32_t tiles_per_block = ((level_end - level_begin) < 8 && tiles >= 512) ? 2 : 1;
        const uint32_t fat_blocks = blocks_for(tiles, tiles_per_block);
        hipLaunchKernelGGL(bounds_kernel, dim3(fat_blocks), dim3(kFat), 0, stream, T, idx[cur], node_of[cur], n, nd, tiles_per_block);
        const uint32_t width = level_end - level_begin;
        if (width > bins_width) {
            bins_width = std::min<uint64_t>(std::max<uint64_t>(width, 2 * bins_width), top_cap);
            HIP_TRY_B(buf.get(&bins.lo, bins_width * kBinSlots * 3));
            HIP_TRY_B(buf.get(&bins.hi, bins_width * kBinSlots * 3));
            HIP_TRY_B(buf.get(&bins.cnt, bins_width * kBinSlots));
        }
        hipLaunchKernelGGL(bins_clear_kernel, dim3(blocks_for((uint64_t)width * kBinSlots * 3)), dim3(kBlock), 0, stream, bins, width);
        const uint32_t node_blocks = blocks_for(width);
        hipLaunchKernelGGL(bin_kernel, dim3(fat_blocks), dim3(kFat), 0, stream, T, idx[cur], node_of[cur], n, nd, bins, level_begin,
                           tiles_per_block);
        hipLaunchKernelGGL(plane_kernel, dim3(blocks_for((uint64_t)width * 4)), dim3(kBlock), 0, stream, nd, bins, level_begin, level_end);
        hipLaunchKernelGGL(flag_kernel, dim3(chunks), dim3(kBlock), 0, stream, T, idx[cur], node_of[cur], n, nd, flag, local, sums);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, stream, sums, chunks);
        hipLaunchKernelGGL(split_kernel, dim3(node_blocks), dim3(kBlock), 0, stream, nd, pre, flag, level_begin, level_end, counters,
                           small_list, level, top_cap);
        hipLaunchKernelGGL(holes_kernel, dim3(pos_blocks), dim3(kBlock), 0, stream, node_of[cur], n, nd, pre, flag, holes, fills);
        hipLaunchKernelGGL(permute_kernel, dim3(pos_blocks), dim3(kBlock), 0, stream, idx[cur], node_of[cur], n, nd, pre, flag, holes,
                           fills, idx[cur ^ 1], node_of[cur ^ 1]);
        cur ^= 1;
        const uint32_t big_before = host_counters.big;
        HIP_TRY_B(hipMemcpyAsync(&host_counters, counters, sizeof host_counters, hipMemcpyDeviceToHost, stream));
        HIP_TRY_B(hipStreamSynchronize(stream));
        if (host_counters.next_id > top_cap) { *overflow = true; return RTX_OK; }
        if (host_counters.next_id != level_end) depth = (int)level + 1;
        if (host_counters.big == big_before) break;
        level_begin = level_end;
        level_end = host_counters.next_id;
    }
    uint32_t total_ids = host_counters.next_id;
    const uint32_t n_small = host_counters.small;
    if (n_small) {
        const uint32_t small_chunks = blocks_for(n_small, kScanChunk);
        hipLaunchKernelGGL(subtree_need_kernel, dim3(blocks_for(n_small)), dim3(kBlock), 0, stream, small_list, n_small, nd, local);
        hipLaunchKernelGGL(local_scan_kernel, dim3(small_chunks), dim3(kBlock), 0, stream, local, n_small, sums);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, stream, sums, small_chunks);
        hipLaunchKernelGGL(subtree_kernel, dim3(blocks_for(n_small, kBlock / 64)), dim3(kBlock), 0, stream, T, idx[cur], small_list,
                           local, sums, n_small, host_counters.next_id, nd, counters);
        uint32_t reserved = 0;
        HIP_TRY_B(hipMemcpyAsync(&reserved, sums + small_chunks, sizeof reserved, hipMemcpyDeviceToHost, stream));
        HIP_TRY_B(hipMemcpyAsync(&host_counters, counters, sizeof host_counters, hipMemcpyDeviceToHost, stream));
        HIP_TRY_B(hipStreamSynchronize(stream));
        total_ids += reserved;
        if (total_ids > cap) { *err = "BVH build: node ids overran their bound"; return RTX_ERR_DEVICE; }
        depth = std::max<int>(depth, (int)host_counters.deepest);
    }
    HIP_TRY_B(hipMemsetAsync(starts, 0, (size_t)n * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(inner_count_kernel, dim3(blocks_for(total_ids)), dim3(kBlock), 0, stream, nd, total_ids, starts);
    hipLaunchKernelGGL(local_scan_kernel, dim3(chunks), dim3(kBlock), 0, stream, starts, n, sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, stream, sums, chunks);
    hipLaunchKernelGGL(emit_kernel, dim3(blocks_for(total_ids)), dim3(kBlock), 0, stream, nd, total_ids, Prefix{starts, sums}, d_out);
    hipLaunchKernelGGL(widen_kernel, dim3(pos_blocks), dim3(kBlock), 0, stream, idx[cur], d_indices, n);
    uint32_t inner = 0;
    HIP_TRY_B(hipMemcpyAsync(&inner, sums + chunks, sizeof inner, hipMemcpyDeviceToHost, stream));
    HIP_TRY_B(hipEventRecord(ev1, stream));
    HIP_TRY_B(hipStreamSynchronize(stream));
    HIP_TRY_B(hipGetLastError());
    const uint64_t n_nodes = 1 + 2 * (uint64_t)inner;
    if (n_nodes > 2 * (uint64_t)n - 1) { *err = "BVH build: inconsistent node count"; return RTX_ERR_DEVICE; }
    float ms = 0;
    HIP_TRY_B(hipEventElapsedTime(&ms, ev0, ev1));
    out->nodes = d_out;
    out->indices = idx[cur];
    out->indices64 = d_indices;
    out->vertices = V;
    out->n_triangles = n;
    out->n_nodes = n_nodes;
    out->depth = depth;
    out->device_ms = ms;
    return RTX_OK;
}

int build_bvh_device(const double *vertices, uint64_t n, rtx_bvh_node *nodes_out, uint64_t *indices_out,
                     uint64_t *n_nodes_out, int32_t *depth_out, double *device_ms_out, std::string *err) {
    DeviceBvh bvh;
    if (int rc = build_bvh_resident(vertices, n, &bvh, err)) return rc;
    HIP_TRY_B(hipMemcpy(nodes_out, bvh.nodes, (size_t)bvh.n_nodes * sizeof(rtx_bvh_node), hipMemcpyDeviceToHost));
    HIP_TRY_B(hipMemcpy(indices_out, bvh.indices64, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    *n_nodes_out = bvh.n_nodes;
    *depth_out = bvh.depth;
    *device_ms_out = bvh.device_ms;
    return RTX_OK;
}

// ---- from the reference's arrays to the render kernel's layout, on the device -------------

namespace {

// Slot = reference index + 1 (slot 0: a box of NaNs that no ray enters); a leaf stamps its
// `first` on its triangles; boxes outside the range the fast slab division admits clear *tame.
__global__ void flatten_nodes_kernel(const rtx_bvh_node *__restrict__ in, uint32_t n_nodes, DevNode *__restrict__ out,
                                     DevTri *__restrict__ tris, int32_t *__restrict__ tame) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    // {first, count} of reference node k as the render layout has them
    auto first_of = [](const rtx_bvh_node &h) { return h.count == 0 ? (uint32_t)(h.first + 1) : (uint32_t)h.first; };
    if (i == 0) {
        DevNode d{};
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        for (int k = 0; k < 3; k++) { d.bmin[k] = nan; d.bmax[k] = nan; }
        d.pad[0] = 0;                          // slot 0 pairs with the root
        d.pad[1] = node_entry(1u, first_of(in[0]), (uint32_t)in[0].count);
        out[0] = d;
    }
    if (i >= n_nodes) return;
    const rtx_bvh_node h = in[i];
    DevNode d{};
    bool ok = true;
    for (int k = 0; k < 3; k++) {
        d.bmin[k] = h.bmin[k];
        d.bmax[k] = h.bmax[k];
        const double a = fabs(h.bmin[k]), b = fabs(h.bmax[k]);
        ok = ok && (a == 0.0 || (a >= 0x1p-300 && a <= 0x1p300)) && (b == 0.0 || (b >= 0x1p-300 && b <= 0x1p300)) &&
             h.bmin[k] <= h.bmax[k];
    }
    d.count = (uint32_t)h.count;
    d.first = first_of(h);
    // an even slot (i + 1 even) is the left node of a sibling pair: the references to itself and to its sibling
    if ((i & 1u) && i + 1 < n_nodes) {
        const rtx_bvh_node sib = in[i + 1];
        d.pad[0] = node_entry(i + 1u, d.first, d.count);
        d.pad[1] = node_entry(i + 2u, first_of(sib), (uint32_t)sib.count);
    }
    out[i + 1] = d;
    if (!ok) atomicAnd(tame, 0);
    for (uint64_t j = 0; j < h.count; j++) tris[h.first + j].leaf_first = (uint32_t)h.first;
}

// Triangle records in BVH::indices order; the Triangle ctor's normal (triangle.hpp:21-23)
// when the caller gave none, and d = -(n . v0) (triangle.hpp:39).
__global__ void flatten_tris_kernel(const double *__restrict__ V, const double *__restrict__ normals,
                                    const int32_t *__restrict__ colors, const int32_t *__restrict__ spec,
                                    const double *__restrict__ refl, const uint32_t *__restrict__ idx, uint32_t n,
                                    DevTri *__restrict__ tris, DevMat *__restrict__ mats) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t id = idx[p];
    const double *v = V + 9 * (size_t)id;
    DevTri t{};
    for (int k = 0; k < 3; k++) { t.v0[k] = v[k]; t.v1[k] = v[3 + k]; t.v2[k] = v[6 + k]; }
    if (normals) {
        for (int k = 0; k < 3; k++) t.n[k] = normals[3 * (size_t)id + k];
    } else {
        const double ax = v[3] - v[0], ay = v[4] - v[1], az = v[5] - v[2];
        const double bx = v[6] - v[0], by = v[7] - v[1], bz = v[8] - v[2];
        const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        const double len = sqrt(cx * cx + cy * cy + cz * cz);
        t.n[0] = cx / len;
        t.n[1] = cy / len;
        t.n[2] = cz / len;
    }
    t.d = -(t.n[0] * v[0] + t.n[1] * v[1] + t.n[2] * v[2]);
    t.orig_id = id;
    const uint32_t keep = tris[p].leaf_first;      // written by flatten_nodes_kernel
    t.leaf_first = keep;
    tris[p] = t;
    DevMat m{};
    for (int k = 0; k < 3; k++) m.color[k] = colors[3 * (size_t)id + k];
    m.specularity = spec[id];
    m.reflectivity = refl[id];
    mats[p] = m;
}

}  // namespace

int flatten_scene_device(const DeviceBvh &bvh, const double *d_normals, const int32_t *d_colors, const int32_t *d_spec,
                         const double *d_refl, DevNode *d_nodes, DevTri *d_tris, DevMat *d_mats, int32_t *tame_out,
                         double root_box[6], std::string *err) {
    int32_t *d_tame = nullptr;
    HIP_TRY_B(hipMalloc(&d_tame, sizeof(int32_t)));
    struct Free { void *p; ~Free() { (void)hipFree(p); } } guard{d_tame};
    const int32_t one = 1;
    HIP_TRY_B(hipMemcpy(d_tame, &one, sizeof one, hipMemcpyHostToDevice));
    const uint32_t n = (uint32_t)bvh.n_triangles, m = (uint32_t)bvh.n_nodes;
    hipLaunchKernelGGL(flatten_nodes_kernel, dim3(blocks_for(m)), dim3(kBlock), 0, nullptr, bvh.nodes, m, d_nodes, d_tris, d_tame);
    hipLaunchKernelGGL(flatten_tris_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, nullptr, bvh.vertices, d_normals, d_colors, d_spec,
                       d_refl, bvh.indices, n, d_tris, d_mats);
    HIP_TRY_B(hipGetLastError());
    rtx_bvh_node root;
    HIP_TRY_B(hipMemcpy(&root, bvh.nodes, sizeof root, hipMemcpyDeviceToHost));
    HIP_TRY_B(hipMemcpy(tame_out, d_tame, sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; k++) { root_box[k] = root.bmin[k]; root_box[3 + k] = root.bmax[k]; }
    return RTX_OK;
}

}  // namespace rtx
