// The render path on the GPU (gfx950): one pixel per lane, 8x8 pixels per wave.
//
// Each lane runs the reference's per-pixel work -- camera ray (src/main.cpp:78-82),
// Scene::traceRay with its reflection chain (include/scene.hpp:89-123), the shadow
// query and lighting of computeLighting (scene.hpp:147-183) -- in FP64 with the
// reference's operation order.  This file is compiled with -ffp-contract=off: a
// fused multiply-add would round differently from the reference's separate
// multiply and add.
//
// BVH::intersectRay's recursion (include/bvh.hpp:164-211) becomes an iterative walk
// with a per-lane stack of node slots in LDS (one u32 per level, lanes adjacent:
// conflict-free).  An inner node's two children are one 128-byte line and are
// slab-tested together.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "rtx_kernels.h"

namespace rtx {

namespace {

#define RTX_INF (__builtin_inf())
// include/common.hpp:12: DBL_EPSILON * 2^20 = 2^-32
#define RTX_EPS (2.3283064365386962890625e-10)

struct V3 {
    double x, y, z;
};

__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // vector3.hpp:42-44
__device__ __forceinline__ V3 cross(V3 a, V3 b) {                                                  // vector3.hpp:46-52
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 smul(double s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
__device__ __forceinline__ V3 muls(V3 v, double s) { return {v.x * s, v.y * s, v.z * s}; }
__device__ __forceinline__ double mag(V3 v) { return __builtin_sqrt(dot(v, v)); }                 // vector3.hpp:84-86
// std::min / std::max on doubles as bvh.hpp:220-231 uses them (NOT fmin/fmax: with a
// NaN operand these return the first argument)
__device__ __forceinline__ double smin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double smax(double a, double b) { return (a < b) ? b : a; }

__device__ __forceinline__ V3 ld3(const double *p) { return {p[0], p[1], p[2]}; }

// ---- exact division by a per-ray constant --------------------------------------
// The slab test divides twelve box distances per node pair by the same three ray
// components (bvh.hpp:218-229).  hipcc expands an IEEE f64 division into
//   y0 = rcp(b); e = fma(-b,y0,1); y1 = fma(y0,e,y0); e = fma(-b,y1,1); y = fma(y1,e,y1);
//   q0 = a*y; r = fma(-b,q0,a); q = fma(r,y,q0)
// wrapped in v_div_scale / v_div_fmas / v_div_fixup, which only act on operands near
// the ends of the exponent range, on zeros, infinities and NaNs.  The first line
// depends on the divisor alone: RayDiv keeps y per ray component and the slab test
// runs the second line only -- the same operations on the same values, hence the same
// correctly rounded quotient -- provided every operand stays far from the exponent
// range's ends (`ok`, checked per ray and per scene) or the divisor is zero (then
// a / (+-0) = a * (+-inf) exactly, NaN for a = 0 included).
struct RayDiv {
    double nbx, nby, nbz;   // -r
    double yx, yy, yz;      // refined reciprocals (+-inf for a zero component)
    bool zx, zy, zz;        // component == 0
    bool ok;                // fast path valid for this ray
};

__device__ __forceinline__ bool exp_safe(double v) {
    // 0, or 2^-300 <= |v| <= 2^300
    const uint32_t e = ((uint32_t)__double2hiint(v) >> 20) & 0x7ffu;
    return v == 0.0 || (e >= 1023u - 300u && e <= 1023u + 300u);
}

__device__ __forceinline__ double refined_rcp(double b) {
    double y0 = __builtin_amdgcn_rcp(b);
    double e0 = __builtin_fma(-b, y0, 1.0);
    double y1 = __builtin_fma(y0, e0, y0);
    double e1 = __builtin_fma(-b, y1, 1.0);
    return __builtin_fma(y1, e1, y1);
}

__device__ __forceinline__ RayDiv make_raydiv(V3 o, V3 r, bool scene_ok) {
    RayDiv d;
    d.nbx = -r.x; d.nby = -r.y; d.nbz = -r.z;
    d.zx = r.x == 0.0; d.zy = r.y == 0.0; d.zz = r.z == 0.0;
    d.yx = d.zx ? __builtin_copysign(RTX_INF, r.x) : refined_rcp(r.x);
    d.yy = d.zy ? __builtin_copysign(RTX_INF, r.y) : refined_rcp(r.y);
    d.yz = d.zz ? __builtin_copysign(RTX_INF, r.z) : refined_rcp(r.z);
    d.ok = scene_ok && exp_safe(o.x) && exp_safe(o.y) && exp_safe(o.z) && exp_safe(r.x) && exp_safe(r.y) && exp_safe(r.z);
    return d;
}

// a / b given nb = -b, y = refined_rcp(b), zero = (b == 0)
__device__ __forceinline__ double div_by(double a, double nb, double y, bool zero) {
    double q0 = a * y;
    double rem = __builtin_fma(nb, q0, a);
    double q1 = __builtin_fma(rem, y, q0);
    return zero ? q0 : q1;
}

// BVH::intersectRayAABB with tLim = +inf (bvh.hpp:213-234; both call sites pass a
// local closestT that is still INFINITY, bvh.hpp:170,174).  *tnear is the entry
// distance the test computed.
__device__ __forceinline__ bool slab(const double *lo, const double *hi, V3 o, V3 r, double *tnear) {
    double tx1 = (lo[0] - o.x) / r.x, tx2 = (hi[0] - o.x) / r.x;
    double tmin = smin(tx1, tx2), tmax = smax(tx1, tx2);
    double ty1 = (lo[1] - o.y) / r.y, ty2 = (hi[1] - o.y) / r.y;
    tmin = smax(tmin, smin(ty1, ty2));
    tmax = smin(tmax, smax(ty1, ty2));
    double tz1 = (lo[2] - o.z) / r.z, tz2 = (hi[2] - o.z) / r.z;
    tmin = smax(tmin, smin(tz1, tz2));
    tmax = smin(tmax, smax(tz1, tz2));
    *tnear = tmin;
    return tmin < RTX_INF && tmax > 0 && tmax >= tmin;
}

// The same test with the divisions taken through RayDiv.
__device__ __forceinline__ bool slab_fast(const double *lo, const double *hi, V3 o, const RayDiv &d, double *tnear) {
    double tx1 = div_by(lo[0] - o.x, d.nbx, d.yx, d.zx), tx2 = div_by(hi[0] - o.x, d.nbx, d.yx, d.zx);
    double tmin = smin(tx1, tx2), tmax = smax(tx1, tx2);
    double ty1 = div_by(lo[1] - o.y, d.nby, d.yy, d.zy), ty2 = div_by(hi[1] - o.y, d.nby, d.yy, d.zy);
    tmin = smax(tmin, smin(ty1, ty2));
    tmax = smin(tmax, smax(ty1, ty2));
    double tz1 = div_by(lo[2] - o.z, d.nbz, d.yz, d.zz), tz2 = div_by(hi[2] - o.z, d.nbz, d.yz, d.zz);
    tmin = smax(tmin, smin(tz1, tz2));
    tmax = smin(tmax, smax(tz1, tz2));
    *tnear = tmin;
    return tmin < RTX_INF && tmax > 0 && tmax >= tmin;
}

// Triangle::intersectRay (include/triangle.hpp:32-55); d = -(n.v0) comes precomputed.
__device__ __forceinline__ bool tri_hit(const DevTri &tr, V3 o, V3 r, double *t_out) {
    V3 n = ld3(tr.n);
    double nd = dot(n, r);
    if (__builtin_fabs(nd) < RTX_EPS) return false;
    double t = -(dot(n, o) + tr.d) / nd;
    V3 p = add(o, smul(t, r));
    V3 v0 = ld3(tr.v0), v1 = ld3(tr.v1), v2 = ld3(tr.v2);
    *t_out = t;
    if (!(dot(n, cross(sub(v1, v0), sub(p, v0))) > RTX_EPS)) return false;
    if (!(dot(n, cross(sub(v2, v1), sub(p, v1))) > RTX_EPS)) return false;
    return dot(n, cross(sub(v0, v2), sub(p, v2))) > RTX_EPS;
}

struct Counters {
    unsigned long long v[9];
};
enum { C_PRIMARY, C_REFLECTION, C_SHADOW, C_OCCLUDED, C_CLOSEST, C_NODES, C_BOXHITS, C_LEAVES, C_TRIS };

struct Hit {
    double t;
    int32_t pos;      // position in the device triangle array, -1 = miss
};

// Closest hit over the BVH.
//   PRUNE = false: visits exactly the node set of the reference's recursion (a child is
//           entered iff its slab test passes) and keeps scanning after a hit.
//   PRUNE = true : near child first, skips a child whose entry distance is beyond the
//           best hit so far (+ margin, see prune_margin) and, when `any` is set (a
//           shadow query only asks whether a hit exists, scene.hpp:155-163), stops
//           at the first accepted triangle.  Results are the same.
// Result definition (bvh.hpp:178-208): smallest t among triangles with hit && tmin < t
// < tmax; on equal t the triangle of the LATER leaf (larger leaf `first`) wins, inside
// one leaf the earlier position wins.
template <bool PRUNE, bool STATS, bool FAST>
__device__ __forceinline__ Hit traverse(const DevScene &sc, V3 o, V3 r, const RayDiv &rd, double tmin, double tmax, bool any,
                                        double margin, uint32_t *stack, uint32_t stride, Counters &c) {
    Hit best{RTX_INF, -1};
    uint32_t best_leaf = 0;
    const DevNode *nodes = sc.nodes;
    double tn;
    if (STATS) c.v[C_NODES]++;
    if (!(FAST ? slab_fast(nodes[1].bmin, nodes[1].bmax, o, rd, &tn) : slab(nodes[1].bmin, nodes[1].bmax, o, r, &tn))) return best;
    if (STATS) c.v[C_BOXHITS]++;
    uint32_t first = nodes[1].first, count = nodes[1].count;
    uint32_t sp = 0;
    double limit = tmax;   // PRUNE: no accepted t can be >= tmax
    for (;;) {
        if (count == 0) {
            const DevNode &L = nodes[first];
            const DevNode &R = nodes[first + 1];
            double tl, trr;
            bool hl = FAST ? slab_fast(L.bmin, L.bmax, o, rd, &tl) : slab(L.bmin, L.bmax, o, r, &tl);
            bool hr = FAST ? slab_fast(R.bmin, R.bmax, o, rd, &trr) : slab(R.bmin, R.bmax, o, r, &trr);
            if (STATS) { c.v[C_NODES] += 2; c.v[C_BOXHITS] += (hl ? 1 : 0) + (hr ? 1 : 0); }
            if (PRUNE) {
                hl = hl && !(tl > limit + margin);
                hr = hr && !(trr > limit + margin);
            }
            uint32_t lf = L.first, lc = L.count, rf = R.first, rc = R.count;
            if (hl && hr) {
                bool right_first = PRUNE && (trr < tl);
                stack[sp * stride] = right_first ? first : first + 1;
                sp++;
                first = right_first ? rf : lf;
                count = right_first ? rc : lc;
                continue;
            } else if (hl) {
                first = lf; count = lc;
                continue;
            } else if (hr) {
                first = rf; count = rc;
                continue;
            }
        } else {
            if (STATS) { c.v[C_LEAVES]++; c.v[C_TRIS] += count; }
            for (uint32_t i = 0; i < count; i++) {
                const DevTri &tr = sc.tris[first + i];
                double t;
                if (tri_hit(tr, o, r, &t) && t > tmin && t < tmax) {
                    if (PRUNE && any) return Hit{t, (int32_t)(first + i)};
                    uint32_t lfirst = first;   // == tr.leaf_first
                    bool better = t < best.t || (t == best.t && (lfirst > best_leaf));
                    // same leaf, equal t: the earlier position stays (strict <, bvh.hpp:186)
                    if (better) {
                        best.t = t; best.pos = (int32_t)(first + i); best_leaf = lfirst;
                        if (PRUNE) limit = t;
                    }
                }
            }
        }
        if (sp == 0) break;
        sp--;
        uint32_t slot = stack[sp * stride];
        first = nodes[slot].first;
        count = nodes[slot].count;
    }
    return best;
}

__device__ __forceinline__ V3 reflect_ray(V3 ray, V3 n) {              // scene.hpp:185-187
    return sub(muls(smul(2.0, n), dot(n, ray)), ray);
}

// Distance slack for PRUNE.  A triangle's t comes from the plane equation
// (triangle.hpp:40) and a box's entry distance from three divisions (bvh.hpp:218-231);
// both describe the same geometric fact with different roundings.  The accepted hit
// point lies within rounding distance of its triangle, hence of every box that holds
// it; along the ray that is a t-difference of at most (point slack) / |r_axis| on the
// axis that bounds the entry.  The slack below is a 2^-20 relative bound on the
// point, 2^32 times the rounding unit, the same factor the reference's own EPSILON
// uses.
__device__ __forceinline__ double prune_margin(V3 o, V3 r, double scene_scale) {
    double ax = __builtin_fabs(r.x), ay = __builtin_fabs(r.y), az = __builtin_fabs(r.z);
    double m = RTX_INF;
    if (ax > 0 && ax < m) m = ax;
    if (ay > 0 && ay < m) m = ay;
    if (az > 0 && az < m) m = az;
    double reach = scene_scale + __builtin_fabs(o.x) + __builtin_fabs(o.y) + __builtin_fabs(o.z);
    return (reach * 9.5367431640625e-07) / m;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }   // color.hpp:23-29

// One lane = one pixel.  The lane runs a small state machine around ONE traversal
// loop: a closest-hit query (kind 0) is followed by one shadow query per light
// (kind 1, scene.hpp:155), then either the path ends or the reflected ray
// (scene.hpp:119-120) becomes the next closest-hit query.  The (localColor,
// reflectivity) pairs waiting for their reflected colour (scene.hpp:122) sit in LDS
// behind the traversal stack.
template <bool PRUNE, bool STATS>
__global__ void __launch_bounds__(256) render_kernel(const DevScene sc, const DevRender rp) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t blocks_x = (uint32_t)rp.tile_w >> 3, blocks_y = (uint32_t)rp.tile_h >> 3;
    const uint32_t per_tile = blocks_x * blocks_y;
    const uint64_t gw = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint32_t local_tile = (uint32_t)(gw / per_tile);
    const uint32_t wb = (uint32_t)(gw % per_tile);
    if (local_tile >= (uint32_t)rp.owned_tiles) return;
    const uint32_t tile = (rp.tile_count > 1) ? local_tile * (uint32_t)rp.tile_count + (uint32_t)rp.tile_rank : local_tile;
    const uint32_t tx = tile % (uint32_t)rp.tiles_x, ty = tile / (uint32_t)rp.tiles_x;
    const uint32_t in_x = (wb % blocks_x) * 8u + (lane & 7u), in_y = (wb / blocks_x) * 8u + (lane >> 3);
    const int col = (int)(tx * (uint32_t)rp.tile_w + in_x), row = (int)(ty * (uint32_t)rp.tile_h + in_y);
    if (col < rp.x0 || col >= rp.x1 || row < rp.y0 || row >= rp.y1) return;

    const uint32_t stride = rp.stack_stride;
    uint32_t *stack = lds + threadIdx.x;
    uint32_t *pend = lds + (uint32_t)(sc.bvh_depth + 1) * stride + threadIdx.x;   // [level][5][lane]
    Counters c;
    if (STATS) for (int i = 0; i < 9; i++) c.v[i] = 0;

    // putPixel inverted (src/main.cpp:90-94): row = -y + H/2 - 1, col = x + W/2
    const int x = col - rp.W / 2, y = rp.H / 2 - 1 - row;
    const uint64_t k = (uint64_t)row * (uint64_t)rp.W + (uint64_t)col;
    const uint32_t n_lights = sc.n_point + sc.n_dir;

    int sum_r = 0, sum_g = 0, sum_b = 0;
    int first_r = 0, first_g = 0, first_b = 0;
    for (int s = 0; s < rp.spp; s++) {
        // Scene::toViewport (scene.hpp:81-87); samples > 0 add a jitter in [0,1)^2
        double fx = (double)x, fy = (double)y;
        if (s > 0) {
            uint64_t z = splitmix64(splitmix64(rp.seed) ^ k);
            z = splitmix64(z ^ ((uint64_t)(uint32_t)s << 32));
            fx = fx + (double)(z >> 32) * (1.0 / 4294967296.0);
            fy = fy + (double)(z & 0xFFFFFFFFull) * (1.0 / 4294967296.0);
        }
        V3 vp = {fx * rp.px, fy * rp.py, sc.viewport[2]};
        // rotateX, rotateY, rotateZ (vector3.hpp:54-82)
        V3 rx = {vp.x, sc.cx * vp.y - sc.sx * vp.z, sc.sx * vp.y + sc.cx * vp.z};
        V3 ry = {sc.cy * rx.x + sc.sy * rx.z, rx.y, (-sc.sy) * rx.x + sc.cy * rx.z};
        V3 r = {sc.cz * ry.x - sc.sz * ry.y, sc.sz * ry.x + sc.cz * ry.y, ry.z};
        V3 o = {sc.cam[0], sc.cam[1], sc.cam[2]};
        double tmin = 1.0, tmax = RTX_INF;                              // src/main.cpp:83

        int kind = 0;          // 0 closest-hit query, 1 shadow query for light `li`
        int d = 0;             // reflections so far = entries in pend[]
        uint32_t li = 0;
        V3 v = {0, 0, 0};      // -ray of the surface hit being shaded
        int32_t pos = -1;      // its triangle (device position)
        double total = 0.0;    // computeTotalLighting accumulator
        int cr = sc.bg[0], cg = sc.bg[1], cb = sc.bg[2];
        for (;;) {
            const double margin = PRUNE ? prune_margin(o, r, sc.scene_scale) : 0.0;
            const RayDiv rd = make_raydiv(o, r, sc.fast_div_ok != 0);
            // the plain-division walk only runs for rays with components at the ends of the exponent range
            Hit h = rd.ok ? traverse<PRUNE, STATS, true>(sc, o, r, rd, tmin, tmax, kind == 1, margin, stack, stride, c)
                          : traverse<PRUNE, STATS, false>(sc, o, r, rd, tmin, tmax, kind == 1, margin, stack, stride, c);
            if (kind == 0) {
                if (STATS) c.v[d == 0 ? C_PRIMARY : C_REFLECTION]++;
                if (d == 0 && s == 0) {
                    if (rp.hit) rp.hit[k] = (h.pos >= 0) ? (int32_t)sc.tris[h.pos].orig_id : -1;
                    if (rp.t) rp.t[k] = h.t;
                }
                if (h.pos < 0) break;                                   // backgroundColor, scene.hpp:101-103
                if (STATS) c.v[C_CLOSEST]++;
                o = add(o, smul(h.t, r));                               // point, scene.hpp:105
                v = muls(r, -1.0);                                      // -ray, vector3.hpp:108-110
                pos = h.pos;
                total = sc.ambient;                                     // scene.hpp:128
                li = 0;
            } else {
                // the rest of Scene::computeLighting (scene.hpp:159-182) with light = r, point = o
                if (STATS) { c.v[C_SHADOW]++; if (h.pos >= 0) c.v[C_OCCLUDED]++; }
                if (h.pos < 0) {
                    const V3 n = ld3(sc.tris[pos].n);
                    const int spec = sc.mats[pos].specularity;
                    const double intensity = sc.lights[li].intensity;
                    double term = 0.0;
                    double nl = dot(n, r);
                    if (nl > 0.0) term += intensity * (nl / (mag(n) * mag(r)));
                    if (spec != -1) {
                        V3 R = reflect_ray(r, n);
                        double rv = dot(R, v);
                        if (rv > 0) {
                            double cosv = rv / (mag(R) * mag(v));
                            // pow(x, 1) is x exactly in every libm; other exponents go through
                            // the device pow, which may differ from glibc's in the last place
                            term += intensity * ((spec == 1) ? cosv : pow(cosv, (double)spec));
                        }
                    }
                    total += term;
                }
                li++;
            }
            if (li < n_lights) {
                // computeTotalLighting (scene.hpp:130-142): point lights first (light = position -
                // point, tMax = 1), then directional (light = direction, tMax = inf)
                const bool is_point = li < sc.n_point;
                V3 lv = ld3(sc.lights[li].v);
                r = is_point ? sub(lv, o) : lv;
                tmin = 0.001;
                tmax = is_point ? 1.0 : RTX_INF;
                kind = 1;
                continue;
            }
            // Color * double truncates each channel (color.hpp:47-53)
            const DevMat &mt = sc.mats[pos];
            const int lr = (int)(mt.color[0] * total), lg = (int)(mt.color[1] * total), lb = (int)(mt.color[2] * total);
            const double refl = mt.reflectivity;
            cr = lr; cg = lg; cb = lb;
            if (d == rp.depth || refl < 0) break;                       // scene.hpp:115-117
            uint32_t *e = pend + (uint32_t)d * 5u * stride;
            e[0] = (uint32_t)lr; e[stride] = (uint32_t)lg; e[2 * stride] = (uint32_t)lb;
            e[3 * stride] = (uint32_t)__double2loint(refl); e[4 * stride] = (uint32_t)__double2hiint(refl);
            d++;
            r = reflect_ray(v, ld3(sc.tris[pos].n));                    // scene.hpp:119
            tmin = 0.001; tmax = RTX_INF;
            kind = 0;
            cr = sc.bg[0]; cg = sc.bg[1]; cb = sc.bg[2];
        }
        // localColor * (1 - reflectivity) + reflectedColor * reflectivity (scene.hpp:122), innermost first
        while (d > 0) {
            d--;
            const uint32_t *e = pend + (uint32_t)d * 5u * stride;
            const int lr = (int)e[0], lg = (int)e[stride], lb = (int)e[2 * stride];
            const double refl = __hiloint2double((int)e[4 * stride], (int)e[3 * stride]);
            const double keep = 1 - refl;
            cr = (int)(lr * keep) + (int)(cr * refl);
            cg = (int)(lg * keep) + (int)(cg * refl);
            cb = (int)(lb * keep) + (int)(cb * refl);
        }
        if (s == 0) { first_r = cr; first_g = cg; first_b = cb; }
        sum_r += clamp255(cr); sum_g += clamp255(cg); sum_b += clamp255(cb);
    }

    uint64_t out = k;
    if (rp.layout == RTX_LAYOUT_PACKED_TILES_)
        out = ((uint64_t)local_tile * (uint32_t)rp.tile_h + in_y) * (uint32_t)rp.tile_w + in_x;
    if (rp.rgb) {
        const int spp = rp.spp;
        rp.rgb[3 * out + 0] = (uint8_t)((sum_r + spp / 2) / spp);
        rp.rgb[3 * out + 1] = (uint8_t)((sum_g + spp / 2) / spp);
        rp.rgb[3 * out + 2] = (uint8_t)((sum_b + spp / 2) / spp);
    }
    if (rp.rgb_i32) {
        rp.rgb_i32[3 * out + 0] = rp.spp == 1 ? first_r : sum_r;
        rp.rgb_i32[3 * out + 1] = rp.spp == 1 ? first_g : sum_g;
        rp.rgb_i32[3 * out + 2] = rp.spp == 1 ? first_b : sum_b;
    }
    if (STATS && rp.stats) {
        for (int i = 0; i < 9; i++)
            if (c.v[i]) atomicAdd(&rp.stats[i], c.v[i]);
    }
}

// Self-test: div_by against the compiler's division on seeded operands, bitwise.
__global__ void __launch_bounds__(256) divtest_kernel(uint64_t seed, uint64_t n, unsigned long long *mismatches) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = splitmix64(seed ^ (i * 0x9E3779B97F4A7C15ull));
    uint64_t w = splitmix64(z);
    // mantissas random, exponents in the range the fast path admits, both signs; every 64th divisor is +-0
    // and every 97th dividend is 0
    uint64_t ea = 1023 - 300 + (splitmix64(w) % 601), eb = 1023 - 300 + (splitmix64(w ^ 0x55) % 601);
    if ((i & 7) < 6) { ea = 1023 - 12 + (splitmix64(w) % 24); eb = 1023 - 12 + (splitmix64(w ^ 0x55) % 24); }
    uint64_t abits = (z & 0x800FFFFFFFFFFFFFull) | (ea << 52), bbits = (w & 0x800FFFFFFFFFFFFFull) | (eb << 52);
    if ((i & 31) == 5) bbits |= 0x000FFFFFFFFFFFFFull;          // all-ones significand
    if ((i & 31) == 6) bbits &= 0xFFF0000000000000ull;          // power of two
    double a = __longlong_as_double((long long)abits), b = __longlong_as_double((long long)bbits);
    if (i % 64 == 63) b = (i & 64) ? 0.0 : -0.0;
    if (i % 97 == 96) a = (i & 128) ? 0.0 : -0.0;
    V3 r = {b, 1.0, 1.0}, o = {0, 0, 0};
    RayDiv d = make_raydiv(o, r, true);
    double q = div_by(a, d.nbx, d.yx, d.zx);
    double ref = a / b;
    unsigned long long qb = (unsigned long long)__double_as_longlong(q), rb = (unsigned long long)__double_as_longlong(ref);
    bool same = qb == rb || (q != q && ref != ref) || (q == 0.0 && ref == 0.0);   // NaN payload / sign of zero are free
    if (!d.ok || !same) atomicAdd(mismatches, 1ull);
}

// [rank][local tile][tile_h][tile_w][bpp] -> row-major frame
__global__ void __launch_bounds__(256) untile_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int W, int H,
                                                     int tile_w, int tile_h, int tiles_x, int n_ranks, uint64_t rank_stride,
                                                     int bpp) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (uint64_t)W * (uint64_t)H) return;
    int row = (int)(k / (uint64_t)W), col = (int)(k % (uint64_t)W);
    uint32_t tile = (uint32_t)(row / tile_h) * (uint32_t)tiles_x + (uint32_t)(col / tile_w);
    uint32_t rank = tile % (uint32_t)n_ranks, local_tile = tile / (uint32_t)n_ranks;
    uint64_t s = (uint64_t)rank * rank_stride +
                 (((uint64_t)local_tile * (uint32_t)tile_h + (uint32_t)(row % tile_h)) * (uint32_t)tile_w + (uint32_t)(col % tile_w)) * (uint32_t)bpp;
    for (int b = 0; b < bpp; b++) dst[k * (uint32_t)bpp + b] = src[s + b];
}

}  // namespace

hipError_t launch_render(const DevScene &sc, const DevRender &rp, bool prune, bool stats, uint32_t block_threads,
                         hipStream_t st) {
    const uint64_t per_tile = (uint64_t)(rp.tile_w / 8) * (uint64_t)(rp.tile_h / 8);
    const uint64_t waves = per_tile * (uint64_t)rp.owned_tiles;
    if (waves == 0) return hipSuccess;
    const uint32_t wpb = block_threads / 64;
    const uint64_t blocks = (waves + wpb - 1) / wpb;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    dim3 grid((uint32_t)blocks), block(block_threads);
    // one u32 per level per lane; a walk never holds more entries than the tree is deep
    const size_t lds = render_lds_bytes(sc.bvh_depth, rp.depth, block_threads);
    if (prune && stats) hipLaunchKernelGGL((render_kernel<true, true>), grid, block, lds, st, sc, rp);
    else if (prune) hipLaunchKernelGGL((render_kernel<true, false>), grid, block, lds, st, sc, rp);
    else if (stats) hipLaunchKernelGGL((render_kernel<false, true>), grid, block, lds, st, sc, rp);
    else hipLaunchKernelGGL((render_kernel<false, false>), grid, block, lds, st, sc, rp);
    return hipGetLastError();
}

hipError_t launch_divtest(uint64_t seed, uint64_t n, unsigned long long *d_mismatches, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(divtest_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, seed, n, d_mismatches);
    return hipGetLastError();
}

hipError_t launch_untile(const uint8_t *src, uint8_t *dst, int W, int H, int tile_w, int tile_h, int tiles_x, int n_ranks,
                         uint64_t rank_stride, int bpp, hipStream_t st) {
    const uint64_t n = (uint64_t)W * (uint64_t)H;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(untile_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, src, dst, W, H, tile_w, tile_h,
                       tiles_x, n_ranks, rank_stride, bpp);
    return hipGetLastError();
}

}  // namespace rtx
