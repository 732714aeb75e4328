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

#include <mutex>

#include "device_layout.h"
#include "rtx_kernels.h"

namespace rtx {

namespace {

#define RTX_INF (__builtin_inf())
// second __launch_bounds__ argument of the render kernel: waves per SIMD the register
// allocator must leave room for
#ifndef RTX_WAVES_PER_EU
#define RTX_WAVES_PER_EU 4
#endif
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

// the value lane `l` (wave-uniform) holds
__device__ __forceinline__ double lane_value(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// A node pair or a triangle is fetched with 16-byte loads issued back to back and kept in registers.  issued() keeps
// the compiler from sinking some of those loads below the first arithmetic (it would, to save registers, and a step would
// then pay two or three memory round trips instead of one): nothing is scheduled across it.  It does not wait: the
// arithmetic on the first pieces starts while the last ones are still on their way (each wait sits at a piece's first use).
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void issued() { __builtin_amdgcn_sched_barrier(0); }
#ifdef RTX_EXPERIMENT_KINDTIME
// MEASUREMENT ONLY: the shader clock, read where the statement stands (the compiler moves neither memory accesses nor other reads of the clock across it)
__device__ __forceinline__ unsigned long long cycles_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#endif

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
    double yx, yy, yz;      // refined reciprocals (+-inf for a zero component)
    uint32_t flags;         // RD_ZX | RD_ZY | RD_ZZ: component == 0;  RD_OK: fast path valid for this ray
};
enum : uint32_t { RD_ZX = 1u, RD_ZY = 2u, RD_ZZ = 4u, RD_OK = 8u };
// RD_OK and neither r.y nor r.z zero: the pair test may use slab_pair_nz
__device__ __forceinline__ bool rd_nz(uint32_t flags) { return (flags & (RD_OK | RD_ZY | RD_ZZ)) == RD_OK; }

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
    const bool zx = r.x == 0.0, zy = r.y == 0.0, zz = r.z == 0.0;
    d.yx = zx ? __builtin_copysign(RTX_INF, r.x) : refined_rcp(r.x);
    d.yy = zy ? __builtin_copysign(RTX_INF, r.y) : refined_rcp(r.y);
    d.yz = zz ? __builtin_copysign(RTX_INF, r.z) : refined_rcp(r.z);
    const bool ok = scene_ok && exp_safe(o.x) && exp_safe(o.y) && exp_safe(o.z) && exp_safe(r.x) && exp_safe(r.y) && exp_safe(r.z);
    d.flags = (zx ? RD_ZX : 0u) | (zy ? RD_ZY : 0u) | (zz ? RD_ZZ : 0u) | (ok ? RD_OK : 0u);
    return d;
}

// a / b given y = refined_rcp(b), zero = (b == 0)
__device__ __forceinline__ double div_by(double a, double b, double y, bool zero) {
    double q0 = a * y;
    double rem = __builtin_fma(-b, q0, a);
    double q1 = __builtin_fma(rem, y, q0);
    return zero ? q0 : q1;
}

// BVH::intersectRayAABB with tLim = +inf (bvh.hpp:213-234; both call sites pass a
// local closestT that is still INFINITY, bvh.hpp:170,174).  *tnear is the entry
// distance the test computed.
// *beyond: some axis' entry distance exceeds that axis' limit `lim` (PRUNE only).
__device__ __forceinline__ bool slab(const double *lo, const double *hi, V3 o, V3 r, V3 lim, double *tnear, bool *beyond) {
    double tx1 = (lo[0] - o.x) / r.x, tx2 = (hi[0] - o.x) / r.x;
    double nx = smin(tx1, tx2);
    double tmin = nx, tmax = smax(tx1, tx2);
    double ty1 = (lo[1] - o.y) / r.y, ty2 = (hi[1] - o.y) / r.y;
    double ny = smin(ty1, ty2);
    tmin = smax(tmin, ny);
    tmax = smin(tmax, smax(ty1, ty2));
    double tz1 = (lo[2] - o.z) / r.z, tz2 = (hi[2] - o.z) / r.z;
    double nz = smin(tz1, tz2);
    tmin = smax(tmin, nz);
    tmax = smin(tmax, smax(tz1, tz2));
    *tnear = tmin;
    *beyond = nx > lim.x || ny > lim.y || nz > lim.z;
    return tmin < RTX_INF && tmax > 0 && tmax >= tmin;
}

// The same test with the divisions taken through RayDiv.
__device__ __forceinline__ bool slab_fast(const double *lo, const double *hi, V3 o, V3 r, const RayDiv &d, V3 lim, double *tnear,
                                          bool *beyond) {
    const bool zx = d.flags & RD_ZX, zy = d.flags & RD_ZY, zz = d.flags & RD_ZZ;
    double tx1 = div_by(lo[0] - o.x, r.x, d.yx, zx), tx2 = div_by(hi[0] - o.x, r.x, d.yx, zx);
    double nx = smin(tx1, tx2);
    double tmin = nx, tmax = smax(tx1, tx2);
    double ty1 = div_by(lo[1] - o.y, r.y, d.yy, zy), ty2 = div_by(hi[1] - o.y, r.y, d.yy, zy);
    double ny = smin(ty1, ty2);
    tmin = smax(tmin, ny);
    tmax = smin(tmax, smax(ty1, ty2));
    double tz1 = div_by(lo[2] - o.z, r.z, d.yz, zz), tz2 = div_by(hi[2] - o.z, r.z, d.yz, zz);
    double nz = smin(tz1, tz2);
    tmin = smax(tmin, nz);
    tmax = smin(tmax, smax(tz1, tz2));
    *tnear = tmin;
    *beyond = nx > lim.x || ny > lim.y || nz > lim.z;
    return tmin < RTX_INF && tmax > 0 && tmax >= tmin;
}

// v_min_f64 / v_max_f64 as single instructions.  Through __builtin_fmin / __builtin_fmax hipcc
// first re-quiets every operand it cannot prove canonical (v_max_f64 x, x, x), eight extra
// instructions per node pair; the operands here are never NaN-signalling (sums, products, fmas
// or other min/max results).
__device__ __forceinline__ double vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The pair test for the common ray: exponents tame (RD_OK) and r.y, r.z non-zero.
// Every y/z quotient is then finite and not NaN, so std::min/std::max and v_min_f64 /
// v_max_f64 agree in value and the compare-select chains of slab_fast collapse.
// A zero r.x stays exact by case analysis of bvh.hpp:218-233 with tx = a * (+-inf):
//   lo.x < o.x < hi.x      -> {tx1,tx2} = {-inf,+inf}: the x slab constrains nothing;
//   o.x outside            -> tmin = tmax = +-inf: rejected (tmin < inf or tmax > 0 fails);
//   o.x on a face (a = 0)  -> NaN first operand poisons tmin, or tmin = tmax = +-inf: rejected.
// So for r.x == 0 the box passes iff lo.x < o.x < hi.x and x is left out of tmin/tmax
// (requires lo <= hi, part of the scene's fast_div_ok).  Slot 0 (NaN bounds) is excluded
// by the caller.
__device__ __forceinline__ void slab_one_nz(const double *lo, const double *hi, V3 o, V3 r, const RayDiv &d, V3 lim, bool zx,
                                            bool *hit, bool *beyond, double *tnear) {
#ifdef RTX_EXPERIMENT_APPROX_SLAB
    // MEASUREMENT ONLY (wrong in the last bit, no fallback): what the frame would cost if every quotient were one
    // multiplication and the decision compared entries with exits of the OTHER axes only (right on flat boxes)
    {
        const double bx1 = (lo[0] - o.x) * d.yx, bx2 = (hi[0] - o.x) * d.yx;
        const double by1 = (lo[1] - o.y) * d.yy, by2 = (hi[1] - o.y) * d.yy;
        const double bz1 = (lo[2] - o.z) * d.yz, bz2 = (hi[2] - o.z) * d.yz;
        double nx = vmin(bx1, bx2), fx = vmax(bx1, bx2);
        bool x_ok = true;
        if (zx) { nx = -RTX_INF; fx = RTX_INF; x_ok = (lo[0] - o.x) < 0.0 && (hi[0] - o.x) > 0.0; }
        const double ny = vmin(by1, by2), fy = vmax(by1, by2), nz = vmin(bz1, bz2), fz = vmax(bz1, bz2);
        const double mxy = vmin(fx, fy);
        *tnear = vmax(vmax(nx, ny), nz);
        *beyond = nx > lim.x || ny > lim.y || nz > lim.z;
        *hit = x_ok && vmin(mxy, fz) > 0 && nx <= vmin(fy, fz) && ny <= vmin(fx, fz) && nz <= mxy;
        return;
    }
#endif
    const double ax1 = lo[0] - o.x, ax2 = hi[0] - o.x;
    const double qx1 = __builtin_fma(__builtin_fma(-r.x, ax1 * d.yx, ax1), d.yx, ax1 * d.yx);
    const double qx2 = __builtin_fma(__builtin_fma(-r.x, ax2 * d.yx, ax2), d.yx, ax2 * d.yx);
    const double ay1 = lo[1] - o.y, ay2 = hi[1] - o.y;
    const double qy1 = __builtin_fma(__builtin_fma(-r.y, ay1 * d.yy, ay1), d.yy, ay1 * d.yy);
    const double qy2 = __builtin_fma(__builtin_fma(-r.y, ay2 * d.yy, ay2), d.yy, ay2 * d.yy);
    const double az1 = lo[2] - o.z, az2 = hi[2] - o.z;
    const double qz1 = __builtin_fma(__builtin_fma(-r.z, az1 * d.yz, az1), d.yz, az1 * d.yz);
    const double qz2 = __builtin_fma(__builtin_fma(-r.z, az2 * d.yz, az2), d.yz, az2 * d.yz);
    double nx = vmin(qx1, qx2), fx = vmax(qx1, qx2);
    bool x_ok = true;
    if (zx) {
        nx = -RTX_INF;
        fx = RTX_INF;
        x_ok = ax1 < 0.0 && ax2 > 0.0;
    }
    const double ny = vmin(qy1, qy2), fy = vmax(qy1, qy2);
    const double nz = vmin(qz1, qz2), fz = vmax(qz1, qz2);
    const double tmin = vmax(vmax(nx, ny), nz);
    const double tmax = vmin(vmin(fx, fy), fz);
    *tnear = tmin;
    *beyond = nx > lim.x || ny > lim.y || nz > lim.z;
    *hit = x_ok && tmax > 0 && tmax >= tmin;
}

// Triangle::intersectRay (include/triangle.hpp:32-55); d = -(n.v0) comes precomputed.
__device__ __forceinline__ bool tri_hit(V3 v0, V3 v1, V3 v2, V3 n, double d, V3 o, V3 r, double *t_out) {
    double nd = dot(n, r);
    if (__builtin_fabs(nd) < RTX_EPS) return false;
    double t = -(dot(n, o) + d) / nd;
    V3 p = add(o, smul(t, r));
    *t_out = t;
    if (!(dot(n, cross(sub(v1, v0), sub(p, v0))) > RTX_EPS)) return false;
    if (!(dot(n, cross(sub(v2, v1), sub(p, v1))) > RTX_EPS)) return false;
    return dot(n, cross(sub(v0, v2), sub(p, v2))) > RTX_EPS;
}

struct Counters {
    unsigned long long v[9];
};
enum { C_PRIMARY, C_REFLECTION, C_SHADOW, C_OCCLUDED, C_CLOSEST, C_NODES, C_BOXHITS, C_LEAVES, C_TRIS };

// pow(x, specularity) of the specular term (scene.hpp:176): the exponent is an int, the base a cosine.
// glibc's pow returns the correctly rounded power in all but a few cases per million; the device
// libm's is within one unit of the last place, a different last bit in a fifth of the cases.  For
// exponents 2 and up and an ordinary base the power is therefore computed here: square-and-multiply
// on the base's significand in double-double arithmetic, binary exponents kept aside (error below
// 2^-98 of the result after at most 60 products), rounded once -- the correctly rounded power unless
// the exact value lies within 2^-98 of a rounding boundary.  Everything else -- other exponents, bases
// that are not normal positive numbers, results outside the normal range -- goes to the device pow.
// Kept out of line: inlined it would raise the whole kernel's register count for a term no shipped
// scene uses (specularity is 1 everywhere, scene.hpp:241).
struct DD { double hi, lo; };
__device__ __forceinline__ DD dd_mul(DD a, DD b) {
    const double p = a.hi * b.hi;
    double e = __builtin_fma(a.hi, b.hi, -p);                      // exact error of p
    e = __builtin_fma(a.hi, b.lo, __builtin_fma(a.lo, b.hi, e));   // a.lo * b.lo is below 2^-106 of the product
    const double s = p + e;
    return {s, e - (s - p)};                                        // |e| << |p|: the sum's error, exactly
}
// factors in [1, 2) give a product in [1, 4): back to [1, 2), exactly
__device__ __forceinline__ void dd_halve_if_2(DD &v, long long &ex) {
    if (v.hi >= 2.0) { v.hi *= 0.5; v.lo *= 0.5; ex++; }
}
__device__ __attribute__((noinline)) double general_pow(double x, double e) {
    const int n = (int)e;
    const uint32_t hi32 = (uint32_t)__double2hiint(x);
    const uint32_t bexp = (hi32 >> 20) & 0x7ffu;                   // biased exponent (the sign bit must be 0)
    const long long ex = (long long)bexp - 1023ll;
    const bool ordinary = (double)n == e && n >= 2 && n <= 1 << 30 && x > 0.0 && bexp != 0u && bexp != 0x7ffu;
    // x^n lies in [2^(ex n), 2^((ex + 1) n)): give up where that leaves the normal range
    if (!ordinary || ex * n < -1021ll || (ex + 1) * n > 1023ll) return pow(x, e);
    DD r = {1.0, 0.0};
    DD b = {__hiloint2double((int)((hi32 & 0x000fffffu) | 0x3ff00000u), __double2loint(x)), 0.0};   // x / 2^ex, in [1, 2)
    long long er = 0, eb = ex;
    for (int k = n;;) {
        if (k & 1) { r = dd_mul(r, b); er += eb; dd_halve_if_2(r, er); }
        k >>= 1;
        if (k == 0) break;
        b = dd_mul(b, b); eb *= 2; dd_halve_if_2(b, eb);
    }
    if (er < -1022ll) return pow(x, e);                             // a subnormal result would be rounded twice
    return ldexp(r.hi, (int)er);
}

__device__ __forceinline__ V3 reflect_ray(V3 ray, V3 n) {              // scene.hpp:185-187
    return sub(muls(smul(2.0, n), dot(n, ray)), ray);
}

// Scene::computeLighting (scene.hpp:147-183) asks its shadow query first and only then looks at the surface: the diffuse
// term needs N.L > 0 (:165-170), the specular term specularity != -1 and R.V > 0 (:172-180).  Where both conditions fail
// the light adds 0 whatever the query answers, so the query need not be asked (the pruned walk only; the reference walk
// traces what the reference traces).  The same expressions as the terms themselves: the same values, the same decisions.
__device__ __forceinline__ bool light_adds_nothing(V3 n, V3 v, int spec, V3 light) {
    if (dot(n, light) > 0.0) return false;
    if (spec == -1) return true;
    return !(dot(reflect_ray(light, n), v) > 0);
}

// Distance pruning (PRUNE) skips a box when, on SOME axis, the ray enters the box's slab
// later than limit + slack, where limit is the best accepted t so far (or tmax).
// Why a slack: a triangle's t comes from the plane equation (triangle.hpp:40), a slab
// entry from a division (bvh.hpp:218-229) -- the same geometry, rounded differently.  The
// accepted hit point o + t*r passed the inside tests and lies within rounding distance
// (a few ulps of the scene's extent) of its triangle, hence of every box that holds the
// triangle; on axis i that bounds t from below by entry_i - (point slack) / |r_i|.
// `slack_point` is 2^-20 of the reach (scene extent + |origin|), 2^32 rounding units --
// the factor the reference's own EPSILON uses -- and 1/|r_i| is taken from the ray's
// refined reciprocal (times 2 for its rounding).  An axis the ray barely moves along gets
// a huge slack and simply never prunes; the other axes still do.
__device__ __forceinline__ double slack_point(V3 o, double scene_scale) {
    const double reach = scene_scale + __builtin_fabs(o.x) + __builtin_fabs(o.y) + __builtin_fabs(o.z);
    return reach * 1.9073486328125e-06;   // 2^-19
}
__device__ __forceinline__ V3 prune_limits(double limit, double slack, const RayDiv &d) {
    // limit + slack * |1/r_i|; a zero component has y = inf -> never prunes on that axis
    return {__builtin_fma(slack, __builtin_fabs(d.yx), limit), __builtin_fma(slack, __builtin_fabs(d.yy), limit),
            __builtin_fma(slack, __builtin_fabs(d.yz), limit)};
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// static_cast<int>(double) of color.hpp:47-53 the way the reference's x86-64 build does it
// (cvttsd2si): a value that does not fit an int, or a NaN, becomes INT_MIN, where this GPU's
// conversion would saturate (NaN -> 0).  Only very bright lights get there.
__device__ __forceinline__ int to_int(double v) {
    return (v < 2147483648.0 && v > -2147483649.0) ? (int)v : (int)0x80000000;
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }   // color.hpp:23-29

// ---------------------------------------------------------------------------------
// The render kernel.
//
// One lane = one pixel, one wave = one 8x8-pixel block of a screen tile.  The per-pixel
// work is cut into three kinds of step, and in every round the wave runs ONE kind, the
// one most lanes are waiting for (ballot + popcount, weighted), instead of running all
// kinds with the lanes that need them masked in turn:
//   INNER  one BVH inner node: both children (one 128-byte line) slab-tested, far child
//          pushed on the lane's LDS stack                      (bvh.hpp:197-208,213-234)
//   LEAF   one triangle of a leaf                               (bvh.hpp:178-195, triangle.hpp:32-55)
//   SHADE  a finished query: a closest-hit result starts the shadow query of the first
//          light, a shadow result adds that light's term and starts the next light's
//          query, the last one finishes the surface colour and either reflects or ends
//          the pixel (scene.hpp:89-183).
// BVH::intersectRay's recursion becomes this walk; its result is kept exactly:
// smallest t with hit && tmin < t < tmax; equal t -> the triangle of the later leaf
// (larger leaf `first`), inside one leaf the earlier triangle (bvh.hpp:186,204-208).
//   PRUNE = false: a child is entered iff its slab test passes, shadow queries walk
//           everything: exactly the reference's node visits.
//   PRUNE = true : near child first; a child entered later than the best hit so far
//           (+ slack, see prune_limits) is skipped; a shadow query stops at its first
//           accepted triangle.  Same results, less work.
//
// Registers are what limits the waves per SIMD here (everything is FP64), so a lane
// keeps only the running query in registers.  What a surface hit needs again after its
// shadow queries (-ray, triangle, lighting sum) and the (localColor, reflectivity) pairs
// that wait for their reflected colour (scene.hpp:122) live in a per-lane slice of a
// global scratch buffer, written and read once per query in the SHADE step.
enum : int { ST_INNER = 0, ST_LEAF = 1, ST_SHADE = 2, ST_FETCH = 3, ST_DONE = 4 };

struct Lane {
    V3 o, r;             // the running query
    RayDiv rd;
    double slack;
    V3 lim;              // PRUNE: per-axis entry limits (prune_limits)
    double best_t;
    int32_t best_pos;
    uint32_t first, count, sp;
    int state;
    // kind (bit 0: 1 = shadow query), reflections so far d (bits 1..3), sample s (bits 4..15),
    // light index li (bits 16..31)
    uint32_t path;
    uint32_t pix;        // the pixel: column | row << 16
    uint32_t out;        // its index in the output buffers (row-major or packed-tile layout)
    uint32_t blk;        // staged launches: the pixel's 8x8 block (index into DevRender::hitmask)
#ifdef RTX_EXPERIMENT_STEPHIST
    uint32_t steps;      // MEASUREMENT ONLY (tools/step_histogram.py): node-pair steps of the running query
#endif
};

__device__ __forceinline__ uint32_t path_kind(uint32_t p) { return p & 1u; }
__device__ __forceinline__ uint32_t path_d(uint32_t p) { return (p >> 1) & 7u; }
__device__ __forceinline__ uint32_t path_s(uint32_t p) { return (p >> 4) & 0xFFFu; }
__device__ __forceinline__ uint32_t path_li(uint32_t p) { return p >> 16; }
__device__ __forceinline__ uint32_t make_path(uint32_t kind, uint32_t d, uint32_t s, uint32_t li) {
    return kind | (d << 1) | (s << 4) | (li << 16);
}
// tMin / tMax of the running query (src/main.cpp:83, scene.hpp:120,132,139,155)
__device__ __forceinline__ double query_tmin(uint32_t p) { return (p & 0xFu) == 0u ? 1.0 : 0.001; }
__device__ __forceinline__ double query_tmax(uint32_t p, uint32_t n_point) {
    return (path_kind(p) == 1u && path_li(p) < n_point) ? 1.0 : RTX_INF;
}

// A lighting job (see render_kernel): the lane computes the lighting of another lane's surface hit.  Only with one
// sample per pixel, so the sample bits of `path` are free: bit 4 marks the job, bits 5..10 name the lane that owns
// the pixel.
constexpr uint32_t kJob = 1u << 4;
__device__ __forceinline__ uint32_t job_owner(uint32_t p) { return (p >> 5) & 63u; }

// per-lane slice of the scratch buffer: field-major arrays of `n` lanes each.  `depth` is the launch's recursion
// depth: a pixel has at most depth + 1 surface hits ("levels").  A level's record is its localColor (color x lighting,
// truncated: scene.hpp:113) and reflectivity; a level whose lighting was handed to a lighting job holds the triangle's
// colour instead, its bit is set in `waiting`, and the job delivers the total lighting.
struct Scratch {
    double *f64;         // [0..2] -ray  [3] lighting sum  [4 + level] reflectivity  [4 + depth + level] total lighting (handed-over levels)
    int32_t *i32;        // [0] triangle position  [1 + 3*level + c] colour channel c  [4 + 3*depth ..] see below
    uint64_t n, id;
    uint32_t depth;
    __device__ __forceinline__ double &f(uint32_t field) const { return f64[(uint64_t)field * n + id]; }
    __device__ __forceinline__ int32_t &i(uint32_t field) const { return i32[(uint64_t)field * n + id]; }
    __device__ __forceinline__ double &refl(uint32_t level) const { return f(4u + level); }
    __device__ __forceinline__ double &total(uint32_t level) const { return f(4u + depth + level); }
    __device__ __forceinline__ int32_t &colour(uint32_t level, uint32_t c) const { return i(1u + 3u * level + c); }
    // what is left of a pixel whose chain of closest-hit queries has ended while lighting jobs are still out
    __device__ __forceinline__ int32_t &chain_end() const { return i(4u + 3u * depth); }     // levels below the innermost colour | (innermost = a waiting level) << 8
    __device__ __forceinline__ int32_t &out_index() const { return i(5u + 3u * depth); }
    __device__ __forceinline__ int32_t &waiting() const { return i(6u + 3u * depth); }       // bit per level: total lighting comes from a job
    __device__ __forceinline__ int32_t &innermost(uint32_t c) const { return i(7u + 3u * depth + c); }
};

// What the traversal steps and the scheduler read in every round, fetched from the argument
// blocks once per wave.
struct Hot {
    const DevNode *nodes;
    const DevTri *tris;
    const d2 *top;       // LDS copy of the top of the tree (DevScene::top), 8 pieces of 16 bytes per pair
    const double *lights;   // LDS copy of the first lights (see kLdsLights)
    uint32_t root_ref;   // where a walk starts: kTopFlag | 0 with a table, else slot 0
    uint32_t n_point;
    int32_t w_inner, w_leaf, w_shade, tail_lanes, also_lanes, also_leaf, also_shade;
};

__device__ __forceinline__ void start_query(Lane &L, const DevScene &sc, bool prune, uint32_t root_ref) {
    L.rd = make_raydiv(L.o, L.r, sc.fast_div_ok != 0);
    L.slack = prune ? slack_point(L.o, sc.scene_scale) : 0.0;
    // no accepted t can reach tmax, so tmax is the first limit
    L.lim = prune ? prune_limits(query_tmax(L.path, sc.n_point), L.slack, L.rd) : V3{RTX_INF, RTX_INF, RTX_INF};
    L.best_t = RTX_INF;
    L.best_pos = -1;
    // slot 0 holds a box no ray enters and slot 1 the root: the walk starts as the pair
    // test of (0, 1), which is the root's own slab test (bvh.hpp:174)
    L.first = root_ref;
    L.count = 0;
    L.sp = 0;
    L.state = ST_INNER;
#ifdef RTX_EXPERIMENT_STEPHIST
    L.steps = 0;
#endif
}

// Enters the node a reference names (node_entry() in device_layout.h): a child of the pair just tested, or what came
// off the stack.
__device__ __forceinline__ void enter_node(Lane &L, uint32_t e, const DevNode *nodes) {
    const bool leaf = (e & 0x80000000u) != 0;
    L.first = leaf ? (e & 0x0FFFFFFFu) : e;
    L.count = leaf ? ((e >> 28) & 7u) + 1u : 0u;
    L.state = leaf ? ST_LEAF : ST_INNER;
    if ((e >> 28) == 0xFu) {               // a leaf of more than 7 triangles: its header says where
        const uint32_t slot = e & 0x0FFFFFFFu;
        L.first = nodes[slot].first;
        L.count = nodes[slot].count;
    }
}

__device__ __forceinline__ void pop_node(Lane &L, const DevNode *nodes, const uint32_t *stack, uint32_t stride) {   // stride: log2 of the lanes per stack level
    if (L.sp == 0) {
        L.state = ST_SHADE;
        return;
    }
    L.sp--;
    enter_node(L, stack[L.sp << stride], nodes);
}

template <bool PRUNE, bool STATS>
__device__ __forceinline__ void inner_step(Lane &L, const Hot &hot, uint32_t *stack, uint32_t stride, Counters &c) {
    const DevNode *nodes = hot.nodes;
    // the sibling pair is one 128-byte line: seven 16-byte loads of it, one round trip -- to LDS for a pair
    // of the top table (every walk starts there, and the vector memory path is what the kernel waits
    // for), else to memory
    d2 q0, q1, q2, q3, q4, q5, q6;
    const bool in_top = (L.first & kTopFlag) != 0;
    if (in_top) {
        const d2 *line = hot.top + (L.first & 0xFFFFu) * 8u;
        q0 = line[0]; q1 = line[1]; q2 = line[2]; q3 = line[3]; q4 = line[4]; q5 = line[5]; q6 = line[6];
    } else {
        // 112 of the pair's 128 bytes: the first record whole (its padding holds the references to both nodes) and the
        // sibling's box -- seven loads; every byte a wave asks its L1 for counts here
        const d2 *line = reinterpret_cast<const d2 *>(nodes + L.first);
        q0 = line[0]; q1 = line[1]; q2 = line[2];
        issued();
        q4 = line[4]; q5 = line[5]; q6 = line[6];
        issued();
        q3 = line[3];
        issued();
    }
    const bool root_pair = (L.first & ~kTopFlag) == 0;                   // slots (0, 1)
    const double a_lo[3] = {q0.x, q0.y, q1.x}, a_hi[3] = {q1.y, q2.x, q2.y};
    const double b_lo[3] = {q4.x, q4.y, q5.x}, b_hi[3] = {q5.y, q6.x, q6.y};
    double ta, tb;
    bool ha, hb, beyond_a, beyond_b;
    if (rd_nz(L.rd.flags)) {
        const bool zx = L.rd.flags & RD_ZX;
        slab_one_nz(a_lo, a_hi, L.o, L.r, L.rd, L.lim, zx, &ha, &beyond_a, &ta);
        slab_one_nz(b_lo, b_hi, L.o, L.r, L.rd, L.lim, zx, &hb, &beyond_b, &tb);
        ha = ha && !root_pair;              // slot 0 is the box no ray enters
    } else if (L.rd.flags & RD_OK) {
        ha = slab_fast(a_lo, a_hi, L.o, L.r, L.rd, L.lim, &ta, &beyond_a);
        hb = slab_fast(b_lo, b_hi, L.o, L.r, L.rd, L.lim, &tb, &beyond_b);
    } else {   // ray components at the ends of the exponent range: plain divisions
        ha = slab(a_lo, a_hi, L.o, L.r, L.lim, &ta, &beyond_a);
        hb = slab(b_lo, b_hi, L.o, L.r, L.lim, &tb, &beyond_b);
    }
#ifdef RTX_EXPERIMENT_STEPHIST
    L.steps++;
#endif
    if (STATS) {
        c.v[C_NODES] += root_pair ? 1 : 2;    // slot 0 is not a node of the tree
        c.v[C_BOXHITS] += (ha ? 1 : 0) + (hb ? 1 : 0);
    }
    if (PRUNE) {
        ha = ha && !beyond_a;
        hb = hb && !beyond_b;
    }
    // the references to both nodes sit ready-made in the fourth 16-byte piece of the first record
    const uint32_t ea = (uint32_t)__double2loint(q3.y), eb = (uint32_t)__double2hiint(q3.y);
    // next node: the only child hit, or of two the nearer one (PRUNE) / the left one, while the other one of two goes on
    // the stack; with no child hit, what comes off the stack
    const bool take_b = hb && (!ha || (PRUNE && tb < ta));
    if (ha && hb) {
        stack[L.sp << stride] = take_b ? ea : eb;
        L.sp++;
    }
    uint32_t e = take_b ? eb : ea;
    if (!(ha || hb)) {
        if (L.sp == 0) {
            L.state = ST_SHADE;
            return;
        }
        L.sp--;
        e = stack[L.sp << stride];
    }
    enter_node(L, e, nodes);
}

template <bool PRUNE, bool STATS>
__device__ __forceinline__ void leaf_step(Lane &L, const Hot &hot, const uint32_t *stack, uint32_t stride, Counters &c) {
    // 112 of the record's 128 bytes: seven 16-byte loads, one round trip
    const d2 *rec = reinterpret_cast<const d2 *>(hot.tris + L.first);
    // issued in the order of use: the plane, then the vertices
    const d2 q0 = rec[0], q1 = rec[1];
    issued();
    const d2 q2 = rec[2], q3 = rec[3], q4 = rec[4];
    issued();
    const d2 q5 = rec[5], q6 = rec[6];
    issued();
    const V3 n = {q0.x, q0.y, q1.x}, v0 = {q2.x, q2.y, q3.x}, v1 = {q3.y, q4.x, q4.y}, v2 = {q5.x, q5.y, q6.x};
    const double plane_d = q1.y;
    const uint32_t leaf_first = (uint32_t)__double2hiint(q6.y);          // {orig_id, leaf_first}
    if (STATS) {
        c.v[C_TRIS]++;
        if (L.first == leaf_first) c.v[C_LEAVES]++;
    }
    double t;
    if (tri_hit(v0, v1, v2, n, plane_d, L.o, L.r, &t) && t > query_tmin(L.path) && t < query_tmax(L.path, hot.n_point)) {
        // equal t: the later leaf wins; inside one leaf the earlier triangle stays (strict <,
        // bvh.hpp:186).  The best hit's leaf is read back only on an exact tie.
        bool better = t < L.best_t;
        if (t == L.best_t) better = leaf_first > hot.tris[L.best_pos].leaf_first;
        if (better) {
            L.best_t = t;
            L.best_pos = (int32_t)L.first;
            if (PRUNE) {
                L.lim = prune_limits(t, L.slack, L.rd);
                if (path_kind(L.path) == 1u) {   // a shadow query only asks whether a hit exists
                    L.state = ST_SHADE;
                    return;
                }
            }
        }
    }
    L.first++;
    L.count--;
    if (L.count == 0) pop_node(L, hot.nodes, stack, stride);
}

// src/main.cpp:78-83: camera ray of pixel (col,row), sample s
__device__ __forceinline__ void primary_ray(Lane &L, const DevScene &sc, const DevRender &rp, int col, int row, uint32_t s, uint32_t frame) {
    // putPixel inverted (src/main.cpp:90-94): row = -y + H/2 - 1, col = x + W/2
    const int x = col - rp.W / 2, y = rp.H / 2 - 1 - row;
    double fx = (double)x, fy = (double)y;
    if (s > 0) {   // samples > 0 add a jitter in [0,1)^2 (this build's extension)
        const uint64_t k = (uint64_t)row * (uint64_t)rp.W + (uint64_t)col;
        uint64_t z = splitmix64(splitmix64(rp.seed) ^ k);
        z = splitmix64(z ^ ((uint64_t)s << 32));
        fx = fx + (double)(z >> 32) * (1.0 / 4294967296.0);
        fy = fy + (double)(z & 0xFFFFFFFFull) * (1.0 / 4294967296.0);
    }
    // Scene::toViewport (scene.hpp:81-87), then rotateX, rotateY, rotateZ (vector3.hpp:54-82)
    // the scene's camera, or the frame's own in a launch over several frames
    const bool own = rp.has_cams != 0;
    const double *cam = rp.cams[own ? frame : 0u];
    const double cx = own ? cam[3] : sc.cx, sx = own ? cam[4] : sc.sx, cy = own ? cam[5] : sc.cy, sy = own ? cam[6] : sc.sy,
                 cz = own ? cam[7] : sc.cz, sz = own ? cam[8] : sc.sz;
    V3 vp = {fx * rp.px, fy * rp.py, sc.viewport[2]};
    V3 rx = {vp.x, cx * vp.y - sx * vp.z, sx * vp.y + cx * vp.z};
    V3 ry = {cy * rx.x + sy * rx.z, rx.y, (-sy) * rx.x + cy * rx.z};
    L.r = {cz * ry.x - sz * ry.y, sz * ry.x + cz * ry.y, ry.z};
    L.o = {own ? cam[0] : sc.cam[0], own ? cam[1] : sc.cam[1], own ? cam[2] : sc.cam[2]};
    L.path = make_path(0, 0, s, 0);
}

// What a pixel accumulates over its samples (only used when spp > 1).
struct PixelSum {
    int r, g, b;
};

// A block keeps the scene's first kLdsLights lights in LDS ({intensity, v[3]} each, as in DevLight): the shading step
// then reads a light without a trip to memory.  Scenes with more lights read all of them from memory.
constexpr uint32_t kLdsLights = 8;
__device__ __forceinline__ double light_intensity(const DevScene &sc, const double *lds_lights, uint32_t n_lights, uint32_t li) {
    return n_lights <= kLdsLights ? lds_lights[4u * li] : sc.lights[li].intensity;
}
__device__ __forceinline__ V3 light_vector(const DevScene &sc, const double *lds_lights, uint32_t n_lights, uint32_t li) {
    if (n_lights <= kLdsLights) return {lds_lights[4u * li + 1u], lds_lights[4u * li + 2u], lds_lights[4u * li + 3u]};
    return ld3(sc.lights[li].v);
}

// A finished query (scene.hpp:89-183).  Leaves the lane with a new query (ST_INNER, its origin, direction and `path` set,
// `new_query` raised: the caller starts all of a round's new queries in one place, start_query() being a good hundred
// instructions that would otherwise stand in five branches of one round) or without a pixel (ST_FETCH).
//
// A pixel is a chain of closest-hit queries (camera ray, then up to `depth` reflections); every surface hit
// ("level") also needs its lighting: one shadow query per light, in the order of computeTotalLighting.  Levels that
// wait for their reflected colour keep their localColor and reflectivity in the lane's scratch, and when the chain
// has ended the levels are blended innermost first (scene.hpp:120-122) with Color's truncating arithmetic.
//
// TAIL = false is the launch's steady state: a lane does all of that for its pixel, one query after the other.
// TAIL = true is the code of a wave whose queues are empty and that has lanes with nothing left to do (see
// render_kernel).  There a surface hit's lighting may have been handed to such a lane as a lighting job
// (`handed_over`, decided by the caller just before this step) while the owner follows the reflection at once -- the
// shadow queries of a hit and its reflection query do not depend on each other, only the blend needs both -- and the
// pixel is finished by whoever finishes last: pend[lane] (LDS) counts the parties still at work on that lane's
// pixel, the owner's chain and its lighting jobs.  A job runs this same code: it starts from a copy of the owner's
// closest-hit result, walks the lights and delivers the total into the OWNER's scratch.
template <bool PRUNE, bool STATS, bool TAIL>
__device__ __forceinline__ void shade_step(Lane &L, PixelSum &acc, const DevScene &sc, const DevRender &rp, const Scratch &scr_in,
                                           Counters &c, bool handed_over, uint32_t *pend, uint32_t *surf,
                                           const double *lds_lights, uint32_t lane_in, bool &new_query,
                                           unsigned long long *section_cycles = nullptr) {
    // The lane's number as a value the compiler cannot see through: every address derived from it (the lane's scratch
    // fields, its LDS slots) is then worked out here, inside the step.  Left visible, they are loop invariants of the
    // kernel's main loop, get computed once, do not fit the registers and are spilled -- and in this step each reload of
    // one waits for every load issued before it (s_waitcnt vmcnt(0)), which turns one trip to memory into four.
    uint32_t lane = lane_in;
    asm volatile("" : "+v"(lane));
    Scratch scr = scr_in;
    scr.id = scr_in.id - lane_in + lane;
#ifdef RTX_EXPERIMENT_KINDTIME
    // MEASUREMENT ONLY: section_cycles[k] adds up the cycles between RTX_SECTION(k - 1) and RTX_SECTION(k)
    unsigned long long section_t = cycles_now();
    struct SectionEnd {
        unsigned long long *acc, &t;
        __device__ ~SectionEnd() { acc[7] += cycles_now() - t; }
    } section_end{section_cycles, section_t};
#define RTX_SECTION(k) { const unsigned long long now_ = cycles_now(); section_cycles[k] += now_ - section_t; section_t = now_; }
#else
    (void)section_cycles;
#define RTX_SECTION(k)
#endif
    const uint32_t n_lights = sc.n_point + sc.n_dir;
    const uint32_t kind = path_kind(L.path), s = path_s(L.path);
    const bool job = TAIL && (L.path & kJob) != 0;
#ifdef RTX_EXPERIMENT_STEPHIST
    // MEASUREMENT ONLY: the caller's counter array continues with [16 + 12 kind + b] queries and [40 + 12 kind + b] their
    // node-pair steps, b = bucket (0: under 8 steps, 1: 8..15, 2: 16..31, ... 8: 1024 and up), kind 0 = closest hit, 1 = shadow
    if (STATS && rp.stats && !(job && kind == 0)) {
        uint32_t b = L.steps < 8u ? 0u : 29u - (uint32_t)__builtin_clz(L.steps);
        b = b > 8u ? 8u : b;
        atomicAdd(&rp.stats[16u + 12u * kind + b], 1ull);
        atomicAdd(&rp.stats[40u + 12u * kind + b], (unsigned long long)L.steps);
    }
#endif
    uint32_t d = path_d(L.path), li = path_li(L.path);
    bool path_done = false;    // the chain has ended: (cr, cg, cb) is its innermost colour
    int cr = sc.bg[0], cg = sc.bg[1], cb = sc.bg[2];
    bool lights = false;       // a shadow query was answered, or the first one is due
    bool elsewhere = false;    // TAIL: the lighting of this hit is a job's business

    // Everything the step can need of the surface it is about comes in ONE trip to memory, whatever the lane goes on to
    // do: the triangle's normal and material and, after a shadow query, what the closest-hit step left in the scratch.
    // (The branches below are taken by different lanes of the wave one after the other; with a trip to memory inside
    // each of them a shading step took seven times as long as a step of the walk.)  Which triangle the surface is
    // waits in LDS (*surf) while the shadow queries run; the lights come from LDS too.
    const int32_t pos = kind == 0 ? L.best_pos : (int32_t)surf[lane];   // -1: the closest-hit query missed
    V3 n = {0, 0, 1};
    V3 v = muls(L.r, -1.0);                 // -ray of the query that found the surface (vector3.hpp:108-110)
    int32_t colour0 = 0, colour1 = 0, colour2 = 0, spec = -1, orig_id = -1;
    double refl = -1.0, lighting = sc.ambient;                          // scene.hpp:128
    const bool wants_id = kind == 0 && d == 0 && s == 0 && rp.hit != nullptr;
    if (pos >= 0) {
        const d2 *tq = reinterpret_cast<const d2 *>(sc.tris + pos);
        const d2 n01 = tq[0], n2d = tq[1];
        const int4 m = *reinterpret_cast<const int4 *>(sc.mats + pos);  // colour, specularity
        refl = sc.mats[pos].reflectivity;
        if (wants_id) orig_id = (int32_t)sc.tris[pos].orig_id;
        if (kind == 1) {
            v = {scr.f(0), scr.f(1), scr.f(2)};
            lighting = scr.f(3);
        }
        issued();
        n = {n01.x, n01.y, n2d.x};
        colour0 = m.x; colour1 = m.y; colour2 = m.z; spec = m.w;
    }
    RTX_SECTION(0)          // the trip to memory
    if (kind == 0) {
        // Scene::traceRay after the closest-hit query (scene.hpp:97-111)
        if (STATS && !job) c.v[d == 0 ? C_PRIMARY : C_REFLECTION]++;
        if (d == 0 && s == 0 && (rp.hit || rp.t)) {                     // (a job carries its marks in the sample bits: s != 0)
            const uint64_t k = (uint64_t)(L.pix >> 16) * (uint64_t)rp.W + (uint64_t)(L.pix & 0xFFFFu);
            if (rp.hit) rp.hit[k] = orig_id;
            if (rp.t) rp.t[k] = L.best_t;
        }
        if (pos < 0) {
            path_done = true;                                           // backgroundColor, scene.hpp:101-103
        } else {
            if (STATS && !job) c.v[C_CLOSEST]++;
            L.o = add(L.o, smul(L.best_t, L.r));                        // point, scene.hpp:105
            scr.f(0) = v.x; scr.f(1) = v.y; scr.f(2) = v.z;
            surf[lane] = (uint32_t)pos;
            li = 0;
            elsewhere = TAIL && handed_over;
            lights = !elsewhere;
        }
    } else {
        // the rest of Scene::computeLighting (scene.hpp:159-182) with light = r, point = o
        if (STATS) { c.v[C_SHADOW]++; if (L.best_pos >= 0) c.v[C_OCCLUDED]++; }
        if (L.best_pos < 0) {
            const double intensity = light_intensity(sc, lds_lights, n_lights, li);
            double term = 0.0;
            const double nl = dot(n, L.r);
            if (nl > 0.0) term += intensity * (nl / (mag(n) * mag(L.r)));
            __builtin_amdgcn_sched_barrier(0);
            if (spec != -1) {
                const V3 R = reflect_ray(L.r, n);
                const double rv = dot(R, v);
                if (rv > 0) {
                    const double cosv = rv / (mag(R) * mag(v));
                    // pow(x, 1) is x exactly in every libm; see general_pow for the other exponents
                    term += intensity * ((spec == 1) ? cosv : general_pow(cosv, (double)spec));
                }
            }
            lighting += term;
        }
        li++;
        lights = true;
    }
    __builtin_amdgcn_sched_barrier(0);
    RTX_SECTION(1)          // closest-hit bookkeeping / lighting term
    Scratch px = scr;          // TAIL: the scratch of the pixel this step finishes, the lane's own or a job's owner's
    bool finish = false;       // TAIL: this lane is the last party of that pixel
    if (lights) {
        for (; li < n_lights; li++) {
            // computeTotalLighting (scene.hpp:130-142): point lights first (light = position -
            // point, tMax = 1), then directional ones (light = direction, tMax = inf)
            const V3 lv = light_vector(sc, lds_lights, n_lights, li);
            const V3 light = (li < sc.n_point) ? sub(lv, L.o) : lv;
            if (PRUNE && light_adds_nothing(n, v, spec, light)) continue;
            L.r = light;
            L.path = (L.path & 0xFFFEu) | 1u | (li << 16);              // a shadow query of light li; depth and sample (or job marks) stay
            scr.f(3) = lighting;
            L.state = ST_INNER; new_query = true;
            return;
        }
        RTX_SECTION(2)      // the lights loop with its shadow query starts
        if (job) {
            // the job is done: the level's total lighting goes to the pixel's owner, and if this was the last party at
            // work on the pixel, this lane finishes it
            const uint32_t owner = job_owner(L.path);
            px.id = scr.id - lane + owner;
            px.total(d) = lighting;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const uint32_t left = __hip_atomic_fetch_sub(pend + owner, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - 1u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            L.state = ST_FETCH;
            if (left != 0u) return;
            finish = true;
        } else {
            // Color * double truncates each channel (color.hpp:47-53)
            const int lr = to_int(colour0 * lighting), lg = to_int(colour1 * lighting), lb = to_int(colour2 * lighting);
            if (d == (uint32_t)rp.depth || refl < 0) {                  // scene.hpp:115-117
                cr = lr; cg = lg; cb = lb;
                path_done = true;
            } else {
                scr.colour(d, 0) = lr; scr.colour(d, 1) = lg; scr.colour(d, 2) = lb;
                scr.refl(d) = refl;
                L.r = reflect_ray(v, n);                                // scene.hpp:119
                L.path = make_path(0, d + 1, s, 0);
                L.state = ST_INNER; new_query = true;
                return;
            }
        }
    }
    RTX_SECTION(3)          // local colour, reflection query start
    uint32_t levels = d;       // levels below the innermost colour
    bool inner_waits = false;  // TAIL: the innermost colour is a level whose lighting is a job's
    if (TAIL && elsewhere) {
        // the level's record holds the triangle's colour until the blend, where the job's total meets it
        scr.colour(d, 0) = colour0; scr.colour(d, 1) = colour1; scr.colour(d, 2) = colour2;
        scr.waiting() |= (int32_t)(1u << d);
        if (d == (uint32_t)rp.depth || refl < 0) {                      // scene.hpp:115-117
            path_done = true;
            inner_waits = true;
        } else {
            scr.refl(d) = refl;
            L.r = reflect_ray(v, n);                                    // scene.hpp:119; v = -(the ray that hit)
            L.path = make_path(0, d + 1, s, 0);
            L.state = ST_INNER; new_query = true;
            return;
        }
    }
    if (!path_done && !finish) return;
    uint32_t out = L.out;
    if (TAIL) {
        if (path_done) {
            // the chain has ended; lighting jobs may still be out: leave what the blend needs and count this party out
            scr.chain_end() = (int32_t)(levels | (inner_waits ? 0x100u : 0u));
            scr.out_index() = (int32_t)out;
            scr.innermost(0) = cr; scr.innermost(1) = cg; scr.innermost(2) = cb;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const uint32_t left = __hip_atomic_fetch_sub(pend + lane, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - 1u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            L.state = ST_FETCH;
            if (left != 0u) return;
        }
        // the last party blends from the pixel's records
        const uint32_t e = (uint32_t)px.chain_end(), waiting = (uint32_t)px.waiting();
        levels = e & 0xFFu;
        out = (uint32_t)px.out_index();
        if (e >> 8) {
            const double total = px.total(levels);
            cr = to_int(px.colour(levels, 0) * total); cg = to_int(px.colour(levels, 1) * total); cb = to_int(px.colour(levels, 2) * total);
        } else {
            cr = px.innermost(0); cg = px.innermost(1); cb = px.innermost(2);
        }
        // localColor * (1 - reflectivity) + reflectedColor * reflectivity (scene.hpp:122), innermost first
        while (levels > 0) {
            levels--;
            int lr = px.colour(levels, 0), lg = px.colour(levels, 1), lb = px.colour(levels, 2);
            if ((waiting >> levels) & 1u) {
                const double total = px.total(levels);
                lr = to_int(lr * total); lg = to_int(lg * total); lb = to_int(lb * total);
            }
            const double refl = px.refl(levels);
            const double keep = 1 - refl;
            cr = (int)((uint32_t)to_int(lr * keep) + (uint32_t)to_int(cr * refl));
            cg = (int)((uint32_t)to_int(lg * keep) + (uint32_t)to_int(cg * refl));
            cb = (int)((uint32_t)to_int(lb * keep) + (uint32_t)to_int(cb * refl));
        }
    } else {
        // localColor * (1 - reflectivity) + reflectedColor * reflectivity (scene.hpp:122), innermost first
        while (levels > 0) {
            // the records of two levels per trip to memory
            const bool two = levels >= 2u;
            const uint32_t la = levels - 1u, lb_ = two ? levels - 2u : la;
            const int ar = scr.colour(la, 0), ag = scr.colour(la, 1), ab = scr.colour(la, 2);
            const double arefl = scr.refl(la);
            const int br = scr.colour(lb_, 0), bg = scr.colour(lb_, 1), bb = scr.colour(lb_, 2);
            const double brefl = scr.refl(lb_);
            issued();
            // int + int wraps on both machines
            const double akeep = 1 - arefl;
            cr = (int)((uint32_t)to_int(ar * akeep) + (uint32_t)to_int(cr * arefl));
            cg = (int)((uint32_t)to_int(ag * akeep) + (uint32_t)to_int(cg * arefl));
            cb = (int)((uint32_t)to_int(ab * akeep) + (uint32_t)to_int(cb * arefl));
            if (two) {
                const double bkeep = 1 - brefl;
                cr = (int)((uint32_t)to_int(br * bkeep) + (uint32_t)to_int(cr * brefl));
                cg = (int)((uint32_t)to_int(bg * bkeep) + (uint32_t)to_int(cg * brefl));
                cb = (int)((uint32_t)to_int(bb * bkeep) + (uint32_t)to_int(cb * brefl));
            }
            levels -= two ? 2u : 1u;
        }
    }
    RTX_SECTION(4)          // blend
    // putPixel (src/main.cpp:90-94) + the clamp of saveCanvas (src/main.cpp:107)
    if (TAIL || rp.spp == 1) {
        if (rp.rgb) {
            rp.rgb[3 * (uint64_t)out + 0] = (uint8_t)clamp255(cr);
            rp.rgb[3 * (uint64_t)out + 1] = (uint8_t)clamp255(cg);
            rp.rgb[3 * (uint64_t)out + 2] = (uint8_t)clamp255(cb);
        }
        if (rp.rgb_i32) { rp.rgb_i32[3 * (uint64_t)out + 0] = cr; rp.rgb_i32[3 * (uint64_t)out + 1] = cg; rp.rgb_i32[3 * (uint64_t)out + 2] = cb; }
        L.state = ST_FETCH;
        return;
    }
    acc.r += clamp255(cr); acc.g += clamp255(cg); acc.b += clamp255(cb);
    if (s + 1 < (uint32_t)rp.spp) {
        primary_ray(L, sc, rp, (int)(L.pix & 0xFFFFu), (int)(L.pix >> 16), s + 1, rp.n_frames > 1 ? L.out / rp.frame_pixels : 0u);
        L.state = ST_INNER; new_query = true;
        return;
    }
    const int spp = rp.spp;
    if (rp.rgb) {
        rp.rgb[3 * (uint64_t)out + 0] = (uint8_t)((acc.r + spp / 2) / spp);
        rp.rgb[3 * (uint64_t)out + 1] = (uint8_t)((acc.g + spp / 2) / spp);
        rp.rgb[3 * (uint64_t)out + 2] = (uint8_t)((acc.b + spp / 2) / spp);
    }
    if (rp.rgb_i32) { rp.rgb_i32[3 * (uint64_t)out + 0] = acc.r; rp.rgb_i32[3 * (uint64_t)out + 1] = acc.g; rp.rgb_i32[3 * (uint64_t)out + 2] = acc.b; }
    L.state = ST_FETCH;
}

// Wave-uniform bookkeeping of the persistent loop.
struct Wave {
    // the 8x8 block being handed out (pixels cursor..63 are still free), where its top-left pixel lies in the frame
    // and in the packed-tile output
    uint32_t cursor, block_x0, block_y0, block_packed0;
    uint32_t frame;            // which frame of a launch over several frames that block belongs to
    bool drained;              // every queue ran past its last block
    uint32_t exhausted;        // queues this wave has seen empty
    uint32_t blocks_x, per_tile, home_queue;
    // STATS: rounds the wave ran of each kind and the lanes that took part (kept by lane 0)
    unsigned long long rounds[3], lane_steps[3];
#ifdef RTX_EXPERIMENT_KINDTIME
    // MEASUREMENT ONLY (tools/kind_time.py): shader-clock cycles this wave spent in the steps of each kind (node pair,
    // triangle, shading and refill) and in the rounds' own bookkeeping; leaves through the wave log
    unsigned long long kind_cycles[4];
    unsigned long long section_cycles[8];   // inside the shading step, see RTX_SECTION; [7] = the rest of the step
#endif
};
#ifdef RTX_EXPERIMENT_KINDTIME
#define RTX_KIND_BEGIN const unsigned long long kind_t0 = cycles_now();
#define RTX_KIND_END(k) w.kind_cycles[k] += cycles_now() - kind_t0;
#else
#define RTX_KIND_BEGIN
#define RTX_KIND_END(k)
#endif

// The next 8x8 block for a wave whose current one is used up: from the queue of this wave's XCD first (its tiles are
// screen neighbours, so the XCD's L2 keeps seeing the same part of the BVH), then from the other queues.  Which queue
// served a block only affects speed.  Every queue hands its blocks out `n_serve` times over, one whole pass after the
// other (the frames of a launch over several frames, the passes of a staged launch); w.frame says which pass the block
// belongs to.  Leaves w.drained set when every queue has run past its last block.
__device__ __forceinline__ void draw_block(Wave &w, const DevRender &rp, uint32_t n_serve, uint32_t lane, uint64_t gw) {
    w.drained = true;
    for (uint32_t k = 0; k < rp.n_queues && w.drained; k++) {
        const uint32_t q = (w.home_queue + k) % rp.n_queues;
        if (w.exhausted & (1u << q)) continue;
        const uint32_t q_begin = rp.tile_order[q], q_end = rp.tile_order[q + 1], q_tiles = q_end - q_begin;
        // a table that is not one (a caller broke the ordering of the launches that share it) must not
        // become an address: the queue then counts as empty
        if (q_end < q_begin || q_end > (uint32_t)rp.owned_tiles) { w.exhausted |= 1u << q; continue; }
        uint32_t i = 0;
        if (lane == 0) i = atomicAdd(rp.work + q * 32u, 1u);
        i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
        const uint32_t per_pass = q_tiles * w.per_tile;
        if (i >= per_pass * n_serve) { w.exhausted |= 1u << q; continue; }
        if (n_serve > 1) { w.frame = i / per_pass; i -= w.frame * per_pass; }
        // all wave-uniform: local tile, block inside it, frame position
        const uint32_t local_tile = rp.tile_order[9u + q_begin + i / w.per_tile], wb = i % w.per_tile;
        const uint32_t tile = (rp.tile_count > 1) ? local_tile * (uint32_t)rp.tile_count + (uint32_t)rp.tile_rank : local_tile;
        const uint32_t bx = (wb % w.blocks_x) * 8u, by = (wb / w.blocks_x) * 8u;
        w.block_x0 = (tile % (uint32_t)rp.tiles_x) * (uint32_t)rp.tile_w + bx;
        w.block_y0 = (tile / (uint32_t)rp.tiles_x) * (uint32_t)rp.tile_h + by;
        w.block_packed0 = (local_tile * (uint32_t)rp.tile_h + by) * (uint32_t)rp.tile_w + bx;
        w.cursor = 0;
        w.drained = false;
        if (rp.wave_log && lane == 0) {      // diagnostics: when and where this wave drew its latest block
            rp.wave_log[4 * gw + 2] = wall_clock64();
            rp.wave_log[4 * gw + 3] = (unsigned long long)w.block_x0 | ((unsigned long long)w.block_y0 << 16) | ((unsigned long long)q << 32);
        }
    }
}

// The rounds of a wave.  TAIL = false: until every lane is done (returns false) or, with `jobs_possible`, until the
// first lane is (returns true: the queues are empty, the caller goes on with TAIL = true).  TAIL = true: no lane takes a
// new pixel any more, lanes without work take lighting jobs; until every lane is done.
template <bool PRUNE, bool STATS, bool TAIL>
__device__ __forceinline__ bool wave_loop(Lane &L, PixelSum &acc, Wave &w, const DevScene &sc, const DevRender &rp, const Hot &hot,
                                          uint32_t *stack, uint32_t stride, const Scratch &scr, Counters &c, uint32_t *pend,
                                          uint32_t *surf, uint32_t lane, uint64_t gw, bool jobs_possible) {
    for (;;) {
#ifdef RTX_EXPERIMENT_KINDTIME
        const unsigned long long round_t0 = cycles_now();
        struct RoundClock {                 // every way out of the round's body adds the round to the total
            Wave &w; unsigned long long t0;
            __device__ ~RoundClock() { w.kind_cycles[3] += cycles_now() - t0; }
        } round_clock{w, round_t0};
#endif
        const unsigned long long m_inner = __ballot(L.state == ST_INNER);
        const unsigned long long m_leaf = __ballot(L.state == ST_LEAF);
        const unsigned long long m_shade = __ballot(L.state == ST_SHADE);
        const unsigned long long m_fetch = __ballot(L.state == ST_FETCH);
        if ((m_inner | m_leaf | m_shade | m_fetch) == 0) return false;  // every lane done
        const unsigned long long m_idle = ~(m_inner | m_leaf | m_shade | m_fetch);
        if (!TAIL && jobs_possible && m_idle != 0ull) return true;
        // wave-uniform choice: the kind of step most lanes wait for (weighted).  With only a
        // few live lanes left (the end of the frame) every kind that has a lane waiting runs, so
        // that no lane sits out a round.
        const int n_inner = __popcll(m_inner), n_leaf = __popcll(m_leaf), n_shade = __popcll(m_shade | m_fetch);
        const int w_inner = n_inner * hot.w_inner, w_leaf = n_leaf * hot.w_leaf, w_shade = n_shade * hot.w_shade;
        const bool all_kinds = n_inner + n_leaf + n_shade <= hot.tail_lanes;
        const bool pick_inner = w_inner >= w_leaf && w_inner >= w_shade;
        const bool pick_leaf = !pick_inner && w_leaf >= w_shade;
        if (pick_inner || (all_kinds && n_inner) || n_inner >= hot.also_lanes) {
            if (STATS && lane == 0) { w.rounds[0]++; w.lane_steps[0] += n_inner; }
            RTX_KIND_BEGIN
            if (L.state == ST_INNER) inner_step<PRUNE, STATS>(L, hot, stack, stride, c);
            RTX_KIND_END(0)
        }
        if (pick_leaf || (all_kinds && n_leaf) || n_leaf >= hot.also_leaf) {
            if (STATS) { const int k = __popcll(__ballot(L.state == ST_LEAF)); if (lane == 0) { w.rounds[1]++; w.lane_steps[1] += k; } }
            RTX_KIND_BEGIN
            if (L.state == ST_LEAF) leaf_step<PRUNE, STATS>(L, hot, stack, stride, c);
            RTX_KIND_END(1)
        }
        if ((!pick_inner && !pick_leaf) || (all_kinds && n_shade) || n_shade >= hot.also_shade) {
#ifdef RTX_EXPERIMENT_KINDTIME
            struct KindClock {
                Wave &w; unsigned long long t0;
                __device__ ~KindClock() { w.kind_cycles[2] += cycles_now() - t0; }
            } shade_clock{w, cycles_now()};
#endif
            unsigned long long handed_over = 0ull;
            if (TAIL) {
                // closest-hit results that found a surface meet the lanes without work, one pair at a time (rare, and
                // wave-uniform): the idle lane takes a copy of the result and continues as that hit's lighting job
                unsigned long long offers = __ballot(L.state == ST_SHADE && (L.path & (1u | kJob)) == 0u && L.best_pos >= 0);
                if (rp.lighting_jobs == 2) offers = 0ull;               // experiments: the second loop without its jobs
                unsigned long long idle = m_idle;
                while (offers != 0ull && idle != 0ull) {
                    const int a = __builtin_ctzll(offers), h = __builtin_ctzll(idle);
                    offers &= offers - 1ull;
                    idle &= idle - 1ull;
                    handed_over |= 1ull << a;
                    const double ox = lane_value(L.o.x, a), oy = lane_value(L.o.y, a), oz = lane_value(L.o.z, a);
                    const double rx = lane_value(L.r.x, a), ry = lane_value(L.r.y, a), rz = lane_value(L.r.z, a);
                    const double bt = lane_value(L.best_t, a);
                    const int bp = __builtin_amdgcn_readlane(L.best_pos, a);
                    const uint32_t pa = (uint32_t)__builtin_amdgcn_readlane((int)L.path, a);
                    if ((int)lane == h) {
                        L.o = {ox, oy, oz}; L.r = {rx, ry, rz};
                        L.best_t = bt; L.best_pos = bp;
                        L.path = pa | kJob | ((uint32_t)a << 5);
                        L.sp = 0;
                        L.state = ST_SHADE;
                    }
                }
                if ((handed_over >> lane) & 1ull) pend[lane] += 1u;     // one more party at work on this lane's pixel
            }
            if (STATS) { const int k = __popcll(__ballot(L.state == ST_SHADE)); if (lane == 0) { w.rounds[2]++; w.lane_steps[2] += k; } }
            bool new_query = false;     // the lane has a new ray in L.o, L.r, L.path: a shadow ray, a reflection or a camera ray
            if (L.state == ST_SHADE)
                {
#ifdef RTX_EXPERIMENT_KINDTIME
                    const unsigned long long t_ = cycles_now();
                    shade_step<PRUNE, STATS, TAIL>(L, acc, sc, rp, scr, c, ((handed_over >> lane) & 1ull) != 0ull, pend, surf, hot.lights, lane, new_query, w.section_cycles);
                    w.section_cycles[6] += cycles_now() - t_;      // the whole step
#else
                    shade_step<PRUNE, STATS, TAIL>(L, acc, sc, rp, scr, c, ((handed_over >> lane) & 1ull) != 0ull, pend, surf, hot.lights, lane, new_query);
#endif
                }
            if (TAIL) {
                if (L.state == ST_FETCH) L.state = ST_DONE;             // the queues are empty
                if (new_query) start_query(L, sc, PRUNE, hot.root_ref);
                continue;
            }
            // hand pixels to the lanes without one, in lane order
            unsigned long long need = __ballot(L.state == ST_FETCH);
            while (need) {
                if (w.cursor == 64u && !w.drained) {
#ifdef RTX_EXPERIMENT_KINDTIME
                    const unsigned long long t_ = cycles_now();
                    draw_block(w, rp, (uint32_t)rp.n_frames, lane, gw);
                    w.section_cycles[5] += cycles_now() - t_;
#else
                    draw_block(w, rp, (uint32_t)rp.n_frames, lane, gw);
#endif
                }
                if (w.cursor == 64u) {                                  // nothing left anywhere
                    if (L.state == ST_FETCH) L.state = ST_DONE;
                    break;
                }
                const uint32_t avail = 64u - w.cursor;
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                if (L.state == ST_FETCH && rank < avail) {
                    const uint32_t in = w.cursor + rank, ix = in & 7u, iy = in >> 3;
                    const uint32_t col = w.block_x0 + ix, row = w.block_y0 + iy;
                    // outside the frame or the window: nothing to render, the lane asks again
                    if ((int)col >= rp.x0 && (int)col < rp.x1 && (int)row >= rp.y0 && (int)row < rp.y1) {
                        L.pix = col | (row << 16);
                        L.out = ((rp.layout == RTX_LAYOUT_PACKED_TILES_) ? w.block_packed0 + iy * (uint32_t)rp.tile_w + ix
                                                                         : row * (uint32_t)rp.W + col) + w.frame * rp.frame_pixels;
                        acc = {0, 0, 0};
                        primary_ray(L, sc, rp, (int)col, (int)row, 0, w.frame);
                        L.state = ST_INNER;
                        new_query = true;
                    }
                }
                const uint32_t n_need = (uint32_t)__popcll(need);
                w.cursor += n_need < avail ? n_need : avail;
                // lanes still without a pixel: those the block could not serve and those whose pixel
                // was outside the window
                need = __ballot(L.state == ST_FETCH);
            }
            // every new ray of the round starts here: shadow rays, reflections, camera rays
            if (new_query) start_query(L, sc, PRUNE, hot.root_ref);
        }
    }
}

// ---------------------------------------------------------------------------------
// The staged pipeline: the same steps, regrouped.  In render_kernel's own loop a lane owns a pixel from its camera
// ray to its last reflection, up to 2 (depth + 1) queries one after the other, and the launch ends when the longest
// such chain does.  Here the frame is depth + 3 launches (DevRender::stage), and between launches every pixel's state
// lies in memory, so that each launch hands out single queries: launch l traces the level-l closest-hit queries of all
// pixels that still have one (and, in a pass of its own, the lighting of their level l-1 hits, whose shadow queries do
// not depend on the reflection: scene.hpp:113 and :119-121 only meet in the blend).  A wave's lanes then run rays of one
// kind -- all reflections of one level, or all shadow rays, which share their direction for a directional light -- lanes
// take the next query of the wave's block as soon as theirs ends, and the longest dependent chain of a launch is one
// query (or one hit's shadow queries).  Which pixels of an 8x8 block have a hit at a level is a 64-bit mask per block and
// level (DevRender::hitmask), set by the lanes that find the hits; the next launch reads a block's mask once and hands
// out exactly those pixels, so a block or pixel whose chain has ended costs nothing later.  Results are the fused
// kernel's, bit for bit: the same operations on the same values, the hit point and ray of a level travel as the doubles
// they are.

// The lighting of a level is complete: its localColor and reflectivity go to the pixel's records.
__device__ __forceinline__ void stage_level_done(const DevScene &sc, const DevRender &rp, const Scratch &scr, uint32_t lv, uint64_t out) {
    // Color * double truncates each channel (color.hpp:47-53); the reflectivity decides in the next launch and in the blend
    const uint64_t P = rp.rec_pixels;
    const int32_t pos = scr.i(0);
    const DevMat &mt = sc.mats[pos];
    const double total = scr.f(3);
    int32_t *local = rp.lvl_local + (uint64_t)lv * 3u * P + out;
    local[0] = to_int(mt.color[0] * total); local[P] = to_int(mt.color[1] * total); local[2 * P] = to_int(mt.color[2] * total);
    rp.lvl_refl[(uint64_t)lv * P + out] = mt.reflectivity;
}

__device__ __forceinline__ void put_pixel(const DevRender &rp, uint64_t out, int cr, int cg, int cb) {
    // putPixel (src/main.cpp:90-94) + the clamp of saveCanvas (src/main.cpp:107)
    if (rp.rgb) {
        rp.rgb[3 * out + 0] = (uint8_t)clamp255(cr);
        rp.rgb[3 * out + 1] = (uint8_t)clamp255(cg);
        rp.rgb[3 * out + 2] = (uint8_t)clamp255(cb);
    }
    if (rp.rgb_i32) { rp.rgb_i32[3 * out + 0] = cr; rp.rgb_i32[3 * out + 1] = cg; rp.rgb_i32[3 * out + 2] = cb; }
}

// Starts the shadow query of the first light from `li` on that can add anything to the surface whose -ray, lighting sum and
// triangle lie in the lane's scratch (computeTotalLighting, scene.hpp:130-142: point lights first, then directional ones).
// False: no such light is left.
template <bool PRUNE>
__device__ __forceinline__ bool stage_next_light(Lane &L, const DevScene &sc, const Scratch &scr, uint32_t root_ref, uint32_t lv, uint32_t li) {
    V3 sn = {0, 0, 0}, sv = {0, 0, 0};
    int sspec = -1;
    if (PRUNE) {
        const int32_t pos = scr.i(0);
        sn = ld3(sc.tris[pos].n); sv = {scr.f(0), scr.f(1), scr.f(2)}; sspec = sc.mats[pos].specularity;
    }
    for (; li < sc.n_point + sc.n_dir; li++) {
        const V3 lvec = ld3(sc.lights[li].v);
        const V3 light = (li < sc.n_point) ? sub(lvec, L.o) : lvec;
        if (PRUNE && light_adds_nothing(sn, sv, sspec, light)) continue;
        L.r = light;
        L.path = make_path(1, lv, 0, li);
        start_query(L, sc, PRUNE, root_ref);
        return true;
    }
    return false;
}

// A finished query of a staged launch.  Leaves the lane with the next shadow query of the same hit or without work.
template <bool PRUNE, bool STATS>
__device__ __forceinline__ void stage_finish(Lane &L, const DevScene &sc, const DevRender &rp, const Scratch &scr, Counters &c,
                                             uint32_t root_ref) {
    const uint64_t P = rp.rec_pixels, out = L.out;
    const uint32_t kind = path_kind(L.path), lv = path_d(L.path);
    if (kind == 0) {
        // Scene::traceRay after the closest-hit query (scene.hpp:97-111): the level's record
        if (STATS) c.v[lv == 0 ? C_PRIMARY : C_REFLECTION]++;
        if (lv == 0 && (rp.hit || rp.t)) {
            const uint64_t k = (uint64_t)(L.pix >> 16) * (uint64_t)rp.W + (uint64_t)(L.pix & 0xFFFFu);
            if (rp.hit) rp.hit[k] = (L.best_pos >= 0) ? (int32_t)sc.tris[L.best_pos].orig_id : -1;
            if (rp.t) rp.t[k] = L.best_t;
        }
        if (L.best_pos >= 0) {
            if (STATS) c.v[C_CLOSEST]++;
            const V3 pt = add(L.o, smul(L.best_t, L.r));               // point, scene.hpp:105
            double *rec_pt = rp.rec_pt + (uint64_t)(lv & 1u) * 3u * P + out, *rec_ray = rp.rec_ray + (uint64_t)(lv & 1u) * 3u * P + out;
            rec_pt[0] = pt.x; rec_pt[P] = pt.y; rec_pt[2 * P] = pt.z;
            rec_ray[0] = L.r.x; rec_ray[P] = L.r.y; rec_ray[2 * P] = L.r.z;
            rp.rec_pos[(uint64_t)lv * P + out] = L.best_pos;
            // the pixel's bit in its block's mask of this level: (row & 7) * 8 + (col & 7), blocks are 8-aligned in the frame
            const uint32_t bit = ((L.pix >> 13) & 0x38u) | (L.pix & 7u);
            (void)__hip_atomic_fetch_or(rp.hitmask + (uint64_t)lv * rp.total_blocks + L.blk, 1ull << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lv == 0) {
            put_pixel(rp, out, sc.bg[0], sc.bg[1], sc.bg[2]);           // backgroundColor, scene.hpp:101-103: the pixel is done
        }
        L.state = ST_FETCH;
        return;
    }
    // the rest of Scene::computeLighting (scene.hpp:159-182) with light = r, point = o
    uint32_t li = path_li(L.path);
    if (STATS) { c.v[C_SHADOW]++; if (L.best_pos >= 0) c.v[C_OCCLUDED]++; }
    if (L.best_pos < 0) {
        const int32_t pos = scr.i(0);
        const V3 n = ld3(sc.tris[pos].n);
        const V3 v = {scr.f(0), scr.f(1), scr.f(2)};
        const int spec = sc.mats[pos].specularity;
        const double intensity = sc.lights[li].intensity;
        double term = 0.0;
        const double nl = dot(n, L.r);
        if (nl > 0.0) term += intensity * (nl / (mag(n) * mag(L.r)));
        __builtin_amdgcn_sched_barrier(0);
        if (spec != -1) {
            const V3 R = reflect_ray(L.r, n);
            const double rv = dot(R, v);
            if (rv > 0) {
                const double cosv = rv / (mag(R) * mag(v));
                term += intensity * ((spec == 1) ? cosv : general_pow(cosv, (double)spec));
            }
        }
        scr.f(3) += term;
    }
    li++;
    if (li < sc.n_point + sc.n_dir && stage_next_light<PRUNE>(L, sc, scr, root_ref, lv, li)) return;
    stage_level_done(sc, rp, scr, lv, out);
    L.state = ST_FETCH;
}

// A lane without work has been given a pixel of the wave's block (L.pix, L.out, L.blk set): start that pixel's item of
// this pass.  Leaves the lane in ST_INNER (a query runs) or in ST_FETCH (nothing to trace: it asks again).
template <bool PRUNE, bool STATS>
__device__ __forceinline__ void stage_begin(Lane &L, const DevScene &sc, const DevRender &rp, const Scratch &scr, uint32_t root_ref,
                                            int32_t kind) {
    const uint64_t P = rp.rec_pixels, out = L.out;
    const uint32_t stage = (uint32_t)rp.stage;
    if (kind == SP_PRIMARY) {
        primary_ray(L, sc, rp, (int)(L.pix & 0xFFFFu), (int)(L.pix >> 16), 0, 0);
        start_query(L, sc, PRUNE, root_ref);
        return;
    }
    if (kind == SP_REFLECT || kind == SP_LIGHT) {
        // the level stage-1 hit of this pixel (its bit is set in the block's mask of that level)
        const uint32_t from = stage - 1u;
        const double *rec_pt = rp.rec_pt + (uint64_t)(from & 1u) * 3u * P + out, *rec_ray = rp.rec_ray + (uint64_t)(from & 1u) * 3u * P + out;
        const int32_t pos = rp.rec_pos[(uint64_t)from * P + out];
        L.o = {rec_pt[0], rec_pt[P], rec_pt[2 * P]};
        const V3 v = muls(V3{rec_ray[0], rec_ray[P], rec_ray[2 * P]}, -1.0);     // -ray, vector3.hpp:108-110
        if (kind == SP_REFLECT) {
            // scene.hpp:115-121: the hit reflects unless its reflectivity is negative (the recursion depth is the host's
            // business: it launches no such pass beyond `depth`)
            const double refl = sc.mats[pos].reflectivity;
            const V3 n = ld3(sc.tris[pos].n);
            if (refl < 0) return;
            L.r = reflect_ray(v, n);                                    // scene.hpp:119
            L.path = make_path(0, stage, 0, 0);
            start_query(L, sc, PRUNE, root_ref);
            return;
        }
        scr.f(0) = v.x; scr.f(1) = v.y; scr.f(2) = v.z;
        scr.f(3) = sc.ambient;                                          // scene.hpp:128
        scr.i(0) = pos;
        if (!stage_next_light<PRUNE>(L, sc, scr, root_ref, from, 0u)) stage_level_done(sc, rp, scr, from, out);
        return;
    }
    // SP_RESOLVE (pixels whose camera ray hit): walk the pixel's levels down to where its chain ended, then blend back
    // up (localColor * (1 - reflectivity) + reflectedColor * reflectivity, scene.hpp:122), innermost first.  Level l + 1
    // was a hit iff the pixel's bit is set in the block's mask of that level.
    const uint32_t bit = ((L.pix >> 13) & 0x38u) | (L.pix & 7u);
    int cr = 0, cg = 0, cb = 0;
    uint32_t levels = 0;
    for (;; levels++) {
        const int32_t *local = rp.lvl_local + (uint64_t)levels * 3u * P + out;
        cr = local[0]; cg = local[P]; cb = local[2 * P];
        if (levels == (uint32_t)rp.depth || rp.lvl_refl[(uint64_t)levels * P + out] < 0) break;   // scene.hpp:115-117
        if (!((rp.hitmask[(uint64_t)(levels + 1u) * rp.total_blocks + L.blk] >> bit) & 1ull)) {
            cr = sc.bg[0]; cg = sc.bg[1]; cb = sc.bg[2];                // the reflected ray missed: backgroundColor, scene.hpp:101-103
            levels++;
            break;
        }
    }
    while (levels > 0) {
        levels--;
        const int32_t *local = rp.lvl_local + (uint64_t)levels * 3u * P + out;
        const int lr = local[0], lg = local[P], lb = local[2 * P];
        const double refl = rp.lvl_refl[(uint64_t)levels * P + out];
        const double keep = 1 - refl;
        // int + int wraps on both machines
        cr = (int)((uint32_t)to_int(lr * keep) + (uint32_t)to_int(cr * refl));
        cg = (int)((uint32_t)to_int(lg * keep) + (uint32_t)to_int(cg * refl));
        cb = (int)((uint32_t)to_int(lb * keep) + (uint32_t)to_int(cb * refl));
    }
    put_pixel(rp, out, cr, cg, cb);
}

// Wave-uniform bookkeeping of a staged launch's hand-out, beside Wave.
struct Run {
    uint32_t next, end, queue;     // blocks next..end-1 of `queue` are this wave's (one atomic draws several blocks)
    uint32_t blk;                  // the current block's index in the launch (local tile * blocks per tile + block in tile)
    unsigned long long mask;       // its pixels that still wait for a lane in this pass
};

// The next block with work for a wave whose current one is used up (run.mask == 0).  A draw takes rp.draw_run
// consecutive blocks of one queue with one atomic -- a pass over a frame is some 10^5 blocks, and a queue's counter is one
// address for the whole chip -- and a block whose mask of the level is empty is passed over at the cost of one scalar load.
__device__ __forceinline__ void draw_staged(Wave &w, Run &run, const DevRender &rp, uint32_t lane, uint64_t gw) {
    for (;;) {
        if (run.next == run.end) {
            w.drained = true;
            for (uint32_t k = 0; k < rp.n_queues && w.drained; k++) {
                const uint32_t q = (w.home_queue + k) % rp.n_queues;
                if (w.exhausted & (1u << q)) continue;
                const uint32_t q_begin = rp.tile_order[q], q_end = rp.tile_order[q + 1];
                if (q_end < q_begin || q_end > (uint32_t)rp.owned_tiles) { w.exhausted |= 1u << q; continue; }
                const uint32_t total = (q_end - q_begin) * w.per_tile * (uint32_t)rp.n_passes;
                uint32_t i = 0;
                if (lane == 0) i = atomicAdd(rp.work + q * 32u, rp.draw_run);
                i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
                if (i >= total) { w.exhausted |= 1u << q; continue; }
                run.queue = q;
                run.next = i;
                run.end = (total - i < rp.draw_run) ? total : i + rp.draw_run;
                w.drained = false;
            }
            if (w.drained) return;
        }
        const uint32_t q_begin = rp.tile_order[run.queue], per_pass = (rp.tile_order[run.queue + 1] - q_begin) * w.per_tile;
        uint32_t i = run.next++;
        w.frame = i / per_pass;                                  // the pass
        i -= w.frame * per_pass;
        const uint32_t local_tile = rp.tile_order[9u + q_begin + i / w.per_tile], wb = i % w.per_tile;
        run.blk = local_tile * w.per_tile + wb;
        const int32_t kind = rp.pass_kind[w.frame];
        // which pixels of the block have an item in this pass: every pixel for the camera rays, else the hits of the
        // level this launch continues (the blend: the hits of the camera rays)
        run.mask = kind == SP_PRIMARY ? ~0ull : rp.hitmask[(uint64_t)(kind == SP_RESOLVE ? 0u : (uint32_t)rp.stage - 1u) * rp.total_blocks + run.blk];
        if (run.mask == 0ull) continue;
        const uint32_t tile = (rp.tile_count > 1) ? local_tile * (uint32_t)rp.tile_count + (uint32_t)rp.tile_rank : local_tile;
        const uint32_t bx = (wb % w.blocks_x) * 8u, by = (wb / w.blocks_x) * 8u;
        w.block_x0 = (tile % (uint32_t)rp.tiles_x) * (uint32_t)rp.tile_w + bx;
        w.block_y0 = (tile / (uint32_t)rp.tiles_x) * (uint32_t)rp.tile_h + by;
        w.block_packed0 = (local_tile * (uint32_t)rp.tile_h + by) * (uint32_t)rp.tile_w + bx;
        if (rp.wave_log && lane == 0) {
            rp.wave_log[4 * gw + 2] = wall_clock64();
            rp.wave_log[4 * gw + 3] = (unsigned long long)w.block_x0 | ((unsigned long long)w.block_y0 << 16) | ((unsigned long long)run.queue << 32);
        }
        return;
    }
}

// The rounds of a wave in a staged launch: render_kernel's scheduler over the same traversal steps.  `mailbox`: 64 words
// of LDS of this wave's own.
template <bool PRUNE, bool STATS>
__device__ __forceinline__ void stage_loop(Lane &L, Wave &w, const DevScene &sc, const DevRender &rp, const Hot &hot, uint32_t *stack,
                                           uint32_t stride, const Scratch &scr, Counters &c, uint32_t *mailbox, uint32_t lane, uint64_t gw) {
    Run run = {0u, 0u, 0u, 0u, 0ull};
    for (;;) {
        const unsigned long long m_inner = __ballot(L.state == ST_INNER);
        const unsigned long long m_leaf = __ballot(L.state == ST_LEAF);
        const unsigned long long m_shade = __ballot(L.state == ST_SHADE);
        unsigned long long m_fetch = __ballot(L.state == ST_FETCH);
        if (m_fetch != 0ull && w.drained && run.mask == 0ull) {         // the queues are empty: lanes without work are done
            if (L.state == ST_FETCH) L.state = ST_DONE;
            m_fetch = 0ull;
        }
        if ((m_inner | m_leaf | m_shade | m_fetch) == 0) return;        // every lane done
        // Starting an item costs the whole wave two trips to memory (the pixel's records, then its triangle), so lanes
        // without work are served in batches: when rp.refill_min of them wait, or when nothing else is left to do
        const bool refill = (int)__popcll(m_fetch) >= rp.refill_min || (m_inner | m_leaf | m_shade) == 0ull;
        const int n_inner = __popcll(m_inner), n_leaf = __popcll(m_leaf), n_shade = __popcll(m_shade | (refill ? m_fetch : 0ull));
        const int w_inner = n_inner * hot.w_inner, w_leaf = n_leaf * hot.w_leaf, w_shade = n_shade * hot.w_shade;
        const bool all_kinds = n_inner + n_leaf + n_shade <= hot.tail_lanes;
        const bool pick_inner = w_inner >= w_leaf && w_inner >= w_shade;
        const bool pick_leaf = !pick_inner && w_leaf >= w_shade;
        if (pick_inner || (all_kinds && n_inner) || n_inner >= hot.also_lanes) {
            if (STATS && lane == 0) { w.rounds[0]++; w.lane_steps[0] += n_inner; }
            if (L.state == ST_INNER) inner_step<PRUNE, STATS>(L, hot, stack, stride, c);
        }
        if (pick_leaf || (all_kinds && n_leaf) || n_leaf >= hot.also_leaf) {
            if (STATS) { const int k = __popcll(__ballot(L.state == ST_LEAF)); if (lane == 0) { w.rounds[1]++; w.lane_steps[1] += k; } }
            if (L.state == ST_LEAF) leaf_step<PRUNE, STATS>(L, hot, stack, stride, c);
        }
        if ((!pick_inner && !pick_leaf) || (all_kinds && n_shade) || n_shade >= hot.also_shade) {
            if (STATS) { const int k = __popcll(__ballot(L.state == ST_SHADE)); if (lane == 0) { w.rounds[2]++; w.lane_steps[2] += k; } }
            if (L.state == ST_SHADE) stage_finish<PRUNE, STATS>(L, sc, rp, scr, c, hot.root_ref);
            // hand the waiting pixels of the wave's block to the lanes without work: the k-th such lane takes the k-th pixel
            unsigned long long need = refill ? __ballot(L.state == ST_FETCH) : 0ull;
            while (need) {
                if (run.mask == 0ull && !w.drained) draw_staged(w, run, rp, lane, gw);
                if (run.mask == 0ull) {                                 // nothing left anywhere
                    if (L.state == ST_FETCH) L.state = ST_DONE;
                    break;
                }
                // lane j speaks for pixel j of the block: if that pixel waits and is among the first n_need waiting
                // ones, it leaves its number in the mailbox slot of its rank, where the lane of that rank finds it
                const uint32_t n_need = (uint32_t)__popcll(need);
                const unsigned long long below = (1ull << lane) - 1ull;
                const bool waits = (run.mask >> lane) & 1ull;
                const uint32_t my_rank = (uint32_t)__popcll(run.mask & below);
                const bool taken = waits && my_rank < n_need;
                if (taken) mailbox[my_rank] = lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's LDS accesses stay in this order
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const unsigned long long m_taken = __ballot(taken);
                const uint32_t rank = (uint32_t)__popcll(need & below);
                if (L.state == ST_FETCH && rank < (uint32_t)__popcll(m_taken)) {
                    const uint32_t in = mailbox[rank], ix = in & 7u, iy = in >> 3;
                    const uint32_t col = w.block_x0 + ix, row = w.block_y0 + iy;
                    // outside the frame or the window: nothing to render, the lane asks again
                    if ((int)col >= rp.x0 && (int)col < rp.x1 && (int)row >= rp.y0 && (int)row < rp.y1) {
                        L.pix = col | (row << 16);
                        L.out = (rp.layout == RTX_LAYOUT_PACKED_TILES_) ? w.block_packed0 + iy * (uint32_t)rp.tile_w + ix : row * (uint32_t)rp.W + col;
                        L.blk = run.blk;
                        stage_begin<PRUNE, STATS>(L, sc, rp, scr, hot.root_ref, rp.pass_kind[w.frame]);
                    }
                }
                run.mask &= ~m_taken;
                // the blend was the last reader of the block's masks: they go back to zero for the slot's next frame
                if (run.mask == 0ull && rp.pass_kind[w.frame] == SP_RESOLVE && lane <= (uint32_t)rp.depth)
                    rp.hitmask[(uint64_t)lane * rp.total_blocks + run.blk] = 0ull;
                need = __ballot(L.state == ST_FETCH);
            }
        }
    }
}

template <bool PRUNE, bool STATS, bool STAGED>
__global__ void __launch_bounds__(256, RTX_WAVES_PER_EU) render_kernel(const DevScene sc_arg, const DevRender rp_arg) {
    // The two argument blocks hold ~100 dwords.  Read as ordinary kernel arguments they are all
    // loaded into scalar registers up front and most of them spilled again; read through the
    // kernel-argument pointer, with its origin hidden from the optimiser, each field is loaded
    // where it is used (the fields the traversal steps use stay in registers, the rest cost one
    // scalar load in the shading step).
    (void)sc_arg;
    (void)rp_arg;
    typedef const char __attribute__((address_space(4))) *KernArgs;
    KernArgs ka = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    constexpr size_t rp_offset = (sizeof(DevScene) + alignof(DevRender) - 1) / alignof(DevRender) * alignof(DevRender);
    const DevScene &sc = *(const DevScene *)(ka);
    const DevRender &rp = *(const DevRender *)(ka + rp_offset);
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * waves_per_block + wave;   // resident wave slot
    const unsigned long long wave_t0 = (rp.wave_log || rp.ticks) ? wall_clock64() : 0;

    // the stack of a lane is strided by the block size, a power of two: levels are shifted, not multiplied
    const uint32_t stride = 31u - (uint32_t)__builtin_clz(rp.stack_stride);
    // the top of the tree goes into LDS behind the block's stacks
    const uint32_t n_top = sc.n_top_pairs;
    // (the layout behind the stacks is render_lds_bytes()'s: two rows of a word per lane, the lights, then the table)
    double *lights_lds = reinterpret_cast<double *>(lds + (uint32_t)(sc.bvh_depth + 3) * rp.stack_stride);
    d2 *top_lds = reinterpret_cast<d2 *>(lights_lds + 4u * kLdsLights);
    // first row behind the stacks: parties at work on each lane's pixel (lighting jobs, see shade_step)
    const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));    // wave-uniform, and known to be
    uint32_t *pend = lds + (uint32_t)(sc.bvh_depth + 1) * rp.stack_stride + wave_first;
    // second row: the triangle of the surface whose shadow queries the lane is running
    uint32_t *surf = lds + (uint32_t)(sc.bvh_depth + 2) * rp.stack_stride + wave_first;
    {
        const d2 *src = reinterpret_cast<const d2 *>(sc.top);
        for (uint32_t i = threadIdx.x; i < n_top * 8u; i += blockDim.x) top_lds[i] = src[i];
        const uint32_t n_lights = sc.n_point + sc.n_dir;
        if (n_lights <= kLdsLights && threadIdx.x < 4u * n_lights)
            lights_lds[threadIdx.x] = (threadIdx.x & 3u) == 0u ? sc.lights[threadIdx.x >> 2].intensity : sc.lights[threadIdx.x >> 2].v[(threadIdx.x & 3u) - 1u];
        __syncthreads();
    }
    const Hot hot = {sc.nodes, sc.tris, top_lds, lights_lds, n_top ? kTopFlag : 0u, sc.n_point, rp.w_inner, rp.w_leaf, rp.w_shade, rp.tail_lanes, rp.also_lanes, rp.also_leaf, rp.also_shade};
    uint32_t *stack = lds + threadIdx.x;
    const Scratch scr = {rp.scratch_f64, rp.scratch_i32, rp.scratch_lanes, gw * 64u + lane, (uint32_t)rp.depth};
    Counters c;
    if (STATS) for (int i = 0; i < 9; i++) c.v[i] = 0;

    Lane L;
    PixelSum acc = {0, 0, 0};
    L.state = ST_FETCH;
    L.best_t = RTX_INF; L.best_pos = -1; L.first = 0; L.count = 0; L.sp = 0; L.slack = 0; L.lim = {0, 0, 0};
    L.o = {0, 0, 0}; L.r = {0, 0, 1}; L.path = 0; L.pix = 0; L.out = 0; L.blk = 0;
    L.rd = RayDiv{0, 0, 0, 0};

    // The launch is persistent: it has only as many waves as the chip holds, and a wave keeps
    // drawing 8x8-pixel blocks from global work queues (`work`, `tile_order`) until they run dry.
    // A lane whose pixel is finished takes the next one right away, so lanes behind cheap pixels
    // (background, early misses) do not idle while their neighbours follow three reflections.
    Wave w;
    w.cursor = 64; w.block_x0 = 0; w.block_y0 = 0; w.block_packed0 = 0; w.frame = 0;
    w.drained = false;
    w.exhausted = 0;
    w.blocks_x = (uint32_t)rp.tile_w >> 3;
    w.per_tile = w.blocks_x * ((uint32_t)rp.tile_h >> 3);
    // HW_REG_XCC_ID (id 20), bits 3:0: the XCD this wave runs on
    w.home_queue = rp.n_queues > 1 ? ((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) % rp.n_queues : 0u;
    for (int i = 0; i < 3; i++) { w.rounds[i] = 0; w.lane_steps[i] = 0; }
#ifdef RTX_EXPERIMENT_KINDTIME
    for (int i = 0; i < 4; i++) w.kind_cycles[i] = 0;
    for (int i = 0; i < 8; i++) w.section_cycles[i] = 0;
#endif

    // Lighting jobs.  Once the queues are empty a wave runs on with ever fewer lanes, and what then sets the length of
    // the launch is the longest pixel: up to depth + 1 closest-hit queries and as many rounds of shadow queries, one
    // after the other on one lane.  From the moment the wave has lanes without work it therefore runs a second copy of
    // the loop, in which a lane that comes out of a closest-hit query with a hit hands the lighting of that hit to an
    // idle lane and follows the reflection at once (see shade_step): the pixel's dependent chain loses its shadow
    // queries.  One sample per pixel only (a job's marks live in the sample bits of `path`), and only in scenes with
    // lights.  The steady state runs the first copy, which knows nothing of all this.
    const bool jobs_possible = rp.spp == 1 && (sc.n_point + sc.n_dir) != 0u && rp.lighting_jobs != 0;
    if (STAGED) {
        stage_loop<PRUNE, STATS>(L, w, sc, rp, hot, stack, stride, scr, c, pend, lane, gw);
    } else if (wave_loop<PRUNE, STATS, false>(L, acc, w, sc, rp, hot, stack, stride, scr, c, pend, surf, lane, gw, jobs_possible)) {
        // lanes only run out of work when every queue is empty: from here on no lane takes a new pixel, and the
        // pixels in progress count one party each, their owner's chain
        pend[lane] = (L.state == ST_DONE) ? 0u : 1u;
        if (L.state != ST_DONE) scr.waiting() = 0;
        wave_loop<PRUNE, STATS, true>(L, acc, w, sc, rp, hot, stack, stride, scr, c, pend, surf, lane, gw, jobs_possible);
    }
    if (STATS && rp.stats) {
        for (int i = 0; i < 9; i++)
            if (c.v[i]) atomicAdd(&rp.stats[i], c.v[i]);
        if (lane == 0)
            for (int i = 0; i < 3; i++) {
                atomicAdd(&rp.stats[9 + i], w.rounds[i]);
                atomicAdd(&rp.stats[12 + i], w.lane_steps[i]);
            }
    }
    if (rp.wave_log && lane == 0) {
        rp.wave_log[4 * gw] = wave_t0;
        rp.wave_log[4 * gw + 1] = wall_clock64();
#ifdef RTX_EXPERIMENT_KINDTIME
        rp.wave_log[4 * gw + 2] = w.kind_cycles[0] | (w.kind_cycles[1] << 32);
        rp.wave_log[4 * gw + 3] = w.kind_cycles[2] | (w.kind_cycles[3] << 32);
        // the sections of the shading step: summed over the waves behind the log's rows (the caller allocates 8 more words there)
        for (int i = 0; i < 8; i++) atomicAdd(&rp.wave_log[4 * (unsigned long long)gridDim.x * waves_per_block + i], w.section_cycles[i]);
#endif
    }
    // first wave start / last wave end of the launch, one pair of atomics per block.  Both are maxima over a pair
    // the caller zeroed (the start as its complement), so that no clearing dispatch has to precede the launch.
    if (rp.ticks && threadIdx.x == 0) {
        atomicMax(&rp.ticks[0], ~wave_t0);
        atomicMax(&rp.ticks[1], (unsigned long long)wall_clock64());
    }
    // The last wave out puts the work counters back to zero for the next launch that uses this scratch slot
    // (launches that share a slot are ordered one behind the other), so a frame costs no dispatch but its own.
    if (lane == 0) {
        __threadfence();
        const uint32_t out = atomicAdd(rp.work + kWavesOutWord, 1u) + 1u;
        if (out == gridDim.x * waves_per_block) {
            for (uint32_t q = 0; q < 8u; q++) atomicExch(rp.work + q * 32u, 0u);
            atomicExch(rp.work + kWavesOutWord, 0u);
        }
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
    double q = div_by(a, b, d.yx, (d.flags & RD_ZX) != 0);
    double ref = a / b;
    unsigned long long qb = (unsigned long long)__double_as_longlong(q), rb = (unsigned long long)__double_as_longlong(ref);
    bool same = qb == rb || (q != q && ref != ref) || (q == 0.0 && ref == 0.0);   // NaN payload / sign of zero are free
    if (!(d.flags & RD_OK) || !same) atomicAdd(mismatches, 1ull);
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

// DevScene::top: the first `max_pairs` sibling pairs of the tree in breadth-first order.  One thread:
// a few hundred records at most, once per scene.
__global__ void __launch_bounds__(64) top_table_kernel(const DevNode *nodes, DevNode *top, uint32_t max_pairs, uint32_t *count) {
    __shared__ uint32_t slot_of[kTopPairsMax];
    if (threadIdx.x != 0) return;
    if (max_pairs > kTopPairsMax) max_pairs = kTopPairsMax;
    uint32_t n = 0;
    if (max_pairs) slot_of[n++] = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t s = slot_of[i];
        DevNode a = nodes[s], b = nodes[s + 1];
        // an inner child whose pair still fits is referred to by its place in the table; slot 0 is no node
        if (s != 0 && a.count == 0 && n < max_pairs) { slot_of[n] = a.first; a.first = kTopFlag | n; n++; }
        if (b.count == 0 && n < max_pairs) { slot_of[n] = b.first; b.first = kTopFlag | n; n++; }
        a.pad[0] = node_entry(s, a.first, a.count);          // the references to both nodes, as in `nodes`
        a.pad[1] = node_entry(s + 1u, b.first, b.count);
        top[2 * i] = a;
        top[2 * i + 1] = b;
    }
    *count = n;
}

// The work-queue table of a launch (layout: DevRender::tile_order), built where it is read.
// The launch's local tiles are grouped into macro tiles of macro x macro screen neighbours; macro
// tile (mx, my) belongs to queue (mx + 3 my) mod n_queues; a queue lists its macro tiles in
// row-major order and a macro tile its own tiles in row-major order.  One block: a count per macro
// tile, one thread per queue turns the counts into offsets, then every macro tile writes its tiles.
// `order` has room for 9 + owned tiles + 2 x macro tiles words (counts and serving order live behind the table).
__global__ void __launch_bounds__(1024) tile_order_kernel(uint32_t *order, int tiles_x, int tiles_y, int tile_rank, int tile_count,
                                                          uint32_t owned, int macro, int n_queues, int centre_first,
                                                          uint32_t *clear, uint32_t n_clear) {
    // `clear` (nullable): the launch's work counters, zeroed here for a caller that wants kernels only on its stream
    for (uint32_t i = threadIdx.x; i < n_clear; i += blockDim.x) clear[i] = 0u;
    const int macros_x = (tiles_x + macro - 1) / macro, macros_y = (tiles_y + macro - 1) / macro;
    const int n_macros = macros_x * macros_y;
    uint32_t *cnt = order + 9u + owned;          // per macro tile: its tiles, then its offset inside its queue
    uint32_t *seq = cnt + n_macros;              // per queue, one after the other: its macro tiles in serving order
    __shared__ uint32_t q_total[8], q_base[9];
    for (int pass = 0; pass < 2; pass++) {
        for (int m = (int)threadIdx.x; m < n_macros; m += (int)blockDim.x) {
            const int mx = m % macros_x, my = m / macros_x;
            const int tx1 = min(tiles_x, (mx + 1) * macro), ty1 = min(tiles_y, (my + 1) * macro);
            uint32_t k = pass ? 9u + q_base[(mx + 3 * my) % n_queues] + cnt[m] : 0u;
            for (int ty = my * macro; ty < ty1; ty++)
                for (int tx = mx * macro; tx < tx1; tx++) {
                    const uint32_t tile = (uint32_t)ty * (uint32_t)tiles_x + (uint32_t)tx;
                    if (tile_count > 1 && tile % (uint32_t)tile_count != (uint32_t)tile_rank) continue;
                    if (pass) order[k] = tile_count > 1 ? tile / (uint32_t)tile_count : tile;
                    k++;
                }
            if (!pass) cnt[m] = k;
        }
        __syncthreads();
        if (!pass) {
            if ((int)threadIdx.x < n_queues) {          // counts -> offsets inside the queue, in serving order
                const int q = (int)threadIdx.x;
                uint32_t first = 0, mine = 0;           // where this queue's macro tiles start in `seq`, how many
                for (int m = 0; m < n_macros; m++) {
                    const int qm = (m % macros_x + 3 * (m / macros_x)) % n_queues;
                    first += qm < q;
                    mine += qm == q;
                }
                uint32_t *list = seq + first;
                uint32_t k = 0;
                for (int m = 0; m < n_macros; m++)
                    if ((m % macros_x + 3 * (m / macros_x)) % n_queues == q) list[k++] = (uint32_t)m;
                // The frame's costly pixels (the model, its reflections) sit around the middle of the picture
                // in every scene of the reference; served first, their long dependent chains run while the
                // chip is still full, and the launch ends on cheap border tiles.  Insertion sort by distance
                // of the macro tile from the frame's centre (a few dozen entries per queue).
                if (centre_first && mine <= 2048u) {
                    auto key = [&](uint32_t m) {     // squared distance from the centre, each axis in units of the frame's extent
                        const long long dx = 2 * (long long)(m % (uint32_t)macros_x) + 1 - macros_x, dy = 2 * (long long)(m / (uint32_t)macros_x) + 1 - macros_y;
                        return dx * dx * macros_y * macros_y + dy * dy * macros_x * macros_x;
                    };
                    for (uint32_t i = 1; i < mine; i++) {
                        const uint32_t m = list[i];
                        const long long km = key(m);
                        uint32_t j = i;
                        while (j > 0 && key(list[j - 1]) > km) { list[j] = list[j - 1]; j--; }
                        list[j] = m;
                    }
                }
                uint32_t run = 0;
                for (uint32_t i = 0; i < mine; i++) {
                    const uint32_t m = list[i], c = cnt[m];
                    cnt[m] = run;
                    run += c;
                }
                q_total[q] = run;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t run = 0;
                for (int q = 0; q < 8; q++) {
                    q_base[q] = run;
                    order[q] = run;
                    if (q < n_queues) run += q_total[q];
                }
                q_base[8] = run;
                order[8] = run;
            }
            __syncthreads();
        }
    }
}

// Scratch of the entry point that allocates nothing (rtx_render_device_noalloc): work counters,
// work-queue table and per-lane shading scratch of ONE launch.  Every GPU the library runs on
// holds its own copy at its own address until the process ends.
static __device__ __align__(256) unsigned char scratch[kNoallocScratchBytes];

}  // namespace

// Device address of `scratch` on `device` (which must be the current device).  In host code the
// array's name is only a handle for hipGetSymbolAddress; the address it returns is what kernels,
// memsets and copies are given.  Looked up once per device.
hipError_t noalloc_scratch(int device, void **ptr) {
    constexpr int kMaxDevices = 64;
    static void *addr[kMaxDevices] = {};
    static std::mutex lock;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(lock);
    if (!addr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        addr[device] = p;
    }
    *ptr = addr[device];
    return hipSuccess;
}

hipError_t launch_top_table(const DevNode *d_nodes, DevNode *d_top, uint32_t max_pairs, uint32_t *d_count, hipStream_t st) {
    hipLaunchKernelGGL(top_table_kernel, dim3(1), dim3(64), 0, st, d_nodes, d_top, max_pairs, d_count);
    return hipGetLastError();
}

hipError_t launch_tile_order(uint32_t *d_order, const DevRender &rp, int macro, int n_queues, int centre_first, bool clear_counters,
                             hipStream_t st) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, d_order, rp.tiles_x, rp.tiles_y, rp.tile_rank, rp.tile_count,
                       (uint32_t)rp.owned_tiles, macro, n_queues, centre_first, clear_counters ? rp.work : nullptr,
                       clear_counters ? (uint32_t)(kScratchHeaderBytes / 4) : 0u);
    return hipGetLastError();
}

int render_blocks_per_cu() { return RTX_WAVES_PER_EU; }

hipError_t launch_render(const DevScene &sc, const DevRender &rp, bool prune, bool stats, uint32_t block_threads,
                         uint32_t grid_blocks, hipStream_t st, bool staged) {
    if (rp.total_blocks == 0 || grid_blocks == 0) return hipSuccess;
    dim3 grid(grid_blocks), block(block_threads);
    const size_t lds = render_lds_bytes(sc.bvh_depth, block_threads) + (size_t)sc.n_top_pairs * 128;
    if (staged) {
        if (prune && stats) hipLaunchKernelGGL((render_kernel<true, true, true>), grid, block, lds, st, sc, rp);
        else if (prune) hipLaunchKernelGGL((render_kernel<true, false, true>), grid, block, lds, st, sc, rp);
        else if (stats) hipLaunchKernelGGL((render_kernel<false, true, true>), grid, block, lds, st, sc, rp);
        else hipLaunchKernelGGL((render_kernel<false, false, true>), grid, block, lds, st, sc, rp);
    } else if (prune && stats) hipLaunchKernelGGL((render_kernel<true, true, false>), grid, block, lds, st, sc, rp);
    else if (prune) hipLaunchKernelGGL((render_kernel<true, false, false>), grid, block, lds, st, sc, rp);
    else if (stats) hipLaunchKernelGGL((render_kernel<false, true, false>), grid, block, lds, st, sc, rp);
    else hipLaunchKernelGGL((render_kernel<false, false, false>), grid, block, lds, st, sc, rp);
    return hipGetLastError();
}

// Self-test input/output for the specular term: out[i] = general_pow(x[i], e[i]).
__global__ void __launch_bounds__(256) powtest_kernel(const double *x, const int32_t *e, double *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = general_pow(x[i], (double)e[i]);
}

// Self-test of the square root as this library's kernels use it (mag(), the normal of a triangle):
// out[i] = sqrt(x[i]).
__global__ void __launch_bounds__(256) sqrttest_kernel(const double *x, double *out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_sqrt(x[i]);
}

hipError_t launch_sqrttest(const double *d_x, double *d_out, uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sqrttest_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, d_x, d_out, n);
    return hipGetLastError();
}

hipError_t launch_powtest(const double *d_x, const int32_t *d_e, double *d_out, uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(powtest_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, d_x, d_e, d_out, n);
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
