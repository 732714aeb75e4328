// TEST INFRASTRUCTURE ONLY -- see rt_oracle.h.  CPU restatement of the reference
// render path; citations are file:line inside khaganmv/ray-tracer.
//
// Build with -ffp-contract=off: every floating-point expression below is
// written in the reference's evaluation order and must round once per
// operation, as the reference's baseline x86-64 -O3 build does.
#include "rt_oracle.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

const double INF = std::numeric_limits<double>::infinity();
// include/common.hpp:12  EPSILON = DBL_EPSILON * 2^20 = 2^-32
const double EPS = std::numeric_limits<double>::epsilon() * 1048576.0;

struct V3 {
    double x, y, z;
};

// include/vector3.hpp:42-44  (x*x' + y*y') + z*z'
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// include/vector3.hpp:46-52
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// include/vector3.hpp:122-160
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 smul(double s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
inline V3 muls(V3 v, double s) { return {v.x * s, v.y * s, v.z * s}; }
inline V3 divs(V3 v, double s) { return {v.x / s, v.y / s, v.z / s}; }
// include/vector3.hpp:108-110  operator-() is *this * -1
inline V3 neg(V3 v) { return muls(v, -1); }
// include/vector3.hpp:84-86
inline double mag(V3 v) { return std::sqrt(dot(v, v)); }
inline double comp(V3 v, int axis) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; }
// include/vector3.hpp:92-106  component-wise (a < b) ? a : b / (a > b) ? a : b
inline V3 vmin(V3 a, V3 b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
inline V3 vmax(V3 a, V3 b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }
// std::min / std::max as the reference uses them on doubles (include/bvh.hpp:220-231):
// min(a,b) = (b < a) ? b : a ; max(a,b) = (a < b) ? b : a  -- NOT fmin/fmax.
inline double smin(double a, double b) { return (b < a) ? b : a; }
inline double smax(double a, double b) { return (a < b) ? b : a; }

// include/util.hpp:12-14 and include/vector3.hpp:54-82.  Each rotate function of the reference
// evaluates cos(radians) and sin(radians) of one argument; g++ -O3 (the reference's Makefile:7)
// turns the pair into one sincos() call, whose sine is not always sin()'s to the last bit, so the
// call is spelled out here instead of left to the optimiser.
inline double to_radians(double deg) { return (deg / 180.0) * M_PI; }
inline V3 rot_x(V3 v, double deg) {
    double s, c;
    sincos(to_radians(deg), &s, &c);
    return {v.x, c * v.y - s * v.z, s * v.y + c * v.z};
}
inline V3 rot_y(V3 v, double deg) {
    double s, c;
    sincos(to_radians(deg), &s, &c);
    return {c * v.x + s * v.z, v.y, -s * v.x + c * v.z};
}
inline V3 rot_z(V3 v, double deg) {
    double s, c;
    sincos(to_radians(deg), &s, &c);
    return {c * v.x - s * v.y, s * v.x + c * v.y, v.z};
}

// include/triangle.hpp:8-25
struct Tri {
    V3 v0, v1, v2;
    int32_t color[3];
    V3 normal;
    int32_t specularity;
    double reflectivity;
    V3 centroid;
};

Tri make_tri(V3 a, V3 b, V3 c, const int32_t col[3], int32_t spec, double refl) {
    Tri t;
    t.v0 = a; t.v1 = b; t.v2 = c;
    t.color[0] = col[0]; t.color[1] = col[1]; t.color[2] = col[2];
    t.specularity = spec;
    t.reflectivity = refl;
    V3 e01 = sub(b, a), e02 = sub(c, a);
    V3 n = cross(e01, e02);
    t.normal = divs(n, mag(n));                 // triangle.hpp:23
    t.centroid = divs(add(add(a, b), c), 3);    // triangle.hpp:24
    return t;
}

// include/triangle.hpp:32-55.  Returns the hit flag; *t_out is written whenever
// the plane test passes (the caller range-checks t).
inline bool tri_hit(const Tri &tr, V3 o, V3 r, double *t_out) {
    double nd = dot(tr.normal, r);
    if (std::fabs(nd) < EPS) { *t_out = 0.0; return false; }
    double d = -dot(tr.normal, tr.v0);
    double t = -(dot(tr.normal, o) + d) / nd;
    V3 p = add(o, smul(t, r));
    V3 e01 = sub(tr.v1, tr.v0), e12 = sub(tr.v2, tr.v1), e20 = sub(tr.v0, tr.v2);
    V3 e0p = sub(p, tr.v0), e1p = sub(p, tr.v1), e2p = sub(p, tr.v2);
    *t_out = t;
    return dot(tr.normal, cross(e01, e0p)) > EPS && dot(tr.normal, cross(e12, e1p)) > EPS &&
           dot(tr.normal, cross(e20, e2p)) > EPS;
}

// include/bvh.hpp:25-30 (64-byte record, dumped as is)
struct Node {
    V3 bmin, bmax;
    uint64_t first, count;
};
static_assert(sizeof(Node) == 64, "node record");

struct Box {
    V3 lo{INF, INF, INF}, hi{-INF, -INF, -INF};
    void grow(V3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }             // bvh.hpp:62-65
    void grow(const Box &b) { if (b.lo.x != INF) { grow(b.lo); grow(b.hi); } }  // bvh.hpp:67-72
    double area() const {                                              // bvh.hpp:74-78
        V3 e = sub(hi, lo);
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
};

struct Light { double intensity; V3 v; };

struct Counters {
    uint64_t primary = 0, reflection = 0, shadow = 0, occluded = 0, closest_hits = 0;
    uint64_t shadow_idle = 0, shadow_idle_occluded = 0;   // shadow queries whose answer cannot change the light's term (it is 0 either way)
    uint64_t nodes = 0, box_hits = 0, leaves = 0, tris = 0;
};

}  // namespace

struct ora_scene {
    V3 viewport, cam_pos, cam_rot;
    int32_t bg[3];
    std::vector<Tri> tris;       // Scene::triangles (materials read from here)
    double ambient;
    std::vector<Light> point, directional;
    // BVH (include/bvh.hpp:32-37); geometry is read through `tris` as well: the
    // reference's BVH holds a copy taken before the material overrides, and only
    // its geometry is ever read.
    std::vector<Node> nodes;
    std::vector<uint64_t> indices;
    uint64_t nodes_used = 0;
    int32_t max_depth = 0;
};

namespace {

thread_local std::string g_err;

// ---------------------------------------------------------------- BVH build
void update_bounds(ora_scene &s, uint64_t idx) {                        // bvh.hpp:100-115
    Node &n = s.nodes[idx];
    n.bmin = {INF, INF, INF};
    n.bmax = {-INF, -INF, -INF};
    for (uint64_t i = 0; i < n.count; i++) {
        const Tri &t = s.tris[s.indices[n.first + i]];
        n.bmin = vmin(n.bmin, t.v0); n.bmin = vmin(n.bmin, t.v1); n.bmin = vmin(n.bmin, t.v2);
        n.bmax = vmax(n.bmax, t.v0); n.bmax = vmax(n.bmax, t.v1); n.bmax = vmax(n.bmax, t.v2);
    }
}

double best_split(const ora_scene &s, uint64_t idx, int *axis_out, double *pos_out) {  // bvh.hpp:236-305
    const Node &n = s.nodes[idx];
    const int NB = 8;
    double best = INF;
    for (int axis = 0; axis < 3; axis++) {
        double lo = INF, hi = -INF;
        for (uint64_t i = 0; i < n.count; i++) {
            double c = comp(s.tris[s.indices[n.first + i]].centroid, axis);
            lo = smin(lo, c);
            hi = smax(hi, c);
        }
        if (lo == hi) continue;
        Box bounds[NB];
        int counts[NB] = {0};
        double scale = NB / (hi - lo);
        for (uint64_t i = 0; i < n.count; i++) {
            const Tri &t = s.tris[s.indices[n.first + i]];
            size_t b = static_cast<size_t>((comp(t.centroid, axis) - lo) * scale);
            if (b > NB - 1) b = NB - 1;
            counts[b]++;
            bounds[b].grow(t.v0); bounds[b].grow(t.v1); bounds[b].grow(t.v2);
        }
        double la[NB - 1], ra[NB - 1];
        int lc[NB - 1], rc[NB - 1];
        Box lbox, rbox;
        int lsum = 0, rsum = 0;
        for (int i = 0; i < NB - 1; i++) {
            lsum += counts[i]; lc[i] = lsum;
            lbox.grow(bounds[i]); la[i] = lbox.area();
            rsum += counts[NB - 1 - i]; rc[NB - 2 - i] = rsum;
            rbox.grow(bounds[NB - 1 - i]); ra[NB - 2 - i] = rbox.area();
        }
        double step = (hi - lo) / NB;
        for (int i = 0; i < NB - 1; i++) {
            // an empty side has area +inf and count 0 -> NaN cost -> never "<"
            double cost = lc[i] * la[i] + rc[i] * ra[i];
            if (cost < best) {
                *axis_out = axis;
                *pos_out = lo + step * static_cast<double>(static_cast<size_t>(i + 1));
                best = cost;
            }
        }
    }
    return best;
}

void subdivide(ora_scene &s, uint64_t idx, int depth) {                 // bvh.hpp:117-162
    if (depth > s.max_depth) s.max_depth = depth;
    int axis = 0;
    double pos = 0;
    double best = best_split(s, idx, &axis, &pos);
    Node &n = s.nodes[idx];
    Box nb; nb.lo = n.bmin; nb.hi = n.bmax;
    if (best >= static_cast<double>(n.count) * nb.area()) return;       // bvh.hpp:80-82,124
    int i = static_cast<int>(n.first);
    int j = static_cast<int>(i + n.count - 1);
    while (i <= j) {
        if (comp(s.tris[s.indices[i]].centroid, axis) < pos) {
            i++;
        } else {
            uint64_t tmp = s.indices[j];
            s.indices[j] = s.indices[i];
            s.indices[i] = tmp;
            j--;
        }
    }
    int left_n = static_cast<int>(i - n.first);
    if (left_n == 0 || left_n == static_cast<int>(n.count)) return;
    uint64_t l = s.nodes_used++, r = s.nodes_used++;
    s.nodes[l].first = n.first;   s.nodes[l].count = left_n;
    s.nodes[r].first = i;         s.nodes[r].count = n.count - left_n;
    n.first = l;
    n.count = 0;
    update_bounds(s, l);
    update_bounds(s, r);
    subdivide(s, l, depth + 1);
    subdivide(s, r, depth + 1);
}

void build_bvh(ora_scene &s) {                                          // bvh.hpp:84-98
    size_t N = s.tris.size();
    s.indices.resize(N);
    for (size_t i = 0; i < N; i++) s.indices[i] = i;
    s.nodes.assign(2 * N - 1, Node{{0, 0, 0}, {0, 0, 0}, 0, 0});
    s.nodes[0].first = 0;
    s.nodes[0].count = N;
    s.nodes_used = 1;
    s.max_depth = 0;
    update_bounds(s, 0);
    subdivide(s, 0, 0);
}

// ---------------------------------------------------------------- queries
struct Hit { bool any; double t; int id; };

inline bool slab(V3 o, V3 r, V3 lo, V3 hi, double tlim) {              // bvh.hpp:213-234
    double tx1 = (lo.x - o.x) / r.x, tx2 = (hi.x - o.x) / r.x;
    double tmin = smin(tx1, tx2), tmax = smax(tx1, tx2);
    double ty1 = (lo.y - o.y) / r.y, ty2 = (hi.y - o.y) / r.y;
    tmin = smax(tmin, smin(ty1, ty2));
    tmax = smin(tmax, smax(ty1, ty2));
    double tz1 = (lo.z - o.z) / r.z, tz2 = (hi.z - o.z) / r.z;
    tmin = smax(tmin, smin(tz1, tz2));
    tmax = smin(tmax, smax(tz1, tz2));
    return tmin < tlim && tmax > 0 && tmax >= tmin;
}

Hit bvh_hit(const ora_scene &s, V3 o, V3 r, double tmin, double tmax, uint64_t idx, Counters &c) {  // bvh.hpp:164-211
    const Node &n = s.nodes[idx];
    c.nodes++;
    if (!slab(o, r, n.bmin, n.bmax, INF)) return {false, INF, -1};
    c.box_hits++;
    if (n.count > 0) {
        c.leaves++;
        Hit h{false, INF, -1};
        for (uint64_t i = 0; i < n.count; i++) {
            uint64_t ti = s.indices[n.first + i];
            double t;
            c.tris++;
            if (tri_hit(s.tris[ti], o, r, &t)) {
                if (t > tmin && t < tmax && t < h.t) {
                    h.any = true; h.t = t; h.id = static_cast<int>(ti);
                }
            }
        }
        return h;
    }
    Hit a = bvh_hit(s, o, r, tmin, tmax, n.first, c);
    Hit b = bvh_hit(s, o, r, tmin, tmax, n.first + 1, c);
    // equal t (incl. both INF): the second child's id is taken
    return {a.any || b.any, smin(a.t, b.t), (a.t < b.t) ? a.id : b.id};
}

Hit brute_hit(const ora_scene &s, V3 o, V3 r, double tmin, double tmax, Counters &c) {  // scene.hpp:189-211
    Hit h{false, INF, -1};
    for (size_t i = 0; i < s.tris.size(); i++) {
        double t;
        c.tris++;
        if (tri_hit(s.tris[i], o, r, &t)) {
            if (t > tmin && t < tmax && t < h.t) { h.any = true; h.t = t; h.id = static_cast<int>(i); }
        }
    }
    return h;
}

inline Hit query(const ora_scene &s, V3 o, V3 r, double tmin, double tmax, bool use_bvh, Counters &c) {
    return use_bvh ? bvh_hit(s, o, r, tmin, tmax, 0, c) : brute_hit(s, o, r, tmin, tmax, c);
}

// ---------------------------------------------------------------- shading
inline V3 reflect_ray(V3 ray, V3 n) {                                   // scene.hpp:185-187
    return sub(muls(smul(2, n), dot(n, ray)), ray);
}

double light_term(const ora_scene &s, double intensity, V3 L, double tmax, V3 p, V3 n, V3 v, int spec,
                  bool use_bvh, Counters &c) {                          // scene.hpp:147-183
    double total = 0.0;
    c.shadow++;
    Hit sh = query(s, p, L, 0.001, tmax, use_bvh, c);
    // census only: would the light add anything if it were visible?  (the GPU path's pruned walk does not ask the
    // query when it would not; the test of that count reads these two)
    const bool idle = !(dot(n, L) > 0.0) && (spec == -1 || !(dot(reflect_ray(L, n), v) > 0));
    if (idle) { c.shadow_idle++; if (sh.any) c.shadow_idle_occluded++; }
    if (sh.any) { c.occluded++; return total; }
    double nl = dot(n, L);
    if (nl > 0.0) {
        double cosv = nl / (mag(n) * mag(L));
        total += intensity * cosv;
    }
    if (spec != -1) {
        V3 R = reflect_ray(L, n);
        double rv = dot(R, v);
        if (rv > 0) {
            double cosv = rv / (mag(R) * mag(v));
            total += intensity * std::pow(cosv, static_cast<double>(spec));
        }
    }
    return total;
}

double total_light(const ora_scene &s, V3 p, V3 n, V3 v, int spec, bool use_bvh, Counters &c) {  // scene.hpp:125-145
    double total = s.ambient;
    for (const Light &l : s.point) total += light_term(s, l.intensity, sub(l.v, p), 1, p, n, v, spec, use_bvh, c);
    for (const Light &l : s.directional) total += light_term(s, l.intensity, l.v, INF, p, n, v, spec, use_bvh, c);
    return total;
}

struct RGB { int r, g, b; };
// include/color.hpp:47-53: each channel truncated toward zero separately
// color.hpp:47-53 converts with static_cast<int>(double).  For a product that does not fit an int
// (a very bright light) or is NaN the language leaves the result open; the reference as built for
// x86-64 gets cvttsd2si's answer, INT_MIN, which the clamp of saveCanvas then turns into 0.  Spelled
// out, so that the restatement does not depend on how a compiler treats the overflow.
inline int to_int(double v) {
    return (v < 2147483648.0 && v > -2147483649.0) ? static_cast<int>(v) : INT32_MIN;
}
inline RGB cmul(RGB c, double k) {
    return {to_int(c.r * k), to_int(c.g * k), to_int(c.b * k)};
}
// int + int of color.hpp:31-37; sums past the int range wrap on x86-64
inline int wrap_add(int a, int b) { return static_cast<int>(static_cast<uint32_t>(a) + static_cast<uint32_t>(b)); }
inline RGB cadd(RGB a, RGB b) { return {wrap_add(a.r, b.r), wrap_add(a.g, b.g), wrap_add(a.b, b.b)}; }

RGB trace(const ora_scene &s, V3 o, V3 r, double tmin, double tmax, int depth, bool use_bvh, Counters &c,
          Hit *first) {                                                  // scene.hpp:89-123
    Hit h = query(s, o, r, tmin, tmax, use_bvh, c);
    if (first) *first = h;
    if (!h.any) return {s.bg[0], s.bg[1], s.bg[2]};
    c.closest_hits++;
    V3 p = add(o, smul(h.t, r));
    const Tri &tr = s.tris[h.id];
    RGB local = cmul({tr.color[0], tr.color[1], tr.color[2]},
                     total_light(s, p, tr.normal, neg(r), tr.specularity, use_bvh, c));
    if (depth == 0 || tr.reflectivity < 0) return local;
    V3 refl = reflect_ray(neg(r), tr.normal);
    c.reflection++;
    RGB rc = trace(s, p, refl, 0.001, INF, depth - 1, use_bvh, c, nullptr);
    return cadd(cmul(local, 1 - tr.reflectivity), cmul(rc, tr.reflectivity));
}

inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }     // color.hpp:23-29

struct Job {
    ora_scene *s;
    int W, H, depth, spp;
    uint64_t seed;
    bool use_bvh;
    int32_t *hit, *rgb_i32;
    uint8_t *rgb_u8;
    double *t;
};

// src/main.cpp:75-94: x outer, y inner; row = -y + H/2 - 1, col = x + W/2
void render_strip(const Job &j, int x0, int x1, Counters *out) {
    Counters c;
    const ora_scene &s = *j.s;
    for (int x = x0; x < x1; x++) {
        for (int y = -(j.H - j.H / 2); y < j.H / 2; y++) {
            size_t row = static_cast<size_t>(-y + j.H / 2 - 1), col = static_cast<size_t>(x + j.W / 2);
            size_t k = row * static_cast<size_t>(j.W) + col;
            int sum[3] = {0, 0, 0};
            RGB c0{0, 0, 0};
            for (int sidx = 0; sidx < j.spp; sidx++) {
                V3 vp;
                if (sidx == 0) {
                    // scene.hpp:81-87 with int x, y
                    vp = {x * (s.viewport.x / j.W), y * (s.viewport.y / j.H), s.viewport.z};
                } else {
                    double jx, jy;
                    ora_jitter(j.seed, k, static_cast<uint32_t>(sidx), &jx, &jy);
                    vp = {(x + jx) * (s.viewport.x / j.W), (y + jy) * (s.viewport.y / j.H), s.viewport.z};
                }
                V3 ray = rot_z(rot_y(rot_x(vp, s.cam_rot.x), s.cam_rot.y), s.cam_rot.z);
                Hit first;
                c.primary++;
                RGB col3 = trace(s, s.cam_pos, ray, 1.0, INF, j.depth, j.use_bvh, c, &first);
                if (sidx == 0) {
                    c0 = col3;
                    if (j.hit) j.hit[k] = first.id;
                    if (j.t) j.t[k] = first.t;
                }
                sum[0] += clamp255(col3.r); sum[1] += clamp255(col3.g); sum[2] += clamp255(col3.b);
            }
            if (j.rgb_i32) {
                if (j.spp == 1) { j.rgb_i32[3 * k] = c0.r; j.rgb_i32[3 * k + 1] = c0.g; j.rgb_i32[3 * k + 2] = c0.b; }
                else { j.rgb_i32[3 * k] = sum[0]; j.rgb_i32[3 * k + 1] = sum[1]; j.rgb_i32[3 * k + 2] = sum[2]; }
            }
            if (j.rgb_u8) {
                for (int ch = 0; ch < 3; ch++)
                    j.rgb_u8[3 * k + ch] = static_cast<uint8_t>((sum[ch] + j.spp / 2) / j.spp);
            }
        }
    }
    *out = c;
}

bool read_file(const char *path, std::string *out) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
    fclose(f);
    return true;
}

}  // namespace

extern "C" {

const char *ora_last_error(void) { return g_err.c_str(); }

uint64_t ora_fnv1a(const void *data, uint64_t n) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

void ora_jitter(uint64_t seed, uint64_t k, uint32_t s, double *jx, double *jy) {
    if (s == 0) { *jx = 0.0; *jy = 0.0; return; }
    uint64_t z = splitmix64(splitmix64(seed) ^ k);
    z = splitmix64(z ^ (static_cast<uint64_t>(s) << 32));
    *jx = static_cast<double>(z >> 32) * (1.0 / 4294967296.0);
    *jy = static_cast<double>(z & 0xFFFFFFFFull) * (1.0 / 4294967296.0);
}

ora_scene *ora_scene_create(int64_t n, const double *v9, const int32_t *color3, const int32_t *spec,
                            const double *refl, const double viewport[3], const double cam_pos[3],
                            const double cam_rot[3], const int32_t bg[3], double ambient, int32_t n_point,
                            const double *point4, int32_t n_dir, const double *dir4) {
    if (n <= 0) { g_err = "scene needs at least one triangle"; return nullptr; }
    ora_scene *s = new ora_scene;
    s->viewport = {viewport[0], viewport[1], viewport[2]};
    s->cam_pos = {cam_pos[0], cam_pos[1], cam_pos[2]};
    s->cam_rot = {cam_rot[0], cam_rot[1], cam_rot[2]};
    for (int i = 0; i < 3; i++) s->bg[i] = bg[i];
    s->ambient = ambient;
    for (int i = 0; i < n_point; i++) s->point.push_back({point4[4 * i], {point4[4 * i + 1], point4[4 * i + 2], point4[4 * i + 3]}});
    for (int i = 0; i < n_dir; i++) s->directional.push_back({dir4[4 * i], {dir4[4 * i + 1], dir4[4 * i + 2], dir4[4 * i + 3]}});
    s->tris.reserve(n);
    for (int64_t i = 0; i < n; i++) {
        const double *p = v9 + 9 * i;
        s->tris.push_back(make_tri({p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}, color3 + 3 * i,
                                   spec[i], refl[i]));
    }
    build_bvh(*s);
    return s;
}

// parseOBJ, include/scene.hpp:213-253.  The reference loops `while (!fs.eof())`,
// reading one non-blank char as the record id, then three doubles ('v') or three
// ints ('f').  When the file ends in white space the last `fs >> id` fails and
// leaves id and the operands as the previous iteration left them, so the last
// record is pushed a second time (observed with the reference built here:
// teapot.obj + "\n" loads 6331 triangles).  That is kept.
ora_scene *ora_scene_load_obj(const char *obj_path, const char *preset) {
    struct Preset { const char *name; double cx, cy, cz; };
    // include/scene.hpp:259,298,337,376,415
    static const Preset presets[] = {{"teapot", -0.015, 4, -11.99},
                                     {"bunny", 0.1425, 2, -5.94},
                                     {"erato", -0.8, 28.7, -86.1},
                                     {"dragon", -0.0425, 0.7115, -3.01725},
                                     {"aurelius", -0.025, 3.05, -20.9}};
    const Preset *pr = nullptr;
    for (const Preset &p : presets) if (strcmp(p.name, preset) == 0) pr = &p;
    if (!pr) { g_err = std::string("unknown preset ") + preset; return nullptr; }
    std::string text;
    if (!read_file(obj_path, &text)) { g_err = "Failed to open file.\n"; return nullptr; }

    std::vector<V3> verts;
    std::vector<double> v9;
    const char *p = text.c_str(), *end = p + text.size();
    char id = 0;
    double xyz[3] = {0, 0, 0};
    long ijk[3] = {0, 0, 0};
    bool at_eof = false;
    while (!at_eof) {
        while (p < end && isspace(static_cast<unsigned char>(*p))) p++;
        bool have_id = p < end;
        if (have_id) id = *p++;
        else at_eof = true;                      // id and operands keep their old values
        if (id == 'v') {
            if (have_id) {
                for (int k = 0; k < 3; k++) {
                    char *q;
                    xyz[k] = strtod(p, &q);
                    if (q == p) { g_err = "unsupported OBJ dialect (vertex record)"; return nullptr; }
                    p = q;
                }
                if (p >= end) at_eof = true;     // the number ran into end-of-file: eofbit is set
            }
            verts.push_back({xyz[0], xyz[1], xyz[2]});
        } else if (id == 'f') {
            if (have_id) {
                for (int k = 0; k < 3; k++) {
                    char *q;
                    ijk[k] = strtol(p, &q, 10);
                    if (q == p) { g_err = "unsupported OBJ dialect (face record)"; return nullptr; }
                    p = q;
                }
                if (p >= end) at_eof = true;
            }
            for (int k = 0; k < 3; k++) {
                if (ijk[k] < 1 || static_cast<size_t>(ijk[k]) > verts.size()) { g_err = "face index out of range"; return nullptr; }
                V3 v = verts[ijk[k] - 1];
                v9.push_back(v.x); v9.push_back(v.y); v9.push_back(v.z);
            }
        }
    }
    size_t n = v9.size() / 9;
    if (n < 10) { g_err = "scene factories need at least 10 triangles"; return nullptr; }
    // scene.hpp:240-242 then the factory overrides, scene.hpp:273-289
    std::vector<int32_t> col(3 * n, 255), spec(n, 1);
    std::vector<double> refl(n, -1.0);
    for (size_t i = 0; i < n - 10; i++) refl[i] = 0.2;
    for (size_t i = n - 10; i < n - 4; i++) refl[i] = 0.4;
    for (size_t i = n - 4; i < n - 2; i++) { col[3 * i] = 0; col[3 * i + 1] = 255; col[3 * i + 2] = 0; refl[i] = 0.4; }
    for (size_t i = n - 2; i < n; i++) { col[3 * i] = 255; col[3 * i + 1] = 0; col[3 * i + 2] = 0; refl[i] = 0.4; }
    const double viewport[3] = {1, 1, 1}, cam[3] = {pr->cx, pr->cy, pr->cz}, rot[3] = {0, 0.1, 0};
    const int32_t bg[3] = {255, 255, 255};
    const double dirl[4] = {0.5, 0, 1, -1};
    return ora_scene_create(static_cast<int64_t>(n), v9.data(), col.data(), spec.data(), refl.data(), viewport,
                            cam, rot, bg, 0.2, 0, nullptr, 1, dirl);
}

void ora_scene_free(ora_scene *s) { delete s; }
int64_t ora_num_triangles(const ora_scene *s) { return static_cast<int64_t>(s->tris.size()); }
int64_t ora_num_nodes(const ora_scene *s) { return static_cast<int64_t>(s->nodes_used); }
int32_t ora_max_depth(const ora_scene *s) { return s->max_depth; }
void ora_get_nodes(const ora_scene *s, void *out) { memcpy(out, s->nodes.data(), s->nodes_used * sizeof(Node)); }
void ora_get_indices(const ora_scene *s, uint64_t *out) { memcpy(out, s->indices.data(), s->indices.size() * 8); }
void ora_get_geometry(const ora_scene *s, double *out) {
    for (const Tri &t : s->tris) {
        const V3 v[5] = {t.v0, t.v1, t.v2, t.normal, t.centroid};
        for (const V3 &a : v) { *out++ = a.x; *out++ = a.y; *out++ = a.z; }
    }
}
void ora_get_materials(const ora_scene *s, int32_t *color3, int32_t *spec, double *refl) {
    for (size_t i = 0; i < s->tris.size(); i++) {
        for (int k = 0; k < 3; k++) color3[3 * i + k] = s->tris[i].color[k];
        spec[i] = s->tris[i].specularity;
        refl[i] = s->tris[i].reflectivity;
    }
}
void ora_get_camera(const ora_scene *s, double *o, int32_t *bg) {
    const V3 v[3] = {s->viewport, s->cam_pos, s->cam_rot};
    for (const V3 &a : v) { *o++ = a.x; *o++ = a.y; *o++ = a.z; }
    *o = s->ambient;
    for (int k = 0; k < 3; k++) bg[k] = s->bg[k];
}

int ora_render(ora_scene *s, int32_t W, int32_t H, int32_t depth, int32_t spp, uint64_t seed, int32_t threads,
               int32_t use_bvh, int32_t *hit, int32_t *rgb_i32, uint8_t *rgb_u8, double *t, ora_census *census) {
    if (W < 0 || H < 0 || depth < 0 || spp < 1 || threads < 1) { g_err = "bad render arguments"; return -1; }
    Job j{s, W, H, depth, spp, seed, use_bvh != 0, hit, rgb_i32, rgb_u8, t};
    std::vector<Counters> cs(threads);
    if (W > 0 && H > 0) {
        std::vector<std::thread> pool;
        int start = -W / 2, step = W / threads;           // src/main.cpp:37-38
        for (int i = 0; i < threads; i++) {
            int x0 = start + i * step;
            int x1 = (i == threads - 1) ? (W - W / 2) : x0 + step;
            pool.emplace_back(render_strip, std::cref(j), x0, x1, &cs[i]);
        }
        for (auto &th : pool) th.join();
    }
    if (census) {
        memset(census, 0, sizeof *census);
        for (const Counters &c : cs) {
            census->primary_rays += c.primary; census->reflection_rays += c.reflection;
            census->shadow_rays += c.shadow; census->shadow_occluded += c.occluded;
            census->closest_hits += c.closest_hits; census->node_visits += c.nodes;
            census->box_hits += c.box_hits; census->leaf_visits += c.leaves; census->triangle_tests += c.tris;
            census->shadow_idle += c.shadow_idle; census->shadow_idle_occluded += c.shadow_idle_occluded;
        }
    }
    return 0;
}

int ora_closest_hit(ora_scene *s, const double o[3], const double r[3], double tmin, double tmax, int32_t use_bvh,
                    double *t_out, int32_t *id_out) {
    Counters c;
    Hit h = query(*s, {o[0], o[1], o[2]}, {r[0], r[1], r[2]}, tmin, tmax, use_bvh != 0, c);
    if (t_out) *t_out = h.t;
    if (id_out) *id_out = h.id;
    return h.any ? 1 : 0;
}

}  // extern "C"
