// The C ABI declared in include/rtx.h: host scenes, flattening to the device layout,
// render launches, tile re-assembly and PPM output.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bvh_build.h"
#include "device_layout.h"
#include "host_scene.hpp"
#include "rtx.h"
#include "rtx_kernels.h"

static_assert(RTX_LAYOUT_PACKED_TILES == rtx::RTX_LAYOUT_PACKED_TILES_, "layout constant");
static_assert(sizeof(rtx_bvh_node) == 64, "rtx_bvh_node must match the reference's BVHNode");

struct rtx_host_scene {
    rtx::HostScene s;
};

struct rtx_scene {
    int device = 0;
    rtx::DevScene dev{};
    void *d_nodes = nullptr, *d_tris = nullptr, *d_mats = nullptr, *d_lights = nullptr, *d_top = nullptr;
    // Per-lane scratch of the render kernel, kept between frames and grown on demand.  kSlots
    // copies, used in turn, let that many frames be in flight on different streams (the end of a
    // frame, when lanes run dry, then overlaps the start of the next one); slot_done[k] orders a
    // launch behind the previous user of its slot.
    static constexpr int kSlots = 8;
    void *d_scratch[kSlots] = {};
    uint64_t scratch_bytes[kSlots] = {};
    hipEvent_t slot_done[kSlots] = {};
    uint32_t next_slot = 0;
    std::mutex slot_lock;
    // launch geometry whose work-queue table each slot's scratch holds (see tile_order_kernel)
    std::vector<int64_t> slot_key[kSlots];
    // record arrays of the staged pipeline (a pixel's state between the launches of one frame), per slot, grown on demand
    void *d_records[kSlots] = {};
    uint64_t record_bytes[kSlots] = {};
    bool records_dirty[kSlots] = {};
    std::vector<int64_t> mask_key[kSlots];
    // rtx_render()'s output buffers (device, and pinned host memory for the way back) and events, kept between calls
    void *d_out_rgb = nullptr, *d_out_hit = nullptr, *d_out_stats = nullptr, *h_out_rgb = nullptr, *h_out_hit = nullptr;
    uint64_t out_rgb_bytes = 0, out_hit_bytes = 0;
    hipEvent_t out_e0 = nullptr, out_e1 = nullptr;
    static constexpr int kOutPieces = 4;          // rtx_render() brings a frame back in this many pieces
    hipEvent_t out_piece[kOutPieces + 1] = {};
    // rtx_render_begin() / rtx_render_end(): frames on their way to host memory, each slot with its own stream,
    // device buffers and pinned host buffers
    struct HostSlot {
        hipStream_t stream = nullptr;
        void *d_rgb = nullptr, *h_rgb = nullptr, *d_hit = nullptr, *h_hit = nullptr;
        uint64_t rgb_bytes = 0, hit_bytes = 0, pixels = 0;
        bool want_hit = false, busy = false;
    };
    static constexpr int kHostSlots = 4;
    HostSlot host_slots[kHostSlots];
    uint64_t bytes = 0;
    uint64_t n_triangles = 0;
    double build_seconds = 0.0;   // BVH build inside rtx_scene_create*, 0 when the caller brought the tree
    int cu_count = 256;
    std::mutex lock;
};

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(RTX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// No exception leaves the library: the body of an entry point that allocates runs in here.
template <typename F>
int guarded(F &&body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(RTX_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(RTX_ERR_INVALID, std::string("unexpected failure: ") + e.what());
    } catch (...) {
        return fail(RTX_ERR_INVALID, "unexpected failure");
    }
}

// RTX_TIMING=1: stage times of scene creation on stderr.
struct StageTimer {
    const bool on = getenv("RTX_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rtx timing] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// Selects a device for the current scope and puts the previous one back.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (dev != prev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int build_into_host_scene(rtx::HostScene &s, int32_t builder, int32_t device, std::string *err) {
    if (builder == RTX_BUILDER_HOST) return rtx::build_bvh(&s, err);
    if (builder != RTX_BUILDER_DEVICE) { *err = "unknown BVH builder"; return RTX_ERR_INVALID; }
    const uint64_t n = s.n_triangles();
    if (n == 0) { *err = "cannot build a BVH over zero triangles"; return RTX_ERR_INVALID; }
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { *err = "no HIP device"; return RTX_ERR_DEVICE; }
    DeviceGuard guard(dev);
    if (!guard.ok) { *err = "cannot select the HIP device"; return RTX_ERR_DEVICE; }
    s.nodes.assign(2 * n - 1, rtx_bvh_node{{0, 0, 0}, {0, 0, 0}, 0, 0});
    s.indices.assign(n, 0);
    uint64_t used = 0;
    double ms = 0;
    if (int rc = rtx::build_bvh_device(s.vertices.data(), n, s.nodes.data(), s.indices.data(), &used, &s.bvh_depth, &ms, err)) return rc;
    s.nodes.resize(used);
    s.bvh_seconds = ms * 1e-3;
    return RTX_OK;
}

int finish_host_scene(rtx_host_scene *hs, bool have_normals, bool have_bvh, int verbose, int32_t builder = RTX_BUILDER_HOST,
                      int32_t device = -1) {
    rtx::HostScene &s = hs->s;
    rtx::setup_triangles(&s, have_normals);
    std::string err;
    if (have_bvh) {
        if (int rc = rtx::validate_bvh(&s, &err)) return fail(rc, err);
        s.bvh_depth = rtx::measure_bvh_depth(s.nodes);
    } else if (builder >= 0) {
        if (int rc = build_into_host_scene(s, builder, device, &err)) return fail(rc, err);
        // Scene::Scene prints the build time in seconds (scene.hpp:49-51)
        if (verbose) printf("[ BVH ] %g seconds.\n", std::floor(s.bvh_seconds * 1000.0) / 1000.0);
    }
    return RTX_OK;
}

// Camera, lights and the other per-scene constants of a device scene whose node,
// triangle and material arrays are in place.
struct SceneCommon {
    const double *viewport, *camera_position, *camera_rotation_deg;
    const int32_t *background;
    double ambient;
    const rtx_light *point_lights;
    uint32_t n_point;
    const rtx_light *directional_lights;
    uint32_t n_directional;
};

int finish_device_scene(rtx_scene *sc, const SceneCommon &c, uint64_t n, uint64_t m, const double root_box[6], bool boxes_tame,
                        int32_t bvh_depth) {
    sc->n_triangles = n;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, sc->device) == hipSuccess && cus > 0) sc->cu_count = cus;
    std::vector<rtx::DevLight> lights;
    for (uint32_t i = 0; i < c.n_point; i++) lights.push_back({c.point_lights[i].intensity, {c.point_lights[i].v[0], c.point_lights[i].v[1], c.point_lights[i].v[2]}});
    for (uint32_t i = 0; i < c.n_directional; i++) lights.push_back({c.directional_lights[i].intensity, {c.directional_lights[i].v[0], c.directional_lights[i].v[1], c.directional_lights[i].v[2]}});
    const size_t light_bytes = lights.size() * sizeof(rtx::DevLight);
    HIP_TRY(hipMalloc(&sc->d_lights, light_bytes ? light_bytes : 16));
    if (light_bytes) HIP_TRY(hipMemcpy(sc->d_lights, lights.data(), light_bytes, hipMemcpyHostToDevice));
    sc->bytes += light_bytes;
    rtx::DevScene &d = sc->dev;
    d.nodes = (const rtx::DevNode *)sc->d_nodes;
    d.tris = (const rtx::DevTri *)sc->d_tris;
    d.mats = (const rtx::DevMat *)sc->d_mats;
    d.lights = (const rtx::DevLight *)sc->d_lights;
    d.n_nodes = (uint32_t)(m + 1);
    d.n_tris = (uint32_t)n;
    d.n_point = c.n_point;
    d.n_dir = c.n_directional;
    double scale = 0;
    for (int k = 0; k < 3; k++) {
        d.cam[k] = c.camera_position[k];
        d.viewport[k] = c.viewport[k];
        d.bg[k] = c.background[k];
        const double a = std::fabs(root_box[k]), b = std::fabs(root_box[3 + k]);
        if (a > scale) scale = a;
        if (b > scale) scale = b;
    }
    d.scene_scale = scale;
    // see RayDiv in rtx_kernels.hip: the slab test may replace its divisions only while
    // every coordinate it subtracts is 0 or well inside the exponent range, and boxes are
    // not inverted
    auto tame = [](double v) { return v == 0.0 || (std::fabs(v) >= 0x1p-300 && std::fabs(v) <= 0x1p300); };
    bool fast_ok = boxes_tame;
    for (int k = 0; k < 3; k++) fast_ok = fast_ok && tame(c.camera_position[k]);
    d.fast_div_ok = fast_ok ? 1 : 0;
    // Vector3::rotateX/Y/Z call cos(radians) and sin(radians) with
    // radians = (degrees / 180.0) * M_PI (util.hpp:12-14).  Built the reference's way (g++ -O3,
    // Makefile:7) each of those functions makes ONE sincos() call for the pair, and glibc's sincos
    // does not always return sin()'s last bit (-29.266614452137418 degrees: ...3194 against
    // ...3199), so sincos it is: same libm entry point, same bits.
    const double rx = (c.camera_rotation_deg[0] / 180.0) * M_PI, ry = (c.camera_rotation_deg[1] / 180.0) * M_PI,
                 rz = (c.camera_rotation_deg[2] / 180.0) * M_PI;
    sincos(rx, &d.sx, &d.cx);
    sincos(ry, &d.sy, &d.cy);
    sincos(rz, &d.sz, &d.cz);
    d.ambient = c.ambient;
    d.bvh_depth = bvh_depth;
    // The top of the tree for the render kernel's LDS: as many sibling pairs as a block has room for
    // behind its traversal stacks (block size as plan_render chooses it).
    uint32_t block = 256;
    while (block > 64 && rtx::render_lds_bytes(bvh_depth, block) > 65536) block >>= 1;
    const char *cap = getenv("RTX_TOP_PAIRS");          // experiments: fewer pairs, 0 = none
    uint32_t pairs = rtx::render_top_pairs(bvh_depth, block, rtx::render_blocks_per_cu());
    if (cap && *cap) pairs = std::min<uint32_t>(pairs, (uint32_t)atoi(cap));
    d.top = nullptr;
    d.n_top_pairs = 0;
    if (pairs) {
        HIP_TRY(hipMalloc(&sc->d_top, (size_t)pairs * 128 + 16));
        uint32_t *d_count = (uint32_t *)((char *)sc->d_top + (size_t)pairs * 128);
        HIP_TRY(rtx::launch_top_table(d.nodes, (rtx::DevNode *)sc->d_top, pairs, d_count, nullptr));
        uint32_t n = 0;
        HIP_TRY(hipMemcpy(&n, d_count, sizeof n, hipMemcpyDeviceToHost));
        d.top = (const rtx::DevNode *)sc->d_top;
        d.n_top_pairs = n;
        sc->bytes += (uint64_t)pairs * 128;
    }
    return RTX_OK;
}

int32_t loader_builder(uint32_t flags) {
    if (flags & RTX_LOAD_NO_BVH) return -1;
    return (flags & RTX_LOAD_DEVICE_BVH) ? RTX_BUILDER_DEVICE : RTX_BUILDER_HOST;
}

struct ResolvedParams {
    rtx::DevRender r{};
};

int resolve_params(const rtx_render_params *p, ResolvedParams *out) {
    if (!p) return fail(RTX_ERR_INVALID, "render params are NULL");
    if (p->width < 0 || p->height < 0) return fail(RTX_ERR_INVALID, "negative canvas size");
    // the kernel keeps a pixel's column and row in 16 bits each
    if (p->width > 65535 || p->height > 65535) return fail(RTX_ERR_UNSUPPORTED, "canvas larger than 65535 pixels in one dimension");
    if (p->depth < 0 || p->depth > 7) return fail(RTX_ERR_INVALID, "depth must be 0..7");
    if (p->spp < 1 || p->spp > 4096) return fail(RTX_ERR_INVALID, "spp must be 1..4096");
    if (p->traversal != RTX_TRAVERSAL_REFERENCE && p->traversal != RTX_TRAVERSAL_PRUNED) return fail(RTX_ERR_INVALID, "unknown traversal mode");
    if (p->layout != RTX_LAYOUT_ROW_MAJOR && p->layout != RTX_LAYOUT_PACKED_TILES) return fail(RTX_ERR_INVALID, "unknown layout");
    rtx::DevRender &r = out->r;
    r.W = p->width; r.H = p->height;
    r.x0 = p->x0; r.x1 = p->x1; r.y0 = p->y0; r.y1 = p->y1;
    if (r.x0 == 0 && r.x1 == 0 && r.y0 == 0 && r.y1 == 0) { r.x1 = r.W; r.y1 = r.H; }
    if (r.x0 < 0 || r.y0 < 0 || r.x1 > r.W || r.y1 > r.H || r.x0 > r.x1 || r.y0 > r.y1) return fail(RTX_ERR_INVALID, "pixel window outside the canvas");
    r.depth = p->depth; r.spp = p->spp; r.seed = p->seed;
    r.tile_w = p->tile_w > 0 ? p->tile_w : 32;
    r.tile_h = p->tile_h > 0 ? p->tile_h : 32;
    if (r.tile_w % 8 || r.tile_h % 8 || r.tile_w > 4096 || r.tile_h > 4096) return fail(RTX_ERR_INVALID, "tile size must be a multiple of 8, at most 4096");
    r.tile_count = p->tile_count > 1 ? p->tile_count : 1;
    r.tile_rank = r.tile_count > 1 ? p->tile_rank : 0;
    if (r.tile_rank < 0 || r.tile_rank >= r.tile_count) return fail(RTX_ERR_INVALID, "tile_rank outside [0, tile_count)");
    r.tiles_x = (r.W + r.tile_w - 1) / r.tile_w;
    r.tiles_y = (r.H + r.tile_h - 1) / r.tile_h;
    const int64_t tiles = (int64_t)r.tiles_x * r.tiles_y;
    // tiles k = rank, rank + count, ...
    r.owned_tiles = tiles > r.tile_rank ? (int32_t)((tiles - r.tile_rank + r.tile_count - 1) / r.tile_count) : 0;
    r.layout = p->layout;
    return RTX_OK;
}

// Equal for every rank, so that the gathered buffer is [rank][max_owned][tile_h][tile_w].
uint64_t packed_tiles_per_rank(const rtx::DevRender &r) {
    const int64_t tiles = (int64_t)r.tiles_x * r.tiles_y;
    return (uint64_t)((tiles + r.tile_count - 1) / r.tile_count);
}

}  // namespace

extern "C" {

const char *rtx_last_error(void) { return g_error.c_str(); }
// for the library's other translation units (rtx_group.cpp): the calling thread's error text
void rtx_set_last_error_(const char *message) { g_error = message ? message : ""; }
#ifndef RTX_KERNEL_HASH
#define RTX_KERNEL_HASH "unknown"
#endif
// the last word is a hash of the render kernels' sources (see the Makefile)
const char *rtx_version(void) { return "rtx-mi355x 0.3 (gfx950) kernels " RTX_KERNEL_HASH; }

// ------------------------------------------------------------------ host scenes
static int host_scene_load_obj_ex_impl(const char *obj_path, const char *preset, uint32_t flags, int32_t device, rtx_host_scene **out) {
    if (!obj_path || !preset || !out) return fail(RTX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    const int verbose = (flags & RTX_LOAD_VERBOSE) != 0;
    std::unique_ptr<rtx_host_scene> hs(new rtx_host_scene);
    std::string err;
    int rc = (flags & RTX_LOAD_TOLERANT) ? rtx::parse_obj_tolerant(obj_path, &hs->s.vertices, &err)
                                         : rtx::parse_obj(obj_path, &hs->s.vertices, &err);
    if (rc == RTX_OK && (flags & RTX_LOAD_TOLERANT) && (flags & RTX_LOAD_LAST_FACE_TWICE) && hs->s.vertices.size() >= 9) {
        const size_t at = hs->s.vertices.size() - 9;
        for (size_t k = 0; k < 9; k++) hs->s.vertices.push_back(hs->s.vertices[at + k]);
    }
    // Scene::parseOBJ prints the face count (scene.hpp:248)
    if (rc == RTX_OK && verbose) printf("[ TRI ] %zu triangles.\n", hs->s.vertices.size() / 9);
    if (rc == RTX_OK) rc = rtx::apply_preset(&hs->s, preset, &err);
    if (rc != RTX_OK) return fail(rc, err);
    rc = finish_host_scene(hs.get(), false, false, verbose, loader_builder(flags), device);
    if (rc != RTX_OK) return rc;
    *out = hs.release();
    return RTX_OK;
}

int rtx_host_scene_load_obj(const char *obj_path, const char *preset, int verbose, rtx_host_scene **out) {
    return rtx_host_scene_load_obj_ex(obj_path, preset, verbose ? RTX_LOAD_VERBOSE : 0, -1, out);
}

static int host_scene_standin_ex_impl(const char *preset, uint64_t n_triangles, uint64_t seed, const char *obj_path_out, uint32_t flags,
                              int32_t device, rtx_host_scene **out) {
    if (!preset) return fail(RTX_ERR_INVALID, "NULL preset");
    if (out) *out = nullptr;
    const int verbose = (flags & RTX_LOAD_VERBOSE) != 0;
    std::vector<double> verts;
    std::vector<uint32_t> faces;
    std::string err;
    if (int rc = rtx::generate_standin(preset, n_triangles, seed, (flags & RTX_STANDIN_KNOT) ? 1 : 0, &verts, &faces, &err)) return fail(rc, err);
    if (obj_path_out)
        if (int rc = rtx::write_obj(obj_path_out, verts, faces, &err)) return fail(rc, err);
    if (!out) return RTX_OK;
    std::unique_ptr<rtx_host_scene> hs(new rtx_host_scene);
    hs->s.vertices.reserve(faces.size() * 3);
    for (uint32_t v : faces) hs->s.vertices.insert(hs->s.vertices.end(), &verts[3 * (size_t)v], &verts[3 * (size_t)v] + 3);
    if (verbose) printf("[ TRI ] %zu triangles.\n", hs->s.vertices.size() / 9);
    int rc = rtx::apply_preset(&hs->s, preset, &err);
    if (rc != RTX_OK) return fail(rc, err);
    rc = finish_host_scene(hs.get(), false, false, verbose, loader_builder(flags), device);
    if (rc != RTX_OK) return rc;
    *out = hs.release();
    return RTX_OK;
}

int rtx_host_scene_standin(const char *preset, uint64_t n_triangles, uint64_t seed, const char *obj_path_out, int verbose,
                           rtx_host_scene **out) {
    return rtx_host_scene_standin_ex(preset, n_triangles, seed, obj_path_out, verbose ? RTX_LOAD_VERBOSE : 0, -1, out);
}

static int host_scene_from_desc_impl(const rtx_scene_desc *d, rtx_host_scene **out) {
    if (!d || !out) return fail(RTX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    const uint64_t n = d->n_triangles;
    if (n == 0) return fail(RTX_ERR_INVALID, "scene has no triangles");
    if (n > 0x7fffffffull) return fail(RTX_ERR_UNSUPPORTED, "more than 2^31-1 triangles");
    if (!d->vertices || !d->colors || !d->specularity || !d->reflectivity) return fail(RTX_ERR_INVALID, "triangle arrays missing");
    if ((d->n_point_lights && !d->point_lights) || (d->n_directional_lights && !d->directional_lights)) return fail(RTX_ERR_INVALID, "light arrays missing");
    const bool have_bvh = d->n_nodes != 0 || d->nodes || d->indices;
    if (have_bvh && (!d->nodes || !d->indices || d->n_nodes == 0 || d->n_nodes > 2 * n - 1)) return fail(RTX_ERR_INVALID, "BVH arrays incomplete");
    std::unique_ptr<rtx_host_scene> hs(new rtx_host_scene);
    rtx::HostScene &s = hs->s;
    s.vertices.assign(d->vertices, d->vertices + 9 * n);
    if (d->normals) s.normals.assign(d->normals, d->normals + 3 * n);
    s.colors.assign(d->colors, d->colors + 3 * n);
    s.specularity.assign(d->specularity, d->specularity + n);
    s.reflectivity.assign(d->reflectivity, d->reflectivity + n);
    if (have_bvh) {
        s.nodes.assign(d->nodes, d->nodes + d->n_nodes);
        s.indices.assign(d->indices, d->indices + n);
    }
    for (int k = 0; k < 3; k++) {
        s.viewport[k] = d->viewport[k];
        s.camera_position[k] = d->camera_position[k];
        s.camera_rotation_deg[k] = d->camera_rotation_deg[k];
        s.background[k] = d->background[k];
    }
    s.ambient = d->ambient;
    s.point_lights.assign(d->point_lights, d->point_lights + d->n_point_lights);
    s.directional_lights.assign(d->directional_lights, d->directional_lights + d->n_directional_lights);
    int rc = finish_host_scene(hs.get(), d->normals != nullptr, have_bvh, 0);
    if (rc != RTX_OK) return rc;
    *out = hs.release();
    return RTX_OK;
}

int rtx_host_scene_desc(const rtx_host_scene *hs, rtx_scene_desc *o) {
    if (!hs || !o) return fail(RTX_ERR_INVALID, "NULL argument");
    const rtx::HostScene &s = hs->s;
    memset(o, 0, sizeof *o);
    o->n_triangles = s.n_triangles();
    o->vertices = s.vertices.data();
    o->normals = s.normals.data();
    o->colors = s.colors.data();
    o->specularity = s.specularity.data();
    o->reflectivity = s.reflectivity.data();
    o->n_nodes = s.nodes.size();
    o->nodes = s.nodes.empty() ? nullptr : s.nodes.data();
    o->indices = s.nodes.empty() ? nullptr : s.indices.data();
    for (int k = 0; k < 3; k++) {
        o->viewport[k] = s.viewport[k];
        o->camera_position[k] = s.camera_position[k];
        o->camera_rotation_deg[k] = s.camera_rotation_deg[k];
        o->background[k] = s.background[k];
    }
    o->ambient = s.ambient;
    o->n_point_lights = (uint32_t)s.point_lights.size();
    o->point_lights = s.point_lights.data();
    o->n_directional_lights = (uint32_t)s.directional_lights.size();
    o->directional_lights = s.directional_lights.data();
    o->device = -1;
    return RTX_OK;
}

int32_t rtx_host_scene_bvh_depth(const rtx_host_scene *hs) { return hs ? hs->s.bvh_depth : -1; }

int rtx_host_scene_set_camera(rtx_host_scene *hs, const double position[3], const double rotation_deg[3]) {
    if (!hs || !position || !rotation_deg) return fail(RTX_ERR_INVALID, "NULL argument");
    for (int k = 0; k < 3; k++) { hs->s.camera_position[k] = position[k]; hs->s.camera_rotation_deg[k] = rotation_deg[k]; }
    return RTX_OK;
}

static int host_scene_set_lights_impl(rtx_host_scene *hs, double ambient, uint32_t n_point, const rtx_light *point,
                              uint32_t n_directional, const rtx_light *directional) {
    if (!hs || (n_point && !point) || (n_directional && !directional)) return fail(RTX_ERR_INVALID, "NULL argument");
    hs->s.ambient = ambient;
    hs->s.point_lights.assign(point, point + n_point);
    hs->s.directional_lights.assign(directional, directional + n_directional);
    return RTX_OK;
}

void rtx_host_scene_free(rtx_host_scene *hs) { delete hs; }

// ------------------------------------------------------------------ BVH build
static int bvh_build_impl(const double *vertices, uint64_t n, int32_t builder, int32_t device, rtx_bvh_node *nodes_out,
                  uint64_t *indices_out, uint64_t *n_nodes_out, int32_t *depth_out, double *seconds_out) {
    if (!vertices || !nodes_out || !indices_out || !n_nodes_out) return fail(RTX_ERR_INVALID, "NULL argument");
    if (n == 0) return fail(RTX_ERR_INVALID, "cannot build a BVH over zero triangles");
    std::string err;
    if (builder == RTX_BUILDER_HOST) {
        rtx::HostScene s;
        s.vertices.assign(vertices, vertices + 9 * n);
        s.colors.assign(3 * n, 0);
        s.specularity.assign(n, -1);
        s.reflectivity.assign(n, -1.0);
        rtx::setup_triangles(&s, false);
        if (int rc = rtx::build_bvh(&s, &err)) return fail(rc, err);
        std::copy(s.nodes.begin(), s.nodes.end(), nodes_out);
        std::copy(s.indices.begin(), s.indices.end(), indices_out);
        *n_nodes_out = s.nodes.size();
        if (depth_out) *depth_out = s.bvh_depth;
        if (seconds_out) *seconds_out = s.bvh_seconds;
        return RTX_OK;
    }
    if (builder != RTX_BUILDER_DEVICE) return fail(RTX_ERR_INVALID, "unknown BVH builder");
    int dev = device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "cannot select the HIP device");
    int32_t depth = 0;
    double ms = 0;
    if (int rc = rtx::build_bvh_device(vertices, n, nodes_out, indices_out, n_nodes_out, &depth, &ms, &err)) return fail(rc, err);
    if (depth_out) *depth_out = depth;
    if (seconds_out) *seconds_out = ms * 1e-3;
    return RTX_OK;
}

static int host_scene_rebuild_bvh_impl(rtx_host_scene *hs, int32_t builder, int32_t device, double *seconds_out) {
    if (!hs) return fail(RTX_ERR_INVALID, "NULL argument");
    std::string err;
    if (int rc = build_into_host_scene(hs->s, builder, device, &err)) return fail(rc, err);
    if (seconds_out) *seconds_out = hs->s.bvh_seconds;
    return RTX_OK;
}

// ------------------------------------------------------------------ device scenes
static int create_scene_built_on_device(const rtx_scene_desc *d, rtx_scene **out);
struct SceneDeleter { void operator()(rtx_scene *sc) const { rtx_scene_destroy(sc); } };

static int scene_create_from_host_impl(const rtx_host_scene *hs, int32_t device, rtx_scene **out) {
    if (!hs || !out) return fail(RTX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    const rtx::HostScene &s = hs->s;
    const uint64_t n = s.n_triangles(), m = s.nodes.size();
    if (n == 0) return fail(RTX_ERR_INVALID, "scene is empty");
    if (m == 0) {
        // loaded with RTX_LOAD_NO_BVH: the tree is built where it is used
        rtx_scene_desc d;
        if (int rc = rtx_host_scene_desc(hs, &d)) return rc;
        d.nodes = nullptr;
        d.indices = nullptr;
        d.device = device;
        return create_scene_built_on_device(&d, out);
    }
    // traversal-stack entries carry 28-bit triangle positions and node slots
    if (m + 1 >= (1ull << 28) || n >= (1ull << 28)) return fail(RTX_ERR_UNSUPPORTED, "more than 2^28 triangles or BVH nodes");
    // the kernel packs the light index of a shadow query into 16 bits
    if (s.point_lights.size() + s.directional_lights.size() > 0xffff) return fail(RTX_ERR_UNSUPPORTED, "more than 65535 lights");

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(RTX_ERR_INVALID, "device ordinal out of range");

    // nodes: slot = reference index + 1; leaf `first` already addresses the re-ordered
    // triangle array because that array IS BVH::indices order
    std::vector<rtx::DevNode> nodes(m + 1);
    memset(nodes.data(), 0, nodes.size() * sizeof(rtx::DevNode));
    // slot 0 pairs with the root (slot 1); NaN bounds fail every slab test, so a walk that
    // starts as the pair test of (0, 1) tests exactly the root
    for (int k = 0; k < 3; k++) { nodes[0].bmin[k] = std::nan(""); nodes[0].bmax[k] = std::nan(""); }
    nodes[0].count = 0; nodes[0].first = 0;
    for (uint64_t i = 0; i < m; i++) {
        const rtx_bvh_node &h = s.nodes[i];
        rtx::DevNode &d = nodes[i + 1];
        for (int k = 0; k < 3; k++) { d.bmin[k] = h.bmin[k]; d.bmax[k] = h.bmax[k]; }
        d.count = (uint32_t)h.count;
        d.first = h.count == 0 ? (uint32_t)(h.first + 1) : (uint32_t)h.first;
    }
    // The left record of a sibling pair holds the references to both nodes of the pair (see DevNode).  A pair is wherever
    // an inner node points: the reference's builder puts every left child at an odd index (even slot), a caller's
    // tree may number its nodes differently, so the pairs are taken from the inner nodes and not from the slot parity.
    nodes[0].pad[0] = 0;                   // the pair (0, 1) a walk starts with
    nodes[0].pad[1] = rtx::node_entry(1u, nodes[1].first, nodes[1].count);
    for (uint64_t i = 0; i < m; i++) {
        if (s.nodes[i].count != 0 || s.nodes[i].first == 0 || s.nodes[i].first + 1 >= m) continue;   // leaf, or a blanked gap
        const uint64_t left = s.nodes[i].first + 1;            // slot of the left child; validate_bvh keeps left + 1 <= m
        nodes[left].pad[0] = rtx::node_entry((uint32_t)left, nodes[left].first, nodes[left].count);
        nodes[left].pad[1] = rtx::node_entry((uint32_t)left + 1u, nodes[left + 1].first, nodes[left + 1].count);
    }
    std::vector<rtx::DevTri> tris(n);
    std::vector<rtx::DevMat> mats(n);
    memset(tris.data(), 0, n * sizeof(rtx::DevTri));
    memset(mats.data(), 0, n * sizeof(rtx::DevMat));
    for (uint64_t pos = 0; pos < n; pos++) {
        const uint64_t id = s.indices[pos];
        rtx::DevTri &t = tris[pos];
        const double *v = &s.vertices[9 * id];
        const double *nn = &s.normals[3 * id];
        for (int k = 0; k < 3; k++) { t.v0[k] = v[k]; t.v1[k] = v[3 + k]; t.v2[k] = v[6 + k]; t.n[k] = nn[k]; }
        // double d = -normal.dot(v0);  (triangle.hpp:39)
        t.d = -(nn[0] * v[0] + nn[1] * v[1] + nn[2] * v[2]);
        t.orig_id = (uint32_t)id;
        rtx::DevMat &mt = mats[pos];
        for (int k = 0; k < 3; k++) mt.color[k] = s.colors[3 * id + k];
        mt.specularity = s.specularity[id];
        mt.reflectivity = s.reflectivity[id];
    }
    for (uint64_t i = 0; i < m; i++) {
        const rtx_bvh_node &h = s.nodes[i];
        for (uint64_t j = 0; j < h.count; j++) tris[h.first + j].leaf_first = (uint32_t)h.first;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "cannot select the device");
    std::unique_ptr<rtx_scene, SceneDeleter> owner(new rtx_scene);
    rtx_scene *sc = owner.get();
    sc->device = device;
    auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
        if (e != hipSuccess) return e;
        sc->bytes += bytes;
        return bytes ? rtx::upload_pageable(*dst, src, bytes, nullptr) : hipSuccess;
    };
    hipError_t e = upload(&sc->d_nodes, nodes.data(), nodes.size() * sizeof(rtx::DevNode));
    if (e == hipSuccess) e = upload(&sc->d_tris, tris.data(), n * sizeof(rtx::DevTri));
    if (e == hipSuccess) e = upload(&sc->d_mats, mats.data(), n * sizeof(rtx::DevMat));
    if (e != hipSuccess) return fail(RTX_ERR_DEVICE, std::string("scene upload: ") + hipGetErrorString(e));
    auto tame = [](double v) { return v == 0.0 || (std::fabs(v) >= 0x1p-300 && std::fabs(v) <= 0x1p300); };
    bool boxes_tame = true;
    for (const rtx_bvh_node &h : s.nodes)
        for (int k = 0; k < 3; k++) boxes_tame = boxes_tame && tame(h.bmin[k]) && tame(h.bmax[k]) && h.bmin[k] <= h.bmax[k];
    const double root_box[6] = {s.nodes[0].bmin[0], s.nodes[0].bmin[1], s.nodes[0].bmin[2],
                                s.nodes[0].bmax[0], s.nodes[0].bmax[1], s.nodes[0].bmax[2]};
    const SceneCommon common{s.viewport, s.camera_position, s.camera_rotation_deg, s.background, s.ambient,
                             s.point_lights.data(), (uint32_t)s.point_lights.size(),
                             s.directional_lights.data(), (uint32_t)s.directional_lights.size()};
    if (int rc = finish_device_scene(sc, common, n, m, root_box, boxes_tame, s.bvh_depth)) return rc;
    *out = owner.release();
    return RTX_OK;
}

// A scene whose BVH is built and flattened on the device: the vertices go up once, the
// tree never comes back to the host.
static int create_scene_built_on_device(const rtx_scene_desc *d, rtx_scene **out) {
    const uint64_t n = d->n_triangles;
    if (n == 0) return fail(RTX_ERR_INVALID, "scene has no triangles");
    if (n >= (1ull << 28)) return fail(RTX_ERR_UNSUPPORTED, "more than 2^28 triangles or BVH nodes");
    if (!d->vertices || !d->colors || !d->specularity || !d->reflectivity) return fail(RTX_ERR_INVALID, "triangle arrays missing");
    if ((d->n_point_lights && !d->point_lights) || (d->n_directional_lights && !d->directional_lights)) return fail(RTX_ERR_INVALID, "light arrays missing");
    if ((uint64_t)d->n_point_lights + d->n_directional_lights > 0xffff) return fail(RTX_ERR_UNSUPPORTED, "more than 65535 lights");
    int ndev = 0, device = d->device;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(RTX_ERR_INVALID, "device ordinal out of range");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "cannot select the device");

    std::string err;
    rtx::DeviceBvh bvh;
    StageTimer timer;
    std::unique_ptr<rtx_scene, SceneDeleter> owner(new rtx_scene);
    rtx_scene *sc = owner.get();
    sc->device = device;
    // The per-triangle arrays the flattening reads (normals if given, colours, exponents, reflectivities) go up on a
    // helper thread and its own stream while this thread builds the tree: the copy engine is idle during the build.
    struct Temp { void *p = nullptr; ~Temp() { (void)hipFree(p); } } t_normals, t_colors, t_spec, t_refl;
    hipError_t staged = hipSuccess;
    std::atomic<int> vertices_up{0};
    std::thread uploader([&]() {
        hipStream_t up = nullptr;
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
        auto stage = [&](Temp &t, const void *src, size_t bytes) {
            if (e == hipSuccess) e = hipMalloc(&t.p, bytes);
            if (e == hipSuccess) e = rtx::upload_pageable(t.p, src, bytes, up);      // waits its turn behind the vertices
        };
        if (e == hipSuccess) e = hipMalloc(&sc->d_tris, n * sizeof(rtx::DevTri));
        if (e == hipSuccess) e = hipMalloc(&sc->d_mats, n * sizeof(rtx::DevMat));
        // the build waits for the vertices, nothing waits for these: sleep through the vertex upload rather than spin on
        // a core that the upload's own copy threads (and, in a scene group, the other devices' uploads) want
        while (!vertices_up.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (d->normals) stage(t_normals, d->normals, 3 * n * sizeof(double));
        stage(t_colors, d->colors, 3 * n * sizeof(int32_t));
        stage(t_spec, d->specularity, n * sizeof(int32_t));
        stage(t_refl, d->reflectivity, n * sizeof(double));
        if (e == hipSuccess) e = hipStreamSynchronize(up);
        if (up) (void)hipStreamDestroy(up);
        staged = e;
    });
    struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_on_exit{uploader};
    const int build_rc = rtx::build_bvh_resident(d->vertices, n, &bvh, &err, &vertices_up);
    uploader.join();
    if (build_rc != RTX_OK) return fail(build_rc, err);
    timer.lap("upload + BVH build");
    const uint64_t m = bvh.n_nodes;
    if (m + 1 >= (1ull << 28)) return fail(RTX_ERR_UNSUPPORTED, "more than 2^28 triangles or BVH nodes");
    hipError_t e = staged;
    if (e == hipSuccess) e = hipMalloc(&sc->d_nodes, (m + 1) * sizeof(rtx::DevNode));
    if (e != hipSuccess) return fail(RTX_ERR_DEVICE, std::string("scene upload: ") + hipGetErrorString(e));
    timer.lap("allocate, materials up");
    sc->bytes = (m + 1) * sizeof(rtx::DevNode) + n * (sizeof(rtx::DevTri) + sizeof(rtx::DevMat));
    int32_t boxes_tame = 0;
    double root_box[6];
    int rc = rtx::flatten_scene_device(bvh, (const double *)t_normals.p, (const int32_t *)t_colors.p, (const int32_t *)t_spec.p,
                                       (const double *)t_refl.p, (rtx::DevNode *)sc->d_nodes, (rtx::DevTri *)sc->d_tris,
                                       (rtx::DevMat *)sc->d_mats, &boxes_tame, root_box, &err);
    if (rc != RTX_OK) return fail(rc, err);
    timer.lap("flatten");
    const SceneCommon common{d->viewport, d->camera_position, d->camera_rotation_deg, d->background, d->ambient,
                             d->point_lights, d->n_point_lights, d->directional_lights, d->n_directional_lights};
    rc = finish_device_scene(sc, common, n, m, root_box, boxes_tame != 0, bvh.depth);
    if (rc != RTX_OK) return rc;
    sc->build_seconds = bvh.device_ms * 1e-3;
    *out = owner.release();
    return RTX_OK;
}

static int scene_create_with_builder_impl(const rtx_scene_desc *desc, int32_t builder, rtx_scene **out) {
    if (!desc || !out) return fail(RTX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    const bool have_bvh = desc->n_nodes != 0 || desc->nodes || desc->indices;
    if (builder != RTX_BUILDER_HOST && builder != RTX_BUILDER_DEVICE) return fail(RTX_ERR_INVALID, "unknown BVH builder");
    if (!have_bvh && builder == RTX_BUILDER_DEVICE) return create_scene_built_on_device(desc, out);
    rtx_host_scene *raw = nullptr;
    if (int rc = rtx_host_scene_from_desc(desc, &raw)) return rc;
    std::unique_ptr<rtx_host_scene> hs(raw);
    int rc = rtx_scene_create_from_host(hs.get(), desc->device, out);
    if (rc == RTX_OK && !have_bvh) (*out)->build_seconds = hs->s.bvh_seconds;
    return rc;
}

int rtx_scene_create(const rtx_scene_desc *desc, rtx_scene **out) {
    return rtx_scene_create_with_builder(desc, RTX_BUILDER_DEVICE, out);
}

double rtx_scene_build_seconds(const rtx_scene *sc) { return sc ? sc->build_seconds : 0.0; }
uint64_t rtx_scene_bvh_nodes(const rtx_scene *sc) { return sc && sc->dev.n_nodes ? sc->dev.n_nodes - 1 : 0; }
int32_t rtx_scene_bvh_depth(const rtx_scene *sc) { return sc ? sc->dev.bvh_depth : -1; }

void rtx_scene_destroy(rtx_scene *sc) {
    if (!sc) return;
    DeviceGuard guard(sc->device);
    (void)hipFree(sc->d_nodes);
    (void)hipFree(sc->d_tris);
    (void)hipFree(sc->d_mats);
    (void)hipFree(sc->d_lights);
    (void)hipFree(sc->d_top);
    for (int k = 0; k < rtx_scene::kSlots; k++) {
        (void)hipFree(sc->d_scratch[k]);
        (void)hipFree(sc->d_records[k]);
        if (sc->slot_done[k]) (void)hipEventDestroy(sc->slot_done[k]);
    }
    (void)hipFree(sc->d_out_rgb);
    (void)hipFree(sc->d_out_hit);
    (void)hipFree(sc->d_out_stats);
    if (sc->h_out_rgb) (void)hipHostFree(sc->h_out_rgb);
    if (sc->h_out_hit) (void)hipHostFree(sc->h_out_hit);
    for (rtx_scene::HostSlot &h : sc->host_slots) {
        if (h.stream) { (void)hipStreamSynchronize(h.stream); (void)hipStreamDestroy(h.stream); }
        (void)hipFree(h.d_rgb);
        (void)hipFree(h.d_hit);
        if (h.h_rgb) (void)hipHostFree(h.h_rgb);
        if (h.h_hit) (void)hipHostFree(h.h_hit);
    }
    if (sc->out_e0) (void)hipEventDestroy(sc->out_e0);
    if (sc->out_e1) (void)hipEventDestroy(sc->out_e1);
    for (hipEvent_t e : sc->out_piece) if (e) (void)hipEventDestroy(e);
    delete sc;
}

uint64_t rtx_scene_device_bytes(const rtx_scene *sc) { return sc ? sc->bytes : 0; }

// ------------------------------------------------------------------ render
void rtx_render_params_default(rtx_render_params *p, int32_t width, int32_t height) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->width = width;
    p->height = height;
    p->depth = 3;              // src/main.cpp:83
    p->spp = 1;
    p->traversal = RTX_TRAVERSAL_PRUNED;      // the same frame as RTX_TRAVERSAL_REFERENCE, bit for bit, in less time
    p->tile_w = 32;
    p->tile_h = 32;
    p->tile_count = 1;
    p->layout = RTX_LAYOUT_ROW_MAJOR;
}

uint64_t rtx_output_pixels(const rtx_render_params *p) {
    ResolvedParams rp;
    if (resolve_params(p, &rp) != RTX_OK) return 0;
    if (rp.r.layout == RTX_LAYOUT_ROW_MAJOR) return (uint64_t)rp.r.W * (uint64_t)rp.r.H;
    return packed_tiles_per_rank(rp.r) * (uint64_t)rp.r.tile_w * (uint64_t)rp.r.tile_h;
}

int rtx_render_device(rtx_scene *sc, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id, void *d_rgb_i32, void *d_t,
                      void *d_stats, void *stream) {
    return rtx_render_device_timed(sc, p, d_rgb_u8, d_hit_id, d_rgb_i32, d_t, d_stats, nullptr, stream);
}

// What a render launch needs besides its scratch memory: the resolved parameters with the output
// pointers and scheduler settings in place, and the launch geometry.
struct LaunchPlan {
    rtx::DevRender r{};
    uint32_t block = 256;        // threads per block (a power of two, 64..256)
    uint64_t grid_blocks = 0;    // blocks of the persistent launch; 0 = nothing to render
    int n_queues = 8, macro = 8; // work distribution (see tile_order_kernel)
    int centre_first = 1;        // a queue serves the macro tiles nearest the frame's centre first
    bool pruned = false;
};

static int plan_render(rtx_scene *sc, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id, void *d_rgb_i32, void *d_t,
                       void *d_stats, void *d_ticks, LaunchPlan *plan) {
    if (!sc) return fail(RTX_ERR_INVALID, "scene is NULL");
    ResolvedParams rp;
    if (int rc = resolve_params(p, &rp)) return rc;
    rtx::DevRender &r = plan->r;
    r = rp.r;
    // what the kernel indexes with must match what it was given
    if (r.layout == RTX_LAYOUT_PACKED_TILES && (d_hit_id || d_t)) return fail(RTX_ERR_INVALID, "hit ids / t are only written in row-major layout");
    r.px = sc->dev.viewport[0] / r.W;     // this->viewport.x / canvasWidth  (scene.hpp:83)
    r.py = sc->dev.viewport[1] / r.H;
    r.rgb = (uint8_t *)d_rgb_u8;
    r.hit = (int32_t *)d_hit_id;
    r.rgb_i32 = (int32_t *)d_rgb_i32;
    r.t = (double *)d_t;
    r.stats = (unsigned long long *)d_stats;
    r.ticks = (unsigned long long *)d_ticks;
    // LDS stack: the largest block whose stack fits 64 KiB
    uint32_t block = 256;
    while (block > 64 && rtx::render_lds_bytes(sc->dev.bvh_depth, block) > 65536) block >>= 1;
    if (rtx::render_lds_bytes(sc->dev.bvh_depth, block) > 65536)
        return fail(RTX_ERR_UNSUPPORTED, "BVH deeper than 255 levels: traversal stack does not fit LDS");
    r.stack_stride = block;
    plan->block = block;
    // Tuning knobs (environment overrides are for experiments).
    auto knob = [](const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; };
    // measured per traversal mode: the pruned walk has shorter runs of node steps
    const bool pruned = p->traversal == RTX_TRAVERSAL_PRUNED;
    plan->pruned = pruned;
    r.w_inner = knob("RTX_W_INNER", pruned ? 3 : 1);
    r.w_leaf = knob("RTX_W_LEAF", pruned ? 4 : 2);
    r.w_shade = knob("RTX_W_SHADE", pruned ? 4 : 2);
    r.tail_lanes = knob("RTX_TAIL_LANES", 16);
    r.also_lanes = knob("RTX_ALSO_LANES", 24);
    r.also_leaf = knob("RTX_ALSO_LEAF", 16);
    r.also_shade = knob("RTX_ALSO_SHADE", r.also_lanes);
    r.lighting_jobs = knob("RTX_LIGHTING_JOBS", 1);
    r.n_frames = 1;
    r.has_cams = 0;
    r.frame_pixels = 0;
    if (r.w_inner < 1 || r.w_leaf < 1 || r.w_shade < 1) return fail(RTX_ERR_INVALID, "scheduler weights must be positive");
    plan->grid_blocks = 0;
    if (r.W == 0 || r.H == 0 || r.owned_tiles == 0) return RTX_OK;
    // Persistent launch: as many blocks as the chip keeps resident (RTX_WAVES_PER_EU waves per
    // SIMD = that many 256-thread blocks per CU), never more than there are 8x8 blocks of work.
    const uint64_t total_blocks = rtx::render_waves(r);
    if (total_blocks >= (1ull << 26)) return fail(RTX_ERR_UNSUPPORTED, "more than 2^32 pixels in one launch");
    r.total_blocks = (uint32_t)total_blocks;
    const uint32_t wpb = block / 64;
    const uint64_t resident = (uint64_t)sc->cu_count * (uint64_t)knob("RTX_BLOCKS_PER_CU", rtx::render_blocks_per_cu()) * (256 / block);
    plan->grid_blocks = std::min<uint64_t>((total_blocks + wpb - 1) / wpb, std::max<uint64_t>(resident, 1));
    // Work queues.  The frame's tiles are grouped into macro tiles (RTX_MACRO x RTX_MACRO tiles of
    // screen neighbours) and the macro tiles dealt to 8 queues, one per XCD; a wave serves its own
    // XCD's queue first.
    plan->n_queues = knob("RTX_XCD_QUEUES", 1) ? 8 : 1;
    plan->macro = std::min(64, std::max(1, knob("RTX_MACRO", 8)));
    plan->centre_first = knob("RTX_CENTRE_FIRST", 1) ? 1 : 0;
    r.n_queues = (uint32_t)plan->n_queues;
    return RTX_OK;
}

// Points the launch at its scratch memory: [work counters][work-queue table][per-lane arrays].
static void bind_scratch(LaunchPlan *plan, void *base) {
    rtx::DevRender &r = plan->r;
    const uint64_t lanes = plan->grid_blocks * plan->block;
    r.work = (uint32_t *)base;
    r.tile_order = (const uint32_t *)((char *)base + rtx::kScratchHeaderBytes);
    r.scratch_lanes = lanes;
    r.scratch_f64 = (double *)((char *)base + rtx::kScratchHeaderBytes + rtx::render_order_bytes(r, plan->macro));
    r.scratch_i32 = (int32_t *)((char *)r.scratch_f64 + lanes * rtx::render_lane_f64s(r.depth) * 8);
}

// Turns a launch plan into one over `n_frames` frames (see DevRender::n_frames): the queues hand out every block once
// per frame, frame f writes f * (pixels of one frame) further on, with its own camera if `cameras` (n_frames x
// {position[3], rotation in degrees[3]}) is given.
static int plan_frames(rtx_scene *sc, const rtx_render_params *p, int32_t n_frames, const double *cameras, LaunchPlan *plan) {
    rtx::DevRender &r = plan->r;
    if (n_frames < 1 || n_frames > 8) return fail(RTX_ERR_INVALID, "a launch renders 1..8 frames");
    if (r.hit || r.t) return fail(RTX_ERR_INVALID, "hit ids / t are written by single-frame launches only");
    const uint64_t frame_pixels = rtx_output_pixels(p);
    if (frame_pixels * (uint64_t)n_frames >= (1ull << 32) || (uint64_t)r.total_blocks * (uint64_t)n_frames >= (1ull << 26))
        return fail(RTX_ERR_UNSUPPORTED, "more than 2^32 pixels in one launch");
    r.n_frames = n_frames;
    r.frame_pixels = (uint32_t)frame_pixels;
    r.has_cams = cameras ? 1 : 0;
    for (int32_t f = 0; cameras && f < n_frames; f++) {
        const double *c = cameras + 6 * f;
        double *o = r.cams[f];
        for (int k = 0; k < 3; k++) o[k] = c[k];
        // one sincos() per axis, as in finish_device_scene (Vector3::rotateX/Y/Z, vector3.hpp:54-82)
        sincos((c[3] / 180.0) * M_PI, &o[4], &o[3]);
        sincos((c[4] / 180.0) * M_PI, &o[6], &o[5]);
        sincos((c[5] / 180.0) * M_PI, &o[8], &o[7]);
    }
    // more work per launch may keep more of the chip busy: the grid was sized for one frame's blocks
    const uint64_t wpb = plan->block / 64;
    const uint64_t resident = (uint64_t)sc->cu_count * (uint64_t)rtx::render_blocks_per_cu() * (256 / plan->block);
    plan->grid_blocks = std::min<uint64_t>(((uint64_t)r.total_blocks * (uint64_t)n_frames + wpb - 1) / wpb, std::max<uint64_t>(resident, plan->grid_blocks));
    return RTX_OK;
}

// The launches of one staged frame (see the staged pipeline in rtx_kernels.hip): launch 0 traces the camera rays, launch l
// (1..depth) the level-l reflections and the lighting of the level l-1 hits, launch depth+1 the lighting of the last level,
// launch depth+2 blends and writes the pixels.  One stream, one after the other.
static int launch_staged_frame(rtx_scene *sc, LaunchPlan &plan, void *records, uint64_t pixels, bool stats, hipStream_t st) {
    rtx::DevRender &r = plan.r;
    const uint64_t levels = (uint64_t)r.depth + 1, P = pixels;
    r.rec_pixels = P;
    r.hitmask = (unsigned long long *)records;
    r.rec_pt = (double *)((char *)records + (rtx::staged_mask_bytes(r.total_blocks, r.depth) + 255) / 256 * 256);
    r.rec_ray = r.rec_pt + 2 * 3 * P;
    r.lvl_refl = r.rec_ray + 2 * 3 * P;
    r.rec_pos = (int32_t *)(r.lvl_refl + levels * P);
    r.lvl_local = r.rec_pos + levels * P;
    const char *run = getenv("RTX_STAGED_RUN");
    r.draw_run = (uint32_t)std::min(64, std::max(1, run && *run ? atoi(run) : 1));
    const char *refill = getenv("RTX_STAGED_REFILL");
    r.refill_min = std::min(64, std::max(1, refill && *refill ? atoi(refill) : 8));
    const char *only = getenv("RTX_STAGED_ONLY");        // experiments: stop after this launch (leaves the masks dirty)
    const int32_t last = only && *only ? std::min(atoi(only), r.depth + 2) : r.depth + 2;
    for (int32_t l = 0; l <= last; l++) {
        r.stage = l;
        if (l == 0) { r.n_passes = 1; r.pass_kind[0] = rtx::SP_PRIMARY; }
        else if (l <= r.depth) { r.n_passes = 2; r.pass_kind[0] = rtx::SP_REFLECT; r.pass_kind[1] = rtx::SP_LIGHT; }
        else if (l == r.depth + 1) { r.n_passes = 1; r.pass_kind[0] = rtx::SP_LIGHT; }
        else { r.n_passes = 1; r.pass_kind[0] = rtx::SP_RESOLVE; }
        if (r.n_passes == 1) r.pass_kind[1] = r.pass_kind[0];
        if ((uint64_t)r.total_blocks * (uint64_t)r.n_passes + 64 >= (1ull << 26)) return fail(RTX_ERR_UNSUPPORTED, "more than 2^32 pixels in one launch");
        HIP_TRY(rtx::launch_render(sc->dev, r, plan.pruned, stats, plan.block, (uint32_t)plan.grid_blocks, st, true));
    }
    return RTX_OK;
}

static int render_device_timed_impl(rtx_scene *sc, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id, void *d_rgb_i32, void *d_t,
                            void *d_stats, void *d_ticks, void *stream, int32_t n_frames = 1, const double *cameras = nullptr,
                            bool staged = false) {
    LaunchPlan plan;
    if (int rc = plan_render(sc, p, d_rgb_u8, d_hit_id, d_rgb_i32, d_t, d_stats, d_ticks, &plan)) return rc;
    if (staged && plan.r.spp != 1) return fail(RTX_ERR_UNSUPPORTED, "the staged pipeline renders one sample per pixel; use rtx_render_device");
    if (plan.grid_blocks == 0) return RTX_OK;
    if (n_frames != 1 || cameras)
        if (int rc = plan_frames(sc, p, n_frames, cameras, &plan)) return rc;
    rtx::DevRender &r = plan.r;
    const hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sc->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "cannot select the scene's device");
    // Scratch belongs to the scene handle and only grows.  kSlots copies are used in turn; a launch is
    // ordered behind the previous user of its slot through the slot's event.
    std::lock_guard<std::mutex> hold_slots(sc->slot_lock);
    const int slot = (int)(sc->next_slot++ % rtx_scene::kSlots);
    const uint64_t need = rtx::kScratchHeaderBytes + rtx::render_order_bytes(r, plan.macro) +
                          plan.grid_blocks * plan.block * rtx::render_lane_bytes(r.depth);
    if (!sc->slot_done[slot]) HIP_TRY(hipEventCreateWithFlags(&sc->slot_done[slot], hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(st, sc->slot_done[slot], 0));   // the slot's previous frame
    if (need > sc->scratch_bytes[slot]) {
        // grow every slot now: an allocation stalls the device, better once than once per slot
        for (int k = 0; k < rtx_scene::kSlots; k++) {
            if (need <= sc->scratch_bytes[k]) continue;
            if (sc->d_scratch[k]) HIP_TRY(hipFree(sc->d_scratch[k]));         // waits for launches still using it
            sc->d_scratch[k] = nullptr;
            sc->scratch_bytes[k] = 0;
            sc->slot_key[k].clear();
            HIP_TRY(hipMalloc(&sc->d_scratch[k], need));
            sc->scratch_bytes[k] = need;
            // the work counters start at zero; from then on every launch leaves them at zero again (the last wave
            // out of render_kernel), so no clearing dispatch sits in front of a frame
            HIP_TRY(hipMemset(sc->d_scratch[k], 0, rtx::kScratchHeaderBytes));
        }
    }
    bind_scratch(&plan, sc->d_scratch[slot]);
    // The work-queue table depends only on the launch geometry: the slot keeps it until a launch with
    // another geometry (or a smaller lane count, which moves nothing: the table sits in front) uses it.
    const std::vector<int64_t> key = {r.W, r.H, r.tile_w, r.tile_h, r.tile_rank, r.tile_count, plan.n_queues, plan.macro, plan.centre_first};
    if (key != sc->slot_key[slot]) {
        HIP_TRY(rtx::launch_tile_order((uint32_t *)r.tile_order, r, plan.macro, plan.n_queues, plan.centre_first, false, st));
        sc->slot_key[slot] = key;
    }
    // Diagnostics: RTX_WAVE_LOG=<file> records every wave's start/end tick (100 MHz) of this
    // launch, when it drew its latest block and which one, as four raw u64 per wave (and two more rows, zero unless the
    // kernel is a measurement build).  Synchronous; never set in production.
    const char *log_path = getenv("RTX_WAVE_LOG");
    struct DeviceBuffer { void *p = nullptr; ~DeviceBuffer() { (void)hipFree(p); } } log;
    const uint64_t n_waves = plan.grid_blocks * (plan.block / 64);
    if (log_path && *log_path) {
        HIP_TRY(hipMalloc(&log.p, n_waves * 32 + 64));      // (+ 8 words that only measurement builds of the kernel use)
        HIP_TRY(hipMemset(log.p, 0, n_waves * 32 + 64));
        r.wave_log = (unsigned long long *)log.p;
    }
    if (staged) {
        const uint64_t pixels = rtx_output_pixels(p), masks = (rtx::staged_mask_bytes(r.total_blocks, r.depth) + 255) / 256 * 256;
        const uint64_t want = masks + rtx::staged_record_bytes(pixels, r.depth);
        if (want > sc->record_bytes[slot]) {
            if (sc->d_records[slot]) HIP_TRY(hipFree(sc->d_records[slot]));   // waits for launches still using it
            sc->d_records[slot] = nullptr;
            sc->record_bytes[slot] = 0;
            HIP_TRY(hipMalloc(&sc->d_records[slot], want));
            sc->record_bytes[slot] = want;
            sc->records_dirty[slot] = true;
        }
        // The hit masks must be zero before a frame.  A frame's last launch leaves them so; only new memory, a frame whose
        // launches did not all go out, or a change of geometry (the masks' place and count move with it) needs the fill.
        const std::vector<int64_t> mask_key = {(int64_t)r.total_blocks, r.depth};
        if (sc->records_dirty[slot] || mask_key != sc->mask_key[slot]) HIP_TRY(hipMemsetAsync(sc->d_records[slot], 0, masks, st));
        sc->mask_key[slot] = mask_key;
        sc->records_dirty[slot] = true;
        if (int rc = launch_staged_frame(sc, plan, sc->d_records[slot], pixels, d_stats != nullptr, st)) return rc;
        sc->records_dirty[slot] = getenv("RTX_STAGED_ONLY") != nullptr;
    } else {
        HIP_TRY(rtx::launch_render(sc->dev, r, plan.pruned, d_stats != nullptr, plan.block, (uint32_t)plan.grid_blocks, st));
    }
    HIP_TRY(hipEventRecord(sc->slot_done[slot], st));
    if (log.p) {
        std::vector<unsigned long long> h(n_waves * 4 + 8);
        HIP_TRY(hipMemcpy(h.data(), log.p, n_waves * 32 + 64, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(log_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    return RTX_OK;
}

// The same launch with every byte of its scratch in the library's file-scope device array
// (rtx::noalloc_scratch): no allocation, no event, no host-side state.  A frame that does not fit
// the array whole runs with fewer resident lanes; one whose work-queue table alone does not fit is
// refused.
static int render_device_noalloc_impl(rtx_scene *sc, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id, void *d_rgb_i32,
                                      void *d_t, void *d_stats, void *stream) {
    LaunchPlan plan;
    if (int rc = plan_render(sc, p, d_rgb_u8, d_hit_id, d_rgb_i32, d_t, d_stats, nullptr, &plan)) return rc;
    if (plan.grid_blocks == 0) return RTX_OK;
    rtx::DevRender &r = plan.r;
    const hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(sc->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "cannot select the scene's device");
    // Replaying these launches from a recorded graph is not something this library has verified (an attempt replayed
    // an empty frame), so a stream that is being captured is refused rather than served with an untested promise.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
    if (capturing != hipStreamCaptureStatusNone && !getenv("RTX_ALLOW_CAPTURE"))
        return fail(RTX_ERR_UNSUPPORTED, "rtx_render_device_noalloc: the stream is being captured; launch it directly");
    void *base = nullptr;
    HIP_TRY(rtx::noalloc_scratch(sc->device, &base));
    const uint64_t fixed = rtx::kScratchHeaderBytes + rtx::render_order_bytes(r, plan.macro);
    const uint64_t per_block = plan.block * rtx::render_lane_bytes(r.depth);
    if (fixed + per_block > rtx::kNoallocScratchBytes)
        return fail(RTX_ERR_UNSUPPORTED, "rtx_render_device_noalloc: the frame has more tiles than the fixed scratch array's "
                                         "work-queue table holds (48 MiB in all; use larger tiles or rtx_render_device)");
    plan.grid_blocks = std::min<uint64_t>(plan.grid_blocks, (rtx::kNoallocScratchBytes - fixed) / per_block);
    bind_scratch(&plan, base);
    // two kernels and nothing else on the stream: the first one clears the last call's counters and writes the table
    HIP_TRY(rtx::launch_tile_order((uint32_t *)r.tile_order, r, plan.macro, plan.n_queues, plan.centre_first, true, st));
    HIP_TRY(rtx::launch_render(sc->dev, r, plan.pruned, d_stats != nullptr, plan.block, (uint32_t)plan.grid_blocks, st));
    return RTX_OK;
}

// A device buffer and its pinned twin on the host, grown on demand and kept with the scene.
static hipError_t grow_pair(void **dev, void **host, uint64_t *have, uint64_t need) {
    if (need <= *have) return hipSuccess;
    (void)hipFree(*dev);
    if (*host) (void)hipHostFree(*host);
    *dev = nullptr; *host = nullptr; *have = 0;
    hipError_t e = hipMalloc(dev, need);
    if (e == hipSuccess) e = hipHostMalloc(host, need, hipHostMallocDefault);
    if (e == hipSuccess) *have = need;
    return e;
}

static int render_impl(rtx_scene *sc, const rtx_render_params *p, uint8_t *rgb_out, int32_t *hit_id_out, rtx_stats *stats) {
    if (!sc) return fail(RTX_ERR_INVALID, "scene is NULL");
    if (!rgb_out) return fail(RTX_ERR_INVALID, "rgb_out is NULL");
    const uint64_t px = rtx_output_pixels(p);
    ResolvedParams rp;
    if (int rc = resolve_params(p, &rp)) return rc;
    if (stats) memset(stats, 0, sizeof *stats);
    if (px == 0) return RTX_OK;
    std::lock_guard<std::mutex> hold(sc->lock);
    DeviceGuard guard(sc->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "cannot select the scene's device");
    // Output lands in device buffers kept with the scene, travels to pinned host memory by DMA and is copied from
    // there into the caller's (pageable) arrays on the worker threads: a pageable device-to-host copy of a 4K frame
    // takes as long as rendering it.
    HIP_TRY(grow_pair(&sc->d_out_rgb, &sc->h_out_rgb, &sc->out_rgb_bytes, px * 3));
    if (hit_id_out) HIP_TRY(grow_pair(&sc->d_out_hit, &sc->h_out_hit, &sc->out_hit_bytes, px * 4));
    if (stats && !sc->d_out_stats) HIP_TRY(hipMalloc(&sc->d_out_stats, 15 * 8));
    if (!sc->out_e0) HIP_TRY(hipEventCreate(&sc->out_e0));
    if (!sc->out_e1) HIP_TRY(hipEventCreate(&sc->out_e1));
    void *d_rgb = sc->d_out_rgb, *d_hit = hit_id_out ? sc->d_out_hit : nullptr, *d_stats = stats ? sc->d_out_stats : nullptr;
    hipStream_t stream = nullptr;
    // pixels outside the window (and the padding of packed tiles) read 0 / -1
    HIP_TRY(hipMemsetAsync(d_rgb, 0, px * 3, stream));
    if (d_hit) HIP_TRY(hipMemsetAsync(d_hit, 0xff, px * 4, stream));
    if (d_stats) HIP_TRY(hipMemsetAsync(d_stats, 0, 15 * 8, stream));
    HIP_TRY(hipEventRecord(sc->out_e0, stream));
    if (int rc = rtx_render_device(sc, p, d_rgb, d_hit, nullptr, nullptr, d_stats, stream)) return rc;
    HIP_TRY(hipEventRecord(sc->out_e1, stream));
    // The frame comes back in pieces: while piece k + 1 is on its way to pinned memory, the worker threads copy piece k
    // into the caller's array (a 4K frame is 25 MB: 0.5 ms of DMA and 0.35 ms of host copies, one behind the other otherwise).
    constexpr int kPieces = rtx_scene::kOutPieces;
    const uint64_t rgb_bytes = px * 3, piece = (rgb_bytes / kPieces + 4095) / 4096 * 4096;
    int n_pieces = 0;
    for (uint64_t off = 0; off < rgb_bytes; off += piece, n_pieces++) {
        const uint64_t len = std::min(piece, rgb_bytes - off);
        HIP_TRY(hipMemcpyAsync((char *)sc->h_out_rgb + off, (const char *)d_rgb + off, len, hipMemcpyDeviceToHost, stream));
        if (!sc->out_piece[n_pieces]) HIP_TRY(hipEventCreateWithFlags(&sc->out_piece[n_pieces], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(sc->out_piece[n_pieces], stream));
    }
    if (d_hit) HIP_TRY(hipMemcpyAsync(sc->h_out_hit, d_hit, px * 4, hipMemcpyDeviceToHost, stream));
    unsigned long long c[15] = {};
    if (d_stats) HIP_TRY(hipMemcpyAsync(c, d_stats, sizeof c, hipMemcpyDeviceToHost, stream));
    for (int k = 0; k < n_pieces; k++) {
        HIP_TRY(hipEventSynchronize(sc->out_piece[k]));
        const uint64_t off = (uint64_t)k * piece;
        rtx::parallel_copy(rgb_out + off, (const char *)sc->h_out_rgb + off, std::min(piece, rgb_bytes - off));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (d_hit) rtx::parallel_copy(hit_id_out, sc->h_out_hit, px * 4);
    if (stats) {
        stats->primary_rays = c[0]; stats->reflection_rays = c[1]; stats->shadow_rays = c[2];
        stats->shadow_occluded = c[3]; stats->closest_hits = c[4]; stats->node_visits = c[5];
        stats->box_hits = c[6]; stats->leaf_visits = c[7]; stats->triangle_tests = c[8];
        for (int i = 0; i < 3; i++) { stats->rounds[i] = c[9 + i]; stats->lane_steps[i] = c[12 + i]; }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, sc->out_e0, sc->out_e1));
        stats->kernel_ms = ms;
    }
    return RTX_OK;
}

// rtx_render() in two halves, so that a frame's trip to host memory overlaps the next frame's render.
static int render_begin_impl(rtx_scene *sc, const rtx_render_params *p, int32_t slot, int32_t want_hit_ids) {
    if (!sc) return fail(RTX_ERR_INVALID, "scene is NULL");
    if (slot < 0 || slot >= rtx_scene::kHostSlots) return fail(RTX_ERR_INVALID, "frame slot must be 0..3");
    ResolvedParams rp;
    if (int rc = resolve_params(p, &rp)) return rc;
    if (want_hit_ids && rp.r.layout == RTX_LAYOUT_PACKED_TILES) return fail(RTX_ERR_INVALID, "hit ids / t are only written in row-major layout");
    const uint64_t px = rtx_output_pixels(p);
    std::lock_guard<std::mutex> hold(sc->lock);
    rtx_scene::HostSlot &h = sc->host_slots[slot];
    if (h.busy) return fail(RTX_ERR_INVALID, "frame slot still holds a frame: call rtx_render_end() first");
    DeviceGuard guard(sc->device);
    if (!guard.ok) return fail(RTX_ERR_DEVICE, "cannot select the scene's device");
    if (!h.stream) HIP_TRY(hipStreamCreateWithFlags(&h.stream, hipStreamNonBlocking));
    h.pixels = px;
    h.want_hit = want_hit_ids != 0;
    h.busy = true;
    // a failure below leaves the slot free again
    struct Release { rtx_scene::HostSlot &h; bool keep = false; ~Release() { if (!keep) h.busy = false; } } release{h};
    if (px == 0) { release.keep = true; return RTX_OK; }
    HIP_TRY(grow_pair(&h.d_rgb, &h.h_rgb, &h.rgb_bytes, px * 3));
    if (h.want_hit) HIP_TRY(grow_pair(&h.d_hit, &h.h_hit, &h.hit_bytes, px * 4));
    // pixels outside the window (and the padding of packed tiles) read 0 / -1
    HIP_TRY(hipMemsetAsync(h.d_rgb, 0, px * 3, h.stream));
    if (h.want_hit) HIP_TRY(hipMemsetAsync(h.d_hit, 0xff, px * 4, h.stream));
    if (int rc = rtx_render_device(sc, p, h.d_rgb, h.want_hit ? h.d_hit : nullptr, nullptr, nullptr, nullptr, h.stream)) return rc;
    HIP_TRY(hipMemcpyAsync(h.h_rgb, h.d_rgb, px * 3, hipMemcpyDeviceToHost, h.stream));
    if (h.want_hit) HIP_TRY(hipMemcpyAsync(h.h_hit, h.d_hit, px * 4, hipMemcpyDeviceToHost, h.stream));
    release.keep = true;
    return RTX_OK;
}

static int render_end_impl(rtx_scene *sc, int32_t slot, uint8_t *rgb_out, int32_t *hit_id_out) {
    if (!sc) return fail(RTX_ERR_INVALID, "scene is NULL");
    if (slot < 0 || slot >= rtx_scene::kHostSlots) return fail(RTX_ERR_INVALID, "frame slot must be 0..3");
    rtx_scene::HostSlot &h = sc->host_slots[slot];
    {
        std::lock_guard<std::mutex> hold(sc->lock);
        if (!h.busy) return fail(RTX_ERR_INVALID, "frame slot holds no frame: call rtx_render_begin() first");
        if (hit_id_out && !h.want_hit) return fail(RTX_ERR_INVALID, "hit ids were not asked for in rtx_render_begin()");
    }
    // whatever happens below, the slot is free again when this call returns: a frame whose wait failed is lost,
    // not a reason to refuse every later rtx_render_begin() on the slot
    struct Release { rtx_scene *sc; rtx_scene::HostSlot &h; ~Release() { std::lock_guard<std::mutex> hold(sc->lock); h.busy = false; } } release{sc, h};
    if (h.pixels) {
        DeviceGuard guard(sc->device);
        if (!guard.ok) return fail(RTX_ERR_DEVICE, "cannot select the scene's device");
        HIP_TRY(hipStreamSynchronize(h.stream));
        if (rgb_out) rtx::parallel_copy(rgb_out, h.h_rgb, h.pixels * 3);
        if (hit_id_out) rtx::parallel_copy(hit_id_out, h.h_hit, h.pixels * 4);
    }
    return RTX_OK;
}

int rtx_untile(const rtx_render_params *p, int32_t n_ranks, const void *d_gathered, void *d_out, int32_t bpp, void *stream) {
    ResolvedParams rp;
    if (int rc = resolve_params(p, &rp)) return rc;
    if (!d_gathered || !d_out) return fail(RTX_ERR_INVALID, "NULL buffer");
    if (n_ranks < 1 || n_ranks != rp.r.tile_count) return fail(RTX_ERR_INVALID, "n_ranks must equal params.tile_count");
    if (bpp < 1 || bpp > 16) return fail(RTX_ERR_INVALID, "bytes_per_pixel must be 1..16");
    const uint64_t rank_stride = packed_tiles_per_rank(rp.r) * (uint64_t)rp.r.tile_w * (uint64_t)rp.r.tile_h * (uint64_t)bpp;
    HIP_TRY(rtx::launch_untile((const uint8_t *)d_gathered, (uint8_t *)d_out, rp.r.W, rp.r.H, rp.r.tile_w, rp.r.tile_h,
                               rp.r.tiles_x, n_ranks, rank_stride, bpp, (hipStream_t)stream));
    return RTX_OK;
}

static int selftest_pow_impl(uint64_t n, uint64_t seed, uint64_t *differing, uint64_t *max_ulps) {
    if (!differing || !max_ulps) return fail(RTX_ERR_INVALID, "NULL argument");
    *differing = 0;
    *max_ulps = 0;
    if (n == 0) return RTX_OK;
    // cosines in (0, 1] and the exponents a material can carry (computeLighting, scene.hpp:172-180)
    std::vector<double> x(n), got(n);
    std::vector<int32_t> e(n);
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return z; };
    for (uint64_t i = 0; i < n; i++) {
        x[i] = (double)((next() >> 11) + 1) * 0x1p-53;                 // (0, 1]
        if (i % 16 == 0) x[i] = 1.0 - (double)(next() >> 40) * 0x1p-53; // close to 1, where pow loses most
        static const int32_t menu[] = {2, 3, 5, 8, 10, 16, 32, 50, 64, 100, 250, 1000};
        e[i] = menu[next() % (sizeof menu / sizeof menu[0])];
    }
    double *d_x = nullptr, *d_out = nullptr;
    int32_t *d_e = nullptr;
    hipError_t err = hipMalloc((void **)&d_x, n * sizeof(double));
    if (err == hipSuccess) err = hipMalloc((void **)&d_out, n * sizeof(double));
    if (err == hipSuccess) err = hipMalloc((void **)&d_e, n * sizeof(int32_t));
    if (err == hipSuccess) err = hipMemcpy(d_x, x.data(), n * sizeof(double), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d_e, e.data(), n * sizeof(int32_t), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = rtx::launch_powtest(d_x, d_e, d_out, n, nullptr);
    if (err == hipSuccess) err = hipMemcpy(got.data(), d_out, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_x); (void)hipFree(d_out); (void)hipFree(d_e);
    if (err != hipSuccess) return fail(RTX_ERR_DEVICE, std::string("pow self-test: ") + hipGetErrorString(err));
    for (uint64_t i = 0; i < n; i++) {
        const double want = std::pow(x[i], (double)e[i]);
        if (want < 2.2250738585072014e-308) continue;     // below the normal range: adds nothing to a lighting sum
        int64_t a, b;
        memcpy(&a, &want, 8);
        memcpy(&b, &got[i], 8);
        const uint64_t ulps = (uint64_t)(a > b ? a - b : b - a);      // same sign, finite: ordered like integers
        if (ulps) (*differing)++;
        if (ulps > *max_ulps) *max_ulps = ulps;
    }
    return RTX_OK;
}

static int selftest_sqrt_impl(uint64_t n, uint64_t seed, uint64_t *differing) {
    if (!differing) return fail(RTX_ERR_INVALID, "NULL argument");
    *differing = 0;
    if (n == 0) return RTX_OK;
    std::vector<double> x(n), got(n);
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return z; };
    for (uint64_t i = 0; i < n; i++) {
        // a full 53-bit significand, exponents -500..500 for three quarters of the operands, -2..2 for the rest
        const double m = 1.0 + (double)(next() >> 12) * 0x1p-52;
        const int span = (i & 3) ? 1001 : 5;
        x[i] = std::ldexp(m, (int)(next() % (uint64_t)span) - span / 2);
    }
    double *d_x = nullptr, *d_out = nullptr;
    hipError_t err = hipMalloc((void **)&d_x, n * sizeof(double));
    if (err == hipSuccess) err = hipMalloc((void **)&d_out, n * sizeof(double));
    if (err == hipSuccess) err = hipMemcpy(d_x, x.data(), n * sizeof(double), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = rtx::launch_sqrttest(d_x, d_out, n, nullptr);
    if (err == hipSuccess) err = hipMemcpy(got.data(), d_out, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_x); (void)hipFree(d_out);
    if (err != hipSuccess) return fail(RTX_ERR_DEVICE, std::string("sqrt self-test: ") + hipGetErrorString(err));
    for (uint64_t i = 0; i < n; i++) {
        const double want = std::sqrt(x[i]);
        if (memcmp(&want, &got[i], 8) != 0) (*differing)++;
    }
    return RTX_OK;
}

int rtx_selftest_division(uint64_t n, uint64_t seed, uint64_t *mismatches) {
    if (!mismatches) return fail(RTX_ERR_INVALID, "NULL argument");
    unsigned long long *d = nullptr, h = 0;
    HIP_TRY(hipMalloc((void **)&d, sizeof h));
    hipError_t e = hipMemset(d, 0, sizeof h);
    if (e == hipSuccess) e = rtx::launch_divtest(seed, n, d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RTX_ERR_DEVICE, std::string("division self-test: ") + hipGetErrorString(e));
    *mismatches = h;
    return RTX_OK;
}

// saveCanvas, src/main.cpp:96-116 (the clamp already happened on the device)
int rtx_save_ppm(const char *path, int32_t width, int32_t height, const uint8_t *rgb) {
    if (!path || !rgb || width < 0 || height < 0) return fail(RTX_ERR_INVALID, "bad argument");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RTX_ERR_IO, "Failed to open file.\n");
    fprintf(f, "P6\n%d %d 255\n", width, height);
    const size_t n = (size_t)width * (size_t)height * 3;
    const bool ok = fwrite(rgb, 1, n, f) == n;
    if (fclose(f) != 0 || !ok) return fail(RTX_ERR_IO, "short write");
    return RTX_OK;
}

// The same canvas as a PNG (8-bit RGB, rows unfiltered, deflate "stored" blocks: no compression
// library needed and nothing to get wrong; a 4K frame is 24.9 MB either way).
static int save_png_impl(const char *path, int32_t width, int32_t height, const uint8_t *rgb) {
    if (!path || !rgb || width <= 0 || height <= 0) return fail(RTX_ERR_INVALID, "bad argument");
    // closed on every way out, the vectors below may throw
    struct File { FILE *f; ~File() { if (f) fclose(f); } } file{fopen(path, "wb")};
    FILE *f = file.f;
    if (!f) return fail(RTX_ERR_IO, "Failed to open file.\n");
    static uint32_t table[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    });
    auto crc = [&](uint32_t c, const uint8_t *p, size_t n) { while (n--) c = table[(c ^ *p++) & 0xff] ^ (c >> 8); return c; };
    auto be32 = [](uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    bool ok = true;
    auto chunk = [&](const char type[4], const std::vector<uint8_t> &data) {
        uint8_t head[8], tail[4];
        be32(head, (uint32_t)data.size());
        memcpy(head + 4, type, 4);
        uint32_t c = crc(0xffffffffu, head + 4, 4);
        c = crc(c, data.data(), data.size()) ^ 0xffffffffu;
        be32(tail, c);
        ok = ok && fwrite(head, 1, 8, f) == 8 && (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) &&
             fwrite(tail, 1, 4, f) == 4;
    };
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    ok = fwrite(magic, 1, 8, f) == 8;
    std::vector<uint8_t> ihdr(13, 0);
    be32(&ihdr[0], (uint32_t)width);
    be32(&ihdr[4], (uint32_t)height);
    ihdr[8] = 8;    // bit depth
    ihdr[9] = 2;    // colour type: RGB
    chunk("IHDR", ihdr);
    // scanlines: filter byte 0 + row
    const size_t row = (size_t)width * 3, raw = (row + 1) * (size_t)height;
    std::vector<uint8_t> z;
    z.reserve(raw + raw / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;                       // Adler-32 of the raw stream
    std::vector<uint8_t> line(row + 1, 0);
    size_t in_block = 0;
    auto open_block = [&](size_t remaining) {
        const size_t n = remaining < 65535 ? remaining : 65535;
        z.push_back(remaining <= 65535 ? 1 : 0);  // BFINAL, BTYPE = 00
        z.push_back(n & 0xff); z.push_back(n >> 8);
        z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        in_block = n;
    };
    size_t remaining = raw;
    open_block(remaining);
    for (int32_t y = 0; y < height; y++) {
        memcpy(&line[1], rgb + (size_t)y * row, row);
        for (size_t i = 0; i < line.size();) {
            if (in_block == 0) open_block(remaining);
            const size_t n = std::min(in_block, line.size() - i);
            z.insert(z.end(), line.begin() + i, line.begin() + i + n);
            for (size_t k = 0; k < n; k++) { a += line[i + k]; if (a >= 65521) a -= 65521; b += a; if (b >= 65521) b -= 65521; }
            i += n; in_block -= n; remaining -= n;
        }
    }
    uint8_t ad[4];
    be32(ad, (b << 16) | a);
    z.insert(z.end(), ad, ad + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
    file.f = nullptr;
    if (fclose(f) != 0 || !ok) return fail(RTX_ERR_IO, "short write");
    return RTX_OK;
}

// ------------------------------------------------------------------ guarded entry points
int rtx_host_scene_load_obj_ex(const char *obj_path, const char *preset, uint32_t flags, int32_t device, rtx_host_scene **out) {
    return guarded([&]() -> int { return host_scene_load_obj_ex_impl(obj_path, preset, flags, device, out); });
}

int rtx_host_scene_standin_ex(const char *preset, uint64_t n_triangles, uint64_t seed, const char *obj_path_out, uint32_t flags,
                              int32_t device, rtx_host_scene **out) {
    return guarded([&]() -> int { return host_scene_standin_ex_impl(preset, n_triangles, seed, obj_path_out, flags, device, out); });
}

int rtx_host_scene_from_desc(const rtx_scene_desc *d, rtx_host_scene **out) {
    return guarded([&]() -> int { return host_scene_from_desc_impl(d, out); });
}

int rtx_host_scene_set_lights(rtx_host_scene *hs, double ambient, uint32_t n_point, const rtx_light *point,
                              uint32_t n_directional, const rtx_light *directional) {
    return guarded([&]() -> int { return host_scene_set_lights_impl(hs, ambient, n_point, point, n_directional, directional); });
}

int rtx_bvh_build(const double *vertices, uint64_t n, int32_t builder, int32_t device, rtx_bvh_node *nodes_out,
                  uint64_t *indices_out, uint64_t *n_nodes_out, int32_t *depth_out, double *seconds_out) {
    return guarded([&]() -> int { return bvh_build_impl(vertices, n, builder, device, nodes_out, indices_out, n_nodes_out, depth_out, seconds_out); });
}

int rtx_host_scene_rebuild_bvh(rtx_host_scene *hs, int32_t builder, int32_t device, double *seconds_out) {
    return guarded([&]() -> int { return host_scene_rebuild_bvh_impl(hs, builder, device, seconds_out); });
}

int rtx_scene_create_from_host(const rtx_host_scene *hs, int32_t device, rtx_scene **out) {
    return guarded([&]() -> int { return scene_create_from_host_impl(hs, device, out); });
}

int rtx_scene_create_with_builder(const rtx_scene_desc *desc, int32_t builder, rtx_scene **out) {
    return guarded([&]() -> int { return scene_create_with_builder_impl(desc, builder, out); });
}

int rtx_render_device_timed(rtx_scene *sc, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id, void *d_rgb_i32, void *d_t,
                            void *d_stats, void *d_ticks, void *stream) {
    // RTX_PIPELINE=staged (experiments, and the test suite's second pass): single-sample frames take the staged pipeline
    const char *pipe = getenv("RTX_PIPELINE");
    const bool staged = pipe && !strcmp(pipe, "staged") && p && p->spp <= 1;
    return guarded([&]() -> int { return render_device_timed_impl(sc, p, d_rgb_u8, d_hit_id, d_rgb_i32, d_t, d_stats, d_ticks, stream, 1, nullptr, staged); });
}

int rtx_render_device_batch(rtx_scene *sc, const rtx_render_params *p, int32_t n_frames, const double *cameras, void *d_rgb_u8,
                            void *d_rgb_i32, void *d_stats, void *d_ticks, void *stream) {
    return guarded([&]() -> int { return render_device_timed_impl(sc, p, d_rgb_u8, nullptr, d_rgb_i32, nullptr, d_stats, d_ticks, stream, n_frames, cameras); });
}

int rtx_render_device_staged(rtx_scene *sc, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id, void *d_rgb_i32, void *d_t,
                             void *d_stats, void *d_ticks, void *stream) {
    return guarded([&]() -> int { return render_device_timed_impl(sc, p, d_rgb_u8, d_hit_id, d_rgb_i32, d_t, d_stats, d_ticks, stream, 1, nullptr, true); });
}

int rtx_render_device_noalloc(rtx_scene *sc, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id, void *d_rgb_i32, void *d_t,
                              void *d_stats, void *stream) {
    return guarded([&]() -> int { return render_device_noalloc_impl(sc, p, d_rgb_u8, d_hit_id, d_rgb_i32, d_t, d_stats, stream); });
}

int rtx_render(rtx_scene *sc, const rtx_render_params *p, uint8_t *rgb_out, int32_t *hit_id_out, rtx_stats *stats) {
    return guarded([&]() -> int { return render_impl(sc, p, rgb_out, hit_id_out, stats); });
}

int rtx_render_begin(rtx_scene *sc, const rtx_render_params *p, int32_t slot, int32_t want_hit_ids) {
    return guarded([&]() -> int { return render_begin_impl(sc, p, slot, want_hit_ids); });
}

int rtx_render_end(rtx_scene *sc, int32_t slot, uint8_t *rgb_out, int32_t *hit_id_out) {
    return guarded([&]() -> int { return render_end_impl(sc, slot, rgb_out, hit_id_out); });
}

int rtx_selftest_pow(uint64_t n, uint64_t seed, uint64_t *differing, uint64_t *max_ulps) {
    return guarded([&]() -> int { return selftest_pow_impl(n, seed, differing, max_ulps); });
}

int rtx_selftest_sqrt(uint64_t n, uint64_t seed, uint64_t *differing) {
    return guarded([&]() -> int { return selftest_sqrt_impl(n, seed, differing); });
}

int rtx_save_png(const char *path, int32_t width, int32_t height, const uint8_t *rgb) {
    return guarded([&]() -> int { return save_png_impl(path, width, height, rgb); });
}

}  // extern "C"
