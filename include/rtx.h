/* rtx.h -- C ABI of the MI355X render path.
 *
 * Drop-in for the reference's threaded per-pixel loop: everything
 * `void render(Scene scene, int xMin, int xMax)` (src/main.cpp:18,75-88) does for
 * one frame -- camera ray, BVH closest hit, shadow rays, shading, the reflection
 * blend, putPixel -- runs on the GPU behind rtx_render().  Scene loading, the
 * binned-SAH BVH build and image output stay on the host.
 *
 * Plain pointers and sizes only; no exceptions cross this boundary.  Every call
 * returns RTX_OK (0) or a negative rtx_status; rtx_last_error() gives the text for
 * the calling thread.  (The reference throws string literals that main() prints
 * and turns into exit code -1, src/main.cpp:27-32,65-70.)
 *
 * Environment: results never depend on it.  For experiments the library reads, per launch,
 * RTX_W_INNER / RTX_W_LEAF / RTX_W_SHADE / RTX_TAIL_LANES / RTX_ALSO_LANES / RTX_ALSO_LEAF /
 * RTX_ALSO_SHADE (the wave scheduler's thresholds), RTX_BLOCKS_PER_CU, RTX_XCD_QUEUES, RTX_MACRO
 * (work distribution), RTX_PIPELINE=staged (rtx_render_device takes the staged pipeline for single-sample frames)
 * and RTX_WAVE_LOG=<file> (per-wave clock values; synchronous); the loaders
 * print stage times to stderr under RTX_TIMING=1 and size their thread pool by OMP_NUM_THREADS.
 *
 * Threading: a scene handle may be used from any thread.  rtx_render() serialises per
 * handle; rtx_render_device() launches may be in flight on several streams at once (up to
 * eight frames of one scene overlap, a ninth waits for the first).  All calls are
 * synchronous unless a stream is passed.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_INVALID = -1,     /* bad argument                                   */
    RTX_ERR_IO = -2,          /* "Failed to open file." and friends             */
    RTX_ERR_DEVICE = -3,      /* HIP runtime error                              */
    RTX_ERR_UNSUPPORTED = -4, /* valid input this build cannot handle           */
    RTX_ERR_NOMEM = -5
} rtx_status;

/* ---- host-side scene data ------------------------------------------------------ */

enum {
    RTX_BUILDER_HOST = 0,        /* scalar restatement of the reference builder on one host core */
    RTX_BUILDER_DEVICE = 1       /* the same tree built by HIP kernels                           */
};

/* BVHNode as the reference lays it out (include/bvh.hpp:25-30): 64 bytes.
 * count == 0: inner node, children are `first` and `first + 1`;
 * count  > 0: leaf over indices[first .. first+count). */
typedef struct rtx_bvh_node {
    double bmin[3];
    double bmax[3];
    uint64_t first;
    uint64_t count;
} rtx_bvh_node;

/* Point (include/light.hpp:8-11): v = position; Directional (:13-16): v = direction. */
typedef struct rtx_light {
    double intensity;
    double v[3];
} rtx_light;

/* What `Scene` holds (include/scene.hpp:15-23), as plain arrays.  Triangle index i
 * below is the reference's index into Scene::triangles, and it is the id that
 * rtx_render reports. */
typedef struct rtx_scene_desc {
    uint64_t n_triangles;
    const double *vertices;      /* [n][9]  v0 v1 v2                                  */
    const double *normals;       /* [n][3]  as the Triangle ctor computes them
                                    (include/triangle.hpp:21-23); NULL = compute here */
    const int32_t *colors;       /* [n][3]  Color r g b (ints, not clamped)            */
    const int32_t *specularity;  /* [n]     -1 = no specular term                      */
    const double *reflectivity;  /* [n]     < 0 = no reflection                        */
    /* BVH built by the caller with the reference builder; NULL/0 = build it here with
     * this library's restatement of that builder (identical arrays). */
    uint64_t n_nodes;
    const rtx_bvh_node *nodes;
    const uint64_t *indices;     /* [n]                                               */
    double viewport[3];
    double camera_position[3];
    double camera_rotation_deg[3];
    int32_t background[3];
    double ambient;
    uint32_t n_point_lights;
    const rtx_light *point_lights;
    uint32_t n_directional_lights;
    const rtx_light *directional_lights;
    int32_t device;              /* HIP device ordinal, -1 = current device            */
} rtx_scene_desc;

/* A host scene: the result of parseOBJ plus one scene factory plus the BVH build
 * (include/scene.hpp:213-253, :255-448, include/bvh.hpp:84-162).  Owns its arrays. */
typedef struct rtx_host_scene rtx_host_scene;

/* Replaces Scene::parseOBJ(path) + Scene::<preset>() (include/scene.hpp:213-448).
 * preset: "teapot", "bunny", "erato", "dragon" or "aurelius" (or "showcase", no factory of the
 * reference: the teapot's camera with two point lights, a directional light and specular
 * exponents 10 and 50 on the model) -- camera position,
 * rotation (0, 0.1, 0) deg, white background, ambient 0.2, one directional light
 * 0.5 * (0,1,-1) and the last-10-triangles material overrides.  Prints the
 * reference's "[ TRI ] n triangles." / "[ BVH ] s seconds." lines when verbose. */
int rtx_host_scene_load_obj(const char *obj_path, const char *preset, int verbose, rtx_host_scene **out);

enum {
    RTX_LOAD_VERBOSE = 1,        /* print the reference's "[ TRI ]" / "[ BVH ]" lines                       */
    RTX_LOAD_DEVICE_BVH = 2,     /* build the BVH with RTX_BUILDER_DEVICE (same arrays, copied back)        */
    RTX_LOAD_TOLERANT = 4,       /* general OBJ text instead of the reference's dialect, see below          */
    RTX_LOAD_NO_BVH = 8,         /* no BVH on the host: rtx_scene_create_from_host builds it on the device  */
    RTX_STANDIN_KNOT = 16,       /* rtx_host_scene_standin_ex only: the knotted-tube stand-in, see below    */
    RTX_LOAD_LAST_FACE_TWICE = 32 /* with RTX_LOAD_TOLERANT: the last face once more, as the reference's own
                                     pipeline yields it (tools/denormalize.py ends its output with a newline,
                                     after which parseOBJ reads the last face twice, include/scene.hpp:223-246):
                                     triangle count, ids and the presets' last-ten materials then match a frame
                                     the reference renders from the same source mesh                         */
};

/* The call above with options.  RTX_LOAD_TOLERANT reads OBJ files the reference's loader
 * cannot: it does what tools/denormalize.py:1-36 does before parseOBJ sees the file, in one
 * pass over the text -- `v` records are collected first, then every `f` record contributes
 * ONE triangle from the leading index of its first three `a`, `a/b`, `a//c` or `a/b/c`
 * groups, negative indices count back from the end of the vertex list -- and every other
 * record (vn, vt, o, g, s, usemtl, comments) is skipped.  Unlike the strict loader it does
 * not repeat the last face when the file ends in a newline (include/scene.hpp:223-246 does,
 * and the strict loader keeps that so that triangle ids match the reference's). */
int rtx_host_scene_load_obj_ex(const char *obj_path, const char *preset, uint32_t flags, int32_t device,
                               rtx_host_scene **out);

/* The stand-in for a missing mesh: a seeded procedural closed surface with exactly
 * n_triangles - 10 faces inside the 10-triangle open box of tools/range.py:40-54,
 * framed for `preset`'s camera.  Written as an .obj in the reference's dialect when
 * obj_path_out != NULL (no trailing newline), and/or loaded like the call above.
 * No counterpart in the reference (its bunny/dragon/erato/aurelius meshes are not
 * redistributable with this repository). */
int rtx_host_scene_standin(const char *preset, uint64_t n_triangles, uint64_t seed, const char *obj_path_out,
                           int verbose, rtx_host_scene **out);
/* The same with RTX_LOAD_* flags (RTX_LOAD_TOLERANT has no meaning here).  RTX_STANDIN_KNOT selects the
 * second stand-in family: the tube of a (2,3) torus knot, tessellated to the same face count -- concave,
 * self-occluding and self-reflecting like a scanned statue -- in a box placed so that its open front
 * fills the view, as the box of the reference's teapot scene does (include/scene.hpp:259). */
int rtx_host_scene_standin_ex(const char *preset, uint64_t n_triangles, uint64_t seed, const char *obj_path_out,
                              uint32_t flags, int32_t device, rtx_host_scene **out);

/* Scene from caller arrays (copies them; builds normals/BVH when absent). */
int rtx_host_scene_from_desc(const rtx_scene_desc *desc, rtx_host_scene **out);

/* Borrow the arrays of a host scene as a desc (valid until the host scene is freed). */
int rtx_host_scene_desc(const rtx_host_scene *hs, rtx_scene_desc *out);
/* Tree depth of the BVH (root = 0). */
int32_t rtx_host_scene_bvh_depth(const rtx_host_scene *hs);
/* Camera / light edits on a loaded scene (the reference edits Scene members directly). */
int rtx_host_scene_set_camera(rtx_host_scene *hs, const double position[3], const double rotation_deg[3]);
int rtx_host_scene_set_lights(rtx_host_scene *hs, double ambient, uint32_t n_point, const rtx_light *point,
                              uint32_t n_directional, const rtx_light *directional);
void rtx_host_scene_free(rtx_host_scene *hs);

/* ---- BVH build ------------------------------------------------------------------- */

/* BVH::BVH (include/bvh.hpp:84-98: updateNodeBounds, subdivide, findBestSplitPlane,
 * :100-162,236-305) over n_triangles triangles given as 9 doubles each.  Both builders
 * return the reference's arrays bit for bit: nodes_out (room for 2n-1 records, *n_nodes_out
 * used, root = 0, children in pairs in depth-first order) and indices_out (n entries).
 * depth_out (nullable): depth of the deepest node, root = 0.  seconds_out (nullable): build
 * time; for the device builder without the upload of the vertices and the download of the
 * result.  device: HIP ordinal for RTX_BUILDER_DEVICE, -1 = current. */
int rtx_bvh_build(const double *vertices, uint64_t n_triangles, int32_t builder, int32_t device,
                  rtx_bvh_node *nodes_out, uint64_t *indices_out, uint64_t *n_nodes_out, int32_t *depth_out,
                  double *seconds_out);

/* Replaces the BVH of a loaded host scene by one built with `builder` (same arrays either
 * way); what the Scene ctor does at include/scene.hpp:47-51. */
int rtx_host_scene_rebuild_bvh(rtx_host_scene *hs, int32_t builder, int32_t device, double *seconds_out);

/* ---- device scene --------------------------------------------------------------- */

typedef struct rtx_scene rtx_scene;

/* Flattens the scene into the device layout and uploads it to desc->device.
 * Takes the place of passing `Scene` by value to each render thread
 * (src/main.cpp:46). */
int rtx_scene_create(const rtx_scene_desc *desc, rtx_scene **out);
/* The same with the builder named (RTX_BUILDER_*, below) for a desc without a BVH.
 * rtx_scene_create uses RTX_BUILDER_DEVICE: the vertices are uploaded once, the tree is
 * built and converted to the render layout on the device and never travels back; the
 * result is the same scene, bit for bit, as with RTX_BUILDER_HOST.  A desc that brings its
 * own BVH is validated on the host and uploaded as is, whatever `builder` says. */
int rtx_scene_create_with_builder(const rtx_scene_desc *desc, int32_t builder, rtx_scene **out);
/* Seconds the BVH build took inside the create call (device builder: kernels only);
 * 0 when the caller supplied the tree.  What Scene::Scene prints as "[ BVH ]"
 * (include/scene.hpp:47-51). */
double rtx_scene_build_seconds(const rtx_scene *scene);
/* BVH::nodesUsed (include/bvh.hpp:36) and the depth of the deepest node (root = 0) of the scene's tree. */
uint64_t rtx_scene_bvh_nodes(const rtx_scene *scene);
int32_t rtx_scene_bvh_depth(const rtx_scene *scene);
int rtx_scene_create_from_host(const rtx_host_scene *hs, int32_t device, rtx_scene **out);
void rtx_scene_destroy(rtx_scene *scene);
/* Bytes of HBM held by the scene (nodes + triangles + materials + lights). */
uint64_t rtx_scene_device_bytes(const rtx_scene *scene);

/* ---- render ---------------------------------------------------------------------- */

enum {
    RTX_TRAVERSAL_REFERENCE = 0, /* visits exactly the nodes BVH::intersectRay visits      */
    RTX_TRAVERSAL_PRUNED = 1     /* near-first order + distance pruning, same results      */
};

enum {
    RTX_LAYOUT_ROW_MAJOR = 0,    /* out[row][col], row 0 = top (putPixel, src/main.cpp:90-94) */
    RTX_LAYOUT_PACKED_TILES = 1  /* out[local tile][tile_h][tile_w]: only this rank's tiles   */
};

/* The arguments of render() + the compile-time constants of src/main.cpp:8-9,83. */
typedef struct rtx_render_params {
    int32_t width, height;       /* CANVAS_WIDTH / CANVAS_HEIGHT                              */
    int32_t x0, x1, y0, y1;      /* pixel window, columns [x0,x1) rows [y0,y1); all 0 = frame */
    int32_t depth;               /* recursionDepth of traceRay (3 in src/main.cpp:83), 0..7   */
    int32_t spp;                 /* 1 = the reference; > 1 = jittered supersampling           */
    uint64_t seed;               /* jitter seed                                               */
    int32_t traversal;           /* RTX_TRAVERSAL_*                                           */
    /* image-space sharding: the frame is cut into tile_w x tile_h tiles (multiples of 8),
     * numbered row-major; this call renders tile k iff k % tile_count == tile_rank.
     * tile_count <= 1 renders every tile. */
    int32_t tile_w, tile_h, tile_rank, tile_count;
    int32_t layout;              /* RTX_LAYOUT_*                                              */
} rtx_render_params;

/* Work counters of one render (all rays: closest-hit and shadow queries). */
typedef struct rtx_stats {
    uint64_t primary_rays, reflection_rays, shadow_rays, shadow_occluded, closest_hits;
    uint64_t node_visits, box_hits, leaf_visits, triangle_tests;
    /* scheduling of the render kernel: rounds a wave ran of each step kind (0 inner node pair,
     * 1 leaf triangle, 2 shading) summed over waves, and the lanes that took part in them;
     * lane_steps / (64 * rounds) is the lane utilisation of that kind of step */
    uint64_t rounds[3], lane_steps[3];
    double kernel_ms;            /* HIP-event time of the render kernel(s)                    */
} rtx_stats;

/* depth 3, one sample, whole frame, RTX_TRAVERSAL_PRUNED, 32x32 tiles, row-major. */
void rtx_render_params_default(rtx_render_params *p, int32_t width, int32_t height);

/* Size in pixels of the output buffers for these params (row-major: W*H; packed:
 * owned tiles * tile_w * tile_h). */
uint64_t rtx_output_pixels(const rtx_render_params *p);

/* One frame into HOST buffers.  Replaces the 8-thread render() fan-out and join
 * (src/main.cpp:45-52).  rgb_out: 3 bytes per pixel, clamped as saveCanvas does
 * (src/main.cpp:107); hit_id_out (nullable): primary-ray triangle id or -1;
 * stats (nullable): asks the kernel to count its work.  The scene keeps the frame-sized
 * device buffer and pinned host buffer this call stages through (grown on demand, released by
 * rtx_scene_destroy). */
int rtx_render(rtx_scene *scene, const rtx_render_params *p, uint8_t *rgb_out, int32_t *hit_id_out,
               rtx_stats *stats);

/* rtx_render() in two halves for a loop over frames (the reference renders three in a row,
 * src/main.cpp:42-61): rtx_render_begin() queues the frame on frame slot `slot` (0..3; each slot
 * has its own stream, device buffers and pinned host buffers, kept with the scene) and returns;
 * rtx_render_end() waits for that slot's frame and copies it into the caller's arrays (either may
 * be NULL; hit ids only if asked for at begin).  With frame n+1 begun before frame n is ended, the
 * copy to host memory and the end of a frame's kernel overlap the next frame's render: about
 * 2.8 ms per 4K frame of the dragon scene delivered to host memory, against 4.0 ms call by call.
 *     rtx_render_begin(s, &p, 0, 0);
 *     for (n = 0; n < frames; n++) {
 *         if (n + 1 < frames) rtx_render_begin(s, &p, (n + 1) % 2, 0);
 *         rtx_render_end(s, n % 2, canvas, NULL);
 *     }
 * A slot holds one frame at a time: begin on a slot that was not ended is an error. */
int rtx_render_begin(rtx_scene *scene, const rtx_render_params *p, int32_t slot, int32_t want_hit_ids);
int rtx_render_end(rtx_scene *scene, int32_t slot, uint8_t *rgb_out, int32_t *hit_id_out);

/* Same frame into DEVICE buffers, asynchronous on `stream` (a hipStream_t, NULL =
 * default stream).  d_rgb_i32 (nullable): the unclamped Color ints, 3 per pixel
 * (the reference's canvas, src/main.cpp:16); d_t (nullable): primary closestT.
 * d_stats (nullable): device pointer to the 15 uint64 counters of rtx_stats (everything
 * before kernel_ms) that the kernel adds into. */
int rtx_render_device(rtx_scene *scene, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id,
                      void *d_rgb_i32, void *d_t, void *d_stats, void *stream);

/* The same launch, which also leaves its own duration on the device: d_ticks (nullable)
 * points to two uint64 that the CALLER HAS ZEROED (the call adds no clearing dispatch in front
 * of the frame); when the launch is over, d_ticks[0] holds the bitwise complement of the start of
 * the first block and d_ticks[1] the end of the last block of the render kernel on the device's
 * 100 MHz clock: duration = d_ticks[1] - ~d_ticks[0].  With several frames in flight, events on
 * a stream also count the wait for the chip; this does not.  No counterpart in the reference. */
int rtx_render_device_timed(rtx_scene *scene, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id,
                            void *d_rgb_i32, void *d_t, void *d_stats, void *d_ticks, void *stream);

/* The same frame through the STAGED pipeline: depth + 3 launches on `stream` instead of one.  rtx_render_device()'s
 * launch keeps a pixel on one lane from its camera ray to its last reflection (Scene::traceRay's recursion,
 * include/scene.hpp:89-123, as a chain of up to 2 (depth + 1) queries), so a frame launched alone ends with the chip
 * waiting for its longest pixels.  Here the rays are regrouped between launches: launch l traces the level-l
 * closest-hit queries of every pixel that still has one, together with the shadow queries of the level l-1 hits
 * (a hit's lighting and its reflection only meet in the blend, scene.hpp:113,119-122), lanes are refilled query by
 * query, and a last launch blends each pixel's levels and writes it.  Same pixels, ids, t and counters, bit for bit.
 * Costs 96 + 24 (depth + 1) bytes of HBM per output pixel (kept with the scene handle per frame in flight).
 * One sample per pixel only (RTX_ERR_UNSUPPORTED otherwise).  Arguments as rtx_render_device_timed(). */
int rtx_render_device_staged(rtx_scene *scene, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id,
                             void *d_rgb_i32, void *d_t, void *d_stats, void *d_ticks, void *stream);

/* Several frames in ONE launch: n_frames (1..8) frames of the scene with the same parameters, frame f into
 * d_rgb_u8 + f * rtx_output_pixels(p) * 3 (d_rgb_i32, nullable: + f * rtx_output_pixels(p) * 12).  `cameras`
 * (nullable) gives every frame its own camera, n_frames x {position[3], rotation in degrees[3]} in host memory,
 * read before the call returns; NULL renders the scene's camera n_frames times.  The reference renders frame after
 * frame (src/main.cpp:42-61); here a frame's last pixels, which leave most of the chip idle when a frame is
 * launched alone, overlap the next frame's first ones inside the one launch -- what frames in flight on several
 * streams achieve only while every stream has a hardware queue to itself.  d_stats / d_ticks as in
 * rtx_render_device_timed(), over the whole launch.  Hit ids and t are not written. */
int rtx_render_device_batch(rtx_scene *scene, const rtx_render_params *p, int32_t n_frames, const double *cameras,
                            void *d_rgb_u8, void *d_rgb_i32, void *d_stats, void *d_ticks, void *stream);

/* rtx_render_device() for callers that cannot allocate: tight serving loops, harnesses that forbid
 * hipMalloc.  No counterpart in the reference.  Same arguments, same
 * frame, bit for bit; but the launch's scratch (work counters, work-queue table, per-lane shading
 * state) lives in a 48 MiB array that is part of the library's device code, one copy per GPU for
 * the life of the process, and the call makes no hipMalloc / hipFree / event or stream creation
 * and no synchronisation: two kernel launches on `stream`, nothing else.  A stream that is
 * being captured into a HIP graph is refused (RTX_ERR_UNSUPPORTED): replay of the two launches is
 * not verified.
 * Limits: a frame whose work-queue table (4 bytes per owned tile + 4 per 8x8 group of tiles) does
 * not leave room for one block's lanes is refused with RTX_ERR_UNSUPPORTED -- about 8 million
 * tiles; at 32x32-pixel tiles no canvas this library accepts comes near -- and a launch that
 * needs more per-lane scratch than is left (depth > 3 on a full chip) runs with fewer resident
 * lanes: same pixels, somewhat slower.
 * Every call on one GPU uses that GPU's array: calls that overlap there (different streams, or
 * different scenes) race.  Keep them on one stream or order them yourself. */
int rtx_render_device_noalloc(rtx_scene *scene, const rtx_render_params *p, void *d_rgb_u8, void *d_hit_id,
                              void *d_rgb_i32, void *d_t, void *d_stats, void *stream);

/* After an all-gather of `n_ranks` packed tile buffers ([rank][local tile][tile_h][tile_w][3]),
 * writes the row-major frame.  Device pointers; asynchronous on `stream`.
 * No counterpart in the reference (its threads write one shared canvas). */
int rtx_untile(const rtx_render_params *p, int32_t n_ranks, const void *d_gathered_u8, void *d_rgb_u8,
               int32_t bytes_per_pixel, void *stream);

/* ---- one frame over several GPUs of a node ------------------------------------------- */

/* The reference spreads a frame over eight host threads that own column strips of one canvas
 * (src/main.cpp:34-52).  A scene group is that fan-out over GPUs: the scene replicated in the HBM
 * of every listed device, the frame cut into tile_w x tile_h screen tiles dealt round-robin
 * (tile k -> device k mod N), one RCCL all-gather of the devices' packed tile buffers over xGMI,
 * and the first device of the list ("root") writing the row-major frame.  One process, one host
 * thread, no Python; RCCL (librccl.so) is bound when the first group is created.
 * devices == NULL or n_devices <= 0: the first n_devices devices, or all of the node's. */
typedef struct rtx_scene_group rtx_scene_group;

int rtx_scene_group_create(const rtx_scene_desc *desc, const int32_t *devices, int32_t n_devices, rtx_scene_group **out);
int rtx_scene_group_create_from_host(const rtx_host_scene *hs, const int32_t *devices, int32_t n_devices,
                                     rtx_scene_group **out);
void rtx_scene_group_destroy(rtx_scene_group *group);
int32_t rtx_scene_group_size(const rtx_scene_group *group);
/* HIP ordinal and scene replica of member i (the replica stays owned by the group). */
int32_t rtx_scene_group_device(const rtx_scene_group *group, int32_t i);
rtx_scene *rtx_scene_group_scene(rtx_scene_group *group, int32_t i);

/* How the members' packed tile buffers come together.  RTX_GATHER_ALL (the default): one ncclAllGather, every
 * member ends with the whole frame's tiles.  RTX_GATHER_ROOT: ncclSend from every other member, ncclRecv on the
 * root: only the root needs the frame, and 1/N of the all-gather's bytes cross the links.  The environment
 * variable RTX_GROUP_GATHER=all|root sets a new group's mode. */
#define RTX_GATHER_ALL 0
#define RTX_GATHER_ROOT 1
int rtx_scene_group_set_gather(rtx_scene_group *group, int32_t mode);

/* The exchange as a list of operations, which is what the group executes (exposed so that the plan can be
 * checked without a second GPU): member `member` runs operation `kind` -- an all-gather of its `bytes`, a send of
 * `bytes` at src_offset of its packed buffer to member `peer`, a receive of `bytes` from `peer` at dst_offset
 * of its gathered buffer, or a local copy from src_offset of its packed buffer to dst_offset of its gathered
 * buffer.  Returns the number of operations written (at most 2 n_members), or a negative rtx_status. */
typedef struct rtx_gather_op {
    int32_t kind, member, peer, reserved;
    uint64_t src_offset, dst_offset, bytes;
} rtx_gather_op;
#define RTX_GATHER_OP_ALLGATHER 0
#define RTX_GATHER_OP_SEND 1
#define RTX_GATHER_OP_RECV 2
#define RTX_GATHER_OP_COPY 3
int32_t rtx_gather_plan(int32_t n_members, uint64_t local_bytes, int32_t mode, rtx_gather_op *ops, int32_t max_ops);

/* One frame into a HOST buffer (3 bytes per pixel, row-major), rendered by all members; replaces
 * the fan-out and join of src/main.cpp:45-52.  p->tile_rank, tile_count and layout are set by the
 * call; hit ids and counters are not gathered (render one member's scene for those). */
int rtx_render_sharded(rtx_scene_group *group, const rtx_render_params *p, uint8_t *rgb_out);

/* The same frame left on the root device, asynchronously: the work is queued on frame slot `slot`
 * (0..3, each with its own render streams and buffers, so up to four frames overlap; the
 * exchanges of all slots share one high-priority stream per member) and *d_rgb_u8
 * receives the root-device pointer of that slot's frame, complete after rtx_scene_group_sync().
 * A slot's next frame is ordered behind its previous one. */
int rtx_render_sharded_device(rtx_scene_group *group, const rtx_render_params *p, int32_t slot, void **d_rgb_u8);
/* Waits for frame slot `slot` on every member; negative = all slots. */
int rtx_scene_group_sync(rtx_scene_group *group, int32_t slot);

/* saveCanvas (src/main.cpp:96-116): binary P6, "P6\n<W> <H> 255\n" + rows. */
int rtx_save_ppm(const char *path, int32_t width, int32_t height, const uint8_t *rgb);
/* The same canvas as an 8-bit RGB PNG (uncompressed deflate blocks).  No counterpart in the
 * reference, whose PNGs under images/ were converted by hand. */
int rtx_save_png(const char *path, int32_t width, int32_t height, const uint8_t *rgb);

/* Device self-test of the slab test's division-by-ray-constant: n seeded operand pairs
 * (zeros, powers of two, all-ones significands, exponents over the admitted range) are
 * divided both ways and compared bitwise; *mismatches must come back 0.
 * No counterpart in the reference (it divides directly, include/bvh.hpp:218-229). */
int rtx_selftest_division(uint64_t n, uint64_t seed, uint64_t *mismatches);

/* Device self-test of the one libm call on the path whose bits are not pinned: pow(x, e) of
 * the specular term (include/scene.hpp:176) for n seeded cosines in (0, 1] and exponents
 * 2..1000, against the host's libm.  *differing: results that are not bit-equal; *max_ulps:
 * the largest distance in units of the last place.  Powers below the normal range (they add
 * nothing to a lighting sum) are left out.  (Exponent 1, the only one the shipped scenes
 * use, does not go through pow here.)  No counterpart in the reference. */
int rtx_selftest_pow(uint64_t n, uint64_t seed, uint64_t *differing, uint64_t *max_ulps);

/* Device self-test of the square root (Vector3::magnitude, include/vector3.hpp:84-86, in the shading terms
 * and in the Triangle constructor's normal, include/triangle.hpp:23): n seeded operands across the exponent
 * range, against the host's correctly rounded sqrt.  *differing must come back 0: the path's bit-exactness
 * rests on it (a GPU toolchain may offer a faster, less exact f64 sqrt).  No counterpart in the reference. */
int rtx_selftest_sqrt(uint64_t n, uint64_t seed, uint64_t *differing);

const char *rtx_last_error(void);
const char *rtx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
