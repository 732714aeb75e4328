/* TEST INFRASTRUCTURE ONLY -- the parity oracle.
 *
 * A scalar CPU restatement of the reference ray tracer's per-pixel render path
 * (khaganmv/ray-tracer).  It exists to CHECK the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (ray_tracer_amd/) never links, imports or executes anything here.
 *
 * Parity status: PINNED.  oracle/_ref (the reference's own headers compiled in
 * the build container, see ref_harness.cpp) and this restatement agree bit for
 * bit on BVH node/index arrays, primary hit ids, primary t and unclamped RGB for
 * the shipped teapot scene and for the stand-in scenes; the resulting hashes and
 * raw arrays are committed under tests/golden/ by tests/golden/make_golden.py.
 *
 * Every function in rt_oracle.cpp cites the reference file:line it follows.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_scene ora_scene;

/* Work counters over ALL rays of a render (closest-hit and shadow queries). */
typedef struct ora_census {
    uint64_t primary_rays;
    uint64_t reflection_rays;  /* closest-hit queries at recursion depth >= 1 */
    uint64_t shadow_rays;
    uint64_t shadow_occluded;
    uint64_t closest_hits;     /* closest-hit queries that found a triangle   */
    uint64_t node_visits;      /* slab tests (one per BVH::intersectRay call)  */
    uint64_t box_hits;         /* slab tests that passed                       */
    uint64_t leaf_visits;
    uint64_t triangle_tests;   /* Triangle::intersectRay calls                 */
    /* shadow queries whose light adds 0 to the surface whether it is visible or not (N.L <= 0 and no specular
     * term, scene.hpp:165-180), and those of them that found an occluder: what a renderer may leave untraced */
    uint64_t shadow_idle;
    uint64_t shadow_idle_occluded;
} ora_census;

/* Scene from raw arrays; the Triangle ctor work (normal, centroid) and the BVH
 * build happen inside.  lights: 4 doubles each {intensity, x, y, z}. */
ora_scene *ora_scene_create(int64_t n_triangles, const double *vertices9, const int32_t *color3,
                            const int32_t *specularity, const double *reflectivity,
                            const double viewport[3], const double camera_position[3],
                            const double camera_rotation_deg[3], const int32_t background[3],
                            double ambient, int32_t n_point, const double *point_lights4,
                            int32_t n_directional, const double *directional_lights4);

/* parseOBJ + one of the five scene factories ("teapot", "bunny", "erato",
 * "dragon", "aurelius"): camera, lights and the last-10-triangles material
 * overrides.  Returns NULL on failure (ora_last_error()). */
ora_scene *ora_scene_load_obj(const char *obj_path, const char *preset);

void ora_scene_free(ora_scene *s);
const char *ora_last_error(void);

int64_t ora_num_triangles(const ora_scene *s);
int64_t ora_num_nodes(const ora_scene *s);
int32_t ora_max_depth(const ora_scene *s);
/* nodes: 64-byte records {double bMin[3], bMax[3]; uint64 triangleFirst, triangleCount}. */
void ora_get_nodes(const ora_scene *s, void *out64);
void ora_get_indices(const ora_scene *s, uint64_t *out);
/* 15 doubles per triangle: v0 v1 v2 normal centroid. */
void ora_get_geometry(const ora_scene *s, double *out15);
/* materials after the factory overrides: color3 (int32), specularity, reflectivity */
void ora_get_materials(const ora_scene *s, int32_t *color3, int32_t *specularity, double *reflectivity);
/* {viewport3, position3, rotation_deg3, ambient} = 10 doubles; background 3 ints */
void ora_get_camera(const ora_scene *s, double *out10, int32_t *background3);

/* Render W x H (row 0 = top, as putPixel lays the canvas out).  Any output may
 * be NULL.  use_bvh = 0 runs the brute-force closestIntersection instead.
 * spp == 1 is the reference; spp > 1 is this build's jittered extension
 * (sample 0 has zero jitter; rgb_u8 = rounded mean of the clamped samples,
 * rgb_i32 = sum of the clamped samples; hit/t from sample 0).
 * Threads own contiguous column strips like the reference's render(). */
int ora_render(ora_scene *s, int32_t W, int32_t H, int32_t depth, int32_t spp, uint64_t seed,
               int32_t threads, int32_t use_bvh, int32_t *hit, int32_t *rgb_i32, uint8_t *rgb_u8,
               double *t, ora_census *census);

/* Jitter of sample s at pixel index k (k = row*W + col): writes offsets in
 * [0,1)^2; sample 0 is (0,0). */
void ora_jitter(uint64_t seed, uint64_t k, uint32_t s, double *jx, double *jy);

/* Single queries, for unit tests. */
int ora_closest_hit(ora_scene *s, const double o[3], const double r[3], double tmin, double tmax,
                    int32_t use_bvh, double *t_out, int32_t *id_out);

uint64_t ora_fnv1a(const void *data, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
