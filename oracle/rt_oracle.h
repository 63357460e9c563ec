/*
 * rt_oracle.h - C interface of the CPU ORACLE (test infrastructure, NOT product).
 *
 * The oracle is a CPU restatement of deadalusai/raytracer's render core
 * (raytracer-impl/src, all .rs files) used ONLY as the checker in tests/, in
 * __graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.  Nothing in
 * raytracer_amd/ may include, link or call it.
 *
 * PARITY PIN STATUS: the reference is Rust; no rustc/cargo exists in the build
 * container, so the reference itself cannot be run.  The reference's own tests
 * pin only `partition_by_key` (raytracer-impl/src/util.rs:41-63) and
 * `create_render_chunks` (raytracer-impl/src/viewport.rs:61-74); both are
 * checked in tests/test_oracle_kat.py.  Intersection, traversal, scatter,
 * shading and pixel values are "parity unpinned" by the reference: they are
 * pinned here by analytic known-answers and by published test vectors of the
 * third-party RNGs (rand_xorshift 0.4.0, Random123 Philox4x32-10).
 *
 * It consumes the same POD scene description as the product (include/rt_hip.h:
 * an interface, not product code).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcScene OrcScene;

/* RNG feeding `rng.random::<f32>()` (implementation.rs:17-18,649-650). */
enum {
    /* reference semantics: one XorShiftRng::seed_from_u64(seed) per chunk,
     * consumed sequentially in pixel order (src/render.rs:142-153)            */
    ORC_RNG_XORSHIFT_CHUNK = 0,
    /* reference semantics with the product's GENERATOR: every rng.random::<f32>()
     * of the reference (lens: implementation.rs:649-650, scatter: :17-18) is a draw
     * of a counter-based RNG, Philox4x32-10 addressed by (seed, x, y, sample, node,
     * purpose); independent of chunking.  The product in RT_OPT_LENS_SAMPLER 0.    */
    ORC_RNG_COUNTER = 1,
    /* NOT a restatement of the reference: ORC_RNG_COUNTER plus the product's
     * stratified lens sampler (the two lens draws of sample s are placed in one of
     * 8 x 8 cells over the lens square inside complete groups of 64 samples, of
     * 4 x 4 inside complete groups of 16 after them; include/rt_hip.h,
     * rt_render_chunk).  The product's default, RT_OPT_LENS_SAMPLER 1.  The one
     * place the oracle follows the product instead of the reference; it exists so
     * that the default configuration can be compared pixel by pixel too.          */
    ORC_RNG_COUNTER_STRATIFIED = 2
};

/* Work counters (per render call, added to). "ref_*" count what the reference's
 * non-shrinking iterator does (bvh.rs:210-236); "min_*" count an ordered,
 * t_max-shrinking closest-hit traversal of the same tree (SURVEY.md 8d:
 * algorithmic minimum) and are only filled when stats != 0. */
typedef struct OrcCounters {
    RtRayCounters rays;
    uint64_t ref_nodes, ref_spheres, ref_planes, ref_tris, ref_xforms;
    uint64_t min_nodes, min_spheres, min_planes, min_tris, min_xforms, min_instances;
    uint64_t min_mismatch;   /* rays where the shrinking traversal disagreed    */
} OrcCounters;

const char* orc_last_error(void);

/* Scene::new + add_* + build_bvh (implementation.rs:340-383). NULL on error. */
OrcScene* orc_scene_create(const RtSceneDesc* desc);
void      orc_scene_destroy(OrcScene* s);

/* src/render.rs:130-164 for one chunk. sum_segment: 0 = the reference's
 * sequential sample sum; k>0 = sum samples in blocks of k, then the block sums
 * in order (the product's association). stats!=0 also fills min_* counters. */
int orc_render_chunk(const OrcScene* s, const RtSettings* settings,
                     const RtRenderChunk* chunk, int rng_mode, uint64_t seed,
                     uint32_t sum_segment, int stats,
                     float* rgb_out, uint8_t* rgba_out, OrcCounters* counters);

/* Sum (not mean) of the clamped samples [sample_first, +sample_count) in one of
 * the two counter modes; the checker for rt_accumulate_chunk_device. */
int orc_accumulate_chunk(const OrcScene* s, const RtSettings* settings,
                         const RtRenderChunk* chunk, int rng_mode, uint64_t seed,
                         uint32_t sample_first, uint32_t sample_count,
                         float* accum_out);

/* Whole image with the reference's threading model (src/render.rs:186-224):
 * `threads` workers pull chunks (create_render_chunks(size,[cx,cy])) from a
 * shared queue. Returns wall milliseconds of the pixel loop in *elapsed_ms. */
int orc_render_image_mt(const OrcScene* s, const RtSettings* settings,
                        uint32_t count_x, uint32_t count_y, uint32_t threads,
                        int rng_mode, uint64_t seed,
                        float* rgb_out, uint8_t* rgba_out,
                        OrcCounters* counters, double* elapsed_ms);

/* Same threading model restricted to the chunks chunk_first + k*chunk_stride
 * (k < chunk_count) of the list: the bounded sample bench.py times. */
int orc_render_chunks_mt(const OrcScene* s, const RtSettings* settings,
                         uint32_t count_x, uint32_t count_y,
                         uint32_t chunk_first, uint32_t chunk_stride, uint32_t chunk_count, uint32_t threads,
                         int rng_mode, uint64_t seed, int stats,
                         OrcCounters* counters, double* elapsed_ms);

/* First hit_closest of the lens-centre ray of each pixel (same outputs as
 * rt_primary_hits). */
int orc_primary_hits(const OrcScene* s, const RtSettings* settings,
                     const RtRenderChunk* chunk,
                     float* t_out, float* normal_out,
                     int32_t* entity_out, int32_t* tri_out);

/* Diagnostics of OrcCounters.min_mismatch (stats mode): the rays on which the ordered, shrinking
 * traversal disagreed with the reference-order one.  `*_spans` = [entry, exit] along the ray, by
 * hit_aabb's arithmetic over the whole line, of three boxes of that winner: its entity box (world
 * space), its mesh's root box and its triangle's own box (entity space; 0 for non-mesh winners). */
typedef struct OrcMismatch {
    float o[3], d[3], t_min, t_max;
    int32_t ref_have, min_have;
    float ref_t, min_t;
    int32_t ref_ent, ref_tri, min_ent, min_tri;
    float ref_spans[6], min_spans[6];
} OrcMismatch;
/* start capturing up to `capacity` records (process-wide; 0 stops and clears) / copy them out */
/* factor (>= 1) by which the shrinking walk of stats mode widens its culling bound; default 1 + 2^-10 */
void     orc_set_cull_slack(float factor);
void     orc_mismatch_capture(uint32_t capacity);
uint32_t orc_mismatch_read(OrcMismatch* out, uint32_t capacity);

/* Candidate order of bvh.rs:210-236 when every box is hit (leaf enumeration).
 * mesh_index<0: TLAS. Returns count. */
int64_t orc_scene_bvh_order(const OrcScene* s, int64_t mesh_index,
                            uint32_t* out, uint64_t capacity);
int orc_scene_bvh_depth(const OrcScene* s, int64_t mesh_index);

/* ---- known-answer hooks (one per reference function) --------------------- */
/* util.rs:6-28 on i32 keys (the reference's own test type). Returns left len,
 * -1 on the empty-slice panic. */
int  orc_partition_by_key_i32(int32_t* data, int n, int32_t split_on);
/* viewport.rs:30-54 */
int  orc_create_render_chunks(uint32_t w, uint32_t h, uint32_t cx, uint32_t cy,
                              RtRenderChunk* out, uint32_t capacity);
/* sphere.rs:4-17: returns 1 and fills ts[2] when discriminant > 0 */
int  orc_intersect_sphere(const float* origin, const float* dir, float radius, float* ts);
/* sphere.rs:36-53: full Sphere::hit -> t (or -1) and normal */
float orc_sphere_hit(const float* origin, const float* dir, float radius,
                     float t_min, float t_max, float* normal_out);
/* mesh.rs:18-76. Returns 1 on hit; out = t, p[3], normal[3], uv[2] */
int  orc_try_hit_tri(const float* origin, const float* dir, float t_min, float t_max,
                     const RtMeshTri* tri, float* out9);
/* plane.rs:4-17: returns 1 + *t */
int  orc_intersect_plane(const float* origin, const float* dir,
                         const float* plane_origin, const float* normal, float* t);
/* implementation.rs:92-118 */
int  orc_hit_aabb(const float* bmin, const float* bmax, const float* origin,
                  const float* dir, float t_min, float t_max);
/* materials.rs:84-87,150-165 */
void  orc_reflect(const float* incident, const float* normal, float* out);
void  orc_refract(const float* v, const float* n, float ni_over_nt, float* out);
float orc_schlick(float cosine, float ref_idx);
/* implementation.rs:506-512 */
void  orc_color_sky_day(const float* dir, float* out);
/* types.rs:191-201 */
void  orc_rotate_about_axis(const float* p, const float* axis, float theta, float* out);
/* src/rgba.rs:6-12 */
void  orc_v3_to_rgba(const float* rgb, uint8_t* out);
/* implementation.rs:462-491: Camera::new + get_ray */
void  orc_camera_new(const float* look_from, const float* look_at, float v_fov,
                     float aspect, float lens_radius, float focus_dist, RtCamera* out);
void  orc_camera_get_ray(const RtCamera* cam, float x, float y, float lx, float ly,
                         float* origin_out, float* dir_out);
/* lights.rs: returns 0 when the light gives None; out = t, dir[3], color[3], intensity */
int   orc_light_record(const RtLight* light, const float* p, float* out8);

/* rand_xorshift 0.4.0 XorShiftRng::from_seed(16 bytes) / seed_from_u64 then n
 * next_u32 outputs. */
void orc_xorshift_from_seed(const uint8_t* seed16, uint32_t* out, int n);
void orc_xorshift_seed_from_u64(uint64_t seed, uint32_t* out, int n);
/* rand 0.9 StandardUniform f32 of one u32 */
float orc_u32_to_f32(uint32_t v);
/* Random123 Philox4x32-10 block */
void orc_philox4x32_10(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4);
/* the product's counter RNG: draw `k` of context (seed,x,y,sample,node,purpose) */
float orc_counter_rng_f32(uint64_t seed, uint32_t x, uint32_t y, uint32_t sample,
                          uint32_t node, uint32_t purpose, uint32_t k);

#ifdef __cplusplus
}
#endif
#endif
