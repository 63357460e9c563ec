/*
 * rt_hip.h - C ABI of the MI355X render core (librt_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of deadalusai/raytracer: the
 * per-pixel render loop of crate `raytracer-impl`
 * (raytracer-impl/src/implementation.rs:638-661 and everything it calls) as
 * driven by the worker loop of the UI crate (src/render.rs:130-164).
 *
 * The reference has no FFI; the seam is the Rust function boundary between
 * `raytracer-ui` and `raytracer-impl`.  Every entry point below cites the
 * reference interface it replaces.  Plain pointers and sizes only.
 *
 * Conventions
 *   - all functions returning `int` return 0 on success, a negative RT_E_* code
 *     otherwise; rt_last_error() then gives a thread-local message.  Nothing
 *     throws or aborts across this boundary.
 *   - `RtScene*` is opaque, immutable after creation and may be used from
 *     several host threads at once.  The host-pointer entry points
 *     (rt_render_chunk, rt_primary_hits) each borrow a private workspace and
 *     stream (up to 8 per scene, later callers wait), so the reference's worker
 *     threads (src/render.rs:186-224) overlap on the GPU.  The *_device entry
 *     points enqueue on the CALLER's stream and share one workspace per scene:
 *     keep them on one stream (or use one scene per stream).
 *   - output buffers are caller allocated.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

/* ---- error codes -------------------------------------------------------- */
#define RT_OK            0
#define RT_E_INVALID    -1   /* bad argument / malformed scene description     */
#define RT_E_HIP        -2   /* a HIP runtime call failed                      */
#define RT_E_NO_DEVICE  -3   /* no GPU visible                                  */
#define RT_E_DEPTH      -4   /* BVH deeper than the traversal stack can hold
                                (reference: ArrayVec<State,30> panic, bvh.rs:192).
                                Found at scene creation; a kernel that nevertheless
                                runs out of stack or deferred-ray slots sets a
                                sticky per-scene error: the blocking entry points
                                return it for the call itself, the *_device ones
                                (which only enqueue) for every LATER call on the scene */
#define RT_E_IO         -5   /* file could not be read (sample scenes / OBJ)    */
#define RT_E_LIMIT      -6   /* size beyond a documented maximum                */
#define RT_E_CANCELLED  -7   /* rt_scene_cancel was called: the call ended early
                                (or was refused), output pixels are unfinished   */

/* ---- POD vector types (types.rs:11-12, :254-255) ------------------------ */
typedef struct RtV3 { float x, y, z; } RtV3;
typedef struct RtV2 { float x, y; } RtV2;

/* ---- Camera (implementation.rs:451-459): the already-derived basis ------ */
typedef struct RtCamera {
    RtV3  lower_left_corner;
    RtV3  horizontal;
    RtV3  vertical;
    RtV3  origin;
    RtV3  u;
    RtV3  v;
    float lens_radius;
} RtCamera;

/* ---- Hitables (shapes/sphere.rs:19-23, plane.rs:19-26, mesh.rs:104-148) -- */
enum { RT_HITABLE_SPHERE = 0, RT_HITABLE_PLANE = 1, RT_HITABLE_MESH = 2 };

typedef struct RtSphere {
    float    radius;
    uint32_t mat_id;
    uint32_t tex_id;
} RtSphere;

typedef struct RtPlane {
    RtV3     normal;      /* unit */
    RtV3     u_basis;
    RtV3     v_basis;
    float    radius;      /* used only when has_radius != 0 (disk) */
    uint32_t has_radius;
    uint32_t mat_id;
    uint32_t tex_id;
} RtPlane;

/* MeshTri (mesh.rs:114-123). tex_key: -1 = None. */
typedef struct RtMeshTri {
    RtV3    a, b, c;
    RtV2    a_uv, b_uv, c_uv;
    int32_t tex_key;
} RtMeshTri;

/* Mesh (mesh.rs:104-106): geometry shared by every MeshObject that names it
 * (the Arc<MeshBvhRoot> of mesh.rs:141-158 => instancing). */
typedef struct RtMesh {
    const RtMeshTri* tris;
    uint64_t         tri_count;
} RtMesh;

typedef struct RtMeshObject {
    uint32_t mesh_index;  /* into RtSceneDesc.meshes */
    uint32_t mat_id;
    uint32_t tex_id;
} RtMeshObject;

/* ---- Entity (implementation.rs:213-230) --------------------------------- */
typedef struct RtRotation { RtV3 axis; float theta; } RtRotation;

typedef struct RtEntity {
    uint32_t hitable_kind;      /* RT_HITABLE_*                               */
    uint32_t hitable_index;     /* into spheres / planes / mesh_objects       */
    int32_t  id;                /* Entity.id, -1 = None                       */
    uint32_t translation_first; /* into RtSceneDesc.translations              */
    uint32_t translation_count;
    uint32_t rotation_first;    /* into RtSceneDesc.rotations                 */
    uint32_t rotation_count;
} RtEntity;

/* ---- Materials (materials.rs:26-28, :56-59, :115-119) -------------------- */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_SPECULAR = 1, RT_MAT_DIELECTRIC = 2 };

typedef struct RtMaterial {
    uint32_t kind;
    float    reflectivity;  /* Lambertian.reflectivity / Specular.reflectiveness / Dielectric.reflectivity */
    float    fuzz;          /* Specular.fuzz       */
    float    opacity;       /* Dielectric.opacity  */
    float    ref_index;     /* Dielectric.ref_index */
} RtMaterial;

/* ---- Textures (texture.rs) ----------------------------------------------- */
enum {
    RT_TEX_COLOR = 0,        /* ColorTexture(V3)            texture.rs:9      */
    RT_TEX_CHECKER = 1,      /* CheckerTexture{size,odd,even} texture.rs:20   */
    RT_TEX_UV_TEST = 2,      /* UvTestTexture               texture.rs:49     */
    RT_TEX_XYZ_TEST = 3,     /* XyzTestTexture(f32)         texture.rs:58     */
    RT_TEX_COLOR_MAP = 4,    /* ColorMap{width,height,pixels} texture.rs:76   */
    RT_TEX_MESH = 5,         /* MeshTexture{Ka,Kd,map}      texture.rs:102    */
    RT_TEX_MESH_SET = 6      /* MeshTextureSet{textures}    texture.rs:120    */
};

typedef struct RtTexture {
    uint32_t kind;
    RtV3     color;       /* COLOR: value; MESH: diffuse_color                 */
    RtV3     color2;      /* MESH: ambient_color                               */
    float    size;        /* CHECKER: size; XYZ_TEST: max                      */
    int32_t  child0;      /* CHECKER: odd tex; MESH: color map tex (-1 = None);
                             MESH_SET: first entry in RtSceneDesc.tex_set_entries */
    int32_t  child1;      /* CHECKER: even tex; MESH_SET: entry count          */
    uint32_t width;       /* COLOR_MAP                                         */
    uint32_t height;      /* COLOR_MAP                                         */
    uint64_t texel_first; /* COLOR_MAP: first texel in RtSceneDesc.texels      */
} RtTexture;

/* ---- Lights (lights.rs:9-13, :52-58, :112-116) --------------------------- */
enum { RT_LIGHT_POINT = 0, RT_LIGHT_LAMP = 1, RT_LIGHT_DIRECTIONAL = 2 };

typedef struct RtLight {
    uint32_t kind;
    RtV3     origin;      /* POINT, LAMP                                       */
    RtV3     direction;   /* LAMP, DIRECTIONAL (already unit)                  */
    RtV3     color;
    float    intensity;
    float    angle_deg;   /* LAMP                                              */
} RtLight;

enum { RT_SKY_DAY = 0, RT_SKY_BLACK = 1 };

/* ---- Scene (implementation.rs:315-324), flattened to POD arrays ---------- *
 * What a `describe()` over the reference's private Scene fields would emit.  */
typedef struct RtSceneDesc {
    uint32_t abi_version;   /* RT_ABI_VERSION */
    RtCamera camera;
    uint32_t sky;           /* RT_SKY_* */

    const RtEntity*     entities;      uint64_t entity_count;
    const RtV3*         translations;  uint64_t translation_count;
    const RtRotation*   rotations;     uint64_t rotation_count;

    const RtSphere*     spheres;       uint64_t sphere_count;
    const RtPlane*      planes;        uint64_t plane_count;
    const RtMeshObject* mesh_objects;  uint64_t mesh_object_count;
    const RtMesh*       meshes;        uint64_t mesh_count;

    const RtMaterial*   materials;     uint64_t material_count;
    const RtTexture*    textures;      uint64_t texture_count;
    const int32_t*      tex_set_entries; uint64_t tex_set_entry_count; /* texture ids */
    const RtV3*         texels;        uint64_t texel_count;
    const RtLight*      lights;        uint64_t light_count;
} RtSceneDesc;

/* ---- RenderSettings (implementation.rs:332-337) --------------------------- */
typedef struct RtSettings {
    uint32_t width;
    uint32_t height;
    uint32_t max_reflections;
    uint32_t samples_per_pixel;
} RtSettings;

/* ---- RenderChunk (viewport.rs:5-11) --------------------------------------- */
typedef struct RtRenderChunk {
    uint32_t left, top, width, height;
} RtRenderChunk;

/* Ray counters. A ray = one Scene::hit_closest call (implementation.rs:372):
 * primary (:567 at top level), secondary (:594,:602), shadow incl. continuation
 * through transparent occluders (:535).  The reference keeps no such counter. */
typedef struct RtRayCounters {
    uint64_t primary;
    uint64_t secondary;
    uint64_t shadow;
    uint64_t samples;
} RtRayCounters;

typedef struct RtScene RtScene;

/* ---- entry points --------------------------------------------------------- */

/* Number of visible GPUs (no counterpart; the reference is CPU-only). */
int rt_device_count(void);

/* Thread-local message for the last failing call on this thread.
 * Replaces the reference's panics / CreateSceneError(String)
 * (src/job_constructing.rs:45-53). */
const char* rt_last_error(void);

/* Replaces Scene::build_bvh (implementation.rs:377-383) + MeshBvhRoot::new
 * (mesh.rs:84-90): validates the description, builds the TLAS over entities and
 * one BLAS per mesh with the reference's algorithm (bvh.rs:129-188,
 * util.rs:6-28), flattens everything and uploads it to `device`.
 * Returns NULL on failure.  Limits (RT_E_LIMIT): fewer than 2^26 tree records, 2^26 triangles (distinct, not
 * instanced) and 2^26 entities - the kernels address them through 32-bit byte offsets. */
RtScene* rt_scene_create(const RtSceneDesc* desc, int device);
/* device == RT_DEVICE_HOST_ONLY: validate and build the acceleration structures
 * but upload nothing; such a scene only serves rt_scene_info / rt_scene_bvh_order
 * (render entry points return RT_E_NO_DEVICE). */
#define RT_DEVICE_HOST_ONLY (-1)
void     rt_scene_destroy(RtScene* scene);

/* Animated scenes: new entity transforms for an existing scene (the reference rebuilds everything per
 * frame: Scene::build_bvh, implementation.rs:377-383, e.g. raytracer-samples/src/scene_interceptor_spin.rs:34-40).
 * `translations` / `rotations` replace the arrays given to rt_scene_create and must have the same sizes; every entity keeps
 * its ranges in them.  Entity boxes (Entity::calculate_origin_aabb, implementation.rs:258-283) and the TLAS boxes above
 * them are recomputed ON THE DEVICE (refit: the tree keeps the topology, leaves and visit ranks it was built with, BLASes
 * are untouched); the host only evaluates sin / cos of the rotation angles, with the libm the reference calls per ray.
 * Hits are those of a fresh rt_scene_create of the moved scene, except which of two entities wins an EXACT tie in t (the
 * visit ranks are those of the original tree).  RT_E_INVALID (nothing changed): some rotation axis is not of unit length - such an
 * entity's box does not bound its hits, and what the reference then reports depends on the leaves of the tree it builds for the
 * new transforms: create the scene anew.  Enqueued on `stream`, then waits for it (the host needs the new root box).
 * No render call on this scene may be in flight. */
int rt_scene_update_transforms(RtScene* scene, const RtV3* translations, uint64_t translation_count,
                               const RtRotation* rotations, uint64_t rotation_count, void* stream);
/* The same call, but with a NEW top-level tree: a linear BVH over the Morton codes of the moved entities' box centres, built
 * on the device (raytracer_amd/csrc/tlas_build.h; the reference's Scene::build_bvh, implementation.rs:377-383 over
 * bvh.rs:129-188, is a CPU mean-split build per frame).  For animations in which entities travel: a refitted tree keeps
 * the grouping of the first frame and degrades.  The tree differs from the reference's, so candidates are pruned by other
 * boxes: the nearest hit is the reference's except where a ray grazes the boundary of an entity's own box (f32 rounding),
 * exact ties keep the reference's visit ranks.  After this call rt_scene_update_transforms refits THIS tree (same gate as
 * the build) and rt_scene_bvh_order(-1) still reports the tree of rt_scene_create.  The first call allocates
 * (about 150 bytes per entity + 64 per entity of node records).  Enqueued on `stream`, then waits for it.  No render call
 * on this scene may be in flight.  RT_E_INVALID (nothing changed): some rotation axis is not of unit length - such an entity's box
 * does not bound its hits, and only the reference's own tree reproduces what the reference then reports.  RT_E_DEPTH: the new tree is deeper than the
 * LDS traversal stack can hold (64 levels); RT_E_LIMIT: its records would exceed the 2^26 the walk addresses.  Both are found BEFORE anything a render
 * call reads is written: a refused call leaves the scene - transforms, boxes, tree - exactly as it was (only rt_scene_update_transforms on a
 * device-built tree is then refused until the next successful rebuild: the working buffers hold the refused topology).  A HIP failure after the
 * commit has begun (none is expected) marks the scene unusable: every later call returns RT_E_HIP. */
int rt_scene_rebuild_tlas(RtScene* scene, const RtV3* translations, uint64_t translation_count,
                          const RtRotation* rotations, uint64_t rotation_count, void* stream);

/* Deforming meshes: new vertex positions (and uvs / texture keys) for ONE mesh of an existing scene, same triangle count, and a NEW tree
 * over them built on the device.  The reference builds a mesh's BVH on the CPU whenever a MeshObject is made (MeshBvhRoot::new,
 * raytracer-impl/src/shapes/mesh.rs:84-90, over bvh.rs:129-188); a mesh that moves every frame would pay that - and a whole
 * rt_scene_create - every frame.  `tris` is the mesh as rt_scene_create takes it (RtMesh.tris, original triangle order; host memory).
 * On the device: triangle boxes (MeshTri::calculate_bounds, mesh.rs:127-133) -> Morton keys -> rocPRIM radix sort -> triangle records
 * in key order (the precomputed cross(b - a, c - a) of mesh.rs:22-24 with the host build's operations) -> a linear BVH over them
 * (raytracer_amd/csrc/tlas_build.h; leaves of up to two triangles, like the reference's) -> two-box records into the mesh's slots of
 * the LDS node cache and behind the scene's records -> the mesh's root box -> boxes of its instances and the TLAS boxes above them
 * refitted (as rt_scene_update_transforms does).  Like rt_scene_rebuild_tlas this sits behind the NEAREST-HIT gate, not the bit-exact
 * one: the tree is not the reference's, so a candidate is pruned by other boxes; every triangle test is the reference's, exact ties in
 * t keep the visit ranks of the tree rt_scene_create built (by original triangle index).  Enqueued on `stream`, then waits for it.  No
 * render call on this scene may be in flight.  The first call per mesh allocates (about 330 bytes per triangle; the first rebuild of any
 * kind on a scene also re-allocates the node array with room for every rebuilt tree).  RT_E_DEPTH / RT_E_LIMIT are found before anything
 * a render call reads is written; rt_scene_bvh_order keeps reporting the trees of rt_scene_create. */
int rt_scene_rebuild_blas(RtScene* scene, uint32_t mesh_index, const RtMeshTri* tris, uint64_t tri_count, void* stream);

/* Switches of the sampler and of the launch schedule of one scene (no reference counterpart except
 * RT_OPT_LENS_SAMPLER 0).  Takes effect for calls that start afterwards.  Defaults can be preset through the
 * environment, read once in rt_scene_create: RT_LENS_STRATIFY, RT_FINE_SPLIT, RT_ROUND_UNITS, RT_LENS_COC. */
enum {
    /* 0: the two lens draws of a sample are independent, as in the reference (implementation.rs:648-653);
     * 1 (default): stratified - see rt_render_chunk.  Changes the image (same expectation). */
    RT_OPT_LENS_SAMPLER = 1,
    /* work items of a launch: -1 (default) chosen per call, 0 one pixel x 8 samples, 1 one pixel x 1 sample.
     * Never changes a bit of the image. */
    RT_OPT_FINE_SPLIT = 2,
    /* 1, 2, 4, 8 (default): how many 8-sample units of a pixel share a 64-lane round.  Schedule only. */
    RT_OPT_ROUND_UNITS = 3,
    /* circle of confusion (pixels, default 3) from which a tile is traced in rounds of one lens cell.  Schedule only. */
    RT_OPT_LENS_COC_PX = 4,
    /* 1 (default): 8 x 8 tiles whose camera rays ALL miss every leaf box of the scene's trees - a packet test: the beam of the tile's rays,
     * every lens sample of every pixel, walks the TLAS and the BLAS of the instances it meets, with a margin far above the rounding of the
     * per-ray box test - are not traced under a black sky: such a ray has no candidate (bvh.rs:210-236), hit_closest returns None and the sample
     * is color_sky_black's (0, 0, 0) (implementation.rs:517-519); their camera rays are counted (the reference casts them).  Under the day sky,
     * whose colour depends on the ray (:502-515), such tiles are rendered sample by sample but their camera rays skip the walk (scenes without
     * mesh instances only).  Not for scenes
     * with rotation axes that are not of unit length, nor with max_reflections 0.
     * 0: every sample is traced.  Never changes a bit of the image or a counter (rt_scene_culled_pixels says how many pixels were skipped). */
    RT_OPT_CULL_EMPTY_TILES = 5,
    /* tests of RT_E_DEPTH: run the kernels with a traversal stack of this many entries (0 = what the trees
     * need) and clear the scene's sticky overflow error.  Synchronises the device. */
    RT_OPT_DEBUG_STACK_ENTRIES = 100,
    /* tests of rt_scene_rebuild_tlas's refusal: a rebuilt tree that needs more traversal-stack entries than this is refused
     * with RT_E_DEPTH as if it exceeded the LDS limit (0 = the LDS limit only). */
    RT_OPT_DEBUG_REBUILD_STACK_LIMIT = 101
};
int rt_scene_set_option(RtScene* scene, uint32_t option, double value);

/* What RT_OPT_CULL_EMPTY_TILES did in the latest render call on this scene (both outputs nullable; 0 when the call's tiles were not
 * classified: small chunks, single-sample work items):
 *   left_out     : pixels in empty tiles under a black sky - never rendered, (0, 0, 0) by construction;
 *   without_walk : pixels in empty tiles under a sky whose colour depends on the ray - rendered sample by sample, their camera rays skip the walk.
 * No reference counterpart: the reference walks its trees for every ray (bvh.rs:210-236); this reports how much of that the packet test
 * replaced.  Readable while the call runs (host-mapped words). */
int rt_scene_culled_pixels(const RtScene* scene, uint64_t* left_out, uint64_t* without_walk);

/* Replaces the whole `for p in chunk.iter_pixels() { cast_rays_into_scene(..);
 * put_pixel(v3_to_rgba(..)) }` loop of src/render.rs:148-159 for one chunk
 * (chunk granularity: a per-pixel FFI call cannot feed a GPU).
 *   rgb_out  : chunk.width*chunk.height*3 floats, row-major, the V3 returned
 *              by cast_rays_into_scene (implementation.rs:638-661); nullable
 *   rgba_out : chunk.width*chunk.height*4 bytes, v3_to_rgba (src/rgba.rs:6-12);
 *              nullable
 *   seed     : replaces XorShiftRng::seed_from_u64(12345) (render.rs:142) by
 *              the key of a counter-based RNG addressed (seed,x,y,sample,node):
 *              the image does not depend on chunking or GPU count.  The two
 *              lens draws of a sample (implementation.rs:649-650) are stratified:
 *              8 x 8 cells over the lens square inside every complete group of 64
 *              samples, one sample per cell; 4 x 4 cells inside every complete group
 *              of 16 after the last of those; 4 x 2 cells for a complete group of 8
 *              after that: same expectation as the reference's independent draws,
 *              less noise (RT_OPT_LENS_SAMPLER 0 selects the reference's
 *              distribution).
 *   counters : nullable; rays are ADDED to it.
 * Host pointers; the call blocks until the pixels are in the buffers.
 * Limits (RT_E_LIMIT): image sides <= 65535, max_reflections <= 31, at most 2^28
 * pixels per chunk call (render larger images chunk by chunk). */
int rt_render_chunk(RtScene* scene, const RtSettings* settings,
                    const RtRenderChunk* chunk, uint64_t seed,
                    float* rgb_out, uint8_t* rgba_out, RtRayCounters* counters);

/* Same, but outputs are DEVICE pointers on the scene's device and the work is
 * only enqueued on `stream` (a hipStream_t, NULL = default stream); nothing is
 * copied to the host.  `pitch_pixels` is the row pitch of the output images in
 * pixels and (`out_x`,`out_y`) the position of the chunk inside them, so tiles
 * can be written straight into a full framebuffer.
 * `counters_dev` (nullable) is a device RtRayCounters that is added to. */
int rt_render_chunk_device(RtScene* scene, const RtSettings* settings,
                           const RtRenderChunk* chunk, uint64_t seed,
                           void* rgb_dev, void* rgba_dev,
                           uint32_t pitch_pixels, uint32_t out_x, uint32_t out_y,
                           void* counters_dev, void* stream);

/* rt_render_chunk_device for callers that cannot allocate.  NO counterpart in the reference: it exists for
 * code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same arguments, same
 * pixels (bit for bit) as rt_render_chunk_device, but the call allocates nothing (no hipMalloc / hipFree, no
 * events): its scratch - segment sums, chunk accumulator, block counters, tile order - is carved out of ONE
 * static array of 32 MiB in device memory that every GPU the library runs on holds until the process ends.
 * Limits: a chunk of at most rt_noalloc_max_pixels() pixels after padding to whole 8x8 tiles (about 1.34 M,
 * e.g. 1024 x 1024; RT_E_LIMIT beyond); chunks above a sixteenth of that run with fewer sample segments per
 * launch (slower, same bits).  Every call on a GPU shares that GPU's array: calls that overlap on one GPU
 * (different streams, or different scenes on one device) race - keep them on one stream.  The scene's
 * rt_scene_profile_* hook does not see these launches. */
int rt_render_chunk_device_noalloc(RtScene* scene, const RtSettings* settings,
                                   const RtRenderChunk* chunk, uint64_t seed,
                                   void* rgb_dev, void* rgba_dev,
                                   uint32_t pitch_pixels, uint32_t out_x, uint32_t out_y,
                                   void* counters_dev, void* stream);
uint64_t rt_noalloc_max_pixels(void);

/* Tile-partitioned rendering for one of N GPUs (partitioning 1).
 * Replaces the shared chunk queue of the reference's worker threads
 * (src/render.rs:186-224, 63-78): the frame is cut into horizontal strips of
 * `strip_height` rows (create_render_chunks(size,[1,n]) with equal strips) and this
 * call renders, in ONE launch, the strips strip_first, strip_first+strip_period, ...
 * `strip_height` must be a multiple of 8 (the kernel's 8x8 ray tiles; RT_E_INVALID otherwise).
 * Outputs (device pointers, nullable) are PACKED: rt_strip_rows() rows of
 * settings.width pixels, the owned strips one after the other.  Pixels are
 * identical to those of rt_render_chunk for the same image rows. */
int rt_render_strips_device(RtScene* scene, const RtSettings* settings,
                            uint32_t strip_height, uint32_t strip_first, uint32_t strip_period,
                            uint64_t seed, void* rgb_dev, void* rgba_dev,
                            void* counters_dev, void* stream);
/* Rows owned by (strip_first, strip_period) in an image of `height` rows; < 0 on bad arguments. */
int rt_strip_rows(uint32_t height, uint32_t strip_height, uint32_t strip_first, uint32_t strip_period);

/* The finer tile partition (same role as rt_render_strips_device: the share of one of N GPUs of the reference's chunk
 * queue, src/render.rs:186-224, 63-78; create_render_chunks(size,[n,1]) cut 8 pixels wide and dealt round-robin): this
 * call renders, in ONE launch, the 8-pixel-wide tile columns col_first, col_first + col_period, ... over the whole height
 * of the frame.  Every rank then holds a share of every region of the image, where 8-row strips cut a glass object or a
 * dense mesh unevenly (the slowest of 8 ranks decides the frame time).  Outputs (device pointers, nullable) are PACKED:
 * settings.height rows of rt_tile_columns_width() pixels, the owned columns side by side (the frame's last tile column may
 * be narrower than 8).  Pixels are identical to those of rt_render_chunk for the same image positions. */
int rt_render_tile_columns_device(RtScene* scene, const RtSettings* settings,
                                  uint32_t col_first, uint32_t col_period,
                                  uint64_t seed, void* rgb_dev, void* rgba_dev,
                                  void* counters_dev, void* stream);
/* Packed width in pixels of the tile columns owned by (col_first, col_period) in a frame `width` pixels wide
 * (0: none); < 0 on bad arguments. */
int rt_tile_columns_width(uint32_t width, uint32_t col_first, uint32_t col_period);

/* spp-split rendering (partitioning 2): adds the SUM of the
 * per-sample clamped colours (implementation.rs:635,656) of samples
 * [sample_first, sample_first+sample_count) to accum_dev (float3 per pixel,
 * full-frame pitch = settings.width), without dividing.  Sample indices address
 * the counter-based RNG, so N ranks rendering disjoint ranges reproduce the
 * samples of one rank rendering all of them. */
int rt_accumulate_chunk_device(RtScene* scene, const RtSettings* settings,
                               const RtRenderChunk* chunk, uint64_t seed,
                               uint32_t sample_first, uint32_t sample_count,
                               void* accum_dev, void* counters_dev, void* stream);

/* One frame on N GPUs, composited on the first.  Replaces the reference's pool of worker threads over one shared chunk
 * queue (start_background_render_threads, src/render.rs:186-224; RenderJob::update's hand-out, :63-78) ACROSS devices, for
 * a host that is one process (the egui front end): `scenes` are n_scenes replicas of the same scene, one RtScene per
 * device (rt_scene_create(desc, device); two replicas may also share a device).  One host thread per replica enqueues
 * that replica's share on a stream of its own; the call returns when the frame is complete in GPU 0's memory
 * (GPU 0 = scenes[0]'s device; rgb_dev0 / rgba_dev0 are device pointers THERE, width * height pixels, either nullable).
 *   RT_MULTI_TILES: 8-pixel tile columns dealt round-robin (replica i owns columns i, i + n, ...), every replica renders
 *                   all its columns in one launch (rt_render_tile_columns_device) and writes them straight into GPU 0's
 *                   frame over its own xGMI link (peer access), or - where peer access is refused - packs them, copies the
 *                   packed image over (hipMemcpyPeerAsync) and GPU 0 puts the columns in place.  The frame is that of ONE
 *                   rt_render_chunk_device call, bit for bit.
 *   RT_MULTI_SPP  : every replica renders the whole frame for its range of whole sample segments into an accumulator
 *                   (rt_accumulate_chunk_device); the accumulators are copied to GPU 0 and added there IN REPLICA ORDER,
 *                   then resolved (rt_resolve_device).  Equal to the one-call frame up to the association of the final
 *                   sums of segment sums (1 ulp-scale); bit-identical for a given n_scenes.
 * `counters` (host, nullable) is ADDED to.  The streams, events and staging buffers are cached in scenes[0] and freed
 * with it.  One frame at a time per process.  The same RtScene twice is RT_E_INVALID. */
#define RT_MULTI_TILES 0
#define RT_MULTI_SPP   1
int rt_render_frame_multi(RtScene* const* scenes, uint32_t n_scenes, const RtSettings* settings,
                          uint32_t mode, uint64_t seed, void* rgb_dev0, void* rgba_dev0,
                          RtRayCounters* counters);

/* The same frame into HOST buffers (the pair rt_render_chunk / rt_render_chunk_device again): what a host without a HIP
 * allocator of its own - the egui front end wants RGBA bytes for its texture, src/job_running.rs:37-41 - calls.  The
 * frame is rendered into buffers scenes[0] keeps on its device and copied out; rgb_out (width * height * 3 floats) and
 * rgba_out (width * height * 4 bytes) are nullable. */
int rt_render_frame_multi_host(RtScene* const* scenes, uint32_t n_scenes, const RtSettings* settings,
                               uint32_t mode, uint64_t seed, float* rgb_out, uint8_t* rgba_out,
                               RtRayCounters* counters);

/* Cancellation and progress INSIDE a call.  The reference checks its cancellation token per work item and per pixel
 * (src/render.rs:131,149) and sends a snapshot of the chunk every 800 ms (:146,156-158); a GPU call of seconds (a 4K frame
 * at 4096 spp runs 10 s) needs the same two things from another host thread.
 *   rt_scene_cancel(scene, 1): every launch on the scene - running or enqueued - stops handing out work: a wave reads the
 *     scene's host-mapped cancel word at its first and then every 4th block claim, so a running kernel drains within a few
 *     blocks per wave (milliseconds).  The blocking entry points (rt_render_chunk, rt_render_frame_multi) then return
 *     RT_E_CANCELLED; enqueue-only entry points return it for calls made while the scene is cancelled.  Output pixels of a
 *     cancelled call are unfinished.  The scene stays usable: rt_scene_cancel(scene, 0) re-arms it (call it once the
 *     cancelled work has drained - after the blocking call returned, or after synchronising the stream).
 *   rt_scene_progress: blocks_total = 64-ray-tile x sample-unit blocks of the call in flight on this scene (all its
 *     launches), blocks_done = about how many of them have been handed out (written by the waves at their poll points;
 *     an estimate that only grows within a launch).  Reads host memory only: callable at any time from any thread.
 *     With several calls in flight on one scene the numbers are those of the latest. */
int rt_scene_cancel(RtScene* scene, int cancel);
int rt_scene_progress(const RtScene* scene, uint64_t* blocks_done, uint64_t* blocks_total);

/* Finishes an accumulator: rgb = accum / spp (implementation.rs:659) and
 * rgba = v3_to_rgba(rgb) (src/rgba.rs:6-12) for n_pixels pixels. Device pointers;
 * rgb_dev / rgba_dev nullable. */
int rt_resolve_device(int device, const void* accum_dev, uint64_t n_pixels,
                      uint32_t samples_per_pixel, void* rgb_dev, void* rgba_dev,
                      void* stream);

/* Parity gate 1: the first Scene::hit_closest(ray, BIAS, f32::MAX)
 * (implementation.rs:567) of the lens-centre ray of every pixel of the chunk.
 * Host output arrays of chunk.width*chunk.height entries (normal: x3):
 * t (<0 = miss), world normal, entity index (scene order, -1 = miss), triangle
 * index inside the mesh (-1 for non-mesh). Any output pointer may be NULL. */
int rt_primary_hits(RtScene* scene, const RtSettings* settings,
                    const RtRenderChunk* chunk,
                    float* t_out, float* normal_out,
                    int32_t* entity_out, int32_t* tri_out);

/* create_render_chunks (viewport.rs:30-54). Writes count_x*count_y chunks in
 * the reference's order; returns the number written or a negative error if
 * `capacity` is too small or a count is 0. */
int rt_create_render_chunks(uint32_t width, uint32_t height,
                            uint32_t count_x, uint32_t count_y,
                            RtRenderChunk* out, uint32_t capacity);

/* v3_to_rgba (src/rgba.rs:6-12) on the host, for callers that kept floats. */
void rt_v3_to_rgba(const float* rgb, uint64_t n_pixels, uint8_t* rgba_out);

/* The fixed summation order of the samples of a pixel: sequential inside segments
 * of this many consecutive sample indices, then the segment sums in index order
 * (implementation.rs:641-659 sums all samples sequentially; identical for
 * samples_per_pixel <= the segment).  spp-split ranges should fall on multiples. */
uint32_t rt_sample_segment(void);

/* Introspection used by tests and the bench (no reference counterpart). */
typedef struct RtSceneInfo {
    uint64_t entity_count;
    uint64_t mesh_count;
    uint64_t triangle_count;        /* unique triangles over all meshes         */
    uint64_t instanced_triangle_count; /* sum over mesh entities                */
    uint64_t tlas_node_count;
    uint64_t blas_node_count;
    uint32_t tlas_depth;
    uint32_t blas_depth;            /* deepest BLAS                             */
    uint64_t device_bytes;          /* HBM resident bytes of the scene          */
} RtSceneInfo;
int rt_scene_info(const RtScene* scene, RtSceneInfo* out);

/* Reference-order leaf enumeration of the TLAS (object indices in the order
 * bvh.rs:210-236 would emit them if every box were hit).  Lets tests compare
 * the product's BVH build against the oracle's without sharing code.
 * mesh_index < 0: TLAS; otherwise that mesh's BLAS. Returns count written. */
int64_t rt_scene_bvh_order(const RtScene* scene, int64_t mesh_index,
                           uint32_t* out, uint64_t capacity);

/* Measurement hook (no reference counterpart; the reference only logs wall time,
 * src/job_running.rs:44): while enabled, every launch of the render kernel on
 * this scene is bracketed by HIP events on the stream it is launched on.
 * rt_scene_profile_read waits for those events, returns the summed kernel time
 * and the number of launches since the last read, and clears the list. */
int rt_scene_profile_enable(RtScene* scene, int enable);
int rt_scene_profile_read(RtScene* scene, double* kernel_ms_total, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
