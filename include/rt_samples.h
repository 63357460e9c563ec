/*
 * rt_samples.h - C ABI of librt_samples.so: host-side scene construction.
 *
 * Mirrors the parts of the reference that only CONSTRUCT a Scene (never on the
 * per-pixel path) so the same sample scenes can drive the render core:
 *   - Scene::new / add_entity / add_material / add_texture / add_light
 *     (raytracer-impl/src/implementation.rs:340-370) and the Entity builder
 *     (:232-256)                                   -> rt_builder_*
 *   - CameraConfiguration::make_camera (raytracer-samples/src/scene.rs:27-48)
 *     + Camera::new (implementation.rs:462-480)    -> rt_make_camera
 *   - load_obj_builder(..).build_mesh() (raytracer-obj/src/builder.rs:30,107;
 *     format.rs:168-313)                           -> rt_builder_load_obj
 *   - the SceneFactory list (raytracer-samples/src/lib.rs:18-37)
 *                                                  -> rt_samples_create
 * The product of every call is a plain RtSceneDesc (rt_hip.h) - exactly what
 * rt_scene_create consumes.  No GPU code lives here.
 */
#ifndef RT_SAMPLES_H
#define RT_SAMPLES_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* CameraConfiguration (raytracer-samples/src/scene.rs:12-20) */
typedef struct RtCameraConfig {
    float width;
    float height;
    float lens_radius;
    float fov;
    float angle_adjust_v;
    float angle_adjust_h;
    float focus_dist_adjust;
} RtCameraConfig;

typedef struct RtSceneBuilder RtSceneBuilder;

const char* rt_samples_last_error(void);
const char* rt_samples_last_warning(void);

/* CameraConfiguration::make_camera(look_to, default_look_from). */
int rt_make_camera(const RtCameraConfig* config, const float* look_to3,
                   const float* look_from3, RtCamera* out);
/* Camera::new(look_from, look_at, v_fov, aspect_ratio, lens_radius, focus_dist) */
int rt_camera_new(const float* look_from3, const float* look_at3, float v_fov,
                  float aspect_ratio, float lens_radius, float focus_dist, RtCamera* out);

/* ---- generic builder: Scene::new + add_* --------------------------------- */
RtSceneBuilder* rt_builder_new(const RtCamera* camera, uint32_t sky);
void            rt_builder_destroy(RtSceneBuilder* b);
/* add_material / add_texture return the MatId / TexId (>= 0) */
int rt_builder_add_material(RtSceneBuilder* b, const RtMaterial* m);
int rt_builder_add_texture(RtSceneBuilder* b, const RtTexture* t);
/* ColorMap: copies width*height texels (row 0 = top), returns TexId */
int rt_builder_add_color_map(RtSceneBuilder* b, uint32_t width, uint32_t height, const float* rgb);
/* load_color_map(path) (raytracer-obj/src/builder.rs:159-172, color_map.rs:16-33):
 * decodes a PNG or BMP file into a ColorMap texture (channel / 255.0, row 0 = top,
 * alpha dropped) and returns its TexId. */
int rt_builder_load_color_map(RtSceneBuilder* b, const char* path);
/* The decoder on its own: RGBA8, row 0 = top.  With rgba_out NULL only the size is
 * returned; otherwise capacity must be >= width*height*4. */
int rt_image_decode(const char* path, uint32_t* width, uint32_t* height,
                    uint8_t* rgba_out, uint64_t capacity);
int rt_builder_add_light(RtSceneBuilder* b, const RtLight* l);
/* Mesh{tris}: copies the triangles, returns the mesh index (shared geometry) */
int rt_builder_add_mesh(RtSceneBuilder* b, const RtMeshTri* tris, uint64_t count);
/* load_obj_builder(path).build_mesh(): parses OBJ (+ MTL when present) and adds
 * the mesh and, as one MESH_SET texture, its MeshTextureSet.  Every map_Kd file of
 * the MTL is decoded into a ColorMap (builder.rs:121-131); a missing or undecodable
 * file fails the load as in the reference, except a git-lfs pointer file, which is
 * skipped (diffuse colour only) and reported by rt_samples_last_warning().
 * Returns mesh index; *texture_set_out (nullable) receives the TexId. */
int rt_builder_load_obj(RtSceneBuilder* b, const char* path, int* texture_set_out);
/* Entity::new(hitable).translate(..)*.rotate(..)*.id(..); `hitable_kind`
 * selects which of sphere/plane/mesh_object is read. Order of translations and
 * rotations is the call order of the reference builder. Returns entity index. */
int rt_builder_add_entity(RtSceneBuilder* b, uint32_t hitable_kind,
                          const RtSphere* sphere, const RtPlane* plane,
                          const RtMeshObject* mesh_object, int32_t id,
                          const RtV3* translations, uint32_t translation_count,
                          const RtRotation* rotations, uint32_t rotation_count);
/* Plane::new(normal, mat, tex) [+ with_radius]: derives u/v bases (plane.rs:29-55) */
int rt_plane_new(const float* normal3, uint32_t mat_id, uint32_t tex_id,
                 int has_radius, float radius, RtPlane* out);
/* The description stays valid until the builder is destroyed or modified. */
const RtSceneDesc* rt_builder_desc(RtSceneBuilder* b);

/* ---- sample scene factories (raytracer-samples/src/lib.rs:18-37) --------- */
int         rt_samples_count(void);
const char* rt_samples_name(int index);
/* `name` is one of rt_samples_name(); `mesh_dir` replaces mesh_path!()'s
 * CARGO_MANIFEST_DIR/meshes (may be NULL for scenes without meshes); `params`
 * are "key=value" SceneControl overrides separated by ';' (nullable), e.g.
 * "Count=30;Spacing=30". Returns NULL on error. */
RtSceneBuilder* rt_samples_create(const char* name, const RtCameraConfig* camera,
                                  const char* mesh_dir, const char* params);

#ifdef __cplusplus
}
#endif
#endif
