// scene_build.h - host side of rt_scene_create: validate the description, build
// the acceleration structures with the reference's algorithm and flatten
// everything into the HBM layout of device_scene.h.  Pure C++ (no HIP calls).
#ifndef RT_SCENE_BUILD_H
#define RT_SCENE_BUILD_H

#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "device_scene.h"

// One TLAS node record and where its two boxes come from when the tree is refitted after a change of the entity transforms
// (rt_scene_update_transforms): count > 0: the union of the entity boxes [first, first + count) (TLAS leaf order);
// count == 0: the union of the two boxes of node record `child`.  Records are listed deepest level first.
struct RefitRec { uint32_t node, l_first, l_count, l_child, r_first, r_count, r_child, pad; };

struct FlatScene {
    std::vector<DevNode> nodes;
    std::vector<DevEntity> entities;
    std::vector<Word4> ent_box;          // 2 words per entity, TLAS leaf order (+ 2 spare words: the root box of a refit)
    std::vector<Word4> xforms;
    std::vector<DevMesh> meshes;
    std::vector<DevTriHot> tri_hot;
    std::vector<DevTriCold> tri_cold;
    std::vector<uint32_t> tri_rank;
    std::vector<DevPlane> planes;
    std::vector<DevMaterial> materials;
    std::vector<DevTexture> textures;
    std::vector<int32_t> tex_set_entries;
    std::vector<float> texels;
    std::vector<DevLight> lights;
    DevSceneView view{};          // scalars filled; pointers set after upload
    RtSceneInfo info{};
    // reference emission order (bvh.rs:210-236 with every box hit) of the TLAS
    // and of each BLAS, for rt_scene_bvh_order
    std::vector<uint32_t> tlas_order;
    std::vector<std::vector<uint32_t>> blas_order;
    // refit of the TLAS on the device
    std::vector<RefitRec> refit;
    std::vector<uint32_t> refit_level_start;   // boundaries of the levels in `refit` (deepest first)
    std::vector<uint32_t> tlas_perm;           // TLAS position -> entity index of the description
    // rebuild of the TLAS on the device (rt_scene_rebuild_tlas): the TLAS's share of the LDS node cache (its records [0, tlas_hot_records)),
    // and the deepest BLAS walk (the traversal stack bound is tlas depth + 1 + blas_walk_depth + 2)
    uint32_t tlas_hot_records = 0, blas_walk_depth = 0;
    // rebuild of one mesh's BLAS on the device (rt_scene_rebuild_blas): the depth of the TLAS walk, and per mesh its share of the LDS node
    // cache (records [blas_hot_base, + blas_hot_count)) and the depth of its walk
    uint32_t tlas_walk_depth = 0;
    std::vector<uint32_t> blas_hot_base, blas_hot_count, blas_walk_depths;
};

// Host half of rt_scene_update_transforms: the transform records of every entity (layout of FlatScene::xforms) for new
// translations / rotations, with sin / cos evaluated by the host libm like at creation; also whether any rotation axis is
// not of unit length (DevSceneView::loose_boxes).
void host_make_xforms(const RtEntity* entities, const std::vector<uint32_t>& tlas_perm, const RtV3* translations, const RtRotation* rotations,
                      std::vector<Word4>& xforms, bool& loose_boxes);
// DevSceneView::coc_rel_max from the camera and the TLAS root box
float host_coc_rel_max(const RtCamera& camera, const float root_min[3], const float root_max[3], bool any_entity);

// Returns RT_OK or a negative code with `error` filled.
int build_flat_scene(const RtSceneDesc* desc, FlatScene& out, std::string& error);

// viewport.rs:30-54
int host_create_render_chunks(uint32_t width, uint32_t height, uint32_t count_x, uint32_t count_y,
                              RtRenderChunk* out, uint32_t capacity);
// src/rgba.rs:6-12
void host_v3_to_rgba(const float* rgb, uint64_t n_pixels, uint8_t* rgba_out);

#endif
