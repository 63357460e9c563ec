// scene_build.h - host side of rt_scene_create: validate the description, build
// the acceleration structures with the reference's algorithm and flatten
// everything into the HBM layout of device_scene.h.  Pure C++ (no HIP calls).
#ifndef RT_SCENE_BUILD_H
#define RT_SCENE_BUILD_H

#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "device_scene.h"

struct FlatScene {
    std::vector<DevNode> nodes;
    std::vector<DevEntity> entities;
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
};

// Returns RT_OK or a negative code with `error` filled.
int build_flat_scene(const RtSceneDesc* desc, FlatScene& out, std::string& error);

// viewport.rs:30-54
int host_create_render_chunks(uint32_t width, uint32_t height, uint32_t count_x, uint32_t count_y,
                              RtRenderChunk* out, uint32_t capacity);
// src/rgba.rs:6-12
void host_v3_to_rgba(const float* rgb, uint64_t n_pixels, uint8_t* rgba_out);

#endif
