// device_scene.h - HBM layout of a built scene (shared by scene_build.cpp, which
// fills it on the host, and render.hip, which reads it on the GPU).
//
// Design rule: every record a lane fetches while traversing is a whole number
// of 16-byte words, so one lane's gather is one or a few dwordx4 loads out of a
// single 64-byte line.  Records are grouped in separate flat arrays by how hot
// they are: traversal data (nodes, triangle planes, entity heads) apart from
// data only touched once per ray (uv, texture keys, materials, lights).
#ifndef RT_DEVICE_SCENE_H
#define RT_DEVICE_SCENE_H

#include <stdint.h>

// ---- 16-byte word usable on host and device --------------------------------
struct alignas(16) Word4 {
    union { float f[4]; uint32_t u[4]; int32_t i[4]; };
};

// ---- BVH node: BOTH children's boxes in one 64-byte record ------------------
// The reference's tree (bvh.rs:18-48: one box per node, enum Branch/Leaf) is
// re-laid so that one visit fetches one line and tests two boxes.  Box values
// are the reference's, unchanged.
// Same-axis planes of the LEFT and RIGHT box sit in adjacent dwords (three dwordx4
// loads bring both boxes; the pairing is what a packed-math test would want, which
// measured slower than scalar on gfx950 - see slab2 in render.hip):
//   w0 = (lmin.x rmin.x lmin.y rmin.y)   w1 = (lmin.z rmin.z lmax.x rmax.x)
//   w2 = (lmax.y rmax.y lmax.z rmax.z)   w3.u = (left ref, right ref, -, -)
// child ref: bit31 = 0 -> inner node: index in the scene's one node array
//            bit31 = 1 -> leaf: bits 28..30 = count-1 (1..8), bits 0..27 = first
//                        object position (relative to the BVH's object base)
struct alignas(64) DevNode { Word4 w[4]; };

#define RT_REF_LEAF_BIT   0x80000000u
#define RT_REF_COUNT_SHIFT 28
#define RT_REF_FIRST_MASK 0x0FFFFFFFu
#define RT_LEAF_MAX       8u

// Threads of a render block, and the LDS node cache of a block: the scene's first `lds_nodes` records,
// RT_NODE_CACHE_STRIDE bytes apart (20 dwords: eight different records fall into eight different groups
// of four banks, where 64-byte spacing would put all of them into two).
#ifndef RT_BLOCK
#define RT_BLOCK 256
#endif
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4     /* resident waves per SIMD the render kernel is built for: 512 / 4 -> 128 VGPRs; 3 and 5 measured slower */
#endif
#define RT_NODE_CACHE_STRIDE 80u

// ---- Entity (implementation.rs:224-230), stored in TLAS leaf order ----------
// w0.f = (t0.x t0.y t0.z radius)  first translation inline (common case: 1)
// w1.u = (kind | n_trans<<8 | n_rot<<16, hitable index (plane / mesh), xform first, mat_id)
// w2.u = (tex_id, original entity index, Entity.id as i32, rank base)
struct alignas(16) DevEntity { Word4 w[3]; };

// ---- transforms: Word4 slots, per entity: n_trans translation slots
// (x y z -) then per rotation 2 slots: (axis.x axis.y axis.z cos(-t))
// (sin(-t) cos(+t) sin(+t) identity-flag).  sin/cos are evaluated once on the host with the
// same libm call the reference makes per ray (types.rs:200).

// ---- mesh header (one BLAS) --------------------------------------------------
// w0.u = (node base, tri base, tri count, root ref)
// w1.f = (root min.xyz, root max.x)   w2.f = (root max.y, root max.z, -, -)
struct alignas(16) DevMesh { Word4 w[3]; };

// ---- triangle, stored in BLAS leaf order -------------------------------------
// hot (48 B):  w0 = (a.xyz, n.x) w1 = (b.xyz, n.y) w2 = (c.xyz, n.z)
//              n = cross(b-a, c-a) exactly as mesh.rs:22-24 computes it per test
// cold (32 B): w0 = (a_uv, b_uv) w1.f = (c_uv.x c_uv.y) w1.i[2] = tex_key w1.i[3] = original index
struct alignas(16) DevTriHot { Word4 w[3]; };
struct alignas(16) DevTriCold { Word4 w[2]; };

// ---- plane (plane.rs:19-26) ---------------------------------------------------
// w0 = (normal.xyz, radius) w1 = (u_basis.xyz, has_radius as uint) w2 = (v_basis.xyz, -)
struct alignas(16) DevPlane { Word4 w[3]; };

// ---- material / texture / light ------------------------------------------------
// material: w0 = (kind as uint, reflectivity, fuzz, opacity) w1 = (ref_index, - - -)
struct alignas(16) DevMaterial { Word4 w[2]; };
// texture:  w0.u = (kind, child0, child1, width) w1 = (height u, texel_first u, size f, -)
//           w2.f = (color.xyz, -) w3.f = (color2.xyz, -)
struct alignas(16) DevTexture { Word4 w[4]; };
// light:    w0 = (kind u, intensity f, angle_deg f, -) w1 = (origin.xyz,-) w2 = (direction.xyz,-) w3 = (color.xyz,-)
struct alignas(16) DevLight { Word4 w[4]; };

// ---- everything a kernel needs, passed by value ---------------------------------
struct DevSceneView {
    // camera (implementation.rs:451-459)
    float cam_llc[3], cam_horizontal[3], cam_vertical[3], cam_origin[3], cam_u[3], cam_v[3];
    float cam_lens_radius;
    uint32_t sky;
    uint32_t light_count;
    uint32_t entity_count;
    // TLAS
    float tlas_min[3], tlas_max[3];
    uint32_t tlas_root_ref;
    uint32_t stack_entries;       // per-lane traversal stack depth the trees need
    uint32_t shadow_any_hit;      // no dielectric material in the scene: any occluder ends a shadow ray
    uint32_t mesh_entities;       // some entity is a mesh instance (the walk enters BLASes)
    uint32_t loose_boxes;         // some entity rotates about a non-unit axis: its box does not bound its hits, so
                                  // TLAS boxes are tested against the ray's whole interval like the reference does
    uint32_t sphere_entities, plane_entities;   // some entity is a sphere / a plane
    uint32_t general_xforms;      // some entity has other transforms than exactly one translation
    float coc_rel_max;            // upper bound of a primary hit's circle of confusion, as a fraction of the image width (inf: unbounded)
    uint32_t rich_shading;        // some light is a lamp (cone), or some texture is not a plain colour / a checker
    uint32_t lds_nodes;           // the first lds_nodes node records are kept in LDS by the render kernel
    uint32_t lds_ents;            // != 0 (then = entity_count): word 0 of every entity record (centre, radius) is kept in LDS behind the node cache - scenes
                                  // of plain translated spheres only, where word 0 is all a leaf test reads (set by rt_scene_create when the block's LDS has the room)
    const DevNode* nodes;         // the top levels of the TLAS and of every BLAS first (breadth first), then the rest of each tree (depth first)
    const DevEntity* entities;    // TLAS leaf order
    const Word4* ent_box;         // per entity (TLAS leaf order) its own world box, 2 words: (lo.xyz, -) (hi.xyz, -); Entity::calculate_origin_aabb
    const Word4* xforms;
    const DevMesh* meshes;
    const DevTriHot* tri_hot;
    const DevTriCold* tri_cold;
    const uint32_t* tri_rank;     // reference visit rank per triangle (read on exact ties only)
    const DevPlane* planes;
    const DevMaterial* materials;
    const DevTexture* textures;
    const int32_t* tex_set_entries;
    const float* texels;          // 3 floats per texel
    const DevLight* lights;
    // (behind everything the render kernel reads: a new field in the middle moves its kernel arguments, and with them its code - weekend -0.7 %)
    uint32_t uniform_cost_class;  // every material of the scene gives a hit the same cost class (classify_kernel: 0 dielectric, 1 reflective, 2 other): that class; 0xff: they differ
};

#endif
