// render.hip - the MI355X (gfx950) render core and its C ABI (include/rt_hip.h).
//
// Replaces, for one RenderChunk at a time, the pixel loop of the reference's
// worker thread (src/render.rs:148-159) and everything it calls in
// raytracer-impl (implementation.rs:498-661, bvh.rs, shapes/*, materials.rs,
// lights.rs, texture.rs).  Written for CDNA4 only:
//   * one lane = one work item (a pixel x a segment of its samples), executed by
//     a per-lane STATE MACHINE whose only heavy loop is one shared two-level BVH
//     traversal: primary, reflection/refraction and shadow rays of different
//     lanes all run through the same traversal code, so a 64-lane wave does not
//     idle while a few lanes chase secondary rays;
//   * the recursion of cast_ray_recursive (implementation.rs:559-633) is turned
//     into a weight-carrying work list (the colour is linear in the children),
//     with the reference's depth-budget split; deferred rays live in a small
//     per-lane private stack, the traversal stack lives in LDS (one dword column
//     per lane: bank = lane, conflict free);
//   * two-box 64-byte nodes, near child first, t_max shrinking; exact ties in t
//     are broken by the reference's visit rank, so the winner is the reference's;
//   * f32 arithmetic in the reference's operation order, no FMA contraction
//     (build: -ffp-contract=off), IEEE divide/sqrt; sin/cos of entity rotations
//     and the camera basis come precomputed from the host;
//   * counter-based RNG (Philox4x32-10) addressed by (seed, x, y, sample, node):
//     the image is independent of chunking, launch shape and GPU count.
// MFMA is deliberately unused: nothing here is a dense contraction.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>      // rocPRIM's device radix sort (rt_scene_rebuild_tlas)

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/rt_hip.h"
#include "device_scene.h"
#include "scene_build.h"

// Samples of one pixel are summed sequentially inside segments of this many
// consecutive sample indices, and the segment sums sequentially in index order.
// (implementation.rs:641-659 sums all samples sequentially; for spp <= the segment
// the two are the same order.)  Fixed, so results do not depend on launch shape.
// A work item is one pixel x one segment: 8 measured best on MI355X (32 / 16 / 8 / 4:
// 57.4 / 53.5 / 52.7 / 56.9 ms for the weekend 1080p 256 spp frame; smaller items
// shorten the tail of a launch, larger ones amortise the per-item set-up).
#define RT_SEGMENT_SHIFT 3u
#define RT_SAMPLE_SEGMENT (1u << RT_SEGMENT_SHIFT)
#define RT_PENDING_MAX 8
#ifndef RT_QUEUES
#define RT_QUEUES 16u            /* block counters per launch, one cache line each */
#endif
#define RT_MAX_REFLECTIONS 31u
#ifndef RT_LENS_CELLS
#define RT_LENS_CELLS 64         /* lens strata of the product's sampler: 64 (8 x 8 cells over complete groups of 64 samples, then 4 x 4 over groups of 16;
                                    measured: mesh_demo 4K x 1024 spp +7.6 %) or 16 (round 2's sampler: 4 x 4 only; the oracle's stratified mode follows 64) */
#endif
#ifndef RT_CANCEL_POLL_BLOCKS
#define RT_CANCEL_POLL_BLOCKS 64u   /* the first wave of every 64th workgroup reads the cancel word at its block claims (power of two) */
#endif
#ifdef RT_PHASE_TIME            /* (make time: a flavour of the diagnostics build) */
#define RT_PHASE_STATS
#endif



namespace {

// ===========================================================================
// device math: f32, reference operation order (types.rs:11-207)
// ===========================================================================
struct F3 { float x, y, z; };

__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float f) { return f3(a.x * f, a.y * f, a.z * f); }
__device__ __forceinline__ F3 operator/(F3 a, float f) { return f3(a.x / f, a.y / f, a.z / f); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
    return f3((a.y * b.z - a.z * b.y), -(a.x * b.z - a.z * b.x), (a.x * b.y - a.y * b.x));
}
__device__ __forceinline__ float length(F3 a) { return sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }
__device__ __forceinline__ F3 unit(F3 a) { float len = length(a); return len == 0.0f ? a : a / len; }
__device__ __forceinline__ bool is_zero(F3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
// types.rs:200 with cos/sin already evaluated
__device__ __forceinline__ F3 rotate_cs(F3 p, F3 axis, float c, float s) {
    return (p * c) + (cross(axis, p) * s) + (axis * dot(axis, p) * (1.0f - c));
}
// A rotation whose host-evaluated cos is exactly 1 and sin exactly 0 maps p to
// (p*1) + (cross*0) + (axis*dot*0) = p + (+-0) + (+-0), which is p itself, bit for
// bit, whenever every component of p is finite, non-zero and far from overflow (a zero
// component would take the sign of the zero terms; an infinite one would turn them into
// NaN).  Only then is the arithmetic skipped; otherwise the full formula runs.
__device__ __forceinline__ bool plain(float x) { const float a = fabsf(x); return a > 0.0f && a < 1.0e30f; }
__device__ __forceinline__ F3 rotate_cs_id(F3 p, F3 axis, float c, float s, bool identity) {
    if (identity && plain(p.x) && plain(p.y) && plain(p.z)) return p;
    return rotate_cs(p, axis, c, s);
}
__device__ __forceinline__ F3 ld3(const float* p) { return f3(p[0], p[1], p[2]); }
__device__ __forceinline__ F3 w3(const Word4& w) { return f3(w.f[0], w.f[1], w.f[2]); }

__device__ __forceinline__ Word4 ldw(const Word4* p) {
    // one 16-byte load
    const float4 v = *reinterpret_cast<const float4*>(p);
    Word4 w; w.f[0] = v.x; w.f[1] = v.y; w.f[2] = v.z; w.f[3] = v.w; return w;
}

#define RT_PI 3.14159265358979323846f
#define RT_BIAS 0.004f                       /* implementation.rs:498 */
#define RT_F32_MAX 3.40282347e+38f
#define RT_NONE 0xFFFFFFFFu
// Boxes are culled against best_t = (t of the best hit so far) x RT_CULL_SLACK, not against that t itself.
// The reference tests every box against the ray's whole interval and takes the minimum afterwards
// (bvh.rs:210-236), and its boxes do not bound the COMPUTED t of their hits to the last bit: an axis-aligned
// triangle lies in a face of its own leaf box and its plane-equation t (plane.rs:12-13) comes out 1-3 ulp
// before the slab entry of that face (implementation.rs:102-115); entity boxes add the rounding of the
// rotated ray (a few hundred ulp for unit-length secondary rays far from the origin).  Culling at exactly
// the best t then drops candidates the reference finds - on the 900-instance scene 450 of 4.4 M sample rays,
// nearly all exact ties between coincident faces that the visit rank must decide (tools/mismatch_diag.py).
// 2^-10 = 8000 ulp of margin costs 0.8 % more node steps there, 0.2 % elsewhere.  Winners are still chosen
// by the exact (t, rank) order, so a candidate visited needlessly never changes the result; the bound never
// exceeds the ray's t_max, so no box is entered that the reference's own test would reject.
#define RT_CULL_SLACK 1.0009765625f
#ifndef RT_OWN_BOX_PRETEST
#define RT_OWN_BOX_PRETEST 1
#endif
// Sticky device-side error word (host-mapped, RtScene::dev_error): a capacity that the host sized from the trees turned
// out too small.  Never expected; reported as RT_E_DEPTH instead of a silently wrong image.
#define RT_DEVERR_STACK   1u      /* traversal stack full: a deferred subtree would have been dropped */
#define RT_DEVERR_PENDING 2u      /* deferred-ray list full: a refraction ray would have been dropped */
#define RT_DEVERR_POISONED 4u     /* set by the HOST: a call failed after it had begun to rewrite the scene's device data (rt_scene_rebuild_tlas) */

// ===========================================================================
// counter-based RNG: Philox4x32-10 (Random123), counter = (x | y<<16, sample,
// node, purpose | block<<8), key = seed.  f32 = (u >> 8) * 2^-24 like rand 0.9.
// ===========================================================================
enum { PURPOSE_LENS = 0, PURPOSE_PATH = 1 };

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: integer
        // multiplies are quarter rate on the VALU, and a sample spends 20 instead of 40 of them per draw
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * (unsigned long long)c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * (unsigned long long)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u32_to_f32(uint32_t v) { return (float)(v >> 8) * (1.0f / 16777216.0f); }

// the two draws every consumer on this path needs (lens jitter: implementation.rs:649-650;
// random_normal_reflection_angle: :17-18)
__device__ __forceinline__ void rng_pair(uint64_t seed, uint32_t pxy, uint32_t sample, uint32_t node, uint32_t purpose, float& r0, float& r1) {
    uint32_t o[4];
    philox4x32_10(pxy, sample, node, purpose, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    r0 = u32_to_f32(o[0]); r1 = u32_to_f32(o[1]);
}

// ===========================================================================
// traversal
// ===========================================================================
struct Hit {
    float t;
    uint32_t ent;      // entity position (TLAS leaf order), RT_NONE = miss
    uint32_t tri;      // triangle position (global tri arrays), RT_NONE for sphere/plane
    uint32_t erank;    // reference visit rank of the entity (tie breaking)
};

// implementation.rs:92-118.  The reference walks the axes in turn:
//     inv_d = 1/dir; t0 = (min-o)*inv_d; t1 = (max-o)*inv_d; if inv_d < 0 swap;
//     t_min = t0 > t_min ? t0 : t_min;  t_max = t1 < t_max ? t1 : t_max;  if t_max < t_min -> miss
// Same values here, arranged for the VALU:
//   * the swap is applied to the box planes before the arithmetic (the products are
//     the same two numbers either way);
//   * `x > t ? x : t` with t never NaN is exactly fmaxf(x, t) (a NaN x is ignored by
//     both; +-0 order is irrelevant to every later comparison), and max / min are
//     associative, so the three axes fold into v_max3 / v_min3;
//   * the per-axis early exit equals one test at the end (t_min only grows, t_max only
//     shrinks, NaN never enters either).
// `entry` = the final t_min (used to order the two children).
__device__ __forceinline__ bool slab(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                     F3 o, F3 inv, float t_min, float t_max, float& entry) {
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const float ax = ((nx ? hix : lox) - o.x) * inv.x, bx = ((nx ? lox : hix) - o.x) * inv.x;
    const float ay = ((ny ? hiy : loy) - o.y) * inv.y, by = ((ny ? loy : hiy) - o.y) * inv.y;
    const float az = ((nz ? hiz : loz) - o.z) * inv.z, bz = ((nz ? loz : hiz) - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fmaxf(ax, ay), az), t_min);
    const float tf = fminf(fminf(fminf(bx, by), bz), t_max);
    entry = tn;
    return !(tf < tn);
}

// The two boxes of one node record (layout: device_scene.h).  Packed f32 math
// (v_pk_add_f32 / v_pk_mul_f32 on the left/right pair) was measured 1-2 % SLOWER than
// two plain tests, and compiler SLP packing of the F3 arithmetic 9 % slower, on
// gfx950: the build uses -fno-slp-vectorize and this stays scalar.
__device__ __forceinline__ void slab2(const Word4& a, const Word4& b, const Word4& c, F3 o, F3 inv, float t_min, float t_max,
                                      bool& hl, bool& hr, float& el, float& er) {
    hl = slab(a.f[0], a.f[2], b.f[0], b.f[2], c.f[0], c.f[2], o, inv, t_min, t_max, el);
    hr = slab(a.f[1], a.f[3], b.f[1], b.f[3], c.f[1], c.f[3], o, inv, t_min, t_max, er);
}

// Record `index` of an array of `stride`-byte records through a 32-bit byte offset from the (wave-uniform) base: the load then takes
// its address as SGPR base + VGPR offset, where base[index] costs a 64-bit multiply-add per lane (quarter rate).  rt_scene_create
// refuses scenes whose arrays exceed 4 GiB.  (Dreadnaught +1 %, mesh_demo +1.3 %, 900 instances +1.6 % for the node records alone.)
template <class T>
__device__ __forceinline__ const Word4* rec32(const T* base, uint32_t index, uint32_t stride) {
    return reinterpret_cast<const Word4*>(reinterpret_cast<const char*>(base) + index * stride);
}

// reference visit rank of the entity at TLAS position `ei`: decides exact ties in t between two entities (implementation.rs:415)
__device__ __forceinline__ uint32_t entity_rank(const DevSceneView& S, uint32_t ei) { return ldw(rec32(S.entities, ei, (uint32_t)sizeof(DevEntity)) + 2).u[3]; }

// Entity::hit's world -> entity ray transform, implementation.rs:288-294
// XFORMS = false: every entity of the scene has exactly one translation and no rotation
// (DevSceneView::general_xforms == 0), so only the inline translation is ever needed.
template <bool XFORMS = true>
__device__ __forceinline__ void ray_to_entity(const DevSceneView& S, const Word4& e0, const Word4& e1, F3& o, F3& d) {
    const uint32_t packed = e1.u[0];
    const uint32_t n_trans = (packed >> 8) & 0xff, n_rot = (packed >> 16) & 0xff;
    if (!XFORMS || (n_trans == 1 && n_rot == 0)) { o = o - w3(e0); return; }     // translation kept inline
    const Word4* x = S.xforms + e1.u[2];
    for (uint32_t k = 0; k < n_trans; k++) o = o - w3(ldw(x + k));
    x += n_trans;
    for (uint32_t k = 0; k < n_rot; k++) {
        const Word4 a = ldw(x + 2 * k), b = ldw(x + 2 * k + 1);
        const F3 axis = w3(a);
        o = rotate_cs_id(o, axis, a.f[3], b.f[0], b.u[3] != 0u);      // cos(-theta), sin(-theta)
        d = rotate_cs_id(d, axis, a.f[3], b.f[0], b.u[3] != 0u);
    }
}

// The shared closest-hit search: Scene::hit_closest (implementation.rs:372-375)
// = EntityBvhRoot::try_hit (:411-416) over Entity::hit (:285-310) over
// Sphere::hit / Plane::hit / MeshBvhRoot::try_hit (sphere.rs:36-53, plane.rs:59-90,
// mesh.rs:92-99).  `stack` is this lane's LDS column (stride RT_BLOCK dwords).
// (WIDE / `wide`: the 4-wide walk of the diagnostics build -DRT_BVH4_EXPERIMENT, rt_debug_bvh4_primary below; never in the product)
template <bool WIDE = false>
__device__ __forceinline__ void trace(const DevSceneView& S, const F3 wo, const F3 wd, const float t_min, const float t_max,
                                      uint32_t* stack, const uint32_t stack_cap, uint32_t* error_flag, Hit& best, const Word4* wide = nullptr) {
    best.t = t_max; best.ent = RT_NONE; best.tri = RT_NONE; best.erank = 0;
    float best_t = t_max;                 // culling bound; primitives test against [t_min, t_max]

    F3 o = wo, d = wd, inv = f3(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);
    uint32_t sp = 0, inst_sp = RT_NONE;    // inst_sp != NONE: inside a BLAS entered at that stack height
    const DevNode* const nodes = S.nodes;  // TLAS and every BLAS: child refs index the one array
    uint32_t tri_base = 0;                 // current BLAS
    uint32_t cur_ent = 0, cur_erank = 0;

    // next deferred subtree; leaving a BLAS restores the world-space ray
    auto pop = [&]() -> uint32_t {
        if (sp == inst_sp) { inst_sp = RT_NONE; o = wo; d = wd; inv = f3(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z); }
        if (sp == 0) return RT_NONE;
        return stack[(--sp) * RT_BLOCK];
    };

    uint32_t cur;
    {
        float e;
        cur = slab(S.tlas_min[0], S.tlas_min[1], S.tlas_min[2], S.tlas_max[0], S.tlas_max[1], S.tlas_max[2], o, inv, t_min, best_t, e)
                  ? S.tlas_root_ref : RT_NONE;
    }
    for (;;) {
        // ---- phase 1: walk inner nodes until this lane holds a leaf (RT_NONE has the
        // leaf bit set and ends the walk too).  One 64-byte record = two boxes. ------
        while (cur < RT_REF_LEAF_BIT) {
#ifdef RT_BVH4_EXPERIMENT
            if (WIDE) {
                // one 128-byte record = up to four child boxes (lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] refs[4]); the children that are hit
                // are visited nearest first: a 5-comparator network on (entry distance, ref), the rest pushed farthest first
                const Word4* n = wide + (size_t)cur * 8;
                const Word4 lx = ldw(n), ly = ldw(n + 1), lz = ldw(n + 2), hx = ldw(n + 3), hy = ldw(n + 4), hz = ldw(n + 5), rf = ldw(n + 6);
                const float bound = (S.loose_boxes && inst_sp == RT_NONE) ? t_max : best_t;
                float k[4]; uint32_t r4[4]; uint32_t hits = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float e;
                    const bool h = slab(lx.f[c], ly.f[c], lz.f[c], hx.f[c], hy.f[c], hz.f[c], o, inv, t_min, bound, e);
                    k[c] = h ? e : __builtin_inff(); r4[c] = rf.u[c]; hits += h ? 1u : 0u;
                }
                auto cswap = [&](int a, int b) {
                    const bool sw = k[b] < k[a];
                    const float ka = sw ? k[b] : k[a], kb = sw ? k[a] : k[b];
                    const uint32_t ra = sw ? r4[b] : r4[a], rb = sw ? r4[a] : r4[b];
                    k[a] = ka; k[b] = kb; r4[a] = ra; r4[b] = rb;
                };
                cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
                if (hits == 0u) { cur = pop(); continue; }
                for (int c = 3; c >= 1; c--)
                    if (hits > (uint32_t)c) {
                        if (sp < stack_cap) stack[(sp++) * RT_BLOCK] = r4[c];
                        else atomicOr(error_flag, RT_DEVERR_STACK);
                    }
                cur = r4[0];
                continue;
            }
#endif
            const Word4* n = nodes[cur].w;
            const Word4 a = ldw(n), b = ldw(n + 1), c = ldw(n + 2), r = ldw(n + 3);
            float el, er; bool hl, hr;
            slab2(a, b, c, o, inv, t_min, (S.loose_boxes && inst_sp == RT_NONE) ? t_max : best_t, hl, hr, el, er);
            if (hl && hr) {
                const bool left_first = el <= er;
                const uint32_t far_ref = left_first ? r.u[1] : r.u[0];
                cur = left_first ? r.u[0] : r.u[1];
                if (sp < stack_cap) stack[(sp++) * RT_BLOCK] = far_ref;
                else atomicOr(error_flag, RT_DEVERR_STACK);
            } else if (hl || hr) cur = hl ? r.u[0] : r.u[1];
            else cur = pop();
        }
        if (cur == RT_NONE) break;
        // ---- phase 2: the leaf ---------------------------------------------------------
        const uint32_t first = cur & RT_REF_FIRST_MASK;
        const uint32_t count = ((cur >> RT_REF_COUNT_SHIFT) & 7u) + 1u;
        if (inst_sp != RT_NONE) {
            // try_hit_tri, mesh.rs:18-76 (normal precomputed with the same ops)
            for (uint32_t k = 0; k < count; k++) {
                const uint32_t ti = tri_base + first + k;
                const Word4* tw = S.tri_hot[ti].w;
                const Word4 wa = ldw(tw), wb = ldw(tw + 1), wc = ldw(tw + 2);
                const F3 A = w3(wa), B = w3(wb), C = w3(wc), normal = f3(wa.f[3], wb.f[3], wc.f[3]);
                const float denominator = dot(d, normal);                       // plane.rs:7-10
                if (fabsf(denominator) < 1.0e-6f) continue;
                const float t = dot(A - o, normal) / denominator;               // plane.rs:12-13
                if (t < t_min || t > t_max) continue;                           // mesh.rs:28
                if (t > best.t) continue;                                       // cannot win
                const F3 p = o + (d * t);
                const F3 n_abp = cross(B - A, p - A);
                if (dot(normal, n_abp) < 0.0f) continue;
                const F3 n_bcp = cross(C - B, p - B);
                if (dot(normal, n_bcp) < 0.0f) continue;
                const F3 n_cap = cross(A - C, p - C);
                if (dot(normal, n_cap) < 0.0f) continue;
                bool better = best.ent == RT_NONE || t < best.t;
                if (!better && t == best.t) {
                    // exact tie: the reference keeps the candidate it visits first
                    // (implementation.rs:415, mesh.rs:96-98); ranks are only fetched here
                    better = cur_erank < best.erank ||
                             (cur_erank == best.erank && best.tri != RT_NONE && S.tri_rank[ti] < S.tri_rank[best.tri]);
                }
                if (better) { best.t = t; best.ent = cur_ent; best.tri = ti; best.erank = cur_erank; best_t = fminf(t * RT_CULL_SLACK, t_max); }
            }
            cur = pop();
            continue;
        }
        // TLAS leaf: its entities in order; a mesh entity defers the rest of the leaf
        // and turns the walk into its BLAS.
        bool entered = false;
        for (uint32_t k = 0; k < count && !entered; k++) {
            const uint32_t ei = first + k;
            const Word4* ew = S.entities[ei].w;
            const Word4 e0 = ldw(ew), e1 = ldw(ew + 1);
            const uint32_t kind = e1.u[0] & 0xff;
            if (!S.loose_boxes && RT_OWN_BOX_PRETEST && kind == RT_HITABLE_MESH) {      // the render kernel's pre-test of the instance's own world box (see there)
                const Word4 bl = ldw(S.ent_box + 2 * ei), bh = ldw(S.ent_box + 2 * ei + 1);
                const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
                const float ax = ((nx ? bh.f[0] : bl.f[0]) - wo.x) * inv.x, bx = ((nx ? bl.f[0] : bh.f[0]) - wo.x) * inv.x;
                const float ay = ((ny ? bh.f[1] : bl.f[1]) - wo.y) * inv.y, by = ((ny ? bl.f[1] : bh.f[1]) - wo.y) * inv.y;
                const float az = ((nz ? bh.f[2] : bl.f[2]) - wo.z) * inv.z, bz = ((nz ? bl.f[2] : bh.f[2]) - wo.z) * inv.z;
                const float tn = fmaxf(fmaxf(fmaxf(ax, ay), az), t_min);
                const float tf = fminf(fminf(fminf(bx, by), bz), best_t);
                if (tf * RT_CULL_SLACK < tn) continue;
            }
            F3 eo = wo, ed = wd;
            ray_to_entity(S, e0, e1, eo, ed);
            float t = 0.0f; bool hit = false;
            if (kind == RT_HITABLE_SPHERE) {
                // intersect_sphere + Sphere::hit, sphere.rs:4-17, 36-39
                const float radius = e0.f[3];
                const float qa = dot(ed, ed);
                const float qb = dot(eo, ed);
                const float qc = dot(eo, eo) - radius * radius;
                const float discriminant = qb * qb - qa * qc;
                if (discriminant > 0.0f) {
                    const float sq = sqrtf(discriminant);
                    const float t0 = (-qb - sq) / qa, t1 = (-qb + sq) / qa;
                    const bool in0 = t_min < t0 && t0 < t_max, in1 = t_min < t1 && t1 < t_max;
                    hit = in0 || in1;
                    t = (in0 && in1) ? fminf(t0, t1) : (in0 ? t0 : t1);
                }
            } else if (kind == RT_HITABLE_PLANE) {
                // intersect_plane + Plane::hit, plane.rs:4-17, 59-70
                const Word4* pw = S.planes[e1.u[1]].w;
                const Word4 p0 = ldw(pw), p1 = ldw(pw + 1);
                const F3 pn = w3(p0);
                const float denominator = dot(ed, pn);
                if (!(fabsf(denominator) < 1.0e-6f)) {
                    t = dot(f3(0.0f, 0.0f, 0.0f) - eo, pn) / denominator;
                    hit = !(t < t_min || t > t_max);
                    if (hit && p1.u[3]) { const F3 p = eo + (ed * t); hit = !(length(p) > p0.f[3]); }
                }
            } else {
                // MeshObject::hit -> BLAS (mesh.rs:161-176): enter the instance
                const Word4* mw = S.meshes[e1.u[1]].w;
                const Word4 m0 = ldw(mw), m1 = ldw(mw + 1), m2 = ldw(mw + 2);
                const F3 einv = f3(1.0f / ed.x, 1.0f / ed.y, 1.0f / ed.z);
                float e;
                if (slab(m1.f[0], m1.f[1], m1.f[2], m1.f[3], m2.f[0], m2.f[1], eo, einv, t_min, best_t, e)) {      // entity space: exact box
                    if (k + 1 < count) {
                        if (sp < stack_cap) stack[(sp++) * RT_BLOCK] = RT_REF_LEAF_BIT | ((count - k - 2) << RT_REF_COUNT_SHIFT) | (ei + 1);
                        else atomicOr(error_flag, RT_DEVERR_STACK);
                    }
                    inst_sp = sp;
                    o = eo; d = ed; inv = einv;
                    tri_base = m0.u[1];
                    cur_ent = ei; cur_erank = ldw(ew + 2).u[3];
                    cur = m0.u[3];
                    entered = true;
                }
            }
            if (hit) {
                const uint32_t erank = ldw(ew + 2).u[3];
                if (best.ent == RT_NONE || t < best.t || (t == best.t && erank < best.erank)) {
                    best.t = t; best.ent = ei; best.tri = RT_NONE; best.erank = erank; best_t = fminf(t * RT_CULL_SLACK, t_max);
                }
            }
        }
        if (!entered) cur = pop();
    }
}

// ===========================================================================
// hit record (implementation.rs:170-179), rebuilt once for the winning candidate
// ===========================================================================
struct HitRec { F3 p, normal; float u, v; uint32_t mat_id, tex_id; int32_t tex_key; int32_t entity_index, tri_index; };

// MESH / SPHERES / PLANES: the kinds of hitable the scene has
template <bool XFORMS = true, bool MESH = true, bool SPHERES = true, bool PLANES = true>
__device__ __forceinline__ void resolve_hit(const DevSceneView& S, F3 o, F3 d, const Hit& h, HitRec& r) {
    const Word4* ew = S.entities[h.ent].w;
    const Word4 e0 = ldw(ew), e1 = ldw(ew + 1), e2 = ldw(ew + 2);
    const uint32_t kind = e1.u[0] & 0xff;
    ray_to_entity<XFORMS>(S, e0, e1, o, d);
    r.mat_id = e1.u[3]; r.tex_id = e2.u[0]; r.entity_index = (int32_t)e2.u[1]; r.tex_key = -1; r.tri_index = -1;
    const float t = h.t;
    F3 p = o + (d * t), normal;
    r.u = 0.0f; r.v = 0.0f;
    if (SPHERES && ((!MESH && !PLANES) || kind == RT_HITABLE_SPHERE)) {
        normal = unit(p / e0.f[3]);                                              // sphere.rs:40-41
    } else if (PLANES && (!MESH || kind == RT_HITABLE_PLANE)) {
        const Word4* pw = S.planes[e1.u[1]].w;
        const Word4 p0 = ldw(pw), p1 = ldw(pw + 1), p2 = ldw(pw + 2);
        const F3 pn = w3(p0);
        normal = dot(d, pn) > 0.0f ? -pn : pn;                                   // plane.rs:73
        r.u = dot(w3(p1), p); r.v = dot(w3(p2), p);                              // plane.rs:75-79
    } else if (MESH) {
        const Word4* tw = S.tri_hot[h.tri].w;
        const Word4 wa = ldw(tw), wb = ldw(tw + 1), wc = ldw(tw + 2);
        const Word4* cw = S.tri_cold[h.tri].w;
        const Word4 c0 = ldw(cw), c1 = ldw(cw + 1);
        const F3 A = w3(wa), B = w3(wb), C = w3(wc), n = f3(wa.f[3], wb.f[3], wc.f[3]);
        const F3 n_abp = cross(B - A, p - A);
        const F3 n_cap = cross(A - C, p - C);
        const float area2 = length(n);                                           // mesh.rs:68-72
        const float bu = length(n_cap) / area2;
        const float bv = length(n_abp) / area2;
        const float bw = 1.0f - bu - bv;
        r.u = (c0.f[0] * bw + c0.f[2] * bu) + c1.f[0] * bv;
        r.v = (c0.f[1] * bw + c0.f[3] * bu) + c1.f[1] * bv;
        normal = unit(n);
        r.tex_key = c1.i[2]; r.tri_index = c1.i[3];
    }
    // implementation.rs:300-306: rotations reversed (+theta), then translations reversed
    const uint32_t n_trans = (e1.u[0] >> 8) & 0xff, n_rot = (e1.u[0] >> 16) & 0xff;
    if (!XFORMS || (n_trans == 1 && n_rot == 0)) p = p + w3(e0);
    else {
        const Word4* x = S.xforms + e1.u[2];
        for (uint32_t k = n_rot; k-- > 0;) {
            const Word4 a = ldw(x + n_trans + 2 * k), b = ldw(x + n_trans + 2 * k + 1);
            p = rotate_cs_id(p, w3(a), b.f[1], b.f[2], b.u[3] != 0u);
            normal = rotate_cs_id(normal, w3(a), b.f[1], b.f[2], b.u[3] != 0u);
        }
        for (uint32_t k = n_trans; k-- > 0;) p = p + w3(ldw(x + k));
    }
    r.p = p; r.normal = normal;
}

// ===========================================================================
// shading pieces
// ===========================================================================
__device__ __forceinline__ F3 color_sky(const DevSceneView& S, F3 dir) {                // implementation.rs:502-519
    if (S.sky == RT_SKY_BLACK) return f3(0.0f, 0.0f, 0.0f);
    const F3 unit_direction = unit(dir);
    const float t = 0.5f * (unit_direction.y + 1.0f);
    return f3(1.0f, 1.0f, 1.0f) * (1.0f - t) + (f3(0.5f, 0.7f, 1.0f) * t);
}

__device__ __forceinline__ uint32_t map_to_rank(float p, uint32_t length) {             // texture.rs:87-91
    p = fmodf(p, 1.0f);
    p = p < 0.0f ? 1.0f + p : p;
    const float v = p * (float)length;
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967040.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

// Texture::value, texture.rs:11-134 (chains MeshTextureSet -> MeshTexture -> ColorMap
// and Checker -> leaf unrolled instead of virtual calls)
// RICH = false: the scene's textures are plain colours and checkers of them only (DevSceneView::rich_shading == 0)
template <bool RICH = true>
__device__ __forceinline__ F3 texture_value(const DevSceneView& S, uint32_t tex_id, const HitRec& h) {
    const F3 not_found = f3(1.0f, 0.41f, 0.70f);                                         // texture.rs:124
    const Word4* tw = S.textures[tex_id].w;
    Word4 t0 = ldw(tw);
    uint32_t kind = t0.u[0];
    if (RICH && kind == RT_TEX_MESH_SET) {
        if (h.tex_key < 0 || h.tex_key >= t0.i[2]) return not_found;
        tw = S.textures[S.tex_set_entries[t0.i[1] + h.tex_key]].w;
        t0 = ldw(tw); kind = t0.u[0];
    }
    bool mesh = false; F3 ambient = f3(0, 0, 0), diffuse = f3(0, 0, 0);
    if (RICH && kind == RT_TEX_MESH) {
        diffuse = w3(ldw(tw + 2)); ambient = w3(ldw(tw + 3));
        if (t0.i[1] < 0) return ambient + diffuse;                                       // texture.rs:111-114
        mesh = true;
        tw = S.textures[t0.i[1]].w; t0 = ldw(tw); kind = t0.u[0];
    }
    if (kind == RT_TEX_CHECKER) {                                                        // texture.rs:33-44
        const float size = ldw(tw + 1).f[2];
        const float sines = sinf(size * h.u) * sinf(size * h.v);
        tw = S.textures[sines < 0.0f ? t0.i[1] : t0.i[2]].w; t0 = ldw(tw); kind = t0.u[0];
    }
    F3 v;
    if (kind == RT_TEX_COLOR) v = w3(ldw(tw + 2));
    else if (RICH && kind == RT_TEX_UV_TEST) v = f3(h.u, h.v, 1.0f - h.u - h.v);
    else if (RICH && kind == RT_TEX_XYZ_TEST) {
        const float mx = ldw(tw + 1).f[2];
        v = f3(0.5f + (h.p.x / mx / 2.0f), 0.5f + (h.p.y / mx / 2.0f), 0.5f + (h.p.z / mx / 2.0f));
    } else if (RICH && kind == RT_TEX_COLOR_MAP) {                                               // texture.rs:93-97
        const Word4 t1 = ldw(tw + 1);
        const uint32_t width = t0.u[3], height = t1.u[0];
        const uint64_t x = map_to_rank(h.u, width), y = map_to_rank(h.v, height);
        const uint64_t offset = ((uint64_t)height - y) * width + x;      // y <= height: no underflow
        if (offset < (uint64_t)width * height) { const float* px = S.texels + ((uint64_t)t1.u[1] + offset) * 3; v = f3(px[0], px[1], px[2]); }
        else v = not_found;
    } else v = not_found;
    return mesh ? ambient + (v * diffuse) : v;
}

struct LightRec { float t; F3 direction, color; float intensity; };

// LightSource::get_direction_and_intensity, lights.rs:35-49, 87-109, 138-148
// RICH = false: no lamp (cone) light in the scene
template <bool RICH = true>
__device__ __forceinline__ bool light_record(const DevSceneView& S, uint32_t k, F3 p, LightRec& out) {
    const Word4* lw = S.lights[k].w;
    const Word4 l0 = ldw(lw), l1 = ldw(lw + 1), l2 = ldw(lw + 2), l3 = ldw(lw + 3);
    out.color = w3(l3);
    if (l0.u[0] == RT_LIGHT_DIRECTIONAL) { out.t = RT_F32_MAX; out.direction = w3(l2); out.intensity = l0.f[1]; return true; }
    const F3 direction_to_p = p - w3(l1);
    if (RICH && l0.u[0] == RT_LIGHT_LAMP) {
        const F3 lamp = w3(l2);
        const float theta = acosf(dot(direction_to_p, lamp) / (length(direction_to_p) * length(lamp)));   // types.rs:71-75
        const float theta_deg = theta / RT_PI * 180.0f;
        if (theta_deg > l0.f[2]) return false;
    }
    const float t = length(direction_to_p);
    out.t = t; out.intensity = l0.f[1] / (4.0f * RT_PI * t); out.direction = unit(direction_to_p);
    return true;
}

__device__ __forceinline__ F3 reflect(F3 incident_direction, F3 surface_normal) {        // materials.rs:84-87
    const F3 dir = unit(incident_direction);
    return dir - (surface_normal * dot(dir, surface_normal) * 2.0f);
}
__device__ __forceinline__ F3 refract(F3 v, F3 n, float ni_over_nt) {                    // materials.rs:150-159
    const F3 uv = unit(v);
    const float dt = dot(uv, n);
    const float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (discriminant <= 0.0f) return f3(0.0f, 0.0f, 0.0f);
    return (uv - (n * dt)) * ni_over_nt - (n * sqrtf(discriminant));
}
__device__ __forceinline__ float schlick_reflect_prob(float cosine, float ref_idx) {     // materials.rs:161-165
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powf(1.0f - cosine, 5.0f);
}
// implementation.rs:16-36
__device__ __forceinline__ F3 random_normal_reflection_angle(F3 normal, float r0, float r1) {
    const float theta1 = r0 * (RT_PI / 2.0f);
    const float theta2 = r1 * (2.0f * RT_PI);
    const F3 px = cross(normal, f3(1.0f, 0.0f, 0.0f));
    const F3 perp = !is_zero(px) ? px : cross(normal, f3(0.0f, 1.0f, 0.0f));
    const F3 a = rotate_cs(normal, perp, cosf(theta1), sinf(theta1));
    return rotate_cs(a, normal, cosf(theta2), sinf(theta2));
}

// Dielectric kr / refraction intensity only (what a shadow ray needs of
// MatDielectric::scatter, materials.rs:168-190)
__device__ __forceinline__ float dielectric_kr(F3 dir, F3 normal, float ref_index, float& ni_over_nt, F3& outward_normal) {
    const float dn = dot(dir, normal);
    float cosine;
    if (dn > 0.0f) { outward_normal = -normal; ni_over_nt = ref_index; cosine = ref_index * dn / length(dir); }
    else { outward_normal = normal; ni_over_nt = 1.0f / ref_index; cosine = -dn / length(dir); }
    return schlick_reflect_prob(cosine, ref_index);
}

// ===========================================================================
// the render kernel
// ===========================================================================
struct Pending { F3 o, d, w; uint32_t limit, node; };
#define RT_SINGLE_ROUNDS_BELOW 6u /* blocks per resident wave below which a launch hands out single rounds (Dreadnaught: 10 per wave -1.8 %, 5 per wave +6.5 %, 2.5 per wave +27 %) */
#define RT_BIN_NOHIT 0x200u       /* tile_bin flag beside the class and 0x100 (rounds by lens cell): none of the tile's camera rays can meet the scene, but the sky's colour depends on the ray */
#define RT_BIN_EMPTY 4u           /* cost class (classify_kernel) of a tile none of whose camera rays can meet the scene: sorted last, never handed out */

struct RenderParams {
    uint32_t width, height, max_reflections, spp;   // RenderSettings
    uint32_t left, top, cw, ch;                     // RenderChunk
    uint64_t seed;
    uint32_t sample_first, sample_count;            // absolute sample index range to render
    uint32_t unit_len;                              // samples per work item: RT_SAMPLE_SEGMENT (coarse) or 1 (fine)
    uint32_t unit_first, unit_count;                // units (absolute index = sample / unit_len) covered by this launch
    uint32_t round_units_shift;                     // log2 of the units whose samples share a round (divides unit_count)
    float lens_coc_px;                              // classify_kernel: circle of confusion (pixels) from which a tile's rounds go by lens cell
    uint32_t lens_rounds;                           // tiles flagged by classify_kernel may run rounds of one sample index (8 units x 8 pixels)
    uint32_t lens_stratify;                         // 1: stratified lens samples (the product's sampler); 0: two independent draws (implementation.rs:648-653)
    uint32_t kind_order;                            // 1: the rays of a burst are of ONE kind - shadow rays light by light, then path rays; lanes of a later kind wait
    uint32_t* error_flag;                           // sticky device-side error word (RT_DEVERR_* bits), read back by the host entry points
    const uint32_t* cancel_word;                    // host-mapped: non-zero = rt_scene_cancel; the waves stop claiming blocks (src/render.rs:131,149: the cancellation token)
    uint32_t* progress_word;                        // host-mapped: about how many blocks of the call have been handed out so far (rt_scene_progress)
    uint32_t progress_base;                         // blocks of the call's earlier launches
    uint32_t tiles_x, pix_slots;                    // 8x8 tiling of the chunk; pix_slots = tiles*64
    float* partial;                                 // [unit_count][tile][64][3] sums of the clamped samples of each unit (8x8 tiles of the chunk)
    unsigned long long* counters;                   // RtRayCounters on the device (nullable)
    uint32_t* block_counter;                        // RT_QUEUES counters, 64 B apart (zeroed per launch): next unclaimed block of each queue
    const uint32_t* tile_order;                     // k-th tile to hand out, expensive ones first; [-1] = how many are expensive, [-2] = how many are handed out at all (the rest is empty, below); null = row-major
    uint32_t cull_empty;                            // 1: empty_tiles_kernel runs - tiles none of whose camera rays can meet anything; 2: ... and the sky is black: such tiles are never handed out, their samples are zeros
                                                    // (1, a sky that depends on the ray: they are handed out flagged, and their camera rays skip the walk)
    uint32_t* culled_word;                          // host-mapped: valid pixels of the call's empty tiles (rt_scene_culled_pixels)
#ifdef RT_PHASE_STATS
    unsigned long long* phase_stats;                // diagnostics build only: [phase][lane-iterations, wave-iterations]
#endif
    uint32_t strip_h, strip_period_rows, strip_first_row;   // strip_h != 0: local row y -> image row (interleaved strips)
    uint32_t single_rounds;                         // launches of fewer blocks than this hand out single ROUNDS of a block (whole-segment units, instantiations without depth of field); 0 = never
    uint32_t col_n, col_r;                          // col_n != 0: the chunk is the packed tile columns col_r, col_r + col_n, ... of the frame (local tile column j -> image tile column j * col_n + col_r)
    uint32_t classify_by_beam;                      // classify_kernel traces nothing: a tile the beam test did not find empty has the scene's one cost class (S.uniform_cost_class), and no launch of this call asks for its circle of confusion
};

// cast_rays_into_scene :645-646: the same two values for every sample of a pixel
__device__ __forceinline__ void pixel_uv(const RenderParams& P, uint32_t px, uint32_t py, float& u, float& v) {
    u = (float)px / (float)P.width;
    v = (float)(P.height - py) / (float)P.height;
}
__device__ __forceinline__ void gen_camera_ray(const DevSceneView& S, float u, float v, float lx, float ly, F3& o, F3& d) {
    // Camera::get_ray :482-491
    const float ldx = lx * S.cam_lens_radius, ldy = ly * S.cam_lens_radius;
    const F3 offset = (ld3(S.cam_u) * ldx) + (ld3(S.cam_v) * ldy);
    o = ld3(S.cam_origin) + offset;
    d = ld3(S.cam_llc) + (ld3(S.cam_horizontal) * u) + (ld3(S.cam_vertical) * v) - ld3(S.cam_origin) - offset;
}

// Sum of the 8 values held by this lane and the 7 lanes STRIDE, 2*STRIDE, ... above it (same row of
// 16 lanes), added one after the other: ((0 + c0) + c1) + ... + c7.  DPP row shifts feed the adds
// directly (no LDS permute); lanes without 7 lanes above them in their row compute a value nobody reads.
template <int CTRL> __device__ __forceinline__ float dpp_read(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int STRIDE> __device__ __forceinline__ float seq_sum8(float c) {
    float s = 0.0f + c;
    s = s + dpp_read<0x100 + 1 * STRIDE>(c); s = s + dpp_read<0x100 + 2 * STRIDE>(c); s = s + dpp_read<0x100 + 3 * STRIDE>(c);
    s = s + dpp_read<0x100 + 4 * STRIDE>(c); s = s + dpp_read<0x100 + 5 * STRIDE>(c); s = s + dpp_read<0x100 + 6 * STRIDE>(c);
    s = s + dpp_read<0x100 + 7 * STRIDE>(c);
    return s;
}

// How many lanes of a wave must be waiting for a new ray before the wave leaves
// the traversal burst to shade / refill them (persistent-thread ray refill).  Every
// threshold below 64 (= synchronous: the burst ends with its last ray) measured slower
// (DESIGN.md 4.3), so it is a build option (-DRT_REFILL_LANES=n), not a run-time one.
#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 64
#endif

// Diagnostics build (make stats): per phase, how many lanes and how many wave
// passes went through it; lane utilisation of a phase = lanes / (64 * passes).
// Second diagnostics build (make time, -DRT_PHASE_TIME): where a wave's CYCLES go - s_memtime at every change of phase.
#ifdef RT_PHASE_TIME
enum { TM_OTHER = 0, TM_NODE, TM_LEAF_TRI, TM_LEAF_TLAS, TM_SHADE, TM_ROUNDS, TM_CLAIM, TM_COUNT };
#define TM(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm_acc[tm_cur] += now_ - tm_last; tm_last = now_; tm_cur = (i); } while (0)
#else
#define TM(i) do { } while (0)
#endif
#ifdef RT_PHASE_STATS
enum { PH_NODE = 0, PH_TRI, PH_ENTITY, PH_SHADE, PH_SHADE_PATH_HIT, PH_SHADE_SHADOW_HIT, PH_STAGE_A, PH_STAGE_C, PH_TRACE_INIT, PH_BURST, PH_LEAF_TRI, PH_LEAF_TLAS, PH_OUTER, PH_POP_INST, PH_NODE_GLOBAL, PH_NODE_NOT_II, PH_INST_HEADER, PH_COUNT };
#ifdef RT_PHASE_TIME
#define PH(i) do { } while (0)
#else
#define PH(i) do { const unsigned long long m_ = __ballot(1); ph_lanes[i]++; ph_waves[i] += (lane == (uint32_t)(__ffsll((long long)m_) - 1)); } while (0)
#endif
#else
#define PH(i) do { } while (0)
#endif

// The kernel is instantiated per combination of scene traits, and a scene runs the instantiation without the
// code and per-lane state it does not need (register pressure is what limits this kernel: weekend +18 %,
// Dreadnaught +4 % over the one-size kernel):
// LOOSE: the scene has entity boxes that do not bound their hits (device_scene.h, loose_boxes): TLAS boxes are
//   then tested against the ray's whole interval, as the reference does for every box (one instantiation,
//   with everything else enabled);
// MESH / SPHERES / PLANES: kinds of hitable among the entities (MESH: the BLAS walk and its state);
// XFORMS: entities with other transforms than exactly one translation (S.general_xforms);
// RICH: lamp lights, or textures other than colours and checkers (S.rich_shading);
// DOF: some primary hit can be out of focus by more than the threshold of the lens-cell rounds (S.coc_rel_max x image
//   width, decided per launch): without it the second round shape is compiled out (Dreadnaught +3 %).
// ECACHE: a scene of plain translated spheres whose entity records fit into LDS beside the node cache (S.lds_ents): leaf tests read them there.
template <bool LOOSE, bool MESH, bool SPHERES, bool PLANES, bool XFORMS, bool RICH, bool DOF, bool MULTI = false, bool ECACHE = false>
__global__ void __launch_bounds__(RT_BLOCK, RT_WAVES_PER_SIMD)
render_kernel(const DevSceneView S, const RenderParams P) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stack = lds_stack + threadIdx.x;
    const uint32_t stack_cap = S.stack_entries;

    // Work items (pixel x sample segment) are dealt in blocks of 64 = one 8x8 pixel
    // tile of one segment.  A wave owns one block at a time and all its lanes move on
    // to the next block together, so the rays of a wave stay spatially coherent.
    // Blocks are claimed dynamically (one atomic per wave and block), so waves that
    // drew cheap tiles simply take more of them.
    // (empty tiles - classify_kernel - are sorted behind all others and never handed out: fold_kernel takes their samples as the zeros they are)
    const uint32_t active_tiles = P.tile_order ? P.tile_order[-2] : P.pix_slots >> 6;
    const uint32_t total_items = active_tiles * 64u * P.unit_count;
    const uint32_t lane = threadIdx.x & 63u;

    // (counting samples and camera rays on the host instead - both are known per launch - frees two registers and was measured: Dreadnaught +0.5 %, 900 instances +0.8 %,
    //  mesh_demo -1.1 %, weekend +-0: register allocation moves more than two registers do.  Not adopted.)
    uint32_t n_primary = 0, n_secondary = 0, n_shadow = 0, n_samples = 0;
    uint32_t queue = (blockIdx.x * (RT_BLOCK / 64u) + (threadIdx.x >> 6)) % RT_QUEUES;      // this wave's current block counter
#if defined(RT_PHASE_STATS) && !defined(RT_PHASE_TIME)
    uint32_t ph_lanes[PH_COUNT] = {}, ph_waves[PH_COUNT] = {};
#endif
#ifdef RT_PHASE_TIME
    unsigned long long tm_acc[TM_COUNT] = {}, tm_last = __builtin_amdgcn_s_memtime(); int tm_cur = TM_OTHER;
#endif

    // Lane state between two rays.  PATH / SHADOW: a ray of that kind is in flight
    // (or, with `tracing` false, has just finished and waits for its shading).
    enum { PATH = 0, SHADOW = 1, NEED_LIGHT = 2, NEED_PATH = 3, NEED_SAMPLE = 4, NEED_BLOCK = 5, FINISHED = 6 };
    int state = NEED_BLOCK;
    bool tracing = false;
#ifdef RT_KIND_ORDER_EXPERIMENT
    bool held = false;                       // a ray is waiting to start
#else
    constexpr bool held = false;
#endif
    uint32_t pxy = 0, s = RT_NONE;           // pixel (x | y << 16) and sample index of the lane's current round (RT_NONE: no sample)
    // the wave's current block (wave-uniform): tile, unit, rounds started
    uint32_t blk_tile = 0, blk_unit = 0, round_next = 0, blk_lens = 0;
    // where the rays of the current block start their walk: the TLAS root - or nowhere, in a tile whose camera rays cannot meet anything (empty_tiles_kernel; a sky that is
    // not black, so the tile is rendered): a camera ray that misses has no child and casts no shadow ray, so every ray of such a block is a camera ray
    // (NOHIT: instantiations without mesh instances only - carrying it cost the mesh scenes 1 % (Dreadnaught 6.92 -> 7.00 ms, mesh_demo 107.0 -> 107.9 ms) and the sample
    //  scenes with meshes have a black sky, under which empty tiles are left out altogether; the host does not flag tiles for the others)
    constexpr bool NOHIT = !MESH;
    uint32_t blk_root = S.tlas_root_ref;
#ifdef RT_CLAIM_AHEAD      /* measurement only: the next block claimed while this one starts, so that the atomic's round trip is hidden - one more register through the walk:
                              weekend -0.8 %, Dreadnaught -1.5 %, mesh_demo -0.5 %, 900 instances -0.9 %, 400x225x16 -8 % (blocks held back by busy waves).  Not adopted. */
    uint32_t next_idx = 0; bool have_next = false;      // the block claimed ahead (lane 0 holds its number)
#endif
    uint32_t round_first = 0, round_end = 0;    // the rounds of the block this claim covers: all of them, or ONE (P.single_rounds)
    uint32_t blk_xy0 = 0, blk_py0 = 0, blk_px0 = 0;   // the tile's first pixel: chunk-local x | y << 16, its image row and its image column
    F3 sample_col = f3(0, 0, 0);
    Pending pend[RT_PENDING_MAX]; uint32_t npend = 0;
    F3 ro = f3(0, 0, 0), rd = f3(0, 0, 1), rw = f3(0, 0, 0); uint32_t rlimit = 0, rnode = 0;   // current path ray
    F3 hp = f3(0, 0, 0), hn = f3(0, 0, 0), wa = f3(0, 0, 0), lsum = f3(0, 0, 0);                  // light phase
    float li = 0.0f;
    uint32_t light_k = 0;
    F3 lcol = f3(0, 0, 0), ldir = f3(0, 0, 1); float l_t = 0.0f, closest = 0.0f;

    // Traversal state of the ray in flight (resumable: survives leaving the burst).
    // The search itself is Scene::hit_closest (implementation.rs:372-375) =
    // EntityBvhRoot::try_hit (:411-416) over Entity::hit (:285-310) over Sphere::hit /
    // Plane::hit / MeshBvhRoot::try_hit (sphere.rs:36-53, plane.rs:59-90, mesh.rs:92-99).
    F3 o = f3(0, 0, 0), d = f3(0, 0, 1), inv = f3(0, 0, 0);
    float t_min = 0.0f, t_max = 0.0f, best_t = 0.0f;
    Hit best; best.t = 0.0f; best.ent = RT_NONE; best.tri = RT_NONE; best.erank = 0;      // (erank: unused here, ranks are fetched on ties)
    uint32_t cur = RT_NONE, sp = 0, inst_sp = RT_NONE, tri_base = 0, cur_ent = 0;
    float ray_dd = 0.0f;                     // ECACHE: |d|^2 of the ray in flight (sphere.rs:6 forms it per test; the same product every time)

    // Node cache: the scene's first S.lds_nodes records (the top levels of the TLAS and of every BLAS,
    // scene_build.cpp) are read from LDS instead of through the vector memory path: the texture
    // addresser is 70-80 % busy with the gathers of this kernel (18 cycles per wave instruction), and
    // every ray walks the top levels (weekend +6 %).
    uint32_t* const node_cache = lds_stack + (stack_cap + (S.mesh_entities ? 3u : 0u)) * RT_BLOCK;      // (three rows park 1/d while a lane is inside an instance; sized by the SCENE, rt_scene_create)
    for (uint32_t i = threadIdx.x; i < S.lds_nodes * 4u; i += RT_BLOCK) {
        const float4 v = *reinterpret_cast<const float4*>(&S.nodes[i >> 2].w[i & 3u]);
        *reinterpret_cast<float4*>(node_cache + (i >> 2) * (RT_NODE_CACHE_STRIDE / 4u) + (i & 3u) * 4u) = v;
    }
    // Sphere cache: in a scene of plain translated spheres a leaf test reads word 0 of the entity record and nothing else; with all of them in LDS
    // the walk of such a scene issues no vector-memory read at all (the leaf pass was latency-bound: 183 SIMD cycles for ~75 of VALU work).
    // An instantiation of its own (ECACHE): a run-time switch between LDS and memory in the leaf test cost scenes WITHOUT the cache 2.4 %.
    static_assert(!ECACHE || (SPHERES && !MESH && !PLANES && !XFORMS && !LOOSE), "the sphere cache holds word 0 of the entity records only");
    uint32_t* const ent_cache = node_cache + S.lds_nodes * (RT_NODE_CACHE_STRIDE / 4u);
    if (ECACHE)
        for (uint32_t i = threadIdx.x; i < S.lds_ents; i += RT_BLOCK) {
            float4 v = *reinterpret_cast<const float4*>(&S.entities[i].w[0]);
            v.w = v.w * v.w;                              // (radius * radius, the product sphere.rs:8 forms per test)
            *reinterpret_cast<float4*>(ent_cache + i * 4u) = v;
        }
    __syncthreads();

    for (;;) {
        PH(PH_OUTER);
        TM(TM_CLAIM);
        // ================= next block of work items: all lanes of the wave together ======
        if (__ballot(state != NEED_BLOCK && state != FINISHED) == 0ull) {
            if (__ballot(state == NEED_BLOCK) != 0ull) {          // (all lanes of a wave are in the same of the two states)
                // Blocks are handed out by RT_QUEUES counters, counter q owning blocks q, q + RT_QUEUES, ...
                // (same-address atomics serialise in L2: one counter cost 3 % of the headline frame and
                // half of a single-sample launch).  A wave starts at its own counter and moves on to the
                // next one when that is exhausted, so the hand-out order stays "expensive first".
                // Single-round claims: a claim is ONE round of a block (64 samples instead of 512), against the tail of launches with few blocks per resident wave (one rank's
                // share of an N-GPU frame, a chunk of the host's chunk grid).  Claims 8b .. 8b + 7 are the rounds of block b.  Same pixels.  A claim is an atomic round trip, a
                // dependent read of the tile order and four integer divisions: on full frames it loses (round 3, every claim of every launch single: weekend -25 %, Dreadnaught -17 %
                // with its sky blocks, -4.5 % without them, mesh_demo -7 %), so only launches that hand out fewer than P.single_rounds blocks (about 6 per resident wave; the kernel
                // knows how many tiles are handed out at all, the host does not) - Dreadnaught's share of 8 ranks 1.57 -> 1.23 ms, of 4 ranks 2.23 -> 2.09 ms.  Not in the instantiations
                // with depth of field: the eight rounds of a LENS-CELL block are chained through one output slot and stay one claim, and mesh_demo's expensive tiles are such blocks
                // (its slowest of 8 ranks: no earlier).  (`make singlerounds` compiles it into every instantiation; RT_SINGLE_ROUNDS=1 / 0 forces / forbids it.)
#ifdef RT_SINGLE_ROUNDS_EXPERIMENT
                const uint32_t single_rounds = (total_items >> 6) < P.single_rounds ? 1u : 0u;
#else
                const uint32_t single_rounds = (!DOF && (total_items >> 6) < P.single_rounds) ? 1u : 0u;
#endif
                const uint32_t total_blocks = single_rounds ? (total_items >> 6) << 3 : total_items >> 6;          // pix_slots is a multiple of 64
                uint32_t blk = total_blocks;
                // Cancellation (the reference checks its token per work item and per pixel, src/render.rs:131,149).  A read of the scene's host-mapped
                // cancel word crosses the host link and costs microseconds - done by every wave at every 4th claim it halved the headline frame rate
                // (10 M reads per second).  So ONE wave of every 64th workgroup is a poller: at each of its claims lane 0 reads the word, and when it is
                // set the poller exhausts all RT_QUEUES block counters (one atomic each): every later claim of every wave then comes back empty and
                // the launch drains within one block per wave.  The other waves pay nothing.
                const bool poller = (blockIdx.x & (RT_CANCEL_POLL_BLOCKS - 1u)) == 0u && threadIdx.x < 64u;
                if (poller) {
                    uint32_t c = 0;
                    if (lane == 0) c = __hip_atomic_load(P.cancel_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (__builtin_amdgcn_readfirstlane(c) != 0u && lane < RT_QUEUES) atomicOr(P.block_counter + lane * 16u, 0x80000000u);      // (idempotent: several pollers do this)
                }
                for (uint32_t tries = 0; tries < RT_QUEUES && blk >= total_blocks; tries++) {
                    uint32_t idx = 0;
#ifdef RT_CLAIM_AHEAD
                    if (tries == 0u && have_next) idx = next_idx;      // claimed while the last block was starting: its round trip is long over
                    else
#endif
                    if (lane == 0) idx = atomicAdd(P.block_counter + queue * 16u, 1u);
                    idx = __builtin_amdgcn_readfirstlane(idx);
                    blk = idx < (total_blocks + RT_QUEUES - 1u) / RT_QUEUES ? idx * RT_QUEUES + queue : total_blocks;      // (guards the multiply)
                    if (blk >= total_blocks) { blk = total_blocks; queue = (queue + 1u) % RT_QUEUES; }
                }
                // progress (src/render.rs:156-158 sends a snapshot every 800 ms): the queues advance evenly, so the number of the block just claimed
                // is about the number of blocks handed out; the pollers store it (plain stores, last writer wins - an estimate, not a count)
                if (poller && lane == 0 && blk < total_blocks)      // (+ the blocks of the empty tiles: done before the launch began)
                    __hip_atomic_store(P.progress_word, P.progress_base + ((P.pix_slots >> 6) - active_tiles) * P.unit_count + (single_rounds ? blk >> 3 : blk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef RT_CLAIM_AHEAD
                have_next = blk < total_blocks;
                if (have_next && lane == 0) next_idx = atomicAdd(P.block_counter + queue * 16u, 1u);      // the NEXT block of this queue; looked at when this one is done
#endif
                if (blk >= total_blocks) state = FINISHED;
                else {
                    round_first = 0u; round_end = P.unit_len;
                    if (single_rounds) { round_first = blk & 7u; round_end = round_first + 1u; blk >>= 3; }
                    // Block numbering: first every unit of each tile whose rays hit something, tile after
                    // tile in cost order (so that the long blocks all start early), then the tiles that
                    // see only sky, unit by unit over the whole image.
                    const uint32_t n_tiles = active_tiles;
                    const uint32_t heavy = P.tile_order ? P.tile_order[-1] : 0u, heavy_blocks = heavy * P.unit_count;
                    uint32_t tile_rank, unit_local;
                    if (blk < heavy_blocks) { tile_rank = blk / P.unit_count; unit_local = blk % P.unit_count; }
                    else { const uint32_t b = blk - heavy_blocks, rest = n_tiles - heavy; unit_local = b / rest; tile_rank = heavy + b % rest; }
                    blk_tile = P.tile_order ? P.tile_order[tile_rank] : tile_rank;      // bit 31: the tile's rounds go by lens cell
                    blk_lens = (DOF && P.lens_rounds) ? blk_tile >> 31 : 0u;
                    if (NOHIT) blk_root = ((blk_tile >> 30) & 1u) ? RT_NONE : S.tlas_root_ref;
                    blk_tile &= 0x3fffffffu;
                    {
                        const uint32_t x0 = (blk_tile % P.tiles_x) * 8u, y0 = (blk_tile / P.tiles_x) * 8u;
                        blk_xy0 = x0 | (y0 << 16);
#ifdef RT_NO_TILE_COLUMNS      /* measurement only: what the column mapping costs the kernel */
                        blk_px0 = P.left + x0;
#else
                        blk_px0 = P.col_n ? ((x0 >> 3) * P.col_n + P.col_r) * 8u : P.left + x0;      // (dealt tile columns: rt_render_tile_columns_device)
#endif
                        // (strips are whole rows of tiles, so the 8 rows of a tile are consecutive image rows)
                        blk_py0 = P.strip_h ? P.strip_first_row + (y0 / P.strip_h) * P.strip_period_rows + (y0 % P.strip_h) : P.top + y0;
                    }
                    {
                        // the units of the block: 8 (2^k) consecutive ones, or - lens-cell rounds over whole groups of 16
                        // units - every second one of 16, so that the 8 samples of a round share one of the 16 lens cells of
                        // their 16-sample groups.  blk_unit = first unit; blk_lens = round shape | unit stride shift << 1 (two bits) | pixel group << 3
                        // (RT_LENS_CELLS 64: every 8th unit of 64, where the launch holds whole groups of 64 units - the 8 x 8 lens cells of groups of 64 samples)
                        const uint32_t k = P.round_units_shift;
                        uint32_t t = 0u;
                        if (DOF && blk_lens) t = (RT_LENS_CELLS == 64 && (P.unit_count & 63u) == 0u && (P.unit_first & 63u) == 0u) ? 3u : ((P.unit_count & 15u) == 0u ? 1u : 0u);
                        const uint32_t m = unit_local >> k;
                        blk_lens |= (t << 1) | ((unit_local & ((1u << k) - 1u)) << 3);
                        blk_unit = t ? ((m >> t) << (k + t)) + (m & ((1u << t) - 1u)) : (m << k);
                    }
                    s = RT_NONE;                          // the block's rays are dealt to the lanes round by round, below
                    state = NEED_SAMPLE;
                    if (single_rounds && DOF && (blk_lens & 1u)) {      // a lens-cell block: whole, under its first claim
                        if (round_first != 0u) state = NEED_BLOCK;
                        round_end = P.unit_len;
                    }
                    round_next = round_first;
                }
            }
            if (__ballot(state != FINISHED) == 0ull) break;      // the wave has no blocks left
        }
        if (!tracing && !held && state != NEED_BLOCK && state != FINISHED) {
            // ================= shading of the ray that just finished =====================
            PH(PH_SHADE);
            TM(TM_SHADE);
            if (state == SHADOW) {
                // cast_light_ray_to_lamp, implementation.rs:535-552
                n_shadow++;
                state = NEED_LIGHT;
                if (best.ent == RT_NONE) {
                    lsum = lsum + (lcol * fmaxf(0.0f, dot(hn, -ldir)));                    // :615-619
                    light_k++;
                } else {
                    // Only a dielectric can let light through (scatter yields a refraction only
                    // there, materials.rs:45-53,90-112,168-190): look at the material first,
                    // rebuild the hit record only if needed.
                    PH(PH_SHADE_SHADOW_HIT);
                    const uint32_t mat_id = ldw(S.entities[best.ent].w + 1).u[3];
                    const Word4* mw = S.materials[mat_id].w;
                    const Word4 m0 = ldw(mw);
                    if (m0.u[0] == RT_MAT_DIELECTRIC) {
                        const F3 td = -ldir;
                        HitRec hr; resolve_hit<XFORMS, MESH, SPHERES, PLANES>(S, hp, td, best, hr);
                        float nint; F3 outn;
                        const float kr = dielectric_kr(td, hr.normal, ldw(mw + 1).f[0], nint, outn);
                        if (!(kr >= 1.0f)) {
                            const float refr_intensity = (1.0f - kr) * (1.0f - m0.f[3]);
                            const F3 albedo = texture_value<RICH>(S, hr.tex_id, hr);
                            lcol = lcol * (albedo * refr_intensity);                          // :541-543
                            closest = best.t + RT_BIAS;
                            // NOT the reference: beyond t = 65 536 the sum rounds back to t, a plane or a triangle accepts t == t_min, and the
                            // reference's loop finds the same occluder for ever (the pixel never returns; here: a kernel that never ends).
                            // Step one ulp past the occluder instead (t > 0), as the oracle does.
                            if (!(closest > best.t)) closest = __uint_as_float(__float_as_uint(best.t) + 1u);
                            if (closest > best.t) state = SHADOW;                             // same ray, further on (an occluder at t = inf: no light, as if opaque)
                        }
                    }
                    if (state == NEED_LIGHT) light_k++;                                      // opaque: ZERO (:548)
                }
            } else if (state == PATH) {
                // cast_ray_recursive, implementation.rs:566-631
                if (rnode == 1u) n_primary++; else n_secondary++;
                if (best.ent == RT_NONE) { sample_col = sample_col + (rw * color_sky(S, rd)); rnode = 0u; state = NEED_PATH; }
                else {
                    PH(PH_SHADE_PATH_HIT);
                    HitRec hr; resolve_hit<XFORMS, MESH, SPHERES, PLANES>(S, ro, rd, best, hr);
                    const Word4* mw = S.materials[hr.mat_id].w;
                    const Word4 m0 = ldw(mw);
                    // Material::scatter
                    bool has_refl = false, has_refr = false; F3 refl_d = f3(0, 0, 0), refr_d = f3(0, 0, 0); float refl_i = 0.0f, refr_i = 0.0f;
                    if (m0.u[0] == RT_MAT_LAMBERTIAN) {                                       // materials.rs:44-53
                        has_refl = true; refl_i = m0.f[1];
                        if (refl_i > 0.0f) {      // the direction is only ever used when the ray is cast (:593)
                            float r0, r1; rng_pair(P.seed, pxy, s, rnode, PURPOSE_PATH, r0, r1);
                            refl_d = random_normal_reflection_angle(hr.normal, r0, r1);
                        }
                    } else if (m0.u[0] == RT_MAT_SPECULAR) {                                  // materials.rs:89-112
                        const F3 reflected = reflect(rd, hr.normal);
                        F3 scattered = reflected;
                        if (m0.f[2] != 0.0f) {
                            float r0, r1; rng_pair(P.seed, pxy, s, rnode, PURPOSE_PATH, r0, r1);
                            scattered = reflected + (unit(random_normal_reflection_angle(hr.normal, r0, r1)) * m0.f[2]);
                        }
                        if (dot(scattered, hr.normal) > 0.0f) { has_refl = true; refl_d = scattered; refl_i = m0.f[1]; }
                    } else {                                                                  // materials.rs:167-204
                        float ni_over_nt; F3 outward_normal;
                        const float kr = dielectric_kr(rd, hr.normal, ldw(mw + 1).f[0], ni_over_nt, outward_normal);
                        if (!(kr >= 1.0f)) {
                            has_refr = true;
                            refr_d = unit(refract(rd, outward_normal, ni_over_nt));
                            refr_i = (1.0f - kr) * (1.0f - m0.f[3]);
                        }
                        has_refl = true; refl_d = unit(reflect(rd, hr.normal)); refl_i = kr * m0.f[1];
                    }
                    // depth budget split, :577-589
                    const uint32_t rl = rlimit - 1u;
                    uint32_t reflect_limit = 0u, refract_limit = 0u;
                    if (has_refl && has_refr) { reflect_limit = rl / 2u; refract_limit = rl - reflect_limit; }
                    else if (has_refl) reflect_limit = rl;
                    else if (has_refr) refract_limit = rl;
                    const bool cast_refl = has_refl && refl_i > 0.0f, cast_refr = has_refr && refr_i > 0.0f;   // :593, :601
                    const float reflection_intensity = cast_refl ? refl_i : 0.0f, refraction_intensity = cast_refr ? refr_i : 0.0f;
                    const F3 albedo = texture_value<RICH>(S, hr.tex_id, hr);                        // :626
                    wa = rw * albedo;
                    li = fmaxf(0.0f, 1.0f - (reflection_intensity + refraction_intensity));   // :625
                    // children: colour is linear in them, so they carry weight w*albedo*intensity.
                    // The reflection is evaluated first (:592-605): it becomes the lane's next path ray
                    // in place (ro/rd/rw are dead from here on); only a second child, the refraction,
                    // goes to the deferred list.  rnode == 0 afterwards means "no child waiting in place".
                    const uint32_t parent = rnode;
                    rnode = 0u;
                    bool refr_in_place = false;
                    if (cast_refr) {
                        const F3 cw_ = wa * refraction_intensity;
                        if (refract_limit == 0u) sample_col = sample_col + (cw_ * color_sky(S, refr_d));   // :562-564
                        else if (!(cast_refl && reflect_limit != 0u)) refr_in_place = true;
                        else if (npend < RT_PENDING_MAX) { Pending& q = pend[npend++]; q.o = hr.p; q.d = refr_d; q.w = cw_; q.limit = refract_limit; q.node = parent * 2u + 1u; }
                        else atomicOr(P.error_flag, RT_DEVERR_PENDING);
                    }
                    if (cast_refl) {
                        const F3 cw_ = wa * reflection_intensity;
                        if (reflect_limit == 0u) sample_col = sample_col + (cw_ * color_sky(S, refl_d));
                        else { ro = hr.p; rd = refl_d; rw = cw_; rlimit = reflect_limit; rnode = parent * 2u; }
                    }
                    if (refr_in_place) { ro = hr.p; rd = refr_d; rw = wa * refraction_intensity; rlimit = refract_limit; rnode = parent * 2u + 1u; }
                    // lights, :608-621
                    hp = hr.p + (hr.normal * RT_BIAS); hn = hr.normal;
                    lsum = f3(0.0f, 0.0f, 0.0f); light_k = 0u;
                    state = NEED_LIGHT;
                }
            }
            // ================= next ray of this lane =======================================
            // stage A: next light of the current hit (implementation.rs:612-621)
            if (state == NEED_LIGHT) {
                PH(PH_STAGE_A);
                while (light_k < S.light_count) {
                    LightRec lr;
                    if (light_record<RICH>(S, light_k, hp, lr)) {
                        lcol = lr.color * lr.intensity; ldir = lr.direction; l_t = lr.t; closest = RT_BIAS;   // :527-532
                        state = SHADOW;
                        break;
                    }
                    light_k++;
                }
                if (state == NEED_LIGHT) {
                    // (color_from_lights * lights_intensity) * albedo, weighted (:628-630)
                    sample_col = sample_col + (wa * (lsum * li));
                    state = NEED_PATH;
                }
            }
            // stage B: next ray of the current sample's recursion tree
            if (state == NEED_PATH) {
                if (rnode != 0u) state = PATH;           // the child left in place by the last hit
                else if (npend != 0) {
                    const Pending& q = pend[--npend];
                    ro = q.o; rd = q.d; rw = q.w; rlimit = q.limit; rnode = q.node;
                    state = PATH;
                } else {
                    // cast_ray's clamp (:635, types.rs:46-50) then col = col + sample (:656)
                    // (the sum over the samples is formed when the round ends, below)
                    sample_col = f3(fmaxf(fminf(sample_col.x, 1.0f), -1.0f), fmaxf(fminf(sample_col.y, 1.0f), -1.0f), fmaxf(fminf(sample_col.z, 1.0f), -1.0f));
                    state = NEED_SAMPLE;
                }
            }
        }
        // ================= rounds =====================================================================
        // A block = 8 rounds: 8 pixels x 8 units with 64 or more samples per pixel, else the 64 pixels of one 8x8
        // tile x the samples of one unit (or fewer pixels x more units).  The wave traces it in rounds
        // of 64 rays and every lane waits for the round to end, so the rays traced side by side are of one
        // kind (camera rays with camera rays, shadow rays with shadow rays), which keeps the lanes in step
        // through the tree (measured against free-running lanes: weekend +5 %, mesh_demo +41 %).
        // A round holds samples of as few pixels as possible: up to 8 units of a tile share their rounds, so
        // with 64 or more samples per pixel a round is 64 samples of ONE pixel (lane = unit * 8 + sample),
        // with fewer units 2, 4 or 8 pixels of a compact group; one-sample units (fine split): one sample of
        // all 64 pixels.  Samples of one pixel meet the same materials, so the lanes of a round have paths
        // of similar length, and a wave waits for the longest path of one pixel instead of 64 (measured
        // against one sample of 64 pixels per round: Dreadnaught +32 %, instanced meshes +76 %).
        // The clamped samples of a unit are added in sample order, ((0 + c0) + c1) + ..., by handing them to
        // the unit's first lane one after the other (cast_rays_into_scene's col = col + sample, :656).
        // Tiles out of focus (blk_lens, set per tile by classify_kernel from the circle of confusion) run the
        // other round shape: round r = the r-th sample of 8 units for the 4x2 pixels of the block, lane = unit * 8
        // + pixel.  With the stratified lens sampler all 64 rays of such a round leave the lens from cell r, so
        // they stay together where the 64 lens points of one pixel would spread over the circle of confusion
        // (weekend +9 %, mesh_demo +18 %).  A lane then owns one pixel of one unit and adds its samples into
        // the output slot round by round: the same additions in the same order.
        TM(TM_ROUNDS);
        if (__ballot(tracing || state != NEED_SAMPLE) == 0ull) {        // wave-uniform: every lane has finished its sample
            // lane = (unit of the group, sample of the unit, pixel of the round); all sizes are powers of two
            const uint32_t l_shift = P.unit_len == 1u ? 0u : RT_SEGMENT_SHIFT, k_shift = P.round_units_shift;
            const bool lens = DOF && (blk_lens & 1u) != 0u;
            const uint32_t g_shift = lens ? 3u : 6u - l_shift - k_shift;                    // log2(pixels per round)
            const uint32_t q = lane & ((1u << g_shift) - 1u), j = lane >> g_shift;
            const uint32_t u_in = lens ? j : j >> l_shift, jj = lens ? 0u : j & (P.unit_len - 1u);
            const uint32_t sub = blk_lens >> 3, unit_local = blk_unit + (u_in << ((blk_lens >> 1) & 3u));
            const uint32_t x0 = blk_xy0 & 0xffffu, y0 = blk_xy0 >> 16;
            const uint32_t s_mine = (P.unit_first + unit_local) * P.unit_len + jj;
            const bool s_ok = s_mine >= P.sample_first && s_mine < P.sample_first + P.sample_count;
            // pixel of this lane in round g: Z-order inside the tile, so that the pixels of a round and of a
            // block form compact rectangles (8 -> 4x2, 4 -> 2x2, 2 -> 2x1)
            auto round_pixel = [&](uint32_t g, uint32_t& x, uint32_t& y) {
                const uint32_t z = lens ? (sub << 3) + q : (sub << (6u - k_shift)) + (g << g_shift) + q;
                x = (z & 1u) | ((z >> 1) & 2u) | ((z >> 2) & 4u);
                y = ((z >> 1) & 1u) | ((z >> 2) & 2u) | ((z >> 3) & 4u);
            };
            for (;;) {
                if (round_next != round_first) {
                    // close the round: segment sums of its pixels, ((0 + c0) + c1) + ... in sample order
                    const F3 c = s != RT_NONE ? sample_col : f3(0.0f, 0.0f, 0.0f);     // (x + 0 == x: the sum is never -0)
                    F3 sum = f3(0.0f, 0.0f, 0.0f);
                    static_assert(RT_SAMPLE_SEGMENT == 8u, "seq_sum8 adds the 8 samples of a segment");
                    if (lens) sum = c;             // (a lane owns one pixel of one unit: its sum is carried in the output slot, below)
                    else if (l_shift != 0u && g_shift == 0u) sum = f3(seq_sum8<1>(c.x), seq_sum8<1>(c.y), seq_sum8<1>(c.z));          // the unit's samples: 8 consecutive lanes
                    else if (l_shift != 0u && g_shift == 1u) sum = f3(seq_sum8<2>(c.x), seq_sum8<2>(c.y), seq_sum8<2>(c.z));     // every second lane
                    else for (uint32_t k = 0; k < P.unit_len; k++) {
                        const int src = (int)(((((u_in << l_shift) + k)) << g_shift) | q);
                        sum = sum + f3(__shfl(c.x, src, 64), __shfl(c.y, src, 64), __shfl(c.z, src, 64));
                    }
                    uint32_t x, y; round_pixel(round_next - 1u, x, y);
                    if (jj == 0u && x0 + x < P.cw && y0 + y < P.ch) {
                        float* out = P.partial + ((uint64_t)(unit_local * (P.pix_slots >> 6) + blk_tile) * 64u + (y * 8u + x)) * 3;
                        if (lens) {                // ((0 + c0) + c1) + ...: the same additions, one per round
                            const F3 acc = round_next == 1u ? f3(0.0f, 0.0f, 0.0f) : f3(out[0], out[1], out[2]);
                            sum = acc + sum;
                        }
                        out[0] = sum.x; out[1] = sum.y; out[2] = sum.z;
                    }
                }
                if (round_next == round_end) { state = NEED_BLOCK; break; }      // wait for the wave's next block
                uint32_t x, y; round_pixel(round_next, x, y);
                round_next++;
                if (lens) { const uint32_t sl = s_mine + (round_next - 1u); s = (x0 + x < P.cw && y0 + y < P.ch && sl >= P.sample_first && sl < P.sample_first + P.sample_count) ? sl : RT_NONE; }
                else
                s = (x0 + x < P.cw && y0 + y < P.ch && s_ok) ? s_mine : RT_NONE;      // lanes off the chunk's edge sit the round out
                if (s != RT_NONE) {
                    PH(PH_STAGE_C);
                    const uint32_t px = blk_px0 + x, py = blk_py0 + y;
                    pxy = px | (py << 16);
                    float pu, pv; pixel_uv(P, px, py, pu, pv);
                    float lx = 0.0f, ly = 0.0f;
                    if (P.spp > 1) {                                                  // :648-653
                        float r0, r1; rng_pair(P.seed, pxy, s, 0u, PURPOSE_LENS, r0, r1);
                        // Lens samples are stratified: inside complete groups of 64 samples, sample s lies in one of 8 x 8 cells over the lens square
                        // (s mod 16 names the 4 x 4 cell, bits 4 and 5 of s the quarter of it); after them in cell s mod 16 of a 4 x 4 grid
                        // (complete groups of 16 samples; 4 x 2 cells for a complete segment of 8 after them), jittered inside the cell (the reference draws them
                        // independently: same expectation, less noise - and the rays of one sample index of
                        // neighbouring pixels leave the lens close together, see the round shapes above).
                        if (!P.lens_stratify) { }                                           // RT_OPT_LENS_SAMPLER 0: the reference's two independent draws
#if RT_LENS_CELLS == 64
                        else if (((s >> 6) + 1u) * 64u <= P.spp) {                         // a complete group of 64 samples: 8 x 8 cells; s mod 16 still names the 4 x 4 cell
                            const uint32_t cx = 2u * (s & 3u) + ((s >> 4) & 1u), cy = 2u * ((s >> 2) & 3u) + ((s >> 5) & 1u);
                            r0 = ((float)cx + r0) * 0.125f; r1 = ((float)cy + r1) * 0.125f;
                        }
#endif
                        else if (((s >> 4) + 1u) * 16u <= P.spp) {                         // a complete group of 16 samples: 4 x 4 cells
                            const uint32_t cell = s & 15u;
                            r0 = ((float)(cell & 3u) + r0) * 0.25f; r1 = ((float)(cell >> 2) + r1) * 0.25f;
                        } else if (((s >> RT_SEGMENT_SHIFT) + 1u) * RT_SAMPLE_SEGMENT <= P.spp) {   // a complete segment of 8 after them: 4 x 2 cells
                            const uint32_t cell = s & 7u;
                            r0 = ((float)(cell & 3u) + r0) * 0.25f; r1 = ((float)(cell >> 2) + r1) * 0.5f;
                        }
                        lx = r0 * 2.0f - 1.0f; ly = r1 * 2.0f - 1.0f;
                    }
                    gen_camera_ray(S, pu, pv, lx, ly, ro, rd);
                    rw = f3(1.0f, 1.0f, 1.0f); rlimit = P.max_reflections; rnode = 1u;
                    sample_col = f3(0.0f, 0.0f, 0.0f); npend = 0; n_samples++;
                    if (rlimit == 0) {                                                // :562-564: sky, no ray cast
                        const F3 sky = color_sky(S, rd);
                        sample_col = f3(fmaxf(fminf(sky.x, 1.0f), -1.0f), fmaxf(fminf(sky.y, 1.0f), -1.0f), fmaxf(fminf(sky.z, 1.0f), -1.0f));
                    } else state = PATH;
                }
                if (__ballot(state == PATH) != 0ull) break;      // (a round without a single ray: go on to the next)
            }
        }
        if (!tracing && state != NEED_BLOCK && state != FINISHED) {
            // ================= start the traversal of the new ray ==========================
            // RT_KIND_ORDER_EXPERIMENT (`make kindorder`, RT_KIND_ORDER=1; measured neutral, and merely compiled in it costs weekend and Dreadnaught 1 %): the rays that start together are of one kind - the shadow rays towards the lowest-numbered light any lane has
            // waiting, path rays only when no lane has a shadow ray left; the other lanes are held back for a later burst (their order of
            // operations is unchanged: only the schedule moves, never a pixel).
            bool go = state == PATH || state == SHADOW;
#ifdef RT_KIND_ORDER_EXPERIMENT
            if (go && P.kind_order) {
                uint32_t first_light = 0xffu;
                for (uint32_t k = 0; k < S.light_count; k++)
                    if (__ballot(state == SHADOW && light_k == k) != 0ull) { first_light = k; break; }
                if (first_light != 0xffu) go = state == SHADOW && light_k == first_light;
                held = !go;
            }
#endif
            if (go) {
                PH(PH_TRACE_INIT);
                TM(TM_OTHER);
                const bool shadow = state == SHADOW;
                o = shadow ? hp : ro;
                d = shadow ? -ldir : rd;
                inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                if (MESH && S.mesh_entities) {      // read back when the walk leaves an instance (three divisions saved per exit)
                    stack[stack_cap * RT_BLOCK] = __float_as_uint(inv.x); stack[(stack_cap + 1u) * RT_BLOCK] = __float_as_uint(inv.y);
                    stack[(stack_cap + 2u) * RT_BLOCK] = __float_as_uint(inv.z);
                }
                t_min = shadow ? closest : RT_BIAS; t_max = shadow ? l_t : RT_F32_MAX;
                best.t = t_max; best.ent = RT_NONE; best.tri = RT_NONE; best_t = t_max;
                sp = 0; inst_sp = RT_NONE;
                float e;
                cur = slab(S.tlas_min[0], S.tlas_min[1], S.tlas_min[2], S.tlas_max[0], S.tlas_max[1], S.tlas_max[2], o, inv, t_min, best_t, e)
                          ? (NOHIT ? blk_root : S.tlas_root_ref) : RT_NONE;      // (blk_root: a tile the packet test found clear of every leaf box has no candidate for any of its rays - hit_closest is None without a walk)
                tracing = true;
                if (ECACHE) ray_dd = dot(d, d);
            }
        }
        if (__ballot(tracing) == 0ull) continue;   // nothing to traverse: the wave is between blocks

        // ================= traversal burst ===================================================
        // next deferred subtree; leaving a BLAS restores the world-space ray
        auto pop = [&]() -> uint32_t {
            if (MESH && sp == inst_sp) {
                PH(PH_POP_INST);
                inst_sp = RT_NONE;
                const bool shadow = state == SHADOW;
                o = shadow ? hp : ro; d = shadow ? -ldir : rd;
                // the world-space reciprocals were parked in this lane's LDS column at the start of the ray
                inv = f3(__uint_as_float(stack[stack_cap * RT_BLOCK]), __uint_as_float(stack[(stack_cap + 1u) * RT_BLOCK]),
                         __uint_as_float(stack[(stack_cap + 2u) * RT_BLOCK]));
            }
            if (sp == 0) return RT_NONE;
            return stack[(--sp) * RT_BLOCK];
        };
#if defined(RT_PHASE_STATS) && !defined(RT_PHASE_TIME)
        // per burst: which kinds of ray it holds (bit 0 camera rays, 1 secondary path rays, 2 shadow rays), how many lanes start, iterations, lane-iterations
        const uint32_t bk_kind = (__ballot(tracing && state == PATH && rnode == 1u) ? 1u : 0u) | (__ballot(tracing && state == PATH && rnode != 1u) ? 2u : 0u) |
                                 (__ballot(tracing && state == SHADOW) ? 4u : 0u);
        const uint32_t bk_start = (uint32_t)__popcll(__ballot(tracing));
        uint32_t bk_iters = 0, bk_lane_iters = 0;
#endif
        for (;;) {
#if defined(RT_PHASE_STATS) && !defined(RT_PHASE_TIME)
            bk_iters++; bk_lane_iters += (uint32_t)__popcll(__ballot(tracing));
#endif
            if (tracing) {
                PH(PH_BURST);
                TM(TM_NODE);
                // ---- phase 1: walk inner nodes until this lane holds a leaf (RT_NONE has
                // the leaf bit set and ends the walk too).  One 64-byte record = two boxes.
                while (cur < RT_REF_LEAF_BIT) {
                    PH(PH_NODE);
                    Word4 na, nb, nc, nr;
                    if (cur < S.lds_nodes) {
                        const float4* c = reinterpret_cast<const float4*>(node_cache + cur * (RT_NODE_CACHE_STRIDE / 4u));
                        const float4 v0 = c[0], v1 = c[1], v2 = c[2], v3 = c[3];
                        na.f[0] = v0.x; na.f[1] = v0.y; na.f[2] = v0.z; na.f[3] = v0.w; nb.f[0] = v1.x; nb.f[1] = v1.y; nb.f[2] = v1.z; nb.f[3] = v1.w;
                        nc.f[0] = v2.x; nc.f[1] = v2.y; nc.f[2] = v2.z; nc.f[3] = v2.w; nr.f[0] = v3.x; nr.f[1] = v3.y; nr.f[2] = v3.z; nr.f[3] = v3.w;
                    } else {
                        // (32-bit byte offset from the uniform base: one shift instead of a 64-bit shift and add; rt_scene_create refuses scenes of 2^26 records)
                        const Word4* n = reinterpret_cast<const Word4*>(reinterpret_cast<const char*>(S.nodes) + (cur << 6));
                        na = ldw(n); nb = ldw(n + 1); nc = ldw(n + 2); nr = ldw(n + 3);
#if defined(RT_PHASE_STATS) && !defined(RT_PHASE_TIME)      /* how many record reads from memory are of records with two INNER children (what a compact inner record could save) */
                        PH(PH_NODE_GLOBAL);
                        if (!(nr.u[0] < RT_REF_LEAF_BIT && nr.u[1] < RT_REF_LEAF_BIT)) PH(PH_NODE_NOT_II);
#endif
#ifdef RT_EXTRA_NODE_LOAD      /* measurement only: what ONE more 16-byte read of the same line costs per node step - the slope along which a 48-byte record would gain */
                        { typedef float v4f __attribute__((ext_vector_type(4))); const v4f extra = *reinterpret_cast<const volatile v4f*>(n + 3); asm volatile("" ::"v"(extra.x), "v"(extra.y), "v"(extra.z), "v"(extra.w)); }
#endif
                    }
                    float el, er; bool hl, hr;
                    slab2(na, nb, nc, o, inv, t_min, (LOOSE && inst_sp == RT_NONE) ? t_max : best_t, hl, hr, el, er);
                    if (hl && hr) {
                        const bool left_first = el <= er;
                        const uint32_t far_ref = left_first ? nr.u[1] : nr.u[0];
                        cur = left_first ? nr.u[0] : nr.u[1];
                        if (sp < stack_cap) stack[(sp++) * RT_BLOCK] = far_ref;
                        else atomicOr(P.error_flag, RT_DEVERR_STACK);
                    } else if (hl || hr) cur = hl ? nr.u[0] : nr.u[1];
                    else cur = pop();
                }
                // ---- phase 2: the leaf ------------------------------------------------------
                if (cur != RT_NONE) {
                    const uint32_t first = cur & RT_REF_FIRST_MASK;
                    const uint32_t count = ((cur >> RT_REF_COUNT_SHIFT) & 7u) + 1u;
                    if (MESH && inst_sp != RT_NONE) {
                        PH(PH_LEAF_TRI);
                        TM(TM_LEAF_TRI);
                        // try_hit_tri, mesh.rs:18-76 (normal precomputed with the same ops)
                        for (uint32_t k = 0; k < count; k++) {
                            PH(PH_TRI);
                            const uint32_t ti = tri_base + first + k;
                            const Word4* tw = rec32(S.tri_hot, ti, (uint32_t)sizeof(DevTriHot));
                            const Word4 ta = ldw(tw), tb = ldw(tw + 1), tc = ldw(tw + 2);
                            const F3 A = w3(ta), B = w3(tb), C = w3(tc), normal = f3(ta.f[3], tb.f[3], tc.f[3]);
                            const float denominator = dot(d, normal);                       // plane.rs:7-10
                            if (fabsf(denominator) < 1.0e-6f) continue;
                            const float t = dot(A - o, normal) / denominator;               // plane.rs:12-13
                            if (t < t_min || t > t_max) continue;                           // mesh.rs:28
                            if (t > best.t) continue;                                       // cannot win
                            const F3 p = o + (d * t);
                            const F3 n_abp = cross(B - A, p - A);
                            if (dot(normal, n_abp) < 0.0f) continue;
                            const F3 n_bcp = cross(C - B, p - B);
                            if (dot(normal, n_bcp) < 0.0f) continue;
                            const F3 n_cap = cross(A - C, p - C);
                            if (dot(normal, n_cap) < 0.0f) continue;
                            bool better = best.ent == RT_NONE || t < best.t;
                            if (!better && t == best.t) {
                                // exact tie: the reference keeps the candidate it visits first
                                // (implementation.rs:415, mesh.rs:96-98); ranks are only fetched here
                                // (ranks are fetched here only: an exact tie is rare, a hit is not)
                                better = cur_ent == best.ent ? (best.tri != RT_NONE && S.tri_rank[ti] < S.tri_rank[best.tri])
                                                             : entity_rank(S, cur_ent) < entity_rank(S, best.ent);
                            }
                            if (better) { best.t = t; best.ent = cur_ent; best.tri = ti; best_t = fminf(t * RT_CULL_SLACK, t_max); }
                        }
                        cur = pop();
                    } else {
                        // TLAS leaf: its entities in order; a mesh entity defers the rest of the
                        // leaf and turns the walk into its BLAS.
                        PH(PH_LEAF_TLAS);
                        TM(TM_LEAF_TLAS);
                        const bool shadow = state == SHADOW;
                        // (without mesh instances the walk never leaves world space: o and d ARE the world ray)
                        const F3 wo = MESH ? (shadow ? hp : ro) : o, wd = MESH ? (shadow ? -ldir : rd) : d;
                        bool entered = false;
                        for (uint32_t k = 0; k < count && !entered; k++) {
                            PH(PH_ENTITY);
                            const uint32_t ei = first + k;
                            const Word4* ew = rec32(S.entities, ei, (uint32_t)sizeof(DevEntity));
                            // A leaf of the reference holds up to two entities and passes when the UNION of their boxes does; the reference then
                            // runs Entity::hit on both, and the walk may reach the leaf long after its box was tested (popped from the stack, the best
                            // hit has moved closer since).  For a mesh instance Entity::hit is the whole ray transform, three divisions and the BLAS root
                            // test - spared here when the ray, cut at the CURRENT culling bound, misses the instance's OWN world box
                            // (calculate_origin_aabb: it contains the rotated mesh box, hence every hit Entity::hit could report) by more than
                            // RT_CULL_SLACK.  `inv` is the world ray's here (a lane at a TLAS leaf is never inside an instance).  900 instances +16 %,
                            // mesh_demo +5 %.  In mesh-only scenes the test comes before the entity record is even fetched.  MULTI: the scene has mesh
                            // instances AND more than one entity; a lone entity's box is the root box, and merely compiling the test in costs that
                            // instantiation 1-6 % (register allocation), so single-entity scenes (Dreadnaught) run instantiations without it.
                            auto misses_own_box = [&]() -> bool {
                                const Word4 bl = ldw(rec32(S.ent_box, ei, 32u)), bh = ldw(rec32(S.ent_box, ei, 32u) + 1);
                                const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
                                const float ax = ((nx ? bh.f[0] : bl.f[0]) - wo.x) * inv.x, bx = ((nx ? bl.f[0] : bh.f[0]) - wo.x) * inv.x;
                                const float ay = ((ny ? bh.f[1] : bl.f[1]) - wo.y) * inv.y, by = ((ny ? bl.f[1] : bh.f[1]) - wo.y) * inv.y;
                                const float az = ((nz ? bh.f[2] : bl.f[2]) - wo.z) * inv.z, bz = ((nz ? bl.f[2] : bh.f[2]) - wo.z) * inv.z;
                                const float tn = fmaxf(fmaxf(fmaxf(ax, ay), az), t_min);
                                const float tf = fminf(fminf(fminf(bx, by), bz), best_t);
                                return tf * RT_CULL_SLACK < tn;
                            };
                            constexpr bool MESH_ONLY = MESH && !SPHERES && !PLANES;
                            if (MULTI && MESH_ONLY && !LOOSE && RT_OWN_BOX_PRETEST && misses_own_box()) continue;
                            Word4 e0, e1;
                            if (ECACHE) {      // (e1 is dead in this instantiation)
                                const float4 v = *reinterpret_cast<const float4*>(ent_cache + ei * 4u);
                                e0.f[0] = v.x; e0.f[1] = v.y; e0.f[2] = v.z; e0.f[3] = v.w; e1 = e0;
                            } else { e0 = ldw(ew); e1 = ldw(ew + 1); }
                            const uint32_t kind = e1.u[0] & 0xff;
                            if (MULTI && MESH && !MESH_ONLY && !LOOSE && RT_OWN_BOX_PRETEST && kind == RT_HITABLE_MESH && misses_own_box()) continue;
                            F3 eo = wo, ed = wd;
                            ray_to_entity<XFORMS>(S, e0, e1, eo, ed);
                            float t = 0.0f; bool hit = false;
                            if (SPHERES && ((!MESH && !PLANES) || kind == RT_HITABLE_SPHERE)) {
                                // intersect_sphere + Sphere::hit, sphere.rs:4-17, 36-39
                                const float radius = e0.f[3];
                                const float qa = ECACHE ? ray_dd : dot(ed, ed);      // (ECACHE: ed is the world ray's direction for every sphere of the ray)
                                const float qb = dot(eo, ed);
                                const float qc = dot(eo, eo) - (ECACHE ? radius : radius * radius);      // (ECACHE: the cache holds the product)
                                const float discriminant = qb * qb - qa * qc;
                                if (discriminant > 0.0f) {
                                    const float sq = sqrtf(discriminant);
                                    // the far root only where the near one is out of range: qa = |d|^2 >= 0 and sq >= 0 give t0 <= t1 (division is monotonic), so an
                                    // accepted t0 IS min(t0, t1) whether or not t1 is in range (sphere.rs:7-16 evaluates both; same result, one division less)
                                    const float t0 = (-qb - sq) / qa;
                                    hit = t_min < t0 && t0 < t_max; t = t0;
                                    if (!hit) { const float t1 = (-qb + sq) / qa; hit = t_min < t1 && t1 < t_max; t = t1; }      // (weekend +0.5 %)
                                }
                            } else if (PLANES && (!MESH || kind == RT_HITABLE_PLANE)) {
                                // intersect_plane + Plane::hit, plane.rs:4-17, 59-70
                                const Word4* pw = S.planes[e1.u[1]].w;
                                const Word4 p0 = ldw(pw), p1 = ldw(pw + 1);
                                const F3 pn = w3(p0);
                                const float denominator = dot(ed, pn);
                                if (!(fabsf(denominator) < 1.0e-6f)) {
                                    t = dot(f3(0.0f, 0.0f, 0.0f) - eo, pn) / denominator;
                                    hit = !(t < t_min || t > t_max);
                                    if (hit && p1.u[3]) { const F3 p = eo + (ed * t); hit = !(length(p) > p0.f[3]); }
                                }
                            } else if (MESH) {
                                // MeshObject::hit -> BLAS (mesh.rs:161-176): enter the instance
                                PH(PH_INST_HEADER);
                                const Word4* mw = rec32(S.meshes, e1.u[1], (uint32_t)sizeof(DevMesh));
                                const Word4 m0 = ldw(mw), m1 = ldw(mw + 1), m2 = ldw(mw + 2);
                                const F3 einv = f3(1.0f / ed.x, 1.0f / ed.y, 1.0f / ed.z);
                                float e;
                                if (slab(m1.f[0], m1.f[1], m1.f[2], m1.f[3], m2.f[0], m2.f[1], eo, einv, t_min, best_t, e)) {
                                    if (k + 1 < count) {
                                        if (sp < stack_cap) stack[(sp++) * RT_BLOCK] = RT_REF_LEAF_BIT | ((count - k - 2) << RT_REF_COUNT_SHIFT) | (ei + 1);
                                        else atomicOr(P.error_flag, RT_DEVERR_STACK);
                                    }
                                    inst_sp = sp;
                                    o = eo; d = ed; inv = einv;
                                    tri_base = m0.u[1];
                                    cur_ent = ei;
                                    cur = m0.u[3];
                                    entered = true;
                                }
                            }
                            if (hit) {
                                if (best.ent == RT_NONE || t < best.t || (t == best.t && entity_rank(S, ei) < entity_rank(S, best.ent))) {
                                    best.t = t; best.ent = ei; best.tri = RT_NONE; best_t = fminf(t * RT_CULL_SLACK, t_max);
                                }
                            }
                        }
                        if (!entered) cur = pop();
                    }
                }
                // a shadow ray of a scene without dielectrics is decided by ANY occluder
                if (S.shadow_any_hit && state == SHADOW && best.ent != RT_NONE) cur = RT_NONE;
                if (cur == RT_NONE) tracing = false;      // closest hit (or miss) is in `best`
            }
            // leave the burst when enough lanes can take a new ray, or nobody is tracing
            const unsigned long long busy = __ballot(tracing);
#if RT_REFILL_LANES < 64
            const unsigned long long refill = __ballot(!tracing && state != FINISHED && state != NEED_BLOCK);
            if (busy == 0ull || (uint32_t)__popcll(refill) >= (uint32_t)RT_REFILL_LANES) break;
#else
            if (busy == 0ull) break;          // synchronous: the burst ends when every ray of the wave has ended
#endif
        }
#if defined(RT_PHASE_STATS) && !defined(RT_PHASE_TIME)
        if (lane == 0) {
            unsigned long long* b = P.phase_stats + 2 * PH_COUNT + 4 * bk_kind;
            atomicAdd(b, 1ull); atomicAdd(b + 1, (unsigned long long)bk_start); atomicAdd(b + 2, (unsigned long long)bk_iters); atomicAdd(b + 3, (unsigned long long)bk_lane_iters);
        }
#endif
    }

#ifdef RT_PHASE_TIME
    TM(TM_OTHER);
    for (int k = 0; k < TM_COUNT; k++) if (lane == 0) { atomicAdd(P.phase_stats + 2 * k, tm_acc[k]); atomicAdd(P.phase_stats + 2 * k + 1, 1ull); }
#elif defined(RT_PHASE_STATS)
    for (int k = 0; k < PH_COUNT; k++) {
        uint32_t a = ph_lanes[k], b = ph_waves[k];
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
        if (lane == 0) { atomicAdd(P.phase_stats + 2 * k, (unsigned long long)a); atomicAdd(P.phase_stats + 2 * k + 1, (unsigned long long)b); }
    }
#endif
    // ---- ray counters: wave reduce, one atomic per wave and class ------------------
    if (P.counters) {
        uint32_t v[4] = {n_primary, n_secondary, n_shadow, n_samples};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(P.counters + k, (unsigned long long)x);
        }
    }
}

// Folds segment sums into the accumulator in segment order:
// acc = (first ? 0 : acc) + seg0 + seg1 + ...   (sequential, implementation.rs:656).
// With one-sample units the segment sums ((0 + c0) + c1) + ... are formed here, with the very
// additions the render kernel performs for a whole-segment unit: both splits give the same bits.
__global__ void fold_kernel(const float* partial, uint32_t unit_count, uint32_t unit_first, uint32_t unit_len, uint32_t npix, float* acc,
                            uint32_t acc_pitch, uint32_t cw, uint32_t acc_x, uint32_t acc_y, int init_zero, uint32_t tiles_x, uint32_t pix_slots, const uint32_t* tile_bin) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint32_t x = i % cw, y = i / cw;
    const uint32_t slot = ((y >> 3) * tiles_x + (x >> 3)) * 64u + (y & 7u) * 8u + (x & 7u);      // the render kernel's tile layout
    float* a = acc + ((uint64_t)(acc_y + y) * acc_pitch + acc_x + x) * 3;
    float r = init_zero ? 0.0f : a[0], g = init_zero ? 0.0f : a[1], b = init_zero ? 0.0f : a[2];
    // an empty tile (classify_kernel) was never rendered: every one of its samples is the black sky's +0 and so is every segment sum, and x + 0 == x
    if (tile_bin && (tile_bin[slot >> 6] & 0xffu) == RT_BIN_EMPTY) { }
    else if (unit_len == RT_SAMPLE_SEGMENT) {             // units are whole segment sums
        for (uint32_t u = 0; u < unit_count; u++) {
            const float* p = partial + ((uint64_t)u * pix_slots + slot) * 3;
            r = r + p[0]; g = g + p[1]; b = b + p[2];
        }
    } else {                                         // units are single samples: form each segment sum first, in sample order
        float sr = 0.0f, sg = 0.0f, sb = 0.0f;
        for (uint32_t u = 0; u < unit_count; u++) {
            const float* p = partial + ((uint64_t)u * pix_slots + slot) * 3;
            sr = sr + p[0]; sg = sg + p[1]; sb = sb + p[2];
            if (((unit_first + u + 1u) % RT_SAMPLE_SEGMENT) == 0u || u + 1u == unit_count) {
                r = r + sr; g = g + sg; b = b + sb;
                sr = 0.0f; sg = 0.0f; sb = 0.0f;
            }
        }
    }
    a[0] = r; a[1] = g; a[2] = b;
}

__device__ __forceinline__ uint8_t f32_as_u8(float f) {      // Rust `as u8`: truncate, saturate, NaN -> 0
    if (!(f > 0.0f)) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}

// col / spp (implementation.rs:659) and v3_to_rgba (src/rgba.rs:6-12).
// Source: acc image (pitch src_pitch, origin sx,sy); destination images (pitch dst_pitch, origin dx,dy).
// strip_h != 0: the source rows are one rank's packed strips and go to their image rows (strip k of the rank = rows first_row + k * period_rows ...);
// col_n != 0: the source columns are one rank's packed tile columns (8 pixels wide) and go to image tile columns col_r, col_r + col_n, ...:
// the destination may then be the frame on ANOTHER device (rt_render_frame_multi writes each GPU's strips straight into GPU 0's frame over xGMI).
__global__ void finish_kernel(const float* acc, uint32_t src_pitch, uint32_t sx, uint32_t sy, uint32_t cw, uint32_t npix, float spp,
                              float* rgb, uint8_t* rgba, uint32_t dst_pitch, uint32_t dx, uint32_t dy,
                              uint32_t strip_h = 0u, uint32_t period_rows = 0u, uint32_t first_row = 0u, uint32_t col_n = 0u, uint32_t col_r = 0u) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint32_t x = i % cw, y = i / cw;
    const float* a = acc + ((uint64_t)(sy + y) * src_pitch + sx + x) * 3;
    const float r = a[0] / spp, g = a[1] / spp, b = a[2] / spp;
    const uint32_t row = strip_h ? first_row + (y / strip_h) * period_rows + (y % strip_h) : dy + y;
    const uint32_t col = col_n ? ((x >> 3) * col_n + col_r) * 8u + (x & 7u) : dx + x;      // col_n != 0: packed tile columns go to their image columns
    const uint64_t o = (uint64_t)row * dst_pitch + col;
    if (rgb) { rgb[o * 3] = r; rgb[o * 3 + 1] = g; rgb[o * 3 + 2] = b; }
    if (rgba) { rgba[o * 4] = f32_as_u8(255.0f * r); rgba[o * 4 + 1] = f32_as_u8(255.0f * g); rgba[o * 4 + 2] = f32_as_u8(255.0f * b); rgba[o * 4 + 3] = 255; }
}

// First hit_closest of the lens-centre ray of each pixel (parity gate 1).
// ---- hand-out order of the tiles ------------------------------------------------------------
// A launch ends when its slowest wave ends, and one 8x8 tile of glass at depth 8 costs ~100x a tile
// of sky.  Tiles are therefore handed out most expensive first, so that the waves still running
// at the end hold cheap tiles.  Cost class of a tile = what the primary rays through 16 of its
// pixel centres hit: 0 a dielectric (two child rays per hit), 1 something reflective, 2 anything
// else, 3 only sky.  The order changes the schedule only, never a pixel.
// Measured on weekend 1080p x 256 spp, per-rank time of the 8-rank strip partition:
// 7.5-8.0 ms row-major, 6.2-6.5 ms ordered (ideal 6.1); Dreadnaught on one GPU 19.97 -> 19.08 ms.
#define RT_COST_BINS 5u
#define RT_GROUP_STRIDE (RT_COST_BINS + 1u)      /* group_counts: tiles per class, then the valid pixels of the group's empty tiles */
#define RT_ORDER_GROUP 1024u      /* tiles per block of order_kernel; classify_kernel counts the classes per such group */

// Empty tiles: a packet test of one 8 x 8 tile against the scene.  A tile is EMPTY when every camera ray of its pixels - every lens sample of every sample index - misses, by a
// margin far above the rounding of the per-ray tests, the box of every leaf of the trees: the reference reports a candidate only for leaves whose box (and whose ancestors'
// boxes) the ray passes (bvh.rs:210-236 with implementation.rs:92-118), so such a ray has no candidate, hit_closest returns None and the sample is the sky's colour.  Under a
// black sky (color_sky_black, implementation.rs:517-519) that is (0, 0, 0) whatever the lens draw, so the tile needs no ray at all; its camera rays are still counted
// (a ray = one call of Scene::hit_closest, which the reference makes).
// Geometry (Camera::get_ray, implementation.rs:482-491): a ray leaves o_s = origin + lens offset, |o_s - origin| <= r_l, and passes the focal-plane point T of its pixel at t = 1,
// |T - Tc| <= r_t for the tile's centre point Tc; so the ray's point at parameter t lies within rho(t) = |1 - t| r_l + t r_t of C(t) = origin + t (Tc - origin), in every axis:
// the BEAM.  It can meet a box only if for some t >= 0 C(t) is inside the box grown by rho(t) + margin; rho is linear on [0, 1] and on [1, inf), so each piece is a slab test
// whose planes move linearly with t - six half-line constraints p t >= q per piece.  Infinite or NaN boxes (planes) give constraints that never bind: never empty.
// The beam walks the trees the render kernel walks (any box it misses prunes everything below: inner boxes contain their leaves' boxes); into a mesh instance it goes as
// Entity::hit takes a ray there (ray_to_entity on the central ray: translations and unit-axis rotations keep distances, so the radii stand; scenes with other axes -
// loose_boxes - are not culled), and the first leaf box it meets - a sphere's or plane's entity box, a triangle leaf - ends the walk: not empty.
struct Beam {
    float a[3], b[3];                 // C(t) = a + t b
    float r_l, r_t, scale, bmax;      // scale: the largest coordinate in play (the margin is relative to it)
    float p[2][3][2], ip[2][3][2];    // per piece and axis: the slopes of the two constraints p t >= q and their reciprocals - the same for every box the beam is tested against
};

__device__ __forceinline__ bool beam_meets_box(const Beam& B, const float lo[3], const float hi[3]) {
    float scale = B.scale;
    for (int i = 0; i < 3; i++) scale = fmaxf(scale, fmaxf(fabsf(lo[i]), fabsf(hi[i])));
    const float margin = 2.5e-4f * scale;          // 10^3 x the relative rounding of the per-ray slab test and of the ray transform
    for (int piece = 0; piece < 2; piece++) {
        // rho'(t) = alpha + beta t >= rho(t) + margin on the piece (beta: beam_set_ray)
        const float alpha = (piece == 0 ? B.r_l : -B.r_l) + margin;
        float t_lo = piece == 0 ? 0.0f : 1.0f, t_hi = piece == 0 ? 1.0f : RT_F32_MAX;
        bool none = false;
        for (int i = 0; i < 3; i++) {
            // inside the grown slab i:  a + t b + rho' >= lo  and  a + t b - rho' <= hi
            const float q[2] = {lo[i] - B.a[i] - alpha, B.a[i] - hi[i] - alpha};
            for (int c = 0; c < 2; c++) {
                const float p = B.p[piece][i][c], r = q[c] * B.ip[piece][i][c];      // (q / p to 1.5 ulp: the slack below is a hundred times that)
                if (p > 0.0f) t_lo = fmaxf(t_lo, r - fabsf(r) * 1.0e-5f);
                else if (p < 0.0f) t_hi = fminf(t_hi, r + fabsf(r) * 1.0e-5f);
                else if (q[c] > 0.0f) none = true;
            }
        }
        if (!(none || t_lo > t_hi)) return true;      // (a NaN anywhere: not provably missed)
    }
    return false;
}

__device__ __forceinline__ void beam_set_ray(Beam& B, F3 a, F3 b) {
    B.a[0] = a.x; B.a[1] = a.y; B.a[2] = a.z; B.b[0] = b.x; B.b[1] = b.y; B.b[2] = b.z;
    B.bmax = fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fabsf(b.z));
    for (int piece = 0; piece < 2; piece++) {
        // rho'(t) = alpha + beta t: beta of beam_meets_box (the slope's own margin covers what grows with t: the rounding of C(t), 1e-7 |b| t)
        const float beta = piece == 0 ? (B.r_t - B.r_l) : (B.r_l + B.r_t) * 1.001f + 1.0e-5f * B.bmax;
        for (int i = 0; i < 3; i++) {
            B.p[piece][i][0] = B.b[i] + beta; B.p[piece][i][1] = beta - B.b[i];
            B.ip[piece][i][0] = 1.0f / B.p[piece][i][0]; B.ip[piece][i][1] = 1.0f / B.p[piece][i][1];      // (a zero slope never reads its reciprocal)
        }
    }
    B.scale = fmaxf(B.scale, fmaxf(fmaxf(fabsf(a.x) + fabsf(b.x), fabsf(a.y) + fabsf(b.y)), fabsf(a.z) + fabsf(b.z)));      // (only ever grows: the margin of the world beam stands inside an instance)
}

// px0, py0: the tile's first pixel in the image.  `stack`: this lane's LDS column, as in trace().
__device__ bool tile_is_empty(const DevSceneView& S, const RenderParams& P, uint32_t px0, uint32_t py0, uint32_t* stack, const uint32_t stack_cap) {
    const F3 org = ld3(S.cam_origin), hor = ld3(S.cam_horizontal), ver = ld3(S.cam_vertical);
    const float uc = ((float)px0 + 3.5f) / (float)P.width, vc = ((float)P.height - ((float)py0 + 3.5f)) / (float)P.height;      // pixel_uv of the tile's centre
    const F3 Tc = ld3(S.cam_llc) + (hor * uc) + (ver * vc);
    Beam W;
    W.r_t = 4.0f * (length(hor) / (float)P.width + length(ver) / (float)P.height);      // (3.5 pixels each way: 4 leaves slack)
    W.r_l = S.cam_lens_radius * (length(ld3(S.cam_u)) + length(ld3(S.cam_v))) * 1.001f;   // lens offsets are in [-R, R) along cam_u and cam_v
    W.scale = W.r_l + W.r_t;
    beam_set_ray(W, org, Tc - org);
    if (!beam_meets_box(W, S.tlas_min, S.tlas_max)) return true;
    Beam B = W;
    uint32_t sp = 0, inst_sp = RT_NONE, cur = S.tlas_root_ref;
    auto pop = [&]() -> uint32_t {
        if (sp == inst_sp) { inst_sp = RT_NONE; B = W; }
        if (sp == 0) return RT_NONE;
        return stack[(--sp) * RT_BLOCK];
    };
    for (;;) {
        while (cur < RT_REF_LEAF_BIT) {
            const Word4* n = S.nodes[cur].w;
            const Word4 a = ldw(n), b = ldw(n + 1), c = ldw(n + 2), r = ldw(n + 3);
            const float llo[3] = {a.f[0], a.f[2], b.f[0]}, lhi[3] = {b.f[2], c.f[0], c.f[2]}, rlo[3] = {a.f[1], a.f[3], b.f[1]}, rhi[3] = {b.f[3], c.f[1], c.f[3]};      // (slab2's layout)
            const bool ml = beam_meets_box(B, llo, lhi), mr = beam_meets_box(B, rlo, rhi);
            if (ml && mr) {
                if (sp >= stack_cap) return false;      // (never expected: the stack is sized from the trees; then simply not culled)
                stack[(sp++) * RT_BLOCK] = r.u[1];
                cur = r.u[0];
            } else if (ml || mr) cur = ml ? r.u[0] : r.u[1];
            else cur = pop();
        }
        if (cur == RT_NONE) return true;                // every subtree pruned
        if (inst_sp != RT_NONE) return false;           // a triangle leaf whose box the beam meets
        const uint32_t first = cur & RT_REF_FIRST_MASK, count = ((cur >> RT_REF_COUNT_SHIFT) & 7u) + 1u;
        bool entered = false;
        for (uint32_t k = 0; k < count && !entered; k++) {
            const uint32_t ei = first + k;
            const Word4 bl = ldw(S.ent_box + 2 * ei), bh = ldw(S.ent_box + 2 * ei + 1);
            const float elo[3] = {bl.f[0], bl.f[1], bl.f[2]}, ehi[3] = {bh.f[0], bh.f[1], bh.f[2]};
            if (!beam_meets_box(W, elo, ehi)) continue;      // the entity's own world box (Entity::calculate_origin_aabb)
            const Word4* ew = S.entities[ei].w;
            const Word4 e0 = ldw(ew), e1 = ldw(ew + 1);
            if ((e1.u[0] & 0xff) != RT_HITABLE_MESH) return false;      // a sphere or a plane in the beam's way
            F3 eo = org, ed = Tc - org;
            ray_to_entity(S, e0, e1, eo, ed);
            Beam E = W;
            beam_set_ray(E, eo, ed);
            const Word4* mw = S.meshes[e1.u[1]].w;
            const Word4 m0 = ldw(mw), m1 = ldw(mw + 1), m2 = ldw(mw + 2);
            const float mlo[3] = {m1.f[0], m1.f[1], m1.f[2]}, mhi[3] = {m1.f[3], m2.f[0], m2.f[1]};
            if (!beam_meets_box(E, mlo, mhi)) continue;
            if (m0.u[3] >= RT_REF_LEAF_BIT) return false;    // a mesh of one leaf (its box is the root box)
            if (k + 1 < count) {
                if (sp >= stack_cap) return false;
                stack[(sp++) * RT_BLOCK] = RT_REF_LEAF_BIT | ((count - k - 2) << RT_REF_COUNT_SHIFT) | (ei + 1);
            }
            inst_sp = sp; B = E; cur = m0.u[3]; entered = true;
        }
        if (!entered) cur = pop();
    }
}

// one thread per tile: RT_BIN_EMPTY (black sky: nothing to render), 0xff | RT_BIN_NOHIT (no camera ray of the tile can hit: sky-only, and the render kernel need not walk), or 0xff = to be classified
__global__ void __launch_bounds__(RT_BLOCK)
empty_tiles_kernel(const DevSceneView S, const RenderParams P, uint32_t n_tiles, uint32_t* tile_bin) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t tx0 = (tile % P.tiles_x) * 8u, ty0 = (tile / P.tiles_x) * 8u;
    const uint32_t px0 = P.col_n ? ((tx0 >> 3) * P.col_n + P.col_r) * 8u : P.left + tx0;
    const uint32_t py0 = P.strip_h ? P.strip_first_row + (ty0 / P.strip_h) * P.strip_period_rows + (ty0 % P.strip_h) : P.top + ty0;
    const bool empty = tile_is_empty(S, P, px0, py0, lds_stack + threadIdx.x, S.stack_entries);
    tile_bin[tile] = !empty ? 0xffu : (P.cull_empty == 2u ? RT_BIN_EMPTY : (0xffu | RT_BIN_NOHIT));
}

__global__ void __launch_bounds__(RT_BLOCK)
classify_kernel(const DevSceneView S, const RenderParams P, uint32_t n_tiles, uint32_t* tile_bin, uint32_t* group_counts) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stack = lds_stack + threadIdx.x;
    // one wave = 4 tiles x 16 pixels (a staggered 4x4 subsample of each 8x8 tile)
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tile = gid >> 4, k = gid & 15u;
    const bool live = tile < n_tiles;
    const uint32_t tl = live ? tile : 0u;
    const uint32_t x = (tl % P.tiles_x) * 8u + 2u * (k & 3u) + ((k >> 2) & 1u), y = (tl / P.tiles_x) * 8u + 2u * (k >> 2) + (k & 1u);
    uint32_t cls = 3u;
    float coc = 0.0f, hits = 0.0f;
    const uint32_t px = P.col_n ? ((x >> 3) * P.col_n + P.col_r) * 8u + (x & 7u) : P.left + x;
    const uint32_t py = P.strip_h ? P.strip_first_row + (y / P.strip_h) * P.strip_period_rows + (y % P.strip_h) : P.top + y;
    // empty tiles: found by empty_tiles_kernel (all 16 lanes of a tile read its answer before lane 0 overwrites it below)
    const uint32_t tx0 = (tl % P.tiles_x) * 8u, ty0 = (tl / P.tiles_x) * 8u;
    const uint32_t pre = (P.cull_empty && live) ? tile_bin[tl] : 0xffu;
    const bool empty = pre == RT_BIN_EMPTY, nohit = (pre & RT_BIN_NOHIT) != 0u;
    if (empty) cls = RT_BIN_EMPTY;
    else if (nohit) { }      // (sky only, by the packet test: no ray to trace here either)
    else if (P.classify_by_beam) cls = live ? S.uniform_cost_class : 3u;      // (whatever a ray of this tile hits, its class is the scene's; the few tiles the beam could not clear without a hit in them are handed out early instead of late)
    else if (live && x < P.cw && y < P.ch) {
        float pu, pv; F3 o, d;
        pixel_uv(P, px, py, pu, pv);
        gen_camera_ray(S, pu, pv, 0.0f, 0.0f, o, d);
        Hit h;
        trace(S, o, d, RT_BIAS, RT_F32_MAX, stack, S.stack_entries, P.error_flag, h);
        if (h.ent != RT_NONE) {
            const Word4 m0 = ldw(S.materials[ldw(S.entities[h.ent].w + 1).u[3]].w);
            cls = m0.u[0] == RT_MAT_DIELECTRIC ? 0u : ((m0.u[0] == RT_MAT_SPECULAR || m0.f[1] > 0.0f) ? 1u : 2u);
            // circle of confusion at the hit, in pixels: camera rays meet the focal plane at t = 1 (Camera::get_ray,
            // :482-491), rays of one pixel are 2 R |1 - t| apart at t, a pixel is t |horizontal| / width wide there
            coc = 2.0f * S.cam_lens_radius * fabsf(1.0f - h.t) * (float)P.width / (h.t * length(ld3(S.cam_horizontal)));
            hits = 1.0f;
        }
    }
    for (int off = 8; off > 0; off >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)cls, off, 64); cls = other < cls ? other : cls;      // (the 16 lanes of an empty tile all say so)
        coc += __shfl_xor(coc, off, 64); hits += __shfl_xor(hits, off, 64);
    }
    // Round shape of the tile (render_kernel): where the rays of a pixel spread over more than P.lens_coc_px pixels,
    // rounds of one sample index (one lens cell) of 8 neighbouring pixels are more coherent than rounds of one pixel.
    const uint32_t by_lens = (hits > 0.0f && coc > P.lens_coc_px * hits) ? 0x100u : 0u;
    if (live && k == 0u) tile_bin[tile] = cls | by_lens | (nohit ? RT_BIN_NOHIT : 0u);
    // class counts of the tile's group, for order_kernel: the four tiles of a wave lie in one group (RT_ORDER_GROUP is a multiple of 4) and are usually of one class,
    // so one add per wave and class present (one add per TILE, 1 024 to the same address, made this kernel 100 us slower)
#pragma unroll
    for (uint32_t c = 0; c < RT_COST_BINS; c++) {
        const unsigned long long m = __ballot(live && k == 0u && cls == c);
        if (m != 0ull && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m) - 1))
            atomicAdd(group_counts + (tile / RT_ORDER_GROUP) * RT_GROUP_STRIDE + c, (uint32_t)__popcll(m));
    }
    // the pixels of the chunk inside the group's empty tiles (their camera rays are counted by order_kernel): one add per wave
    if (P.cull_empty) {
        uint32_t npx = (live && k == 0u && (empty || nohit)) ? (P.cw - tx0 < 8u ? P.cw - tx0 : 8u) * (P.ch - ty0 < 8u ? P.ch - ty0 : 8u) : 0u;
        npx += (uint32_t)__shfl_xor((int)npx, 16, 64); npx += (uint32_t)__shfl_xor((int)npx, 32, 64);
        if ((threadIdx.x & 63u) == 0u && npx) atomicAdd(group_counts + (tile / RT_ORDER_GROUP) * RT_GROUP_STRIDE + RT_COST_BINS, npx);
    }
}

// Stable counting sort of the tiles by class (row-major order kept inside a class, so that waves running at the same time stay on
// neighbouring tiles), in parallel: block g owns the tiles [g, g + 1) x RT_ORDER_GROUP.  classify_kernel has counted every group's tiles per
// class (group_counts, zeroed per call), so a block finds where its tiles of each class start from the counts of the groups before it, and a
// tile's place among its block's from a ballot per wave and the counts of the waves before it.  (Round 2's version was one block whose scan
// over 1024 partial counts ran on four threads: 65 us per launch whatever the share of the frame - a fixed cost that an 8-rank split of a
// 10 ms frame pays in full.  This one: a few us.)
__global__ void __launch_bounds__(RT_ORDER_GROUP)
order_kernel(const uint32_t* tile_bin, uint32_t n_tiles, const uint32_t* group_counts, uint32_t* tile_order, unsigned long long* counters, uint32_t sample_count, uint32_t* culled_word,
             uint32_t cull_mode) {
    __shared__ uint32_t wave_count[RT_ORDER_GROUP / 64u][RT_COST_BINS];
    __shared__ uint32_t group_base[RT_COST_BINS];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, g = blockIdx.x, n_groups = gridDim.x;
    const uint32_t i = g * RT_ORDER_GROUP + t;
    const bool live = i < n_tiles;
    const uint32_t bin = live ? tile_bin[i] : 0u, cls = bin & 0xffu;
    // where this block's tiles of each class start: the class's base (all tiles of the classes before it) + the groups before this one
    if (wave == 0u) {
        uint32_t total[RT_COST_BINS], before[RT_COST_BINS];
#pragma unroll
        for (uint32_t b = 0; b < RT_COST_BINS; b++) {
            uint32_t a = 0, c = 0;
            for (uint32_t k = lane; k < n_groups; k += 64u) { const uint32_t v = group_counts[k * RT_GROUP_STRIDE + b]; a += v; c += k < g ? v : 0u; }
            for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); c += __shfl_xor(c, off, 64); }
            total[b] = a; before[b] = c;
        }
        if (lane == 0u) {
            uint32_t run = 0;
#pragma unroll
            for (uint32_t b = 0; b < RT_COST_BINS; b++) { group_base[b] = run + before[b]; run += total[b]; }
            if (g == 0u) {
                tile_order[-1] = total[0] + total[1] + total[2];      // every tile that hits something: all classes but "sky only" and "empty"
                tile_order[-2] = total[0] + total[1] + total[2] + total[3];      // every tile that is handed out: all but the empty ones
            }
        }
        if (g == 0u) {
            // empty tiles: their samples are never rendered, but their camera rays were cast as far as the reference is concerned (hit_closest -> None at the root)
            uint32_t px = 0;
            for (uint32_t k = lane; k < n_groups; k += 64u) px += group_counts[k * RT_GROUP_STRIDE + RT_COST_BINS];
            for (int off = 32; off > 0; off >>= 1) px += __shfl_xor(px, off, 64);
            if (lane == 0u) {
                if (cull_mode == 2u && counters && px) {      // (under another sky the tiles are rendered and count their own rays)
                    atomicAdd(counters + 0, (unsigned long long)px * sample_count);      // primary
                    atomicAdd(counters + 3, (unsigned long long)px * sample_count);      // samples
                }
                if (culled_word && cull_mode) __hip_atomic_store(culled_word + (cull_mode == 2u ? 0 : 1), px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t b = 0; b < RT_COST_BINS; b++) {
        const unsigned long long m = __ballot(live && cls == b);
        if (cls == b) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_count[wave][b] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (!live) return;
    uint32_t pos = group_base[cls] + rank;
    for (uint32_t w = 0; w < wave; w++) pos += wave_count[w][cls];
    tile_order[pos] = i | (((bin >> 8) & 1u) << 31) | (((bin >> 9) & 1u) << 30);      // bit 31: rounds by lens cell; bit 30: no camera ray of the tile can hit anything
}

__global__ void __launch_bounds__(RT_BLOCK)
primary_kernel(const DevSceneView S, const RenderParams P, float* t_out, float* n_out, int32_t* e_out, int32_t* tri_out) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stack = lds_stack + threadIdx.x;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.cw * P.ch) return;
    const uint32_t x = i % P.cw, y = i / P.cw;
    F3 o, d;
    float pu, pv;
    pixel_uv(P, P.left + x, P.top + y, pu, pv);
    gen_camera_ray(S, pu, pv, 0.0f, 0.0f, o, d);
    Hit h;
    trace(S, o, d, RT_BIAS, RT_F32_MAX, stack, S.stack_entries, P.error_flag, h);
    float t = -1.0f; F3 n = f3(0, 0, 0); int32_t e = -1, tri = -1;
    if (h.ent != RT_NONE) { HitRec hr; resolve_hit(S, o, d, h, hr); t = h.t; n = hr.normal; e = hr.entity_index; tri = hr.tri_index; }
    if (t_out) t_out[i] = t;
    if (n_out) { n_out[i * 3] = n.x; n_out[i * 3 + 1] = n.y; n_out[i * 3 + 2] = n.z; }
    if (e_out) e_out[i] = e;
    if (tri_out) tri_out[i] = tri;
}

#ifdef RT_BVH4_EXPERIMENT
// diagnostics build only: primary hits through the 4-wide walk (rt_debug_bvh4_primary)
__global__ void __launch_bounds__(RT_BLOCK)
primary_wide_kernel(const DevSceneView S, const RenderParams P, const Word4* wide, uint32_t stack_cap, float* t_out, int32_t* e_out, int32_t* tri_out) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t* stack = lds_stack + threadIdx.x;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.cw * P.ch) return;
    const uint32_t x = i % P.cw, y = i / P.cw;
    F3 o, d;
    float pu, pv;
    pixel_uv(P, P.left + x, P.top + y, pu, pv);
    gen_camera_ray(S, pu, pv, 0.0f, 0.0f, o, d);
    Hit h;
    if (wide) trace<true>(S, o, d, RT_BIAS, RT_F32_MAX, stack, stack_cap, P.error_flag, h, wide);
    else trace<false>(S, o, d, RT_BIAS, RT_F32_MAX, stack, stack_cap, P.error_flag, h);
    t_out[i] = h.ent != RT_NONE ? h.t : -1.0f;
    e_out[i] = (int32_t)h.ent; tri_out[i] = (int32_t)h.tri;
}
#endif

// ===========================================================================
// refit of the TLAS after a change of the entity transforms (rt_scene_update_transforms)
// ===========================================================================
// Entity::calculate_origin_aabb (implementation.rs:258-283) for every entity, with the arithmetic of the host build
// (scene_build.cpp: rotate() = types.rs:200 with cos / sin evaluated by the host libm and handed over in the transform
// records), so the boxes are bit for bit those a fresh rt_scene_create would compute.  Also refreshes the entity's inline
// first translation.  ent_box: 8 floats per TLAS position (lo.xyz, -, hi.xyz, -).
// ent_out may be another array than ent_in (rt_scene_rebuild_tlas stages the moved records before it commits): the whole record is then copied.
__global__ void refit_entities_kernel(const DevEntity* ent_in, DevEntity* ent_out, const Word4* xforms, const DevMesh* meshes, float* ent_box, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Word4* ew = ent_out[i].w;
    const Word4 e0 = ldw(ent_in[i].w), e1 = ldw(ent_in[i].w + 1);
    if (ent_in != ent_out) {
        const Word4 e2 = ldw(ent_in[i].w + 2);
        *reinterpret_cast<float4*>(ew) = *reinterpret_cast<const float4*>(&e0); *reinterpret_cast<float4*>(ew + 1) = *reinterpret_cast<const float4*>(&e1);
        *reinterpret_cast<float4*>(ew + 2) = *reinterpret_cast<const float4*>(&e2);
    }
    const uint32_t kind = e1.u[0] & 0xff, n_trans = (e1.u[0] >> 8) & 0xff, n_rot = (e1.u[0] >> 16) & 0xff;
    const Word4* x = xforms + e1.u[2];
    const float inf = __builtin_inff();
    F3 lo, hi;
    if (kind == RT_HITABLE_SPHERE) { const float r = e0.f[3]; lo = f3(0.0f - r, 0.0f - r, 0.0f - r); hi = f3(0.0f + r, 0.0f + r, 0.0f + r); }      // sphere.rs:55-58
    else if (kind == RT_HITABLE_PLANE) { lo = f3(-inf, -inf, -inf); hi = f3(inf, inf, inf); }                                                   // plane.rs:92-95
    else { const Word4* mw = meshes[e1.u[1]].w; const Word4 m1 = ldw(mw + 1), m2 = ldw(mw + 2); lo = f3(m1.f[0], m1.f[1], m1.f[2]); hi = f3(m1.f[3], m2.f[0], m2.f[1]); }
    // implementation.rs:264 (`!aabb.is_infinite()`): rotations are skipped for ANY hitable whose box is infinite on all six sides - a plane, but also a
    // sphere of infinite radius or a mesh with infinite bounds (scene_build.cpp: box_is_infinite) - not by the kind of hitable
    const bool infinite = lo.x == -inf && lo.y == -inf && lo.z == -inf && hi.x == inf && hi.y == inf && hi.z == inf;
    // AABB::corners, implementation.rs:120-131
    F3 c[8] = {lo, f3(lo.x, lo.y, hi.z), f3(lo.x, hi.y, lo.z), f3(hi.x, lo.y, lo.z), hi, f3(hi.x, hi.y, lo.z), f3(hi.x, lo.y, hi.z), f3(lo.x, hi.y, hi.z)};
    if (!infinite)
        for (uint32_t k = n_rot; k-- > 0;) {
            const Word4 a = ldw(x + n_trans + 2 * k), b = ldw(x + n_trans + 2 * k + 1);
#pragma unroll
            for (int q = 0; q < 8; q++) c[q] = rotate_cs(c[q], w3(a), b.f[1], b.f[2]);      // + theta
        }
    for (uint32_t k = 0; k < n_trans; k++) {
        const F3 t = w3(ldw(x + k));
#pragma unroll
        for (int q = 0; q < 8; q++) c[q] = c[q] + t;
    }
    F3 bl = c[0], bh = c[0];
#pragma unroll
    for (int q = 1; q < 8; q++) {
        bl = f3(fminf(bl.x, c[q].x), fminf(bl.y, c[q].y), fminf(bl.z, c[q].z));
        bh = f3(fmaxf(bh.x, c[q].x), fmaxf(bh.y, c[q].y), fmaxf(bh.z, c[q].z));
    }
    float* o = ent_box + (size_t)i * 8;
    o[0] = bl.x; o[1] = bl.y; o[2] = bl.z; o[3] = 0.0f; o[4] = bh.x; o[5] = bh.y; o[6] = bh.z; o[7] = 0.0f;
    if (n_trans > 0) { const Word4 t0 = ldw(x); float* w0 = ew[0].f; w0[0] = t0.f[0]; w0[1] = t0.f[1]; w0[2] = t0.f[2]; }      // (w0.w keeps the radius)
}

// Box of one child of a TLAS record: the union of a range of entity boxes (count > 0) or of the two boxes of the child's own
// record (count == 0, written by the previous, deeper level).  Unions are exact (min / max), like the host build's.
__device__ __forceinline__ void refit_child_box(const DevNode* nodes, const float* ent_box, uint32_t first, uint32_t count, uint32_t child, F3& lo, F3& hi) {
    if (count > 0) {
        const float* b = ent_box + (size_t)first * 8;
        lo = f3(b[0], b[1], b[2]); hi = lo;                                                   // bvh.rs:109-116 folds [min, max] of each object in order
        for (uint32_t k = 0; k < count; k++) {
            const float* q = ent_box + (size_t)(first + k) * 8;
            lo = f3(fminf(fminf(lo.x, q[0]), q[4]), fminf(fminf(lo.y, q[1]), q[5]), fminf(fminf(lo.z, q[2]), q[6]));
            hi = f3(fmaxf(fmaxf(hi.x, q[0]), q[4]), fmaxf(fmaxf(hi.y, q[1]), q[5]), fmaxf(fmaxf(hi.z, q[2]), q[6]));
        }
    } else {
        const Word4* n = nodes[child].w;
        const Word4 a = ldw(n), b = ldw(n + 1), c = ldw(n + 2);
        lo = f3(fminf(a.f[0], a.f[1]), fminf(a.f[2], a.f[3]), fminf(b.f[0], b.f[1]));
        hi = f3(fmaxf(b.f[2], b.f[3]), fmaxf(c.f[0], c.f[1]), fmaxf(c.f[2], c.f[3]));
    }
}

__global__ void refit_level_kernel(DevNode* nodes, const RefitRec* recs, uint32_t first, uint32_t count, const float* ent_box) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const RefitRec r = recs[first + i];
    F3 ll, lh, rl, rh;
    refit_child_box(nodes, ent_box, r.l_first, r.l_count, r.l_child, ll, lh);
    refit_child_box(nodes, ent_box, r.r_first, r.r_count, r.r_child, rl, rh);
    float* w = nodes[r.node].w[0].f;      // layout: device_scene.h
    w[0] = ll.x; w[1] = rl.x; w[2] = ll.y; w[3] = rl.y;
    w[4] = ll.z; w[5] = rl.z; w[6] = lh.x; w[7] = rh.x;
    w[8] = lh.y; w[9] = rh.y; w[10] = lh.z; w[11] = rh.z;
}

// The TLAS root box (DevSceneView::tlas_min / tlas_max): the root record's two boxes, or - a scene of one leaf - its entities.
__global__ void refit_root_kernel(const DevNode* nodes, uint32_t root_ref, uint32_t n_entities, const float* ent_box, float* root_box) {
    F3 lo, hi;
    if (root_ref & RT_REF_LEAF_BIT) refit_child_box(nodes, ent_box, 0u, n_entities, 0u, lo, hi);
    else refit_child_box(nodes, ent_box, 0u, 0u, root_ref, lo, hi);
    root_box[0] = lo.x; root_box[1] = lo.y; root_box[2] = lo.z; root_box[3] = hi.x; root_box[4] = hi.y; root_box[5] = hi.z;
}

#include "tlas_build.h"      // device-side BUILD of the TLAS (rt_scene_rebuild_tlas)

// ===========================================================================
// host side
// ===========================================================================
thread_local std::string g_error;

int set_error(int code, const std::string& msg) { g_error = msg; return code; }

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) return set_error(RT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// After work or copies were enqueued on `stream`, a failing call must not return while they can still
// write the caller's buffers or this function's locals: drain the stream first.
#define HIP_TRY_SYNC(expr, stream)                                                                         \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) { (void)hipStreamSynchronize(stream); return set_error(RT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } \
    } while (0)

struct DeviceBuffer {
    void* p = nullptr; size_t bytes = 0;
    bool fixed = false;      // a slice of the allocation-free entry point's static scratch: never grown, never freed
    int ensure(size_t n) {
        if (n <= bytes) return RT_OK;
        if (fixed) return set_error(RT_E_LIMIT, "rt_render_chunk_device_noalloc: the call needs more scratch than the library's static array holds");
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        HIP_TRY(hipMalloc(&p, n));
        bytes = n; return RT_OK;
    }
    void release() { if (p && !fixed) (void)hipFree(p); p = nullptr; bytes = 0; }
};

}  // namespace

// ---- scratch of the allocation-free entry point (rt_render_chunk_device_noalloc) ---------------------------------
// One static array in device memory per GPU the library runs on (held until the process ends): the segment sums, the
// chunk accumulator, the block counters and the tile order of ONE call are carved out of it.  `scratch` is a host-side
// stand-in, not a device address: host code only ever uses the per-device address hipGetSymbolAddress returned
// (scratch_address(), taken once per device in rt_scene_create with that device current).
#define RT_NOALLOC_SCRATCH_BYTES (32u << 20)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

#define RT_MAX_DEVICES 64
static std::mutex g_scratch_lock;
static unsigned char* g_scratch_ptr[RT_MAX_DEVICES] = {};      // [device]: that GPU's copy of the array
static std::mutex g_scratch_call_lock[RT_MAX_DEVICES];          // serialises the ENQUEUEING of no-alloc calls per device

// Address of `device`'s copy of the scratch array; the first call per device (with that device current) asks HIP for it.
static int scratch_address(int device, unsigned char** out) {
    if (device < 0 || device >= RT_MAX_DEVICES) return set_error(RT_E_LIMIT, "device index beyond RT_MAX_DEVICES");
    std::lock_guard<std::mutex> g(g_scratch_lock);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return set_error(RT_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = (unsigned char*)ptr;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

// Everything one in-flight render call writes besides the caller's outputs.
struct Workspace {
    hipStream_t stream = nullptr;    // own stream (host-pointer entry points only)
    bool no_alloc = false;           // buffers are fixed slices of the static scratch: nothing may be allocated on their behalf
    DeviceBuffer partial, accum, out_rgb, out_rgba, ctrl, prim, order;
    int init(bool with_stream) {
        if (with_stream) HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return ctrl.ensure(1024 + RT_QUEUES * 64);      // [0, 32) ray counters, [1024, ...) block counters
    }
    void release() {
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); stream = nullptr; }
        partial.release(); accum.release(); out_rgb.release(); out_rgba.release(); ctrl.release(); prim.release(); order.release();
    }
};

// rt_render_frame_multi: per-replica streams, events and buffers (owned by the scene passed first)
struct MultiContext {
    std::vector<RtScene*> scenes;
    struct PerDevice {
        int device = 0;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        bool direct = false;                  // this device can write GPU 0's memory (the same device, or peer access enabled)
        DeviceBuffer counters, packed_rgb, packed_rgba, accum;
        DeviceBuffer staging, staging_rgba;   // on GPU 0: this device's accumulator (spp mode) / packed tile columns (tile mode without peer access), copied over
        RtRayCounters host_counters{};
        int rc = RT_OK; std::string error;
    };
    std::vector<PerDevice> dev;
    DeviceBuffer total;                       // on GPU 0 (spp mode)
    hipStream_t stream0 = nullptr;
    void release() {
        for (auto& d : dev) {
            (void)hipSetDevice(d.device);
            if (d.stream) { (void)hipStreamSynchronize(d.stream); (void)hipStreamDestroy(d.stream); }
            if (d.done) (void)hipEventDestroy(d.done);
            d.counters.release(); d.packed_rgb.release(); d.packed_rgba.release(); d.accum.release();
        }
        if (!dev.empty()) { (void)hipSetDevice(dev[0].device); for (auto& d : dev) { d.staging.release(); d.staging_rgba.release(); } total.release(); }
        dev.clear(); scenes.clear();
    }
};


#define RT_MAX_WORKSPACES 8

// Switches of the schedule and the sampler (rt_scene_set_option).  Their defaults can be preset through the
// environment, which is read ONCE, in rt_scene_create: RT_LENS_STRATIFY, RT_FINE_SPLIT, RT_ROUND_UNITS, RT_LENS_COC.
struct SceneOptions {
    int lens_stratify = 1;          // RT_OPT_LENS_SAMPLER
    int fine_split = -1;            // RT_OPT_FINE_SPLIT: -1 automatic, 0 whole segments, 1 single samples
    int round_units = 8;            // RT_OPT_ROUND_UNITS
    float lens_coc_px = 3.0f;       // RT_OPT_LENS_COC_PX (with 8 x 8 lens cells: mesh_demo 4K x 1024 spp 16 871 / 16 819 / 16 732 / 16 484 / 16 000 Mrays/s at 2 / 3 / 4 / 6 / 8 px, weekend within 0.6 %)
    int kind_order = 0;             // RT_KIND_ORDER=1 in the RT_KIND_ORDER_EXPERIMENT build
    int cull_empty_tiles = 1;       // RT_OPT_CULL_EMPTY_TILES: tiles whose camera rays all miss the root box under a black sky are not traced (same bits, same counters)
    uint32_t debug_stack_entries = 0;   // RT_OPT_DEBUG_STACK_ENTRIES: 0 = what the trees need
    uint32_t debug_rebuild_stack_limit = 0;   // RT_OPT_DEBUG_REBUILD_STACK_LIMIT: rt_scene_rebuild_tlas refuses trees that need more stack entries (0 = the LDS limit)
};

struct RtScene {
    int device = 0;
    std::mutex opt_lock;
    SceneOptions opt;               // every call works on a snapshot taken under opt_lock
    uint32_t* dev_error_host = nullptr;   // host-mapped word the kernels set on a capacity overflow (RT_DEVERR_*): readable without a sync
    uint32_t* dev_error_dev = nullptr;
    // the same host-mapped allocation, 64 bytes apart: [16] the cancel word (rt_scene_cancel; read by the render kernel at its block claims),
    // [32] the progress word (written by the render kernel); the totals live on the host
    volatile uint32_t* cancel_host() const { return dev_error_host + 16; }
    volatile uint32_t* progress_host() const { return dev_error_host + 32; }
    volatile uint32_t* culled_host() const { return dev_error_host + 48; }      // [0] pixels of the latest call in empty tiles that were left out (black sky), [1] in tiles rendered without a walk (order_kernel)
    std::atomic<uint64_t> progress_total{0};        // blocks of the call in flight (all its launches)
    struct MultiContext* multi = nullptr;           // rt_render_frame_multi: streams, events and buffers, owned by the scene passed first
    std::mutex multi_frame_lock;
    DeviceBuffer multi_frame_rgb, multi_frame_rgba; // rt_render_frame_multi_host: the frame on this device before it is copied to the host
    DevSceneView view{};
    RtSceneInfo info{};
    std::vector<uint32_t> tlas_order;
    std::vector<std::vector<uint32_t>> blas_order;
    std::vector<void*> allocations;
    unsigned char* scratch_ptr = nullptr;   // this device's copy of the static scratch array (rt_render_chunk_device_noalloc)
    // rt_scene_update_transforms: what the host needs to re-derive the transform records, and the device-side refit tables
    RtCamera camera{};
    std::vector<RtEntity> desc_entities;
    std::vector<uint32_t> tlas_perm;
    uint64_t n_translations = 0, n_rotations = 0;
    std::vector<Word4> xforms_host;
    size_t xforms_words = 0;
    std::vector<uint32_t> refit_level_start;
    const RefitRec* refit_dev = nullptr;
    float* ent_box_dev = nullptr;           // 8 floats per entity + 8 for the root box
    bool refit_valid = true;                // false once rt_scene_rebuild_tlas has replaced the topology the refit tables describe
    // rt_scene_rebuild_tlas: sizes of the build, and the buffers of the device-side build (allocated by its first call)
    size_t node_count = 0;                  // records of the scene as built: the rebuilt TLAS's records are appended behind them
    uint32_t tlas_hot_records = 0, blas_walk_depth = 0, lds_nodes_built = 0;
    struct TlasRebuild {
        bool ready = false;
        unsigned long long *keys_a = nullptr, *keys_b = nullptr;
        void* sort_temp = nullptr; size_t sort_temp_bytes = 0;
        LbvhNode* lnodes = nullptr;
        uint32_t *parent = nullptr, *height = nullptr, *arrived = nullptr, *number = nullptr, *queue = nullptr;
        float *box = nullptr, *bounds = nullptr;
        DevEntity* entities_alt = nullptr;      // the build's OUTPUT (key order); swapped with the scene's arrays when a build commits
        float* ent_box_alt = nullptr;
        DevEntity* entities_stage = nullptr;    // moved entity records / boxes / transform records before the permutation: a build that is refused
        float* ent_box_stage = nullptr;         // (RT_E_DEPTH) has written nothing a render call reads
        Word4* xforms_stage = nullptr;
        bool topology_stale = false;            // a refused build left ANOTHER topology in lnodes / number than the scene's records: no refit from them
    } rebuild;
    // rt_scene_rebuild_blas: where each mesh's records and triangles live, and the buffers of its device-side build (allocated by the first call per mesh)
    uint32_t tlas_walk_depth = 0;
    std::vector<uint32_t> blas_hot_base, blas_hot_count, blas_walk_depths;
    std::vector<DevMesh> meshes_host;
    bool node_room = false;                 // the node array has been re-allocated with room for every rebuilt tree (ensure_node_room)
    struct BlasRebuild {
        bool ready = false;
        RtMeshTri* tris = nullptr;
        float *box_in = nullptr, *box_sorted = nullptr, *ibox = nullptr, *bounds = nullptr, *root = nullptr;
        unsigned long long *keys_a = nullptr, *keys_b = nullptr;
        void* sort_temp = nullptr; size_t sort_temp_bytes = 0;
        LbvhNode* lnodes = nullptr;
        uint32_t *parent = nullptr, *height = nullptr, *arrived = nullptr, *number = nullptr, *queue = nullptr, *where = nullptr, *rank_stage = nullptr;
        DevTriHot* hot_stage = nullptr; DevTriCold* cold_stage = nullptr;
    };
    std::vector<BlasRebuild> blas_rebuild;
    // Device-pointer entry points run on the CALLER's stream and share this workspace:
    // `lock` serialises their enqueueing, the caller keeps them on one stream.
    std::mutex lock;
    Workspace shared;
    // Host-pointer entry points (rt_render_chunk, rt_primary_hits) each borrow a workspace
    // with its own stream, so that calls from several host threads (the reference's
    // worker threads, src/render.rs:186-224) overlap on the GPU.
    std::mutex pool_lock;
    std::condition_variable pool_cv;
    std::vector<std::unique_ptr<Workspace>> pool;
    std::vector<Workspace*> pool_free;
    std::atomic<int> host_calls_in_flight{0};
    int grid_blocks = 0;
    size_t lds_bytes = 0, stack_lds_bytes = 0;
    std::mutex profile_lock;
    bool profiling = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> profile_events;
};

namespace {

// Borrows a workspace for the duration of one host-pointer call.
struct WorkspaceLease {
    RtScene* scene; Workspace* ws = nullptr; int rc = RT_OK;
    explicit WorkspaceLease(RtScene* s) : scene(s) {
        s->host_calls_in_flight.fetch_add(1);
        std::unique_lock<std::mutex> g(s->pool_lock);
        for (;;) {
            if (!s->pool_free.empty()) { ws = s->pool_free.back(); s->pool_free.pop_back(); return; }
            if (s->pool.size() < RT_MAX_WORKSPACES) {
                std::unique_ptr<Workspace> w(new Workspace());
                rc = w->init(true);
                if (rc) { w->release(); return; }
                ws = w.get(); s->pool.push_back(std::move(w));
                return;
            }
            s->pool_cv.wait(g);
        }
    }
    ~WorkspaceLease() {
        scene->host_calls_in_flight.fetch_sub(1);
        if (!ws) return;
        { std::lock_guard<std::mutex> g(scene->pool_lock); scene->pool_free.push_back(ws); }
        scene->pool_cv.notify_one();
    }
};

}  // namespace

namespace {

template <class T>
int upload(RtScene* s, const std::vector<T>& v, const T** dst) {
    void* p = nullptr;
    const size_t bytes = v.size() * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    s->allocations.push_back(p);
    if (bytes) HIP_TRY(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
    *dst = reinterpret_cast<const T*>(p);
    s->info.device_bytes += bytes;
    return RT_OK;
}

// The kernels' sticky error word (host-mapped: no synchronisation needed to look at it).
int check_device_error(const RtScene* scene) {
    const uint32_t e = scene->dev_error_host ? *(volatile const uint32_t*)scene->dev_error_host : 0u;
    if (e == 0u) return RT_OK;
    if (e & RT_DEVERR_POISONED) return set_error(RT_E_HIP, "an earlier rt_scene_rebuild_tlas / rt_scene_update_transforms failed half way: the scene's device data is inconsistent; create the scene anew");
    return set_error(RT_E_DEPTH, (e & RT_DEVERR_STACK) ? "a render kernel ran out of traversal stack (BVH deeper than the scene build accounted for): pixels of that launch are incomplete"
                                                       : "a render kernel ran out of deferred-ray slots: pixels of that launch are incomplete");
}

int validate_call(const RtScene* scene, const RtSettings* st, const RtRenderChunk* ch) {
    if (!scene || !st || !ch) return set_error(RT_E_INVALID, "null argument");
    if (scene->device == RT_DEVICE_HOST_ONLY) return set_error(RT_E_NO_DEVICE, "scene was created host-only (RT_DEVICE_HOST_ONLY): nothing to render on");
    if (int rc = check_device_error(scene)) return rc;      // sticky: an earlier launch on this scene overflowed a capacity
    if (scene->dev_error_host && *scene->cancel_host() != 0u) return set_error(RT_E_CANCELLED, "the scene is cancelled (rt_scene_cancel): clear it with rt_scene_cancel(scene, 0)");
    if (st->width == 0 || st->height == 0 || st->samples_per_pixel == 0) return set_error(RT_E_INVALID, "RenderSettings: width, height and samples_per_pixel must be > 0");
    if (st->width > 65535 || st->height > 65535) return set_error(RT_E_LIMIT, "RenderSettings: image sides are limited to 65535 (RNG counter packs x | y<<16)");
    if (st->max_reflections > RT_MAX_REFLECTIONS) return set_error(RT_E_LIMIT, "RenderSettings: max_reflections is limited to 31 (recursion-tree node id is 32 bit)");
    if (ch->width == 0 || ch->height == 0) return set_error(RT_E_INVALID, "RenderChunk: empty chunk");
    if ((uint64_t)ch->left + ch->width > st->width || (uint64_t)ch->top + ch->height > st->height) return set_error(RT_E_INVALID, "RenderChunk: outside the image");
    // work items of a launch are counted in 32 bits, scratch is sized per chunk: 2^28 pixels (16384 x 16384) per call
    if ((((uint64_t)ch->width + 7) / 8) * (((uint64_t)ch->height + 7) / 8) * 64 > (1ull << 28))
        return set_error(RT_E_LIMIT, "RenderChunk: more than 2^28 pixels in one call; render the image in several chunks (create_render_chunks)");
    return RT_OK;
}

// bytes of the tile-order scratch of a launch over n_tiles tiles (enqueue_accumulate)
size_t order_buffer_bytes(size_t n_tiles) { return ((n_tiles + RT_ORDER_GROUP - 1) / RT_ORDER_GROUP) * RT_GROUP_STRIDE * 4 + n_tiles * 8 + 16; }

// Enqueue: accumulate the sum of clamped samples [first, first+count) of the chunk
// into `acc` (float3 image with pitch acc_pitch, chunk origin at acc_x/acc_y).
struct Interleave { uint32_t strip_h = 0, period_rows = 0, first_row = 0, col_n = 0, col_r = 0; };

int enqueue_accumulate(RtScene* scene, Workspace& ws, const RtSettings* st, const RtRenderChunk* ch, uint64_t seed, uint32_t sample_first,
                       uint32_t sample_count, float* acc, uint32_t acc_pitch, uint32_t acc_x, uint32_t acc_y, bool init_zero,
                       unsigned long long* counters_dev, hipStream_t stream, Interleave il = Interleave()) {
    if (sample_count == 0 || (uint64_t)sample_first + sample_count > st->samples_per_pixel) return set_error(RT_E_INVALID, "sample range outside samples_per_pixel");
    const uint32_t npix = ch->width * ch->height;
    const uint32_t seg_lo = sample_first / RT_SAMPLE_SEGMENT;
    const uint32_t seg_hi = (sample_first + sample_count - 1) / RT_SAMPLE_SEGMENT;     // inclusive
    const uint32_t nseg = seg_hi - seg_lo + 1;
    const uint32_t tiles_x = (ch->width + 7) / 8;
    const uint32_t pix_slots = tiles_x * ((ch->height + 7) / 8) * 64;
    // Work split.  Coarse: one work item = one pixel x one whole segment (best lane utilisation:
    // a wave waits for its slowest lane once per 8 samples).  Fine: one item = one sample, for
    // launches that would leave the GPU under-filled or be dominated by their slowest wave
    // (small frames, small chunks).  Same bits either way (see fold_kernel).
    // Measured (weekend): a full GPU runs the fine split 20 % slower (44 -> 54 ms per 1080p frame: eight
    // times the block claims, scratch writes and fold reads), an under-filled one up to 5x faster
    // (400x225x16: 2.48 -> 0.52 ms); so: fine only when the coarse
    // blocks would not even occupy the resident waves and no other host call is rendering on this
    // scene at the same time (concurrent chunk calls fill the GPU together).
    const uint64_t coarse_blocks = (uint64_t)(pix_slots / 64) * nseg;
    SceneOptions opt;
    { std::lock_guard<std::mutex> og(scene->opt_lock); opt = scene->opt; }
    bool fine = coarse_blocks < (uint64_t)scene->grid_blocks * (RT_BLOCK / 64) && scene->host_calls_in_flight.load() <= 1;
    if (opt.fine_split >= 0) fine = opt.fine_split != 0;                                 // RT_OPT_FINE_SPLIT (experiments / tests): force either split
    if (fine && ws.partial.fixed && ws.partial.bytes < (uint64_t)pix_slots * 12ull * RT_SAMPLE_SEGMENT) fine = false;      // (same bits either way)
    const uint32_t unit_len = fine ? 1u : RT_SAMPLE_SEGMENT;
    // segments per launch: keep the scratch below ~2 GiB (a 4K frame then still runs 16 units per launch)
    const uint32_t units_per_seg = RT_SAMPLE_SEGMENT / unit_len;
    const uint64_t partial_cap = ws.partial.fixed ? ws.partial.bytes : (RT_LENS_CELLS == 64 ? (1ull << 33) : (1ull << 31));      // (the static scratch of the no-alloc entry point is what it is)
    uint32_t batch = (uint32_t)(partial_cap / ((uint64_t)pix_slots * 12ull * units_per_seg));
    if (batch == 0) batch = 1;
    if (batch > nseg) batch = nseg;
    if (RT_LENS_CELLS == 64 && batch >= 64) batch &= ~63u;
    else if (batch >= 16) batch &= ~15u;      // (whole groups of units that share their rounds, see round_units_shift
    else if (batch > 8) batch = 8;            //  and the lens-cell rounds over groups of 16 units)
    int rc = ws.partial.ensure((size_t)batch * units_per_seg * pix_slots * 12);
    if (rc) return rc;

    RenderParams P{};
    P.width = st->width; P.height = st->height; P.max_reflections = st->max_reflections; P.spp = st->samples_per_pixel;
    P.left = ch->left; P.top = ch->top; P.cw = ch->width; P.ch = ch->height;
    P.seed = seed; P.sample_first = sample_first; P.sample_count = sample_count;
    P.tiles_x = tiles_x; P.pix_slots = pix_slots; P.unit_len = unit_len;
    P.partial = (float*)ws.partial.p;
    P.counters = counters_dev;
    P.strip_h = il.strip_h; P.strip_period_rows = il.period_rows; P.strip_first_row = il.first_row; P.col_n = il.col_n; P.col_r = il.col_r;
    P.lens_coc_px = opt.lens_coc_px;      // circle of confusion in pixels from which a tile runs lens-cell rounds (a huge value turns them off)
    P.lens_stratify = opt.lens_stratify ? 1u : 0u;
    P.kind_order = opt.kind_order ? 1u : 0u;
    P.error_flag = scene->dev_error_dev;
    P.cancel_word = scene->dev_error_dev + 16; P.progress_word = scene->dev_error_dev + 32; P.progress_base = 0;
    {
        // progress of this call: blocks over all its launches (an estimate for concurrent calls on one scene: they share the words)
        uint64_t blocks_total = 0;
        for (uint32_t b0 = 0; b0 < nseg; b0 += batch) {
            const uint32_t bn = (nseg - b0 < batch) ? nseg - b0 : batch;
            const uint32_t s_lo = (seg_lo + b0) * RT_SAMPLE_SEGMENT > sample_first ? (seg_lo + b0) * RT_SAMPLE_SEGMENT : sample_first;
            const uint32_t s_hi = (seg_lo + b0 + bn) * RT_SAMPLE_SEGMENT < sample_first + sample_count ? (seg_lo + b0 + bn) * RT_SAMPLE_SEGMENT : sample_first + sample_count;
            blocks_total += (uint64_t)(pix_slots / 64) * ((s_hi - 1) / unit_len - s_lo / unit_len + 1);
        }
        *scene->progress_host() = 0u;
        scene->progress_total.store(blocks_total);
    }
    DevSceneView view = scene->view;
    if (opt.debug_stack_entries && opt.debug_stack_entries < view.stack_entries) view.stack_entries = opt.debug_stack_entries;      // (tests of RT_E_DEPTH)
    // Empty tiles: a tile whose every camera ray misses the scene's root box (tile_misses_root) is all black sky - no block of it is handed out, fold_kernel takes its
    // samples as the zeros they are, order_kernel counts its camera rays.  Only where a miss is (0, 0, 0) whatever the ray: a black sky and at least one ray per sample
    // (:562-564: max_reflections 0 casts none); not for scenes whose boxes do not bound (loose_boxes).  Same bits, same counters (RT_OPT_CULL_EMPTY_TILES 0 traces them).
    P.cull_empty = (opt.cull_empty_tiles && !view.loose_boxes && st->max_reflections > 0u) ? (view.sky == RT_SKY_BLACK ? 2u : (view.mesh_entities ? 0u : 1u)) : 0u;      // (1: render_kernel's NOHIT)
    P.culled_word = scene->dev_error_dev + 48;
    // (the 16 probe rays per tile cost Dreadnaught's frame 0.19 ms of 6.9 and a rank's share of 8 the same 0.15-0.19 ms of 1.2: a traversal's latency, not its work)
    P.classify_by_beam = (P.cull_empty && view.uniform_cost_class != 0xffu && !(view.coc_rel_max * (float)st->width > opt.lens_coc_px)) ? 1u : 0u;
    scene->culled_host()[0] = 0u; scene->culled_host()[1] = 0u;
    const uint32_t* tile_bin_dev = nullptr;
    // expensive tiles first (whole-segment launches over enough tiles to have a tail worth hiding)
    const uint32_t n_tiles = pix_slots / 64;
    static const bool order_tiles = []() { const char* e = getenv("RT_TILE_ORDER"); return !e || atoi(e) != 0; }();
    if (!fine && order_tiles && n_tiles >= 256) {
        const uint32_t n_groups = (n_tiles + RT_ORDER_GROUP - 1u) / RT_ORDER_GROUP;
        if ((rc = ws.order.ensure(order_buffer_bytes(n_tiles)))) return rc;
        // layout: [group_counts: n_groups x 4][tile_bin: n_tiles][3 spare + the expensive-tile count = tile_order[-1]][tile_order: n_tiles]
        uint32_t* group_counts = (uint32_t*)ws.order.p; uint32_t* tile_bin = group_counts + (size_t)n_groups * RT_GROUP_STRIDE; uint32_t* tile_order = tile_bin + n_tiles + 4;
        tile_bin_dev = tile_bin;
        HIP_TRY(hipMemsetAsync(group_counts, 0, (size_t)n_groups * RT_GROUP_STRIDE * 4, stream));
        if (P.cull_empty) {
            // (workgroups of ONE wave: a tile's walk is one thread's chain of dependent reads - a rank's share of a 4K frame is 16 200 tiles, 64 workgroups of 256 on 256 CUs)
            hipLaunchKernelGGL(empty_tiles_kernel, dim3((n_tiles + 63u) / 64u), dim3(64), scene->stack_lds_bytes, stream, view, P, n_tiles, tile_bin);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(classify_kernel, dim3((n_tiles * 16 + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), scene->stack_lds_bytes, stream, view, P,
                           n_tiles, tile_bin, group_counts);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(order_kernel, dim3(n_groups), dim3(RT_ORDER_GROUP), 0, stream, (const uint32_t*)tile_bin, n_tiles, (const uint32_t*)group_counts, tile_order,
                           counters_dev, sample_count, P.culled_word, P.cull_empty);
        HIP_TRY(hipGetLastError());
        P.tile_order = tile_order;
    }
    for (uint32_t b0 = 0; b0 < nseg; b0 += batch) {
        const uint32_t bn = (nseg - b0 < batch) ? nseg - b0 : batch;
        // the units of segments [seg_lo + b0, seg_lo + b0 + bn) that hold samples of the range
        const uint32_t s_lo = (seg_lo + b0) * RT_SAMPLE_SEGMENT > sample_first ? (seg_lo + b0) * RT_SAMPLE_SEGMENT : sample_first;
        const uint32_t s_hi = (seg_lo + b0 + bn) * RT_SAMPLE_SEGMENT < sample_first + sample_count ? (seg_lo + b0 + bn) * RT_SAMPLE_SEGMENT
                                                                                                    : sample_first + sample_count;     // exclusive
        P.unit_first = s_lo / unit_len; P.unit_count = (s_hi - 1) / unit_len - P.unit_first + 1;
        {
            // up to 8 units share their rounds (64 samples of one pixel side by side); RT_OPT_ROUND_UNITS = 1 / 2 / 4 lowers the cap (experiments, tests)
            const int cap = opt.round_units;
            const uint32_t max_shift = cap >= 8 ? 3u : cap >= 4 ? 2u : cap >= 2 ? 1u : 0u;
            P.round_units_shift = 0;
            P.lens_rounds = P.tile_order != nullptr;          // (whole-segment units over enough tiles: the tiles are classified)
            while (!fine && P.round_units_shift < max_shift && P.unit_count % (2u << P.round_units_shift) == 0) P.round_units_shift++;
            // (without the stratified sampler the samples of one index do not share a lens cell: nothing to gain from that round shape)
            if (P.round_units_shift != 3u || !P.lens_stratify) P.lens_rounds = 0u;
            // single-round claims for launches of few blocks per resident wave (render_kernel: block claims); RT_SINGLE_ROUNDS=1 always (where compiled in), 0 never
            static const int single = []() { const char* e = getenv("RT_SINGLE_ROUNDS"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
            const uint32_t below = single < 0 ? (uint32_t)scene->grid_blocks * (RT_BLOCK / 64u) * RT_SINGLE_ROUNDS_BELOW : (single ? 0xffffffffu : 0u);
            P.single_rounds = (!fine && unit_len == RT_SAMPLE_SEGMENT) ? below : 0u;
        }
        const uint64_t items = (uint64_t)P.pix_slots * P.unit_count;
        uint64_t blocks = (items + RT_BLOCK - 1) / RT_BLOCK;
        if (blocks > (uint64_t)scene->grid_blocks) blocks = scene->grid_blocks;
        P.block_counter = (uint32_t*)((char*)ws.ctrl.p + 1024);
        HIP_TRY(hipMemsetAsync(P.block_counter, 0, RT_QUEUES * 64, stream));
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        bool profiling;
        { std::lock_guard<std::mutex> pg(scene->profile_lock); profiling = scene->profiling && !ws.no_alloc; }      // (events are created per launch)
        if (profiling) { HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventRecord(ev0, stream)); }
#ifdef RT_PHASE_STATS
        if (ws.no_alloc) return set_error(RT_E_INVALID, "the phase-statistics build allocates: not with the no-alloc entry point");
        static unsigned long long* phase_dev = nullptr;
        if (!phase_dev) HIP_TRY(hipMalloc((void**)&phase_dev, PH_COUNT * 16 + 8 * 32));
        HIP_TRY(hipMemsetAsync(phase_dev, 0, PH_COUNT * 16 + 8 * 32, stream));
        P.phase_stats = phase_dev;
#endif
        {
            // one instantiation per combination of scene traits (code and per-lane state a scene does not need are compiled out)
            using Kernel = void (*)(const DevSceneView, const RenderParams);
            // index bits: 1 mesh instances, 2 spheres, 4 planes, 8 general transforms, 16 rich shading, 32 depth of field
#define RT_K(M, S_, P_, X, R, D) render_kernel<false, M, S_, P_, X, R, D>
#define RT_K8(X, R, D) nullptr, RT_K(true, false, false, X, R, D), RT_K(false, true, false, X, R, D), RT_K(true, true, false, X, R, D), \
                       RT_K(false, false, true, X, R, D), RT_K(true, false, true, X, R, D), RT_K(false, true, true, X, R, D), RT_K(true, true, true, X, R, D)
            static const Kernel kernels[64] = {RT_K8(false, false, false), RT_K8(true, false, false), RT_K8(false, true, false), RT_K8(true, true, false),
                                               RT_K8(false, false, true),  RT_K8(true, false, true),  RT_K8(false, true, true),  RT_K8(true, true, true)};
#undef RT_K8
#undef RT_K
            // the same table for scenes with mesh instances and more than one entity (MULTI: own-box pre-test of instances); only the mesh columns exist
#define RT_K(M, S_, P_, X, R, D) render_kernel<false, M, S_, P_, X, R, D, true>
#define RT_K8(X, R, D) nullptr, RT_K(true, false, false, X, R, D), nullptr, RT_K(true, true, false, X, R, D), \
                       nullptr, RT_K(true, false, true, X, R, D), nullptr, RT_K(true, true, true, X, R, D)
            static const Kernel kernels_multi[64] = {RT_K8(false, false, false), RT_K8(true, false, false), RT_K8(false, true, false), RT_K8(true, true, false),
                                                     RT_K8(false, false, true),  RT_K8(true, false, true),  RT_K8(false, true, true),  RT_K8(true, true, true)};
#undef RT_K8
#undef RT_K
            const DevSceneView& v = view;
            const bool dof = v.coc_rel_max * (float)st->width > P.lens_coc_px;       // (inf x width = inf: unbounded scenes keep both round shapes)
            int which = (v.mesh_entities ? 1 : 0) | (v.sphere_entities ? 2 : 0) | (v.plane_entities ? 4 : 0) | (v.general_xforms ? 8 : 0) | (v.rich_shading ? 16 : 0) |
                        (dof ? 32 : 0);
            if ((which & 7) == 0) which |= 2;                                        // (an empty scene: any instantiation will do)
            static const bool show_traits = getenv("RT_DEBUG_TRAITS") != nullptr;      // diagnostics: which instantiation a scene runs
            if (show_traits) fprintf(stderr, "traits: which=%d loose=%u any_hit=%u lds_nodes=%u lds_ents=%u stack=%u lds_bytes=%zu\n", which, v.loose_boxes, v.shadow_any_hit, v.lds_nodes, v.lds_ents, v.stack_entries, scene->lds_bytes);
            const bool multi = v.mesh_entities && v.entity_count > 1u;
            // plain translated spheres with their records in LDS (S.lds_ents, size_sphere_cache): index bits 1 rich shading, 2 depth of field
            static const Kernel kernels_ecache[4] = {render_kernel<false, false, true, false, false, false, false, false, true>, render_kernel<false, false, true, false, false, true, false, false, true>,
                                                     render_kernel<false, false, true, false, false, false, true, false, true>,  render_kernel<false, false, true, false, false, true, true, false, true>};
            const bool ecache = v.lds_ents != 0u && !v.loose_boxes && (which & 15) == 2;
            const Kernel kernel = v.loose_boxes ? render_kernel<true, true, true, true, true, true, true> : ecache ? kernels_ecache[(which >> 4) & 3] :
                                  (multi ? kernels_multi[which] : kernels[which]);
            hipLaunchKernelGGL(kernel, dim3((uint32_t)blocks), dim3(RT_BLOCK), scene->lds_bytes, stream, view, P);
        }
        HIP_TRY(hipGetLastError());
#ifdef RT_PHASE_STATS
        {
            unsigned long long h[PH_COUNT * 2 + 32];
            HIP_TRY(hipMemcpyAsync(h, phase_dev, sizeof(h), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
#ifdef RT_PHASE_TIME
            static const char* tnames[TM_COUNT] = {"other", "node walk", "leaf: triangles", "leaf: entities", "shading", "rounds", "claim / idle"};
            unsigned long long total = 0;
            for (int k = 0; k < TM_COUNT; k++) total += h[2 * k];
            for (int k = 0; k < TM_COUNT; k++)
                fprintf(stderr, "time  %-18s %6.2f %% of the wave cycles (%llu waves)\n", tnames[k], total ? 100.0 * (double)h[2 * k] / (double)total : 0.0, h[2 * k + 1]);
#else
            static const char* names[PH_COUNT] = {"node", "tri", "entity", "shade", "shade_path_hit", "shade_shadow_hit", "stage_a", "stage_c", "trace_init", "burst", "leaf_tri", "leaf_tlas", "outer", "pop_inst", "node_global", "node_not_ii", "inst_header"};
            for (int k = 0; k < PH_COUNT; k++)
                fprintf(stderr, "phase %-18s lanes %14llu passes %12llu util %.3f\n", names[k], h[2 * k], h[2 * k + 1], h[2 * k + 1] ? (double)h[2 * k] / (64.0 * (double)h[2 * k + 1]) : 0.0);
            for (int k = 1; k < 8; k++) {      // bursts by the kinds of ray they hold: c camera, s secondary, h shadow
                const unsigned long long* b = h + 2 * PH_COUNT + 4 * k;
                if (b[0]) fprintf(stderr, "bursts %c%c%c  n %11llu  start lanes %5.1f  iterations %7.1f  (share of all iterations below)  lanes tracing per iteration %5.1f  total iterations %13llu\n",
                                  (k & 1) ? 'c' : '-', (k & 2) ? 's' : '-', (k & 4) ? 'h' : '-', b[0], (double)b[1] / (double)b[0], (double)b[2] / (double)b[0], (double)b[3] / (double)b[2], b[2]);
            }
#endif
        }
#endif
        if (profiling) {
            HIP_TRY(hipEventRecord(ev1, stream));
            std::lock_guard<std::mutex> pg(scene->profile_lock);
            scene->profile_events.emplace_back(ev0, ev1);
        }
        hipLaunchKernelGGL(fold_kernel, dim3((npix + 255) / 256), dim3(256), 0, stream, (const float*)ws.partial.p, P.unit_count, P.unit_first, unit_len, npix, acc, acc_pitch,
                           ch->width, acc_x, acc_y, (init_zero && b0 == 0) ? 1 : 0, tiles_x, pix_slots, (P.tile_order && P.cull_empty == 2u) ? tile_bin_dev : (const uint32_t*)nullptr);
        HIP_TRY(hipGetLastError());
        P.progress_base += (uint32_t)((uint64_t)(pix_slots / 64) * P.unit_count);
    }
    return RT_OK;
}

// The sphere cache of the render kernel (DevSceneView::lds_ents): all or nothing - every entity's word 0, 16 bytes each, behind the node cache, when the
// scene is plain translated spheres only and a block's share of the LDS (four blocks per CU) still has the room.  Returns the bytes it takes.
size_t size_sphere_cache(DevSceneView& v, size_t lds_bytes_so_far) {
    static const bool off = getenv("RT_NO_SPHERE_CACHE") != nullptr;      // (A/B measurements)
    const size_t block_budget = (size_t)160 * 1024 * RT_BLOCK / (256 * RT_WAVES_PER_SIMD);
    const bool plain_spheres = v.sphere_entities && !v.plane_entities && !v.mesh_entities && !v.general_xforms && !v.loose_boxes;
    const size_t bytes = (size_t)v.entity_count * 16u;
    v.lds_ents = (!off && plain_spheres && lds_bytes_so_far + bytes <= block_budget) ? v.entity_count : 0u;
    return v.lds_ents ? bytes : 0u;
}

}  // namespace

extern "C" {

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* rt_last_error(void) { return g_error.c_str(); }

RtScene* rt_scene_create(const RtSceneDesc* desc, int device) {
    FlatScene flat; std::string err;
    int rc = build_flat_scene(desc, flat, err);
    if (rc) { set_error(rc, err); return nullptr; }
    if (device == RT_DEVICE_HOST_ONLY) {
        // acceleration structures only (rt_scene_info / rt_scene_bvh_order); nothing is uploaded
        RtScene* s = new RtScene();
        s->device = RT_DEVICE_HOST_ONLY;
        s->info = flat.info;
        s->tlas_order = std::move(flat.tlas_order); s->blas_order = std::move(flat.blas_order);
        return s;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error(RT_E_NO_DEVICE, "no HIP device visible"); return nullptr; }
    if (device < 0 || device >= ndev) { set_error(RT_E_INVALID, "device index out of range"); return nullptr; }
    RtScene* s = new RtScene();
    s->device = device;
    auto fail = [&](void) -> RtScene* { rt_scene_destroy(s); return nullptr; };
    if (hipSetDevice(device) != hipSuccess) { set_error(RT_E_HIP, "hipSetDevice failed"); return fail(); }
    s->view = flat.view; s->info = flat.info;
    s->tlas_order = std::move(flat.tlas_order); s->blas_order = std::move(flat.blas_order);
    if (scratch_address(device, &s->scratch_ptr)) return fail();      // (this device is current: its copy of the array)
    s->camera = desc->camera;
    s->desc_entities.assign(desc->entities, desc->entities + desc->entity_count);
    s->tlas_perm = std::move(flat.tlas_perm);
    s->n_translations = desc->translation_count; s->n_rotations = desc->rotation_count;
    s->xforms_words = flat.xforms.size();
    s->refit_level_start = std::move(flat.refit_level_start);
    s->node_count = flat.nodes.size(); s->tlas_hot_records = flat.tlas_hot_records; s->blas_walk_depth = flat.blas_walk_depth; s->lds_nodes_built = flat.view.lds_nodes;
    s->tlas_walk_depth = flat.tlas_walk_depth; s->blas_hot_base = flat.blas_hot_base; s->blas_hot_count = flat.blas_hot_count; s->blas_walk_depths = flat.blas_walk_depths;
    s->meshes_host = flat.meshes; s->blas_rebuild.resize(flat.meshes.size());
    if (flat.refit.empty()) flat.refit.emplace_back();
    if (upload(s, flat.refit, &s->refit_dev)) return fail();
    if (upload(s, flat.ent_box, &s->view.ent_box)) return fail();      // the build's entity boxes; rt_scene_update_transforms recomputes them in place
    s->ent_box_dev = reinterpret_cast<float*>(const_cast<Word4*>(s->view.ent_box));
    if (flat.nodes.size() >= ((size_t)1 << 26) || flat.tri_hot.size() >= ((size_t)1 << 26) || flat.entities.size() >= ((size_t)1 << 26)) {
        set_error(RT_E_LIMIT, "more than 2^26 tree records, triangles or entities (the walk addresses them by 32-bit byte offsets)"); return fail();
    }
    if (upload(s, flat.nodes, &s->view.nodes) || upload(s, flat.entities, &s->view.entities) || upload(s, flat.xforms, &s->view.xforms) ||
        upload(s, flat.meshes, &s->view.meshes) || upload(s, flat.tri_hot, &s->view.tri_hot) || upload(s, flat.tri_cold, &s->view.tri_cold) ||
        upload(s, flat.tri_rank, &s->view.tri_rank) || upload(s, flat.planes, &s->view.planes) || upload(s, flat.materials, &s->view.materials) ||
        upload(s, flat.textures, &s->view.textures) || upload(s, flat.tex_set_entries, &s->view.tex_set_entries) ||
        upload(s, flat.texels, &s->view.texels) || upload(s, flat.lights, &s->view.lights))
        return fail();
    // launch shape: a resident grid (persistent lanes stride over the work items)
    s->lds_bytes = ((size_t)s->view.stack_entries + (s->view.mesh_entities ? 3 : 0)) * RT_BLOCK * sizeof(uint32_t);      // + 3 rows: parked world-space 1/d
    s->stack_lds_bytes = s->lds_bytes;
    s->lds_bytes += (size_t)s->view.lds_nodes * RT_NODE_CACHE_STRIDE;      // the node cache (sized by scene_build.cpp to fit four blocks per CU)
    s->lds_bytes += size_sphere_cache(s->view, s->lds_bytes);
    if (s->stack_lds_bytes > 64 * 1024) { set_error(RT_E_DEPTH, "BVH too deep for the LDS traversal stack (more than 64 levels)"); return fail(); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_error(RT_E_HIP, "hipGetDeviceProperties failed"); return fail(); }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_kernel<true, true, true, true, true, true, true>, RT_BLOCK, s->lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
    s->grid_blocks = prop.multiProcessorCount * per_cu;
    if (s->shared.init(false)) return fail();
    if (hipHostMalloc((void**)&s->dev_error_host, 256, hipHostMallocMapped) != hipSuccess) { s->dev_error_host = nullptr; set_error(RT_E_HIP, "hipHostMalloc (error word) failed"); return fail(); }
    memset(s->dev_error_host, 0, 256);      // error word, cancel word, progress word, culled-pixels word: one cache line each
    if (hipHostGetDevicePointer((void**)&s->dev_error_dev, s->dev_error_host, 0) != hipSuccess) { set_error(RT_E_HIP, "hipHostGetDevicePointer failed"); return fail(); }
    // defaults of the options from the environment, read once
    if (const char* e = getenv("RT_LENS_STRATIFY")) s->opt.lens_stratify = atoi(e) != 0;
    if (const char* e = getenv("RT_FINE_SPLIT")) s->opt.fine_split = atoi(e) != 0;
    if (const char* e = getenv("RT_ROUND_UNITS")) s->opt.round_units = atoi(e);
    if (const char* e = getenv("RT_LENS_COC")) s->opt.lens_coc_px = (float)atof(e);
    if (const char* e = getenv("RT_KIND_ORDER")) s->opt.kind_order = atoi(e) != 0;
    if (const char* e = getenv("RT_CULL_TILES")) s->opt.cull_empty_tiles = atoi(e) != 0;
    return s;
}

int rt_scene_set_option(RtScene* scene, uint32_t option, double value) {
    if (!scene) return set_error(RT_E_INVALID, "null scene");
    std::lock_guard<std::mutex> og(scene->opt_lock);
    switch (option) {
        case RT_OPT_LENS_SAMPLER:
            if (value != 0.0 && value != 1.0) return set_error(RT_E_INVALID, "RT_OPT_LENS_SAMPLER: 0 (reference) or 1 (stratified)");
            scene->opt.lens_stratify = (int)value; return RT_OK;
        case RT_OPT_FINE_SPLIT:
            if (value != -1.0 && value != 0.0 && value != 1.0) return set_error(RT_E_INVALID, "RT_OPT_FINE_SPLIT: -1 (automatic), 0 or 1");
            scene->opt.fine_split = (int)value; return RT_OK;
        case RT_OPT_ROUND_UNITS:
            if (value != 1.0 && value != 2.0 && value != 4.0 && value != 8.0) return set_error(RT_E_INVALID, "RT_OPT_ROUND_UNITS: 1, 2, 4 or 8");
            scene->opt.round_units = (int)value; return RT_OK;
        case RT_OPT_LENS_COC_PX:
            if (!(value >= 0.0)) return set_error(RT_E_INVALID, "RT_OPT_LENS_COC_PX: a circle of confusion in pixels, >= 0");
            scene->opt.lens_coc_px = (float)value; return RT_OK;
        case RT_OPT_CULL_EMPTY_TILES:
            if (value != 0.0 && value != 1.0) return set_error(RT_E_INVALID, "RT_OPT_CULL_EMPTY_TILES: 0 or 1");
            scene->opt.cull_empty_tiles = (int)value; return RT_OK;
        case RT_OPT_DEBUG_STACK_ENTRIES:
            if (!(value >= 0.0 && value <= 1024.0)) return set_error(RT_E_INVALID, "RT_OPT_DEBUG_STACK_ENTRIES: 0 (as built) .. 1024");
            scene->opt.debug_stack_entries = (uint32_t)value;
            if (scene->dev_error_host) {          // also re-arms the sticky error word (no launch may be in flight)
                if (scene->device != RT_DEVICE_HOST_ONLY) { HIP_TRY(hipSetDevice(scene->device)); HIP_TRY(hipDeviceSynchronize()); }
                *scene->dev_error_host = 0u;
            }
            return RT_OK;
        case RT_OPT_DEBUG_REBUILD_STACK_LIMIT:
            if (!(value >= 0.0 && value <= 1024.0)) return set_error(RT_E_INVALID, "RT_OPT_DEBUG_REBUILD_STACK_LIMIT: 0 (the LDS limit) .. 1024");
            scene->opt.debug_rebuild_stack_limit = (uint32_t)value; return RT_OK;
        default: return set_error(RT_E_INVALID, "rt_scene_set_option: unknown option");
    }
}

void rt_scene_destroy(RtScene* s) {
    if (!s) return;
    if (s->device == RT_DEVICE_HOST_ONLY) { delete s; return; }
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    if (s->multi) { s->multi->release(); delete s->multi; s->multi = nullptr; (void)hipSetDevice(s->device); }
    s->multi_frame_rgb.release(); s->multi_frame_rgba.release();
    for (auto& w : s->pool) w->release();
    s->shared.release();
    for (void* p : s->allocations) (void)hipFree(p);
    if (s->dev_error_host) (void)hipHostFree(s->dev_error_host);
    delete s;
}

// body of rt_render_chunk_device on workspace `ws` (the shared one: the caller holds scene->lock)
static int render_chunk_device_on(RtScene* scene, Workspace& ws, const RtSettings* st, const RtRenderChunk* ch, uint64_t seed, void* rgb_dev, void* rgba_dev,
                                      uint32_t pitch_pixels, uint32_t out_x, uint32_t out_y, void* counters_dev, hipStream_t sm) {
    if ((uint64_t)out_x + ch->width > pitch_pixels) return set_error(RT_E_INVALID, "output pitch too small for the chunk");
    HIP_TRY(hipSetDevice(scene->device));
    const uint32_t npix = ch->width * ch->height;
    int rc = ws.accum.ensure((size_t)npix * 12);
    if (rc) return rc;
    rc = enqueue_accumulate(scene, ws, st, ch, seed, 0, st->samples_per_pixel, (float*)ws.accum.p, ch->width, 0, 0, true,
                            (unsigned long long*)counters_dev, sm);
    if (rc) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3((npix + 255) / 256), dim3(256), 0, sm, (const float*)ws.accum.p, ch->width, 0u, 0u, ch->width, npix,
                       (float)st->samples_per_pixel, (float*)rgb_dev, (uint8_t*)rgba_dev, pitch_pixels, out_x, out_y);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_render_chunk_device(RtScene* scene, const RtSettings* st, const RtRenderChunk* ch, uint64_t seed, void* rgb_dev, void* rgba_dev,
                           uint32_t pitch_pixels, uint32_t out_x, uint32_t out_y, void* counters_dev, void* stream) {
    int rc = validate_call(scene, st, ch);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(scene->lock);
    return render_chunk_device_on(scene, scene->shared, st, ch, seed, rgb_dev, rgba_dev, pitch_pixels, out_x, out_y, counters_dev, (hipStream_t)stream);
}

uint64_t rt_noalloc_max_pixels(void) {
    // control words + tile order (8 B per 64 pixels) + accumulator (12 B) + the segment sums of at least one unit (12 B)
    return ((uint64_t)RT_NOALLOC_SCRATCH_BYTES - 4096u - 1024u) / (12u + 12u + 1u) / 64u * 64u;
}

int rt_render_chunk_device_noalloc(RtScene* scene, const RtSettings* st, const RtRenderChunk* ch, uint64_t seed, void* rgb_dev, void* rgba_dev,
                                   uint32_t pitch_pixels, uint32_t out_x, uint32_t out_y, void* counters_dev, void* stream) {
    int rc = validate_call(scene, st, ch);
    if (rc) return rc;
    const uint64_t pix_slots = (((uint64_t)ch->width + 7) / 8) * (((uint64_t)ch->height + 7) / 8) * 64;      // whole 8x8 ray tiles
    if (pix_slots > rt_noalloc_max_pixels())
        return set_error(RT_E_LIMIT, "rt_render_chunk_device_noalloc: chunk of " + std::to_string(pix_slots) + " tile-padded pixels exceeds the static scratch (" +
                                     std::to_string(rt_noalloc_max_pixels()) + " pixels, rt_noalloc_max_pixels()); render it in smaller chunks");
    if (!scene->scratch_ptr) return set_error(RT_E_HIP, "no scratch address for this scene's device");
    // Carve this call's workspace out of the device's static array (all offsets multiples of 256 bytes).  Every call on a GPU
    // shares the array and finds the previous call's contents in it: the block counters are cleared per launch
    // (enqueue_accumulate), everything else is written before it is read.
    auto up = [](size_t n) { return (n + 255) & ~(size_t)255; };
    unsigned char* base = scene->scratch_ptr;
    size_t at = 0;
    Workspace ws; ws.no_alloc = true;
    auto carve = [&](DeviceBuffer& b, size_t n) { b.p = base + at; b.bytes = n; b.fixed = true; at += up(n); };
    carve(ws.ctrl, 1024 + RT_QUEUES * 64);
    carve(ws.order, order_buffer_bytes((size_t)(pix_slots / 64)));
    carve(ws.accum, (size_t)ch->width * ch->height * 12);
    carve(ws.partial, RT_NOALLOC_SCRATCH_BYTES - at);           // the rest: as many units per launch as fit
    std::lock_guard<std::mutex> g(g_scratch_call_lock[scene->device]);
    return render_chunk_device_on(scene, ws, st, ch, seed, rgb_dev, rgba_dev, pitch_pixels, out_x, out_y, counters_dev, (hipStream_t)stream);
}

int rt_render_chunk(RtScene* scene, const RtSettings* st, const RtRenderChunk* ch, uint64_t seed, float* rgb_out, uint8_t* rgba_out,
                    RtRayCounters* counters) {
    int rc = validate_call(scene, st, ch);
    if (rc) return rc;
    const size_t npix = (size_t)ch->width * ch->height;
    HIP_TRY(hipSetDevice(scene->device));
    WorkspaceLease lease(scene);          // concurrent callers get a workspace and a stream each
    if (lease.rc) return lease.rc;
    Workspace& ws = *lease.ws;
    if ((rc = ws.out_rgb.ensure(npix * 12)) || (rc = ws.out_rgba.ensure(npix * 4))) return rc;
    HIP_TRY_SYNC(hipMemsetAsync(ws.ctrl.p, 0, sizeof(RtRayCounters), ws.stream), ws.stream);
    rc = render_chunk_device_on(scene, ws, st, ch, seed, ws.out_rgb.p, ws.out_rgba.p, ch->width, 0, 0, ws.ctrl.p, ws.stream);
    if (rc) { (void)hipStreamSynchronize(ws.stream); return rc; }
    RtRayCounters c{};
    if (rgb_out) HIP_TRY_SYNC(hipMemcpyAsync(rgb_out, ws.out_rgb.p, npix * 12, hipMemcpyDeviceToHost, ws.stream), ws.stream);
    if (rgba_out) HIP_TRY_SYNC(hipMemcpyAsync(rgba_out, ws.out_rgba.p, npix * 4, hipMemcpyDeviceToHost, ws.stream), ws.stream);
    HIP_TRY_SYNC(hipMemcpyAsync(&c, ws.ctrl.p, sizeof(c), hipMemcpyDeviceToHost, ws.stream), ws.stream);      // (`c` is a local: never return with this in flight)
    HIP_TRY(hipStreamSynchronize(ws.stream));
    if ((rc = check_device_error(scene))) return rc;
    if (*scene->cancel_host() != 0u) return set_error(RT_E_CANCELLED, "rt_render_chunk: cancelled (rt_scene_cancel); the output buffers hold unfinished pixels");
    if (counters) { counters->primary += c.primary; counters->secondary += c.secondary; counters->shadow += c.shadow; counters->samples += c.samples; }
    return RT_OK;
}

// Device half of rt_scene_update_transforms (also the last step of rt_scene_rebuild_blas): entity boxes from the transform records and mesh
// boxes on the device, then the TLAS boxes bottom-up, then the root box; waits for the stream (the host needs the root box).
static int refit_tlas_on_device(RtScene* scene, hipStream_t sm) {
    Word4* xforms_dev = const_cast<Word4*>(scene->view.xforms);
    DevNode* nodes_dev = const_cast<DevNode*>(scene->view.nodes);
    DevEntity* entities_dev = const_cast<DevEntity*>(scene->view.entities);
    // device: entity boxes, then the tree bottom-up (one launch per level, deepest first), then the root box
    const uint32_t n = scene->view.entity_count;
    if (n == 0) { HIP_TRY(hipStreamSynchronize(sm)); return RT_OK; }      // an empty scene keeps the empty root box of rt_scene_create
    hipLaunchKernelGGL(refit_entities_kernel, dim3((n + 255) / 256), dim3(256), 0, sm, (const DevEntity*)entities_dev, entities_dev, (const Word4*)xforms_dev, scene->view.meshes,
                       scene->ent_box_dev, n);
    HIP_TRY_SYNC(hipGetLastError(), sm);
    if (scene->refit_valid) {
        for (size_t l = 0; l + 1 < scene->refit_level_start.size(); l++) {
            const uint32_t first = scene->refit_level_start[l], count = scene->refit_level_start[l + 1] - first;
            if (count == 0) continue;
            hipLaunchKernelGGL(refit_level_kernel, dim3((count + 255) / 256), dim3(256), 0, sm, nodes_dev, scene->refit_dev, first, count, (const float*)scene->ent_box_dev);
            HIP_TRY_SYNC(hipGetLastError(), sm);
        }
    } else if (n > RT_LBVH_LEAF) {
        // the tree was built on the device (rt_scene_rebuild_tlas): its topology and record numbers are still in the build's buffers, so a refit
        // is the build without the sort: boxes bottom-up, records rewritten
        RtScene::TlasRebuild& rb = scene->rebuild;
        const dim3 blk(256), per_entity((n + 255) / 256), per_inner((n - 1 + 255) / 256);
        HIP_TRY_SYNC(hipMemsetAsync(rb.arrived, 0, (size_t)(n - 1) * 4, sm), sm);
        hipLaunchKernelGGL(lbvh_boxes_kernel, per_entity, blk, 0, sm, (const LbvhNode*)rb.lnodes, (const uint32_t*)rb.parent, n, (const float*)scene->ent_box_dev, rb.box,
                           rb.height, rb.arrived);
        hipLaunchKernelGGL(lbvh_emit_kernel, per_inner, blk, 0, sm, (const LbvhNode*)rb.lnodes, n, (const uint32_t*)rb.number, (const float*)rb.box,
                           (const float*)scene->ent_box_dev, nodes_dev);
        HIP_TRY_SYNC(hipGetLastError(), sm);
    }
    float* root_dev = scene->ent_box_dev + (size_t)n * 8;
    hipLaunchKernelGGL(refit_root_kernel, dim3(1), dim3(1), 0, sm, (const DevNode*)nodes_dev, scene->view.tlas_root_ref, n, (const float*)scene->ent_box_dev, root_dev);
    HIP_TRY_SYNC(hipGetLastError(), sm);
    float root[6];
    HIP_TRY_SYNC(hipMemcpyAsync(root, root_dev, sizeof(root), hipMemcpyDeviceToHost, sm), sm);
    HIP_TRY(hipStreamSynchronize(sm));
    for (int k = 0; k < 3; k++) { scene->view.tlas_min[k] = root[k]; scene->view.tlas_max[k] = root[3 + k]; }
    scene->view.loose_boxes = 0u;      // (transforms with a non-unit rotation axis are refused before they reach the device)
    scene->view.coc_rel_max = host_coc_rel_max(scene->camera, scene->view.tlas_min, scene->view.tlas_max, n > 0);
    return RT_OK;
}

// One re-allocation of the node array with room behind the scene's records for every tree a rebuild can emit: n_entities - 1 records for the TLAS
// (rt_scene_rebuild_tlas), then tri_count - 1 per mesh (rt_scene_rebuild_blas).  (The old array stays allocated until the scene is destroyed:
// launches in flight may still read it.)
static size_t rebuilt_tlas_room(const RtScene* scene) { return scene->view.entity_count > 1 ? scene->view.entity_count - 1 : 1; }
static size_t rebuilt_blas_base(const RtScene* scene, size_t mesh) {
    size_t at = scene->node_count + rebuilt_tlas_room(scene);
    for (size_t m = 0; m < mesh; m++) at += scene->meshes_host[m].w[0].u[2] > 1 ? scene->meshes_host[m].w[0].u[2] - 1 : 1;
    return at;
}
static int ensure_node_room(RtScene* scene, hipStream_t sm) {
    if (scene->node_room) return RT_OK;
    const size_t total = rebuilt_blas_base(scene, scene->meshes_host.size());
    if (total >= ((size_t)1 << 26)) return set_error(RT_E_LIMIT, "rebuilt trees would not fit the 2^26 records the walk can address");
    DevNode* bigger = nullptr;
    HIP_TRY(hipMalloc((void**)&bigger, total * sizeof(DevNode)));
    scene->allocations.push_back(bigger);
    HIP_TRY(hipMemcpyAsync(bigger, scene->view.nodes, scene->node_count * sizeof(DevNode), hipMemcpyDeviceToDevice, sm));
    HIP_TRY(hipStreamSynchronize(sm));
    scene->view.nodes = bigger;           // (same contents)
    scene->node_room = true;
    return RT_OK;
}

int rt_scene_update_transforms(RtScene* scene, const RtV3* translations, uint64_t translation_count, const RtRotation* rotations,
                               uint64_t rotation_count, void* stream) {
    if (!scene) return set_error(RT_E_INVALID, "null scene");
    if (scene->device == RT_DEVICE_HOST_ONLY) return set_error(RT_E_NO_DEVICE, "scene was created host-only");
    if (translation_count != scene->n_translations || rotation_count != scene->n_rotations)
        return set_error(RT_E_INVALID, "rt_scene_update_transforms: the arrays must have the sizes given to rt_scene_create (entities keep their ranges)");
    if ((translation_count && !translations) || (rotation_count && !rotations)) return set_error(RT_E_INVALID, "null transform array");
    std::lock_guard<std::mutex> g(scene->lock);
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t sm = (hipStream_t)stream;
    // host: transform records with sin / cos from the host libm, as at creation (types.rs:200 evaluates them per ray)
    bool loose = false;
    host_make_xforms(scene->desc_entities.data(), scene->tlas_perm, translations, rotations, scene->xforms_host, loose);
    if (scene->xforms_host.size() != scene->xforms_words) return set_error(RT_E_INVALID, "rt_scene_update_transforms: internal size mismatch");
    // A rotation about an axis that is not of unit length scales the entity: its box does not bound its hits, and which hits the reference reports then
    // depends on the LEAVES of the tree it builds for the moved entities - which a refitted tree does not have (measured: 12 of 20 such scenes differ
    // from a fresh rt_scene_create in 1-57 pixels).  Nothing on the device has been touched yet.
    if (loose) return set_error(RT_E_INVALID, "rt_scene_update_transforms: a rotation axis is not of unit length - entity boxes do not bound their hits and only the tree "
                                              "the reference builds for these transforms reproduces them; use rt_scene_create");
    if (!scene->refit_valid && scene->rebuild.topology_stale)
        return set_error(RT_E_INVALID, "rt_scene_update_transforms: the last rt_scene_rebuild_tlas on this scene was refused (RT_E_DEPTH) and left its working buffers describing "
                                       "another tree than the scene's; call rt_scene_rebuild_tlas again (or create the scene anew)");
    Word4* xforms_dev = const_cast<Word4*>(scene->view.xforms);
    HIP_TRY(hipMemcpyAsync(xforms_dev, scene->xforms_host.data(), scene->xforms_host.size() * sizeof(Word4), hipMemcpyHostToDevice, sm));
    return refit_tlas_on_device(scene, sm);
}

// A NEW top-level tree over the moved entities, built on the device (tlas_build.h).
int rt_scene_rebuild_tlas(RtScene* scene, const RtV3* translations, uint64_t translation_count, const RtRotation* rotations,
                          uint64_t rotation_count, void* stream) {
    if (!scene) return set_error(RT_E_INVALID, "null scene");
    if (scene->device == RT_DEVICE_HOST_ONLY) return set_error(RT_E_NO_DEVICE, "scene was created host-only");
    if (translation_count != scene->n_translations || rotation_count != scene->n_rotations)
        return set_error(RT_E_INVALID, "rt_scene_rebuild_tlas: the arrays must have the sizes given to rt_scene_create (entities keep their ranges)");
    if ((translation_count && !translations) || (rotation_count && !rotations)) return set_error(RT_E_INVALID, "null transform array");
    std::lock_guard<std::mutex> g(scene->lock);
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t sm = (hipStream_t)stream;
    const uint32_t n = scene->view.entity_count;
    if (n == 0) return RT_OK;                 // an empty scene keeps the empty root box and the never-followed root ref of rt_scene_create
    RtScene::TlasRebuild& rb = scene->rebuild;
    const size_t m = n > 1 ? n - 1 : 1;
    if (!rb.ready) {
        // first call: the build's buffers (about 250 bytes per entity) and a node array with room for n - 1 more records
        auto dev_alloc = [&](void** p, size_t bytes) -> int {
            HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
            scene->allocations.push_back(*p);
            return RT_OK;
        };
        int rc;
        if ((rc = dev_alloc((void**)&rb.keys_a, (size_t)(n + 1) * 8)) || (rc = dev_alloc((void**)&rb.keys_b, (size_t)(n + 1) * 8)) ||
            (rc = dev_alloc((void**)&rb.lnodes, m * sizeof(LbvhNode))) || (rc = dev_alloc((void**)&rb.parent, (size_t)(2 * (size_t)n + 1) * 4)) ||
            (rc = dev_alloc((void**)&rb.height, m * 4)) || (rc = dev_alloc((void**)&rb.arrived, m * 4)) || (rc = dev_alloc((void**)&rb.number, m * 4)) ||
            (rc = dev_alloc((void**)&rb.queue, ((size_t)scene->tlas_hot_records + 1) * 4)) || (rc = dev_alloc((void**)&rb.box, m * 32)) ||
            (rc = dev_alloc((void**)&rb.bounds, 64)) || (rc = dev_alloc((void**)&rb.entities_alt, (size_t)(n + 1) * sizeof(DevEntity))) ||
            (rc = dev_alloc((void**)&rb.ent_box_alt, ((size_t)n + 1) * 32)) || (rc = dev_alloc((void**)&rb.entities_stage, (size_t)(n + 1) * sizeof(DevEntity))) ||
            (rc = dev_alloc((void**)&rb.ent_box_stage, ((size_t)n + 1) * 32)) || (rc = dev_alloc((void**)&rb.xforms_stage, (scene->xforms_words + 1) * sizeof(Word4))))
            return rc;
        rb.sort_temp_bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, rb.sort_temp_bytes, rb.keys_a, rb.keys_b, (int)n, 0, 62, sm));
        if ((rc = dev_alloc(&rb.sort_temp, rb.sort_temp_bytes))) return rc;
        if ((rc = ensure_node_room(scene, sm))) return rc;
        rb.ready = true;
    }
    // host: transform records, as in rt_scene_update_transforms
    bool loose = false;
    host_make_xforms(scene->desc_entities.data(), scene->tlas_perm, translations, rotations, scene->xforms_host, loose);
    if (scene->xforms_host.size() != scene->xforms_words) return set_error(RT_E_INVALID, "rt_scene_rebuild_tlas: internal size mismatch");
    // A rotation about an axis that is not of unit length scales the entity (types.rs:200), so its box (calculate_origin_aabb) does not bound its
    // hits and WHICH hits the reference reports depends on the boxes of its own tree (DevSceneView::loose_boxes): no other tree reproduces them
    // (measured: 62 of 323 such random scenes differ).  Nothing on the device has been touched yet.
    if (loose) return set_error(RT_E_INVALID, "rt_scene_rebuild_tlas: a rotation axis is not of unit length - entity boxes do not bound their hits, only the reference's own tree "
                                              "reproduces them; use rt_scene_update_transforms (refit) or rt_scene_create");
    // ---- phase 1: everything up to the height of the new tree goes into the build's OWN buffers.  The scene a render call sees (transform records,
    // entity records and boxes, node records, view.*) is untouched until the new tree is known to fit the traversal stack, so a refusal leaves the
    // scene exactly as it was.
    DevNode* nodes_dev = const_cast<DevNode*>(scene->view.nodes);
    const DevEntity* ent_live = scene->view.entities;
    DevEntity* ent_new = rb.entities_alt; float* box_new = rb.ent_box_alt;      // key order: what the scene will use
    const uint32_t hot = scene->tlas_hot_records, append_base = (uint32_t)scene->node_count;
    const dim3 blk(256), per_entity((n + 255) / 256), per_inner((n > 1 ? n - 1 + 255 : 255) / 256);
    uint32_t height0 = 0;
    HIP_TRY(hipMemcpyAsync(rb.xforms_stage, scene->xforms_host.data(), scene->xforms_host.size() * sizeof(Word4), hipMemcpyHostToDevice, sm));
    hipLaunchKernelGGL(refit_entities_kernel, per_entity, blk, 0, sm, ent_live, rb.entities_stage, (const Word4*)rb.xforms_stage, scene->view.meshes, rb.ent_box_stage, n);
    hipLaunchKernelGGL(lbvh_bounds_kernel, dim3(1), blk, 0, sm, (const float*)rb.ent_box_stage, n, rb.bounds);
    hipLaunchKernelGGL(lbvh_keys_kernel, per_entity, blk, 0, sm, (const float*)rb.ent_box_stage, n, (const float*)rb.bounds, rb.keys_a);
    HIP_TRY_SYNC(hipGetLastError(), sm);
    size_t temp_bytes = rb.sort_temp_bytes;
    HIP_TRY_SYNC(hipcub::DeviceRadixSort::SortKeys(rb.sort_temp, temp_bytes, rb.keys_a, rb.keys_b, (int)n, 0, 62, sm), sm);
    hipLaunchKernelGGL(lbvh_permute_kernel, per_entity, blk, 0, sm, (const unsigned long long*)rb.keys_b, n, (const DevEntity*)rb.entities_stage, ent_new,
                       (const float*)rb.ent_box_stage, box_new, (uint32_t*)nullptr);
    uint32_t root_ref = RT_REF_LEAF_BIT | ((n - 1u) << RT_REF_COUNT_SHIFT);      // all entities in one leaf ...
    if (n > RT_LBVH_LEAF) {                                                        // ... or a tree
        HIP_TRY_SYNC(hipMemsetAsync(rb.arrived, 0, (size_t)(n - 1) * 4, sm), sm);
        hipLaunchKernelGGL(lbvh_topology_kernel, per_inner, blk, 0, sm, (const unsigned long long*)rb.keys_b, n, rb.lnodes, rb.parent);
        hipLaunchKernelGGL(lbvh_boxes_kernel, per_entity, blk, 0, sm, (const LbvhNode*)rb.lnodes, (const uint32_t*)rb.parent, n, (const float*)box_new, rb.box,
                           rb.height, rb.arrived);
        hipLaunchKernelGGL(lbvh_number_init_kernel, per_inner, blk, 0, sm, rb.number, n - 1, append_base);
        hipLaunchKernelGGL(lbvh_number_kernel, dim3(1), dim3(64), 0, sm, (const LbvhNode*)rb.lnodes, n, hot, rb.number, rb.queue);
        HIP_TRY_SYNC(hipGetLastError(), sm);
        root_ref = hot > 0 ? 0u : append_base;      // (lbvh_number_kernel: the root is the first record in breadth-first order)
        HIP_TRY_SYNC(hipMemcpyAsync(&height0, rb.height, 4, hipMemcpyDeviceToHost, sm), sm);
    }
    HIP_TRY(hipStreamSynchronize(sm));
    // traversal stack for the new depth (scene_build.cpp's bound), and what that leaves the node cache
    const uint32_t stack_entries = height0 + 1u + scene->blas_walk_depth + 2u;
    const size_t stack_bytes = ((size_t)stack_entries + (scene->view.mesh_entities ? 3 : 0)) * RT_BLOCK * sizeof(uint32_t);
    uint32_t debug_limit;
    { std::lock_guard<std::mutex> og(scene->opt_lock); debug_limit = scene->opt.debug_rebuild_stack_limit; }
    if (stack_bytes > 64 * 1024 || (debug_limit && stack_entries > debug_limit)) {
        // (a refit of the device-built tree would now find another topology in the build's buffers than the records describe: the next refit must be a rebuild)
        if (!scene->refit_valid) rb.topology_stale = true;
        return set_error(RT_E_DEPTH, "rt_scene_rebuild_tlas: the tree over these entities is too deep for the LDS traversal stack (more than 64 levels); the scene is unchanged "
                                     "(it keeps its previous transforms and tree)");
    }
    // ---- phase 2: commit.  From the first of these writes on the old scene is gone; a HIP failure in here (none is expected: the same launches and copies have
    // succeeded in phase 1) poisons the scene instead of leaving a half-written one in use.
    auto poisoned = [&](const char* what, hipError_t e) -> int {
        (void)hipStreamSynchronize(sm);
        if (scene->dev_error_host) *(volatile uint32_t*)scene->dev_error_host |= RT_DEVERR_POISONED;
        return set_error(RT_E_HIP, std::string("rt_scene_rebuild_tlas: ") + what + ": " + hipGetErrorString(e) + " - the scene is unusable, create it anew");
    };
    hipError_t he;
    if ((he = hipMemcpyAsync(const_cast<Word4*>(scene->view.xforms), rb.xforms_stage, scene->xforms_host.size() * sizeof(Word4), hipMemcpyDeviceToDevice, sm)) != hipSuccess)
        return poisoned("copy of the transform records", he);
    if (n > RT_LBVH_LEAF) {
        hipLaunchKernelGGL(lbvh_emit_kernel, per_inner, blk, 0, sm, (const LbvhNode*)rb.lnodes, n, (const uint32_t*)rb.number, (const float*)rb.box,
                           (const float*)box_new, nodes_dev);
        if ((he = hipGetLastError()) != hipSuccess) return poisoned("lbvh_emit_kernel", he);
    }
    float* root_dev = box_new + (size_t)n * 8;
    hipLaunchKernelGGL(refit_root_kernel, dim3(1), dim3(1), 0, sm, (const DevNode*)nodes_dev, root_ref, n, (const float*)box_new, root_dev);
    if ((he = hipGetLastError()) != hipSuccess) return poisoned("refit_root_kernel", he);
    float root[6];
    if ((he = hipMemcpyAsync(root, root_dev, sizeof(root), hipMemcpyDeviceToHost, sm)) != hipSuccess) return poisoned("copy of the root box", he);
    if ((he = hipStreamSynchronize(sm)) != hipSuccess) return poisoned("hipStreamSynchronize", he);
    // the scene now IS the new tree (entity arrays in key order, TLAS cache slots and appended records rewritten); the arrays it used become the next build's output
    rb.entities_alt = const_cast<DevEntity*>(scene->view.entities);
    rb.ent_box_alt = scene->ent_box_dev;
    scene->view.entities = ent_new;
    scene->view.ent_box = reinterpret_cast<const Word4*>(box_new);
    scene->ent_box_dev = box_new;
    scene->refit_valid = false;               // the refit tables of rt_scene_create describe a tree that is gone
    rb.topology_stale = false;
    scene->view.tlas_root_ref = root_ref;
    for (int k = 0; k < 3; k++) { scene->view.tlas_min[k] = root[k]; scene->view.tlas_max[k] = root[3 + k]; }
    scene->view.loose_boxes = 0u;
    scene->view.coc_rel_max = host_coc_rel_max(scene->camera, scene->view.tlas_min, scene->view.tlas_max, n > 0);
    scene->view.stack_entries = stack_entries;
    scene->tlas_walk_depth = height0;
    const size_t block_budget = (size_t)160 * 1024 * RT_BLOCK / (256 * RT_WAVES_PER_SIMD);
    const size_t room = stack_bytes < block_budget ? (block_budget - stack_bytes) / RT_NODE_CACHE_STRIDE : 0;
    scene->view.lds_nodes = (uint32_t)std::min<size_t>(scene->lds_nodes_built, room);
    scene->stack_lds_bytes = stack_bytes;
    scene->lds_bytes = stack_bytes + (size_t)scene->view.lds_nodes * RT_NODE_CACHE_STRIDE;
    scene->lds_bytes += size_sphere_cache(scene->view, scene->lds_bytes);
    scene->info.tlas_depth = height0;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_kernel<true, true, true, true, true, true, true>, RT_BLOCK, scene->lds_bytes) == hipSuccess && per_cu >= 1) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, scene->device) == hipSuccess) scene->grid_blocks = prop.multiProcessorCount * per_cu;
    }
    return RT_OK;
}

uint32_t rt_sample_segment(void) { return RT_SAMPLE_SEGMENT; }

int rt_strip_rows(uint32_t height, uint32_t strip_height, uint32_t strip_first, uint32_t strip_period) {
    if (strip_height == 0 || strip_period == 0 || strip_first >= strip_period) return RT_E_INVALID;
    uint64_t rows = 0;
    for (uint64_t top = (uint64_t)strip_first * strip_height; top < height; top += (uint64_t)strip_period * strip_height)
        rows += (height - top < strip_height) ? height - top : strip_height;
    return (int)rows;
}

// Body of rt_render_strips_device.  interleave_dst: write every strip to its own rows of a full frame of pitch st->width (which may live on another
// device with peer access) instead of packing the strips.
static int render_strips_on(RtScene* scene, const RtSettings* st, uint32_t strip_height, uint32_t strip_first, uint32_t strip_period,
                            uint64_t seed, void* rgb_dev, void* rgba_dev, bool interleave_dst, void* counters_dev, hipStream_t sm) {
    if (!st) return set_error(RT_E_INVALID, "null argument");
    const int rows = rt_strip_rows(st->height, strip_height, strip_first, strip_period);
    if (rows < 0) return set_error(RT_E_INVALID, "rt_render_strips_device: strip_height and strip_period must be > 0, strip_first < strip_period");
    // the kernel maps a whole 8x8 ray tile through the strip interleave by its first row: a strip must hold whole tile rows
    if (strip_height % 8u != 0u) return set_error(RT_E_INVALID, "rt_render_strips_device: strip_height must be a multiple of 8 (the kernel's 8x8 ray tiles)");
    if (rows == 0) return RT_OK;                         // this rank owns no rows of so small an image
    RtRenderChunk local{0, 0, st->width, (uint32_t)rows};   // packed local rows; validated against a frame that tall
    RtSettings probe = *st; probe.height = probe.height > (uint32_t)rows ? probe.height : (uint32_t)rows;
    int rc = validate_call(scene, &probe, &local);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(scene->lock);
    HIP_TRY(hipSetDevice(scene->device));
    const uint32_t npix = local.width * local.height;
    Workspace& ws = scene->shared;
    if ((rc = ws.accum.ensure((size_t)npix * 12))) return rc;
    Interleave il; il.strip_h = strip_height; il.period_rows = strip_period * strip_height; il.first_row = strip_first * strip_height;
    rc = enqueue_accumulate(scene, ws, st, &local, seed, 0, st->samples_per_pixel, (float*)ws.accum.p, local.width, 0, 0, true,
                            (unsigned long long*)counters_dev, sm, il);
    if (rc) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3((npix + 255) / 256), dim3(256), 0, sm, (const float*)ws.accum.p, local.width, 0u, 0u, local.width, npix,
                       (float)st->samples_per_pixel, (float*)rgb_dev, (uint8_t*)rgba_dev, local.width, 0u, 0u,
                       interleave_dst ? il.strip_h : 0u, il.period_rows, il.first_row);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_render_strips_device(RtScene* scene, const RtSettings* st, uint32_t strip_height, uint32_t strip_first, uint32_t strip_period,
                            uint64_t seed, void* rgb_dev, void* rgba_dev, void* counters_dev, void* stream) {
    return render_strips_on(scene, st, strip_height, strip_first, strip_period, seed, rgb_dev, rgba_dev, false, counters_dev, (hipStream_t)stream);
}

// ---- dealt tile columns: the finer partition of a frame over N GPUs ---------------------------------------------------------------------
// Rank r of N owns the 8-pixel-wide tile columns r, r + N, r + 2N, ... over the whole height of the frame: every rank gets a share of every
// region of the image, where 8-row strips cut a glass object unevenly (per-rank kernel times of the 8-rank split, emulated on one GPU:
// DESIGN.md section 6).  Packed layout: `height` rows of rt_tile_columns_width() pixels, tile column j of the rank at packed columns 8j..8j+7.
int rt_tile_columns_width(uint32_t width, uint32_t col_first, uint32_t col_period) {
    if (col_period == 0 || col_first >= col_period) return RT_E_INVALID;
    const uint64_t tiles_x = ((uint64_t)width + 7) / 8;
    if (col_first >= tiles_x) return 0;
    const uint64_t cols = (tiles_x - col_first + col_period - 1) / col_period;
    uint64_t w = cols * 8;
    if (col_first + (cols - 1) * col_period == tiles_x - 1) w -= tiles_x * 8 - width;      // the frame's last tile column may be narrower
    return (int)w;
}

// Body of rt_render_tile_columns_device.  interleave_dst: write every column to its place in a full frame of pitch st->width (which may live on
// another device with peer access) instead of packing the columns.
static int render_columns_on(RtScene* scene, const RtSettings* st, uint32_t col_first, uint32_t col_period, uint64_t seed, void* rgb_dev, void* rgba_dev,
                             bool interleave_dst, void* counters_dev, hipStream_t sm) {
    if (!st) return set_error(RT_E_INVALID, "null argument");
    const int pw = rt_tile_columns_width(st->width, col_first, col_period);
    if (pw < 0) return set_error(RT_E_INVALID, "rt_render_tile_columns_device: col_period must be > 0, col_first < col_period");
    if (pw == 0) return RT_OK;                           // this rank owns no column of so narrow an image
    RtRenderChunk local{0, 0, (uint32_t)pw, st->height};   // the packed columns
    int rc = validate_call(scene, st, &local);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(scene->lock);
    HIP_TRY(hipSetDevice(scene->device));
    const uint32_t npix = local.width * local.height;
    Workspace& ws = scene->shared;
    if ((rc = ws.accum.ensure((size_t)npix * 12))) return rc;
    Interleave il; il.col_n = col_period; il.col_r = col_first;
    rc = enqueue_accumulate(scene, ws, st, &local, seed, 0, st->samples_per_pixel, (float*)ws.accum.p, local.width, 0, 0, true,
                            (unsigned long long*)counters_dev, sm, il);
    if (rc) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3((npix + 255) / 256), dim3(256), 0, sm, (const float*)ws.accum.p, local.width, 0u, 0u, local.width, npix,
                       (float)st->samples_per_pixel, (float*)rgb_dev, (uint8_t*)rgba_dev, interleave_dst ? st->width : local.width, 0u, 0u,
                       0u, 0u, 0u, interleave_dst ? il.col_n : 0u, il.col_r);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_render_tile_columns_device(RtScene* scene, const RtSettings* st, uint32_t col_first, uint32_t col_period, uint64_t seed, void* rgb_dev, void* rgba_dev,
                                  void* counters_dev, void* stream) {
    return render_columns_on(scene, st, col_first, col_period, seed, rgb_dev, rgba_dev, false, counters_dev, (hipStream_t)stream);
}

int rt_accumulate_chunk_device(RtScene* scene, const RtSettings* st, const RtRenderChunk* ch, uint64_t seed, uint32_t sample_first,
                               uint32_t sample_count, void* accum_dev, void* counters_dev, void* stream) {
    int rc = validate_call(scene, st, ch);
    if (rc) return rc;
    if (!accum_dev) return set_error(RT_E_INVALID, "null accumulator");
    std::lock_guard<std::mutex> g(scene->lock);
    HIP_TRY(hipSetDevice(scene->device));
    return enqueue_accumulate(scene, scene->shared, st, ch, seed, sample_first, sample_count, (float*)accum_dev, st->width, ch->left, ch->top, false,
                              (unsigned long long*)counters_dev, (hipStream_t)stream);
}

// A NEW tree for ONE mesh whose triangles have moved, built on the device (tlas_build.h), then the TLAS boxes above its instances refitted.
int rt_scene_rebuild_blas(RtScene* scene, uint32_t mesh_index, const RtMeshTri* tris, uint64_t tri_count, void* stream) {
    if (!scene || !tris) return set_error(RT_E_INVALID, "rt_scene_rebuild_blas: null argument");
    if (scene->device == RT_DEVICE_HOST_ONLY) return set_error(RT_E_NO_DEVICE, "scene was created host-only");
    if (mesh_index >= scene->meshes_host.size()) return set_error(RT_E_INVALID, "rt_scene_rebuild_blas: no such mesh");
    std::lock_guard<std::mutex> g(scene->lock);
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t sm = (hipStream_t)stream;
    DevMesh& mh = scene->meshes_host[mesh_index];
    const uint32_t n = mh.w[0].u[2], tri_base = mh.w[0].u[1];
    if (tri_count != n) return set_error(RT_E_INVALID, "rt_scene_rebuild_blas: the mesh keeps its triangle count (" + std::to_string(n) + "); a mesh of another size is another scene");
    if (scene->view.loose_boxes) return set_error(RT_E_INVALID, "rt_scene_rebuild_blas: the scene has entities whose boxes do not bound their hits (non-unit rotation axes): only the reference's own trees reproduce them");
    if (!scene->refit_valid && scene->rebuild.topology_stale)
        return set_error(RT_E_INVALID, "rt_scene_rebuild_blas: the last rt_scene_rebuild_tlas on this scene was refused; call it again first (the TLAS refit that follows a mesh rebuild needs its buffers)");
    RtScene::BlasRebuild& rb = scene->blas_rebuild[mesh_index];
    const size_t m = n > 1 ? n - 1 : 1;
    const uint32_t hot = scene->blas_hot_count[mesh_index], hot_base = scene->blas_hot_base[mesh_index];
    int rc;
    if (!rb.ready) {
        // first call for this mesh: the build's buffers (about 330 bytes per triangle), and - once per scene - a node array with room for the new records
        auto dev_alloc = [&](void** p, size_t bytes) -> int {
            HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
            scene->allocations.push_back(*p);
            return RT_OK;
        };
        if ((rc = dev_alloc((void**)&rb.tris, (size_t)n * sizeof(RtMeshTri))) || (rc = dev_alloc((void**)&rb.box_in, ((size_t)n + 1) * 32)) ||
            (rc = dev_alloc((void**)&rb.box_sorted, ((size_t)n + 1) * 32)) || (rc = dev_alloc((void**)&rb.ibox, m * 32)) || (rc = dev_alloc((void**)&rb.bounds, 64)) ||
            (rc = dev_alloc((void**)&rb.root, 64)) || (rc = dev_alloc((void**)&rb.keys_a, ((size_t)n + 1) * 8)) || (rc = dev_alloc((void**)&rb.keys_b, ((size_t)n + 1) * 8)) ||
            (rc = dev_alloc((void**)&rb.lnodes, m * sizeof(LbvhNode))) || (rc = dev_alloc((void**)&rb.parent, (2 * (size_t)n + 1) * 4)) ||
            (rc = dev_alloc((void**)&rb.height, m * 4)) || (rc = dev_alloc((void**)&rb.arrived, m * 4)) || (rc = dev_alloc((void**)&rb.number, m * 4)) ||
            (rc = dev_alloc((void**)&rb.queue, ((size_t)hot + 1) * 4)) || (rc = dev_alloc((void**)&rb.where, (size_t)n * 4)) ||
            (rc = dev_alloc((void**)&rb.rank_stage, (size_t)n * 4)) || (rc = dev_alloc((void**)&rb.hot_stage, (size_t)n * sizeof(DevTriHot))) ||
            (rc = dev_alloc((void**)&rb.cold_stage, (size_t)n * sizeof(DevTriCold))))
            return rc;
        rb.sort_temp_bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, rb.sort_temp_bytes, rb.keys_a, rb.keys_b, (int)n, 0, 62, sm));
        if ((rc = dev_alloc(&rb.sort_temp, rb.sort_temp_bytes))) return rc;
        if ((rc = ensure_node_room(scene, sm))) return rc;
        rb.ready = true;
    }
    // ---- phase 1: the new triangle records, topology and boxes, all in the build's own buffers (nothing a render call reads is written) ----
    DevNode* nodes_dev = const_cast<DevNode*>(scene->view.nodes);
    DevTriHot* hot_live = const_cast<DevTriHot*>(scene->view.tri_hot) + tri_base;
    DevTriCold* cold_live = const_cast<DevTriCold*>(scene->view.tri_cold) + tri_base;
    uint32_t* rank_live = const_cast<uint32_t*>(scene->view.tri_rank) + tri_base;
    const uint32_t append_base = (uint32_t)rebuilt_blas_base(scene, mesh_index);
    const dim3 blk(256), per_tri((n + 255) / 256), per_inner((n > 1 ? n - 1 + 255 : 255) / 256);
    uint32_t height0 = 0;
    HIP_TRY(hipMemcpyAsync(rb.tris, tris, (size_t)n * sizeof(RtMeshTri), hipMemcpyHostToDevice, sm));
    hipLaunchKernelGGL(blas_tri_boxes_kernel, per_tri, blk, 0, sm, (const RtMeshTri*)rb.tris, n, rb.box_in);
    hipLaunchKernelGGL(lbvh_bounds_kernel, dim3(1), blk, 0, sm, (const float*)rb.box_in, n, rb.bounds);
    hipLaunchKernelGGL(lbvh_keys_kernel, per_tri, blk, 0, sm, (const float*)rb.box_in, n, (const float*)rb.bounds, rb.keys_a);
    hipLaunchKernelGGL(blas_where_kernel, per_tri, blk, 0, sm, (const DevTriCold*)cold_live, n, rb.where);
    HIP_TRY_SYNC(hipGetLastError(), sm);
    size_t temp_bytes = rb.sort_temp_bytes;
    HIP_TRY_SYNC(hipcub::DeviceRadixSort::SortKeys(rb.sort_temp, temp_bytes, rb.keys_a, rb.keys_b, (int)n, 0, 62, sm), sm);
    hipLaunchKernelGGL(blas_permute_kernel, per_tri, blk, 0, sm, (const unsigned long long*)rb.keys_b, n, (const RtMeshTri*)rb.tris, (const float*)rb.box_in, (const uint32_t*)rb.where,
                       (const uint32_t*)rank_live, rb.hot_stage, rb.cold_stage, rb.rank_stage, rb.box_sorted);
    uint32_t root_ref = RT_REF_LEAF_BIT | ((n - 1u) << RT_REF_COUNT_SHIFT);      // every triangle in one leaf ...
    if (n > RT_LBVH_LEAF) {                                                        // ... or a tree
        HIP_TRY_SYNC(hipMemsetAsync(rb.arrived, 0, (size_t)(n - 1) * 4, sm), sm);
        hipLaunchKernelGGL(lbvh_topology_kernel, per_inner, blk, 0, sm, (const unsigned long long*)rb.keys_b, n, rb.lnodes, rb.parent);
        hipLaunchKernelGGL(lbvh_boxes_kernel, per_tri, blk, 0, sm, (const LbvhNode*)rb.lnodes, (const uint32_t*)rb.parent, n, (const float*)rb.box_sorted, rb.ibox, rb.height,
                           rb.arrived);
        hipLaunchKernelGGL(lbvh_number_init_kernel, per_inner, blk, 0, sm, rb.number, n - 1, append_base);
        hipLaunchKernelGGL(lbvh_number_kernel, dim3(1), dim3(64), 0, sm, (const LbvhNode*)rb.lnodes, n, hot, rb.number, rb.queue, hot_base);
        root_ref = hot > 0 ? hot_base : append_base;
        HIP_TRY_SYNC(hipMemcpyAsync(&height0, rb.height, 4, hipMemcpyDeviceToHost, sm), sm);
    }
    hipLaunchKernelGGL(blas_root_kernel, dim3(1), dim3(1), 0, sm, (const float*)rb.ibox, (const float*)rb.box_sorted, n, rb.root);
    HIP_TRY_SYNC(hipGetLastError(), sm);
    float root[6];
    HIP_TRY_SYNC(hipMemcpyAsync(root, rb.root, sizeof(root), hipMemcpyDeviceToHost, sm), sm);
    HIP_TRY(hipStreamSynchronize(sm));
    // the traversal stack the scene needs with this tree in it
    uint32_t blas_depth = height0;
    for (size_t k = 0; k < scene->blas_walk_depths.size(); k++) if (k != mesh_index && scene->blas_walk_depths[k] > blas_depth) blas_depth = scene->blas_walk_depths[k];
    const uint32_t stack_entries = scene->tlas_walk_depth + 1u + blas_depth + 2u;
    const size_t stack_bytes = ((size_t)stack_entries + (scene->view.mesh_entities ? 3 : 0)) * RT_BLOCK * sizeof(uint32_t);
    uint32_t debug_limit;
    { std::lock_guard<std::mutex> og(scene->opt_lock); debug_limit = scene->opt.debug_rebuild_stack_limit; }
    if (stack_bytes > 64 * 1024 || (debug_limit && stack_entries > debug_limit))
        return set_error(RT_E_DEPTH, "rt_scene_rebuild_blas: the tree over these triangles is too deep for the LDS traversal stack (more than 64 levels); the scene is unchanged");
    // ---- phase 2: commit (a HIP failure from here on poisons the scene: it would be half old, half new) ----
    auto poisoned = [&](const char* what, hipError_t e) -> int {
        (void)hipStreamSynchronize(sm);
        if (scene->dev_error_host) *(volatile uint32_t*)scene->dev_error_host |= RT_DEVERR_POISONED;
        return set_error(RT_E_HIP, std::string("rt_scene_rebuild_blas: ") + what + ": " + hipGetErrorString(e) + " - the scene is unusable, create it anew");
    };
    hipError_t he;
    if ((he = hipMemcpyAsync(hot_live, rb.hot_stage, (size_t)n * sizeof(DevTriHot), hipMemcpyDeviceToDevice, sm)) != hipSuccess) return poisoned("copy of the triangle records", he);
    if ((he = hipMemcpyAsync(cold_live, rb.cold_stage, (size_t)n * sizeof(DevTriCold), hipMemcpyDeviceToDevice, sm)) != hipSuccess) return poisoned("copy of the triangle records", he);
    if ((he = hipMemcpyAsync(rank_live, rb.rank_stage, (size_t)n * 4, hipMemcpyDeviceToDevice, sm)) != hipSuccess) return poisoned("copy of the visit ranks", he);
    if (n > RT_LBVH_LEAF) {
        hipLaunchKernelGGL(lbvh_emit_kernel, per_inner, blk, 0, sm, (const LbvhNode*)rb.lnodes, n, (const uint32_t*)rb.number, (const float*)rb.ibox, (const float*)rb.box_sorted,
                           nodes_dev);
        if ((he = hipGetLastError()) != hipSuccess) return poisoned("lbvh_emit_kernel", he);
    }
    DevMesh header = mh;
    header.w[0].u[3] = root_ref;
    header.w[1].f[0] = root[0]; header.w[1].f[1] = root[1]; header.w[1].f[2] = root[2]; header.w[1].f[3] = root[3];
    header.w[2].f[0] = root[4]; header.w[2].f[1] = root[5];
    if ((he = hipMemcpyAsync(const_cast<DevMesh*>(scene->view.meshes) + mesh_index, &header, sizeof(DevMesh), hipMemcpyHostToDevice, sm)) != hipSuccess)
        return poisoned("copy of the mesh header", he);
    if ((he = hipStreamSynchronize(sm)) != hipSuccess) return poisoned("hipStreamSynchronize", he);      // (`header` is a local)
    mh = header;
    scene->blas_walk_depths[mesh_index] = height0;
    scene->blas_walk_depth = blas_depth;
    scene->info.blas_depth = std::max(scene->info.blas_depth, height0);
    scene->view.stack_entries = stack_entries;
    const size_t block_budget = (size_t)160 * 1024 * RT_BLOCK / (256 * RT_WAVES_PER_SIMD);
    const size_t room = stack_bytes < block_budget ? (block_budget - stack_bytes) / RT_NODE_CACHE_STRIDE : 0;
    scene->view.lds_nodes = (uint32_t)std::min<size_t>(scene->lds_nodes_built, room);
    scene->stack_lds_bytes = stack_bytes;
    scene->lds_bytes = stack_bytes + (size_t)scene->view.lds_nodes * RT_NODE_CACHE_STRIDE;
    scene->lds_bytes += size_sphere_cache(scene->view, scene->lds_bytes);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_kernel<true, true, true, true, true, true, true>, RT_BLOCK, scene->lds_bytes) == hipSuccess && per_cu >= 1) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, scene->device) == hipSuccess) scene->grid_blocks = prop.multiProcessorCount * per_cu;
    }
    // the mesh's box has changed: the boxes of its instances and the TLAS boxes above them follow (the refit of rt_scene_update_transforms)
    if ((rc = refit_tlas_on_device(scene, sm))) {
        if (scene->dev_error_host) *(volatile uint32_t*)scene->dev_error_host |= RT_DEVERR_POISONED;
        return rc;
    }
    return RT_OK;
}

// ---- cancellation and progress inside a call (src/render.rs:131,149,156-158) ------------------------------------------------------
int rt_scene_cancel(RtScene* scene, int cancel) {
    if (!scene) return set_error(RT_E_INVALID, "null scene");
    if (scene->device == RT_DEVICE_HOST_ONLY || !scene->dev_error_host) return set_error(RT_E_NO_DEVICE, "scene was created host-only");
    __atomic_store_n((uint32_t*)scene->cancel_host(), cancel ? 1u : 0u, __ATOMIC_SEQ_CST);      // host-mapped: the waves see it at their next poll
    return RT_OK;
}

int rt_scene_culled_pixels(const RtScene* scene, uint64_t* left_out, uint64_t* without_walk) {
    if (!scene) return set_error(RT_E_INVALID, "null scene");
    if (scene->device == RT_DEVICE_HOST_ONLY || !scene->dev_error_host) return set_error(RT_E_NO_DEVICE, "scene was created host-only");
    if (left_out) *left_out = scene->culled_host()[0];
    if (without_walk) *without_walk = scene->culled_host()[1];
    return RT_OK;
}

int rt_scene_progress(const RtScene* scene, uint64_t* blocks_done, uint64_t* blocks_total) {
    if (!scene) return set_error(RT_E_INVALID, "null scene");
    if (scene->device == RT_DEVICE_HOST_ONLY || !scene->dev_error_host) return set_error(RT_E_NO_DEVICE, "scene was created host-only");
    const uint64_t total = scene->progress_total.load();
    uint64_t done = *scene->progress_host();
    if (done > total) done = total;
    if (blocks_done) *blocks_done = done;
    if (blocks_total) *blocks_total = total;
    return RT_OK;
}

// ---- one frame on N GPUs, composited on the first (src/render.rs:186-224 across devices) -------------------------------------------
// total[p] = acc_0[p] + acc_1[p] + ... in rank order (the deterministic sum of raytracer_amd/dist.py: reduce_accumulators "ordered")
// one rank's packed tile columns (copied to this GPU) to their places in the frame: the composite where a replica cannot write GPU 0's memory itself
__global__ void unpack_columns_kernel(const float* src_rgb, const uint8_t* src_rgba, uint32_t pw, uint32_t npix, uint32_t col_n, uint32_t col_r,
                                      float* rgb, uint8_t* rgba, uint32_t dst_pitch) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const uint32_t x = i % pw, y = i / pw;
    const uint64_t o = (uint64_t)y * dst_pitch + ((x >> 3) * col_n + col_r) * 8u + (x & 7u);
    if (rgb) { rgb[o * 3] = src_rgb[(uint64_t)i * 3]; rgb[o * 3 + 1] = src_rgb[(uint64_t)i * 3 + 1]; rgb[o * 3 + 2] = src_rgb[(uint64_t)i * 3 + 2]; }
    if (rgba) reinterpret_cast<uint32_t*>(rgba)[o] = reinterpret_cast<const uint32_t*>(src_rgba)[i];
}

__global__ void sum_ranks_kernel(float* total, const float* const* parts, uint32_t n_parts, uint64_t n_floats) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats) return;
    float v = parts[0][i];
    for (uint32_t r = 1; r < n_parts; r++) v = v + parts[r][i];
    total[i] = v;
}

int rt_render_frame_multi(RtScene* const* scenes, uint32_t n_scenes, const RtSettings* st, uint32_t mode, uint64_t seed,
                          void* rgb_dev0, void* rgba_dev0, RtRayCounters* counters) {
    if (!scenes || n_scenes == 0 || !st) return set_error(RT_E_INVALID, "rt_render_frame_multi: null argument");
    if (n_scenes > RT_MAX_DEVICES) return set_error(RT_E_LIMIT, "rt_render_frame_multi: more scene replicas than RT_MAX_DEVICES");
    if (mode != RT_MULTI_TILES && mode != RT_MULTI_SPP) return set_error(RT_E_INVALID, "rt_render_frame_multi: mode is RT_MULTI_TILES or RT_MULTI_SPP");
    for (uint32_t i = 0; i < n_scenes; i++) {
        if (!scenes[i]) return set_error(RT_E_INVALID, "rt_render_frame_multi: null scene");
        if (scenes[i]->device == RT_DEVICE_HOST_ONLY) return set_error(RT_E_NO_DEVICE, "rt_render_frame_multi: a scene was created host-only");
        for (uint32_t j = 0; j < i; j++)
            if (scenes[j] == scenes[i]) return set_error(RT_E_INVALID, "rt_render_frame_multi: the same scene twice (every replica needs its own RtScene: they render at the same time)");
    }
    const RtRenderChunk whole{0, 0, st->width, st->height};
    if (int rc = validate_call(scenes[0], st, &whole)) return rc;
    RtScene* s0 = scenes[0];
    const int dev0 = s0->device;
    const uint64_t npix = (uint64_t)st->width * st->height;
    static std::mutex multi_lock;                    // one frame at a time per process (the contexts hold per-device streams and buffers)
    std::lock_guard<std::mutex> mg(multi_lock);
    // ---- context: streams, events, peer access; rebuilt when the list of replicas changes
    MultiContext* mc = s0->multi;
    bool same = mc && mc->scenes == std::vector<RtScene*>(scenes, scenes + n_scenes);
    for (uint32_t i = 0; same && i < n_scenes; i++) same = mc->dev[i].device == scenes[i]->device;      // (a replica destroyed and another created at its address)
    if (!same) {
        if (mc) { mc->release(); delete mc; s0->multi = nullptr; }
        mc = new MultiContext();
        mc->scenes.assign(scenes, scenes + n_scenes);
        mc->dev.resize(n_scenes);
        const bool no_peer = getenv("RT_MULTI_NO_PEER") != nullptr;      // tests: force the copy path even where direct writes are possible (read when a context is built)
        for (uint32_t i = 0; i < n_scenes; i++) {
            MultiContext::PerDevice& d = mc->dev[i];
            d.device = scenes[i]->device;
            hipError_t e = hipSetDevice(d.device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&d.done, hipEventDisableTiming);
            if (e != hipSuccess) { mc->release(); delete mc; return set_error(RT_E_HIP, std::string("rt_render_frame_multi: stream / event: ") + hipGetErrorString(e)); }
            d.direct = d.device == dev0;
            if (!d.direct && !no_peer) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, d.device, dev0) == hipSuccess && can) {
                    e = hipDeviceEnablePeerAccess(dev0, 0);
                    d.direct = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
                    (void)hipGetLastError();
                }
            }
            if (no_peer) d.direct = false;
        }
        s0->multi = mc;
    }
    // ---- per device: enqueue its share on its own stream, from its own host thread (the reference's worker threads, one per GPU)
    auto work = [&](uint32_t i) {
        MultiContext::PerDevice& d = mc->dev[i];
        RtScene* sc = scenes[i];
        d.rc = RT_OK; d.error.clear(); d.host_counters = RtRayCounters{};
        auto fail = [&](int rc) { d.rc = rc; d.error = g_error; };
        auto hip = [&](hipError_t e, const char* what) -> bool { if (e == hipSuccess) return true; set_error(RT_E_HIP, std::string("rt_render_frame_multi: ") + what + ": " + hipGetErrorString(e)); fail(RT_E_HIP); return false; };
        if (!hip(hipSetDevice(d.device), "hipSetDevice")) return;
        int rc;
        if ((rc = d.counters.ensure(sizeof(RtRayCounters)))) return fail(rc);
        if (!hip(hipMemsetAsync(d.counters.p, 0, sizeof(RtRayCounters), d.stream), "hipMemsetAsync")) return;
        if (mode == RT_MULTI_TILES) {
            const int pw = rt_tile_columns_width(st->width, i, n_scenes);
            if (pw > 0) {
                if (d.direct) {
                    // every tile column straight to its place in GPU 0's frame (over this GPU's own xGMI link when it is another device)
                    if ((rc = render_columns_on(sc, st, i, n_scenes, seed, rgb_dev0, rgba_dev0, true, d.counters.p, d.stream))) return fail(rc);
                } else {
                    // no peer access: packed columns here, one copy per output to GPU 0 (staged through the host by the runtime), unpacked there below
                    const size_t np = (size_t)pw * st->height;
                    if (rgb_dev0 && (rc = d.packed_rgb.ensure(np * 12))) return fail(rc);
                    if (rgba_dev0 && (rc = d.packed_rgba.ensure(np * 4))) return fail(rc);
                    if ((rc = render_columns_on(sc, st, i, n_scenes, seed, rgb_dev0 ? d.packed_rgb.p : nullptr, rgba_dev0 ? d.packed_rgba.p : nullptr, false, d.counters.p, d.stream)))
                        return fail(rc);
                    if (rgb_dev0 && !hip(hipMemcpyPeerAsync(d.staging.p, dev0, d.packed_rgb.p, d.device, np * 12, d.stream), "hipMemcpyPeerAsync")) return;
                    if (rgba_dev0 && !hip(hipMemcpyPeerAsync(d.staging_rgba.p, dev0, d.packed_rgba.p, d.device, np * 4, d.stream), "hipMemcpyPeerAsync")) return;
                }
            }
        } else {
            // this replica's range of whole sample segments of the full frame (raytracer_amd/dist.py: sample_range_of_rank)
            const uint64_t seg = RT_SAMPLE_SEGMENT, segs = (st->samples_per_pixel + seg - 1) / seg;
            const uint64_t lo = segs * i / n_scenes * seg, hi_ = segs * (i + 1) / n_scenes * seg, hi = hi_ < st->samples_per_pixel ? hi_ : st->samples_per_pixel;
            if ((rc = d.accum.ensure(npix * 12))) return fail(rc);
            if (!hip(hipMemsetAsync(d.accum.p, 0, npix * 12, d.stream), "hipMemsetAsync")) return;
            if (hi > lo && (rc = rt_accumulate_chunk_device(sc, st, &whole, seed, (uint32_t)lo, (uint32_t)(hi - lo), d.accum.p, d.counters.p, d.stream))) return fail(rc);
            if (d.device != dev0) {          // the accumulator goes to GPU 0 (a peer copy; staged through the host where peer access is missing)
                if (!hip(hipMemcpyPeerAsync(d.staging.p, dev0, d.accum.p, d.device, npix * 12, d.stream), "hipMemcpyPeerAsync")) return;
            }
        }
        if (!hip(hipMemcpyAsync(&d.host_counters, d.counters.p, sizeof(RtRayCounters), hipMemcpyDeviceToHost, d.stream), "hipMemcpyAsync")) return;
        hip(hipEventRecord(d.done, d.stream), "hipEventRecord");
    };
    if (mode == RT_MULTI_SPP) {
        // staging buffers live on GPU 0: allocate them before the workers start
        HIP_TRY(hipSetDevice(dev0));
        for (uint32_t i = 0; i < n_scenes; i++)
            if (mc->dev[i].device != dev0) { if (int rc = mc->dev[i].staging.ensure(npix * 12)) return rc; }
        if (int rc = mc->total.ensure(npix * 12 + (size_t)n_scenes * sizeof(float*))) return rc;
    } else {
        HIP_TRY(hipSetDevice(dev0));
        for (uint32_t i = 0; i < n_scenes; i++) {
            if (mc->dev[i].direct) continue;
            const int pw = rt_tile_columns_width(st->width, i, n_scenes);
            if (pw <= 0) continue;
            if (rgb_dev0) { if (int rc = mc->dev[i].staging.ensure((size_t)pw * st->height * 12)) return rc; }
            if (rgba_dev0) { if (int rc = mc->dev[i].staging_rgba.ensure((size_t)pw * st->height * 4)) return rc; }
        }
    }
    {
        std::vector<std::thread> threads;
        for (uint32_t i = 1; i < n_scenes; i++) {
            try { threads.emplace_back(work, i); }
            catch (...) { work(i); }          // (no thread to be had: this replica's share is enqueued from here - later, not never)
        }
        work(0);
        for (auto& t : threads) t.join();
    }
    // ---- drain every stream (also after a failure: nothing may still write the caller's buffers when this returns)
    int first_rc = RT_OK; std::string first_error;
    for (uint32_t i = 0; i < n_scenes; i++) {
        MultiContext::PerDevice& d = mc->dev[i];
        (void)hipSetDevice(d.device);
        const hipError_t e = hipStreamSynchronize(d.stream);
        if (d.rc == RT_OK && e != hipSuccess) { d.rc = RT_E_HIP; d.error = std::string("rt_render_frame_multi: hipStreamSynchronize: ") + hipGetErrorString(e); }
        if (d.rc != RT_OK && first_rc == RT_OK) { first_rc = d.rc; first_error = d.error; }
    }
    if (first_rc != RT_OK) return set_error(first_rc, first_error);
    HIP_TRY(hipSetDevice(dev0));
    if (mode == RT_MULTI_TILES) {
        // replicas that could not write GPU 0's frame themselves: their packed columns were copied here; put them in place
        hipStream_t s = mc->dev[0].stream;
        bool any = false;
        for (uint32_t i = 0; i < n_scenes; i++) {
            if (mc->dev[i].direct) continue;
            const int pw = rt_tile_columns_width(st->width, i, n_scenes);
            if (pw <= 0) continue;
            const uint32_t np = (uint32_t)pw * st->height;
            hipLaunchKernelGGL(unpack_columns_kernel, dim3((np + 255) / 256), dim3(256), 0, s, rgb_dev0 ? (const float*)mc->dev[i].staging.p : nullptr,
                               rgba_dev0 ? (const uint8_t*)mc->dev[i].staging_rgba.p : nullptr, (uint32_t)pw, np, n_scenes, i, (float*)rgb_dev0, (uint8_t*)rgba_dev0, st->width);
            HIP_TRY_SYNC(hipGetLastError(), s);
            any = true;
        }
        if (any) HIP_TRY(hipStreamSynchronize(s));
    }
    if (mode == RT_MULTI_SPP) {
        // sum in replica order on GPU 0, then col / spp and v3_to_rgba (implementation.rs:659, src/rgba.rs:6-12)
        std::vector<const float*> parts(n_scenes);
        for (uint32_t i = 0; i < n_scenes; i++) parts[i] = (const float*)(mc->dev[i].device == dev0 ? mc->dev[i].accum.p : mc->dev[i].staging.p);
        const float** parts_dev = (const float**)((char*)mc->total.p + npix * 12);
        hipStream_t s = mc->dev[0].stream;
        HIP_TRY_SYNC(hipMemcpyAsync(parts_dev, parts.data(), n_scenes * sizeof(float*), hipMemcpyHostToDevice, s), s);
        const uint64_t nf = npix * 3;
        hipLaunchKernelGGL(sum_ranks_kernel, dim3((uint32_t)((nf + 255) / 256)), dim3(256), 0, s, (float*)mc->total.p, (const float* const*)parts_dev, n_scenes, nf);
        HIP_TRY_SYNC(hipGetLastError(), s);
        const uint32_t n = (uint32_t)npix;
        hipLaunchKernelGGL(finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)mc->total.p, n, 0u, 0u, n, n, (float)st->samples_per_pixel,
                           (float*)rgb_dev0, (uint8_t*)rgba_dev0, n, 0u, 0u, 0u, 0u, 0u);
        HIP_TRY_SYNC(hipGetLastError(), s);
        HIP_TRY(hipStreamSynchronize(s));
    }
    for (uint32_t i = 0; i < n_scenes; i++) {
        if (int rc = check_device_error(scenes[i])) return rc;
        if (*scenes[i]->cancel_host() != 0u) return set_error(RT_E_CANCELLED, "rt_render_frame_multi: cancelled (rt_scene_cancel); the frame holds unfinished pixels");
    }
    if (counters)
        for (uint32_t i = 0; i < n_scenes; i++) {
            const RtRayCounters& c = mc->dev[i].host_counters;
            counters->primary += c.primary; counters->secondary += c.secondary; counters->shadow += c.shadow; counters->samples += c.samples;
        }
    return RT_OK;
}

int rt_render_frame_multi_host(RtScene* const* scenes, uint32_t n_scenes, const RtSettings* st, uint32_t mode, uint64_t seed,
                               float* rgb_out, uint8_t* rgba_out, RtRayCounters* counters) {
    if (!scenes || n_scenes == 0 || !scenes[0] || !st) return set_error(RT_E_INVALID, "rt_render_frame_multi_host: null argument");
    RtScene* s0 = scenes[0];
    if (s0->device == RT_DEVICE_HOST_ONLY) return set_error(RT_E_NO_DEVICE, "rt_render_frame_multi_host: a scene was created host-only");
    const RtRenderChunk whole{0, 0, st->width, st->height};
    if (int rc = validate_call(s0, st, &whole)) return rc;
    const size_t npix = (size_t)st->width * st->height;
    std::lock_guard<std::mutex> g(s0->multi_frame_lock);
    HIP_TRY(hipSetDevice(s0->device));
    int rc;
    if (rgb_out && (rc = s0->multi_frame_rgb.ensure(npix * 12))) return rc;
    if (rgba_out && (rc = s0->multi_frame_rgba.ensure(npix * 4))) return rc;
    if ((rc = rt_render_frame_multi(scenes, n_scenes, st, mode, seed, rgb_out ? s0->multi_frame_rgb.p : nullptr, rgba_out ? s0->multi_frame_rgba.p : nullptr, counters))) return rc;
    HIP_TRY(hipSetDevice(s0->device));
    if (rgb_out) HIP_TRY(hipMemcpy(rgb_out, s0->multi_frame_rgb.p, npix * 12, hipMemcpyDeviceToHost));
    if (rgba_out) HIP_TRY(hipMemcpy(rgba_out, s0->multi_frame_rgba.p, npix * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_resolve_device(int device, const void* accum_dev, uint64_t n_pixels, uint32_t spp, void* rgb_dev, void* rgba_dev, void* stream) {
    if (!accum_dev || spp == 0 || n_pixels == 0 || n_pixels > 0xFFFFFFFFull) return set_error(RT_E_INVALID, "rt_resolve_device: bad arguments");
    HIP_TRY(hipSetDevice(device));
    const uint32_t n = (uint32_t)n_pixels;
    hipLaunchKernelGGL(finish_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)accum_dev, n, 0u, 0u, n, n, (float)spp,
                       (float*)rgb_dev, (uint8_t*)rgba_dev, n, 0u, 0u);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_primary_hits(RtScene* scene, const RtSettings* st, const RtRenderChunk* ch, float* t_out, float* normal_out, int32_t* entity_out,
                    int32_t* tri_out) {
    int rc = validate_call(scene, st, ch);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    WorkspaceLease lease(scene);
    if (lease.rc) return lease.rc;
    Workspace& ws = *lease.ws;
    const size_t npix = (size_t)ch->width * ch->height;
    if ((rc = ws.prim.ensure(npix * 24))) return rc;
    float* d_t = (float*)ws.prim.p; float* d_n = d_t + npix; int32_t* d_e = (int32_t*)(d_n + 3 * npix); int32_t* d_tri = d_e + npix;
    RenderParams P{};
    P.width = st->width; P.height = st->height; P.max_reflections = st->max_reflections; P.spp = 1;
    P.left = ch->left; P.top = ch->top; P.cw = ch->width; P.ch = ch->height;
    P.error_flag = scene->dev_error_dev;
    DevSceneView view = scene->view;
    {
        std::lock_guard<std::mutex> og(scene->opt_lock);
        if (scene->opt.debug_stack_entries && scene->opt.debug_stack_entries < view.stack_entries) view.stack_entries = scene->opt.debug_stack_entries;
    }
    hipLaunchKernelGGL(primary_kernel, dim3((uint32_t)((npix + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), scene->stack_lds_bytes, ws.stream,
                       view, P, d_t, d_n, d_e, d_tri);
    HIP_TRY_SYNC(hipGetLastError(), ws.stream);
    if (t_out) HIP_TRY_SYNC(hipMemcpyAsync(t_out, d_t, npix * 4, hipMemcpyDeviceToHost, ws.stream), ws.stream);
    if (normal_out) HIP_TRY_SYNC(hipMemcpyAsync(normal_out, d_n, npix * 12, hipMemcpyDeviceToHost, ws.stream), ws.stream);
    if (entity_out) HIP_TRY_SYNC(hipMemcpyAsync(entity_out, d_e, npix * 4, hipMemcpyDeviceToHost, ws.stream), ws.stream);
    if (tri_out) HIP_TRY_SYNC(hipMemcpyAsync(tri_out, d_tri, npix * 4, hipMemcpyDeviceToHost, ws.stream), ws.stream);
    HIP_TRY(hipStreamSynchronize(ws.stream));
    return check_device_error(scene);
}

#ifdef RT_BVH4_EXPERIMENT
// Diagnostics build only (make bvh4, tools/bvh4_probe.py): the measurement behind DESIGN.md 4.2's "a 4-wide node was not built".
// Collapses the scene's two-box records into 4-wide ones (children of an inner child take its place; leaves stay), walks both
// trees with the primary rays of a chunk `reps` times and reports kernel times and the number of pixels whose hit differs (0).
int rt_debug_bvh4_primary(RtScene* scene, const RtSettings* st, const RtRenderChunk* ch, int reps, double* ms_binary, double* ms_wide, uint64_t* mismatches) {
    int rc = validate_call(scene, st, ch);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    const size_t nn = scene->node_count, npix = (size_t)ch->width * ch->height;
    std::vector<DevNode> bin(nn);
    HIP_TRY(hipMemcpy(bin.data(), scene->view.nodes, nn * sizeof(DevNode), hipMemcpyDeviceToHost));
    std::vector<Word4> wide(nn * 8);
    const float inf = __builtin_inff();
    for (size_t i = 0; i < nn; i++) {
        Word4* w = &wide[i * 8];
        for (int c = 0; c < 4; c++) { w[0].f[c] = w[1].f[c] = w[2].f[c] = inf; w[3].f[c] = w[4].f[c] = w[5].f[c] = -inf; w[6].u[c] = RT_NONE; w[7].u[c] = 0; }
        int slot = 0;
        auto put = [&](const DevNode& src, int side, uint32_t ref) {      // child `side` (0 left, 1 right) of record src
            const float* f = src.w[0].f;
            w[0].f[slot] = f[0 + side]; w[1].f[slot] = f[2 + side]; w[2].f[slot] = f[4 + side];
            w[3].f[slot] = f[6 + side]; w[4].f[slot] = f[8 + side]; w[5].f[slot] = f[10 + side];
            w[6].u[slot] = ref; slot++;
        };
        for (int side = 0; side < 2; side++) {
            const uint32_t ref = bin[i].w[3].u[side];
            if (ref < RT_REF_LEAF_BIT && ref < nn) { put(bin[ref], 0, bin[ref].w[3].u[0]); put(bin[ref], 1, bin[ref].w[3].u[1]); }
            else put(bin[i], side, ref);
        }
    }
    Word4* wide_dev = nullptr; float* t_dev[2]; int32_t *e_dev[2], *tri_dev[2];
    HIP_TRY(hipMalloc((void**)&wide_dev, wide.size() * sizeof(Word4)));
    HIP_TRY(hipMemcpy(wide_dev, wide.data(), wide.size() * sizeof(Word4), hipMemcpyHostToDevice));
    for (int v = 0; v < 2; v++) { HIP_TRY(hipMalloc((void**)&t_dev[v], npix * 4)); HIP_TRY(hipMalloc((void**)&e_dev[v], npix * 4)); HIP_TRY(hipMalloc((void**)&tri_dev[v], npix * 4)); }
    RenderParams P{};
    P.width = st->width; P.height = st->height; P.max_reflections = st->max_reflections; P.spp = 1;
    P.left = ch->left; P.top = ch->top; P.cw = ch->width; P.ch = ch->height;
    P.error_flag = scene->dev_error_dev;
    // stack: the two-box walk gets what the scene build sized; a wide step pushes up to three refs, so the wide walk gets RT_BVH4_STACK_PCT
    // per cent of that (default 150; an overflow sets the scene's sticky error).  LDS per block, hence occupancy, follows.
    const char* pct_env = getenv("RT_BVH4_STACK_PCT");
    const uint32_t pct = pct_env ? (uint32_t)atoi(pct_env) : 150u;
    const uint32_t caps[2] = {scene->view.stack_entries, (scene->view.stack_entries * pct + 99u) / 100u};
    hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    double ms[2] = {0, 0};
    for (int v = 0; v < 2; v++) {
        for (int r = -1; r < reps; r++) {                             // (one untimed launch first)
            if (r == 0) HIP_TRY(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(primary_wide_kernel, dim3((uint32_t)((npix + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), (size_t)caps[v] * RT_BLOCK * sizeof(uint32_t), 0, scene->view, P,
                               v ? (const Word4*)wide_dev : (const Word4*)nullptr, caps[v], t_dev[v], e_dev[v], tri_dev[v]);
        }
        HIP_TRY(hipEventRecord(e1, 0)); HIP_TRY(hipEventSynchronize(e1));
        float f = 0; HIP_TRY(hipEventElapsedTime(&f, e0, e1)); ms[v] = f / (reps > 0 ? reps : 1);
    }
    std::vector<float> t[2]; std::vector<int32_t> e[2], tri[2];
    for (int v = 0; v < 2; v++) {
        t[v].resize(npix); e[v].resize(npix); tri[v].resize(npix);
        HIP_TRY(hipMemcpy(t[v].data(), t_dev[v], npix * 4, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(e[v].data(), e_dev[v], npix * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(tri[v].data(), tri_dev[v], npix * 4, hipMemcpyDeviceToHost));
        (void)hipFree(t_dev[v]); (void)hipFree(e_dev[v]); (void)hipFree(tri_dev[v]);
    }
    (void)hipFree(wide_dev); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    uint64_t bad = 0;
    for (size_t i = 0; i < npix; i++) bad += (memcmp(&t[0][i], &t[1][i], 4) != 0 || e[0][i] != e[1][i] || tri[0][i] != tri[1][i]) ? 1 : 0;
    if (ms_binary) *ms_binary = ms[0];
    if (ms_wide) *ms_wide = ms[1];
    if (mismatches) *mismatches = bad;
    return check_device_error(scene);
}
#endif

int rt_create_render_chunks(uint32_t width, uint32_t height, uint32_t count_x, uint32_t count_y, RtRenderChunk* out, uint32_t capacity) {
    int rc = host_create_render_chunks(width, height, count_x, count_y, out, capacity);
    if (rc < 0) set_error(rc, "rt_create_render_chunks: zero count or capacity too small");
    return rc;
}

void rt_v3_to_rgba(const float* rgb, uint64_t n_pixels, uint8_t* rgba_out) { host_v3_to_rgba(rgb, n_pixels, rgba_out); }

int rt_scene_info(const RtScene* scene, RtSceneInfo* out) {
    if (!scene || !out) return set_error(RT_E_INVALID, "null argument");
    *out = scene->info;
    return RT_OK;
}

int rt_scene_profile_enable(RtScene* scene, int enable) {
    if (!scene) return set_error(RT_E_INVALID, "null scene");
    std::lock_guard<std::mutex> g(scene->profile_lock);
    scene->profiling = enable != 0;
    return RT_OK;
}

int rt_scene_profile_read(RtScene* scene, double* kernel_ms_total, uint64_t* launches) {
    if (!scene) return set_error(RT_E_INVALID, "null scene");
    std::lock_guard<std::mutex> g(scene->profile_lock);
    if (scene->device != RT_DEVICE_HOST_ONLY) HIP_TRY(hipSetDevice(scene->device));
    double total = 0.0;
    for (auto& ev : scene->profile_events) {
        HIP_TRY(hipEventSynchronize(ev.second));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        total += ms;
        (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
    }
    if (kernel_ms_total) *kernel_ms_total = total;
    if (launches) *launches = scene->profile_events.size();
    scene->profile_events.clear();
    return RT_OK;
}

int64_t rt_scene_bvh_order(const RtScene* scene, int64_t mesh_index, uint32_t* out, uint64_t capacity) {
    if (!scene || mesh_index >= (int64_t)scene->blas_order.size()) return set_error(RT_E_INVALID, "rt_scene_bvh_order: bad arguments");
    const std::vector<uint32_t>& o = mesh_index < 0 ? scene->tlas_order : scene->blas_order[mesh_index];
    if (out) for (size_t i = 0; i < o.size() && i < capacity; i++) out[i] = o[i];
    return (int64_t)o.size();
}

}  // extern "C"
