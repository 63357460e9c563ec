/*
 * rt_oracle.cpp - CPU ORACLE: a restatement of deadalusai/raytracer's render core.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h): the checker for the HIP path and
 * the timed `cpu_baseline` leg of bench.py.  Never linked into the product.
 *
 * Follows the reference operation-for-operation in f32 (compile with
 * -O2 -ffp-contract=off, no -ffast-math): same operand order, same libm calls
 * (Rust's f32::sin/cos/tan/acos/powf are the platform libm on Linux), same BVH
 * build, same depth-first candidate order, same reducers.  Each function cites
 * the reference file:line it restates (paths relative to /root/reference).
 *
 * One deliberate difference: the shadow ray's loop through transparent occluders (cast_light_ray_to_lamp) terminates
 * where the reference's does not (see there).
 *
 * "parity unpinned": the reference cannot be built here (no rustc); only
 * partition_by_key and create_render_chunks are pinned by the reference's own
 * tests.  XorShift/seed_from_u64 restate rand_xorshift 0.4.0 / rand_core 0.9.3
 * (Cargo.lock:2561,2570) from their published algorithms and are pinned by
 * rand_xorshift's published test vector (tests/test_oracle_kat.py).
 */
#include "rt_oracle.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

// ---------------------------------------------------------------------------
// types.rs:11-207  V3 / V2 / Ray
// ---------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct V2 { float x, y; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }       // types.rs:78-85
inline V3 operator+(V3 a, float f) { return v3(a.x + f, a.y + f, a.z + f); }           // :87-94
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }       // :96-103
inline V3 operator-(V3 a, float f) { return v3(a.x - f, a.y - f, a.z - f); }           // :105-112
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }       // :114-121
inline V3 operator*(V3 a, float f) { return v3(a.x * f, a.y * f, a.z * f); }           // :123-130
inline V3 operator/(V3 a, float f) { return v3(a.x / f, a.y / f, a.z / f); }           // :141-148
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }                             // :150-157
inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(V3 a, V3 b) { return !(a == b); }

inline V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
inline V2 operator*(V2 a, float f) { return V2{a.x * f, a.y * f}; }

const V3 ZERO = {0.0f, 0.0f, 0.0f};
const V3 POS_X = {1.0f, 0.0f, 0.0f};
const V3 POS_Y = {0.0f, 1.0f, 0.0f};
const float F_INF = INFINITY;
const float F32_MAX = 3.40282347e+38f;

inline float length_squared(V3 a) { return (a.x * a.x) + (a.y * a.y) + (a.z * a.z); } // types.rs:56-58
inline float length(V3 a) { return sqrtf(length_squared(a)); }                        // :52-54
inline V3 unit(V3 a) { float len = length(a); return len == 0.0f ? a : a / len; }     // :41-44
inline V3 clamp(V3 a) {                                                               // :46-50
    return v3(fmaxf(fminf(a.x, 1.0f), -1.0f), fmaxf(fminf(a.y, 1.0f), -1.0f), fmaxf(fminf(a.z, 1.0f), -1.0f));
}
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }            // :60-62
inline V3 cross(V3 a, V3 b) {                                                         // :64-68
    return v3((a.y * b.z - a.z * b.y), -(a.x * b.z - a.z * b.x), (a.x * b.y - a.y * b.x));
}
inline float theta_between(V3 a, V3 b) {                                              // :71-75
    return acosf(dot(a, b) / (length(a) * length(b)));
}
// types.rs:191-201  Rodrigues rotation, cos/sin evaluated per call
inline V3 rotate_about_axis(V3 p, V3 axis, float theta) {
    return (p * cosf(theta)) + (cross(axis, p) * sinf(theta)) + (axis * dot(axis, p) * (1.0f - cosf(theta)));
}

struct Ray { V3 origin, direction; };
inline V3 point_at_parameter(const Ray& r, float t) { return r.origin + (r.direction * t); } // :180-182

inline V3 from(RtV3 v) { return v3(v.x, v.y, v.z); }
inline V2 from(RtV2 v) { return V2{v.x, v.y}; }

// ---------------------------------------------------------------------------
// RNG: rand_xorshift 0.4.0 + rand_core 0.9.3 seed_from_u64 + rand 0.9.2 f32
// ---------------------------------------------------------------------------
struct XorShift {
    uint32_t x, y, z, w;
    void from_seed(const uint8_t* s) {   // XorShiftRng::from_seed: 4 LE u32, all-zero -> constants
        uint32_t v[4];
        for (int i = 0; i < 4; i++)
            v[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) | ((uint32_t)s[4 * i + 3] << 24);
        if (v[0] == 0 && v[1] == 0 && v[2] == 0 && v[3] == 0) { v[0] = v[1] = v[2] = v[3] = 0x0BAD5EEDu; }
        x = v[0]; y = v[1]; z = v[2]; w = v[3];
    }
    void seed_from_u64(uint64_t state) { // rand_core SeedableRng::seed_from_u64 (PCG32 expansion)
        uint8_t seed[16];
        for (int i = 0; i < 4; i++) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            uint32_t v = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
            seed[4 * i] = v & 0xff; seed[4 * i + 1] = (v >> 8) & 0xff; seed[4 * i + 2] = (v >> 16) & 0xff; seed[4 * i + 3] = (v >> 24) & 0xff;
        }
        from_seed(seed);
    }
    uint32_t next_u32() {
        uint32_t t = x ^ (x << 11);
        x = y; y = z; z = w;
        w = w ^ (w >> 19) ^ (t ^ (t >> 8));
        return w;
    }
};

inline float u32_to_f32(uint32_t v) { return (float)(v >> 8) * (1.0f / 16777216.0f); } // rand StandardUniform f32

void philox4x32_10(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4]) {
    uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3];
    uint32_t k0 = k_in[0], k1 = k_in[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { PURPOSE_LENS = 0, PURPOSE_PATH = 1, PURPOSE_SHADOW = 2 };

// The RNG the render loop sees.  XorShift ignores contexts (sequential stream,
// reference semantics); the counter RNG is addressed by them.
struct Rng {
    int mode = ORC_RNG_XORSHIFT_CHUNK;     // XORSHIFT_CHUNK or COUNTER (the generator); COUNTER_STRATIFIED = COUNTER + `stratified`
    bool stratified = false;
    XorShift xs;
    // counter mode
    uint32_t key[2] = {0, 0};
    uint32_t px = 0, py = 0, sample = 0, node = 0, purpose = 0, k = 0;
    uint32_t block[4]; uint32_t block_id = 0xffffffffu;

    void set_pixel_sample(uint32_t x, uint32_t y, uint32_t s) { px = x; py = y; sample = s; block_id = 0xffffffffu; }
    void set_ctx(uint32_t n, uint32_t p) { node = n; purpose = p; k = 0; block_id = 0xffffffffu; }
    float next_f32() {
        if (mode == ORC_RNG_XORSHIFT_CHUNK) return u32_to_f32(xs.next_u32());
        uint32_t b = k >> 2;
        if (b != block_id) {
            uint32_t ctr[4] = { px | (py << 16), sample, node, purpose | (b << 8) };
            philox4x32_10(ctr, key, block);
            block_id = b;
        }
        return u32_to_f32(block[(k++) & 3]);
    }
};

// ---------------------------------------------------------------------------
// implementation.rs:41-132  AABB
// ---------------------------------------------------------------------------
struct AABB { V3 min, max; };

inline AABB aabb_infinite() { return AABB{v3(-F_INF, -F_INF, -F_INF), v3(F_INF, F_INF, F_INF)}; } // :54-56
inline bool aabb_is_infinite(const AABB& b) {                                                     // :88-90
    return b.min == v3(-F_INF, -F_INF, -F_INF) && b.max == v3(F_INF, F_INF, F_INF);
}
template <class It> AABB aabb_from_vertices(It begin, It end) {                                  // :69-86
    V3 mn = *begin, mx = *begin;
    for (It it = begin + 1; it != end; ++it) {
        mn.x = fminf(mn.x, it->x); mn.y = fminf(mn.y, it->y); mn.z = fminf(mn.z, it->z);
        mx.x = fmaxf(mx.x, it->x); mx.y = fmaxf(mx.y, it->y); mx.z = fmaxf(mx.z, it->z);
    }
    return AABB{mn, mx};
}
inline void aabb_corners(const AABB& b, V3 c[8]) {                                                // :120-131
    c[0] = b.min;
    c[1] = v3(b.min.x, b.min.y, b.max.z);
    c[2] = v3(b.min.x, b.max.y, b.min.z);
    c[3] = v3(b.max.x, b.min.y, b.min.z);
    c[4] = b.max;
    c[5] = v3(b.max.x, b.max.y, b.min.z);
    c[6] = v3(b.max.x, b.min.y, b.max.z);
    c[7] = v3(b.min.x, b.max.y, b.max.z);
}
// implementation.rs:92-118
inline bool hit_aabb(const AABB& b, const Ray& ray, float t_min, float t_max) {
    if (aabb_is_infinite(b)) return true;
    const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z};
    const float d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
    const float mn[3] = {b.min.x, b.min.y, b.min.z};
    const float mx[3] = {b.max.x, b.max.y, b.max.z};
    for (int k = 0; k <= 2; k++) {
        float inv_d = 1.0f / d[k];
        float t0 = (mn[k] - o[k]) * inv_d;
        float t1 = (mx[k] - o[k]) * inv_d;
        if (inv_d < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max < t_min) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------
// bvh.rs + util.rs
// ---------------------------------------------------------------------------
struct BvhBounds { V3 centroid; AABB aabb; };
struct BvhObjectBounds { size_t object_index; BvhBounds bounds; };
struct BvhNode { AABB aabb; bool leaf; size_t a, b; };  // branch: a=left_offset b=right_offset; leaf: a=offset b=length
struct Bvh {
    std::vector<BvhObjectBounds> object_bounds;
    std::vector<BvhNode> nodes;
};

// util.rs:6-28.  Returns i+1 (the length of the left partition).
template <class T, class K, class Sel>
size_t partition_by_key(T* slice, size_t len, K split_on, Sel selector) {
    size_t i = 0, j = len - 1;
    while (i < j) {
        if (selector(slice[i]) <= split_on) {
            i += 1;
        } else {
            T tmp = slice[i]; slice[i] = slice[j]; slice[j] = tmp;
            j -= 1;
        }
    }
    return i + 1;
}

BvhNode create_leaf(size_t offset, const BvhObjectBounds* ob, size_t n) {            // bvh.rs:109-116
    std::vector<V3> verts; verts.reserve(2 * n);
    for (size_t i = 0; i < n; i++) { verts.push_back(ob[i].bounds.aabb.min); verts.push_back(ob[i].bounds.aabb.max); }
    return BvhNode{aabb_from_vertices(verts.begin(), verts.end()), true, offset, n};
}

inline int select_longest_axis(V3 extent) {                                            // bvh.rs:118-127
    int axis = 0; float len = fabsf(extent.x);
    if (extent.y > len) { axis = 1; len = fabsf(extent.y); }
    if (extent.z > len) { axis = 2; }
    return axis;
}
inline float axis_value(int axis, V3 v) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

void subdivide(size_t offset, std::vector<BvhNode>& nodes, std::vector<BvhObjectBounds>& ob) { // bvh.rs:144-188
    const size_t leaf_offset = nodes[offset].a, leaf_length = nodes[offset].b;
    if (leaf_length <= 2) return;
    BvhObjectBounds* range = ob.data() + leaf_offset;
    // bounds of the centroids
    V3 mn = range[0].bounds.centroid, mx = mn;
    for (size_t i = 1; i < leaf_length; i++) {
        V3 c = range[i].bounds.centroid;
        mn.x = fminf(mn.x, c.x); mn.y = fminf(mn.y, c.y); mn.z = fminf(mn.z, c.z);
        mx.x = fmaxf(mx.x, c.x); mx.y = fmaxf(mx.y, c.y); mx.z = fmaxf(mx.z, c.z);
    }
    V3 extent = mx - mn;
    int axis = select_longest_axis(extent);
    float sum = 0.0f;
    for (size_t i = 0; i < leaf_length; i++) sum += axis_value(axis, range[i].bounds.centroid);
    float split_on = sum / (float)leaf_length;
    size_t left_len = partition_by_key(range, leaf_length, split_on,
                                       [axis](const BvhObjectBounds& o) { return axis_value(axis, o.bounds.centroid); });
    size_t right_len = leaf_length - left_len;
    if (left_len == 0 || right_len == 0) return;
    BvhNode l1 = create_leaf(leaf_offset, range, left_len);
    BvhNode l2 = create_leaf(leaf_offset + left_len, range + left_len, right_len);
    size_t left_offset = nodes.size(); nodes.push_back(l1);
    size_t right_offset = nodes.size(); nodes.push_back(l2);
    nodes[offset].leaf = false; nodes[offset].a = left_offset; nodes[offset].b = right_offset;
    subdivide(left_offset, nodes, ob);
    subdivide(right_offset, nodes, ob);
}

Bvh bvh_build(std::vector<BvhObjectBounds> ob) {                                       // bvh.rs:129-142
    Bvh b;
    b.nodes.reserve(ob.size() * 2);
    b.nodes.push_back(create_leaf(0, ob.data(), ob.size()));
    b.object_bounds = std::move(ob);
    subdivide(0, b.nodes, b.object_bounds);
    return b;
}

int bvh_depth(const Bvh& b, size_t node = 0) {
    const BvhNode& n = b.nodes[node];
    if (n.leaf) return 1;
    int l = bvh_depth(b, n.a), r = bvh_depth(b, n.b);
    return 1 + (l > r ? l : r);
}

const int REF_STACK_CAP = 30;   // ArrayVec<State, 30>, bvh.rs:192
struct StackOverflow {};

// bvh.rs:190-236: the candidate iterator. `visit(object_index)` is called in
// emission order; `node_tests` counts hit_aabb evaluations.
template <class F>
void hit_candidates(const Bvh& bvh, const Ray& ray, float t_min, float t_max, uint64_t& node_tests, F visit) {
    struct State { size_t node_index, offset; };
    State stack[REF_STACK_CAP]; int sp = 0;
    stack[sp++] = State{0, 0};
    while (sp > 0) {
        State st = stack[--sp];
        const BvhNode& node = bvh.nodes[st.node_index];
        node_tests++;
        if (!hit_aabb(node.aabb, ray, t_min, t_max)) continue;
        if (!node.leaf) {
            if (sp + 2 > REF_STACK_CAP) throw StackOverflow{};
            stack[sp++] = State{node.a, 0};
            stack[sp++] = State{node.b, 0};
        } else {
            if (st.offset < node.b - 1) {
                if (sp + 1 > REF_STACK_CAP) throw StackOverflow{};
                stack[sp++] = State{st.node_index, st.offset + 1};
            }
            visit(bvh.object_bounds[node.a + st.offset].object_index);
        }
    }
}

// ---------------------------------------------------------------------------
// HitRecord (implementation.rs:170-179) + oracle-side bookkeeping ids
// ---------------------------------------------------------------------------
struct HitRecord {
    int32_t entity_id;   // Option<u32>, -1 = None
    float t;
    V3 p, normal;
    V2 uv;
    uint32_t mat_id, tex_id;
    int32_t tex_key;     // Option<usize>, -1 = None
    int32_t entity_index = -1, tri_index = -1;   // parity outputs only
};

struct Counters {
    uint64_t primary = 0, secondary = 0, shadow = 0, samples = 0;
    uint64_t ref_nodes = 0, ref_spheres = 0, ref_planes = 0, ref_tris = 0, ref_xforms = 0;
    uint64_t min_nodes = 0, min_spheres = 0, min_planes = 0, min_tris = 0, min_xforms = 0, min_instances = 0, min_mismatch = 0;
    void add_to(OrcCounters* o) const {
        o->rays.primary += primary; o->rays.secondary += secondary; o->rays.shadow += shadow; o->rays.samples += samples;
        o->ref_nodes += ref_nodes; o->ref_spheres += ref_spheres; o->ref_planes += ref_planes; o->ref_tris += ref_tris; o->ref_xforms += ref_xforms;
        o->min_nodes += min_nodes; o->min_spheres += min_spheres; o->min_planes += min_planes; o->min_tris += min_tris;
        o->min_xforms += min_xforms; o->min_instances += min_instances; o->min_mismatch += min_mismatch;
    }
};

// ---------------------------------------------------------------------------
// shapes/sphere.rs
// ---------------------------------------------------------------------------
inline bool intersect_sphere(const Ray& ray, V3 origin, float radius, float ts[2]) {  // sphere.rs:4-17
    V3 oc = ray.origin - origin;
    float a = dot(ray.direction, ray.direction);
    float b = dot(oc, ray.direction);
    float c = dot(oc, oc) - radius * radius;
    float discriminant = b * b - a * c;
    if (discriminant > 0.0f) {
        ts[0] = (-b - sqrtf(discriminant)) / a;
        ts[1] = (-b + sqrtf(discriminant)) / a;
        return true;
    }
    return false;
}

inline bool sphere_hit(const RtSphere& s, const Ray& ray, float t_min, float t_max, HitRecord& out) { // sphere.rs:36-53
    float ts[2];
    if (!intersect_sphere(ray, ZERO, s.radius, ts)) return false;
    bool have = false; float t = 0.0f;
    for (int i = 0; i < 2; i++) {
        if (t_min < ts[i] && ts[i] < t_max) { t = have ? fminf(t, ts[i]) : ts[i]; have = true; }
    }
    if (!have) return false;
    V3 p = point_at_parameter(ray, t);
    V3 normal = unit(p / s.radius);
    out = HitRecord{-1, t, p, normal, V2{0.0f, 0.0f}, s.mat_id, s.tex_id, -1};
    return true;
}

// ---------------------------------------------------------------------------
// shapes/plane.rs
// ---------------------------------------------------------------------------
inline bool intersect_plane(const Ray& ray, V3 origin, V3 normal, float& t) {          // plane.rs:4-17
    float denominator = dot(ray.direction, normal);
    if (fabsf(denominator) < 1.0e-6f) return false;
    float numerator = dot(origin - ray.origin, normal);
    t = numerator / denominator;
    return true;
}

inline bool plane_hit(const RtPlane& pl, const Ray& ray, float t_min, float t_max, HitRecord& out) { // plane.rs:59-90
    float t;
    V3 n = from(pl.normal);
    if (!intersect_plane(ray, ZERO, n, t)) return false;
    if (t < t_min || t > t_max) return false;
    V3 p = point_at_parameter(ray, t);
    if (pl.has_radius) { if (length(p) > pl.radius) return false; }
    V3 normal = dot(ray.direction, n) > 0.0f ? -n : n;
    V2 uv = V2{dot(from(pl.u_basis), p), dot(from(pl.v_basis), p)};
    out = HitRecord{-1, t, p, normal, uv, pl.mat_id, pl.tex_id, -1};
    return true;
}

// ---------------------------------------------------------------------------
// shapes/mesh.rs
// ---------------------------------------------------------------------------
struct MeshTriHit { V3 p, normal; float t; V2 uv; int32_t tex_key; int32_t tri_index; };

inline bool try_hit_tri(const Ray& ray, float t_min, float t_max, const RtMeshTri& tri, MeshTriHit& out) { // mesh.rs:18-76
    V3 a = from(tri.a), b = from(tri.b), c = from(tri.c);
    V3 edge_ab = b - a;
    V3 edge_ac = c - a;
    V3 normal = cross(edge_ab, edge_ac);
    float t;
    if (!intersect_plane(ray, a, normal, t)) return false;
    if (t < t_min || t > t_max) return false;
    V3 p = point_at_parameter(ray, t);
    V3 normal_abp = cross(edge_ab, p - a);
    if (dot(normal, normal_abp) < 0.0f) return false;
    V3 normal_bcp = cross(c - b, p - b);
    if (dot(normal, normal_bcp) < 0.0f) return false;
    V3 normal_cap = cross(a - c, p - c);
    if (dot(normal, normal_cap) < 0.0f) return false;
    float area2 = length(normal);
    float u = length(normal_cap) / area2;
    float v = length(normal_abp) / area2;
    float w = 1.0f - u - v;
    V2 uv = (from(tri.a_uv) * w) + (from(tri.b_uv) * u) + (from(tri.c_uv) * v);
    out = MeshTriHit{p, unit(normal), t, uv, tri.tex_key, -1};
    return true;
}

struct MeshBvhRoot {       // mesh.rs:78-100
    Bvh bvh;
    std::vector<RtMeshTri> tris;
};

inline BvhBounds tri_bounds(const RtMeshTri& t) {                                      // mesh.rs:127-133
    V3 v[3] = {from(t.a), from(t.b), from(t.c)};
    AABB aabb = aabb_from_vertices(v, v + 3);
    V3 centroid = (v[0] + v[1] + v[2]) * 0.33333f;
    return BvhBounds{centroid, aabb};
}

// ---------------------------------------------------------------------------
// Scene
// ---------------------------------------------------------------------------
struct Entity {            // implementation.rs:224-230
    RtEntity desc;
    std::vector<V3> translations;
    std::vector<RtRotation> rotations;
};

}  // namespace

struct OrcScene {
    RtCamera camera;
    uint32_t sky;
    std::vector<Entity> entities;
    std::vector<RtSphere> spheres;
    std::vector<RtPlane> planes;
    std::vector<RtMeshObject> mesh_objects;
    std::vector<MeshBvhRoot> meshes;
    std::vector<RtMaterial> materials;
    std::vector<RtTexture> textures;
    std::vector<int32_t> tex_set_entries;
    std::vector<V3> texels;
    std::vector<RtLight> lights;
    Bvh tlas;    // EntityBvhRoot.bvh, implementation.rs:398-409
};

namespace {

AABB hitable_aabb(const OrcScene& s, const RtEntity& e) {
    switch (e.hitable_kind) {
        case RT_HITABLE_SPHERE: { float r = s.spheres[e.hitable_index].radius; return AABB{ZERO - r, ZERO + r}; } // sphere.rs:55-58
        case RT_HITABLE_PLANE: return aabb_infinite();                                                            // plane.rs:92-95
        default: return s.meshes[s.mesh_objects[e.hitable_index].mesh_index].bvh.nodes[0].aabb;                   // mesh.rs:178-180
    }
}

// implementation.rs:258-283
void calculate_origin_aabb(const OrcScene& s, const Entity& e, V3& origin, AABB& aabb) {
    aabb = hitable_aabb(s, e.desc);
    V3 corners[8]; aabb_corners(aabb, corners);
    if (!aabb_is_infinite(aabb)) {
        for (size_t i = e.rotations.size(); i-- > 0;) {
            const RtRotation& r = e.rotations[i];
            for (int c = 0; c < 8; c++) corners[c] = rotate_about_axis(corners[c], from(r.axis), r.theta);
        }
    }
    origin = ZERO;
    for (const V3& t : e.translations) {
        for (int c = 0; c < 8; c++) corners[c] = corners[c] + t;
        origin = origin + t;
    }
    aabb = aabb_from_vertices(corners, corners + 8);
}

// implementation.rs:433-440
BvhBounds entity_bounds(const OrcScene& s, const Entity& e) {
    V3 origin; AABB aabb;
    calculate_origin_aabb(s, e, origin, aabb);
    V3 centroid = aabb_is_infinite(aabb) ? origin : aabb.min * 0.5f + aabb.max * 0.5f;
    return BvhBounds{centroid, aabb};
}

// mesh.rs:92-99
bool mesh_try_hit(const MeshBvhRoot& m, const Ray& ray, float t_min, float t_max, MeshTriHit& best, Counters& cn) {
    bool have = false;
    hit_candidates(m.bvh, ray, t_min, t_max, cn.ref_nodes, [&](size_t idx) {
        MeshTriHit h;
        cn.ref_tris++;
        if (try_hit_tri(ray, t_min, t_max, m.tris[idx], h)) {
            h.tri_index = (int32_t)idx;
            if (!have) { best = h; have = true; }
            else if (h.t < best.t) best = h;
        }
    });
    return have;
}

// implementation.rs:285-310
bool entity_hit(const OrcScene& s, const Entity& e, Ray ray, float t_min, float t_max, HitRecord& hit, Counters& cn) {
    for (const V3& t : e.translations) ray.origin = ray.origin - t;
    for (const RtRotation& r : e.rotations) {
        ray.origin = rotate_about_axis(ray.origin, from(r.axis), -r.theta);
        ray.direction = rotate_about_axis(ray.direction, from(r.axis), -r.theta);
        cn.ref_xforms++;
    }
    bool ok = false;
    switch (e.desc.hitable_kind) {
        case RT_HITABLE_SPHERE: cn.ref_spheres++; ok = sphere_hit(s.spheres[e.desc.hitable_index], ray, t_min, t_max, hit); break;
        case RT_HITABLE_PLANE: cn.ref_planes++; ok = plane_hit(s.planes[e.desc.hitable_index], ray, t_min, t_max, hit); break;
        default: {                                                                     // mesh.rs:161-176
            const RtMeshObject& mo = s.mesh_objects[e.desc.hitable_index];
            MeshTriHit mh;
            ok = mesh_try_hit(s.meshes[mo.mesh_index], ray, t_min, t_max, mh, cn);
            if (ok) { hit = HitRecord{-1, mh.t, mh.p, mh.normal, mh.uv, mo.mat_id, mo.tex_id, mh.tex_key}; hit.tri_index = mh.tri_index; }
        }
    }
    if (!ok) return false;
    for (size_t i = e.rotations.size(); i-- > 0;) {
        const RtRotation& r = e.rotations[i];
        hit.p = rotate_about_axis(hit.p, from(r.axis), r.theta);
        hit.normal = rotate_about_axis(hit.normal, from(r.axis), r.theta);
    }
    for (size_t i = e.translations.size(); i-- > 0;) hit.p = hit.p + e.translations[i];
    hit.entity_id = e.desc.id;
    return true;
}

// Scene::hit_closest -> EntityBvhRoot::try_hit, implementation.rs:372-375, :411-416
bool hit_closest_ref(const OrcScene& s, const Ray& ray, float t_min, float t_max, HitRecord& best, Counters& cn) {
    bool have = false;
    hit_candidates(s.tlas, ray, t_min, t_max, cn.ref_nodes, [&](size_t idx) {
        HitRecord h;
        if (entity_hit(s, s.entities[idx], ray, t_min, t_max, h, cn)) {
            h.entity_index = (int32_t)idx;
            if (!have) { best = h; have = true; }
            else if (h.t < best.t) best = h;
        }
    });
    return have;
}

// --- stats only: ordered, t_max-shrinking closest-hit traversal of the same
// trees (SURVEY.md 8d "algorithmic minimum"); result must equal the reference
// order's (ties broken by reference visit rank). Not reference code. ----------
struct MinHit { bool have = false; float t = 0; int32_t entity = -1, tri = -1; };

struct RankTables {   // visit rank of every object in reference emission order
    std::vector<uint32_t> tlas_rank;
    std::vector<std::vector<uint32_t>> blas_rank;
};

void bvh_emission_order(const Bvh& b, std::vector<uint32_t>& out) {
    std::vector<std::pair<size_t, size_t>> st; st.push_back({0, 0});
    while (!st.empty()) {
        auto [ni, off] = st.back(); st.pop_back();
        const BvhNode& n = b.nodes[ni];
        if (!n.leaf) { st.push_back({n.a, 0}); st.push_back({n.b, 0}); }
        else {
            if (off < n.b - 1) st.push_back({ni, off + 1});
            out.push_back((uint32_t)b.object_bounds[n.a + off].object_index);
        }
    }
}

inline bool slab_entry(const AABB& b, const Ray& ray, float t_min, float t_max, float& entry) {
    if (aabb_is_infinite(b)) { entry = t_min; return true; }
    const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z};
    const float d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
    const float mn[3] = {b.min.x, b.min.y, b.min.z};
    const float mx[3] = {b.max.x, b.max.y, b.max.z};
    for (int k = 0; k <= 2; k++) {
        float inv_d = 1.0f / d[k];
        float t0 = (mn[k] - o[k]) * inv_d, t1 = (mx[k] - o[k]) * inv_d;
        if (inv_d < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max < t_min) return false;
    }
    entry = t_min;
    return true;
}

// Near-child-first traversal; `nodes` counts box tests (root 1, then 2 per
// inner node visited). A popped node whose entry lies beyond the current best
// hit is skipped without a new box test.
// `best_t` is the CULLING bound, not the best hit's t: the boxes of the reference do not bound the
// computed t of their hits to the last bit (an axis-aligned triangle lies in a face of its own box, and
// its plane-equation t is often 1-3 ulp before the box's slab entry; rotated entities add the rounding of
// the rotated ray), so a walk that culls at exactly the best t loses candidates the reference finds -
// 450 of 4.4 M sample rays on the 900-instance scene, nearly all exact ties between coincident faces.
// The caller therefore keeps best_t = t_best * ORC_CULL_SLACK (the product's RT_CULL_SLACK).
const float ORC_CULL_SLACK = 1.0009765625f;    // 1 + 2^-10
template <class Leaf>
void ordered_traverse(const Bvh& bvh, const Ray& ray, float t_min, const float& best_t, uint64_t& nodes, Leaf leaf) {
    struct Item { size_t node; float entry; };
    std::vector<Item> st;
    float e0;
    nodes++;
    if (!slab_entry(bvh.nodes[0].aabb, ray, t_min, best_t, e0)) return;
    st.push_back(Item{0, e0});
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        if (best_t < it.entry) continue;
        const BvhNode& n = bvh.nodes[it.node];
        if (n.leaf) { for (size_t k = 0; k < n.b; k++) leaf(bvh.object_bounds[n.a + k].object_index); continue; }
        float el = 0, er = 0;
        nodes += 2;
        bool hl = slab_entry(bvh.nodes[n.a].aabb, ray, t_min, best_t, el);
        bool hr = slab_entry(bvh.nodes[n.b].aabb, ray, t_min, best_t, er);
        if (hl && hr) {
            if (el <= er) { st.push_back(Item{n.b, er}); st.push_back(Item{n.a, el}); }
            else { st.push_back(Item{n.a, el}); st.push_back(Item{n.b, er}); }
        } else if (hl) st.push_back(Item{n.a, el});
        else if (hr) st.push_back(Item{n.b, er});
    }
}

float g_cull_slack = ORC_CULL_SLACK;      // (diagnostics can set it: orc_set_cull_slack)
MinHit hit_closest_min(const OrcScene& s, const RankTables& rk, const Ray& ray, float t_min, float t_max, Counters& cn) {
    MinHit best; float best_t = t_max; uint32_t best_erank = 0, best_trank = 0;
    ordered_traverse(s.tlas, ray, t_min, best_t, cn.min_nodes, [&](size_t ei) {
        const Entity& e = s.entities[ei];
        Ray r = ray;
        for (const V3& t : e.translations) r.origin = r.origin - t;
        for (const RtRotation& rot : e.rotations) {
            r.origin = rotate_about_axis(r.origin, from(rot.axis), -rot.theta);
            r.direction = rotate_about_axis(r.direction, from(rot.axis), -rot.theta);
            cn.min_xforms++;
        }
        auto consider = [&](float t, int32_t tri, uint32_t trank) {
            uint32_t er = rk.tlas_rank[ei];
            bool better = !best.have || t < best.t || (t == best.t && (er < best_erank || (er == best_erank && trank < best_trank)));
            if (better) { best.have = true; best.t = t; best.entity = (int32_t)ei; best.tri = tri; best_erank = er; best_trank = trank; best_t = fminf(t * g_cull_slack, t_max); }
        };
        HitRecord h;
        switch (e.desc.hitable_kind) {
            case RT_HITABLE_SPHERE: cn.min_spheres++; if (sphere_hit(s.spheres[e.desc.hitable_index], r, t_min, t_max, h)) consider(h.t, -1, 0); break;
            case RT_HITABLE_PLANE: cn.min_planes++; if (plane_hit(s.planes[e.desc.hitable_index], r, t_min, t_max, h)) consider(h.t, -1, 0); break;
            default: {
                cn.min_instances++;
                uint32_t mi = s.mesh_objects[e.desc.hitable_index].mesh_index;
                const MeshBvhRoot& m = s.meshes[mi];
                ordered_traverse(m.bvh, r, t_min, best_t, cn.min_nodes, [&](size_t ti) {
                    MeshTriHit mh; cn.min_tris++;
                    if (try_hit_tri(r, t_min, t_max, m.tris[ti], mh)) consider(mh.t, (int32_t)ti, rk.blas_rank[mi][ti]);
                });
            }
        }
    });
    return best;
}

struct TraceCtx {
    const OrcScene& s;
    Counters& cn;
    const RankTables* ranks;   // non-null => stats mode
};

// --- diagnostics of a disagreement (stats mode only; not reference code): where do the boxes of the
// two winners lie along the ray?  A hit whose t is outside the [entry, exit] of a box that is supposed
// to bound it is what a t_max-shrinking walk can lose.
std::mutex g_mismatch_lock;
std::vector<OrcMismatch> g_mismatches;
uint32_t g_mismatch_cap = 0;

inline void slab_span(const AABB& b, const Ray& ray, float& entry, float& exit_) {
    // hit_aabb's arithmetic (implementation.rs:102-115) over (-inf, +inf), never rejecting
    const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z};
    const float d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
    const float mn[3] = {b.min.x, b.min.y, b.min.z};
    const float mx[3] = {b.max.x, b.max.y, b.max.z};
    float t_min = -F_INF, t_max = F_INF;
    for (int k = 0; k <= 2; k++) {
        float inv_d = 1.0f / d[k];
        float t0 = (mn[k] - o[k]) * inv_d, t1 = (mx[k] - o[k]) * inv_d;
        if (inv_d < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
    }
    entry = t_min; exit_ = t_max;
}

void describe_winner(const OrcScene& s, const Ray& ray, int32_t ent, int32_t tri, float* spans /* 6 */) {
    for (int k = 0; k < 6; k++) spans[k] = 0.0f;
    if (ent < 0) return;
    const Entity& e = s.entities[(size_t)ent];
    BvhBounds eb = entity_bounds(s, e);
    slab_span(eb.aabb, ray, spans[0], spans[1]);
    if (e.desc.hitable_kind != RT_HITABLE_MESH || tri < 0) return;
    Ray r = ray;
    for (const V3& t : e.translations) r.origin = r.origin - t;
    for (const RtRotation& rot : e.rotations) {
        r.origin = rotate_about_axis(r.origin, from(rot.axis), -rot.theta);
        r.direction = rotate_about_axis(r.direction, from(rot.axis), -rot.theta);
    }
    const MeshBvhRoot& m = s.meshes[s.mesh_objects[e.desc.hitable_index].mesh_index];
    slab_span(m.bvh.nodes[0].aabb, r, spans[2], spans[3]);
    slab_span(tri_bounds(m.tris[(size_t)tri]).aabb, r, spans[4], spans[5]);
}

void record_mismatch(const OrcScene& s, const Ray& ray, float t_min, float t_max, bool have, const HitRecord& best, const MinHit& m) {
    std::lock_guard<std::mutex> g(g_mismatch_lock);
    if (g_mismatches.size() >= g_mismatch_cap) return;
    OrcMismatch r{};
    r.o[0] = ray.origin.x; r.o[1] = ray.origin.y; r.o[2] = ray.origin.z;
    r.d[0] = ray.direction.x; r.d[1] = ray.direction.y; r.d[2] = ray.direction.z;
    r.t_min = t_min; r.t_max = t_max;
    r.ref_have = have ? 1 : 0; r.min_have = m.have ? 1 : 0;
    r.ref_t = have ? best.t : 0.0f; r.ref_ent = have ? best.entity_index : -1; r.ref_tri = have ? best.tri_index : -1;
    r.min_t = m.have ? m.t : 0.0f; r.min_ent = m.have ? m.entity : -1; r.min_tri = m.have ? m.tri : -1;
    describe_winner(s, ray, r.ref_ent, r.ref_tri, r.ref_spans);
    describe_winner(s, ray, r.min_ent, r.min_tri, r.min_spans);
    g_mismatches.push_back(r);
}

bool hit_closest(TraceCtx& cx, const Ray& ray, float t_min, float t_max, HitRecord& best) {
    bool have = hit_closest_ref(cx.s, ray, t_min, t_max, best, cx.cn);
    if (cx.ranks) {
        MinHit m = hit_closest_min(cx.s, *cx.ranks, ray, t_min, t_max, cx.cn);
        if (m.have != have || (have && (m.t != best.t || m.entity != best.entity_index || m.tri != best.tri_index))) {
            cx.cn.min_mismatch++;
            if (g_mismatch_cap) record_mismatch(cx.s, ray, t_min, t_max, have, best, m);
        }
    }
    return have;
}

// ---------------------------------------------------------------------------
// implementation.rs:16-36
// ---------------------------------------------------------------------------
const float PI = 3.14159265358979323846f;
const float TWO_PI = 2.0f * PI;
const float HALF_PI = PI / 2.0f;

V3 random_normal_reflection_angle(V3 normal, Rng& rng) {
    float theta1 = rng.next_f32() * HALF_PI;
    float theta2 = rng.next_f32() * TWO_PI;
    V3 p = cross(normal, POS_X);
    V3 perp = (p != ZERO) ? p : cross(normal, POS_Y);
    return rotate_about_axis(rotate_about_axis(normal, perp, theta1), normal, theta2);
}

// ---------------------------------------------------------------------------
// materials.rs
// ---------------------------------------------------------------------------
struct MatRecord { bool has_reflection; Ray refl_ray; float refl_intensity; bool has_refraction; Ray refr_ray; float refr_intensity; };

inline V3 reflect(V3 incident_direction, V3 surface_normal) {                          // materials.rs:84-87
    V3 dir = unit(incident_direction);
    return dir - (surface_normal * dot(dir, surface_normal) * 2.0f);
}
inline V3 refract(V3 v, V3 n, float ni_over_nt) {                                      // materials.rs:150-159
    V3 uv = unit(v);
    float dt = dot(uv, n);
    float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (discriminant <= 0.0f) return ZERO;
    return (uv - (n * dt)) * ni_over_nt - (n * sqrtf(discriminant));
}
inline float schlick_reflect_prob(float cosine, float ref_idx) {                       // materials.rs:161-165
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powf(1.0f - cosine, 5.0f);
}

MatRecord scatter(const RtMaterial& m, const Ray& ray, const HitRecord& hit, Rng& rng) {
    MatRecord rec{};
    switch (m.kind) {
        case RT_MAT_LAMBERTIAN: {                                                      // materials.rs:44-53
            V3 direction = random_normal_reflection_angle(hit.normal, rng);
            rec.has_reflection = true; rec.refl_ray = Ray{hit.p, direction}; rec.refl_intensity = m.reflectivity;
            break;
        }
        case RT_MAT_SPECULAR: {                                                        // materials.rs:89-112
            V3 reflected = reflect(ray.direction, hit.normal);
            V3 scattered = (m.fuzz == 0.0f) ? reflected
                                            : reflected + (unit(random_normal_reflection_angle(hit.normal, rng)) * m.fuzz);
            if (dot(scattered, hit.normal) > 0.0f) { rec.has_reflection = true; rec.refl_ray = Ray{hit.p, scattered}; rec.refl_intensity = m.reflectivity; }
            break;
        }
        default: {                                                                     // materials.rs:167-204
            float d = dot(ray.direction, hit.normal);
            V3 outward_normal; float ni_over_nt, cosine;
            if (d > 0.0f) { outward_normal = -hit.normal; ni_over_nt = m.ref_index; cosine = m.ref_index * d / length(ray.direction); }
            else { outward_normal = hit.normal; ni_over_nt = 1.0f / m.ref_index; cosine = -d / length(ray.direction); }
            float kr = schlick_reflect_prob(cosine, m.ref_index);
            if (!(kr >= 1.0f)) {
                V3 refraction_direction = unit(refract(ray.direction, outward_normal, ni_over_nt));
                rec.has_refraction = true; rec.refr_ray = Ray{hit.p, refraction_direction};
                rec.refr_intensity = (1.0f - kr) * (1.0f - m.opacity);
            }
            V3 reflection_direction = unit(reflect(ray.direction, hit.normal));
            rec.has_reflection = true; rec.refl_ray = Ray{hit.p, reflection_direction}; rec.refl_intensity = kr * m.reflectivity;
        }
    }
    return rec;
}

// ---------------------------------------------------------------------------
// texture.rs
// ---------------------------------------------------------------------------
const V3 NOT_FOUND_COLOR = {1.0f, 0.41f, 0.70f};                                       // texture.rs:124

inline size_t f32_as_usize(float f) {   // Rust `as usize`: saturating, NaN -> 0
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551615.0f) return (size_t)-1;
    return (size_t)f;
}
inline size_t map_to_rank(float p, size_t length) {                                    // texture.rs:87-91
    p = fmodf(p, 1.0f);
    p = p < 0.0f ? 1.0f + p : p;
    return f32_as_usize(p * (float)length);
}

V3 texture_value(const OrcScene& s, int32_t tex, const HitRecord& hit) {
    const RtTexture& t = s.textures[tex];
    switch (t.kind) {
        case RT_TEX_COLOR: return from(t.color);                                       // texture.rs:11-15
        case RT_TEX_CHECKER: {                                                         // texture.rs:32-45
            float sines = sinf(t.size * hit.uv.x) * sinf(t.size * hit.uv.y);
            return sines < 0.0f ? texture_value(s, t.child0, hit) : texture_value(s, t.child1, hit);
        }
        case RT_TEX_UV_TEST: return v3(hit.uv.x, hit.uv.y, 1.0f - hit.uv.x - hit.uv.y); // texture.rs:51-56
        case RT_TEX_XYZ_TEST: {                                                        // texture.rs:60-71
            float mx = t.size;
            return v3(0.5f + (hit.p.x / mx / 2.0f), 0.5f + (hit.p.y / mx / 2.0f), 0.5f + (hit.p.z / mx / 2.0f));
        }
        case RT_TEX_COLOR_MAP: {                                                       // texture.rs:82-99
            size_t x = map_to_rank(hit.uv.x, t.width);
            size_t y = map_to_rank(hit.uv.y, t.height);
            size_t offset = ((size_t)t.height - y) * (size_t)t.width + x;
            if (offset < (size_t)t.width * (size_t)t.height) return s.texels[t.texel_first + offset];
            return NOT_FOUND_COLOR;
        }
        case RT_TEX_MESH: {                                                            // texture.rs:109-116
            V3 diffuse = from(t.color);
            V3 rhs = t.child0 >= 0 ? texture_value(s, t.child0, hit) * diffuse : diffuse;
            return from(t.color2) + rhs;
        }
        default: {                                                                     // texture.rs:126-134
            if (hit.tex_key < 0 || hit.tex_key >= t.child1) return NOT_FOUND_COLOR;
            return texture_value(s, s.tex_set_entries[t.child0 + hit.tex_key], hit);
        }
    }
}

// ---------------------------------------------------------------------------
// lights.rs
// ---------------------------------------------------------------------------
struct LightRecord { float t; V3 direction, color; float intensity; };

bool light_record(const RtLight& l, V3 p, LightRecord& out) {
    switch (l.kind) {
        case RT_LIGHT_POINT: {                                                         // lights.rs:35-49
            V3 direction_to_p = p - from(l.origin);
            float t = length(direction_to_p);
            float intensity = l.intensity / (4.0f * PI * t);
            out = LightRecord{t, unit(direction_to_p), from(l.color), intensity};
            return true;
        }
        case RT_LIGHT_LAMP: {                                                          // lights.rs:87-109
            V3 direction_to_p = p - from(l.origin);
            float theta = theta_between(direction_to_p, from(l.direction));
            float theta_deg = theta / PI * 180.0f;
            if (theta_deg > l.angle_deg) return false;
            float t = length(direction_to_p);
            float intensity = l.intensity / (4.0f * PI * t);
            out = LightRecord{t, unit(direction_to_p), from(l.color), intensity};
            return true;
        }
        default:                                                                       // lights.rs:138-148
            out = LightRecord{F32_MAX, from(l.direction), from(l.color), l.intensity};
            return true;
    }
}

// ---------------------------------------------------------------------------
// implementation.rs:498-661  core routine
// ---------------------------------------------------------------------------
const float BIAS = 0.004f;

V3 color_sky(const Ray& ray, const OrcScene& s) {                                      // :502-519
    if (s.sky == RT_SKY_BLACK) return ZERO;
    V3 unit_direction = unit(ray.direction);
    float t = 0.5f * (unit_direction.y + 1.0f);
    V3 white = v3(1.0f, 1.0f, 1.0f), sky_blue = v3(0.5f, 0.7f, 1.0f);
    return white * (1.0f - t) + (sky_blue * t);
}

// :524-553
V3 cast_light_ray_to_lamp(V3 hit_point, const LightRecord& lr, TraceCtx& cx, Rng& rng, uint32_t node) {
    Ray light_ray{hit_point, -lr.direction};
    float t_max = lr.t;
    V3 light_color = lr.color * lr.intensity;
    float closest_so_far = BIAS;
    HitRecord shadow_hit;
    for (;;) {
        cx.cn.shadow++;
        if (!hit_closest(cx, light_ray, closest_so_far, t_max, shadow_hit)) break;
        rng.set_ctx(node, PURPOSE_SHADOW);
        MatRecord shadow_mat = scatter(cx.s.materials[shadow_hit.mat_id], light_ray, shadow_hit, rng);
        if (shadow_mat.has_refraction) {
            V3 albedo = texture_value(cx.s, (int32_t)shadow_hit.tex_id, shadow_hit);
            light_color = light_color * (albedo * shadow_mat.refr_intensity);
            closest_so_far = shadow_hit.t + BIAS;
            // NOT the reference: beyond t = 65 536 the sum rounds back to t (ulp(t) > 2 BIAS), a plane or triangle accepts t == t_min
            // (plane.rs:14, mesh.rs:28), and the reference's loop (:535-549) finds the same occluder for ever - it never returns
            // the pixel.  The product cannot restate a hang: both it and this oracle step one ulp past the occluder instead.
            if (!(closest_so_far > shadow_hit.t)) closest_so_far = std::nextafterf(shadow_hit.t, INFINITY);
            if (!(closest_so_far > shadow_hit.t)) return ZERO;      // (t = inf: nothing lies further on)
            continue;
        }
        return ZERO;
    }
    return light_color;
}

// :559-633.  `node` is the position in the recursion tree (root 1, reflection
// 2n, refraction 2n+1); only the counter RNG looks at it.
V3 cast_ray_recursive(const Ray& ray, TraceCtx& cx, Rng& rng, uint32_t recurse_limit, uint32_t node) {
    const OrcScene& scene = cx.s;
    if (recurse_limit == 0) return color_sky(ray, scene);
    if (node == 1) cx.cn.primary++; else cx.cn.secondary++;
    HitRecord hit_record;
    if (!hit_closest(cx, ray, BIAS, F32_MAX, hit_record)) return color_sky(ray, scene);

    rng.set_ctx(node, PURPOSE_PATH);
    MatRecord mat_record = scatter(scene.materials[hit_record.mat_id], ray, hit_record, rng);

    uint32_t reflect_limit, refract_limit;
    {
        uint32_t rl = recurse_limit - 1;
        if (mat_record.has_reflection && mat_record.has_refraction) { reflect_limit = rl / 2; refract_limit = rl - reflect_limit; }
        else if (mat_record.has_reflection) { reflect_limit = rl; refract_limit = 0; }
        else if (mat_record.has_refraction) { reflect_limit = 0; refract_limit = rl; }
        else { reflect_limit = 0; refract_limit = 0; }
    }
    V3 color_from_reflection = ZERO; float reflection_intensity = 0.0f;
    if (mat_record.has_reflection && mat_record.refl_intensity > 0.0f) {
        color_from_reflection = cast_ray_recursive(mat_record.refl_ray, cx, rng, reflect_limit, node * 2);
        reflection_intensity = mat_record.refl_intensity;
    }
    V3 color_from_refraction = ZERO; float refraction_intensity = 0.0f;
    if (mat_record.has_refraction && mat_record.refr_intensity > 0.0f) {
        color_from_refraction = cast_ray_recursive(mat_record.refr_ray, cx, rng, refract_limit, node * 2 + 1);
        refraction_intensity = mat_record.refr_intensity;
    }
    V3 hit_point = hit_record.p + (hit_record.normal * BIAS);
    V3 color_from_lights = ZERO;
    for (const RtLight& light : scene.lights) {
        LightRecord lr;
        if (light_record(light, hit_point, lr)) {
            V3 light_color = cast_light_ray_to_lamp(hit_point, lr, cx, rng, node) *
                             fmaxf(0.0f, dot(hit_record.normal, -lr.direction));
            color_from_lights = color_from_lights + light_color;
        }
    }
    float lights_intensity = fmaxf(0.0f, 1.0f - (reflection_intensity + refraction_intensity));
    V3 albedo = texture_value(scene, (int32_t)hit_record.tex_id, hit_record);
    return ((color_from_reflection * reflection_intensity) +
            (color_from_refraction * refraction_intensity) +
            (color_from_lights * lights_intensity)) * albedo;
}

inline V3 cast_ray(const Ray& ray, TraceCtx& cx, Rng& rng, uint32_t max_reflections) {  // :556,:635
    return clamp(cast_ray_recursive(ray, cx, rng, max_reflections, 1));
}

inline Ray camera_get_ray(const RtCamera& c, float x, float y, V2 lens_deflection) {   // :482-491
    float ldx = lens_deflection.x * c.lens_radius;
    float ldy = lens_deflection.y * c.lens_radius;
    V3 offset = (from(c.u) * ldx) + (from(c.v) * ldy);
    V3 origin = from(c.origin) + offset;
    V3 direction = from(c.lower_left_corner) + (from(c.horizontal) * x) + (from(c.vertical) * y) - from(c.origin) - offset;
    return Ray{origin, direction};
}

// one sample of cast_rays_into_scene's loop body, :645-656
V3 cast_one_sample(TraceCtx& cx, const RtSettings& st, uint32_t x, uint32_t y, Rng& rng) {
    float u = (float)x / (float)st.width;
    float v = (float)(st.height - y) / (float)st.height;
    V2 lens_deflection{0.0f, 0.0f};
    if (st.samples_per_pixel > 1) {
        rng.set_ctx(0, PURPOSE_LENS);
        float r0 = rng.next_f32(), r1 = rng.next_f32();
        // ORC_RNG_XORSHIFT_CHUNK and ORC_RNG_COUNTER stop here: the reference uses the two draws as they are (:649-652).
        // ORC_RNG_COUNTER_STRATIFIED - NOT reference semantics, the product's default sampler (RT_OPT_LENS_SAMPLER 1):
        // the lens samples are stratified: inside complete groups of 64 samples, sample s lies in one of 8 x 8 cells over the
        // lens square (s mod 16 names the 4 x 4 cell, bits 4 and 5 of s the quarter of it); after them, in cell s mod 16 of a
        // 4 x 4 grid inside complete groups of 16 samples (s mod 8 of 4 x 2 cells in a complete segment of 8 after those),
        // jittered inside the cell.  Same expectation, less noise.
        if (!rng.stratified) { }
        else if ((rng.sample / 64u + 1u) * 64u <= st.samples_per_pixel) {
            const uint32_t s = rng.sample;                   // complete group of 64 samples: 8 x 8 cells
            const uint32_t cx = 2u * (s & 3u) + ((s >> 4) & 1u), cy = 2u * ((s >> 2) & 3u) + ((s >> 5) & 1u);
            r0 = ((float)cx + r0) * 0.125f;
            r1 = ((float)cy + r1) * 0.125f;
        } else if ((rng.sample / 16u + 1u) * 16u <= st.samples_per_pixel) {
            const uint32_t cell = rng.sample & 15u;          // complete group of 16 samples: 4 x 4 cells
            r0 = ((float)(cell & 3u) + r0) * 0.25f;
            r1 = ((float)(cell >> 2) + r1) * 0.25f;
        } else if ((rng.sample / 8u + 1u) * 8u <= st.samples_per_pixel) {
            const uint32_t cell = rng.sample & 7u;           // complete segment of 8 after them: 4 x 2 cells
            r0 = ((float)(cell & 3u) + r0) * 0.25f;
            r1 = ((float)(cell >> 2) + r1) * 0.5f;
        }
        float a = r0 * 2.0f - 1.0f;
        float b = r1 * 2.0f - 1.0f;
        lens_deflection = V2{a, b};
    }
    Ray ray = camera_get_ray(cx.s.camera, u, v, lens_deflection);
    cx.cn.samples++;
    return cast_ray(ray, cx, rng, st.max_reflections);
}

// :638-661
V3 cast_rays_into_scene(TraceCtx& cx, const RtSettings& st, uint32_t x, uint32_t y, Rng& rng, uint32_t sum_segment) {
    V3 col = v3(0.0f, 0.0f, 0.0f);
    if (sum_segment == 0) {
        for (uint32_t s = 0; s < st.samples_per_pixel; s++) {
            rng.set_pixel_sample(x, y, s);
            col = col + cast_one_sample(cx, st, x, y, rng);
        }
    } else {
        for (uint32_t s0 = 0; s0 < st.samples_per_pixel; s0 += sum_segment) {
            V3 seg = v3(0.0f, 0.0f, 0.0f);
            for (uint32_t s = s0; s < s0 + sum_segment && s < st.samples_per_pixel; s++) {
                rng.set_pixel_sample(x, y, s);
                seg = seg + cast_one_sample(cx, st, x, y, rng);
            }
            col = col + seg;
        }
    }
    return col / (float)st.samples_per_pixel;
}

inline uint8_t f32_as_u8(float f) {   // Rust `as u8`: saturating, truncating, NaN -> 0
    if (!(f > 0.0f)) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}
inline void v3_to_rgba(V3 c, uint8_t* out) {                                           // src/rgba.rs:6-12
    out[0] = f32_as_u8(255.0f * c.x); out[1] = f32_as_u8(255.0f * c.y); out[2] = f32_as_u8(255.0f * c.z); out[3] = 255;
}

void rng_init(Rng& rng, int mode, uint64_t seed) {
    rng.stratified = mode == ORC_RNG_COUNTER_STRATIFIED;
    if (mode == ORC_RNG_COUNTER_STRATIFIED) mode = ORC_RNG_COUNTER;
    rng.mode = mode;
    if (mode == ORC_RNG_XORSHIFT_CHUNK) rng.xs.seed_from_u64(seed);
    else { rng.key[0] = (uint32_t)seed; rng.key[1] = (uint32_t)(seed >> 32); }
}

// src/render.rs:142-159 for one chunk.  Outputs are chunk-local row-major unless
// frame_pitch != 0 (then written at the chunk's position in a full frame).
void render_chunk_impl(const OrcScene& s, const RtSettings& st, const RtRenderChunk& ch, int rng_mode, uint64_t seed,
                       uint32_t sum_segment, const RankTables* ranks, float* rgb, uint8_t* rgba, uint32_t frame_pitch, Counters& cn) {
    Rng rng; rng_init(rng, rng_mode, seed);
    TraceCtx cx{s, cn, ranks};
    for (uint32_t y = 0; y < ch.height; y++) {
        for (uint32_t x = 0; x < ch.width; x++) {
            V3 color = cast_rays_into_scene(cx, st, ch.left + x, ch.top + y, rng, sum_segment);
            size_t idx = frame_pitch ? ((size_t)(ch.top + y) * frame_pitch + ch.left + x) : ((size_t)y * ch.width + x);
            if (rgb) { rgb[idx * 3] = color.x; rgb[idx * 3 + 1] = color.y; rgb[idx * 3 + 2] = color.z; }
            if (rgba) v3_to_rgba(color, rgba + idx * 4);
        }
    }
}

RankTables make_ranks(const OrcScene& s) {
    RankTables rk;
    std::vector<uint32_t> order;
    bvh_emission_order(s.tlas, order);
    rk.tlas_rank.resize(order.size());
    for (size_t i = 0; i < order.size(); i++) rk.tlas_rank[order[i]] = (uint32_t)i;
    rk.blas_rank.resize(s.meshes.size());
    for (size_t m = 0; m < s.meshes.size(); m++) {
        order.clear(); bvh_emission_order(s.meshes[m].bvh, order);
        rk.blas_rank[m].resize(order.size());
        for (size_t i = 0; i < order.size(); i++) rk.blas_rank[m][order[i]] = (uint32_t)i;
    }
    return rk;
}

std::vector<RtRenderChunk> create_render_chunks(uint32_t w, uint32_t h, uint32_t cx, uint32_t cy) {  // viewport.rs:30-54
    struct Segment { uint32_t offset, size; };
    auto dim_segments = [](uint32_t length, uint32_t count) {
        std::vector<Segment> v;
        uint32_t size = length / count, remainder = length % count;
        for (uint32_t i = 0; i < count; i++) v.push_back(Segment{i * size, i == count - 1 ? size + remainder : size});
        return v;
    };
    std::vector<RtRenderChunk> out;
    for (const Segment& v : dim_segments(h, cy))
        for (const Segment& hs : dim_segments(w, cx))
            out.push_back(RtRenderChunk{hs.offset, v.offset, hs.size, v.size});
    return out;
}

bool validate_settings(const RtSettings* st, const RtRenderChunk* ch) {
    if (!st || st->width == 0 || st->height == 0 || st->samples_per_pixel == 0) return false;
    if (st->width > 65535 || st->height > 65535 || st->max_reflections > 31) return false;
    if (ch && ((uint64_t)ch->left + ch->width > st->width || (uint64_t)ch->top + ch->height > st->height)) return false;
    return true;
}

}  // namespace

// ===========================================================================
// C interface
// ===========================================================================
extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

OrcScene* orc_scene_create(const RtSceneDesc* d) {
    if (!d || d->abi_version != RT_ABI_VERSION) { fail(RT_E_INVALID, "bad scene description"); return nullptr; }
    OrcScene* s = new OrcScene();
    s->camera = d->camera; s->sky = d->sky;
    s->spheres.assign(d->spheres, d->spheres + d->sphere_count);
    s->planes.assign(d->planes, d->planes + d->plane_count);
    s->mesh_objects.assign(d->mesh_objects, d->mesh_objects + d->mesh_object_count);
    s->materials.assign(d->materials, d->materials + d->material_count);
    s->textures.assign(d->textures, d->textures + d->texture_count);
    s->tex_set_entries.assign(d->tex_set_entries, d->tex_set_entries + d->tex_set_entry_count);
    s->texels.resize(d->texel_count);
    for (uint64_t i = 0; i < d->texel_count; i++) s->texels[i] = from(d->texels[i]);
    s->lights.assign(d->lights, d->lights + d->light_count);
    // MeshBvhRoot::new, mesh.rs:84-90
    s->meshes.resize(d->mesh_count);
    for (uint64_t m = 0; m < d->mesh_count; m++) {
        if (d->meshes[m].tri_count == 0) { delete s; fail(RT_E_INVALID, "mesh without triangles"); return nullptr; }
        s->meshes[m].tris.assign(d->meshes[m].tris, d->meshes[m].tris + d->meshes[m].tri_count);
        std::vector<BvhObjectBounds> ob(s->meshes[m].tris.size());
        for (size_t i = 0; i < ob.size(); i++) ob[i] = BvhObjectBounds{i, tri_bounds(s->meshes[m].tris[i])};
        s->meshes[m].bvh = bvh_build(std::move(ob));
    }
    s->entities.resize(d->entity_count);
    for (uint64_t i = 0; i < d->entity_count; i++) {
        Entity& e = s->entities[i];
        e.desc = d->entities[i];
        for (uint32_t k = 0; k < e.desc.translation_count; k++) e.translations.push_back(from(d->translations[e.desc.translation_first + k]));
        for (uint32_t k = 0; k < e.desc.rotation_count; k++) e.rotations.push_back(d->rotations[e.desc.rotation_first + k]);
    }
    if (d->entity_count == 0) { delete s; fail(RT_E_INVALID, "scene without entities"); return nullptr; }
    // Scene::build_bvh, implementation.rs:377-383
    std::vector<BvhObjectBounds> ob(s->entities.size());
    for (size_t i = 0; i < ob.size(); i++) ob[i] = BvhObjectBounds{i, entity_bounds(*s, s->entities[i])};
    s->tlas = bvh_build(std::move(ob));
    return s;
}

void orc_scene_destroy(OrcScene* s) { delete s; }

int orc_render_chunk(const OrcScene* s, const RtSettings* st, const RtRenderChunk* ch, int rng_mode, uint64_t seed,
                     uint32_t sum_segment, int stats, float* rgb, uint8_t* rgba, OrcCounters* counters) {
    if (!s || !validate_settings(st, ch)) return fail(RT_E_INVALID, "orc_render_chunk: bad arguments");
    Counters cn;
    try {
        RankTables rk; if (stats) rk = make_ranks(*s);
        render_chunk_impl(*s, *st, *ch, rng_mode, seed, sum_segment, stats ? &rk : nullptr, rgb, rgba, 0, cn);
    } catch (StackOverflow&) { return fail(RT_E_DEPTH, "BVH traversal stack overflow (ArrayVec<State,30>)"); }
    if (counters) cn.add_to(counters);
    return RT_OK;
}

int orc_accumulate_chunk(const OrcScene* s, const RtSettings* st, const RtRenderChunk* ch, int rng_mode, uint64_t seed,
                         uint32_t sample_first, uint32_t sample_count, float* accum) {
    if (!s || !validate_settings(st, ch) || !accum || (rng_mode != ORC_RNG_COUNTER && rng_mode != ORC_RNG_COUNTER_STRATIFIED))
        return fail(RT_E_INVALID, "orc_accumulate_chunk: bad arguments");
    Counters cn; Rng rng; rng_init(rng, rng_mode, seed);
    TraceCtx cx{*s, cn, nullptr};
    try {
        for (uint32_t y = 0; y < ch->height; y++)
            for (uint32_t x = 0; x < ch->width; x++) {
                V3 col = v3(0, 0, 0);
                for (uint32_t k = 0; k < sample_count; k++) {
                    rng.set_pixel_sample(ch->left + x, ch->top + y, sample_first + k);
                    col = col + cast_one_sample(cx, *st, ch->left + x, ch->top + y, rng);
                }
                size_t idx = (size_t)y * ch->width + x;
                accum[idx * 3] = col.x; accum[idx * 3 + 1] = col.y; accum[idx * 3 + 2] = col.z;
            }
    } catch (StackOverflow&) { return fail(RT_E_DEPTH, "BVH traversal stack overflow"); }
    return RT_OK;
}

int orc_render_chunks_mt(const OrcScene* s, const RtSettings* st, uint32_t count_x, uint32_t count_y,
                         uint32_t chunk_first, uint32_t chunk_stride, uint32_t chunk_count, uint32_t threads, int rng_mode, uint64_t seed,
                         int stats, OrcCounters* counters, double* elapsed_ms) {
    if (!s || !validate_settings(st, nullptr) || count_x == 0 || count_y == 0 || threads == 0)
        return fail(RT_E_INVALID, "orc_render_chunks_mt: bad arguments");
    std::vector<RtRenderChunk> chunks = create_render_chunks(st->width, st->height, count_x, count_y);
    if (chunk_stride == 0 || chunk_count == 0 || (uint64_t)chunk_first + (uint64_t)(chunk_count - 1) * chunk_stride >= chunks.size())
        return fail(RT_E_INVALID, "chunk sample out of bounds");
    RankTables rk; if (stats) rk = make_ranks(*s);
    std::atomic<uint32_t> next{0}; std::atomic<int> err{0};
    std::vector<Counters> per_thread(threads);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < threads; t++) pool.emplace_back([&, t]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= chunk_count) break;
            try { render_chunk_impl(*s, *st, chunks[chunk_first + (size_t)i * chunk_stride], rng_mode, seed, 0, stats ? &rk : nullptr, nullptr, nullptr, 0, per_thread[t]); }
            catch (StackOverflow&) { err = RT_E_DEPTH; break; }
        }
    });
    for (auto& th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (err) return fail(err, "BVH traversal stack overflow");
    if (counters) for (auto& c : per_thread) c.add_to(counters);
    return RT_OK;
}

int orc_render_image_mt(const OrcScene* s, const RtSettings* st, uint32_t count_x, uint32_t count_y, uint32_t threads,
                        int rng_mode, uint64_t seed, float* rgb, uint8_t* rgba, OrcCounters* counters, double* elapsed_ms) {
    if (!s || !validate_settings(st, nullptr) || count_x == 0 || count_y == 0 || threads == 0)
        return fail(RT_E_INVALID, "orc_render_image_mt: bad arguments");
    std::vector<RtRenderChunk> chunks = create_render_chunks(st->width, st->height, count_x, count_y);
    std::atomic<uint32_t> next{0}; std::atomic<int> err{0};
    std::vector<Counters> per_thread(threads);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < threads; t++) pool.emplace_back([&, t]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= chunks.size()) break;
            try { render_chunk_impl(*s, *st, chunks[i], rng_mode, seed, 0, nullptr, rgb, rgba, st->width, per_thread[t]); }
            catch (StackOverflow&) { err = RT_E_DEPTH; break; }
        }
    });
    for (auto& th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (err) return fail(err, "BVH traversal stack overflow");
    if (counters) for (auto& c : per_thread) c.add_to(counters);
    return RT_OK;
}

int orc_primary_hits(const OrcScene* s, const RtSettings* st, const RtRenderChunk* ch,
                     float* t_out, float* normal_out, int32_t* entity_out, int32_t* tri_out) {
    if (!s || !validate_settings(st, ch)) return fail(RT_E_INVALID, "orc_primary_hits: bad arguments");
    Counters cn;
    try {
        for (uint32_t y = 0; y < ch->height; y++)
            for (uint32_t x = 0; x < ch->width; x++) {
                uint32_t gx = ch->left + x, gy = ch->top + y;
                float u = (float)gx / (float)st->width;
                float v = (float)(st->height - gy) / (float)st->height;
                Ray ray = camera_get_ray(s->camera, u, v, V2{0.0f, 0.0f});
                HitRecord h;
                bool ok = hit_closest_ref(*s, ray, BIAS, F32_MAX, h, cn);
                size_t idx = (size_t)y * ch->width + x;
                if (t_out) t_out[idx] = ok ? h.t : -1.0f;
                if (normal_out) { normal_out[idx * 3] = ok ? h.normal.x : 0; normal_out[idx * 3 + 1] = ok ? h.normal.y : 0; normal_out[idx * 3 + 2] = ok ? h.normal.z : 0; }
                if (entity_out) entity_out[idx] = ok ? h.entity_index : -1;
                if (tri_out) tri_out[idx] = ok ? h.tri_index : -1;
            }
    } catch (StackOverflow&) { return fail(RT_E_DEPTH, "BVH traversal stack overflow"); }
    return RT_OK;
}

void orc_set_cull_slack(float factor) { g_cull_slack = factor >= 1.0f ? factor : ORC_CULL_SLACK; }

void orc_mismatch_capture(uint32_t capacity) {
    std::lock_guard<std::mutex> g(g_mismatch_lock);
    g_mismatch_cap = capacity; g_mismatches.clear();
}

uint32_t orc_mismatch_read(OrcMismatch* out, uint32_t capacity) {
    std::lock_guard<std::mutex> g(g_mismatch_lock);
    uint32_t n = 0;
    for (; n < g_mismatches.size() && n < capacity; n++) out[n] = g_mismatches[n];
    return n;
}

int64_t orc_scene_bvh_order(const OrcScene* s, int64_t mesh_index, uint32_t* out, uint64_t capacity) {
    if (!s) return RT_E_INVALID;
    if (mesh_index >= (int64_t)s->meshes.size()) return RT_E_INVALID;
    std::vector<uint32_t> order;
    bvh_emission_order(mesh_index < 0 ? s->tlas : s->meshes[mesh_index].bvh, order);
    if (out) for (size_t i = 0; i < order.size() && i < capacity; i++) out[i] = order[i];
    return (int64_t)order.size();
}

int orc_scene_bvh_depth(const OrcScene* s, int64_t mesh_index) {
    if (!s || mesh_index >= (int64_t)s->meshes.size()) return RT_E_INVALID;
    return bvh_depth(mesh_index < 0 ? s->tlas : s->meshes[mesh_index].bvh);
}

// ---- known-answer hooks ---------------------------------------------------
int orc_partition_by_key_i32(int32_t* data, int n, int32_t split_on) {
    if (n <= 0) return -1;   // util.rs:10-12 panics
    return (int)partition_by_key(data, (size_t)n, split_on, [](int32_t v) { return v; });
}

int orc_create_render_chunks(uint32_t w, uint32_t h, uint32_t cx, uint32_t cy, RtRenderChunk* out, uint32_t capacity) {
    if (cx == 0 || cy == 0) return RT_E_INVALID;
    std::vector<RtRenderChunk> c = create_render_chunks(w, h, cx, cy);
    if (c.size() > capacity) return RT_E_INVALID;
    for (size_t i = 0; i < c.size(); i++) out[i] = c[i];
    return (int)c.size();
}

static Ray mkray(const float* o, const float* d) { return Ray{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])}; }

int orc_intersect_sphere(const float* origin, const float* dir, float radius, float* ts) {
    return intersect_sphere(mkray(origin, dir), ZERO, radius, ts) ? 1 : 0;
}

float orc_sphere_hit(const float* origin, const float* dir, float radius, float t_min, float t_max, float* normal_out) {
    HitRecord h; RtSphere s{radius, 0, 0};
    if (!sphere_hit(s, mkray(origin, dir), t_min, t_max, h)) return -1.0f;
    if (normal_out) { normal_out[0] = h.normal.x; normal_out[1] = h.normal.y; normal_out[2] = h.normal.z; }
    return h.t;
}

int orc_try_hit_tri(const float* origin, const float* dir, float t_min, float t_max, const RtMeshTri* tri, float* out9) {
    MeshTriHit h;
    if (!try_hit_tri(mkray(origin, dir), t_min, t_max, *tri, h)) return 0;
    out9[0] = h.t; out9[1] = h.p.x; out9[2] = h.p.y; out9[3] = h.p.z;
    out9[4] = h.normal.x; out9[5] = h.normal.y; out9[6] = h.normal.z; out9[7] = h.uv.x; out9[8] = h.uv.y;
    return 1;
}

int orc_intersect_plane(const float* origin, const float* dir, const float* po, const float* n, float* t) {
    return intersect_plane(mkray(origin, dir), v3(po[0], po[1], po[2]), v3(n[0], n[1], n[2]), *t) ? 1 : 0;
}

int orc_hit_aabb(const float* bmin, const float* bmax, const float* origin, const float* dir, float t_min, float t_max) {
    AABB b{v3(bmin[0], bmin[1], bmin[2]), v3(bmax[0], bmax[1], bmax[2])};
    return hit_aabb(b, mkray(origin, dir), t_min, t_max) ? 1 : 0;
}

void orc_reflect(const float* i, const float* n, float* out) {
    V3 r = reflect(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_refract(const float* v, const float* n, float ni_over_nt, float* out) {
    V3 r = refract(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]), ni_over_nt); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_schlick(float cosine, float ref_idx) { return schlick_reflect_prob(cosine, ref_idx); }

void orc_color_sky_day(const float* dir, float* out) {
    OrcScene s; s.sky = RT_SKY_DAY;
    V3 c = color_sky(Ray{ZERO, v3(dir[0], dir[1], dir[2])}, s); out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

void orc_rotate_about_axis(const float* p, const float* axis, float theta, float* out) {
    V3 r = rotate_about_axis(v3(p[0], p[1], p[2]), v3(axis[0], axis[1], axis[2]), theta); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void orc_v3_to_rgba(const float* rgb, uint8_t* out) { v3_to_rgba(v3(rgb[0], rgb[1], rgb[2]), out); }

void orc_camera_new(const float* lf, const float* la, float v_fov, float aspect_ratio, float lens_radius, float focus_dist, RtCamera* out) {
    // implementation.rs:462-480
    V3 look_from = v3(lf[0], lf[1], lf[2]), look_at = v3(la[0], la[1], la[2]);
    V3 v_up = v3(0.0f, 1.0f, 0.0f);
    float theta = v_fov * PI / 180.0f;
    float half_height = tanf(theta / 2.0f);
    float half_width = aspect_ratio * half_height;
    V3 w = unit(look_from - look_at);
    V3 u = unit(cross(v_up, w));
    V3 v = cross(w, u);
    V3 llc = look_from - (u * half_width * focus_dist) - (v * half_height * focus_dist) - (w * focus_dist);
    V3 horizontal = u * (2.0f * half_width * focus_dist);
    V3 vertical = v * (2.0f * half_height * focus_dist);
    auto to = [](V3 a) { return RtV3{a.x, a.y, a.z}; };
    out->lower_left_corner = to(llc); out->horizontal = to(horizontal); out->vertical = to(vertical);
    out->origin = to(look_from); out->u = to(u); out->v = to(v); out->lens_radius = lens_radius;
}

void orc_camera_get_ray(const RtCamera* cam, float x, float y, float lx, float ly, float* o, float* d) {
    Ray r = camera_get_ray(*cam, x, y, V2{lx, ly});
    o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z; d[0] = r.direction.x; d[1] = r.direction.y; d[2] = r.direction.z;
}

int orc_light_record(const RtLight* light, const float* p, float* out8) {
    LightRecord lr;
    if (!light_record(*light, v3(p[0], p[1], p[2]), lr)) return 0;
    out8[0] = lr.t; out8[1] = lr.direction.x; out8[2] = lr.direction.y; out8[3] = lr.direction.z;
    out8[4] = lr.color.x; out8[5] = lr.color.y; out8[6] = lr.color.z; out8[7] = lr.intensity;
    return 1;
}

void orc_xorshift_from_seed(const uint8_t* seed16, uint32_t* out, int n) {
    XorShift x; x.from_seed(seed16);
    for (int i = 0; i < n; i++) out[i] = x.next_u32();
}
void orc_xorshift_seed_from_u64(uint64_t seed, uint32_t* out, int n) {
    XorShift x; x.seed_from_u64(seed);
    for (int i = 0; i < n; i++) out[i] = x.next_u32();
}
float orc_u32_to_f32(uint32_t v) { return u32_to_f32(v); }
void orc_philox4x32_10(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) { philox4x32_10(ctr4, key2, out4); }

float orc_counter_rng_f32(uint64_t seed, uint32_t x, uint32_t y, uint32_t sample, uint32_t node, uint32_t purpose, uint32_t k) {
    Rng rng; rng_init(rng, ORC_RNG_COUNTER, seed);
    rng.set_pixel_sample(x, y, sample);
    rng.set_ctx(node, purpose);
    float f = 0;
    for (uint32_t i = 0; i <= k; i++) f = rng.next_f32();
    return f;
}

}  // extern "C"
