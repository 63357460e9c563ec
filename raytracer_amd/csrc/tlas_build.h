// Device-side BUILD of the top-level tree (included by render.hip, which provides Word4 / F3 / ldw / DevNode).
//
// The reference rebuilds its entity BVH on the CPU for every frame (Scene::build_bvh, implementation.rs:377-383; bvh.rs:129-188:
// mean split of the longest axis).  rt_scene_update_transforms keeps the topology and refits the boxes; this file is the other
// half: a NEW topology over the moved entities, built where the boxes are - a linear BVH over the Morton codes of the box centres
// (Karras 2012: one thread per inner node finds its range and split from the sorted keys; boxes bottom-up with one atomic
// counter per node).  The tree differs from the reference's, so it sits behind the nearest-hit gate (tests/test_gpu_rebuild.py),
// not the bit-exact one: a candidate is now pruned by the boxes of THIS tree.  What stays the reference's: every entity's own
// box (refit_entities_kernel: Entity::calculate_origin_aabb), the visit ranks that decide exact ties (they travel with the
// entity records), and everything below the entities (BLASes are untouched).
//
// Stages, all on the caller's stream:
//   keys      64-bit key = 30-bit Morton code of the box centre (10 bits per axis over the bounds of all centres) << 32 | position
//   sort      hipcub::DeviceRadixSort (rocPRIM)
//   permute   entity records and boxes into key order (leaves of the tree are ranges of positions, like the host build's)
//   topology  lbvh_topology_kernel
//   boxes     lbvh_boxes_kernel (+ subtree height, for the traversal stack bound)
//   number    lbvh_number_kernel: the first records in breadth-first order go to the TLAS slots of the LDS node cache
//   emit      lbvh_emit_kernel: two-box records of device_scene.h; ranges of <= RT_LBVH_LEAF entities become leaf refs
#ifndef RT_TLAS_BUILD_H
#define RT_TLAS_BUILD_H

#define RT_LBVH_LEAF 2u          /* entities per leaf (the reference's trees end in leaves of up to two, bvh.rs:144-150) */
#define RT_LBVH_NONE 0xffffffffu

struct LbvhNode {               // inner node i of the Karras tree over n sorted keys (n - 1 of them)
    uint32_t left, right;       // child: bit 31 set = leaf (sorted position), else inner node index
    uint32_t first, last;       // range of sorted positions it covers
};

__device__ __forceinline__ uint32_t morton_spread10(uint32_t v) {      // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// bounds of the box centres (finite boxes only: a plane's box is infinite and gets code 0).  One block.
__global__ void lbvh_bounds_kernel(const float* ent_box, uint32_t n, float* bounds /* lo.xyz hi.xyz */) {
    __shared__ float red[6][256];
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float* b = ent_box + (size_t)i * 8;
        for (int k = 0; k < 3; k++) {
            const float c = 0.5f * b[k] + 0.5f * b[4 + k];
            if (c - c == 0.0f) { lo[k] = fminf(lo[k], c); hi[k] = fmaxf(hi[k], c); }      // finite
        }
    }
    for (int k = 0; k < 3; k++) { red[k][threadIdx.x] = lo[k]; red[3 + k][threadIdx.x] = hi[k]; }
    __syncthreads();
    for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 3; k++) {
                red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + s]);
                red[3 + k][threadIdx.x] = fmaxf(red[3 + k][threadIdx.x], red[3 + k][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) bounds[threadIdx.x] = red[threadIdx.x][0];
}

__global__ void lbvh_keys_kernel(const float* ent_box, uint32_t n, const float* bounds, unsigned long long* keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* b = ent_box + (size_t)i * 8;
    uint32_t code = 0;
    for (int k = 0; k < 3; k++) {
        const float c = 0.5f * b[k] + 0.5f * b[4 + k], lo = bounds[k], ext = bounds[3 + k] - bounds[k];
        uint32_t q = 0;
        if (c - c == 0.0f && ext > 0.0f) q = (uint32_t)fminf(fmaxf((c - lo) / ext * 1024.0f, 0.0f), 1023.0f);
        code |= morton_spread10(q) << (2 - k);
    }
    keys[i] = ((unsigned long long)code << 32) | i;
}

// entity records and boxes into key order; `order` (optional) keeps the permutation for the host's bookkeeping
__global__ void lbvh_permute_kernel(const unsigned long long* keys, uint32_t n, const DevEntity* ent_in, DevEntity* ent_out,
                                    const float* box_in, float* box_out, uint32_t* order) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t src = (uint32_t)keys[j];
    for (int k = 0; k < 3; k++) *reinterpret_cast<float4*>(ent_out[j].w + k) = *reinterpret_cast<const float4*>(ent_in[src].w + k);
    for (int k = 0; k < 2; k++) *reinterpret_cast<float4*>(box_out + (size_t)j * 8 + 4 * k) = *reinterpret_cast<const float4*>(box_in + (size_t)src * 8 + 4 * k);
    if (order) order[j] = src;
}

// length of the common prefix of keys i and j (keys are distinct: the position is part of them); -1 outside the array
__device__ __forceinline__ int lbvh_delta(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// Karras, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees" (HPG 2012), section 4
__global__ void lbvh_topology_kernel(const unsigned long long* keys, uint32_t n, LbvhNode* nodes, uint32_t* parent /* [0, n-1): inner; [n-1, 2n-1): leaves */) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x), nn = (int)n;
    if (i >= nn - 1) return;
    const int d = lbvh_delta(keys, nn, i, i + 1) - lbvh_delta(keys, nn, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, nn, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, nn, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, nn, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, nn, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (lbvh_delta(keys, nn, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    LbvhNode nd;
    nd.first = (uint32_t)lo; nd.last = (uint32_t)hi;
    nd.left = lo == gamma ? (RT_REF_LEAF_BIT | (uint32_t)gamma) : (uint32_t)gamma;
    nd.right = hi == gamma + 1 ? (RT_REF_LEAF_BIT | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    nodes[i] = nd;
    parent[lo == gamma ? (nn - 1) + gamma : gamma] = (uint32_t)i;
    parent[hi == gamma + 1 ? (nn - 1) + gamma + 1 : gamma + 1] = (uint32_t)i;
    if (i == 0) parent[0] = RT_LBVH_NONE;      // (node 0 is the root; written by nobody else)
}

// Boxes of the inner nodes, bottom-up: the second thread to arrive at a node owns it (its children are complete).
// box: 8 floats per inner node (lo.xyz, -, hi.xyz, -); height: levels of RECORDS in the subtree (a node whose range fits a
// leaf emits no record: 0).
__global__ void lbvh_boxes_kernel(const LbvhNode* nodes, const uint32_t* parent, uint32_t n, const float* ent_box, float* box, uint32_t* height,
                                  uint32_t* arrived) {
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    uint32_t i = parent[(n - 1) + leaf];
    while (i != RT_LBVH_NONE) {
        if (atomicAdd(arrived + i, 1u) == 0u) return;          // first to arrive: the sibling subtree is not done yet
        __threadfence();
        const LbvhNode nd = nodes[i];
        float lo[3], hi[3]; uint32_t h = 0;
        for (int c = 0; c < 2; c++) {
            const uint32_t ref = c ? nd.right : nd.left;
            const volatile float* b = (ref & RT_REF_LEAF_BIT) ? ent_box + (size_t)(ref & ~RT_REF_LEAF_BIT) * 8 : box + (size_t)ref * 8;
            for (int k = 0; k < 3; k++) {
                // [min, max] of each child folded in order (bvh.rs:109-116); fminf / fmaxf are exact
                const float l = fminf(b[k], b[4 + k]), u = fmaxf(b[k], b[4 + k]);
                lo[k] = c ? fminf(lo[k], l) : l; hi[k] = c ? fmaxf(hi[k], u) : u;
            }
            if (!(ref & RT_REF_LEAF_BIT)) { const uint32_t hc = ((volatile uint32_t*)height)[ref]; h = hc > h ? hc : h; }
        }
        float* o = box + (size_t)i * 8;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = 0.0f; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = 0.0f;
        height[i] = (nd.last - nd.first + 1u > RT_LBVH_LEAF) ? h + 1u : 0u;
        __threadfence();
        i = parent[i];
    }
}

// Record numbers.  Default: inner node i -> append_base + i (set by lbvh_number_init_kernel).  The first `hot` records in
// breadth-first order take the TLAS slots [0, hot) of the LDS node cache instead.  One thread: hot is a few hundred.
__global__ void lbvh_number_init_kernel(uint32_t* number, uint32_t n_inner, uint32_t append_base) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_inner) number[i] = append_base + i;
}
__global__ void lbvh_number_kernel(const LbvhNode* nodes, uint32_t n, uint32_t hot, uint32_t* number, uint32_t* queue /* hot entries */, uint32_t hot_base = 0u) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || n <= RT_LBVH_LEAF || hot == 0) return;
    uint32_t head = 0, tail = 0;
    queue[tail++] = 0u;                                        // the root (its range is the whole array: > RT_LBVH_LEAF entities here)
    while (head < tail) {
        const uint32_t i = queue[head];
        number[i] = hot_base + head++;
        const LbvhNode nd = nodes[i];
        for (int c = 0; c < 2; c++) {
            const uint32_t ref = c ? nd.right : nd.left;
            if (ref & RT_REF_LEAF_BIT) continue;
            const LbvhNode ch = nodes[ref];
            if (ch.last - ch.first + 1u > RT_LBVH_LEAF && tail < hot) queue[tail++] = ref;
        }
    }
}

// the two-box records of device_scene.h
__global__ void lbvh_emit_kernel(const LbvhNode* nodes, uint32_t n, const uint32_t* number, const float* box, const float* ent_box, DevNode* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    const LbvhNode nd = nodes[i];
    if (nd.last - nd.first + 1u <= RT_LBVH_LEAF) return;          // folded into a leaf ref of its parent
    float lo[2][3], hi[2][3]; uint32_t refs[2];
    for (int c = 0; c < 2; c++) {
        const uint32_t ref = c ? nd.right : nd.left;
        const float* b;
        if (ref & RT_REF_LEAF_BIT) { const uint32_t pos = ref & ~RT_REF_LEAF_BIT; b = ent_box + (size_t)pos * 8; refs[c] = RT_REF_LEAF_BIT | pos; }
        else {
            const LbvhNode ch = nodes[ref];
            b = box + (size_t)ref * 8;
            const uint32_t count = ch.last - ch.first + 1u;
            refs[c] = count <= RT_LBVH_LEAF ? (RT_REF_LEAF_BIT | ((count - 1u) << RT_REF_COUNT_SHIFT) | ch.first) : number[ref];
        }
        for (int k = 0; k < 3; k++) { lo[c][k] = fminf(b[k], b[4 + k]); hi[c][k] = fmaxf(b[k], b[4 + k]); }
    }
    float* w = out[number[i]].w[0].f;      // layout: device_scene.h (left / right interleaved)
    w[0] = lo[0][0]; w[1] = lo[1][0]; w[2] = lo[0][1]; w[3] = lo[1][1];
    w[4] = lo[0][2]; w[5] = lo[1][2]; w[6] = hi[0][0]; w[7] = hi[1][0];
    w[8] = hi[0][1]; w[9] = hi[1][1]; w[10] = hi[0][2]; w[11] = hi[1][2];
    uint32_t* r = out[number[i]].w[3].u;
    r[0] = refs[0]; r[1] = refs[1]; r[2] = 0u; r[3] = 0u;
}

// ---- the same build for ONE mesh's triangles (rt_scene_rebuild_blas) ------------------------------------------------------------------
// The reference builds a mesh's BVH on the CPU whenever a MeshObject is made (MeshBvhRoot::new, mesh.rs:84-90, over bvh.rs:129-188); a mesh
// whose vertices move every frame would pay that every frame.  Same stages as above over the triangles' boxes (MeshTri::calculate_bounds,
// mesh.rs:127-133: AABB::from_vertices over a, b, c); leaves are ranges of <= RT_LBVH_LEAF triangles in key order.

// box of every triangle (8 floats: lo.xyz, -, hi.xyz, -), in the order of the caller's array
__global__ void blas_tri_boxes_kernel(const RtMeshTri* tris, uint32_t n, float* box) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RtMeshTri t = tris[i];
    float* o = box + (size_t)i * 8;
    o[0] = fminf(fminf(t.a.x, t.b.x), t.c.x); o[1] = fminf(fminf(t.a.y, t.b.y), t.c.y); o[2] = fminf(fminf(t.a.z, t.b.z), t.c.z); o[3] = 0.0f;
    o[4] = fmaxf(fmaxf(t.a.x, t.b.x), t.c.x); o[5] = fmaxf(fmaxf(t.a.y, t.b.y), t.c.y); o[6] = fmaxf(fmaxf(t.a.z, t.b.z), t.c.z); o[7] = 0.0f;
}

// position (in the mesh's slice of the triangle arrays) of every ORIGINAL triangle index: the visit ranks that decide exact ties in t
// (mesh.rs:96-98) stay those of the tree rt_scene_create built, whatever order the triangles are stored in
__global__ void blas_where_kernel(const DevTriCold* cold, uint32_t n, uint32_t* where) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t orig = (uint32_t)ldw(cold[p].w + 1).i[3];
    if (orig < n) where[orig] = p;
}

// triangle records in key order: hot = a, b, c + cross(b - a, c - a) with the host build's operations (mesh.rs:22-24, types.rs:64-68),
// cold = uvs, tex_key, original index; rank = the stored visit rank of that original triangle; box permuted alongside
__global__ void blas_permute_kernel(const unsigned long long* keys, uint32_t n, const RtMeshTri* tris, const float* box_in, const uint32_t* where,
                                    const uint32_t* rank_old, DevTriHot* hot, DevTriCold* cold, uint32_t* rank, float* box_out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t src = (uint32_t)keys[j];
    const RtMeshTri t = tris[src];
    const float abx = t.b.x - t.a.x, aby = t.b.y - t.a.y, abz = t.b.z - t.a.z;
    const float acx = t.c.x - t.a.x, acy = t.c.y - t.a.y, acz = t.c.z - t.a.z;
    const float nx = aby * acz - abz * acy, ny = -(abx * acz - abz * acx), nz = abx * acy - aby * acx;
    float* h = hot[j].w[0].f;
    h[0] = t.a.x; h[1] = t.a.y; h[2] = t.a.z; h[3] = nx;
    h[4] = t.b.x; h[5] = t.b.y; h[6] = t.b.z; h[7] = ny;
    h[8] = t.c.x; h[9] = t.c.y; h[10] = t.c.z; h[11] = nz;
    Word4* c = cold[j].w;
    c[0].f[0] = t.a_uv.x; c[0].f[1] = t.a_uv.y; c[0].f[2] = t.b_uv.x; c[0].f[3] = t.b_uv.y;
    c[1].f[0] = t.c_uv.x; c[1].f[1] = t.c_uv.y; c[1].i[2] = t.tex_key; c[1].i[3] = (int32_t)src;
    rank[j] = rank_old[where[src]];
    for (int k = 0; k < 2; k++) *reinterpret_cast<float4*>(box_out + (size_t)j * 8 + 4 * k) = *reinterpret_cast<const float4*>(box_in + (size_t)src * 8 + 4 * k);
}

// the mesh's root box (DevMesh words 1, 2): the union of all triangle boxes = the two boxes of inner node 0, or - a mesh of <= RT_LBVH_LEAF
// triangles - of its triangles
__global__ void blas_root_kernel(const float* inner_box, const float* tri_box, uint32_t n, float* root /* lo.xyz hi.xyz */) {
    float lo[3], hi[3];
    if (n > RT_LBVH_LEAF) { for (int k = 0; k < 3; k++) { lo[k] = inner_box[k]; hi[k] = inner_box[4 + k]; } }
    else {
        for (int k = 0; k < 3; k++) { lo[k] = tri_box[k]; hi[k] = tri_box[4 + k]; }
        for (uint32_t i = 1; i < n; i++) for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], tri_box[i * 8 + k]); hi[k] = fmaxf(hi[k], tri_box[i * 8 + 4 + k]); }
    }
    for (int k = 0; k < 3; k++) { root[k] = lo[k]; root[3 + k] = hi[k]; }
}

#endif
