// scene_build.cpp - rt_scene_create's host half.
//
// Replaces Scene::build_bvh (raytracer-impl/src/implementation.rs:377-383,
// 398-440) and MeshBvhRoot::new (shapes/mesh.rs:84-90, 127-133).  The tree
// TOPOLOGY and the object order inside it are the reference's, because they
// decide which candidate wins an exact tie in t:
//   - bounds + centroids per object       bvh.rs:72-83, implementation.rs:258-283,433-440
//   - longest centroid axis, mean split   bvh.rs:118-127, 155-163
//   - two-pointer unstable partition      util.rs:6-28 (incl. its untested last element)
//   - stop at <= 2 objects / empty side   bvh.rs:149, 169
// The LAYOUT is new: an index permutation is partitioned in place over SoA
// centroid arrays, and the result is emitted as 64-byte two-box nodes
// (device_scene.h) with objects re-ordered into leaf order.
#include "scene_build.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstring>
#include <memory>

namespace {

struct P3 { float x, y, z; };
struct Box { P3 lo, hi; };

inline P3 p3(const RtV3& v) { return P3{v.x, v.y, v.z}; }
inline float comp(const P3& p, int axis) { return axis == 0 ? p.x : (axis == 1 ? p.y : p.z); }

inline void grow(Box& b, const P3& v) {
    b.lo.x = fminf(b.lo.x, v.x); b.lo.y = fminf(b.lo.y, v.y); b.lo.z = fminf(b.lo.z, v.z);
    b.hi.x = fmaxf(b.hi.x, v.x); b.hi.y = fmaxf(b.hi.y, v.y); b.hi.z = fmaxf(b.hi.z, v.z);
}
inline bool box_is_infinite(const Box& b) {
    const float inf = INFINITY;
    return b.lo.x == -inf && b.lo.y == -inf && b.lo.z == -inf && b.hi.x == inf && b.hi.y == inf && b.hi.z == inf;
}

// Rodrigues rotation with the same expression shape as types.rs:200
inline P3 rotate(const P3& p, const P3& k, float theta) {
    const float c = cosf(theta), s = sinf(theta);
    const P3 kxp = {k.y * p.z - k.z * p.y, -(k.x * p.z - k.z * p.x), k.x * p.y - k.y * p.x};
    const float kd = k.x * p.x + k.y * p.y + k.z * p.z;
    const float omc = 1.0f - c;
    return P3{(p.x * c + kxp.x * s) + (k.x * kd) * omc,
              (p.y * c + kxp.y * s) + (k.y * kd) * omc,
              (p.z * c + kxp.z * s) + (k.z * kd) * omc};
}

// ---------------------------------------------------------------------------
// Binary tree in the reference's shape
// ---------------------------------------------------------------------------
struct TreeNode {
    Box box;
    uint32_t first, count;   // range of positions (always valid: the objects below this node)
    int32_t left, right;     // -1 for a leaf
};

struct Tree {
    std::vector<TreeNode> nodes;     // node 0 = root
    std::vector<uint32_t> perm;      // position -> object index (the reference's object_bounds order)
};

Box range_box(const std::vector<Box>& boxes, const std::vector<uint32_t>& perm, uint32_t first, uint32_t count) {
    Box b{boxes[perm[first]].lo, boxes[perm[first]].lo};   // bvh.rs:109-116 folds [min,max] of each object in order
    for (uint32_t i = 0; i < count; i++) { grow(b, boxes[perm[first + i]].lo); grow(b, boxes[perm[first + i]].hi); }
    return b;
}

// Large inputs (point clouds, big meshes) are built by several host threads: the top of the tree
// serially, every subtree below kParallelGrain objects as a task of its own on a disjoint slice of
// `perm`.  Subtrees are spliced in task order, so the result does not depend on the thread count.
constexpr uint32_t kParallelMin = 200000, kParallelGrain = 32768;

template <class Task, class Fn>
void run_tasks(std::vector<Task>& tasks, Fn fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_threads = std::min<unsigned>({hw ? hw : 1u, 16u, (unsigned)tasks.size()});
    std::atomic<size_t> next{0};
    auto worker = [&]() { for (size_t i; (i = next.fetch_add(1)) < tasks.size();) fn(tasks[i]); };
    std::vector<std::thread> pool;
    for (unsigned k = 1; k < n_threads; k++) pool.emplace_back(worker);
    worker();
    for (std::thread& th : pool) th.join();
}

// bvh.rs:144-188 on the subtree of nodes[root] (children appended to `nodes`); nodes of at most
// `defer_below` objects are left unsplit and reported in `deferred` instead (0: split everything).
void subdivide_reference(std::vector<TreeNode>& nodes, uint32_t root, std::vector<uint32_t>& perm, const std::vector<Box>& boxes,
                         const std::vector<P3>& centroids, uint32_t defer_below, std::vector<uint32_t>* deferred) {
    std::vector<uint32_t> todo{root};
    while (!todo.empty()) {
        const uint32_t ni = todo.back(); todo.pop_back();
        const uint32_t first = nodes[ni].first, count = nodes[ni].count;
        if (count <= 2) continue;
        if (count <= defer_below) { deferred->push_back(ni); continue; }
        uint32_t* idx = perm.data() + first;
        // extent of the centroids -> longest axis (note: the comparisons use the signed
        // extent, only the remembered length is abs'ed - bvh.rs:118-127)
        Box cb{centroids[idx[0]], centroids[idx[0]]};
        for (uint32_t i = 1; i < count; i++) grow(cb, centroids[idx[i]]);
        const P3 ext{cb.hi.x - cb.lo.x, cb.hi.y - cb.lo.y, cb.hi.z - cb.lo.z};
        int axis = 0; float len = fabsf(ext.x);
        if (ext.y > len) { axis = 1; len = fabsf(ext.y); }
        if (ext.z > len) { axis = 2; }
        // mean of the centroid coordinate, summed in current object order
        float sum = 0.0f;
        for (uint32_t i = 0; i < count; i++) sum += comp(centroids[idx[i]], axis);
        const float split = sum / (float)count;
        // util.rs:13-27
        uint32_t i = 0, j = count - 1;
        while (i < j) {
            if (comp(centroids[idx[i]], axis) <= split) i++;
            else { const uint32_t tmp = idx[i]; idx[i] = idx[j]; idx[j] = tmp; j--; }
        }
        const uint32_t left_len = i + 1, right_len = count - left_len;
        if (left_len == 0 || right_len == 0) continue;
        const int32_t l = (int32_t)nodes.size();
        nodes.push_back(TreeNode{range_box(boxes, perm, first, left_len), first, left_len, -1, -1});
        nodes.push_back(TreeNode{range_box(boxes, perm, first + left_len, right_len), first + left_len, right_len, -1, -1});
        nodes[ni].left = l; nodes[ni].right = l + 1;
        todo.push_back((uint32_t)l); todo.push_back((uint32_t)l + 1);
    }
}

// Appends a subtree built in its own vector (node 0 = the subtree's root, a copy of dst[at]) to dst.
void splice_subtree(std::vector<TreeNode>& dst, uint32_t at, const std::vector<TreeNode>& sub) {
    const int32_t shift = (int32_t)dst.size() - 1;            // sub node k (k >= 1) becomes dst node k + shift
    dst[at] = sub[0];
    if (sub[0].left >= 0) { dst[at].left += shift; dst[at].right += shift; }
    for (size_t k = 1; k < sub.size(); k++) {
        TreeNode n = sub[k];
        if (n.left >= 0) { n.left += shift; n.right += shift; }
        dst.push_back(n);
    }
}

Tree build_tree(const std::vector<Box>& boxes, const std::vector<P3>& centroids) {
    Tree t;
    const uint32_t n = (uint32_t)boxes.size();
    t.perm.resize(n);
    for (uint32_t i = 0; i < n; i++) t.perm[i] = i;
    t.nodes.reserve((size_t)n * 2);
    t.nodes.push_back(TreeNode{range_box(boxes, t.perm, 0, n), 0, n, -1, -1});
    if (n < kParallelMin) { subdivide_reference(t.nodes, 0, t.perm, boxes, centroids, 0, nullptr); return t; }
    struct Task { uint32_t node; std::vector<TreeNode> sub; };
    std::vector<uint32_t> deferred;
    subdivide_reference(t.nodes, 0, t.perm, boxes, centroids, kParallelGrain, &deferred);
    std::vector<Task> tasks(deferred.size());
    for (size_t k = 0; k < deferred.size(); k++) tasks[k].node = deferred[k];
    run_tasks(tasks, [&](Task& task) {
        task.sub.push_back(t.nodes[task.node]);
        subdivide_reference(task.sub, 0, t.perm, boxes, centroids, 0, nullptr);      // (its slice of perm is this task's alone)
    });
    for (const Task& task : tasks) splice_subtree(t.nodes, task.node, task.sub);
    return t;
}

// Emission order of bvh.rs:210-236 when every box passes: right child first,
// objects of a leaf in ascending offset.  Gives positions, in visit order.
void emission_positions(const Tree& t, std::vector<uint32_t>& out) {
    std::vector<int32_t> st{0};
    out.clear(); out.reserve(t.perm.size());
    while (!st.empty()) {
        const TreeNode& n = t.nodes[st.back()]; st.pop_back();
        if (n.left >= 0) { st.push_back(n.left); st.push_back(n.right); }
        else for (uint32_t k = 0; k < n.count; k++) out.push_back(n.first + k);
    }
}

// ---------------------------------------------------------------------------
// Emit 64-byte two-box nodes
// ---------------------------------------------------------------------------
constexpr uint32_t RT_NONE_U32 = 0xFFFFFFFFu;

struct Emitter {
    const Tree& t;
    uint32_t depth = 0;      // max inner nodes on a root-to-leaf path

    // the tree as it will be emitted: one record per inner node, children = leaf refs or records
    // l_src / r_src: where a child's box comes from when the tree is refitted (rt_scene_update_transforms): the union of the
    // object boxes of a range of positions (count > 0: a reference leaf - or, below a leaf of more than RT_LEAF_MAX objects,
    // the whole of that leaf), or the union of the two boxes of the child's own record (count == 0).
    struct ENode { Box lb, rb; uint32_t l, r; bool l_inner, r_inner; uint32_t level; uint32_t l_src_first, l_src_count, r_src_first, r_src_count; };
    std::vector<ENode> en;
    std::vector<uint32_t> number;      // record index of every ENode in the scene's node array (after write())
    uint32_t root = 0; bool root_inner = false;

    explicit Emitter(const Tree& tree) : t(tree) { root = build(0, 0, root_inner); }

    static uint32_t leaf_ref(uint32_t first, uint32_t count) {
        return RT_REF_LEAF_BIT | ((count - 1) << RT_REF_COUNT_SHIFT) | (first & RT_REF_FIRST_MASK);
    }
    // one 16-byte word per axis pair - see device_scene.h
    static void put_boxes(DevNode& d, const Box& l, const Box& r) {
        d.w[0].f[0] = l.lo.x; d.w[0].f[1] = r.lo.x; d.w[0].f[2] = l.lo.y; d.w[0].f[3] = r.lo.y;
        d.w[1].f[0] = l.lo.z; d.w[1].f[1] = r.lo.z; d.w[1].f[2] = l.hi.x; d.w[1].f[3] = r.hi.x;
        d.w[2].f[0] = l.hi.y; d.w[2].f[1] = r.hi.y; d.w[2].f[2] = l.hi.z; d.w[2].f[3] = r.hi.z;
    }
    // A leaf of more than RT_LEAF_MAX objects (only when a split left one side
    // empty, bvh.rs:169) becomes a small subtree whose boxes all equal the leaf's
    // box: every object is still tested exactly when that box is hit.
    uint32_t build_leaf(uint32_t first, uint32_t count, const Box& box, uint32_t level, bool& inner, uint32_t src_first, uint32_t src_count) {
        if (count <= RT_LEAF_MAX) { if (level > depth) depth = level; inner = false; return leaf_ref(first, count); }
        const uint32_t me = (uint32_t)en.size(); en.emplace_back();
        const uint32_t half = count / 2;
        bool li, ri;
        const uint32_t l = build_leaf(first, half, box, level + 1, li, src_first, src_count);
        const uint32_t r = build_leaf(first + half, count - half, box, level + 1, ri, src_first, src_count);
        en[me] = ENode{box, box, l, r, li, ri, level, src_first, src_count, src_first, src_count};      // both boxes = the whole leaf's
        inner = true;
        return me;
    }
    uint32_t build(int32_t ni, uint32_t level, bool& inner) {
        const TreeNode n = t.nodes[ni];
        if (n.left < 0) return build_leaf(n.first, n.count, n.box, level, inner, n.first, n.count);
        const uint32_t me = (uint32_t)en.size(); en.emplace_back();
        bool li, ri;
        const uint32_t l = build(n.left, level + 1, li);
        const uint32_t r = build(n.right, level + 1, ri);
        const TreeNode& cl = t.nodes[n.left]; const TreeNode& cr = t.nodes[n.right];
        en[me] = ENode{cl.box, cr.box, l, r, li, ri, level, cl.first, cl.left < 0 ? cl.count : 0u, cr.first, cr.left < 0 ? cr.count : 0u};
        inner = true;
        return me;
    }
    // Writes the records into `nodes` (the scene's one node array; child refs are indices in it):
    // the first `n_hot` records in breadth-first order - the top levels, which every ray visits and the
    // render kernel keeps in LDS - from nodes[hot_base] on, the others depth first (a child next to its
    // parent) from nodes[cold_base] on.  Returns the ref of the root.
    uint32_t write(std::vector<DevNode>& nodes, uint32_t hot_base, uint32_t n_hot, uint32_t cold_base) {
        if (!root_inner) return root;
        number.assign(en.size(), RT_NONE_U32);
        std::vector<uint32_t> order{root};
        for (size_t i = 0; i < order.size() && i < n_hot; i++) {
            const ENode& e = en[order[i]];
            number[order[i]] = hot_base + (uint32_t)i;
            if (e.l_inner) order.push_back(e.l);
            if (e.r_inner) order.push_back(e.r);
        }
        uint32_t next_cold = cold_base;
        std::vector<uint32_t> st{root};
        while (!st.empty()) {
            const uint32_t k = st.back(); st.pop_back();
            if (number[k] == RT_NONE_U32) number[k] = next_cold++;
            if (en[k].r_inner) st.push_back(en[k].r);
            if (en[k].l_inner) st.push_back(en[k].l);
        }
        for (size_t k = 0; k < en.size(); k++) {
            const ENode& e = en[k];
            DevNode& d = nodes[number[k]];
            put_boxes(d, e.lb, e.rb);
            d.w[3].u[0] = e.l_inner ? number[e.l] : e.l;
            d.w[3].u[1] = e.r_inner ? number[e.r] : e.r;
            d.w[3].u[2] = d.w[3].u[3] = 0;
        }
        return number[root];
    }
};

// ---------------------------------------------------------------------------
// The tree the GPU walks: binned SAH over the REFERENCE's leaves
// ---------------------------------------------------------------------------
// Which objects the reference tests for a ray is decided by the leaf boxes alone: a leaf's
// objects are candidates iff the leaf's box and all its ancestors' boxes pass hit_aabb
// (bvh.rs:210-236), every ancestor box is the exact union (component-wise min / max) of the boxes
// below it, and hit_aabb is monotone under box containment (f32 subtraction, multiplication by
// a fixed sign and max / min are monotone; a NaN product - plane through the origin of an
// axis-parallel ray - is ignored, never restrictive), so a leaf box that passes implies that all
// its ancestors pass.  Any tree over the same leaves whose inner boxes are exact unions therefore
// yields the same candidates, and the hit is chosen by (t, reference visit rank), not by visiting
// order.  So the device tree is built for walking cost, not copied from the reference's
// mean-split topology; the reference tree still provides the leaves, their boxes and the ranks.
#ifndef RT_SAH_BINS
#define RT_SAH_BINS 64
#endif
struct LeafRec { Box box; uint32_t first, count; };

double clamp30(float v) { return v != v ? 0.0 : (v > 1.0e30f ? 1.0e30 : (v < -1.0e30f ? -1.0e30 : (double)v)); }
double half_area(const double lo[3], const double hi[3]) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

struct WalkJob { uint32_t node, lo, hi; };

// Splits the leaf range of every job reachable from `root_job` (children appended to `nodes`); ranges of at
// most `defer_below` leaves are reported in `deferred` instead (0: split everything).
void subdivide_walk(std::vector<TreeNode>& nodes, WalkJob root_job, const std::vector<LeafRec>& leaves, std::vector<uint32_t>& ids,
                    const std::vector<double>& cen, uint32_t defer_below, std::vector<WalkJob>* deferred) {
    constexpr int BINS = RT_SAH_BINS;
    std::vector<WalkJob> todo{root_job};
    while (!todo.empty()) {
        const WalkJob j = todo.back(); todo.pop_back();
        const uint32_t n = j.hi - j.lo;
        if (n > 1 && n <= defer_below) { deferred->push_back(j); continue; }
        // exact union of the leaf boxes of the range
        Box bx = leaves[ids[j.lo]].box;
        for (uint32_t k = j.lo + 1; k < j.hi; k++) { grow(bx, leaves[ids[k]].box.lo); grow(bx, leaves[ids[k]].box.hi); }
        nodes[j.node].box = bx;
        if (n == 1) { const LeafRec& l = leaves[ids[j.lo]]; nodes[j.node].first = l.first; nodes[j.node].count = l.count; continue; }
        // binned SAH on the centroids
        double cb_lo[3] = {1e300, 1e300, 1e300}, cb_hi[3] = {-1e300, -1e300, -1e300};
        for (uint32_t k = j.lo; k < j.hi; k++) for (int a = 0; a < 3; a++) {
            const double c = cen[3 * (size_t)ids[k] + a];
            if (c < cb_lo[a]) cb_lo[a] = c;
            if (c > cb_hi[a]) cb_hi[a] = c;
        }
        double best_cost = 1e300; int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; a++) {
            const double ext = cb_hi[a] - cb_lo[a];
            if (!(ext > 0.0)) continue;
            double blo[BINS][3], bhi[BINS][3]; uint64_t bcnt[BINS];
            for (int b = 0; b < BINS; b++) { bcnt[b] = 0; for (int c = 0; c < 3; c++) { blo[b][c] = 1e300; bhi[b][c] = -1e300; } }
            for (uint32_t k = j.lo; k < j.hi; k++) {
                const LeafRec& l = leaves[ids[k]];
                int b = (int)((cen[3 * (size_t)ids[k] + a] - cb_lo[a]) / ext * BINS); if (b >= BINS) b = BINS - 1; if (b < 0) b = 0;
                bcnt[b] += l.count;
                const double lo3[3] = {clamp30(l.box.lo.x), clamp30(l.box.lo.y), clamp30(l.box.lo.z)}, hi3[3] = {clamp30(l.box.hi.x), clamp30(l.box.hi.y), clamp30(l.box.hi.z)};
                for (int c = 0; c < 3; c++) { if (lo3[c] < blo[b][c]) blo[b][c] = lo3[c]; if (hi3[c] > bhi[b][c]) bhi[b][c] = hi3[c]; }
            }
            double right_area[BINS]; uint64_t right_cnt[BINS];
            double rlo[3] = {1e300, 1e300, 1e300}, rhi[3] = {-1e300, -1e300, -1e300}; uint64_t rc = 0;
            for (int b = BINS - 1; b > 0; b--) {
                if (bcnt[b]) for (int c = 0; c < 3; c++) { if (blo[b][c] < rlo[c]) rlo[c] = blo[b][c]; if (bhi[b][c] > rhi[c]) rhi[c] = bhi[b][c]; }
                rc += bcnt[b]; right_cnt[b] = rc; right_area[b] = rc ? half_area(rlo, rhi) : 0.0;
            }
            double llo[3] = {1e300, 1e300, 1e300}, lhi[3] = {-1e300, -1e300, -1e300}; uint64_t lc = 0;
            for (int b = 0; b + 1 < BINS; b++) {
                if (bcnt[b]) for (int c = 0; c < 3; c++) { if (blo[b][c] < llo[c]) llo[c] = blo[b][c]; if (bhi[b][c] > lhi[c]) lhi[c] = bhi[b][c]; }
                lc += bcnt[b];
                if (lc == 0 || right_cnt[b + 1] == 0) continue;
                const double cost = half_area(llo, lhi) * (double)lc + right_area[b + 1] * (double)right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b + 1; }
            }
        }
        uint32_t mid = j.lo + n / 2;                            // no usable split (coincident centroids): halve the range
        if (best_axis >= 0) {
            const double ext = cb_hi[best_axis] - cb_lo[best_axis];
            uint32_t* first = ids.data() + j.lo; uint32_t* last = ids.data() + j.hi;
            uint32_t* m = std::partition(first, last, [&](uint32_t id) {
                int b = (int)((cen[3 * (size_t)id + best_axis] - cb_lo[best_axis]) / ext * BINS); if (b >= BINS) b = BINS - 1; if (b < 0) b = 0;
                return b < best_split;
            });
            const uint32_t pm = (uint32_t)(m - ids.data());
            if (pm > j.lo && pm < j.hi) mid = pm;
        }
        const int32_t l = (int32_t)nodes.size();
        nodes.push_back(TreeNode{Box{}, 0, 0, -1, -1});
        nodes.push_back(TreeNode{Box{}, 0, 0, -1, -1});
        nodes[j.node].left = l; nodes[j.node].right = l + 1;
        todo.push_back(WalkJob{(uint32_t)l, j.lo, mid});
        todo.push_back(WalkJob{(uint32_t)l + 1, mid, j.hi});
    }
}

Tree build_walk_tree(const Tree& ref) {
    std::vector<LeafRec> leaves;
    for (const TreeNode& n : ref.nodes) if (n.left < 0) leaves.push_back(LeafRec{n.box, n.first, n.count});
    Tree t;
    const uint32_t n_leaves = (uint32_t)leaves.size();
    std::vector<uint32_t> ids(n_leaves);
    for (uint32_t i = 0; i < n_leaves; i++) ids[i] = i;
    std::vector<double> cen((size_t)n_leaves * 3);
    for (uint32_t i = 0; i < n_leaves; i++) {
        const Box& b = leaves[i].box;
        cen[3 * i] = 0.5 * (clamp30(b.lo.x) + clamp30(b.hi.x)); cen[3 * i + 1] = 0.5 * (clamp30(b.lo.y) + clamp30(b.hi.y));
        cen[3 * i + 2] = 0.5 * (clamp30(b.lo.z) + clamp30(b.hi.z));
    }
    t.nodes.reserve((size_t)n_leaves * 2);
    t.nodes.push_back(TreeNode{Box{}, 0, 0, -1, -1});
    if (n_leaves < kParallelMin) { subdivide_walk(t.nodes, WalkJob{0, 0, n_leaves}, leaves, ids, cen, 0, nullptr); return t; }
    struct Task { WalkJob job; std::vector<TreeNode> sub; };
    std::vector<WalkJob> deferred;
    subdivide_walk(t.nodes, WalkJob{0, 0, n_leaves}, leaves, ids, cen, kParallelGrain, &deferred);
    std::vector<Task> tasks(deferred.size());
    for (size_t k = 0; k < deferred.size(); k++) tasks[k].job = deferred[k];
    run_tasks(tasks, [&](Task& task) {
        task.sub.push_back(TreeNode{Box{}, 0, 0, -1, -1});
        subdivide_walk(task.sub, WalkJob{0, task.job.lo, task.job.hi}, leaves, ids, cen, 0, nullptr);      // (its slice of ids is this task's alone)
    });
    for (const Task& task : tasks) splice_subtree(t.nodes, task.job.node, task.sub);
    return t;
}

// Depth of the node array the Emitter would produce for `t` (inner levels on the longest path)
uint32_t emitted_depth(const Tree& t) {
    uint32_t depth = 0;
    std::vector<std::pair<int32_t, uint32_t>> st{{0, 0}};
    while (!st.empty()) {
        const auto [ni, level] = st.back(); st.pop_back();
        const TreeNode& n = t.nodes[ni];
        if (n.left >= 0) { st.push_back({n.left, level + 1}); st.push_back({n.right, level + 1}); continue; }
        uint32_t extra = 0;
        for (uint32_t c = n.count; c > RT_LEAF_MAX; c = c - c / 2) extra++;      // emit_leaf halves: the larger half decides
        if (level + extra > depth) depth = level + extra;
    }
    return depth;
}

int bad(std::string& err, const std::string& msg) { err = msg; return RT_E_INVALID; }

}  // namespace

int build_flat_scene(const RtSceneDesc* d, FlatScene& out, std::string& err) {
    if (!d) return bad(err, "rt_scene_create: null description");
    if (d->abi_version != RT_ABI_VERSION) return bad(err, "rt_scene_create: abi_version mismatch");
    if (d->entity_count == 0) return bad(err, "rt_scene_create: scene has no entities");
    if (d->entity_count > RT_REF_FIRST_MASK) { err = "too many entities"; return RT_E_LIMIT; }
    const uint64_t n_mat = d->material_count, n_tex = d->texture_count;

    // ---- validation ---------------------------------------------------------
    for (uint64_t i = 0; i < d->texture_count; i++) {
        const RtTexture& t = d->textures[i];
        auto child_ok = [&](int32_t c) { return c >= 0 && (uint64_t)c < n_tex; };
        switch (t.kind) {
            case RT_TEX_COLOR: case RT_TEX_UV_TEST: case RT_TEX_XYZ_TEST: break;
            case RT_TEX_CHECKER:
                if (!child_ok(t.child0) || !child_ok(t.child1)) return bad(err, "checker texture: bad child");
                for (int32_t c : {t.child0, t.child1}) {
                    const uint32_t k = d->textures[c].kind;
                    if (k == RT_TEX_CHECKER || k == RT_TEX_MESH || k == RT_TEX_MESH_SET) return bad(err, "checker texture: children must be leaf textures");
                }
                break;
            case RT_TEX_COLOR_MAP:
                if (t.width == 0 || t.height == 0 || t.texel_first + (uint64_t)t.width * t.height > d->texel_count) return bad(err, "color map: texels out of range");
                break;
            case RT_TEX_MESH:
                if (t.child0 != -1 && (!child_ok(t.child0) || d->textures[t.child0].kind != RT_TEX_COLOR_MAP)) return bad(err, "mesh texture: map must be a color map");
                break;
            case RT_TEX_MESH_SET:
                if (t.child0 < 0 || t.child1 < 0 || (uint64_t)t.child0 + (uint64_t)t.child1 > d->tex_set_entry_count) return bad(err, "mesh texture set: entries out of range");
                for (int32_t k = 0; k < t.child1; k++) {
                    const int32_t e = d->tex_set_entries[t.child0 + k];
                    if (!child_ok(e) || d->textures[e].kind != RT_TEX_MESH) return bad(err, "mesh texture set: entries must be mesh textures");
                }
                break;
            default: return bad(err, "unknown texture kind");
        }
    }
    for (uint64_t i = 0; i < n_mat; i++) if (d->materials[i].kind > RT_MAT_DIELECTRIC) return bad(err, "unknown material kind");
    for (uint64_t i = 0; i < d->light_count; i++) if (d->lights[i].kind > RT_LIGHT_DIRECTIONAL) return bad(err, "unknown light kind");
    if (d->sky > RT_SKY_BLACK) return bad(err, "unknown sky");
    for (uint64_t i = 0; i < d->sphere_count; i++) if (d->spheres[i].mat_id >= n_mat || d->spheres[i].tex_id >= n_tex) return bad(err, "sphere: material/texture id out of range");
    for (uint64_t i = 0; i < d->plane_count; i++) if (d->planes[i].mat_id >= n_mat || d->planes[i].tex_id >= n_tex) return bad(err, "plane: material/texture id out of range");
    for (uint64_t i = 0; i < d->mesh_object_count; i++) {
        const RtMeshObject& m = d->mesh_objects[i];
        if (m.mesh_index >= d->mesh_count || m.mat_id >= n_mat || m.tex_id >= n_tex) return bad(err, "mesh object: index out of range");
    }
    for (uint64_t i = 0; i < d->mesh_count; i++) {
        if (d->meshes[i].tri_count == 0 || !d->meshes[i].tris) return bad(err, "mesh without triangles");
        if (d->meshes[i].tri_count > RT_REF_FIRST_MASK) { err = "mesh too large"; return RT_E_LIMIT; }
    }
    for (uint64_t i = 0; i < d->entity_count; i++) {
        const RtEntity& e = d->entities[i];
        const uint64_t lim = e.hitable_kind == RT_HITABLE_SPHERE ? d->sphere_count : e.hitable_kind == RT_HITABLE_PLANE ? d->plane_count
                           : e.hitable_kind == RT_HITABLE_MESH ? d->mesh_object_count : 0;
        if (e.hitable_index >= lim) return bad(err, "entity: hitable out of range");
        if ((uint64_t)e.translation_first + e.translation_count > d->translation_count) return bad(err, "entity: translations out of range");
        if ((uint64_t)e.rotation_first + e.rotation_count > d->rotation_count) return bad(err, "entity: rotations out of range");
        if (e.translation_count > 255 || e.rotation_count > 255) { err = "entity: more than 255 transforms"; return RT_E_LIMIT; }
    }

    const bool timing = getenv("RT_BUILD_TIMING") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "scene build: %-28s %.3f s\n", what, std::chrono::duration<double>(now - lap_t).count());
        lap_t = now;
    };
    lap("validation");
    // ---- camera / scalars -----------------------------------------------------
    DevSceneView& v = out.view;
    auto put3 = [](float* dst, const RtV3& s) { dst[0] = s.x; dst[1] = s.y; dst[2] = s.z; };
    put3(v.cam_llc, d->camera.lower_left_corner); put3(v.cam_horizontal, d->camera.horizontal); put3(v.cam_vertical, d->camera.vertical);
    put3(v.cam_origin, d->camera.origin); put3(v.cam_u, d->camera.u); put3(v.cam_v, d->camera.v);
    v.cam_lens_radius = d->camera.lens_radius;
    v.sky = d->sky; v.light_count = (uint32_t)d->light_count; v.entity_count = (uint32_t)d->entity_count;

    // ---- BLAS per mesh (MeshBvhRoot::new) ------------------------------------
    // TLAS nodes are emitted first so that node 0.. belongs to the TLAS; reserve
    // by building the BLAS trees now and emitting them after the TLAS.
    const uint64_t n_mesh = d->mesh_count;
    std::vector<Tree> blas(n_mesh);
    std::vector<Box> mesh_root_box(n_mesh);
    for (uint64_t m = 0; m < n_mesh; m++) {
        const RtMesh& mesh = d->meshes[m];
        std::vector<Box> boxes(mesh.tri_count); std::vector<P3> cents(mesh.tri_count);
        for (uint64_t i = 0; i < mesh.tri_count; i++) {
            const P3 a = p3(mesh.tris[i].a), b = p3(mesh.tris[i].b), c = p3(mesh.tris[i].c);
            Box bx{a, a}; grow(bx, b); grow(bx, c);
            boxes[i] = bx;
            // (a + b + c) * 0.33333, mesh.rs:130
            cents[i] = P3{((a.x + b.x) + c.x) * 0.33333f, ((a.y + b.y) + c.y) * 0.33333f, ((a.z + b.z) + c.z) * 0.33333f};
        }
        blas[m] = build_tree(boxes, cents);
        mesh_root_box[m] = blas[m].nodes[0].box;
    }

    // ---- TLAS over entities (EntityBvhRoot::new) ------------------------------
    const uint64_t n_ent = d->entity_count;
    std::vector<Box> ebox(n_ent); std::vector<P3> ecent(n_ent);
    for (uint64_t i = 0; i < n_ent; i++) {
        const RtEntity& e = d->entities[i];
        Box hb;
        if (e.hitable_kind == RT_HITABLE_SPHERE) { const float r = d->spheres[e.hitable_index].radius; hb = Box{P3{0.0f - r, 0.0f - r, 0.0f - r}, P3{0.0f + r, 0.0f + r, 0.0f + r}}; }
        else if (e.hitable_kind == RT_HITABLE_PLANE) hb = Box{P3{-INFINITY, -INFINITY, -INFINITY}, P3{INFINITY, INFINITY, INFINITY}};
        else hb = mesh_root_box[d->mesh_objects[e.hitable_index].mesh_index];
        // calculate_origin_aabb, implementation.rs:258-283
        P3 c[8] = {hb.lo, {hb.lo.x, hb.lo.y, hb.hi.z}, {hb.lo.x, hb.hi.y, hb.lo.z}, {hb.hi.x, hb.lo.y, hb.lo.z},
                   hb.hi, {hb.hi.x, hb.hi.y, hb.lo.z}, {hb.hi.x, hb.lo.y, hb.hi.z}, {hb.lo.x, hb.hi.y, hb.hi.z}};
        if (!box_is_infinite(hb))
            for (uint32_t k = e.rotation_count; k-- > 0;) {
                const RtRotation& r = d->rotations[e.rotation_first + k];
                for (P3& p : c) p = rotate(p, p3(r.axis), r.theta);
            }
        P3 origin{0.0f, 0.0f, 0.0f};
        for (uint32_t k = 0; k < e.translation_count; k++) {
            const P3 t = p3(d->translations[e.translation_first + k]);
            for (P3& p : c) p = P3{p.x + t.x, p.y + t.y, p.z + t.z};
            origin = P3{origin.x + t.x, origin.y + t.y, origin.z + t.z};
        }
        Box bx{c[0], c[0]};
        for (int k = 1; k < 8; k++) grow(bx, c[k]);
        ebox[i] = bx;
        // implementation.rs:437
        ecent[i] = box_is_infinite(bx) ? origin
                                       : P3{bx.lo.x * 0.5f + bx.hi.x * 0.5f, bx.lo.y * 0.5f + bx.hi.y * 0.5f, bx.lo.z * 0.5f + bx.hi.z * 0.5f};
    }
    lap("BLAS trees, entity boxes");
    Tree tlas = build_tree(ebox, ecent);
    lap("TLAS reference tree");

    // ---- emit nodes -------------------------------------------------------------
    out.nodes.clear();
    static const bool walk_trees = []() { const char* e = getenv("RT_WALK_TREE"); return !e || atoi(e) != 0; }();      // 0: walk the reference's own topology
    constexpr uint32_t kMaxWalkDepth = 26;                  // keeps the LDS stack of TLAS + BLAS within 64 KB
    Tree tlas_walk;
    bool tlas_own = false;
    if (walk_trees) { tlas_walk = build_walk_tree(tlas); tlas_own = emitted_depth(tlas_walk) <= std::max(kMaxWalkDepth, emitted_depth(tlas)); }
    lap("TLAS walk tree");
    // every tree as it will be emitted; the records are written once the LDS budget of the node cache is known
    std::vector<std::unique_ptr<Emitter>> emitters;
    emitters.emplace_back(new Emitter(tlas_own ? tlas_walk : tlas));
    put3(v.tlas_min, RtV3{tlas.nodes[0].box.lo.x, tlas.nodes[0].box.lo.y, tlas.nodes[0].box.lo.z});
    put3(v.tlas_max, RtV3{tlas.nodes[0].box.hi.x, tlas.nodes[0].box.hi.y, tlas.nodes[0].box.hi.z});
    out.info.tlas_node_count = emitters[0]->en.size();
    out.info.tlas_depth = emitted_depth(tlas);             // of the reference-shaped tree
    const uint32_t tlas_walk_depth = emitters[0]->depth;
    tlas_walk = Tree();
    lap("TLAS nodes");

    out.meshes.resize(n_mesh);
    out.blas_order.resize(n_mesh);
    uint32_t blas_depth = 0, blas_walk_depth = 0;
    uint64_t tri_total = 0;
    std::vector<uint32_t> pos_order;
    for (uint64_t m = 0; m < n_mesh; m++) {
        const RtMesh& mesh = d->meshes[m];
        const Tree& t = blas[m];
        const uint32_t tri_base = (uint32_t)out.tri_hot.size();
        {
            Tree walk;
            bool own = false;
            if (walk_trees) { walk = build_walk_tree(t); own = emitted_depth(walk) <= std::max(kMaxWalkDepth, emitted_depth(t)); }
            emitters.emplace_back(new Emitter(own ? walk : t));     // (the emitter keeps no reference to `walk` after its constructor)
        }
        const Emitter& be = *emitters.back();
        if (be.depth > blas_walk_depth) blas_walk_depth = be.depth;
        out.blas_walk_depths.push_back(be.depth);
        blas_depth = std::max(blas_depth, emitted_depth(t));
        DevMesh& dm = out.meshes[m];
        dm.w[0].u[0] = 0; dm.w[0].u[1] = tri_base; dm.w[0].u[2] = (uint32_t)mesh.tri_count; dm.w[0].u[3] = 0;      // root ref: below
        const Box& rb = t.nodes[0].box;
        dm.w[1].f[0] = rb.lo.x; dm.w[1].f[1] = rb.lo.y; dm.w[1].f[2] = rb.lo.z; dm.w[1].f[3] = rb.hi.x;
        dm.w[2].f[0] = rb.hi.y; dm.w[2].f[1] = rb.hi.z; dm.w[2].f[2] = 0.0f; dm.w[2].f[3] = 0.0f;
        // triangles in leaf order
        emission_positions(t, pos_order);
        std::vector<uint32_t> rank(mesh.tri_count);
        out.blas_order[m].resize(mesh.tri_count);
        for (uint32_t k = 0; k < pos_order.size(); k++) { rank[pos_order[k]] = k; out.blas_order[m][k] = t.perm[pos_order[k]]; }
        for (uint64_t p = 0; p < mesh.tri_count; p++) {
            const RtMeshTri& tr = mesh.tris[t.perm[p]];
            // normal = cross(b - a, c - a), mesh.rs:22-24 / types.rs:64-68
            const P3 ab{tr.b.x - tr.a.x, tr.b.y - tr.a.y, tr.b.z - tr.a.z};
            const P3 ac{tr.c.x - tr.a.x, tr.c.y - tr.a.y, tr.c.z - tr.a.z};
            const P3 n{ab.y * ac.z - ab.z * ac.y, -(ab.x * ac.z - ab.z * ac.x), ab.x * ac.y - ab.y * ac.x};
            DevTriHot h;
            h.w[0].f[0] = tr.a.x; h.w[0].f[1] = tr.a.y; h.w[0].f[2] = tr.a.z; h.w[0].f[3] = n.x;
            h.w[1].f[0] = tr.b.x; h.w[1].f[1] = tr.b.y; h.w[1].f[2] = tr.b.z; h.w[1].f[3] = n.y;
            h.w[2].f[0] = tr.c.x; h.w[2].f[1] = tr.c.y; h.w[2].f[2] = tr.c.z; h.w[2].f[3] = n.z;
            out.tri_hot.push_back(h);
            DevTriCold c;
            c.w[0].f[0] = tr.a_uv.x; c.w[0].f[1] = tr.a_uv.y; c.w[0].f[2] = tr.b_uv.x; c.w[0].f[3] = tr.b_uv.y;
            c.w[1].f[0] = tr.c_uv.x; c.w[1].f[1] = tr.c_uv.y; c.w[1].i[2] = tr.tex_key; c.w[1].i[3] = (int32_t)t.perm[p];
            out.tri_cold.push_back(c);
            out.tri_rank.push_back(rank[p]);
        }
        tri_total += mesh.tri_count;
    }
    // traversal stack: one deferred sibling per inner level of the TLAS, one
    // TLAS-leaf continuation, one per inner level of the deepest BLAS, + margin
    // (a scene without meshes needs exactly tlas_walk_depth entries - Emitter::depth is the maximum of inner records on a path, each defers one sibling at
    //  most - and gets one to spare: the LDS a block saves holds the sphere cache of the headline scene, render.hip size_sphere_cache)
    v.stack_entries = d->mesh_object_count ? tlas_walk_depth + 1 + blas_walk_depth + 2 : tlas_walk_depth + 1;

    // ---- node records: the top levels of every tree first (the LDS node cache), then the rest --------
    // LDS of a render block = its lanes' traversal stacks + the cache; four blocks share a CU's 160 KB.
    {
        const size_t stack_bytes = ((size_t)v.stack_entries + (d->mesh_object_count ? 3 : 0)) * RT_BLOCK * sizeof(uint32_t);      // as rt_scene_create sizes it
        const size_t block_budget = (size_t)160 * 1024 * RT_BLOCK / (256 * RT_WAVES_PER_SIMD);      // RT_WAVES_PER_SIMD x 256 lanes share a CU's LDS
        static const int cap = []() { const char* e = getenv("RT_LDS_NODES"); return e ? atoi(e) : 1 << 20; }();     // 0 turns the cache off
        size_t budget = stack_bytes < block_budget ? (block_budget - stack_bytes) / RT_NODE_CACHE_STRIDE : 0;
        if (budget > (size_t)cap) budget = (size_t)cap;
        // equal shares for the trees, what a small tree leaves goes to the others (in order)
        std::vector<uint32_t> hot(emitters.size(), 0);
        size_t open = 0; for (auto& e : emitters) if (!e->en.empty()) open++;
        size_t left = budget;
        for (int pass = 0; pass < 4 && left && open; pass++) {
            const size_t share = std::max<size_t>(left / open, 1);
            open = 0;
            for (size_t k = 0; k < emitters.size() && left; k++) {
                const size_t room = emitters[k]->en.size() - hot[k];
                const size_t take = std::min({room, share, left});
                hot[k] += (uint32_t)take; left -= take;
                if (emitters[k]->en.size() > hot[k]) open++;
            }
        }
        size_t n_hot = 0, n_all = 0;
        for (size_t k = 0; k < emitters.size(); k++) { n_hot += hot[k]; n_all += emitters[k]->en.size(); }
        out.nodes.assign(n_all ? n_all : 1, DevNode{});
        uint32_t hot_at = 0, cold_at = (uint32_t)n_hot;
        for (size_t k = 0; k < emitters.size(); k++) {
            const uint32_t ref = emitters[k]->write(out.nodes, hot_at, hot[k], cold_at);
            if (k == 0) {
                v.tlas_root_ref = ref;
                // refit tables of the TLAS (rt_scene_update_transforms): records by level, deepest first
                const Emitter& em = *emitters[0];
                uint32_t levels = 0;
                for (const Emitter::ENode& e : em.en) levels = std::max(levels, e.level + 1);
                std::vector<std::vector<uint32_t>> by_level(levels);
                for (uint32_t i = 0; i < em.en.size(); i++) by_level[em.en[i].level].push_back(i);
                out.refit_level_start.assign(1, 0u);
                for (uint32_t l = levels; l-- > 0;) {
                    for (uint32_t i : by_level[l]) {
                        const Emitter::ENode& e = em.en[i];
                        RefitRec r{};
                        r.node = em.number[i];
                        r.l_first = e.l_src_first; r.l_count = e.l_src_count; r.l_child = e.l_inner ? em.number[e.l] : 0u;
                        r.r_first = e.r_src_first; r.r_count = e.r_src_count; r.r_child = e.r_inner ? em.number[e.r] : 0u;
                        out.refit.push_back(r);
                    }
                    out.refit_level_start.push_back((uint32_t)out.refit.size());
                }
            } else { out.meshes[k - 1].w[0].u[3] = ref; out.blas_hot_base.push_back(hot_at); out.blas_hot_count.push_back(hot[k]); }
            hot_at += hot[k]; cold_at += (uint32_t)emitters[k]->en.size() - hot[k];
            emitters[k].reset();
        }
        v.lds_nodes = (uint32_t)n_hot;
        out.tlas_hot_records = hot.empty() ? 0u : hot[0];
        out.blas_walk_depth = blas_walk_depth;
        out.tlas_walk_depth = tlas_walk_depth;
    }
    out.info.blas_node_count = out.nodes.size() - out.info.tlas_node_count;
    out.info.blas_depth = blas_depth;
    out.info.mesh_count = n_mesh;
    out.info.triangle_count = tri_total;
    out.info.entity_count = n_ent;

    lap("BLAS walk trees, nodes, triangles");
    // ---- entities in TLAS leaf order ------------------------------------------
    emission_positions(tlas, pos_order);
    std::vector<uint32_t> erank(n_ent);
    out.tlas_order.resize(n_ent);
    for (uint32_t k = 0; k < pos_order.size(); k++) { erank[pos_order[k]] = k; out.tlas_order[k] = tlas.perm[pos_order[k]]; }
    out.entities.resize(n_ent);
    out.tlas_perm = tlas.perm;
    out.ent_box.resize((n_ent + 1) * 2);
    for (uint64_t p = 0; p < n_ent; p++) {
        const Box& bx = ebox[tlas.perm[p]];
        Word4 lo, hi; lo.f[0] = bx.lo.x; lo.f[1] = bx.lo.y; lo.f[2] = bx.lo.z; lo.f[3] = 0.0f; hi.f[0] = bx.hi.x; hi.f[1] = bx.hi.y; hi.f[2] = bx.hi.z; hi.f[3] = 0.0f;
        out.ent_box[2 * p] = lo; out.ent_box[2 * p + 1] = hi;
    }
    uint64_t inst_tris = 0;
    uint32_t xform_at = 0;
    for (uint64_t p = 0; p < n_ent; p++) {
        const uint32_t ei = tlas.perm[p];
        const RtEntity& e = d->entities[ei];
        DevEntity& de = out.entities[p];
        memset(&de, 0, sizeof(de));
        uint32_t mat = 0, tex = 0, hidx = 0; float radius = 0.0f;
        if (e.hitable_kind == RT_HITABLE_SPHERE) { const RtSphere& s = d->spheres[e.hitable_index]; mat = s.mat_id; tex = s.tex_id; radius = s.radius; }
        else if (e.hitable_kind == RT_HITABLE_PLANE) { const RtPlane& s = d->planes[e.hitable_index]; mat = s.mat_id; tex = s.tex_id; hidx = e.hitable_index; }
        else { const RtMeshObject& s = d->mesh_objects[e.hitable_index]; mat = s.mat_id; tex = s.tex_id; hidx = s.mesh_index; inst_tris += d->meshes[s.mesh_index].tri_count; }
        if (e.translation_count > 0) { const RtV3& t0 = d->translations[e.translation_first]; de.w[0].f[0] = t0.x; de.w[0].f[1] = t0.y; de.w[0].f[2] = t0.z; }
        de.w[0].f[3] = radius;
        de.w[1].u[0] = e.hitable_kind | (e.translation_count << 8) | (e.rotation_count << 16);
        de.w[1].u[1] = hidx;
        de.w[1].u[2] = xform_at;                       // (host_make_xforms lays the records out in this order)
        xform_at += e.translation_count + 2u * e.rotation_count;
        de.w[1].u[3] = mat;
        de.w[2].u[0] = tex; de.w[2].u[1] = ei; de.w[2].i[2] = e.id; de.w[2].u[3] = erank[p];
    }
    out.info.instanced_triangle_count = inst_tris;
    {
        bool loose = false;
        host_make_xforms(d->entities, tlas.perm, d->translations, d->rotations, out.xforms, loose);
        v.loose_boxes = loose ? 1u : 0u;
    }

    lap("entity records");
    // ---- planes / materials / textures / lights ---------------------------------
    out.planes.resize(d->plane_count ? d->plane_count : 1);
    for (uint64_t i = 0; i < d->plane_count; i++) {
        const RtPlane& s = d->planes[i]; DevPlane& p = out.planes[i];
        p.w[0].f[0] = s.normal.x; p.w[0].f[1] = s.normal.y; p.w[0].f[2] = s.normal.z; p.w[0].f[3] = s.radius;
        p.w[1].f[0] = s.u_basis.x; p.w[1].f[1] = s.u_basis.y; p.w[1].f[2] = s.u_basis.z; p.w[1].u[3] = s.has_radius ? 1u : 0u;
        p.w[2].f[0] = s.v_basis.x; p.w[2].f[1] = s.v_basis.y; p.w[2].f[2] = s.v_basis.z; p.w[2].f[3] = 0.0f;
    }
    // cast_light_ray_to_lamp (implementation.rs:535-552) only looks past an occluder whose
    // scatter yields a refraction, i.e. a MatDielectric; without any, the first hit found decides.
    v.shadow_any_hit = 1;
    for (uint64_t i = 0; i < n_mat; i++) if (d->materials[i].kind == RT_MAT_DIELECTRIC) v.shadow_any_hit = 0;
    // what classify_kernel (render.hip) finds out per tile by tracing 16 rays, where the scene's materials leave nothing to find out
    v.uniform_cost_class = n_mat ? 0u : 0xffu;
    for (uint64_t i = 0; i < n_mat; i++) {
        const RtMaterial& m = d->materials[i];
        const uint32_t c = m.kind == RT_MAT_DIELECTRIC ? 0u : ((m.kind == RT_MAT_SPECULAR || m.reflectivity > 0.0f) ? 1u : 2u);
        if (i == 0) v.uniform_cost_class = c; else if (c != v.uniform_cost_class) v.uniform_cost_class = 0xffu;
        if (v.uniform_cost_class == 0xffu) break;
    }
    v.mesh_entities = d->mesh_object_count ? 1u : 0u;
    // Circle of confusion of any primary hit, bounded from the TLAS root box: camera rays meet the focal plane at
    // t = 1 and t is the depth along the view axis over the focus distance, rays of one pixel are 2 R |1 - t| apart at t
    // and the image is t |horizontal| wide there.  Decides whether a launch needs the lens-cell round shape at all.
    {
        const Box& rb = tlas.nodes[0].box;
        const float lo[3] = {rb.lo.x, rb.lo.y, rb.lo.z}, hi[3] = {rb.hi.x, rb.hi.y, rb.hi.z};
        v.coc_rel_max = host_coc_rel_max(d->camera, lo, hi, n_ent > 0);
    }
    v.rich_shading = 0;
    for (uint64_t i = 0; i < d->light_count; i++) if (d->lights[i].kind == RT_LIGHT_LAMP) v.rich_shading = 1;
    for (uint64_t i = 0; i < n_tex; i++) if (d->textures[i].kind != RT_TEX_COLOR && d->textures[i].kind != RT_TEX_CHECKER) v.rich_shading = 1;
    v.general_xforms = v.sphere_entities = v.plane_entities = 0;
    for (uint64_t i = 0; i < n_ent; i++) {
        const RtEntity& e = d->entities[i];
        if (e.hitable_kind == RT_HITABLE_SPHERE) v.sphere_entities = 1;
        if (e.hitable_kind == RT_HITABLE_PLANE) v.plane_entities = 1;
        if (e.translation_count != 1 || e.rotation_count != 0) v.general_xforms = 1;
    }
    out.materials.resize(n_mat ? n_mat : 1);
    for (uint64_t i = 0; i < n_mat; i++) {
        const RtMaterial& s = d->materials[i]; DevMaterial& m = out.materials[i];
        m.w[0].u[0] = s.kind; m.w[0].f[1] = s.reflectivity; m.w[0].f[2] = s.fuzz; m.w[0].f[3] = s.opacity;
        m.w[1].f[0] = s.ref_index; m.w[1].f[1] = m.w[1].f[2] = m.w[1].f[3] = 0.0f;
    }
    out.textures.resize(n_tex ? n_tex : 1);
    for (uint64_t i = 0; i < n_tex; i++) {
        const RtTexture& s = d->textures[i]; DevTexture& t = out.textures[i];
        if (s.texel_first > 0xFFFFFFFFull / 3) { err = "too many texels"; return RT_E_LIMIT; }
        t.w[0].u[0] = s.kind; t.w[0].i[1] = s.child0; t.w[0].i[2] = s.child1; t.w[0].u[3] = s.width;
        t.w[1].u[0] = s.height; t.w[1].u[1] = (uint32_t)s.texel_first; t.w[1].f[2] = s.size; t.w[1].f[3] = 0.0f;
        t.w[2].f[0] = s.color.x; t.w[2].f[1] = s.color.y; t.w[2].f[2] = s.color.z; t.w[2].f[3] = 0.0f;
        t.w[3].f[0] = s.color2.x; t.w[3].f[1] = s.color2.y; t.w[3].f[2] = s.color2.z; t.w[3].f[3] = 0.0f;
    }
    out.tex_set_entries.assign(d->tex_set_entries, d->tex_set_entries + d->tex_set_entry_count);
    if (out.tex_set_entries.empty()) out.tex_set_entries.push_back(0);
    out.texels.resize(d->texel_count ? d->texel_count * 3 : 4);
    for (uint64_t i = 0; i < d->texel_count; i++) { out.texels[3 * i] = d->texels[i].x; out.texels[3 * i + 1] = d->texels[i].y; out.texels[3 * i + 2] = d->texels[i].z; }
    out.lights.resize(d->light_count ? d->light_count : 1);
    for (uint64_t i = 0; i < d->light_count; i++) {
        const RtLight& s = d->lights[i]; DevLight& l = out.lights[i];
        l.w[0].u[0] = s.kind; l.w[0].f[1] = s.intensity; l.w[0].f[2] = s.angle_deg; l.w[0].f[3] = 0.0f;
        l.w[1].f[0] = s.origin.x; l.w[1].f[1] = s.origin.y; l.w[1].f[2] = s.origin.z; l.w[1].f[3] = 0.0f;
        l.w[2].f[0] = s.direction.x; l.w[2].f[1] = s.direction.y; l.w[2].f[2] = s.direction.z; l.w[2].f[3] = 0.0f;
        l.w[3].f[0] = s.color.x; l.w[3].f[1] = s.color.y; l.w[3].f[2] = s.color.z; l.w[3].f[3] = 0.0f;
    }
    if (out.meshes.empty()) out.meshes.emplace_back();
    if (out.tri_hot.empty()) { out.tri_hot.emplace_back(); out.tri_cold.emplace_back(); out.tri_rank.push_back(0); }

    lap("materials, textures, lights");
    return RT_OK;
}

void host_make_xforms(const RtEntity* entities, const std::vector<uint32_t>& tlas_perm, const RtV3* translations, const RtRotation* rotations,
                      std::vector<Word4>& xforms, bool& loose_boxes) {
    xforms.clear();
    loose_boxes = false;
    for (uint32_t ei : tlas_perm) {
        const RtEntity& e = entities[ei];
        for (uint32_t k = 0; k < e.translation_count; k++) {
            const RtV3& t = translations[e.translation_first + k];
            Word4 w; w.f[0] = t.x; w.f[1] = t.y; w.f[2] = t.z; w.f[3] = 0.0f;
            xforms.push_back(w);
        }
        for (uint32_t k = 0; k < e.rotation_count; k++) {
            const RtRotation& r = rotations[e.rotation_first + k];
            Word4 w0, w1;
            w0.f[0] = r.axis.x; w0.f[1] = r.axis.y; w0.f[2] = r.axis.z; w0.f[3] = cosf(-r.theta);
            w1.f[0] = sinf(-r.theta); w1.f[1] = cosf(r.theta); w1.f[2] = sinf(r.theta);
            // exact identity (see rotate_cs_id in render.hip): both directions, tame axis
            const bool identity = w0.f[3] == 1.0f && w1.f[0] == 0.0f && w1.f[1] == 1.0f && w1.f[2] == 0.0f &&
                                  fabsf(r.axis.x) <= 2.0f && fabsf(r.axis.y) <= 2.0f && fabsf(r.axis.z) <= 2.0f;
            w1.u[3] = identity ? 1u : 0u;
            xforms.push_back(w0); xforms.push_back(w1);
            // rotate_about_axis assumes a unit axis (types.rs:190); with any other axis the map is no
            // rotation, Entity::hit's inverse is not its inverse, and the reference's entity boxes
            // (calculate_origin_aabb) no longer bound what Entity::hit finds -> see loose_boxes
            const float len2 = r.axis.x * r.axis.x + r.axis.y * r.axis.y + r.axis.z * r.axis.z;
            if (!identity && !(fabsf(len2 - 1.0f) <= 1.0e-4f)) loose_boxes = true;
        }
    }
    if (xforms.empty()) xforms.emplace_back();
}

// Circle of confusion of any primary hit, bounded from the TLAS root box: camera rays meet the focal plane at
// t = 1 and t is the depth along the view axis over the focus distance, rays of one pixel are 2 R |1 - t| apart at t
// and the image is t |horizontal| wide there.  Decides whether a launch needs the lens-cell round shape at all.
float host_coc_rel_max(const RtCamera& c, const float lo[3], const float hi[3], bool any_entity) {
    if (!(c.lens_radius > 0.0f)) return 0.0f;
    const bool infinite = lo[0] == -INFINITY && lo[1] == -INFINITY && lo[2] == -INFINITY && hi[0] == INFINITY && hi[1] == INFINITY && hi[2] == INFINITY;
    if (!any_entity || infinite) return INFINITY;
    const double ctr[3] = {c.lower_left_corner.x + 0.5 * c.horizontal.x + 0.5 * c.vertical.x - c.origin.x,
                           c.lower_left_corner.y + 0.5 * c.horizontal.y + 0.5 * c.vertical.y - c.origin.y,
                           c.lower_left_corner.z + 0.5 * c.horizontal.z + 0.5 * c.vertical.z - c.origin.z};
    const double f = std::sqrt(ctr[0] * ctr[0] + ctr[1] * ctr[1] + ctr[2] * ctr[2]);
    const double hlen = std::sqrt((double)c.horizontal.x * c.horizontal.x + (double)c.horizontal.y * c.horizontal.y + (double)c.horizontal.z * c.horizontal.z);
    double zmin = 1e300, zmax = -1e300;
    for (int k = 0; k < 8; k++) {
        const double px = (k & 1) ? hi[0] : lo[0], py = (k & 2) ? hi[1] : lo[1], pz = (k & 4) ? hi[2] : lo[2];
        const double z = ((px - c.origin.x) * ctr[0] + (py - c.origin.y) * ctr[1] + (pz - c.origin.z) * ctr[2]) / f;
        zmin = std::min(zmin, z); zmax = std::max(zmax, z);
    }
    if (f > 0.0 && hlen > 0.0 && zmin > 1e-6 * f && std::isfinite(zmax)) {
        const double t_lo = zmin / f, t_hi = zmax / f;
        const double worst = std::max(std::fabs(1.0 - t_lo) / t_lo, std::fabs(1.0 - t_hi) / t_hi);
        return (float)(1.05 * 2.0 * c.lens_radius * worst / hlen);
    }
    return INFINITY;
}

int host_create_render_chunks(uint32_t width, uint32_t height, uint32_t count_x, uint32_t count_y, RtRenderChunk* out, uint32_t capacity) {
    if (count_x == 0 || count_y == 0 || !out) return RT_E_INVALID;
    if ((uint64_t)count_x * count_y > capacity) return RT_E_INVALID;
    // viewport.rs:35-43: equal segments, the remainder goes to the last one
    const uint32_t sw = width / count_x, rw = width % count_x, sh = height / count_y, rh = height % count_y;
    uint32_t n = 0;
    for (uint32_t j = 0; j < count_y; j++)
        for (uint32_t i = 0; i < count_x; i++)
            out[n++] = RtRenderChunk{i * sw, j * sh, i == count_x - 1 ? sw + rw : sw, j == count_y - 1 ? sh + rh : sh};
    return (int)n;
}

void host_v3_to_rgba(const float* rgb, uint64_t n_pixels, uint8_t* o) {
    // `(255.0 * c) as u8`: truncating, saturating, NaN -> 0 (src/rgba.rs:6-12)
    auto cvt = [](float c) -> uint8_t { const float s = 255.0f * c; if (!(s > 0.0f)) return 0; if (s >= 255.0f) return 255; return (uint8_t)s; };
    for (uint64_t i = 0; i < n_pixels; i++) { o[4 * i] = cvt(rgb[3 * i]); o[4 * i + 1] = cvt(rgb[3 * i + 1]); o[4 * i + 2] = cvt(rgb[3 * i + 2]); o[4 * i + 3] = 255; }
}
