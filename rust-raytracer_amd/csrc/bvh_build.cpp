// See bvh_build.h.  Every arithmetic step that decides the topology follows the
// reference line by line (same f32 operations in the same order); what differs is the
// bookkeeping: index ranges partitioned in place instead of Vec<Object> moves, object
// boxes and centroids computed once, subtrees built concurrently.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <limits>
#include <thread>

namespace rtx {

namespace {

struct Box {
    float mn[3], mx[3];
};

inline void range3(const V3 *p, int n, float mn[3], float mx[3]) {
    // utils::component_wise_range (utils.rs:20-29)
    const float big = std::numeric_limits<float>::max();
    for (int k = 0; k < 3; k++) {
        mn[k] = big;
        mx[k] = -big;
    }
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            mn[k] = std::fmin(mn[k], p[i][k]);
            mx[k] = std::fmax(mx[k], p[i][k]);
        }
}

inline float surface_area(const float mn[3], const float mx[3]) {
    // AABB::surface_area (bvh.rs:147-150)
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.0f * dx * dy + 2.0f * dx * dz + 2.0f * dz * dy;
}

constexpr int kBuckets = 12;   // N_BUCKETS (bvh.rs:241)
constexpr uint32_t kLeaf = 4;  // BVH_LEAF_SIZE (bvh.rs:331)

struct Ctx {
    const rtx_scene_desc &d;
    std::vector<Box> boxes;
    std::vector<V3> cents;
    std::vector<uint32_t> idx, scratch;
    std::vector<uint8_t> bucket;
    std::vector<BuildNode> nodes;
    std::atomic<uint32_t> next_node{1};
    std::atomic<bool> degenerate{false};
    std::atomic<int> spare_threads{0};
    uint32_t par_threshold = 1u << 16;
    explicit Ctx(const rtx_scene_desc &desc) : d(desc) {}

    void build(uint32_t node, uint32_t b, uint32_t e);
};

void Ctx::build(uint32_t ni, uint32_t b, uint32_t e) {
    BuildNode &nd = nodes[ni];
    nd.first = b;
    nd.count = e - b;
    uint32_t n = e - b;
    if (n <= kLeaf) {  // BvhTree::Leaf (bvh.rs:336-345): box = union of the objects' boxes
        nd.left = nd.right = -1;
        for (int k = 0; k < 3; k++) {
            nd.mn[k] = n ? boxes[idx[b]].mn[k] : 0.0f;
            nd.mx[k] = n ? boxes[idx[b]].mx[k] : 0.0f;
        }
        for (uint32_t i = b + 1; i < e; i++)
            for (int k = 0; k < 3; k++) {
                nd.mn[k] = std::fmin(nd.mn[k], boxes[idx[i]].mn[k]);
                nd.mx[k] = std::fmax(nd.mx[k], boxes[idx[i]].mx[k]);
            }
        return;
    }
    if (degenerate.load(std::memory_order_relaxed)) {
        nd.left = nd.right = -1;
        return;
    }
    // bvh_split (bvh.rs:195-217): widest axis of the centroid range
    const float big = std::numeric_limits<float>::max();
    float cmin[3] = {big, big, big}, cmax[3] = {-big, -big, -big};
    for (uint32_t i = b; i < e; i++) {
        const V3 &c = cents[idx[i]];
        for (int k = 0; k < 3; k++) {
            cmin[k] = std::fmin(cmin[k], c[k]);
            cmax[k] = std::fmax(cmax[k], c[k]);
        }
    }
    float diff[3] = {cmax[0] - cmin[0], cmax[1] - cmin[1], cmax[2] - cmin[2]};
    int dim = 0;
    float mx = diff[0];
    if (diff[1] > mx) {
        mx = diff[1];
        dim = 1;
    }
    if (diff[2] > mx) dim = 2;
    // bvh_split_by_sah (bvh.rs:245-321); n > 4 here so the Basic branch is unreachable
    uint32_t count[kBuckets] = {0};
    Box bb[kBuckets];
    for (int i = 0; i < kBuckets; i++)
        for (int k = 0; k < 3; k++) bb[i].mn[k] = bb[i].mx[k] = 0.0f;  // AABB::empty() = the origin (bvh.rs:54-56)
    float dim_range = cmax[dim] - cmin[dim];
    for (uint32_t i = b; i < e; i++) {
        uint32_t o = idx[i];
        float f = 12.0f * (cents[o][dim] - cmin[dim]) / dim_range;
        float ft = std::trunc(f);
        int bi;  // Rust `as i32`: NaN -> 0, saturating
        if (!(ft == ft)) bi = 0;
        else if (ft >= 2147483648.0f) bi = 2147483647;
        else if (ft <= -2147483648.0f) bi = -2147483647 - 1;
        else bi = (int)ft;
        if (bi == kBuckets) bi -= 1;
        if (bi < 0 || bi >= kBuckets) {  // the reference's debug_assert / index panic
            degenerate.store(true);
            nd.left = nd.right = -1;
            return;
        }
        count[bi] += 1;
        for (int k = 0; k < 3; k++) {
            bb[bi].mn[k] = std::fmin(bb[bi].mn[k], boxes[o].mn[k]);
            bb[bi].mx[k] = std::fmax(bb[bi].mx[k], boxes[o].mx[k]);
        }
        bucket[i] = (uint8_t)bi;
    }
    float sa_global = surface_area(cmin, cmax);
    float best_cost = 0.0f;
    int best_i = 0;
    for (int i = 0; i < kBuckets - 1; i++) {
        Box l = bb[0], r = bb[i + 1];
        float cl = 0.0f, cr = 0.0f;
        for (int j = 0; j <= i; j++) {
            for (int k = 0; k < 3; k++) {
                l.mn[k] = std::fmin(l.mn[k], bb[j].mn[k]);
                l.mx[k] = std::fmax(l.mx[k], bb[j].mx[k]);
            }
            cl += (float)count[j];
        }
        for (int j = i + 1; j < kBuckets; j++) {
            for (int k = 0; k < 3; k++) {
                r.mn[k] = std::fmin(r.mn[k], bb[j].mn[k]);
                r.mx[k] = std::fmax(r.mx[k], bb[j].mx[k]);
            }
            cr += (float)count[j];
        }
        float cost = 0.125f + (cl * surface_area(l.mn, l.mx) + cr * surface_area(r.mn, r.mx)) / sa_global;
        if (i == 0 || cost < best_cost) {  // strict <, first minimum wins (bvh.rs:296-304)
            best_cost = cost;
            best_i = i;
        }
    }
    // stable partition (bvh.rs:306-319)
    uint32_t nl = 0;
    for (uint32_t i = b; i < e; i++)
        if (bucket[i] <= best_i) nl++;
    if (nl == 0 || nl == n) {  // the reference would now recurse on the same set forever
        degenerate.store(true);
        nd.left = nd.right = -1;
        return;
    }
    uint32_t pl = b, pr = b + nl;
    for (uint32_t i = b; i < e; i++) {
        if (bucket[i] <= best_i) scratch[pl++] = idx[i];
        else scratch[pr++] = idx[i];
    }
    std::copy(scratch.begin() + b, scratch.begin() + e, idx.begin() + b);
    uint32_t ci = next_node.fetch_add(2);
    nd.left = (int32_t)ci;
    nd.right = (int32_t)ci + 1;
    uint32_t mid = b + nl;
    bool forked = false;
    std::thread th;
    if (nl > par_threshold && (e - mid) > par_threshold && spare_threads.fetch_sub(1) > 0) {
        forked = true;
        th = std::thread([this, ci, b, mid]() { build(ci, b, mid); });
    } else if (nl > par_threshold && (e - mid) > par_threshold) {
        spare_threads.fetch_add(1);
    }
    if (!forked) build(ci, b, mid);
    build(ci + 1, mid, e);
    if (forked) {
        th.join();
        spare_threads.fetch_add(1);
    }
    // Node box = union(left, right) (bvh.rs:353)
    const BuildNode &L = nodes[ci], &R = nodes[ci + 1];
    BuildNode &me = nodes[ni];
    for (int k = 0; k < 3; k++) {
        me.mn[k] = std::fmin(L.mn[k], R.mn[k]);
        me.mx[k] = std::fmax(L.mx[k], R.mx[k]);
    }
}

uint32_t depth_of(const std::vector<BuildNode> &nodes, uint32_t root) {
    // iterative BvhTree::get_depth (bvh.rs:404-409)
    uint32_t best = 0;
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.push_back({root, 0});
    while (!st.empty()) {
        auto [n, dpt] = st.back();
        st.pop_back();
        if (nodes[n].left < 0) {
            best = std::max(best, dpt);
            continue;
        }
        st.push_back({(uint32_t)nodes[n].left, dpt + 1});
        st.push_back({(uint32_t)nodes[n].right, dpt + 1});
    }
    return best;
}

}  // namespace

void object_bounding_box(const rtx_scene_desc &d, uint32_t i, float mn[3], float mx[3]) {
    const float *v = d.obj_verts + (size_t)i * 12;
    M4 M;
    std::memcpy(M.m, d.xforms + (size_t)d.obj_xform[i] * 32, sizeof(M.m));
    int type = d.obj_type[i];
    if (type == RTX_OBJ_SPHERE) {
        V3 c = transform_point(M, v3(v[0], v[1], v[2]));
        // radius * get_axis_scaling(M) (utils.rs:33-44)
        V3 sc = v3(magnitude(transform_vector(M, v3(1, 0, 0))), magnitude(transform_vector(M, v3(0, 1, 0))),
                   magnitude(transform_vector(M, v3(0, 0, 1))));
        V3 r = v[3] * sc;
        V3 lo = c - r, hi = c + r;
        mn[0] = lo.x; mn[1] = lo.y; mn[2] = lo.z;
        mx[0] = hi.x; mx[1] = hi.y; mx[2] = hi.z;
    } else {
        int nv = type == RTX_OBJ_QUAD ? 4 : 3;
        V3 p[4];
        for (int k = 0; k < nv; k++) p[k] = transform_point(M, v3(v[k * 3], v[k * 3 + 1], v[k * 3 + 2]));
        range3(p, nv, mn, mx);
    }
}

int build_bvh(const rtx_scene_desc &d, int threads, BuiltBvh &out, std::string &err) {
    uint32_t n = d.n_objects;
    out.nodes.clear();
    out.leaf_order.clear();
    out.depth = 0;
    if (n == 0) return RTX_OK;
    if (threads < 1) threads = 1;
    // the boxes below index the transform table: reject bad indices before touching it
    for (uint32_t i = 0; i < n; i++) {
        int type = d.obj_type[i], xf = d.obj_xform[i];
        if (type < 0 || type > 2 || xf < 0 || (uint32_t)xf >= d.n_xforms) {
            err = "object with bad type / transform index";
            return RTX_ERR_INVALID;
        }
    }
    Ctx c(d);
    c.boxes.resize(n);
    c.cents.resize(n);
    {
        std::vector<std::thread> pool;
        uint32_t chunk = (n + threads - 1) / threads;
        for (int t = 0; t < threads; t++) {
            uint32_t b = t * chunk, e = std::min(n, b + chunk);
            if (b >= e) break;
            pool.emplace_back([&, b, e]() {
                for (uint32_t i = b; i < e; i++) {
                    object_bounding_box(d, i, c.boxes[i].mn, c.boxes[i].mx);
                    // Point3::centroid(&[min,max]) (bvh.rs:200): ((0 + min) + max) / 2
                    const Box &bx = c.boxes[i];
                    c.cents[i] = v3(((0.0f + bx.mn[0]) + bx.mx[0]) / 2.0f, ((0.0f + bx.mn[1]) + bx.mx[1]) / 2.0f,
                                    ((0.0f + bx.mn[2]) + bx.mx[2]) / 2.0f);
                }
            });
        }
        for (auto &t : pool) t.join();
    }
    c.idx.resize(n);
    for (uint32_t i = 0; i < n; i++) c.idx[i] = i;
    c.scratch.resize(n);
    c.bucket.resize(n);
    c.nodes.resize((size_t)2 * n + 1);
    c.spare_threads.store(threads - 1);
    c.par_threshold = std::max<uint32_t>(1u << 14, n / 256);
    c.build(0, 0, n);
    if (c.degenerate.load()) {
        err = "BVH build cannot separate a set of more than 4 objects (identical centroids on the widest "
              "axis, or non-finite geometry); the reference recurses forever on this input (bvh.rs:221-230)";
        return RTX_ERR_DEGENERATE;
    }
    c.nodes.resize(c.next_node.load());
    out.nodes = std::move(c.nodes);
    out.leaf_order = std::move(c.idx);
    out.depth = depth_of(out.nodes, 0);
    return RTX_OK;
}

}  // namespace rtx
