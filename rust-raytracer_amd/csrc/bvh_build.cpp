// See bvh_build.h.  Every arithmetic step that decides the topology follows the
// reference line by line (same f32 operations in the same order); what differs is the
// bookkeeping: index ranges partitioned in place instead of Vec<Object> moves, object
// boxes and centroids computed once, subtrees built concurrently.
#include "bvh_build.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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

constexpr int kBuckets = kSahBuckets;
constexpr uint32_t kLeaf = 4;  // BVH_LEAF_SIZE (bvh.rs:331)

// bvh_split (bvh.rs:208-216): widest axis of the centroid range; y beats x, z beats the running maximum, only if
// strictly greater
inline int widest_axis(const float cmin[3], const float cmax[3]) {
    float diff[3] = {cmax[0] - cmin[0], cmax[1] - cmin[1], cmax[2] - cmin[2]};
    int dim = 0;
    float mx = diff[0];
    if (diff[1] > mx) {
        mx = diff[1];
        dim = 1;
    }
    if (diff[2] > mx) dim = 2;
    return dim;
}

// Rust `as i32` of trunc(12 (c - min) / range): NaN -> 0, saturating; 12 -> 11 (bvh.rs:254-260)
inline int bucket_of(float c, float cmin_dim, float dim_range) {
    float f = 12.0f * (c - cmin_dim) / dim_range;
    float ft = std::trunc(f);
    int bi;
    if (!(ft == ft)) bi = 0;
    else if (ft >= 2147483648.0f) bi = 2147483647;
    else if (ft <= -2147483648.0f) bi = -2147483647 - 1;
    else bi = (int)ft;
    if (bi == kBuckets) bi -= 1;
    return bi;
}

// bvh_split_by_sah's choice (bvh.rs:275-304): cost_i = 0.125 + (nL SA(L) + nR SA(R)) / SA(centroid box), strict <,
// the first minimum wins
inline int sah_choose(const uint32_t count[kBuckets], const Box bb[kBuckets], const float cmin[3], const float cmax[3]) {
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
    return best_i;
}

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
    const int dim = widest_axis(cmin, cmax);
    // bvh_split_by_sah (bvh.rs:245-321); n > 4 here so the Basic branch is unreachable
    uint32_t count[kBuckets] = {0};
    Box bb[kBuckets];
    for (int i = 0; i < kBuckets; i++)
        for (int k = 0; k < 3; k++) bb[i].mn[k] = bb[i].mx[k] = 0.0f;  // AABB::empty() = the origin (bvh.rs:54-56)
    float dim_range = cmax[dim] - cmin[dim];
    for (uint32_t i = b; i < e; i++) {
        uint32_t o = idx[i];
        const int bi = bucket_of(cents[o][dim], cmin[dim], dim_range);
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
    const int best_i = sah_choose(count, bb, cmin, cmax);
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

static_assert(sizeof(Box) == 24 && sizeof(V3) == 12, "TopSplitter takes the boxes and centroids as packed floats");

// TopSplitter on the host, single-threaded: the loops of Ctx::build behind the interface.
struct HostTopSplitter : TopSplitter {
    const Box *boxes = nullptr;
    const V3 *cents = nullptr;
    std::vector<uint32_t> idx, scratch;
    std::vector<uint8_t> bucket;
    bool begin(const float *b, const float *c, uint32_t n) override {
        boxes = reinterpret_cast<const Box *>(b);
        cents = reinterpret_cast<const V3 *>(c);
        idx.resize(n);
        for (uint32_t i = 0; i < n; i++) idx[i] = i;
        scratch.resize(n);
        bucket.resize(n);
        return true;
    }
    bool centroid_range(uint32_t b, uint32_t e, float cmin[3], float cmax[3]) override {
        const float big = std::numeric_limits<float>::max();
        for (int k = 0; k < 3; k++) {
            cmin[k] = big;
            cmax[k] = -big;
        }
        for (uint32_t i = b; i < e; i++)
            for (int k = 0; k < 3; k++) {
                cmin[k] = std::fmin(cmin[k], cents[idx[i]][k]);
                cmax[k] = std::fmax(cmax[k], cents[idx[i]][k]);
            }
        return true;
    }
    bool bin(uint32_t b, uint32_t e, int dim, float cmin_dim, float dim_range, uint32_t count[kBuckets], float bb[kBuckets][6],
             bool *bad) override {
        *bad = false;
        for (int i = 0; i < kBuckets; i++) {
            count[i] = 0;
            for (int k = 0; k < 6; k++) bb[i][k] = 0.0f;
        }
        for (uint32_t i = b; i < e; i++) {
            const uint32_t o = idx[i];
            const int bi = bucket_of(cents[o][dim], cmin_dim, dim_range);
            if (bi < 0 || bi >= kBuckets) {
                *bad = true;
                return true;
            }
            count[bi] += 1;
            for (int k = 0; k < 3; k++) {
                bb[bi][k] = std::fmin(bb[bi][k], boxes[o].mn[k]);
                bb[bi][3 + k] = std::fmax(bb[bi][3 + k], boxes[o].mx[k]);
            }
            bucket[i] = (uint8_t)bi;
        }
        return true;
    }
    bool partition(uint32_t b, uint32_t e, int best_i) override {
        uint32_t nl = 0;
        for (uint32_t i = b; i < e; i++) nl += bucket[i] <= best_i;
        uint32_t pl = b, pr = b + nl;
        for (uint32_t i = b; i < e; i++) {
            if (bucket[i] <= best_i) scratch[pl++] = idx[i];
            else scratch[pr++] = idx[i];
        }
        std::copy(scratch.begin() + b, scratch.begin() + e, idx.begin() + b);
        return true;
    }
    bool finish(uint32_t *out) override {
        std::copy(idx.begin(), idx.end(), out);
        return true;
    }
};

// The top of the tree through a TopSplitter: nodes of at least `piece` objects are split breadth first (the same
// decisions as Ctx::build, from the splitter's per-node results); what is left -- `frontier`, pieces of fewer
// objects -- is for the host threads.  `tops` lists the split nodes in creation order (a parent before its
// children).  Returns 1 done, 0 the splitter failed (nothing usable), -1 degenerate input.
int split_top(Ctx &c, TopSplitter &top, uint32_t n, uint32_t piece, std::vector<std::array<uint32_t, 3>> &frontier,
              std::vector<uint32_t> &tops) {
    if (!top.begin(&c.boxes[0].mn[0], reinterpret_cast<const float *>(c.cents.data()), n)) return 0;
    std::vector<std::array<uint32_t, 3>> todo{{0u, 0u, n}};
    for (size_t at = 0; at < todo.size(); at++) {
        const uint32_t ni = todo[at][0], b = todo[at][1], e = todo[at][2], cnt = e - b;
        if (cnt < piece || cnt <= kLeaf) {
            frontier.push_back(todo[at]);
            continue;
        }
        BuildNode &nd = c.nodes[ni];
        nd.first = b;
        nd.count = cnt;
        float cmin[3], cmax[3];
        if (!top.centroid_range(b, e, cmin, cmax)) return 0;
        const int dim = widest_axis(cmin, cmax);
        uint32_t count[kBuckets];
        float bbf[kBuckets][6];
        bool bad = false;
        if (!top.bin(b, e, dim, cmin[dim], cmax[dim] - cmin[dim], count, bbf, &bad)) return 0;
        if (bad) return -1;
        Box bb[kBuckets];
        for (int i = 0; i < kBuckets; i++)
            for (int k = 0; k < 3; k++) {
                bb[i].mn[k] = bbf[i][k];
                bb[i].mx[k] = bbf[i][3 + k];
            }
        const int best_i = sah_choose(count, bb, cmin, cmax);
        uint32_t nl = 0;
        for (int i = 0; i <= best_i; i++) nl += count[i];
        if (nl == 0 || nl == cnt) return -1;
        if (!top.partition(b, e, best_i)) return 0;
        const uint32_t ci = c.next_node.fetch_add(2);
        nd.left = (int32_t)ci;
        nd.right = (int32_t)ci + 1;
        tops.push_back(ni);
        todo.push_back({ci, b, b + nl});
        todo.push_back({ci + 1, b + nl, e});
    }
    return top.finish(c.idx.data()) ? 1 : 0;
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

TopSplitter *new_host_top_splitter() { return new HostTopSplitter(); }

int build_bvh(const rtx_scene_desc &d, int threads, BuiltBvh &out, std::string &err, TopSplitter *top, uint32_t top_min) {
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
    c.par_threshold = std::max<uint32_t>(1u << 14, n / 256);
    bool built = false;
    if (top && n >= std::max<uint32_t>(top_min, 64u * (kLeaf + 1))) {
        std::vector<std::array<uint32_t, 3>> frontier;
        std::vector<uint32_t> tops;
        uint32_t pieces = 64;
        if (const char *tp = std::getenv("RTX_TOP_PIECES")) pieces = (uint32_t)std::max(2, std::atoi(tp));
        const int rc = split_top(c, *top, n, std::max<uint32_t>(n / pieces, kLeaf + 1), frontier, tops);
        if (rc < 0) c.degenerate.store(true);
        if (rc > 0) {
            // the pieces, largest first, on the host threads (each piece single-threaded: there are >= 64 of them)
            std::sort(frontier.begin(), frontier.end(),
                      [](const std::array<uint32_t, 3> &a, const std::array<uint32_t, 3> &b) { return a[2] - a[1] > b[2] - b[1]; });
            c.spare_threads.store(0);
            c.par_threshold = 0xFFFFFFFFu;
            std::atomic<size_t> next{0};
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; t++)
                pool.emplace_back([&]() {
                    for (size_t k = next.fetch_add(1); k < frontier.size(); k = next.fetch_add(1))
                        c.build(frontier[k][0], frontier[k][1], frontier[k][2]);
                });
            for (auto &t : pool) t.join();
            for (size_t k = tops.size(); k-- > 0;) {  // node box = union(left, right) (bvh.rs:353), children first
                BuildNode &me = c.nodes[tops[k]];
                const BuildNode &L = c.nodes[me.left], &R = c.nodes[me.right];
                for (int a = 0; a < 3; a++) {
                    me.mn[a] = std::fmin(L.mn[a], R.mn[a]);
                    me.mx[a] = std::fmax(L.mx[a], R.mx[a]);
                }
            }
        }
        built = rc != 0;
        if (rc == 0) {  // the splitter gave up (said aloud: the build is slower without it): nothing it did is kept
            std::fprintf(stderr, "rtx: the top of the BVH could not be split on the GPU (allocation or launch failed); the host builds the whole tree\n");
            for (uint32_t i = 0; i < n; i++) c.idx[i] = i;
            c.next_node.store(1);
        }
    }
    if (!built) {
        c.spare_threads.store(threads - 1);
        c.build(0, 0, n);
    }
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
