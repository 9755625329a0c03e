// Host BVH builder that reproduces the reference's tree topology exactly
// (Bvh::new -> BvhTree::new -> bvh_split(SAH), src/bvh.rs:13-26, 195-357) but in place,
// O(n log n), and task-parallel over subtrees.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "hostmath.h"

namespace rtx {

struct BuildNode {
    float mn[3], mx[3];
    int32_t left, right;      // child node indices; -1 for a leaf
    uint32_t first, count;    // range in leaf_order (all nodes: the objects below)
};

struct BuiltBvh {
    std::vector<BuildNode> nodes;       // nodes[0] is the root (absent for an empty scene)
    std::vector<uint32_t> leaf_order;   // object indices, DFS left-first = reference Vec order
    uint32_t depth = 0;                 // BvhTree::get_depth (bvh.rs:404-409)
};

// Object::get_bounding_box (object.rs:290-313) of object i of the description.
void object_bounding_box(const rtx_scene_desc &d, uint32_t i, float mn[3], float mx[3]);

constexpr int kSahBuckets = 12;   // N_BUCKETS (bvh.rs:241)

// The three passes over a node's objects that one split of bvh_split_by_sah needs (bvh.rs:195-321), as a
// service: the top of a big tree -- nodes of millions of objects, where a host build has one thread per node --
// runs them on the GPU (gpu_build.hip: SURVEY 8 f1, "GPU-assisted binning"); the decisions in between (widest
// axis, SAH cost of the 11 splits) stay with build_bvh, in the reference's f32 arithmetic.  Every result is a
// min / max / count over the node's objects or a stable partition, so it does not depend on who computes it:
// the topology is the host build's, bit for bit (tests/test_host.py, tests/test_gpu_build.py).
// All ranges [b, e) index the permutation of the objects that the splitter keeps (identity at begin()).
struct TopSplitter {
    virtual ~TopSplitter() {}
    // boxes: n x (min xyz, max xyz); cents: n x xyz.  false: cannot run (build_bvh then splits everything itself)
    virtual bool begin(const float *boxes, const float *cents, uint32_t n) = 0;
    // range of the centroids (bvh.rs:197-206)
    virtual bool centroid_range(uint32_t b, uint32_t e, float cmin[3], float cmax[3]) = 0;
    // bucket of every object along `dim` (bvh.rs:254-260, `as i32` saturating, 12 -> 11), kept for partition();
    // per bucket the number of objects and the union of their boxes with the ORIGIN (bvh.rs:251,269).
    // *bad: some index fell outside 0..11 (the reference's index panic)
    virtual bool bin(uint32_t b, uint32_t e, int dim, float cmin_dim, float dim_range, uint32_t count[kSahBuckets],
                     float bb[kSahBuckets][6], bool *bad) = 0;
    // stable partition: objects whose bucket is <= best_i first (bvh.rs:306-319)
    virtual bool partition(uint32_t b, uint32_t e, int best_i) = 0;
    // the permutation -> idx_out (n entries)
    virtual bool finish(uint32_t *idx_out) = 0;
};
// The same passes on the host, one thread (what build_bvh does inside its recursion): for tests of the top phase.
TopSplitter *new_host_top_splitter();

// Returns RTX_OK, or RTX_ERR_DEGENERATE when a split cannot separate the objects (the
// reference recurses until its stack overflows in that case, bvh.rs:221-230,334-357).
// top != nullptr: nodes of at least top_min objects are split through `top`, breadth first, until the pieces are
// small and many enough for the host threads (n / 64 objects); if `top` fails the host builds everything.
int build_bvh(const rtx_scene_desc &d, int threads, BuiltBvh &out, std::string &err, TopSplitter *top = nullptr,
              uint32_t top_min = 1u << 20);

}  // namespace rtx
