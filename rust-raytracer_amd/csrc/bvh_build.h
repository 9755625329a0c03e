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

// Returns RTX_OK, or RTX_ERR_DEGENERATE when a split cannot separate the objects (the
// reference recurses until its stack overflows in that case, bvh.rs:221-230,334-357).
int build_bvh(const rtx_scene_desc &d, int threads, BuiltBvh &out, std::string &err);

}  // namespace rtx
