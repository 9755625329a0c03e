// Flattens the host scene + built BVH into the 64-byte records the kernels traverse.
#pragma once
#include <string>
#include <vector>

#include "bvh_build.h"
#include "device_types.h"

namespace rtx {

struct FlatScene {
    std::vector<BvhNode> nodes;         // internal nodes, DFS preorder (root = 0)
    std::vector<WideNode> wnodes;       // the same tree, two levels per 128-byte record (root = 0)
    uint32_t wide_root_ref = NO_PRIM;   // wide node 0, or the root leaf reference
    uint32_t wide_depth = 0;            // levels of wide nodes above the deepest leaf
    std::vector<Prim> prims;            // leaf order
    std::vector<uint32_t> prim_obj;     // leaf order -> object index
    std::vector<Xform> xforms;
    std::vector<MaterialRec> mats;
    std::vector<MatInstr> prog;
    std::vector<TextureRec> tex;
    std::vector<uint8_t> tex_data;
    std::vector<LightRec> lights;
    std::vector<uint32_t> preorder_sizes;  // object count of every tree node, preorder (digest)
    uint32_t root_ref = NO_PRIM;
    float root_mn[3] = {0, 0, 0}, root_mx[3] = {0, 0, 0};
    uint32_t n_leaves = 0;
    bool any_xform = false;             // more than an eighth of the objects have a transform of their own
    bool xf0_identity = false;          // transform 0 is exactly the identity matrix pair
    uint32_t max_bounce = 0;            // largest number of bounce leaves of any material in use
    uint32_t bounce_mat_lo = 1, bounce_mat_hi = 0;  // range of the material ids that have bounce leaves (empty: lo > hi)
    float total_sa = 0.0f;              // BvhTree::total_sa (bvh.rs:419-426)
    bool nonfinite_boxes = false;       // some node box has a NaN / infinite coordinate: the kernels then decide every box by the reference's divisions
    float overhang = 0.0f;              // how far a hit point can lie outside its object's reference box (flatten.cpp)
};

// Largest distance by which a hit can lie outside its object's reference box; per_object gets
// every object's own value.  Expects validated object indices.
float hit_overhang(const rtx_scene_desc &d, std::vector<float> &per_object);

int flatten_scene(const rtx_scene_desc &d, const BuiltBvh &bvh, FlatScene &out, std::string &err, int threads = 1);

}  // namespace rtx
