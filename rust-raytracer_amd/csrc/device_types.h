// Data layout shared by the host flattener and the HIP kernels.
//
// Everything the kernels chase per ray is a 64-byte, 64-byte-aligned record so that one
// node or primitive visit is exactly one cache line (four dwordx4 loads per lane).
#pragma once
#include <stdint.h>

namespace rtx {

// One internal BVH node (BvhTree::Node, bvh.rs:323-326).  The boxes of BOTH children are
// stored in the parent, so a visit decides both descents from one line; the reference
// tests a node's own box on entry (bvh.rs:362,377), which is the same predicate evaluated
// one level later.  Child reference: bit 31 clear = internal node index; bit 31 set =
// leaf, bits 30..3 = first primitive (leaf order), bits 2..0 = primitive count (<= 4).
struct alignas(64) BvhNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t left, right;
    uint32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "node must be one cache line");

// Two levels of the same tree fetched as ONE 128-byte line (the L2 / fabric line size on
// MI355X): the boxes of up to four descendants of a reference node, obtained by repeatedly
// replacing the internal slot of largest surface area by its two children.  Every slot is
// a node of the reference tree with the reference's box; leaves are never merged or
// split, so leaf contents, leaf order and leaf boxes are exactly bvh.rs's.  Slot reference:
// bit 31 set = leaf (same encoding as BvhNode), else index of the wide node of that
// internal reference node; WIDE_EMPTY = unused slot.  Boxes are stored per axis for the
// four slots (SoA) so one dwordx4 load feeds four slab computations.
struct alignas(128) WideNode {
    float mnx[4], mny[4], mnz[4];
    float mxx[4], mxy[4], mxz[4];
    uint32_t ref[4];
    float ovh[4];  // per slot: how far a hit in that subtree can lie outside its object's box (0 in most scenes)
};
static_assert(sizeof(WideNode) == 128, "wide node must be one 128-byte line");
constexpr uint32_t WIDE_EMPTY = 0xFFFFFFFFu;

constexpr uint32_t LEAF_BIT = 0x80000000u;
constexpr uint32_t NO_PRIM = 0xFFFFFFFFu;
constexpr uint32_t INVALID_PRIM = 0xFFFFFFFEu;  // sample slot outside the image

// One primitive in leaf order (Object, object.rs:20-26, minus the per-object matrices,
// which live in the Xform table).  v[0..11]: sphere = centre, radius, radius^2;
// triangle = a,b,c; quad = a,b,c,d.  n = normalize((b-a)x(c|d - a)) precomputed with the
// reference's arithmetic (object.rs:168,191).  meta: bits 1..0 type, 14..2 xform id, bit 15 =
// that transform is exactly the identity, 31..16 material id.
struct alignas(64) Prim {
    float v[12];
    float n[3];
    uint32_t meta;
};
static_assert(sizeof(Prim) == 64, "primitive must be one cache line");

struct Xform {
    float o2w[16];
    float w2o[16];
};

// Flattened material: a post-order program evaluated by the resolve kernel with a small
// accumulator stack (material.rs:118-206).
enum MatOp : uint32_t {
    OP_NONE = 0,        // val = gray 0.5                       (material.rs:204)
    OP_PHONG = 1,       // val = Phong(kd,ks,shininess)         (material.rs:144-184)
    OP_REFLECT = 2,     // val = colour of bounce child #arg    (material.rs:185-193)
    OP_REFRACT = 3,     // val = colour of bounce child #arg    (material.rs:194-203)
    OP_BEGIN = 4,       // push acc = (0,0,0)                   (material.rs:143)
    OP_ITEM = 5,        // acc = acc + coef * val               (material.rs:132-143)
    OP_END = 6          // val = pop acc
};
struct MatInstr {
    uint32_t op;
    uint32_t arg;       // bounce child slot for REFLECT/REFRACT
    float p0, p1, p2;   // Phong kd,ks,shininess | refraction index | ITEM coefficient
    uint32_t pad[3];
};
struct MaterialRec {
    uint32_t prog_first, prog_count;
    uint32_t n_bounce;          // bounce leaves (children spawned per hit)
    uint32_t tex_kind, tex_id;
    float tex_color[3];
};
struct TextureRec {
    uint32_t w, h;
    uint64_t offset;
};
struct LightRec {
    uint32_t type;
    float pos[3];
    float dir[3];
    float angle_deg;
    float color[3];
    // Cone lights: cos of the cut-off angle, for the cheap pre-filter in front of the reference's
    // atan2 comparison (light.rs:113-116); 2 = always outside, -2 = never outside (cut-off >= pi).
    float cos_cut;
    // Cone lights: cos of (cut-off - 8.01 degrees): inside that, the falloff's cone factor clamp((cut-off -
    // angle) / 8, 0, 1) is exactly 1 (light.rs:65-77) and the atan2 behind it need not run; 2 = never.
    float cos_full;
    // Directional lights: the direction of the shadow query, Ray::new(point, -direction).offset(1e-4)
    // (light.rs:109-110, ray.rs:11-16,43-46): ndir1 = normalize(-direction) moves the origin, ndir2 = normalize(ndir1)
    // is the ray's direction -- two square roots and two divisions per query that depend on the light alone, done
    // once by the flattener in the same f32 operations.
    float ndir1[3], ndir2[3];
};

enum : uint32_t { LIGHT_AMBIENT = 0, LIGHT_POINT = 1, LIGHT_DIRECTIONAL = 2, LIGHT_CONE = 3 };  // light.rs:12-17

constexpr int MAX_LIGHTS = 16;
constexpr int MAX_BOUNCE_LEAVES = 4;   // bounce leaves per material
constexpr int MAX_COMP_DEPTH = 4;      // nesting of Composition
#ifndef RTX_LDS_STACK
#define RTX_LDS_STACK 12
#endif
constexpr int LDS_STACK = RTX_LDS_STACK;          // per-lane traversal stack entries kept in LDS (the host sizes the spill for this, the smaller number)
constexpr int LDS_STACK_DEEP = RTX_LDS_STACK + 2; // ... in the kernel variants for scenes of long rays, which list their waiting hits in global memory instead of LDS
constexpr int MAX_TREE_DEPTH = 200;    // LDS_STACK + global spill entries per lane

}  // namespace rtx
