// Launch interface between the host API (rtx_api.cpp) and the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

namespace rtx {

// Scene as the kernels see it: pointers into HBM plus the small constants.
struct DeviceScene {
    const BvhNode *nodes;       // binary nodes: RTX_FLAG_REFERENCE_TRAVERSAL only
    const WideNode *wnodes;     // the same tree, two levels per 128-byte line (production traversal)
    const Prim *prims;
    const Xform *xforms;
    const MaterialRec *mats;
    const MatInstr *prog;
    const TextureRec *tex;
    const uint8_t *tex_data;
    const LightRec *lights;
    const uint32_t *prim_obj;   // leaf-order primitive -> object index of the description
    uint32_t n_prims, n_lights, n_wnodes;
    uint32_t root_ref;          // child-reference encoding; NO_PRIM for an empty scene
    uint32_t wroot_ref;         // root in the wide tree (wide node 0 or the root leaf)
    uint32_t xf0_identity;      // transform 0 is exactly the identity
    float root_mn[3], root_mx[3];
    float coord_scale;          // largest |coordinate| of the root box (pruning slack)
    float overhang;             // largest distance by which a hit can lie outside its object's box (0 for most scenes)
    float background[3];
    float c2w[16];
    uint32_t width, height;
};

// Counter block (uint32 words) in device memory, zeroed before every batch.
enum {
    CTR_LEVEL_COUNT = 0,    // [8]  path nodes per bounce level
    CTR_SHADOW_COUNT = 8,   // [8][MAX_LIGHTS] shadow-queue entries per level and light
    CTR_OVERFLOW = 8 + 8 * MAX_LIGHTS,
    CTR_VISIT_NODES_LO = CTR_OVERFLOW + 2,  // trace_closest: 64-bit counters as lo/hi pairs (8-byte aligned)
    CTR_VISIT_NODES_HI,
    CTR_VISIT_PRIMS_LO,
    CTR_VISIT_PRIMS_HI,
    CTR_SVISIT_NODES_LO,                    // trace_shadow
    CTR_SVISIT_NODES_HI,
    CTR_SVISIT_PRIMS_LO,
    CTR_SVISIT_PRIMS_HI,
    CTR_MAX_STEPS,                          // most records visited by a single ray (COUNT)
    CTR_UTIL = 148,                         // [2][5] 64-bit lane-utilisation census (RTX_UTIL_STATS builds)
    CTR_WORDS = 168
};
// Queue heads of the persistent trace kernels live in their own buffer (Frame::work): per
// bounce level WORK_QUEUES queues (0 = trace_closest, 1 + l = trace_shadow of light l), each
// split into QPARTS sub-queues whose head counters sit on separate 128-byte lines -- one
// hot counter sustains only ~90 atomics/us, which bounds small frames (multi-GPU shares).
enum {
    WORK_QUEUES = 1 + MAX_LIGHTS,
    QPARTS = 16,
    QHEAD_STRIDE = 32,  // words
    WORK_WORDS = 8 * WORK_QUEUES * QPARTS * QHEAD_STRIDE
};
static_assert(CTR_VISIT_NODES_LO % 2 == 0 && CTR_MAX_STEPS < CTR_WORDS, "64-bit counters must be 8-byte aligned");

// Wavefront state of one batch of primary samples ("path nodes": one per traced ray).
struct Frame {
    float4 *ray_o;      // xyz origin,    w = hit t
    float4 *ray_d;      // xyz direction, w = bits of hit primitive (leaf order)
    uint2 *link;        // x = lit-light mask, y = first bounce child (NO_PRIM: none)
    float4 *color;      // resolved colour of the node
    uint32_t *shadow_q; // [level][n_lights][q_cap] node ids waiting for a shadow query
    uint32_t *counters; // CTR_WORDS
    uint32_t *work;     // WORK_WORDS queue heads
    uint32_t *spill;    // traversal-stack overflow, [MAX_TREE_DEPTH-LDS_STACK][threads] of (ref, t)
    uint32_t node_cap, q_cap;
};

struct BatchInfo {
    uint32_t q0, npix;      // local pixel slots [q0, q0+npix) of this rank
    uint32_t s0, ns;        // samples [s0, s0+ns)
    uint32_t spp;           // total samples per pixel of the render
    uint32_t use_jitter;
    uint32_t seed_lo, seed_hi;
    uint32_t tile, rank, nranks, tiles_x, tiles_y;
    uint32_t max_levels;    // = max_ray_bounces
};

struct LaunchCfg {
    hipStream_t stream;
    int persistent_blocks;  // grid of the persistent trace kernels
    bool count_visits;
    bool reference_traversal;
};

void launch_raygen(const DeviceScene &sc, const Frame &fr, const BatchInfo &bi, const LaunchCfg &lc);
void launch_trace_closest(const DeviceScene &sc, const Frame &fr, uint32_t level, const LaunchCfg &lc);
void launch_shade_forward(const DeviceScene &sc, const Frame &fr, uint32_t level, uint32_t max_levels,
                          const LaunchCfg &lc);
void launch_trace_shadow(const DeviceScene &sc, const Frame &fr, uint32_t level, const LaunchCfg &lc);
// Resolves the colours of a level's path nodes (material.rs:118-206).  RESOLVE_LEAVES: misses
// and hits without bounce children (ready once the level's shadow answers exist);
// RESOLVE_PARENTS: hits with bounce children (need the deeper levels); RESOLVE_ALL: both.
enum ResolvePhase : uint32_t { RESOLVE_ALL = 0, RESOLVE_LEAVES = 1, RESOLVE_PARENTS = 2 };
void launch_resolve(const DeviceScene &sc, const Frame &fr, uint32_t level, ResolvePhase phase, const LaunchCfg &lc);
// Adds the batch's sample colours to `sum` in sample order (world.rs:54-61); on the last
// sample chunk divides by spp, clamps and quantises (world.rs:62-63, color.rs:25-31).
void launch_accumulate(const DeviceScene &sc, const Frame &fr, const BatchInfo &bi, float4 *sum,
                       uint8_t *tiles_rgb8, float *tiles_f32_or_null, const LaunchCfg &lc);
// Hit export for rtx_trace_primary / rtx_trace_rays: level-0 nodes -> (object id, t).
void launch_export_hits(const DeviceScene &sc, const Frame &fr, uint32_t n, uint32_t *out_obj, float *out_t,
                        const LaunchCfg &lc);
// Loads caller rays (origin, direction) into level 0, normalising as Ray::new does.
void launch_load_rays(const Frame &fr, const float *org_dir, uint32_t n, const LaunchCfg &lc);
void launch_export_colors(const Frame &fr, uint32_t n, float *out_rgb, const LaunchCfg &lc);
int query_persistent_blocks(int device);

// rtx_render_device_noalloc: the wavefront state of a small render lives in a file-scope
// `static __device__` array (kernels.hip) instead of hipMalloc'ed buffers.  Every GPU holds its
// own copy at its own address: noalloc_scratch_address looks it up (hipGetSymbolAddress) with
// `device` current, once per device, and caches it.
constexpr size_t NOALLOC_SCRATCH_BYTES = 64ull << 20;
hipError_t noalloc_scratch_address(int device, void **ptr);

}  // namespace rtx
