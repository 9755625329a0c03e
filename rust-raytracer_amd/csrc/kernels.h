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
    const LightRec *lights;     // per-lane look-ups (a shadow query's light)
    const uint32_t *prim_obj;   // leaf-order primitive -> object index of the description
    uint32_t n_prims, n_lights, n_wnodes;
    uint32_t max_bounce;        // most bounce children any material spawns per hit
    uint32_t bounce_mat_lo, bounce_mat_hi;  // the material ids with bounce leaves lie in [lo, hi] (empty: lo > hi)
    uint32_t root_ref;          // child-reference encoding; NO_PRIM for an empty scene
    uint32_t wroot_ref;         // root in the wide tree (wide node 0 or the root leaf)
    uint32_t xf0_identity;      // transform 0 is exactly the identity
    float root_mn[3], root_mx[3];
    float coord_scale;          // largest |coordinate| of the root box (pruning slack)
    float overhang;             // largest distance by which a hit can lie outside its object's box (0 for most scenes)
    float background[3];
    float c2w[16];
    uint32_t width, height;
    // The lights again, inside the kernel arguments: a loop over the lights whose counter is the same in every lane
    // reads them with scalar loads from the argument segment (through `lights` the compiler must assume that a
    // store may have changed them: one vector-memory round trip per light and per field group, one after the
    // other, in every shading pass and every Phong term).
    LightRec lights_k[MAX_LIGHTS];
};

// Counter block (uint32 words) in device memory, zeroed before every batch.
enum {
    CTR_LEVEL_COUNT = 0,    // primary samples of the batch (written by raygen / load_rays)
    CTR_POOL = 1,           // path nodes handed out from the pool of bounce children (in runs of POOL_RUN)
    CTR_OVERFLOW = 2,
    CTR_BOUNCE_LO = 4,      // 64-bit counters as lo/hi pairs (8-byte aligned): bounce rays traced,
    CTR_BOUNCE_HI,
    CTR_SHADOW_LO,          // shadow queries traced,
    CTR_SHADOW_HI,
    CTR_VISIT_NODES_LO,     // records visited by closest-hit rays (COUNT)
    CTR_VISIT_NODES_HI,
    CTR_VISIT_PRIMS_LO,
    CTR_VISIT_PRIMS_HI,
    CTR_SVISIT_NODES_LO,    // ... by shadow rays
    CTR_SVISIT_NODES_HI,
    CTR_SVISIT_PRIMS_LO,
    CTR_SVISIT_PRIMS_HI,
    CTR_MAX_STEPS,          // most records visited by a single ray (COUNT)
    CTR_PRE_DEEP,           // primaries-first launches: primary hits whose material spawns bounce rays,
    CTR_PRE_SHALLOW,        // ... and the other primary hits (written by pre_scan_kernel)
    CTR_UTIL = 20,          // [22] 64-bit lane-utilisation census (RTX_UTIL_STATS builds)
    CTR_WORDS = 64
};
// Queue heads of the persistent path kernel live in their own buffer (Frame::work): the batch's
// primary samples are ONE queue split into QPARTS sub-queues whose head counters sit on separate
// 128-byte lines -- one hot counter sustains only ~90 atomics/us, which bounds small frames
// (multi-GPU shares).
enum {
    QPARTS = 16,
    QHEAD_STRIDE = 32,  // words
    WORK_HALF = QPARTS * QHEAD_STRIDE,
    WORK_WORDS = 2 * WORK_HALF,  // two sets of heads: a primaries-first launch runs two persistent kernels per batch
    PRE_SEG = 4096,     // primaries per block of the classification passes between them
    POOL_RUN = 128,     // bounce-child path nodes a wave reserves with one atomic
    SHADE_LIST = 192,   // words behind a wave's work stack (round 2: its shading list; now in LDS -- the diagnostics keep two entries here)
#ifndef RTX_SHADE_LDS
#define RTX_SHADE_LDS 128
#endif
    SHADE_LDS = RTX_SHADE_LDS  // hits a wave can have waiting for its shading pass (LDS; at most RTX_SHADE_MIN - 1 + 64 ever wait)
};
static_assert(CTR_BOUNCE_LO % 2 == 0 && CTR_MAX_STEPS < CTR_UTIL && CTR_UTIL + 44 <= CTR_WORDS, "64-bit counters must be 8-byte aligned");

// link.x of a path node: bits 15..0 mask of the lights that reach the hit point, 19..16 bounce
// level, 22..20 number of bounce children; link.y first bounce child (NO_PRIM: none).
constexpr uint32_t LINK_LEVEL_SHIFT = 16, LINK_NB_SHIFT = 20, LINK_LEVEL_HOLE = 15;

// Wavefront state of one batch of primary samples ("path nodes": one per traced closest-hit ray).
// Nodes [0, n0) are the primary samples; bounce children live in the pool [n0, node_cap), handed
// out to the waves in runs (unused slots of a run are marked as holes when the wave retires).
struct Frame {
    float4 *ray_o;      // xyz origin,    w = hit t
    float4 *ray_d;      // xyz direction, w = bits of hit primitive (leaf order)
    uint2 *link;        // see LINK_*
    float4 *color;      // resolved colour of the node
    uint2 *wstack;      // [wave][wstack_cap + SHADE_LIST] work stacks of the path kernel's waves (node, tag)
    uint32_t *counters; // CTR_WORDS
    uint32_t *work;     // WORK_WORDS queue heads
    uint32_t *spill;    // traversal-stack overflow, [spill entries][threads] of (ref, t)
    uint32_t node_cap, wstack_cap;
    uint32_t stripe;    // entries of the queue of primaries that make one stripe (WaveQueue): 64 pixel slots x the batch's samples
    uint32_t *preq;     // primaries-first launches: the primary hits in queue order, those with bounce materials from the front, the others from the back (n0 entries); else nullptr
    uint32_t *pre_blocks;  // ... and the two counts / offsets of every PRE_SEG primaries
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
    int persistent_blocks;  // grid of the persistent path kernel
    bool count_visits;
    bool reference_traversal;
    bool leaf_bank = false;  // the path kernel variant that sets leaves aside during node steps (kernels.hip, wave_step)
    bool xform_cache = false;  // scenes where many objects have transforms: the kernel variant that keeps the ray in the last transform's space (kernels.hip, RayCtx)
    bool primaries_first = false;  // trace the primaries alone, then the rest from their hits, bounce materials first (launch_path)
};

void launch_raygen(const DeviceScene &sc, const Frame &fr, const BatchInfo &bi, const LaunchCfg &lc);
// Loads caller rays (origin, direction) as the batch's primaries, normalising as Ray::new does.
void launch_load_rays(const Frame &fr, const float *org_dir, uint32_t n, const LaunchCfg &lc);
// The trace path of the batch's primaries in ONE persistent launch: a lane takes a primary from the
// queue and follows World::trace_ray (world.rs:115-131) through its whole path tree, depth-first:
// closest hit, the shadow query of every light (light.rs:81-122), the bounce rays of
// material.rs:185-203 -- each of them a path node whose record the resolve pass reads.
// shade == false: closest hits of the primaries only (rtx_trace_primary / rtx_trace_rays).
// lc.primaries_first (shade, production traversal, no counters, fr.preq set, n0 primaries in the frame): the same
// in two persistent launches -- the primaries' closest hits first; then the hits are listed, those whose
// material spawns bounce rays first (three small launches: count, scan, scatter; stable, so the queue keeps its
// order within a class), and the second launch starts from that list: the long dependent chains of a frame begin
// when the launch begins, not when their primary's turn comes.
void launch_path(const DeviceScene &sc, const Frame &fr, uint32_t max_levels, bool shade, const LaunchCfg &lc, uint32_t n0 = 0);
// Resolves the colours of the path nodes of one bounce level (material.rs:118-206), deepest level
// first: a node's bounce children have their colours by then.
void launch_resolve(const DeviceScene &sc, const Frame &fr, uint32_t level, const LaunchCfg &lc);
// Adds the batch's sample colours to `sum` in sample order (world.rs:54-61); on the last
// sample chunk divides by spp, clamps and quantises (world.rs:62-63, color.rs:25-31).
void launch_accumulate(const DeviceScene &sc, const Frame &fr, const BatchInfo &bi, float4 *sum,
                       uint8_t *tiles_rgb8, float *tiles_f32_or_null, const LaunchCfg &lc);
// Hit export for rtx_trace_primary / rtx_trace_rays: primaries -> (object id, t).
void launch_export_hits(const DeviceScene &sc, const Frame &fr, uint32_t n, uint32_t *out_obj, float *out_t,
                        const LaunchCfg &lc);
void launch_export_colors(const Frame &fr, uint32_t n, float *out_rgb, const LaunchCfg &lc);
// The frame's exchange step behind the collective: the all-gathered tile buffers of every rank
// ([rank][local tile][row in tile][col in tile][rgb], rank_stride bytes apart) -> one row-major RGB8 image.
// Replaces the receive loop of world.rs:70-74 (pixels[x] = colors).
void launch_deinterleave(const uint8_t *all_tiles, size_t rank_stride, uint32_t nranks, uint32_t tile, uint32_t width,
                         uint32_t height, uint8_t *image, hipStream_t stream);
int query_persistent_blocks(int device);

// rtx_render_device_noalloc: the wavefront state of a small render lives in a file-scope
// `static __device__` array (kernels.hip) instead of hipMalloc'ed buffers.  Every GPU holds its
// own copy at its own address: noalloc_scratch_address looks it up (hipGetSymbolAddress) with
// `device` current, once per device, and caches it.
constexpr size_t NOALLOC_SCRATCH_BYTES = 64ull << 20;
hipError_t noalloc_scratch_address(int device, void **ptr);

}  // namespace rtx
