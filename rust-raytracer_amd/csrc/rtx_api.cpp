// C ABI of include/rtx.h: context, scene upload, wavefront render driver.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <limits.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtx.h"
#include "../../include/rtx_host.h"
#include "bvh_build.h"
#include "gpu_build.h"
#include "flatten.h"
#include "hostmath.h"
#include "kernels.h"

using namespace rtx;

namespace {
thread_local std::string g_init_error;  // errors with no context to hold them (rtx_last_error(NULL))
double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// One render slot: a stream, the wavefront buffers of the batch in flight on it, and what a
// render needs besides -- renders in different slots do not share anything they write, so
// their kernels overlap on the GPU (one slot's drain and resolve passes under the other's bulk).
struct Pipe {
    Frame fr{};
    std::vector<void *> allocs;
    uint64_t n0_cap = 0, node_cap = 0;
    uint64_t preq_cap = 0;            // primaries the list of a primaries-first launch holds (ensure_preq)
    uint32_t *preq = nullptr, *pre_blocks = nullptr;   // ... (kept here: ensure_frame starts its Frame afresh)
    uint32_t spill_entries = 0, wstack_cap = 0;
    hipStream_t main = nullptr;
    float4 *sum = nullptr;            // per-pixel sums of a render cut into several batches
    uint64_t pix_cap = 0;
    uint32_t *h_counters = nullptr;   // pinned, one CTR_WORDS block per batch
    uint32_t h_counter_slots = 0;
    std::vector<hipEvent_t> events;   // timing (renders with stats)
    uint32_t pending_batches = 0;     // batches of the last asynchronous render whose overflow flags nobody has looked at
    // the last render of this slot, for rtx_slot_stats
    uint32_t last_batches = 0, last_levels = 0;
    uint64_t last_primary = 0;
    double last_t0 = 0;               // host time the render was enqueued at (synchronous renders: after the stream had drained)
    // Second lane of a render that is cut into several batches: odd batches run on a stream and in
    // wavefront buffers of their own, so that batch k's tail and resolve launches hide under batch
    // k+1's path kernel; the accumulates stay in sample order through an event chain.  Created on
    // the first such render.
    std::unique_ptr<Pipe> lane2;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_acc[2] = {nullptr, nullptr};
};

struct rtx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;     // = pipes[0].main, the stream rtx_stream() hands out
    std::vector<Pipe> pipes;          // the render slots (rtx_render_device_slot); everything else uses slot 0
    std::string err;
    int persistent_blocks = 1024;
    // scene
    bool has_scene = false;
    DeviceScene ds{};
    std::vector<void *> scene_allocs;
    std::vector<uint32_t> leaf_order, preorder_sizes;
    rtx_bvh_info info{};
    uint32_t max_bounce = 0;
    uint32_t per_hit = 0;             // work entries one closest hit can push: bounce children + non-ambient lights
    // multi-GPU (rtx_comm_init): one process per GPU, this context's rank in the frame's tile partition
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    // host-image renders (rtx_render, slot 0)
    uint8_t *tiles = nullptr;
    float *tiles_f32 = nullptr;
    uint64_t tiles_cap = 0;
    uint64_t node_budget_bytes = 16ull << 30;  // wavefront state of ONE batch (a slot in use holds one, two with the second lane)
    bool blocks_fixed = false;        // RTX_PERSISTENT_BLOCKS given: every path launch uses exactly that grid
    int leaf_bank = -1;               // RTX_LEAF_BANK: -1 = by scene size
    int batch_lanes = 2;              // RTX_BATCH_LANES=1: the batches of a render all run on the slot's own stream
    uint32_t force_flags = 0;         // RTX_REFERENCE_TRAVERSAL=1: every call runs the verbatim traversal (A/B checks)
    int build_threads = 0;            // rtx_set_build_threads (0: min(16, hardware threads))
    uint32_t spill_entries = 1;       // traversal-stack entries beyond LDS_STACK that the uploaded tree can need
    uint8_t *scratch = nullptr;       // this device's copy of the static scratch array (rtx_render_device_noalloc)
    uint32_t min_batches = 1;         // RTX_MIN_BATCHES: cut big frames into at least this many batches
    uint32_t *h_noalloc_flag = nullptr;  // pinned word: CTR_OVERFLOW of the last rtx_render_device_noalloc (allocated by rtx_init)
    bool noalloc_pending = false;     // ... which nobody has looked at yet (rtx_synchronize / rtx_slot_synchronize(0) do)
    bool any_xform = false;           // many objects of the uploaded scene have a transform (flatten.cpp): LaunchCfg::xform_cache
    int xform_cache_force = -1;       // RTX_XFORM_CACHE=0/1: force the kernel variant without / with the transform cache (tests; results are identical)
    int gpu_binning = -1;             // RTX_GPU_BINNING: 0 never, 1 from 4096 objects on, -1 = from 2^20 objects on
    int primaries_first = -1;         // RTX_PRIMARIES_FIRST: 0 never, 1 always, -1 = by batch size
    uint32_t primaries_first_below = 0;  // RTX_PRIMARIES_FIRST_BELOW: ... batches of fewer primaries per wave than this
    uint32_t stripe_blocks = 1;       // RTX_STRIPE_BLOCKS: 8x8 pixel blocks (with all their samples) per stripe of the queue of primaries
    long test_pool_nodes = -1;        // RTX_TEST_POOL_NODES: tests shrink the pool of bounce children to force the overflow report
};

namespace {

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return RTX_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

template <class T>
int upload(rtx_ctx *c, const std::vector<T> &v, const T **out, std::vector<void *> &allocs) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 64);
    void *p = nullptr;
    HIP_TRY(c, hipMalloc(&p, bytes));
    allocs.push_back(p);
    if (!v.empty()) HIP_TRY(c, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T *>(p);
    c->info.device_bytes += bytes;
    return RTX_OK;
}

// Device scratch of one call, released on every return path.
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 256)); }
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

void free_all(std::vector<void *> &allocs) {
    for (void *p : allocs) (void)hipFree(p);
    allocs.clear();
}

struct Partition {
    uint32_t tile, rank, nranks, tiles_x, tiles_y, my_tiles;
    uint64_t npix;
};

int make_partition(rtx_ctx *c, const rtx_partition *p, Partition &out) {
    out.tile = p ? p->tile : 32;
    out.rank = p ? p->rank : 0;
    out.nranks = p ? p->nranks : 1;
    if (out.tile == 0 || (out.tile & 7u) || out.tile > 1024 || out.nranks == 0 || out.rank >= out.nranks) {
        c->err = "partition: tile must be a multiple of 8 (<= 1024) and rank < nranks";
        return RTX_ERR_INVALID;
    }
    out.tiles_x = (c->ds.width + out.tile - 1) / out.tile;
    out.tiles_y = (c->ds.height + out.tile - 1) / out.tile;
    uint64_t nt = (uint64_t)out.tiles_x * out.tiles_y;
    out.my_tiles = (uint32_t)((nt + out.nranks - 1 - out.rank) / out.nranks);
    out.npix = (uint64_t)out.my_tiles * out.tile * out.tile;
    return RTX_OK;
}

// host mirror of slot_to_pixel in kernels.hip
bool slot_to_pixel(const Partition &pt, uint32_t W, uint32_t H, uint64_t lp, uint32_t &px, uint32_t &py) {
    uint32_t T = pt.tile, tt = T * T;
    uint32_t lt = (uint32_t)(lp / tt), rr = (uint32_t)(lp % tt);
    uint32_t blk = rr >> 6, in = rr & 63u, bpr = T >> 3;
    uint32_t x_in = (blk % bpr) * 8u + (in & 7u), y_in = (blk / bpr) * 8u + (in >> 3);
    uint64_t gt = (uint64_t)lt * pt.nranks + pt.rank;
    if (gt >= (uint64_t)pt.tiles_x * pt.tiles_y) return false;
    px = (uint32_t)(gt % pt.tiles_x) * T + x_in;
    py = (uint32_t)(gt / pt.tiles_x) * T + y_in;
    return px < W && py < H;
}

uint64_t geo_sum(uint32_t K, uint32_t levels) {
    uint64_t s = 0, g = 1;
    for (uint32_t k = 0; k < levels; k++) {
        s += g;
        g *= K;
        if (K == 0) break;
    }
    return std::max<uint64_t>(s, 1);
}

// Path nodes that `n0` primary samples traced `levels` deep can need: the samples themselves plus a
// pool for their bounce children -- at most K per hit -- which the waves of the path kernel take in
// runs of POOL_RUN (the slack: every wave can strand one run).
uint64_t node_capacity(const rtx_ctx *c, uint64_t n0, uint32_t levels, size_t threads) {
    const uint64_t children = n0 * (geo_sum(c->max_bounce, levels) - 1);
    if (children && c->test_pool_nodes >= 0) return n0 + (uint64_t)c->test_pool_nodes;
    return n0 + (children ? children + (uint64_t)(threads / 64) * POOL_RUN : 0);
}

// Entries of a wave's work stack (path kernel): its depth-first walk keeps at most the untraced
// children and shadow queries of the <= 64 paths it has in flight, level by level; new primaries
// are admitted only while the stack is under half of this.
uint32_t wstack_capacity(const rtx_ctx *c, uint32_t levels) {
    return std::max<uint32_t>(1024, 64u * std::max(c->per_hit, 1u) * (levels + 2));
}

// Lays the wavefront buffers of one batch (`n0` primary samples, `node_cap` path nodes, `threads`
// persistent threads, `npix` pixel sums or 0) out in [base, ...) and returns the bytes used; with
// base == nullptr it only measures.  Used for the static scratch of rtx_render_device_noalloc.
size_t layout_frame(uint8_t *base, uint64_t n0, uint64_t node_cap, size_t threads, uint32_t spill_entries,
                    uint32_t wstack_cap, uint64_t npix, Frame &fr, float4 **sum) {
    size_t off = 0;
    auto take = [&](size_t bytes) -> uint8_t * {
        uint8_t *p = base ? base + off : nullptr;
        off += (std::max<size_t>(bytes, 256) + 255) & ~(size_t)255;
        return p;
    };
    fr.ray_o = (float4 *)take(node_cap * sizeof(float4));
    fr.ray_d = (float4 *)take(node_cap * sizeof(float4));
    fr.link = (uint2 *)take(node_cap * sizeof(uint2));
    fr.color = (float4 *)take(node_cap * sizeof(float4));
    fr.wstack = (uint2 *)take((threads / 64) * (wstack_cap + SHADE_LIST) * sizeof(uint2));
    // queue heads and counter block are one stretch: one memset clears both (reset_counters)
    fr.work = (uint32_t *)take((WORK_WORDS + CTR_WORDS) * sizeof(uint32_t));
    fr.counters = fr.work ? fr.work + WORK_WORDS : nullptr;
    fr.spill = (uint32_t *)take(threads * spill_entries * sizeof(uint32_t) * 2);
    if (sum) *sum = (float4 *)take(npix * sizeof(float4));
    fr.node_cap = (uint32_t)node_cap;
    fr.wstack_cap = wstack_cap;
    (void)n0;
    return off;
}

// Makes sure the pipe's wavefront buffers hold `n0` primary samples traced `levels` deep.
int ensure_frame(rtx_ctx *c, Pipe &pp, uint64_t n0, uint32_t levels) {
    const size_t threads = (size_t)c->persistent_blocks * 256;
    const uint64_t node_cap = node_capacity(c, n0, levels, threads);
    if (node_cap >= 0xFFFFFF00ull) {
        c->err = "batch exceeds 2^32 path nodes";
        return RTX_ERR_CAPACITY;
    }
    // (a frame laid out for more primaries keeps its pool: the pool begins at the batch's own n0)
    const uint32_t wcap = wstack_capacity(c, levels);
    if (pp.fr.ray_o && node_cap <= pp.node_cap && node_cap - n0 <= pp.node_cap - pp.n0_cap && c->spill_entries <= pp.spill_entries &&
        wcap <= pp.wstack_cap) {
        pp.fr.node_cap = (uint32_t)pp.node_cap;
        pp.fr.wstack_cap = pp.wstack_cap;
        return RTX_OK;
    }
    HIP_TRY(c, hipStreamSynchronize(pp.main));
    free_all(pp.allocs);
    pp.fr = Frame{};
    pp.node_cap = pp.n0_cap = 0;
    auto alloc = [&](void **p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 256));
        if (e == hipSuccess) pp.allocs.push_back(*p);
        return e;
    };
    HIP_TRY(c, alloc((void **)&pp.fr.ray_o, node_cap * sizeof(float4)));
    HIP_TRY(c, alloc((void **)&pp.fr.ray_d, node_cap * sizeof(float4)));
    HIP_TRY(c, alloc((void **)&pp.fr.link, node_cap * sizeof(uint2)));
    HIP_TRY(c, alloc((void **)&pp.fr.color, node_cap * sizeof(float4)));
    HIP_TRY(c, alloc((void **)&pp.fr.wstack, (threads / 64) * ((size_t)wcap + SHADE_LIST) * sizeof(uint2)));
    // queue heads, counter block (+ straggler list of diagnostic builds): one stretch, cleared by one memset
    HIP_TRY(c, alloc((void **)&pp.fr.work, (WORK_WORDS + CTR_WORDS + 256) * sizeof(uint32_t)));
    pp.fr.counters = pp.fr.work + WORK_WORDS;
    // stack overflow beyond the LDS entries, sized from the uploaded tree (not MAX_TREE_DEPTH)
    HIP_TRY(c, alloc((void **)&pp.fr.spill, threads * c->spill_entries * sizeof(uint32_t) * 2));  // (ref, t)
    pp.spill_entries = c->spill_entries;
    pp.wstack_cap = wcap;
    pp.fr.wstack_cap = wcap;
    pp.node_cap = node_cap;
    pp.n0_cap = n0;
    pp.fr.node_cap = (uint32_t)node_cap;
    return RTX_OK;
}

// The list of primary hits of a primaries-first launch (launch_path): 4 bytes per primary plus the block counts,
// outside the node budget, grown on demand.
int ensure_preq(rtx_ctx *c, Pipe &pp, uint64_t n0) {
    if (!(pp.preq && n0 <= pp.preq_cap)) {
        HIP_TRY(c, hipStreamSynchronize(pp.main));
        if (pp.preq) (void)hipFree(pp.preq);
        if (pp.pre_blocks) (void)hipFree(pp.pre_blocks);
        pp.preq = pp.pre_blocks = nullptr;
        pp.preq_cap = 0;
        HIP_TRY(c, hipMalloc((void **)&pp.preq, std::max<size_t>(n0 * sizeof(uint32_t), 256)));
        HIP_TRY(c, hipMalloc((void **)&pp.pre_blocks, std::max<size_t>(((n0 + PRE_SEG - 1) / PRE_SEG) * 2 * sizeof(uint32_t), 256)));
        pp.preq_cap = n0;
    }
    pp.fr.preq = pp.preq;
    pp.fr.pre_blocks = pp.pre_blocks;
    return RTX_OK;
}

int reset_counters(rtx_ctx *c, Pipe &pp) {
    static const bool util_print = std::getenv("RTX_UTIL_PRINT") != nullptr;  // diagnostic builds (-DRTX_UTIL_STATS)
    HIP_TRY(c, hipMemsetAsync(pp.fr.work, 0, (WORK_WORDS + CTR_WORDS) * sizeof(uint32_t), pp.main));
    static const bool hist_print = std::getenv("RTX_HIST_PRINT") != nullptr;
    if (hist_print && pp.allocs.size()) HIP_TRY(c, hipMemsetAsync(pp.fr.counters + CTR_WORDS, 0, 256 * sizeof(uint32_t), pp.main));
    if (util_print) {
        if (pp.allocs.size()) HIP_TRY(c, hipMemsetAsync(pp.fr.counters + CTR_WORDS, 0, 256 * sizeof(uint32_t), pp.main));
        HIP_TRY(c, hipMemsetAsync(pp.fr.counters + CTR_UTIL + 20, 0xFF, 8, pp.main));  // earliest wave begin: atomicMin
    }
    return RTX_OK;
}

int ensure_pixels(rtx_ctx *c, Pipe &pp, uint64_t npix, bool own_tiles, bool want_f32) {
    if (npix > pp.pix_cap || !pp.sum) {
        HIP_TRY(c, hipStreamSynchronize(pp.main));
        if (pp.sum) (void)hipFree(pp.sum);
        pp.sum = nullptr;
        HIP_TRY(c, hipMalloc((void **)&pp.sum, std::max<size_t>(npix * sizeof(float4), 256)));
        pp.pix_cap = npix;
    }
    if (own_tiles && (npix > c->tiles_cap || !c->tiles)) {
        if (c->tiles) (void)hipFree(c->tiles);
        if (c->tiles_f32) (void)hipFree(c->tiles_f32);
        c->tiles = nullptr;
        c->tiles_f32 = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->tiles, std::max<size_t>(npix * 3, 256)));
        c->tiles_cap = npix;
    }
    if (own_tiles && want_f32 && !c->tiles_f32)
        HIP_TRY(c, hipMalloc((void **)&c->tiles_f32, std::max<size_t>(c->tiles_cap * 3 * sizeof(float), 256)));
    return RTX_OK;
}

int ensure_counter_slots(rtx_ctx *c, Pipe &pp, uint32_t slots) {
    if (slots <= pp.h_counter_slots) return RTX_OK;
    HIP_TRY(c, hipStreamSynchronize(pp.main));  // an earlier render may still copy into the old block
    if (pp.h_counters) (void)hipHostFree(pp.h_counters);
    pp.h_counters = nullptr;
    HIP_TRY(c, hipHostMalloc((void **)&pp.h_counters, (size_t)slots * CTR_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    pp.h_counter_slots = slots;
    return RTX_OK;
}

int ensure_events(rtx_ctx *c, Pipe &pp, size_t n) {
    while (pp.events.size() < n) {
        hipEvent_t e;
        HIP_TRY(c, hipEventCreate(&e));
        pp.events.push_back(e);
    }
    return RTX_OK;
}

struct Batch {
    uint32_t q0, npix, s0, ns;
};

// Runs the trace path of one batch whose primary rays are already in the frame: the persistent
// path kernel (every ray of every bounce level), then the colours bottom-up, one streaming pass
// per level.  Records 3 timing events starting at events[ev0] when timing is on: path begin, path
// end, resolves end.
int run_path(rtx_ctx *c, Pipe &lane, std::vector<hipEvent_t> &events, LaunchCfg lc, uint32_t levels, bool timing, size_t ev0,
             uint64_t n0 = 0) {
    lc.stream = lane.main;
    // Primaries first (launch_path): RTX_PRIMARIES_FIRST=0/1 forces it off / on; by default batches of fewer than
    // primaries_first_below primaries per wave of the grid, whose launch is mostly its end phase
    lc.primaries_first = false;
    if (n0 != 0 && levels != 0 && !lc.count_visits && !lc.reference_traversal && !lc.leaf_bank && c->primaries_first != 0) {
        const uint64_t waves = (uint64_t)std::max(1, lc.persistent_blocks) * 4;
        if (c->primaries_first > 0 || n0 < (uint64_t)c->primaries_first_below * waves) {
            const int rc = ensure_preq(c, lane, n0);
            if (rc != RTX_OK) return rc;
            lc.primaries_first = true;
        }
    }
    if (timing) (void)hipEventRecord(events[ev0], lane.main);
    if (levels) launch_path(c->ds, lane.fr, levels, true, lc, (uint32_t)n0);   // trace_ray(.., 0) is the background (world.rs:116-117)
    if (timing) (void)hipEventRecord(events[ev0 + 1], lane.main);
    // a level deeper than 0 exists only if some material spawns bounce rays
    for (int k = (c->max_bounce ? (int)levels - 1 : 0); k >= 1; k--) launch_resolve(c->ds, lane.fr, (uint32_t)k, lc);
    launch_resolve(c->ds, lane.fr, 0, lc);
    if (timing) (void)hipEventRecord(events[ev0 + 2], lane.main);
    return RTX_OK;
}

constexpr size_t EV_PER_BATCH = 3;  // path begin, path end, resolves end
constexpr uint32_t RTX_LEAF_BANK_MIN_PRIMS = 8u << 20;  // scenes from this size on run the leaf-banking path kernel

// Waits for the slot's last render and reports it (overflow flags, counters, event times).
int slot_stats(rtx_ctx *c, Pipe &pp, rtx_stats *stats) {
    HIP_TRY(c, hipStreamSynchronize(pp.main));
    const double t1 = now_s();
    const uint32_t nb = std::min(pp.last_batches, pp.h_counter_slots);
    bool overflow = false;
    for (uint32_t bi = 0; bi < pp.pending_batches && bi < pp.h_counter_slots; bi++)
        if (pp.h_counters[(size_t)bi * CTR_WORDS + CTR_OVERFLOW]) overflow = true;
    pp.pending_batches = 0;
    if (&pp == &c->pipes[0] && c->noalloc_pending) {  // rtx_render_device_noalloc runs on slot 0's stream
        c->noalloc_pending = false;
        if (*c->h_noalloc_flag) overflow = true;
    }
    if (overflow) {
        c->err = "wavefront pool overflow (internal sizing error)";
        return RTX_ERR_CAPACITY;
    }
    if (!stats) return RTX_OK;
    std::memset(stats, 0, sizeof(*stats));
    stats->trace_ms = (t1 - pp.last_t0) * 1e3;
    stats->batches = nb;
    stats->primary_rays = pp.last_primary;
    for (size_t bi = 0; bi < nb; bi++) {
        const uint32_t *h = pp.h_counters + bi * CTR_WORDS;
        stats->bounce_rays += (uint64_t)h[CTR_BOUNCE_LO] | ((uint64_t)h[CTR_BOUNCE_HI] << 32);
        stats->shadow_rays += (uint64_t)h[CTR_SHADOW_LO] | ((uint64_t)h[CTR_SHADOW_HI] << 32);
        stats->closest_nodes += (uint64_t)h[CTR_VISIT_NODES_LO] | ((uint64_t)h[CTR_VISIT_NODES_HI] << 32);
        stats->closest_prims += (uint64_t)h[CTR_VISIT_PRIMS_LO] | ((uint64_t)h[CTR_VISIT_PRIMS_HI] << 32);
        stats->shadow_nodes += (uint64_t)h[CTR_SVISIT_NODES_LO] | ((uint64_t)h[CTR_SVISIT_NODES_HI] << 32);
        stats->shadow_prims += (uint64_t)h[CTR_SVISIT_PRIMS_LO] | ((uint64_t)h[CTR_SVISIT_PRIMS_HI] << 32);
        stats->max_ray_records = std::max(stats->max_ray_records, h[CTR_MAX_STEPS]);
        if (std::getenv("RTX_WAVE_TIMES") && pp.allocs.size() && bi + 1 == nb) {  // diagnostic builds (-DRTX_WAVE_TIMES)
            const size_t per_wave = (size_t)pp.wstack_cap + SHADE_LIST, waves = (size_t)c->persistent_blocks * 4;
            std::vector<uint2> w(waves * per_wave);
            if (hipMemcpy(w.data(), pp.fr.wstack, w.size() * sizeof(uint2), hipMemcpyDeviceToHost) == hipSuccess) {
                std::vector<double> dry, end;
                uint32_t t0 = 0;
                bool have = false;
                for (size_t k = 0; k < waves; k++) {
                    const uint2 a = w[k * per_wave + per_wave - 2], b = w[k * per_wave + per_wave - 1];
                    if (b.y != 0x57415645u) continue;
                    if (!have || (int32_t)(a.x - t0) < 0) t0 = a.x;
                    have = true;
                }
                for (size_t k = 0; k < waves; k++) {
                    const uint2 a = w[k * per_wave + per_wave - 2], b = w[k * per_wave + per_wave - 1];
                    if (b.y != 0x57415645u) continue;
                    end.push_back((double)(uint32_t)(b.x - t0) * 1e-5);
                    dry.push_back(a.y ? (double)(uint32_t)(a.y - t0) * 1e-5 : end.back());
                }
                std::sort(dry.begin(), dry.end());
                std::sort(end.begin(), end.end());
                auto q = [](const std::vector<double> &v, double f) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
                std::fprintf(stderr, "wave times (ms after the first wave began), %zu waves: queue dry at min %.3f / 10%% %.3f / median %.3f / 90%% %.3f / max %.3f; "
                                     "ended at min %.3f / 10%% %.3f / median %.3f / 90%% %.3f / 99%% %.3f / max %.3f\n",
                             end.size(), q(dry, 0), q(dry, 0.1), q(dry, 0.5), q(dry, 0.9), q(dry, 1.0), q(end, 0), q(end, 0.1), q(end, 0.5), q(end, 0.9),
                             q(end, 0.99), q(end, 1.0));
            }
        }
        if (std::getenv("RTX_HIST_PRINT") && pp.allocs.size()) {  // diagnostic builds (-DRTX_RAY_HIST), COUNT launches
            uint32_t hist[256];
            if (hipMemcpy(hist, pp.fr.counters + CTR_WORDS, sizeof hist, hipMemcpyDeviceToHost) == hipSuccess) {
                static const char *names[8] = {"primary", "bounce", "shadow (point/cone)", "shadow (directional)", "primary, ill axis",
                                               "bounce, ill axis", "shadow (point/cone), ill axis", "shadow (directional), ill axis"};
                for (int cl = 0; cl < 8; cl++) {
                    std::fprintf(stderr, "hist %-32s", names[cl]);
                    for (int b = 0; b < 16; b++)
                        if (hist[cl * 16 + b]) std::fprintf(stderr, " <%u: %u rays %u rec |", 1u << b, hist[cl * 16 + b], hist[128 + cl * 16 + b]);
                    std::fprintf(stderr, "\n");
                }
            }
        }
        if (std::getenv("RTX_UTIL_PRINT")) {  // diagnostic builds (-DRTX_UTIL_STATS) fill these words
            const uint64_t *u = reinterpret_cast<const uint64_t *>(&h[CTR_UTIL]);
            std::fprintf(stderr, "util path node_steps %llu prim_steps %llu node_lanes %llu prim_lanes %llu done_lanes %llu "
                                 "events %llu shade_events %llu shade_lanes %llu start_lanes %llu fin_lanes %llu\n",
                         (unsigned long long)u[0], (unsigned long long)u[1], (unsigned long long)u[2],
                         (unsigned long long)u[3], (unsigned long long)u[4], (unsigned long long)u[5],
                         (unsigned long long)u[6], (unsigned long long)u[7], (unsigned long long)u[8],
                         (unsigned long long)u[9]);
            std::fprintf(stderr, "util node steps that ran the division path: %llu (%llu lanes needed it)\n", (unsigned long long)u[20], (unsigned long long)u[21]);
            uint32_t dbg[256];
            if (pp.allocs.size() && hipMemcpy(dbg, pp.fr.counters + CTR_WORDS, sizeof dbg, hipMemcpyDeviceToHost) == hipSuccess) {
                std::fprintf(stderr, "util stragglers (> 0.6 ms after the queue ran dry): %u\n", dbg[0]);
                for (uint32_t k = 0; k < std::min(dbg[0], 24u); k++) {
                    const uint32_t *r = dbg + 1 + 10 * k;
                    std::fprintf(stderr, "   %.3f ms: %u events, %u wave steps, %u rays finished (%u closest), %u entries + rays held when dry; "
                                         "of it %.3f ms in steps, %.3f finished rays + stack/queue, %.3f shading pass, %.3f ray begin\n",
                                 r[0] * 1e-5, r[1], r[2], r[3], r[5], r[4], r[6] * 1e-5, r[7] * 1e-5, r[8] * 1e-5, r[9] * 1e-5);
                }
            }
            std::fprintf(stderr, "util waves: kernel %.3f ms, mean wave lifetime %.3f ms (%.3f ms of it in traversal steps), mean / max time after the queue ran dry %.3f / %.3f ms\n",
                         (double)(u[11] - u[10]) * 1e-5, (double)u[12] * 1e-5 / (c->persistent_blocks * 4.0),
                         (double)u[15] * 1e-5 / (c->persistent_blocks * 4.0),
                         (double)u[13] * 1e-5 / (c->persistent_blocks * 4.0), (double)u[14] * 1e-5);
            std::fprintf(stderr, "util events, mean per wave: finished rays %.3f ms, shading pass %.3f ms, stack / queue %.3f ms, ray begin %.3f ms\n",
                         (double)u[16] * 1e-5 / (c->persistent_blocks * 4.0), (double)u[17] * 1e-5 / (c->persistent_blocks * 4.0),
                         (double)u[18] * 1e-5 / (c->persistent_blocks * 4.0), (double)u[19] * 1e-5 / (c->persistent_blocks * 4.0));
        }
        float ms = 0;
        const size_t e = EV_PER_BATCH * bi;
        if (e + 2 < pp.events.size()) {
            if (pp.last_levels && hipEventElapsedTime(&ms, pp.events[e], pp.events[e + 1]) == hipSuccess) {
                stats->path_kernel_ms += ms;
                stats->path_launches++;
            }
            if (hipEventElapsedTime(&ms, pp.events[e + 1], pp.events[e + 2]) == hipSuccess) stats->resolve_ms += ms;
        }
    }
    return RTX_OK;
}

// Waits for every render slot (a slot's stream ends behind its second lane) and reports the first error.
int drain_slots(rtx_ctx *c) {
    int result = RTX_OK;
    for (auto &pp : c->pipes) {
        if (pp.lane2 && pp.lane2->main) (void)hipStreamSynchronize(pp.lane2->main);
        int rc = slot_stats(c, pp, nullptr);
        if (rc != RTX_OK && result == RTX_OK) result = rc;
    }
    return result;
}

int render_impl(rtx_ctx *c, uint32_t slot, uint32_t spp, uint32_t bounces, uint64_t seed, const rtx_partition *part,
                uint32_t flags, void *d_tiles_out, bool want_f32, rtx_stats *stats, Partition &pt) {
    if (!c->has_scene) {
        c->err = "no scene uploaded";
        return RTX_ERR_NO_SCENE;
    }
    if (slot >= c->pipes.size()) {
        c->err = "render slot out of range (rtx_slot_count)";
        return RTX_ERR_INVALID;
    }
    flags |= c->force_flags;
    if (spp == 0) {
        c->err = "samples_per_pixel must be non-zero (world.rs:30)";
        return RTX_ERR_INVALID;
    }
    if (bounces > 8) {
        c->err = "max_ray_bounces > 8 not supported";
        return RTX_ERR_CAPACITY;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = make_partition(c, part, pt);
    if (rc != RTX_OK) return rc;
    if (pt.npix >= 0xFFFFFFFFull) {
        c->err = "more than 2^32 pixel slots per rank";
        return RTX_ERR_CAPACITY;
    }
    Pipe &pp = c->pipes[slot];
    uint32_t levels = bounces;
    uint32_t K = c->max_bounce;
    uint64_t per_sample = geo_sum(K, std::max(levels, 1u)) * 60;
    // Batches: a frame whose path nodes exceed the slot's share of the node budget is cut into
    // pieces (by sample, else by pixel range) that run one after the other on the slot's stream.
    uint64_t total = pt.npix * spp;
    uint64_t cap0 = std::max<uint64_t>(c->node_budget_bytes / per_sample, 4096);
    const uint32_t want = total >= (1u << 20) ? std::max(1u, c->min_batches) : 1u;
    std::vector<Batch> batches;
    if (pt.npix <= cap0 && spp >= want) {
        uint32_t nsc = (uint32_t)std::min<uint64_t>(cap0 / std::max<uint64_t>(pt.npix, 1), (spp + want - 1) / want);
        nsc = std::max(nsc, 1u);
        for (uint32_t s0 = 0; s0 < spp; s0 += nsc) batches.push_back(Batch{0, (uint32_t)pt.npix, s0, std::min(nsc, spp - s0)});
    } else {
        uint64_t parts = std::max<uint64_t>((pt.npix + cap0 - 1) / cap0, (want + spp - 1) / spp);
        uint64_t chunk = (((pt.npix + parts - 1) / parts) + 63) & ~63ull;
        chunk = std::max<uint64_t>(std::min<uint64_t>(chunk, cap0 & ~63ull), 64);
        for (uint64_t q0 = 0; q0 < pt.npix; q0 += chunk)
            for (uint32_t s0 = 0; s0 < spp; s0++)
                batches.push_back(Batch{(uint32_t)q0, (uint32_t)std::min<uint64_t>(chunk, pt.npix - q0), s0, 1});
    }
    if (batches.empty()) batches.push_back(Batch{0, 0, 0, spp});
    uint64_t n0max = 0;
    for (auto &b : batches) n0max = std::max<uint64_t>(n0max, (uint64_t)b.npix * b.ns);
    rc = ensure_frame(c, pp, std::max<uint64_t>(n0max, 1), std::max(levels, 1u));
    if (rc != RTX_OK) return rc;
    const bool two_lanes = batches.size() >= 2 && c->batch_lanes >= 2;
    if (two_lanes) {
        if (!pp.lane2) {
            pp.lane2.reset(new Pipe());
            HIP_TRY(c, hipStreamCreateWithFlags(&pp.lane2->main, hipStreamNonBlocking));
            for (hipEvent_t *e : {&pp.ev_begin, &pp.ev_end, &pp.ev_acc[0], &pp.ev_acc[1]})
                HIP_TRY(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        rc = ensure_frame(c, *pp.lane2, std::max<uint64_t>(n0max, 1), std::max(levels, 1u));
        if (rc != RTX_OK) return rc;
    }
    rc = ensure_pixels(c, pp, pt.npix, d_tiles_out == nullptr, want_f32);
    if (rc != RTX_OK) return rc;
    rc = ensure_counter_slots(c, pp, (uint32_t)batches.size());
    if (rc != RTX_OK) return rc;
    const bool timing = true;  // three event records per batch cost nothing; rtx_slot_stats reads them later
    const size_t ev_per_batch = EV_PER_BATCH;
    rc = ensure_events(c, pp, ev_per_batch * batches.size());
    if (rc != RTX_OK) return rc;
    LaunchCfg lc{pp.main, c->persistent_blocks, (flags & RTX_FLAG_COUNT_VISITS) != 0,
                 (flags & RTX_FLAG_REFERENCE_TRAVERSAL) != 0};
    lc.xform_cache = c->any_xform;
    // Leaf banking pays where rays are long and the kernel is bound by VALU issue: measured -9 % on the
    // 20M-triangle soup, +-0 on the 4M one (bound by fabric reads), +3..6 % on the scenes of a few
    // million primitives and less.  RTX_LEAF_BANK=0/1 overrides the size rule.
    lc.leaf_bank = c->leaf_bank < 0 ? c->ds.n_prims >= RTX_LEAF_BANK_MIN_PRIMS : c->leaf_bank != 0;
    uint8_t *tiles = d_tiles_out ? static_cast<uint8_t *>(d_tiles_out) : c->tiles;
    const bool async = !stats && d_tiles_out;
    if (!async) HIP_TRY(c, hipStreamSynchronize(pp.main));
    pp.last_t0 = now_s();
    if (two_lanes) {  // the second lane begins behind everything the slot's stream holds (the previous frame, its gather)
        HIP_TRY(c, hipEventRecord(pp.ev_begin, pp.main));
        HIP_TRY(c, hipStreamWaitEvent(pp.lane2->main, pp.ev_begin, 0));
    }
    // (a lambda so that every error return inside joins the second lane below)
    auto issue = [&]() -> int {
        for (size_t bi = 0; bi < batches.size(); bi++) {
            const Batch &b = batches[bi];
            Pipe &lane = (two_lanes && (bi & 1)) ? *pp.lane2 : pp;
            lc.stream = lane.main;
            BatchInfo info{};
            info.q0 = b.q0;
            info.npix = b.npix;
            info.s0 = b.s0;
            info.ns = b.ns;
            info.spp = spp;
            info.use_jitter = spp != 1;  // world.rs:47-51
            info.seed_lo = (uint32_t)seed;
            info.seed_hi = (uint32_t)(seed >> 32);
            info.tile = pt.tile;
            info.rank = pt.rank;
            info.nranks = pt.nranks;
            info.tiles_x = pt.tiles_x;
            info.tiles_y = pt.tiles_y;
            info.max_levels = levels;
            rc = reset_counters(c, lane);
            if (rc != RTX_OK) return rc;
            // A wave ends with the subtrees of its last 64 primaries at ever fewer active lanes: a small batch on
            // the full grid is mostly that phase.  Small batches get a smaller grid (>= 1024 primaries per block;
            // measured on 1/16 of the headline frame: 1.49 -> 1.40 ms alone, 0.69 -> 0.61 ms with four slots in rotation).
            lc.persistent_blocks = c->blocks_fixed ? c->persistent_blocks
                                                   : (int)std::min<uint64_t>((uint64_t)c->persistent_blocks,
                                                                             std::max<uint64_t>(256, (uint64_t)b.npix * b.ns / 1024));
            lane.fr.stripe = 64u * c->stripe_blocks * std::min(b.ns, 4096u);
            launch_raygen(c->ds, lane.fr, info, lc);
            rc = run_path(c, lane, pp.events, lc, levels, timing, ev_per_batch * bi, (uint64_t)b.npix * b.ns);
            if (rc != RTX_OK) return rc;
            // sums are folded in sample order (world.rs:54-61): the batches follow each other on the stream,
            // or, with two lanes, batch k's accumulate waits for batch k-1's on the other stream
            if (two_lanes && bi > 0) HIP_TRY(c, hipStreamWaitEvent(lane.main, pp.ev_acc[(bi - 1) & 1], 0));
            launch_accumulate(c->ds, lane.fr, info, pp.sum, tiles, want_f32 ? c->tiles_f32 : nullptr, lc);
            if (two_lanes) HIP_TRY(c, hipEventRecord(pp.ev_acc[bi & 1], lane.main));
            HIP_TRY(c, hipMemcpyAsync(pp.h_counters + bi * CTR_WORDS, lane.fr.counters, CTR_WORDS * sizeof(uint32_t),
                                      hipMemcpyDeviceToHost, lane.main));
        }
        if (two_lanes) {  // the slot's stream ends behind the second lane: whoever waits for the slot waits for the frame
            HIP_TRY(c, hipEventRecord(pp.ev_end, pp.lane2->main));
            HIP_TRY(c, hipStreamWaitEvent(pp.main, pp.ev_end, 0));
        }
        return RTX_OK;
    };
    rc = issue();
    if (rc != RTX_OK) {
        // the second lane may still use the slot's sums and pinned counters: nothing of the slot is
        // freed or reused before it has drained
        if (two_lanes) (void)hipStreamSynchronize(pp.lane2->main);
        (void)hipStreamSynchronize(pp.main);
        return rc;
    }
    HIP_TRY(c, hipGetLastError());
    // valid primary samples of this rank
    uint64_t valid = 0;
    for (uint32_t lt = 0; lt < pt.my_tiles; lt++) {
        uint64_t gt = (uint64_t)lt * pt.nranks + pt.rank;
        uint32_t tx = (uint32_t)(gt % pt.tiles_x), ty = (uint32_t)(gt / pt.tiles_x);
        uint32_t w = std::min(pt.tile, c->ds.width - std::min(c->ds.width, tx * pt.tile));
        uint32_t h = std::min(pt.tile, c->ds.height - std::min(c->ds.height, ty * pt.tile));
        valid += (uint64_t)w * h;
    }
    pp.last_batches = (uint32_t)batches.size();
    pp.last_levels = levels;
    pp.last_primary = levels ? valid * spp : 0;
    pp.pending_batches = (uint32_t)batches.size();
    if (async) return RTX_OK;  // rtx_synchronize / rtx_slot_stats look at the overflow flags
    return slot_stats(c, pp, stats);
}

}  // namespace

extern "C" {

int rtx_init(int device, rtx_ctx **out) {
    if (!out) return RTX_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_init_error = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                       " (this backend has no CPU fallback)";
        return RTX_ERR_HIP;
    }
    if (device < 0 || device >= n) {
        g_init_error = "device index out of range";
        return RTX_ERR_INVALID;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_init_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return RTX_ERR_HIP;
    }
    rtx_ctx *c = new rtx_ctx();
    c->device = device;
    int n_pipes = 4;  // buffers of a slot are allocated by its first render
    if (const char *np = std::getenv("RTX_PIPES")) n_pipes = std::min(4, std::max(1, std::atoi(np)));
    c->pipes.resize((size_t)n_pipes);
    e = hipSuccess;
    for (auto &pp : c->pipes)
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&pp.main, hipStreamNonBlocking);
    if (e != hipSuccess) {
        g_init_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        rtx_destroy(c);
        return RTX_ERR_HIP;
    }
    c->stream = c->pipes[0].main;
    c->persistent_blocks = query_persistent_blocks(device);
    if (std::getenv("RTX_VERBOSE")) std::fprintf(stderr, "rtx_init: device %d, persistent grid %d blocks of 256 threads\n", device, c->persistent_blocks);
    if (const char *pb = std::getenv("RTX_PERSISTENT_BLOCKS"))  // tests: grids whose wave count is not a multiple of QPARTS
        if (std::atoi(pb) > 0) {
            c->persistent_blocks = std::atoi(pb);
            c->blocks_fixed = true;
        }
    // this device's copy of the static scratch array of rtx_render_device_noalloc, looked up once, here
    e = noalloc_scratch_address(device, (void **)&c->scratch);
    if (e != hipSuccess || !c->scratch) {
        g_init_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
        rtx_destroy(c);
        return RTX_ERR_HIP;
    }
    e = hipHostMalloc((void **)&c->h_noalloc_flag, sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        g_init_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        rtx_destroy(c);
        return RTX_ERR_HIP;
    }
    *c->h_noalloc_flag = 0;
    if (const char *xc = std::getenv("RTX_XFORM_CACHE")) c->xform_cache_force = std::atoi(xc) != 0;
    if (const char *gb = std::getenv("RTX_GPU_BINNING")) c->gpu_binning = std::atoi(gb) != 0;
    if (const char *pf = std::getenv("RTX_PRIMARIES_FIRST")) c->primaries_first = std::atoi(pf) != 0;
    if (const char *pb = std::getenv("RTX_PRIMARIES_FIRST_BELOW")) c->primaries_first_below = (uint32_t)std::max(0, std::atoi(pb));
    if (const char *sb = std::getenv("RTX_STRIPE_BLOCKS")) c->stripe_blocks = (uint32_t)std::min(1024, std::max(1, std::atoi(sb)));
    if (const char *tp = std::getenv("RTX_TEST_POOL_NODES")) c->test_pool_nodes = std::max(0L, std::atol(tp));
    if (const char *lb = std::getenv("RTX_LEAF_BANK")) c->leaf_bank = std::atoi(lb) != 0;
    if (const char *bl = std::getenv("RTX_BATCH_LANES")) c->batch_lanes = std::atoi(bl);
    if (const char *rv = std::getenv("RTX_REFERENCE_TRAVERSAL"))
        if (std::atoi(rv)) c->force_flags |= RTX_FLAG_REFERENCE_TRAVERSAL;
    if (const char *mn = std::getenv("RTX_MIN_BATCHES")) c->min_batches = (uint32_t)std::min(64, std::max(1, std::atoi(mn)));
    if (const char *mb = std::getenv("RTX_NODE_BUDGET_MB")) {
        long v = std::atol(mb);
        if (v > 0) c->node_budget_bytes = (uint64_t)v << 20;
    }
    *out = c;
    return RTX_OK;
}

void rtx_destroy(rtx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto &pp : c->pipes)
        if (pp.main) (void)hipStreamSynchronize(pp.main);
    free_all(c->scene_allocs);
    for (auto &pp : c->pipes) {
        if (pp.lane2) {
            if (pp.lane2->main) (void)hipStreamSynchronize(pp.lane2->main);
            free_all(pp.lane2->allocs);
            if (pp.lane2->preq) (void)hipFree(pp.lane2->preq);
            if (pp.lane2->pre_blocks) (void)hipFree(pp.lane2->pre_blocks);
            if (pp.lane2->main) (void)hipStreamDestroy(pp.lane2->main);
        }
        for (hipEvent_t e : {pp.ev_begin, pp.ev_end, pp.ev_acc[0], pp.ev_acc[1]})
            if (e) (void)hipEventDestroy(e);
        free_all(pp.allocs);
        if (pp.preq) (void)hipFree(pp.preq);
        if (pp.pre_blocks) (void)hipFree(pp.pre_blocks);
        if (pp.sum) (void)hipFree(pp.sum);
        if (pp.h_counters) (void)hipHostFree(pp.h_counters);
        for (auto e : pp.events) (void)hipEventDestroy(e);
        if (pp.main) (void)hipStreamDestroy(pp.main);
    }
    if (c->tiles) (void)hipFree(c->tiles);
    if (c->tiles_f32) (void)hipFree(c->tiles_f32);
    if (c->h_noalloc_flag) (void)hipHostFree(c->h_noalloc_flag);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}

const char *rtx_last_error(rtx_ctx *c) { return c ? c->err.c_str() : g_init_error.c_str(); }

int rtx_upload_scene(rtx_ctx *c, const rtx_scene_desc *d) {
    if (!c || !d) return RTX_ERR_INVALID;
    if (d->width == 0 || d->height == 0) {
        c->err = "camera width/height must be non-zero";
        return RTX_ERR_INVALID;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    // every slot (and its second lane, which the slot's stream ends behind) may still be reading the old
    // scene; a render's unread overflow flag is reported here rather than carried over to the new scene
    int pending_rc = drain_slots(c);
    if (pending_rc != RTX_OK) return pending_rc;
    free_all(c->scene_allocs);
    c->has_scene = false;
    c->info = rtx_bvh_info{};
    int threads = (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (c->build_threads > 0) threads = c->build_threads;
    if (const char *t = std::getenv("RTX_BUILD_THREADS")) threads = std::max(1, std::atoi(t));
    double t0 = now_s();
    BuiltBvh bvh;
    // GPU-assisted binning (gpu_build.hip): the nodes at the top of a big tree are split by kernels, the pieces by
    // the host threads.  RTX_GPU_BINNING=0 / 1: never / for every scene of at least 4096 objects; default: from
    // 2^20 objects on (below that the host's own top levels cost less than the transfer of the boxes)
    int rc;
    {
        const int mode = c->gpu_binning;
        std::unique_ptr<TopSplitter> top(mode != 0 ? new_gpu_top_splitter(c->device) : nullptr);
        rc = build_bvh(*d, threads, bvh, c->err, top.get(), mode > 0 ? 4096u : 1u << 20);
    }
    if (rc != RTX_OK) return rc;
    double t1 = now_s();
    FlatScene fs;
    rc = flatten_scene(*d, bvh, fs, c->err, threads);
    if (rc != RTX_OK) return rc;
    // stack need: one entry per level (reference traversal), up to three per wide level
    if (bvh.depth + 2 > (uint32_t)MAX_TREE_DEPTH || 3 * fs.wide_depth + 2 > (uint32_t)MAX_TREE_DEPTH) {
        c->err = "BVH deeper than the traversal stack";
        return RTX_ERR_CAPACITY;
    }
    {
        const uint32_t need = std::max(bvh.depth, 3 * fs.wide_depth) + 2;
        c->spill_entries = need > (uint32_t)LDS_STACK ? need - (uint32_t)LDS_STACK : 1u;
    }
    double t2 = now_s();
    DeviceScene &ds = c->ds;
    ds = DeviceScene{};
    if ((rc = upload(c, fs.nodes, &ds.nodes, c->scene_allocs))) return rc;
    if ((rc = upload(c, fs.wnodes, &ds.wnodes, c->scene_allocs))) return rc;
    if ((rc = upload(c, fs.prims, &ds.prims, c->scene_allocs))) return rc;
    if ((rc = upload(c, fs.xforms, &ds.xforms, c->scene_allocs))) return rc;
    if ((rc = upload(c, fs.mats, &ds.mats, c->scene_allocs))) return rc;
    if ((rc = upload(c, fs.prog, &ds.prog, c->scene_allocs))) return rc;
    if ((rc = upload(c, fs.tex, &ds.tex, c->scene_allocs))) return rc;
    if ((rc = upload(c, fs.tex_data, &ds.tex_data, c->scene_allocs))) return rc;
    if ((rc = upload(c, fs.lights, &ds.lights, c->scene_allocs))) return rc;
    if ((rc = upload(c, fs.prim_obj, &ds.prim_obj, c->scene_allocs))) return rc;
    ds.n_prims = d->n_objects;
    ds.n_lights = (uint32_t)fs.lights.size();
    for (size_t l = 0; l < (size_t)MAX_LIGHTS; l++) ds.lights_k[l] = l < fs.lights.size() ? fs.lights[l] : LightRec{};
    ds.n_wnodes = (uint32_t)fs.wnodes.size();
    ds.root_ref = fs.root_ref;
    ds.wroot_ref = fs.wide_root_ref;
    ds.xf0_identity = fs.xf0_identity ? 1u : 0u;
    c->any_xform = c->xform_cache_force < 0 ? fs.any_xform : c->xform_cache_force != 0;
    float scale = 0.0f;
    for (int k = 0; k < 3; k++) {
        ds.root_mn[k] = fs.root_mn[k];
        ds.root_mx[k] = fs.root_mx[k];
        scale = std::max(scale, std::max(std::fabs(fs.root_mn[k]), std::fabs(fs.root_mx[k])));
        float bgc = d->background[k];
        ds.background[k] = bgc < 0.0f ? 0.0f : (bgc > 1.0f ? 1.0f : bgc);
    }
    // (non-finite boxes: an infinite scale makes every ray exact_only and turns pruning off, init_ray)
    ds.coord_scale = fs.nonfinite_boxes ? INFINITY : scale;
    ds.overhang = fs.overhang;
    std::memcpy(ds.c2w, d->camera_to_world, sizeof(ds.c2w));
    ds.width = d->width;
    ds.height = d->height;
    double t3 = now_s();
    c->leaf_order = std::move(bvh.leaf_order);
    c->preorder_sizes = std::move(fs.preorder_sizes);
    c->max_bounce = fs.max_bounce;
    ds.max_bounce = fs.max_bounce;
    ds.bounce_mat_lo = fs.bounce_mat_lo;
    ds.bounce_mat_hi = fs.bounce_mat_hi;
    c->per_hit = fs.max_bounce;
    for (const auto &l : fs.lights) c->per_hit += l.type != LIGHT_AMBIENT ? 1u : 0u;
    c->info.n_objects = d->n_objects;
    c->info.n_internal = (uint32_t)fs.nodes.size();
    c->info.n_leaves = fs.n_leaves;
    c->info.depth = bvh.depth;
    c->info.total_sa = fs.total_sa;
    c->info.build_seconds = t1 - t0;
    c->info.flatten_seconds = t2 - t1;
    c->info.upload_seconds = t3 - t2;
    c->has_scene = true;
    return RTX_OK;
}

int rtx_set_build_threads(rtx_ctx *c, int threads) {
    if (!c || threads < 0 || threads > 1024) return RTX_ERR_INVALID;
    c->build_threads = threads;
    return RTX_OK;
}

int rtx_bvh_info_get(rtx_ctx *c, rtx_bvh_info *out) {
    if (!c || !out) return RTX_ERR_INVALID;
    if (!c->has_scene) {
        c->err = "no scene uploaded";
        return RTX_ERR_NO_SCENE;
    }
    *out = c->info;
    return RTX_OK;
}

int rtx_bvh_digest(rtx_ctx *c, uint32_t *leaf_order, uint32_t *preorder_sizes) {
    if (!c) return RTX_ERR_INVALID;
    if (!c->has_scene) {
        c->err = "no scene uploaded";
        return RTX_ERR_NO_SCENE;
    }
    if (leaf_order) std::copy(c->leaf_order.begin(), c->leaf_order.end(), leaf_order);
    if (preorder_sizes) std::copy(c->preorder_sizes.begin(), c->preorder_sizes.end(), preorder_sizes);
    return (int)c->preorder_sizes.size();
}

uint64_t rtx_partition_pixels(rtx_ctx *c, const rtx_partition *part) {
    if (!c || !c->has_scene) return 0;
    Partition pt;
    if (make_partition(c, part, pt) != RTX_OK) return 0;
    return pt.npix;
}

void *rtx_stream(rtx_ctx *c) { return c ? (void *)c->stream : nullptr; }

static int sync_slot(rtx_ctx *c, Pipe &pp) { return slot_stats(c, pp, nullptr); }

int rtx_synchronize(rtx_ctx *c) {
    if (!c) return RTX_ERR_INVALID;
    int result = RTX_OK;
    for (auto &pp : c->pipes) {
        int rc = sync_slot(c, pp);
        if (rc != RTX_OK) result = rc;
    }
    return result;
}

int rtx_slot_count(rtx_ctx *c) { return c ? (int)c->pipes.size() : RTX_ERR_INVALID; }

void *rtx_slot_stream(rtx_ctx *c, uint32_t slot) { return c && slot < c->pipes.size() ? (void *)c->pipes[slot].main : nullptr; }

int rtx_slot_synchronize(rtx_ctx *c, uint32_t slot) {
    if (!c || slot >= c->pipes.size()) return RTX_ERR_INVALID;
    return sync_slot(c, c->pipes[slot]);
}

int rtx_slot_stats(rtx_ctx *c, uint32_t slot, rtx_stats *out) {
    if (!c || !out || slot >= c->pipes.size()) return RTX_ERR_INVALID;
    return slot_stats(c, c->pipes[slot], out);
}

int rtx_render_device_slot(rtx_ctx *c, uint32_t slot, uint32_t spp, uint32_t bounces, uint64_t seed, const rtx_partition *part,
                           uint32_t flags, void *d_tiles, rtx_stats *stats) {
    if (!c || !d_tiles) return RTX_ERR_INVALID;
    Partition pt;
    return render_impl(c, slot, spp, bounces, seed, part, flags, d_tiles, false, stats, pt);
}

int rtx_render_device(rtx_ctx *c, uint32_t spp, uint32_t bounces, uint64_t seed, const rtx_partition *part,
                      uint32_t flags, void *d_tiles, rtx_stats *stats) {
    if (!c || !d_tiles) return RTX_ERR_INVALID;
    Partition pt;
    return render_impl(c, 0, spp, bounces, seed, part, flags, d_tiles, false, stats, pt);
}

int rtx_render_device_noalloc(rtx_ctx *c, uint32_t spp, uint32_t bounces, uint64_t seed, const rtx_partition *part,
                              uint32_t flags, void *d_tiles) {
    if (!c || !d_tiles) return RTX_ERR_INVALID;
    if (!c->has_scene) {
        c->err = "no scene uploaded";
        return RTX_ERR_NO_SCENE;
    }
    if (spp == 0) {
        c->err = "samples_per_pixel must be non-zero (world.rs:30)";
        return RTX_ERR_INVALID;
    }
    if (bounces > 8) {
        c->err = "max_ray_bounces > 8 not supported";
        return RTX_ERR_CAPACITY;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    Partition pt;
    int rc = make_partition(c, part, pt);
    if (rc != RTX_OK) return rc;
    // ONE batch whose wavefront buffers are carved out of the static scratch array; a grid of at
    // most one block per CU keeps the stack-spill part of it small
    const uint32_t levels = bounces, lv = std::max(levels, 1u), K = c->max_bounce;
    const int blocks = std::min(c->persistent_blocks, 256);
    const uint64_t n0 = std::max<uint64_t>(pt.npix * spp, 1);
    const long double est = (long double)n0 * (long double)geo_sum(K, lv);
    Pipe np;  // no stream or events of its own: runs on rtx_stream()
    float4 *sum = nullptr;
    size_t need = ~(size_t)0;
    if (est < 4.0e9L)
        need = layout_frame(c->scratch, n0, node_capacity(c, n0, lv, (size_t)blocks * 256), (size_t)blocks * 256,
                            c->spill_entries, wstack_capacity(c, lv), pt.npix, np.fr, &sum);
    if (need > NOALLOC_SCRATCH_BYTES) {
        char msg[256];
        std::snprintf(msg, sizeof msg,
                      "rtx_render_device_noalloc: %llu pixel slots x %u spp, %u bounces need %.1f MiB of wavefront state; the "
                      "static scratch holds %.0f MiB (use rtx_render_device)",
                      (unsigned long long)pt.npix, spp, bounces, est < 4.0e9L ? need / 1048576.0 : 1e9,
                      NOALLOC_SCRATCH_BYTES / 1048576.0);
        c->err = msg;
        return RTX_ERR_CAPACITY;
    }
    np.main = c->stream;
    flags |= c->force_flags;
    LaunchCfg lc{c->stream, blocks, false, (flags & RTX_FLAG_REFERENCE_TRAVERSAL) != 0};
    lc.xform_cache = c->any_xform;
    BatchInfo info{};
    info.q0 = 0;
    info.npix = (uint32_t)pt.npix;
    info.s0 = 0;
    info.ns = spp;
    info.spp = spp;
    info.use_jitter = spp != 1;  // world.rs:47-51
    info.seed_lo = (uint32_t)seed;
    info.seed_hi = (uint32_t)(seed >> 32);
    info.tile = pt.tile;
    info.rank = pt.rank;
    info.nranks = pt.nranks;
    info.tiles_x = pt.tiles_x;
    info.tiles_y = pt.tiles_y;
    info.max_levels = levels;
    // the array keeps the previous call's contents: counters and queue heads must start at zero
    rc = reset_counters(c, np);
    if (rc != RTX_OK) return rc;
    np.fr.stripe = 64u * c->stripe_blocks * std::min(spp, 4096u);
    launch_raygen(c->ds, np.fr, info, lc);
    rc = run_path(c, np, np.events, lc, levels, false, 0);
    if (rc != RTX_OK) return rc;
    launch_accumulate(c->ds, np.fr, info, sum, static_cast<uint8_t *>(d_tiles), nullptr, lc);
    // the overflow flag goes to the pinned word of rtx_init (no allocation here); rtx_synchronize /
    // rtx_slot_synchronize(0) report it as RTX_ERR_CAPACITY
    HIP_TRY(c, hipMemcpyAsync(c->h_noalloc_flag, np.fr.counters + CTR_OVERFLOW, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    c->noalloc_pending = true;
    HIP_TRY(c, hipGetLastError());
    return RTX_OK;
}

int rtx_render(rtx_ctx *c, uint32_t spp, uint32_t bounces, uint64_t seed, const rtx_partition *part, uint32_t flags,
               uint8_t *out_rgb8, float *out_f32, rtx_stats *stats) {
    if (!c || !out_rgb8) return RTX_ERR_INVALID;
    Partition pt;
    rtx_stats local;
    int rc = render_impl(c, 0, spp, bounces, seed, part, flags, nullptr, out_f32 != nullptr, stats ? stats : &local, pt);
    if (rc != RTX_OK) return rc;
    std::vector<uint8_t> h8(pt.npix * 3);
    HIP_TRY(c, hipMemcpy(h8.data(), c->tiles, h8.size(), hipMemcpyDeviceToHost));
    std::vector<float> hf;
    if (out_f32) {
        hf.resize(pt.npix * 3);
        HIP_TRY(c, hipMemcpy(hf.data(), c->tiles_f32, hf.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    // tiles [local tile][row][col] -> row-major image
    uint32_t T = pt.tile, W = c->ds.width, H = c->ds.height;
    for (uint32_t lt = 0; lt < pt.my_tiles; lt++) {
        uint64_t gt = (uint64_t)lt * pt.nranks + pt.rank;
        uint32_t tx = (uint32_t)(gt % pt.tiles_x), ty = (uint32_t)(gt / pt.tiles_x);
        for (uint32_t y = 0; y < T; y++) {
            uint32_t py = ty * T + y;
            if (py >= H) break;
            uint32_t x0 = tx * T;
            if (x0 >= W) break;
            uint32_t w = std::min(T, W - x0);
            size_t src = ((size_t)lt * T * T + (size_t)y * T) * 3;
            size_t dst = ((size_t)py * W + x0) * 3;
            std::memcpy(out_rgb8 + dst, h8.data() + src, (size_t)w * 3);
            if (out_f32) std::memcpy(out_f32 + dst, hf.data() + src, (size_t)w * 3 * sizeof(float));
        }
    }
    return RTX_OK;
}

#define NCCL_TRY(ctx, expr)                                                                    \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            (ctx)->err = std::string(#expr) + ": " + ncclGetErrorString(r_);                   \
            return RTX_ERR_COMM;                                                               \
        }                                                                                      \
    } while (0)

int rtx_comm_unique_id(void *id_128_bytes) {
    if (!id_128_bytes) return RTX_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == RTX_COMM_ID_BYTES, "rtx.h states the size of the RCCL id");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) {
        g_init_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r);
        return RTX_ERR_COMM;
    }
    std::memcpy(id_128_bytes, &id, sizeof id);
    return RTX_OK;
}

int rtx_comm_init(rtx_ctx *c, const void *id_128_bytes, int rank, int nranks) {
    if (!c || !id_128_bytes || nranks < 1 || rank < 0 || rank >= nranks) return RTX_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->comm) {
        int rc = drain_slots(c);  // a gather of the old communicator may still be in flight on a slot's stream
        if (rc != RTX_OK) return rc;
        NCCL_TRY(c, ncclCommDestroy(c->comm));
        c->comm = nullptr;
    }
    ncclUniqueId id;
    std::memcpy(&id, id_128_bytes, sizeof id);
    NCCL_TRY(c, ncclCommInitRank(&c->comm, nranks, id, rank));
    c->comm_rank = rank;
    c->comm_size = nranks;
    return RTX_OK;
}

int rtx_comm_info(rtx_ctx *, char *buf, size_t size) {
    if (!buf || size == 0) return RTX_ERR_INVALID;
    int nv = 0, hv = 0;
    (void)ncclGetVersion(&nv);
    (void)hipRuntimeGetVersion(&hv);
    Dl_info dn{}, dh{};
    const char *pn = dladdr(reinterpret_cast<void *>(&ncclAllGather), &dn) && dn.dli_fname ? dn.dli_fname : "?";
    const char *ph = dladdr(reinterpret_cast<void *>(&hipStreamSynchronize), &dh) && dh.dli_fname ? dh.dli_fname : "?";
    char real_n[PATH_MAX], real_h[PATH_MAX];
    if (realpath(pn, real_n)) pn = real_n;
    if (realpath(ph, real_h)) ph = real_h;
    std::snprintf(buf, size, "RCCL %d.%d.%d (%s); HIP runtime %d (%s)", nv / 10000, (nv / 100) % 100, nv % 100, pn, hv, ph);
    return RTX_OK;
}

int rtx_comm_rank(rtx_ctx *c) { return c ? c->comm_rank : RTX_ERR_INVALID; }
int rtx_comm_size(rtx_ctx *c) { return c ? c->comm_size : RTX_ERR_INVALID; }

uint64_t rtx_gather_bytes_per_rank(rtx_ctx *c, uint32_t tile) {
    if (!c || !c->has_scene) return 0;
    rtx_partition p{tile ? tile : 32u, 0, (uint32_t)c->comm_size};  // rank 0 owns the most tiles
    Partition pt;
    if (make_partition(c, &p, pt) != RTX_OK) return 0;
    return pt.npix * 3;
}

int rtx_gather_image(rtx_ctx *c, uint32_t slot, uint32_t tile, const void *d_tiles, void *d_all, void *d_image) {
    if (!c || !d_tiles || !d_all || slot >= c->pipes.size()) return RTX_ERR_INVALID;
    if (!c->has_scene) {
        c->err = "no scene uploaded";
        return RTX_ERR_NO_SCENE;
    }
    if (!c->comm) {
        c->err = "rtx_comm_init has not been called";
        return RTX_ERR_COMM;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t bytes = rtx_gather_bytes_per_rank(c, tile);
    if (bytes == 0) {
        c->err = "bad tile size";
        return RTX_ERR_INVALID;
    }
    hipStream_t st = c->pipes[slot].main;  // behind the slot's render, before its next one
    NCCL_TRY(c, ncclAllGather(d_tiles, d_all, (size_t)bytes, ncclUint8, c->comm, st));
    if (d_image)
        launch_deinterleave(static_cast<const uint8_t *>(d_all), (size_t)bytes, (uint32_t)c->comm_size, tile ? tile : 32u, c->ds.width,
                            c->ds.height, static_cast<uint8_t *>(d_image), st);
    HIP_TRY(c, hipGetLastError());
    return RTX_OK;
}

int rtx_assemble_image(rtx_ctx *c, uint32_t slot, uint32_t tile, uint32_t nranks, const void *d_all, void *d_image) {
    if (!c || !d_all || !d_image || slot >= c->pipes.size()) return RTX_ERR_INVALID;
    if (!c->has_scene) {
        c->err = "no scene uploaded";
        return RTX_ERR_NO_SCENE;
    }
    rtx_partition p0{tile ? tile : 32u, 0, nranks};
    Partition pt;
    int rc = make_partition(c, &p0, pt);  // rank 0 owns the largest share: the stride between the ranks' buffers
    if (rc != RTX_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    launch_deinterleave(static_cast<const uint8_t *>(d_all), (size_t)pt.npix * 3, nranks, pt.tile, c->ds.width, c->ds.height,
                        static_cast<uint8_t *>(d_image), c->pipes[slot].main);
    HIP_TRY(c, hipGetLastError());
    return RTX_OK;
}

int rtx_trace_primary(rtx_ctx *c, uint32_t sample, int use_jitter, uint64_t seed, uint32_t flags, uint32_t *out_obj,
                      float *out_t) {
    if (!c || !out_obj || !out_t) return RTX_ERR_INVALID;
    if (!c->has_scene) {
        c->err = "no scene uploaded";
        return RTX_ERR_NO_SCENE;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    Partition pt;
    int rc = make_partition(c, nullptr, pt);
    if (rc != RTX_OK) return rc;
    uint64_t chunk = std::min<uint64_t>(pt.npix, 1ull << 24);
    Pipe &pp = c->pipes[0];
    rc = ensure_frame(c, pp, chunk, 1);
    if (rc != RTX_OK) return rc;
    flags |= c->force_flags;
    LaunchCfg lc{c->stream, c->persistent_blocks, false, (flags & RTX_FLAG_REFERENCE_TRAVERSAL) != 0};
    lc.xform_cache = c->any_xform;
    DevBuf b_obj, b_t;
    HIP_TRY(c, b_obj.alloc(chunk * 4));
    HIP_TRY(c, b_t.alloc(chunk * 4));
    uint32_t *d_obj = b_obj.as<uint32_t>();
    float *d_t = b_t.as<float>();
    std::vector<uint32_t> h_obj(chunk);
    std::vector<float> h_t(chunk);
    int result = RTX_OK;
    for (uint64_t q0 = 0; q0 < pt.npix && result == RTX_OK; q0 += chunk) {
        uint32_t n = (uint32_t)std::min<uint64_t>(chunk, pt.npix - q0);
        BatchInfo info{};
        info.q0 = (uint32_t)q0;
        info.npix = n;
        info.s0 = sample;
        info.ns = 1;
        info.spp = 1;
        info.use_jitter = use_jitter ? 1 : 0;
        info.seed_lo = (uint32_t)seed;
        info.seed_hi = (uint32_t)(seed >> 32);
        info.tile = pt.tile;
        info.rank = 0;
        info.nranks = 1;
        info.tiles_x = pt.tiles_x;
        info.tiles_y = pt.tiles_y;
        info.max_levels = 1;
        (void)reset_counters(c, pp);
        pp.fr.stripe = 64u;
        launch_raygen(c->ds, pp.fr, info, lc);
        launch_path(c->ds, pp.fr, 1, false, lc);
        launch_export_hits(c->ds, pp.fr, n, d_obj, d_t, lc);
        hipError_t e = hipMemcpyAsync(h_obj.data(), d_obj, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h_t.data(), d_t, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            c->err = std::string("trace_primary: ") + hipGetErrorString(e);
            result = RTX_ERR_HIP;
            break;
        }
        for (uint32_t i = 0; i < n; i++) {
            uint32_t px, py;
            if (slot_to_pixel(pt, c->ds.width, c->ds.height, q0 + i, px, py)) {
                out_obj[(size_t)py * c->ds.width + px] = h_obj[i];
                out_t[(size_t)py * c->ds.width + px] = h_t[i];
            }
        }
    }
    return result;
}

static int rays_impl(rtx_ctx *c, const float *org_dir, uint64_t n, uint32_t flags, uint32_t depth, uint32_t *out_obj,
                     float *out_t, float *out_rgb) {
    if (!c->has_scene) {
        c->err = "no scene uploaded";
        return RTX_ERR_NO_SCENE;
    }
    if (n == 0) return RTX_OK;
    flags |= c->force_flags;
    if (depth > 8) {
        c->err = "depth > 8 not supported";
        return RTX_ERR_CAPACITY;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t levels = out_rgb ? depth : 1;
    uint64_t chunk = std::min<uint64_t>(n, 1ull << 22);
    Pipe &pp = c->pipes[0];
    int rc = ensure_frame(c, pp, chunk, std::max(levels, 1u));
    if (rc != RTX_OK) return rc;
    LaunchCfg lc{c->stream, c->persistent_blocks, false, (flags & RTX_FLAG_REFERENCE_TRAVERSAL) != 0};
    lc.xform_cache = c->any_xform;
    DevBuf b_rays, b_f, b_obj;
    HIP_TRY(c, b_rays.alloc(chunk * 24));
    HIP_TRY(c, b_f.alloc(chunk * 12));
    HIP_TRY(c, b_obj.alloc(chunk * 4));
    float *d_rays = b_rays.as<float>(), *d_f = b_f.as<float>();
    uint32_t *d_obj = b_obj.as<uint32_t>();
    int result = RTX_OK;
    for (uint64_t i0 = 0; i0 < n; i0 += chunk) {
        uint32_t m = (uint32_t)std::min<uint64_t>(chunk, n - i0);
        hipError_t e = hipMemcpyAsync(d_rays, org_dir + i0 * 6, (size_t)m * 24, hipMemcpyHostToDevice, c->stream);
        (void)reset_counters(c, pp);
        pp.fr.stripe = 64u;
        launch_load_rays(pp.fr, d_rays, m, lc);
        if (out_rgb) {
            if (run_path(c, pp, pp.events, lc, levels, false, 0) != RTX_OK) e = hipErrorUnknown;
            launch_export_colors(pp.fr, m, d_f, lc);
            if (e == hipSuccess) e = hipMemcpyAsync(out_rgb + i0 * 3, d_f, (size_t)m * 12, hipMemcpyDeviceToHost, c->stream);
        } else {
            launch_path(c->ds, pp.fr, 1, false, lc);
            launch_export_hits(c->ds, pp.fr, m, d_obj, d_f, lc);
            if (e == hipSuccess) e = hipMemcpyAsync(out_obj + i0, d_obj, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(out_t + i0, d_f, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            c->err = std::string("trace rays: ") + hipGetErrorString(e);
            result = RTX_ERR_HIP;
            break;
        }
    }
    return result;
}

int rtx_trace_rays(rtx_ctx *c, const float *org_dir, uint64_t n, uint32_t flags, uint32_t *out_obj, float *out_t) {
    if (!c || (n && (!org_dir || !out_obj || !out_t))) return RTX_ERR_INVALID;
    return rays_impl(c, org_dir, n, flags, 1, out_obj, out_t, nullptr);
}

int rtx_shade_rays(rtx_ctx *c, const float *org_dir, uint64_t n, uint32_t depth, float *out_rgb) {
    if (!c || (n && (!org_dir || !out_rgb))) return RTX_ERR_INVALID;
    return rays_impl(c, org_dir, n, 0, depth, nullptr, nullptr, out_rgb);
}

int rtx_host_bvh_digest(const rtx_scene_desc *d, int threads, uint32_t *leaf_order, uint32_t *preorder_sizes,
                        uint32_t *depth, float *total_sa, uint32_t *n_internal) {
    if (!d) return RTX_ERR_INVALID;
    BuiltBvh bvh;
    std::string err;
    // RTX_TOP_SPLITTER=host: the top of the tree through the TopSplitter interface (its host implementation), as
    // rtx_upload_scene runs it through the GPU one -- the CPU tests compare the two ways of building
    std::unique_ptr<TopSplitter> top;
    if (const char *ts = std::getenv("RTX_TOP_SPLITTER"))
        if (std::string(ts) == "host") top.reset(new_host_top_splitter());
    int rc = build_bvh(*d, threads, bvh, err, top.get(), 512u);
    if (rc != RTX_OK) {
        g_init_error = err;
        return rc;
    }
    FlatScene fs;
    rc = flatten_scene(*d, bvh, fs, err, threads > 0 ? threads : 1);
    if (rc != RTX_OK) {
        g_init_error = err;
        return rc;
    }
    if (leaf_order) std::copy(bvh.leaf_order.begin(), bvh.leaf_order.end(), leaf_order);
    if (preorder_sizes) std::copy(fs.preorder_sizes.begin(), fs.preorder_sizes.end(), preorder_sizes);
    if (depth) *depth = bvh.depth;
    if (total_sa) *total_sa = fs.total_sa;
    if (n_internal) *n_internal = (uint32_t)fs.nodes.size();
    return (int)fs.preorder_sizes.size();
}

int rtx_host_m4(int op, const float *in, float *out) {
    if (!in || !out) return RTX_ERR_INVALID;
    auto put = [&](const M4 &m) { std::memcpy(out, m.m, sizeof(m.m)); };
    auto get = [&](const float *p) {
        M4 m;
        std::memcpy(m.m, p, sizeof(m.m));
        return m;
    };
    auto put3 = [&](V3 v) {
        out[0] = v.x;
        out[1] = v.y;
        out[2] = v.z;
    };
    switch (op) {
        case RTX_M4_FROM_ANGLE_X: put(m4_from_angle_x(in[0])); return RTX_OK;
        case RTX_M4_FROM_ANGLE_Y: put(m4_from_angle_y(in[0])); return RTX_OK;
        case RTX_M4_FROM_ANGLE_Z: put(m4_from_angle_z(in[0])); return RTX_OK;
        case RTX_M4_FROM_TRANSLATION: put(m4_from_translation(v3(in[0], in[1], in[2]))); return RTX_OK;
        case RTX_M4_FROM_NONUNIFORM_SCALE: put(m4_from_nonuniform_scale(in[0], in[1], in[2])); return RTX_OK;
        case RTX_M4_MUL: put(get(in) * get(in + 16)); return RTX_OK;
        case RTX_M4_INVERT: {
            M4 r;
            if (!m4_invert(get(in), r)) return RTX_ERR_DEGENERATE;
            put(r);
            return RTX_OK;
        }
        case RTX_M4_LOOK_AT: put(m4_look_at(v3(in[0], in[1], in[2]), v3(in[3], in[4], in[5]), v3(in[6], in[7], in[8]))); return RTX_OK;
        case RTX_M4_AXIS_SCALING: {
            const M4 m = get(in);
            put3(v3(magnitude(transform_vector(m, v3(1, 0, 0))), magnitude(transform_vector(m, v3(0, 1, 0))),
                    magnitude(transform_vector(m, v3(0, 0, 1)))));
            return RTX_OK;
        }
        case RTX_M4_TRANSFORM_POINT: put3(transform_point(get(in), v3(in[16], in[17], in[18]))); return RTX_OK;
        case RTX_M4_TRANSFORM_VECTOR: put3(transform_vector(get(in), v3(in[16], in[17], in[18]))); return RTX_OK;
        case RTX_M4_OBJECT_BOX: {  // the builder's own box routine on a one-object description
            const int32_t type = (int32_t)in[0], zero = 0;
            if (type < 0 || type > 2) return RTX_ERR_INVALID;
            float xf[32];
            std::memcpy(xf, in + 13, 16 * sizeof(float));
            std::memset(xf + 16, 0, 16 * sizeof(float));  // (world_to_object is not looked at by the box)
            rtx_scene_desc d{};
            d.n_objects = 1;
            d.obj_type = &type;
            d.obj_verts = in + 1;
            d.obj_xform = &zero;
            d.n_xforms = 1;
            d.xforms = xf;
            object_bounding_box(d, 0, out, out + 3);
            return RTX_OK;
        }
        default: return RTX_ERR_INVALID;
    }
}

float rtx_scene_hit_overhang(const rtx_scene_desc *d) {
    if (!d) return (float)RTX_ERR_INVALID;
    for (uint32_t i = 0; i < d->n_objects; i++) {
        int type = d->obj_type[i], xf = d->obj_xform[i];
        if (type < 0 || type > 2 || xf < 0 || (uint32_t)xf >= d->n_xforms) {
            g_init_error = "object with bad type / transform index";
            return (float)RTX_ERR_INVALID;
        }
    }
    std::vector<float> per_object;
    return hit_overhang(*d, per_object);
}

}  // extern "C"
