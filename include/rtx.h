/*
 * rtx.h -- C ABI of the MI355X-native trace backend for ellishg/rust-raytracer.
 *
 * The reference has no FFI of its own: its hot path sits behind the free function
 * `world::render(camera, objects, lights, background_color, samples_per_pixel,
 * max_ray_bounces, path, num_threads)` (src/world.rs:17-29).  These entry points are
 * what a thin `rtx-sys` crate binds so that `render` keeps its signature and replaces
 * the thread-pool region src/world.rs:38-74 with one call.
 *
 * Conventions: every function returns RTX_OK (0) or a negative rtx_status and never
 * throws or aborts across the boundary; rtx_last_error() gives the text.  Input
 * pointers are borrowed for the duration of the call; output buffers are owned by the
 * caller.  One rtx_ctx per GPU, called from one host thread at a time (a render runs on the
 * stream of its render slot; slot 0's is rtx_stream()).  The library has no CPU fallback:
 * without a usable HIP device rtx_init fails.
 *
 * Limits the reference does not have (checked at rtx_upload_scene / rtx_render*, reported as
 * RTX_ERR_CAPACITY or RTX_ERR_INVALID with a message, never silently wrong): <= 16 lights;
 * max_ray_bounces <= 8 (reference: u16); <= 4 reflective / refractive leaves per material;
 * Composition nesting <= 4; <= 8,192 distinct transforms; <= 65,536 materials; <= 2^28 objects;
 * BVH depth <= 198; < 2^32 path nodes per batch (larger frames are split into batches by the
 * library itself, results unchanged).
 */
#ifndef RTX_H
#define RTX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_INVALID = -1,    /* bad argument / malformed scene */
    RTX_ERR_HIP = -2,        /* a HIP runtime call failed */
    RTX_ERR_NO_SCENE = -3,   /* render before rtx_upload_scene */
    RTX_ERR_CAPACITY = -4,   /* scene exceeds a documented device limit */
    RTX_ERR_DEGENERATE = -5, /* BVH build cannot terminate (reference recurses forever, bvh.rs:221-230) */
    RTX_ERR_IO = -6,
    RTX_ERR_COMM = -7        /* an RCCL call failed (multi-GPU gather) */
} rtx_status;

/* Object / material / light vocabularies of the reference. */
enum { RTX_OBJ_SPHERE = 0, RTX_OBJ_TRIANGLE = 1, RTX_OBJ_QUAD = 2 };           /* object.rs:14-18 */
enum { RTX_MAT_NONE = 0, RTX_MAT_PHONG = 1, RTX_MAT_REFLECTIVE = 2,
       RTX_MAT_REFRACTIVE = 3, RTX_MAT_COMPOSITION = 4 };                      /* material.rs:21-31 */
enum { RTX_TEX_NONE = 0, RTX_TEX_FLAT = 1, RTX_TEX_IMAGE = 2 };                /* material.rs:14-18 */
enum { RTX_LIGHT_AMBIENT = 0, RTX_LIGHT_POINT = 1, RTX_LIGHT_DIRECTIONAL = 2,
       RTX_LIGHT_CONE = 3 };                                                   /* light.rs:12-17 */

/* Flat restatement of what `render` receives: Vec<Object> (object.rs:20-26), the
 * Material trees (material.rs:33-37), Vec<Light> (light.rs:19-22), background
 * (main.rs:111) and the Camera (camera.rs:5-9).  All matrices column-major f32. */
typedef struct rtx_scene_desc {
    uint32_t n_objects;
    const int32_t *obj_type;
    const float *obj_verts;       /* n*12: sphere = centre xyz + radius; tri = a,b,c; quad = a,b,c,d */
    const int32_t *obj_xform;     /* n: index into xforms */
    const int32_t *obj_material;  /* n: index into materials */
    uint32_t n_xforms;
    const float *xforms;          /* n_xforms*32: object_to_world (16) then world_to_object (16) */
    uint32_t n_materials;
    const int32_t *mat_root;      /* root node of the material tree */
    const int32_t *mat_tex_kind;
    const float *mat_tex_color;   /* n*3 (Flat) */
    const int32_t *mat_tex_id;    /* (Image) */
    uint32_t n_mat_nodes;
    const int32_t *node_type;
    const float *node_params;     /* n*3: Phong diffuse,specular,shininess | Refractive index */
    const int32_t *node_child_first;
    const int32_t *node_child_count;
    uint32_t n_mat_children;
    const int32_t *child_node;    /* Composition entries: (material node, coefficient) */
    const float *child_coef;
    uint32_t n_textures;
    const int32_t *tex_w;
    const int32_t *tex_h;
    const uint64_t *tex_offset;   /* byte offset of each RGB8 image in tex_data */
    const uint8_t *tex_data;
    uint32_t n_lights;
    const int32_t *light_type;
    const float *light_pos;       /* n*3 */
    const float *light_dir;       /* n*3, normalised as Light::new_* does (light.rs:42,54) */
    const float *light_angle_deg; /* n */
    const float *light_color;     /* n*3 */
    float background[3];
    float camera_to_world[16];    /* camera.rs:20 */
    uint32_t width, height;
} rtx_scene_desc;

typedef struct rtx_ctx rtx_ctx;

/* Counters of one render call.  One ray = one Bvh::get_closest_intersection query of the
 * reference (world.rs:118, light.rs:90,111). */
typedef struct rtx_stats {
    uint64_t primary_rays, bounce_rays, shadow_rays;
    /* records fetched, per kernel class; 0 unless RTX_FLAG_COUNT_VISITS is set.  Nodes are the
     * 128-byte four-slot nodes (64-byte binary nodes under RTX_FLAG_REFERENCE_TRAVERSAL),
     * primitives 64-byte records. */
    uint64_t closest_nodes, closest_prims;  /* closest-hit rays: internal nodes, primitive records */
    uint64_t shadow_nodes, shadow_prims;    /* shadow queries */
    double trace_ms;            /* host wall time raygen..accumulate (stream-synchronised) */
    double path_kernel_ms;      /* HIP-event time of the persistent path kernel, summed over the batches
                                   (one launch per batch: every ray of every bounce level) */
    double resolve_ms;          /* HIP-event time of the resolve passes behind it, summed over the batches */
    uint32_t path_launches;
    uint32_t batches;
    uint32_t max_ray_records;   /* most node + primitive records visited by one ray (COUNT_VISITS) */
    uint32_t reserved;
} rtx_stats;

/* BVH facts after upload (bvh.rs:18-24 logs the same numbers). */
typedef struct rtx_bvh_info {
    uint32_t n_objects, n_internal, n_leaves, depth;
    float total_sa;
    double build_seconds, flatten_seconds, upload_seconds;
    uint64_t device_bytes;
} rtx_bvh_info;

/* Image-tile partition (one process per GPU): the image is cut into tile x tile pixel
 * squares numbered row-major; this context renders tiles with index % nranks == rank.
 * Replaces the column-per-job thread pool of world.rs:38-68. */
typedef struct rtx_partition {
    uint32_t tile;    /* tile edge in pixels, multiple of 8 (default 32) */
    uint32_t rank;
    uint32_t nranks;
} rtx_partition;

enum { RTX_FLAG_COUNT_VISITS = 1u,   /* fill rtx_stats.nodes_visited / prims_tested (slower) */
       RTX_FLAG_REFERENCE_TRAVERSAL = 2u /* visit both children, no pruning (bvh.rs:359-395 verbatim) */ };

/* Create a context on HIP device `device`.  Replaces ThreadPool::new (world.rs:38). */
int rtx_init(int device, rtx_ctx **out);
void rtx_destroy(rtx_ctx *ctx);
const char *rtx_last_error(rtx_ctx *ctx); /* ctx may be NULL: last rtx_init error */

/* Replaces World::new -> Bvh::new (world.rs:98-111, bvh.rs:13-26, 334-357): builds the
 * reference's SAH tree topology on the host, flattens it into 64-byte node / primitive
 * records and uploads everything to HBM once. */
int rtx_upload_scene(rtx_ctx *ctx, const rtx_scene_desc *desc);
/* Host threads the next rtx_upload_scene builds the BVH with: render's `num_threads`
 * (world.rs:28, -t/--threads of main.rs:40-47); 0 = min(16, hardware threads).  The tree is the
 * same for every thread count. */
int rtx_set_build_threads(rtx_ctx *ctx, int threads);
int rtx_bvh_info_get(rtx_ctx *ctx, rtx_bvh_info *out);
/* Leaf-order permutation of object indices and preorder node sizes of the built tree
 * (parity check of the topology against the reference builder). Returns node count. */
int rtx_bvh_digest(rtx_ctx *ctx, uint32_t *leaf_order, uint32_t *preorder_sizes);

/* Replaces the pool + gather of world.rs:38-74 plus the quantisation of world.rs:76-80:
 * traces every (pixel, sample) of this rank's tiles and resolves them to RGB8.
 * out_rgb8: host buffer W*H*3 (row-major, pixel (0,0) top-left); only this rank's pixels
 * are written.  out_rgb_f32 (optional) receives the averaged, clamped colour (W*H*3).
 * spp == 1 uses no jitter (world.rs:47-51); spp > 1 uses the counter-based jitter(seed). */
int rtx_render(rtx_ctx *ctx, uint32_t spp, uint32_t max_bounces, uint64_t seed,
               const rtx_partition *part_or_null, uint32_t flags,
               uint8_t *out_rgb8, float *out_rgb_f32_or_null, rtx_stats *stats_or_null);

/* Same, but the result stays in HBM: d_tiles_rgb8 is a DEVICE buffer of
 * rtx_partition_pixels()*3 bytes receiving this rank's tiles packed
 * [local tile][row in tile][col in tile][rgb] (pixels outside the image are zero).
 * The launch is asynchronous on rtx_stream(); used for the multi-GPU RCCL gather. */
int rtx_render_device(rtx_ctx *ctx, uint32_t spp, uint32_t max_bounces, uint64_t seed,
                      const rtx_partition *part_or_null, uint32_t flags,
                      void *d_tiles_rgb8, rtx_stats *stats_or_null);
/* Render slots: a context holds rtx_slot_count() (4; RTX_PIPES) independent slots, each with its own
 * stream and wavefront buffers, so that consecutive frames -- or the tile shares of a multi-GPU
 * frame sequence -- overlap on the GPU: a render ends with a drain of a few long rays that leaves
 * most of the machine idle, and the next frame's bulk fills it (the thread pool of world.rs:38-68
 * keeps its workers busy the same way, column by column).  rtx_render_device(...) is slot 0.
 * Renders in one slot follow each other on rtx_slot_stream(slot); order consumers of d_tiles after
 * that stream, and do not reuse one d_tiles buffer for two slots in flight.  No counterpart in
 * the reference. */
int rtx_slot_count(rtx_ctx *ctx);
int rtx_render_device_slot(rtx_ctx *ctx, uint32_t slot, uint32_t spp, uint32_t max_bounces, uint64_t seed,
                           const rtx_partition *part_or_null, uint32_t flags,
                           void *d_tiles_rgb8, rtx_stats *stats_or_null);
void *rtx_slot_stream(rtx_ctx *ctx, uint32_t slot);
int rtx_slot_synchronize(rtx_ctx *ctx, uint32_t slot);
/* Waits for the slot's last render and reports its counters and kernel times (what stats_or_null
 * would have received, had the call not been asynchronous; trace_ms: enqueue to completion). */
int rtx_slot_stats(rtx_ctx *ctx, uint32_t slot, rtx_stats *out);

/* rtx_render_device for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same result bytes as
 * rtx_render_device, but the call makes no hipMalloc / hipFree / hipHostMalloc and creates no event:
 * the wavefront state of the render (path nodes, shadow queues, counters, queue heads, stack spill,
 * pixel sums) lives in a static 64 MiB device array that each GPU holds from library load to process
 * end.  The render runs as ONE batch on a reduced grid, serially on rtx_stream(), asynchronously;
 * no stats.  rtx_synchronize / rtx_slot_synchronize(0) wait for it and return RTX_ERR_CAPACITY if its
 * wavefront pool or a wave's work stack overflowed (the image is then incomplete; the flag travels
 * through a pinned word that rtx_init allocated).  A render whose state does not fit -- roughly
 * pixel slots x spp x 72 B x (bounce levels actually reachable) plus 5-20 MiB of stack spill --
 * is rejected with RTX_ERR_CAPACITY and rtx_last_error says how much it needed.  All contexts of
 * one GPU share the array: calls that overlap on one GPU (two contexts, or a second call before
 * the first has finished on another stream) race; calls in sequence on one context are ordered. */
int rtx_render_device_noalloc(rtx_ctx *ctx, uint32_t spp, uint32_t max_bounces, uint64_t seed,
                              const rtx_partition *part_or_null, uint32_t flags, void *d_tiles_rgb8);
/* Number of pixel slots (tiles * tile * tile) this rank owns under `part`. */
uint64_t rtx_partition_pixels(rtx_ctx *ctx, const rtx_partition *part_or_null);
void *rtx_stream(rtx_ctx *ctx); /* hipStream_t of slot 0: rtx_render_device, rtx_render_device_noalloc, the trace calls */
int rtx_synchronize(rtx_ctx *ctx); /* waits for every slot; reports a capacity overflow of an asynchronous render */

/* Multi-GPU, one process per GPU: the frame is cut into tiles dealt round-robin over the ranks
 * (rtx_partition), the scene is replicated, and ONE collective per frame brings the finished RGB8
 * tiles together -- the receive loop of world.rs:70-74.  The context owns the RCCL communicator.
 *   rank 0: rtx_comm_unique_id(id); hand the 128 bytes to every rank by any means (file, socket, MPI)
 *   every rank: rtx_init(local device) ... rtx_comm_init(ctx, id, rank, nranks) (collective)
 *   per frame:  rtx_render_device_slot(ctx, slot, .., part = {tile, rank, nranks}, .., d_tiles, NULL)
 *               rtx_gather_image(ctx, slot, tile, d_tiles, d_all, d_image)   (collective, asynchronous)
 * d_tiles must hold rtx_gather_bytes_per_rank() bytes (the largest rank's share: ranks with one
 * tile fewer leave the tail untouched), d_all nranks times that, d_image (optional) W*H*3: the
 * assembled row-major image, on every rank.  The gather runs on the slot's stream: behind the
 * slot's render, ahead of its next one.  With one rank it is a device copy plus the reordering. */
#define RTX_COMM_ID_BYTES 128
int rtx_comm_unique_id(void *id_128_bytes);
int rtx_comm_init(rtx_ctx *ctx, const void *id_128_bytes, int rank, int nranks);
int rtx_comm_rank(rtx_ctx *ctx);
int rtx_comm_size(rtx_ctx *ctx);
/* Which collective library and HIP runtime this process has bound the library's symbols to: "RCCL <version>
 * (<path of the shared object that ncclAllGather resolves to>); HIP runtime <version> (<path>)", NUL-terminated,
 * truncated to `size`.  A process that loads another copy of these libraries first (a framework that bundles
 * its own) runs rtx_gather_image on THAT copy: bench.py prints this string next to the gather it timed.  No
 * counterpart in the reference.  ctx may be NULL. */
int rtx_comm_info(rtx_ctx *ctx, char *buf, size_t size);
uint64_t rtx_gather_bytes_per_rank(rtx_ctx *ctx, uint32_t tile);
int rtx_gather_image(rtx_ctx *ctx, uint32_t slot, uint32_t tile, const void *d_tiles_rgb8, void *d_all_tiles,
                     void *d_image_rgb8_or_null);
/* The reordering half of rtx_gather_image on its own, for hosts that bring their own collective (MPI,
 * a communicator of their own): d_all_tiles holds the tile buffers of `nranks` ranks, each
 * rtx_partition_pixels(ctx, {tile, 0, nranks}) * 3 bytes apart, as an all-gather leaves them;
 * d_image_rgb8 receives the row-major W*H*3 image.  Asynchronous, on the slot's stream. */
int rtx_assemble_image(rtx_ctx *ctx, uint32_t slot, uint32_t tile, uint32_t nranks, const void *d_all_tiles,
                       void *d_image_rgb8);

/* Hit-ID parity: closest hit of the primary ray of sample `sample` of every pixel
 * (camera.rs:32-47 + bvh.rs:33-35).  out_obj: object index in rtx_scene_desc order,
 * 0xFFFFFFFF = miss; out_t: world-space t.  Both W*H, row-major. */
int rtx_trace_primary(rtx_ctx *ctx, uint32_t sample, int use_jitter, uint64_t seed,
                      uint32_t flags, uint32_t *out_obj, float *out_t);
/* Closest hit of n caller-supplied rays (origin xyz, direction xyz; the direction is
 * normalised as Ray::new does, ray.rs:11-16).  Bvh::get_closest_intersection, bvh.rs:33-35. */
int rtx_trace_rays(rtx_ctx *ctx, const float *org_dir, uint64_t n, uint32_t flags,
                   uint32_t *out_obj, float *out_t);
/* World::trace_ray(ray, depth) for n caller-supplied rays (world.rs:115-131): rgb f32. */
int rtx_shade_rays(rtx_ctx *ctx, const float *org_dir, uint64_t n, uint32_t depth,
                   float *out_rgb);

#ifdef __cplusplus
}
#endif
#endif
