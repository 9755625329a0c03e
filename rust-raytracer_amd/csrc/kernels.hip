// HIP kernels of the trace path for gfx950 (wave64).  Build with -ffp-contract=off: every
// f32 operation that decides a hit (box slab test, primitive predicates, world t) is the
// same IEEE operation in the same order as the reference's Rust code, so hit ids and t
// come out bit-identical; only libm calls (atan2/acos/pow) differ in the last ulp.
//
//   raygen          camera.rs:32-47 + the spp loop of world.rs:54-61
//   path            World::trace_ray (world.rs:115-131) for a queue of primary samples in ONE persistent
//                   launch: per lane, depth-first through the sample's path tree -- closest hit
//                   (bvh.rs:60-131,359-395 + object.rs:145-220), one any-hit shadow query per light
//                   (light.rs:81-122), the bounce rays of material.rs:185-203; per-lane stack in LDS
//   resolve         material.rs:53-76,98-206,220-240 + color.rs (clamped algebra), bottom-up per level
//   accumulate      world.rs:61-63 + color.rs:25-31
#include "kernels.h"

#include <algorithm>

namespace rtx {

// Scratch of rtx_render_device_noalloc (path nodes, shadow queues, counters, queue heads, stack
// spill, pixel sums of one small render).  In host code this name is only a handle for
// hipGetSymbolAddress: kernels and memsets get the address that noalloc_scratch_address returns.
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

hipError_t noalloc_scratch_address(int device, void **ptr) {
    constexpr int MAX_DEVICES = 64;
    static void *cached[MAX_DEVICES] = {};
    *ptr = nullptr;
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    if (!cached[device]) {  // the caller has made `device` current (rtx_init)
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        cached[device] = p;
    }
    *ptr = cached[device];
    return hipSuccess;
}

namespace {

constexpr int BLOCK = 256;
constexpr int RESOLVE_SCAN = 8;  // stretches of BLOCK pool nodes a resolve block lists before it works them off
// Threads per block of the two persistent trace kernels (measured: 64 / 128 / 256 equal on
// the headline frame, 512 / 1024 -10..15 %: wave slots come back block by block).
#ifndef RTX_TRACE_BLOCK
#define RTX_TRACE_BLOCK 256
#endif
constexpr int TBLOCK = RTX_TRACE_BLOCK;
// The first TOP_NODES wide nodes (numbered breadth-first by the flattener) are kept in LDS,
// one dwordx4 piece per array ([piece][node]: lanes on different nodes read consecutive
// 16-byte words, no bank conflicts): the top of the tree is where most node visits go, and
// an LDS read costs neither an L1 tag lookup per lane nor an L2 round trip.
#ifndef RTX_TOP_NODES
#define RTX_TOP_NODES (RTX_TRACE_BLOCK / 16)
#endif
constexpr uint32_t TOP_NODES = RTX_TOP_NODES;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) v4f *LdsNodePtr;  // typed so the reads stay ds_read_b128
constexpr uint32_t PRIM_IDENTITY_XFORM = 0x8000u;  // Prim::meta bit 15
// |1/d_i| beyond which axis i is left out of pruning, and the pruning slack per unit of scene
// scale (init_ray): a hit point and the faces of its leaf box agree to delta ~ 2.2e-7 * (scale
// + |o|) in space (f32 matrix products on coordinates of that size), i.e. delta * ILL_INV in t
// over the axes pruning looks at; SLACK covers that (1.4e-5) with the f32 error of t itself.
#ifndef RTX_ILL_INV
#define RTX_ILL_INV 64.0f
#endif
#ifndef RTX_SLACK
#define RTX_SLACK 2.4e-5f
#endif
constexpr float ILL_INV = RTX_ILL_INV;  // |1/d_i| beyond which axis i is left out of pruning (init_ray)
#ifndef RTX_TRACE_WAVES
#define RTX_TRACE_WAVES 5  // waves per SIMD the trace kernels are register-allocated for (<= 96 VGPRs)
#endif

// Experiment switches of round 3, second half (each measured on its own, tests/run_gpu_r03_trim.sh)
#ifndef RTX_LIGHTS_K
#define RTX_LIGHTS_K 1      // loops over the lights read them from the kernel arguments (scalar loads)
#endif
#ifndef RTX_FAST_THR
#define RTX_FAST_THR 1      // the pass threshold (a scheduling heuristic) with v_rcp / v_sqrt instead of IEEE division and square root
#endif
#ifndef RTX_ONE_SQRT
#define RTX_ONE_SQRT 1      // a shadow query's direction and distance from ONE square root
#endif
#ifndef RTX_ROOM_MUL
#define RTX_ROOM_MUL 1      // admission of new primaries without the integer division in the common case
#endif
#ifndef RTX_SHADE_IN_LDS
#define RTX_SHADE_IN_LDS 1  // the hits that wait for a shading pass are listed in LDS (else behind the wave's work stack)
#endif
#ifndef RTX_FAST_INV
#define RTX_FAST_INV 0      // 1/d for the slab tests from v_rcp + one Newton step instead of the IEEE division
#endif
#if RTX_LIGHTS_K
#define RTX_LIGHT(sc, l) (sc).lights_k[l]
#else
#define RTX_LIGHT(sc, l) (sc).lights[l]
#endif

// ---------------------------------------------------------------- cgmath-order arithmetic
struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float magnitude(F3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ F3 normalize(F3 a) { return a * (1.0f / magnitude(a)); }
__device__ __forceinline__ float angle_rad(F3 a, F3 b) { return atan2f(magnitude(cross(a, b)), dot(a, b)); }
__device__ __forceinline__ F3 transform_point(const float *M, F3 p) {
    float h[4];
#pragma unroll
    for (int r = 0; r < 4; r++) h[r] = ((M[r] * p.x + M[4 + r] * p.y) + M[8 + r] * p.z) + M[12 + r] * 1.0f;
    float iw = 1.0f / h[3];
    return f3(h[0] * iw, h[1] * iw, h[2] * iw);
}
__device__ __forceinline__ F3 transform_vector(const float *M, F3 v) {
    float h[3];
#pragma unroll
    for (int r = 0; r < 3; r++) h[r] = ((M[r] * v.x + M[4 + r] * v.y) + M[8 + r] * v.z) + M[12 + r] * 0.0f;
    return f3(h[0], h[1], h[2]);
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) {  // utils.rs:5-13
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
__device__ __forceinline__ F3 reflect(F3 v, F3 n) { return normalize(v - 2.0f * dot(v, n) * n); }  // utils.rs:15-17
__device__ __forceinline__ F3 refract(F3 v, F3 normal, float idx) {  // utils.rs:46-59
    float n = dot(v, normal) <= 0.0f ? 1.0f / idx : idx / 1.0f;
    float cos_in = fabsf(dot(v, normal));
    float cos_out = sqrtf(1.0f - (n * n) * (1.0f - (cos_in * cos_in)));  // powf(x, 2.0) == x*x
    return normalize(v * n + (n * cos_in - cos_out) * normal);
}
__device__ __forceinline__ F3 crgb(float r, float g, float b) {  // Color::rgb (color.rs:17-23)
    return f3(clampf(r, 0.0f, 1.0f), clampf(g, 0.0f, 1.0f), clampf(b, 0.0f, 1.0f));
}
__device__ __forceinline__ F3 cadd(F3 a, F3 b) { return crgb(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 cmul(F3 a, F3 b) { return crgb(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 cscale(F3 a, float s) { return crgb(a.x * s, a.y * s, a.z * s); }

constexpr float DEG_TO_RAD = (float)(3.14159265358979323846 / 180.0);
constexpr float RAD_TO_DEG = (float)(180.0 / 3.14159265358979323846);
constexpr float PI_F = 3.14159265358979323846f;

// ---------------------------------------------------------------- jitter (same definition as the oracle)
__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
__device__ __forceinline__ uint32_t jitter_u32(uint32_t seed_lo, uint32_t seed_hi, uint32_t px, uint32_t py,
                                               uint32_t sample, uint32_t dim) {
    uint32_t h = pcg(seed_lo ^ 0x9E3779B9u);
    h = pcg(h ^ seed_hi);
    h = pcg(h + px);
    h = pcg(h + py);
    h = pcg(h + sample * 2u + dim);
    return h;
}

// ---------------------------------------------------------------- wave-level queue allocation
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
// set bits of `m` below this lane (v_mbcnt: no lane mask kept in registers)
__device__ __forceinline__ uint32_t rank_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Queue head of a persistent kernel: lane 0 takes `n` work items, the wave shares the base.
__device__ __forceinline__ uint32_t wave_fetch(uint32_t *head, uint32_t n) {
    uint32_t b = 0;
    if (lane_id() == 0) b = atomicAdd(head, n);
    return __builtin_amdgcn_readfirstlane(b);
}

// ---------------------------------------------------------------- ray state for traversal
struct RayCtx {
    F3 o, d;        // world ray (d normalised by Ray::new)
    F3 inv;         // 1/d, correctly rounded
    // The ray in object space (ray.rs:18-23), two ways (template parameter XFC of the kernels; the host picks):
    // XFC = false, scenes whose objects all have the identity transform (meshes: the headline scene, the soup): the
    //   origin is o, and od is the direction normalised once more, as Ray::new does -- 4 registers and a branch per
    //   test less than a cache: -3 % on the headline frame; an object with a transform gets its ray per test (+4 % on
    //   the Suzanne scene, hence:)
    // XFC = true, scenes with transforms: (oo, od) is the ray in the space of transform `xf`, the last one a test needed.
    F3 oo, od;
    uint32_t xf;
    bool exact_only;  // a zero / non-finite component, or slab products that could overflow: always take the division path
    float slack0;     // absolute pruning slack
    // axis i is ill-conditioned for pruning when |1/d_i| > ILL_INV (see init_ray): derived from
    // inv where needed, not kept in a register
    __device__ __forceinline__ float cap_x() const { return fabsf(inv.x) <= ILL_INV ? INFINITY : -INFINITY; }
    __device__ __forceinline__ float cap_y() const { return fabsf(inv.y) <= ILL_INV ? INFINITY : -INFINITY; }
    __device__ __forceinline__ float cap_z() const { return fabsf(inv.z) <= ILL_INV ? INFINITY : -INFINITY; }
    __device__ __forceinline__ bool ill_any() const {
        return !(fmaxf(fmaxf(fabsf(inv.x), fabsf(inv.y)), fabsf(inv.z)) <= ILL_INV);
    }
};

// Object space of an identity transform: M*(p,1) and M*(v,0) reproduce p and v bit for bit (x*1 + y*0 + z*0 + 0), so
// the products are skipped; Ray::new normalises the direction again (ray.rs:14), which changes a unit vector's last
// bits more often than not.
template <bool XFC>
__device__ __forceinline__ void init_ray(RayCtx &r, const DeviceScene &sc, F3 o, F3 d) {
    r.o = o;
    r.d = d;
#if RTX_FAST_INV
    {   // 1/d within 1 ulp: v_rcp_f32 (1 ulp) + one Newton step; the slab tests' error band allows for 2 u on it
        float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
        ix = __builtin_fmaf(ix, __builtin_fmaf(-d.x, ix, 1.0f), ix);
        iy = __builtin_fmaf(iy, __builtin_fmaf(-d.y, iy, 1.0f), iy);
        iz = __builtin_fmaf(iz, __builtin_fmaf(-d.z, iz, 1.0f), iz);
        r.inv = f3(ix, iy, iz);
    }
#else
    r.inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
#endif
    const bool fin = isfinite(r.inv.x) && isfinite(r.inv.y) && isfinite(r.inv.z) && isfinite(o.x) && isfinite(o.y) &&
                     isfinite(o.z);
    const float big = sc.coord_scale + fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    const float mi = fmaxf(fmaxf(fabsf(r.inv.x), fabsf(r.inv.y)), fabsf(r.inv.z));
    // (every slab parameter (plane - o) * inv of a ray that is not exact_only is finite: |plane|, |o| <= big)
    r.exact_only = !fin || d.x == 0.0f || d.y == 0.0f || d.z == 0.0f || !(big * mi < 1e37f);
    // Pruning compares a box's entry / exit t with hit t values.  A hit point and the faces
    // of its leaf box agree only to eps ~ 1e-6 * scale in space (both come from f32 matrix
    // products), which is eps * |1/d_i| in t along axis i: for a ray nearly parallel to a
    // box face that is unbounded, and a box can "begin" after a hit that lies inside it (found
    // by rays aimed at mesh vertices with one direction component ~1e-4).  Axes with
    // |1/d_i| > ILL_INV are therefore left out of the entry t that pruning looks at (a
    // normalised direction always keeps one axis with |1/d_i| <= 1.74; the centre row and
    // column of every image are such rays), and such rays do not prune boxes behind the
    // origin; slack0 covers eps * ILL_INV for the others.  (Measured instead, round 3: a slack
    // that grows with the worst axis -- the 5 % of rays with such an axis then prune so little
    // that single rays visit 1,500 records and make the launch's tail; and slab parameters in
    // fma form, fma(plane, 1/d, -(o/d)) -- their absolute error u |o_i / d_i| widens the band
    // of those same rays until they visit every box near their path.)
    r.slack0 = RTX_SLACK * big;
    // Pruning also equates the slab parameter t with the distance that a hit reports, which
    // holds for |d| = 1.  Ray::new's normalisation loses that for directions whose squared
    // length leaves the normal f32 range (|d| ~ 1e-23: the dot product is a denormal with a
    // few bits; |d| ~ 1e20: it overflows): such a ray prunes nothing -- an infinite slack
    // turns every pruning comparison off -- and visits what the reference visits.
    if (!(fabsf(dot(d, d) - 1.0f) <= 4e-6f)) r.slack0 = INFINITY;
    // (Hits that lie outside their object's box -- non-planar quads, sheared spheres -- are
    // allowed for per node slot, see node_step.)
    if constexpr (XFC) {
        r.xf = 0xFFFFFFFFu;  // (no transform yet: the first test computes its own)
        r.oo = r.od = f3(0.f, 0.f, 0.f);
    } else {
        r.od = normalize(d);
    }
}

// AABB::intersect decision exactly as bvh.rs:60-131 (true divisions, Interval algebra in
// the reference's nesting order x ∩ (y ∩ z)).
struct Ival {
    int kind;  // 0 Infinite, 1 Closed, 2 Empty
    float a, b;
};
__device__ __forceinline__ Ival ival_new(float a, float b, float x, float slope) {
    Ival r;
    if (slope == 0.0f) {
        r.kind = (a <= x && x <= b) ? 0 : 2;
        r.a = r.b = 0.0f;
        return r;
    }
    float ta = (a - x) / slope, tb = (b - x) / slope;
    r.kind = 1;
    r.a = fminf(ta, tb);
    r.b = fmaxf(ta, tb);
    return r;
}
__device__ __forceinline__ Ival ival_intersect(Ival s, Ival o) {
    if (s.kind == 0) return o;
    if (s.kind == 2) return s;
    if (o.kind == 0) return s;
    if (o.kind == 2) return o;
    Ival r;
    r.a = fmaxf(s.a, o.a);
    r.b = fminf(s.b, o.b);
    r.kind = (r.a <= r.b) ? 1 : 2;
    return r;
}
__device__ __forceinline__ bool box_exact(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, F3 o, F3 d,
                                       float &tmin, float &tmax) {
    Ival xi = ival_new(mnx, mxx, o.x, d.x);
    Ival yi = ival_new(mny, mxy, o.y, d.y);
    Ival zi = ival_new(mnz, mxz, o.z, d.z);
    Ival t = ival_intersect(xi, ival_intersect(yi, zi));
    if (t.kind == 2) return false;
    if (t.kind == 1) {
        tmin = t.a;
        tmax = t.b;
    } else {  // unreachable!() in the reference
        tmin = -INFINITY;
        tmax = INFINITY;
    }
    return true;
}

// Box decision + pruning.
//
// Exactness argument: the reference tests a primitive iff its leaf box and every ancestor
// box pass AABB::intersect.  A parent box is the min/max union of its children, and the
// slab arithmetic ((a-x)/slope, min, max) is monotone under correct rounding, so a ray that
// passes a box passes every ancestor of it: only the LEAF box decision can change a result.
// Internal boxes are therefore tested conservatively (reciprocal multiply, accept unless
// clearly separated); leaf boxes are decided exactly: the fast path settles clearly
// separated intervals (the reciprocal form is within 2^-22 relative of the division) and
// everything else re-runs the reference's divisions.  Pruning (box behind the origin or
// beyond the best hit) keeps a slack far above the f32 error of t.
// kind: 0 = internal child (conservative), 1 = leaf child / root (exact), 2 = exact, no pruning.
__device__ __forceinline__ bool box_test(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                         const RayCtx &r, float limit_plus, bool exact, bool prune, float &t_entry) {
    float tmin, tmax;
    bool hit;
    if (!r.exact_only) {
        float ax = (mnx - r.o.x) * r.inv.x, bx = (mxx - r.o.x) * r.inv.x;
        float ay = (mny - r.o.y) * r.inv.y, by = (mxy - r.o.y) * r.inv.y;
        float az = (mnz - r.o.z) * r.inv.z, bz = (mxz - r.o.z) * r.inv.z;
        tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        float delta = fmaxf(fabsf(tmin), fabsf(tmax)) * 4.8e-7f;
        bool clear_miss = tmin > tmax + delta;
        bool clear_hit = tmin <= tmax - delta;
        hit = !clear_miss;
        if (exact && !clear_miss && !clear_hit) hit = box_exact(mnx, mny, mnz, mxx, mxy, mxz, r.o, r.d, tmin, tmax);
    } else {
        hit = box_exact(mnx, mny, mnz, mxx, mxy, mxz, r.o, r.d, tmin, tmax);
    }
    t_entry = tmin;
    if (!prune) return hit;
    return hit && !(!r.ill_any() && tmax < -r.slack0) && !(tmin > limit_plus);
}

// Object::get_intersection (object.rs:145-220) against one 64-byte primitive record.
template <bool XFC, bool TRI>
__device__ __forceinline__ bool prim_intersect(const Prim *pp, RayCtx &r, const Xform *xforms, float &t_world) {
    const float4 *q4 = reinterpret_cast<const float4 *>(pp);
    float4 p0 = q4[0], p1 = q4[1], p2 = q4[2], p3 = q4[3];
    uint32_t meta = __float_as_uint(p3.w);
    uint32_t type = meta & 3u, xf = (meta >> 2) & 0x1FFFu;
    const bool identity = (meta & PRIM_IDENTITY_XFORM) != 0u;
    if constexpr (!XFC && TRI) {
        // Every lane of this step at a triangle without a transform -- all but a few steps of a mesh scene: straight
        // code, the reference's nested conditions (object.rs:190-211) as one predicate; q IS position + t direction.
        // (TRI: the leaf-banking kernels, whose scenes are big triangle meshes: -2 % on the 20 M soup; on the headline
        // frame +-0, on a scene of spheres, where the test never passes, +7 %.)
        if (__ballot(!(type == 1u && identity)) == 0ull) {
            const F3 a = f3(p0.x, p0.y, p0.z), b = f3(p0.w, p1.x, p1.y), c = f3(p1.z, p1.w, p2.x);
            const F3 normal = f3(p3.x, p3.y, p3.z);
            const float dn = dot(r.od, normal);
            const float tt = dot(a - r.o, normal) / dn;
            const F3 q = r.o + tt * r.od;
            const float e0 = dot(cross(b - a, q - a), normal);
            const float e1 = dot(cross(c - b, q - b), normal);
            const float e2 = dot(cross(a - c, q - c), normal);
            t_world = dot(q - r.o, r.d);
            return dn < 0.0f && tt > 0.0f && !signbit(e0) && !signbit(e1) && !signbit(e2);
        }
    }
    F3 position, direction;
    if constexpr (XFC) {
        if (xf != r.xf) {
            if (identity) {
                r.oo = r.o;
                r.od = normalize(r.d);
            } else {
                const float *M = xforms[xf].w2o;
                r.oo = transform_point(M, r.o);
                r.od = normalize(transform_vector(M, r.d));
            }
            r.xf = xf;
        }
        position = r.oo;
        direction = r.od;
    } else {
        position = r.o;
        direction = r.od;
        if (!identity) {  // ray.transform_using(world_to_object) (object.rs:146-148, ray.rs:18-23)
            const float *M = xforms[xf].w2o;
            position = transform_point(M, r.o);
            direction = normalize(transform_vector(M, r.d));
        }
    }
    float t;
    bool hit = false;
    if (type == 0u) {  // Sphere: centre p0.xyz, radius p0.w, radius^2 p1.x
        F3 center = f3(p0.x, p0.y, p0.z);
        float tc = dot(center - position, direction);
        F3 closest = position + tc * direction;
        float radius_sqrd = p1.x;
        F3 cd = center - closest;
        float dist_sqrd = dot(cd, cd);
        if (dist_sqrd <= radius_sqrd) {
            float delta = sqrtf(radius_sqrd - dist_sqrd);
            float c0 = tc - delta, c1 = tc + delta;
            bool s0 = !signbit(c0), s1 = !signbit(c1);
            if (s0 && s1) {
                t = (c0 > c1) ? c1 : c0;
                hit = true;
            } else if (s0) {
                t = c0;
                hit = true;
            } else if (s1) {
                t = c1;
                hit = true;
            }
        }
    } else {
        F3 a = f3(p0.x, p0.y, p0.z), b = f3(p0.w, p1.x, p1.y), c = f3(p1.z, p1.w, p2.x);
        F3 normal = f3(p3.x, p3.y, p3.z);
        float dn = dot(direction, normal);
        if (dn < 0.0f) {
            float tt = dot(a - position, normal) / dn;
            if (tt > 0.0f) {
                F3 q = position + tt * direction;
                float e0 = dot(cross(b - a, q - a), normal);
                float e1 = dot(cross(c - b, q - b), normal);
                bool inside = !signbit(e0) && !signbit(e1);
                if (type == 2u) {  // Quad (object.rs:167-189)
                    F3 d = f3(p2.y, p2.z, p2.w);
                    float e2 = dot(cross(d - c, q - c), normal);
                    float e3 = dot(cross(a - d, q - d), normal);
                    inside = inside && !signbit(e2) && !signbit(e3);
                } else {  // Triangle (object.rs:190-211)
                    float e2 = dot(cross(a - c, q - c), normal);
                    inside = inside && !signbit(e2);
                }
                if (inside) {
                    t = tt;
                    hit = true;
                }
            }
        }
    }
    if (!hit) return false;
    F3 osp = position + t * direction;                 // object.rs:214
    F3 wp = identity ? osp : transform_point(xforms[xf].o2w, osp);  // object.rs:216-217
    t_world = dot(wp - r.o, r.d);                      // object.rs:218 / ray.rs:31-33
    return true;
}

// ---------------------------------------------------------------- BVH traversal
// A lane's query is one of two kinds (per lane, at run time):
// closest: closest hit, ties resolved by (t, leaf-order index) = the reference's "first minimum
//          wins" over its left-to-right recursion (bvh.rs:367-371,385-389).
// shadow:  light.rs:81-97 as any-hit: stop at the first hit with t + 1e-4 < dist; otherwise report
//          whether anything was hit at all.  A directional light (light.rs:106-112: any hit at all
//          blocks) is the same query with dist = infinity.
constexpr uint32_t TRAV_DONE = 0xFFFFFFFFu;  // has LEAF_BIT set: ends the descent loop

// Per-lane traversal stack: LDS_STACK entries of (node reference, entry t) in LDS, laid
// out [entry][thread] so a wave's accesses are conflict-free, overflowing to global memory.
// The two pointers carry their address spaces so that the LDS path stays ds_write_b64 /
// ds_read_b64 (with generic pointers the compiler merges both paths into flat accesses).
typedef __attribute__((address_space(3))) unsigned long long *LdsStackPtr;   // entry = ref | t bits << 32
typedef __attribute__((address_space(1))) unsigned long long *GlobalStackPtr;
struct Stack {
    LdsStackPtr lds;        // this thread's column
    GlobalStackPtr spill;   // this thread's column of [MAX_TREE_DEPTH - LDS_STACK][threads]
    uint32_t spill_stride;
    int sp;
    int nlds;               // entries kept in LDS: a constant of the kernel variant (path_body), folded after inlining
    __device__ __forceinline__ void push(uint32_t ref, float t) {
        unsigned long long e = (unsigned long long)ref | ((unsigned long long)__float_as_uint(t) << 32);
        if (sp < nlds) lds[sp * TBLOCK] = e;
        else spill[(size_t)(sp - nlds) * spill_stride] = e;
        sp++;
    }
    // Pops the next entry whose box can still matter; TRAV_DONE when the stack is empty.
    __device__ __forceinline__ uint32_t pop(float limit_plus) {
        while (sp > 0) {
            sp--;
            unsigned long long e;
            if (sp < nlds) e = lds[sp * TBLOCK];
            else e = spill[(size_t)(sp - nlds) * spill_stride];
            if (!(__uint_as_float((uint32_t)(e >> 32)) > limit_plus)) return (uint32_t)e;
        }
        return TRAV_DONE;
    }
};

// Compare-exchange of two (entry t, reference) pairs: afterwards ta <= tb.
__device__ __forceinline__ void cswap(float &ta, uint32_t &ra, float &tb, uint32_t &rb) {
    bool sw = tb < ta;
    float tlo = sw ? tb : ta, thi = sw ? ta : tb;
    uint32_t rlo = sw ? rb : ra, rhi = sw ? ra : rb;
    ta = tlo;
    tb = thi;
    ra = rlo;
    rb = rhi;
}

// Reciprocal slab test of one box (not for exact_only rays): the entry t that pruning may use, the
// exit t, and the two numbers that decide the box.  (plane - o) * (1/d) is within 3 u of the
// reference's (plane - o) / d (u = 2^-24: the rounded reciprocal, the product, the division), the two
// parameters that decide a box therefore within 6 u max(|tmin|, |tmax|) together; delta = 8 u of it.
// diff = tmin - tmax (tmin = entry t over all three axes): the box is clearly missed when
// diff > delta, clearly hit when diff < -delta; anything else is within the error of the
// reference's division form and a LEAF box is then decided by box_exact.  Empty slots hold NaN
// boxes: diff is NaN and every comparison with it false.
// tprune: the entry t over the well-conditioned axes only (<= tmin): cx, cy, cz are +inf for
// those and -inf for the others (RayCtx::cap_*).
__device__ __forceinline__ void slab_fast(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                          const RayCtx &r, float cx, float cy, float cz, float &tprune, float &tmax,
                                          float &diff, float &delta) {
    float ax = (mnx - r.o.x) * r.inv.x, bx = (mxx - r.o.x) * r.inv.x;
    float ay = (mny - r.o.y) * r.inv.y, by = (mxy - r.o.y) * r.inv.y;
    float az = (mnz - r.o.z) * r.inv.z, bz = (mxz - r.o.z) * r.inv.z;
    float nx = fminf(ax, bx), ny = fminf(ay, by), nz = fminf(az, bz);
    float tmin = fmaxf(fmaxf(nx, ny), nz);
    tprune = fmaxf(fmaxf(fminf(nx, cx), fminf(ny, cy)), fminf(nz, cz));
    tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    diff = tmin - tmax;
    delta = fmaxf(fabsf(tmin), fabsf(tmax)) * 4.8e-7f;
}

// Traversal state of one lane.  trav_begin() starts a ray, trav_step() advances it by one
// wide node or one leaf; cur == TRAV_DONE when the ray has its answer.
// REF = false: production traversal over the wide nodes (ordered, pruned).
// REF = true:  bvh.rs:359-395 verbatim over the binary nodes (both children, no pruning).
struct Trav {
    RayCtx r;
    uint32_t cur;
    uint32_t bank;          // a leaf set aside while the wave ran node steps (wave_step), or TRAV_DONE
    __device__ __forceinline__ bool done() const { return (cur & bank) == TRAV_DONE; }
    float limit_plus;
    uint32_t prim;
    float t;                // closest: best hit so far; shadow: distance light -> shaded point
    bool any, occluded;
    uint32_t nodes, prims;  // visit counters of the lane's current ray (COUNT)
};

template <bool REF, bool OVH>
__device__ __forceinline__ void trav_begin(const DeviceScene &sc, Trav &s, Stack &st) {
    s.prim = NO_PRIM;
    s.any = false;
    s.occluded = false;
    s.limit_plus = INFINITY;
    s.cur = TRAV_DONE;
    s.bank = TRAV_DONE;
    st.sp = 0;
    uint32_t root = REF ? sc.root_ref : sc.wroot_ref;
    if (root == NO_PRIM) return;
    if (!REF && !(root & LEAF_BIT)) {
        // The reference tests the root's own box first (bvh.rs:362).  A ray that misses it misses every
        // box inside it (see box_test), so the production traversal leaves that to the root node's slots
        // and saves the test -- divisions and all -- at every ray start; a root LEAF keeps it (exact).
        s.cur = root;
        return;
    }
    float te;
    // (no pruning of the root box in scenes whose hits can lie outside their boxes)
    if (box_test(sc.root_mn[0], sc.root_mn[1], sc.root_mn[2], sc.root_mx[0], sc.root_mx[1], sc.root_mx[2], s.r,
                 s.limit_plus, true, !REF && !OVH, te))
        s.cur = root;
}

// Lane-utilisation census of the scheduling steps (diagnostic build -DRTX_UTIL_STATS only;
// tools/util_probe.py prints it): wave steps and executing lanes per kind.
struct Util {
    uint32_t node_steps = 0, prim_steps = 0, node_lanes = 0, prim_lanes = 0, done_lanes = 0;
    uint32_t events = 0, shade_events = 0, shade_lanes = 0, start_lanes = 0, fin_lanes = 0;  // path kernel: lanes-between-rays events
    uint32_t exact_steps = 0, exact_lanes = 0;  // node steps that ran the division path, and the lanes that needed it
    unsigned long long step_ticks = 0;  // 100 MHz ticks inside the stepping loop (the rest of a wave's life is events)
    unsigned long long sec_ticks[4] = {0, 0, 0, 0};  // ... and inside an event: finished rays, shading pass, stack / queue, ray begin
    __device__ __forceinline__ void flush(uint32_t *ctr) {
#ifdef RTX_UTIL_STATS
        if (lane_id() == 0) {
            unsigned long long *c = reinterpret_cast<unsigned long long *>(ctr + CTR_UTIL);
            atomicAdd(c + 0, (unsigned long long)node_steps);
            atomicAdd(c + 1, (unsigned long long)prim_steps);
            atomicAdd(c + 2, (unsigned long long)node_lanes);
            atomicAdd(c + 3, (unsigned long long)prim_lanes);
            atomicAdd(c + 4, (unsigned long long)done_lanes);
            atomicAdd(c + 5, (unsigned long long)events);
            atomicAdd(c + 6, (unsigned long long)shade_events);
            atomicAdd(c + 7, (unsigned long long)shade_lanes);
            atomicAdd(c + 8, (unsigned long long)start_lanes);
            atomicAdd(c + 9, (unsigned long long)fin_lanes);
            atomicAdd(c + 15, step_ticks);
            for (int k = 0; k < 4; k++) atomicAdd(c + 16 + k, sec_ticks[k]);
            atomicAdd(c + 20, (unsigned long long)exact_steps);
            atomicAdd(c + 21, (unsigned long long)exact_lanes);
        }
#endif
    }
};

// Advances a lane that holds an internal node by one node.
template <bool COUNT, bool REF, bool OVH>
__device__ __forceinline__ void node_step(const DeviceScene &sc, Trav &s, Stack &st, LdsNodePtr lds_top, Util &u) {
    uint32_t cur = s.cur;
    if (COUNT) s.nodes++;
    if constexpr (!REF) {
        float4 mnx, mny, mnz, mxx, mxy, mxz, rf4;
        if (TOP_NODES != 0u && cur < TOP_NODES) {
            LdsNodePtr lp = lds_top + cur;
            v4f a = lp[0 * TOP_NODES], b = lp[1 * TOP_NODES], c = lp[2 * TOP_NODES], d = lp[3 * TOP_NODES];
            v4f e = lp[4 * TOP_NODES], f = lp[5 * TOP_NODES], g = lp[6 * TOP_NODES];
            mnx = make_float4(a.x, a.y, a.z, a.w);
            mny = make_float4(b.x, b.y, b.z, b.w);
            mnz = make_float4(c.x, c.y, c.z, c.w);
            mxx = make_float4(d.x, d.y, d.z, d.w);
            mxy = make_float4(e.x, e.y, e.z, e.w);
            mxz = make_float4(f.x, f.y, f.z, f.w);
            rf4 = make_float4(g.x, g.y, g.z, g.w);
        } else {
            const float4 *np = reinterpret_cast<const float4 *>(sc.wnodes + cur);
            mnx = np[0];
            mny = np[1];
            mnz = np[2];
            mxx = np[3];
            mxy = np[4];
            mxz = np[5];
            rf4 = np[6];
        }
        uint32_t r0 = __float_as_uint(rf4.x), r1 = __float_as_uint(rf4.y), r2 = __float_as_uint(rf4.z),
                 r3 = __float_as_uint(rf4.w);
        // per slot: entry t / exit t, and whether the box is hit (inside the error band: yes, for now)
        float t0, t1, t2, t3, x0, x1, x2, x3;
        bool k0, k1, k2, k3, marginal;
        {
            float f0, f1, f2, f3, d0, d1, d2, d3;
            const float cx = s.r.cap_x(), cy = s.r.cap_y(), cz = s.r.cap_z();
            slab_fast(mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x, s.r, cx, cy, cz, t0, x0, f0, d0);
            slab_fast(mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y, s.r, cx, cy, cz, t1, x1, f1, d1);
            slab_fast(mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z, s.r, cx, cy, cz, t2, x2, f2, d2);
            slab_fast(mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w, s.r, cx, cy, cz, t3, x3, f3, d3);
            k0 = f0 <= d0;
            k1 = f1 <= d1;
            k2 = f2 <= d2;
            k3 = f3 <= d3;
            marginal = fabsf(f0) <= d0 || fabsf(f1) <= d1 || fabsf(f2) <= d2 || fabsf(f3) <= d3;
        }
        // leaf slots inside the error band (and every slot of an exact_only ray) re-run the
        // reference's divisions; rare, so it sits behind ONE wave-uniform branch and reloads
        // the slot's box instead of keeping it live
        const bool eo = s.r.exact_only;
        if (__ballot(eo | marginal) != 0ull) {
#ifdef RTX_UTIL_STATS
            u.exact_steps++;
            u.exact_lanes += (uint32_t)__popcll(__ballot(eo | marginal));
#endif
#pragma unroll 1
            for (uint32_t k = 0; k < 4u; k++) {
                const float *nb = reinterpret_cast<const float *>(sc.wnodes + cur) + k;
                const uint32_t rk = reinterpret_cast<const uint32_t *>(sc.wnodes + cur)[24 + k];
                const float bx0 = nb[0], by0 = nb[4], bz0 = nb[8], bx1 = nb[12], by1 = nb[16], bz1 = nb[20];
                float tk, xk, fk, dk;
                slab_fast(bx0, by0, bz0, bx1, by1, bz1, s.r, INFINITY, INFINITY, INFINITY, tk, xk, fk, dk);
                if ((eo & (rk != WIDE_EMPTY)) | (((int32_t)rk < -1) & (fabsf(fk) <= dk))) {
                    float a, b;
                    const bool h = box_exact(bx0, by0, bz0, bx1, by1, bz1, s.r.o, s.r.d, a, b);
                    a = fminf(a, 3.0e38f);  // (a box that is hit keeps a finite key in the sort below)
                    if (s.r.ill_any()) a = -INFINITY;  // the exact entry t is as ill-conditioned as the fast one
                    if (k == 0u) { k0 = h; t0 = a; x0 = b; }
                    if (k == 1u) { k1 = h; t1 = a; x1 = b; }
                    if (k == 2u) { k2 = h; t2 = a; x2 = b; }
                    if (k == 3u) { k3 = h; t3 = a; x3 = b; }
                } else if (eo) {  // an empty slot: the NaN box of an exact_only ray decides nothing
                    if (k == 0u) k0 = false;
                    if (k == 1u) k1 = false;
                    if (k == 2u) k2 = false;
                    if (k == 3u) k3 = false;
                }
            }
        }
        // A hit can lie up to ovh[k] outside its object's box in space (non-planar quads, sheared
        // spheres: flatten.cpp, hit_overhang; folded up the tree per slot), which is
        // ovh * |1/d_i| in t on the axis that decides the box's entry or exit.  OVH: only the
        // kernels launched for scenes with an overhang carry this code; the node's line is in
        // the L1 by now.
        if constexpr (OVH) {
            const float4 ov = reinterpret_cast<const float4 *>(sc.wnodes + cur)[7];
            const float gx = fabsf(s.r.inv.x), gy = fabsf(s.r.inv.y), gz = fabsf(s.r.inv.z);
            const float g = fmaxf(fmaxf(gx <= ILL_INV ? gx : 0.0f, gy <= ILL_INV ? gy : 0.0f), gz <= ILL_INV ? gz : 0.0f);
            t0 -= ov.x * g;
            x0 += ov.x * g;
            t1 -= ov.y * g;
            x1 += ov.y * g;
            t2 -= ov.z * g;
            x2 += ov.z * g;
            t3 -= ov.w * g;
            x3 += ov.w * g;
        }
        // pruning: box behind the origin or beyond the best hit
        const float back = s.r.ill_any() ? -INFINITY : -s.r.slack0, lim = s.limit_plus;
        const bool h0 = k0 && !(x0 < back) && !(t0 > lim);
        const bool h1 = k1 && !(x1 < back) && !(t1 > lim);
        const bool h2 = k2 && !(x2 < back) && !(t2 > lim);
        const bool h3 = k3 && !(x3 < back) && !(t3 > lim);
        t0 = h0 ? t0 : INFINITY;
        t1 = h1 ? t1 : INFINITY;
        t2 = h2 ? t2 : INFINITY;
        t3 = h3 ? t3 : INFINITY;
        r0 = h0 ? r0 : WIDE_EMPTY;
        r1 = h1 ? r1 : WIDE_EMPTY;
        r2 = h2 ? r2 : WIDE_EMPTY;
        r3 = h3 ? r3 : WIDE_EMPTY;
        // sort the four slots by entry t (rejected slots last): the nearest is continued,
        // the others are pushed farthest first
        cswap(t0, r0, t1, r1);
        cswap(t2, r2, t3, r3);
        cswap(t0, r0, t2, r2);
        cswap(t1, r1, t3, r3);
        cswap(t1, r1, t2, r2);
        if (r1 != WIDE_EMPTY) {
            if (r2 != WIDE_EMPTY) {
                if (r3 != WIDE_EMPTY) st.push(r3, t3);
                st.push(r2, t2);
            }
            st.push(r1, t1);
        }
        s.cur = (r0 != WIDE_EMPTY) ? r0 : st.pop(s.limit_plus);
    } else {
        const float4 *np = reinterpret_cast<const float4 *>(sc.nodes + cur);
        float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        uint32_t left = __float_as_uint(n3.x), right = __float_as_uint(n3.y);
        float tl, tr;
        bool hl = box_test(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, s.r, s.limit_plus, true, false, tl);
        bool hr = box_test(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, s.r, s.limit_plus, true, false, tr);
        if (hl && hr) {
            st.push(right, -INFINITY);
            s.cur = left;
        } else if (hl) {
            s.cur = left;
        } else if (hr) {
            s.cur = right;
        } else {
            s.cur = st.pop(s.limit_plus);
        }
    }
}

// Advances a lane that holds a leaf by ONE primitive (Vec order, bvh.rs:378-389): the leaf
// reference (first << 3 | count) shrinks from the front until it is used up.  The banked leaf
// goes first (it was reached first); only the leaf in `cur` is followed by a pop.
template <bool COUNT, bool REF, bool XFC, bool TRI>
__device__ __forceinline__ void prim_step(const DeviceScene &sc, Trav &s, Stack &st, bool shadow) {
    const bool banked = s.bank != TRAV_DONE;
    uint32_t cur = banked ? s.bank : s.cur;
    uint32_t first = (cur & 0x7FFFFFFFu) >> 3;
    float tw;
    if (COUNT) s.prims++;
    if (prim_intersect<XFC, TRI>(sc.prims + first, s.r, sc.xforms, tw)) {
        if (!shadow) {
            if (!s.any || tw < s.t || (tw == s.t && first < s.prim)) {
                s.any = true;
                s.t = tw;
                s.prim = first;
                if (!REF) s.limit_plus = tw + (s.r.slack0 + 1.5e-5f * fabsf(tw));
            }
        } else {
            const float dist = s.t;
            s.any = true;
            if (tw + 1e-4f < dist) {
                s.occluded = true;
                if (!REF) {  // (the verbatim traversal goes on: the reference asks for the closest hit, light.rs:90)
                    s.cur = s.bank = TRAV_DONE;
                    return;
                }
            }
            if (!REF) s.limit_plus = dist + (s.r.slack0 + 1.5e-5f * fabsf(dist));
        }
    }
    const bool more = (cur & 7u) > 1u;  // then: first + 1, count - 1
    if (banked) s.bank = more ? cur + 7u : TRAV_DONE;
    else s.cur = more ? cur + 7u : st.pop(s.limit_plus);
}

// One scheduling step of a wave: lanes are either at an internal node or at a primitive of
// a leaf; executing both kinds every iteration would run each at a fraction of the lanes.
// The wave executes ONE kind per step, the one that advances more work (a primitive test
// counts RTX_LEAF_WEIGHT times a node visit, it is the shorter code), the other lanes wait.
#ifndef RTX_LEAF_WEIGHT
#define RTX_LEAF_WEIGHT 1
#endif
#ifndef RTX_NODE_WEIGHT
#define RTX_NODE_WEIGHT 1
#endif
// BANK (leaf banking, below) is a kernel variant of its own: it pays where rays are long and the
// kernel is bound by VALU issue (a 20M-triangle soup: -9 % of the frame time), and costs 3-6 % where
// rays are short -- as a run-time switch inside one kernel it cost the short-ray scenes the same
// (registers and code in the stepping loop, not the extra visits).  The host picks per scene.
template <bool COUNT, bool REF, bool OVH, bool BANK, bool XFC>
__device__ __forceinline__ void wave_step(const DeviceScene &sc, Trav &s, Stack &st, bool shadow, LdsNodePtr lds_top,
                                          Util &u) {
    const bool cur_leaf = (s.cur & LEAF_BIT) && s.cur != TRAV_DONE;
    const bool at_leaf = cur_leaf || s.bank != TRAV_DONE;
    // A lane at a leaf need not sit out a node step: it sets the leaf aside (one per lane) and goes
    // on with the next entry of its stack.  The order of the visits changes nothing in the answer
    // (closest hit: the minimum over (t, leaf-order index); shadow query: two flags), only what
    // the limit prunes; the lane would have waited otherwise.
    const bool bankable = BANK && cur_leaf && s.bank == TRAV_DONE && st.sp > 0;
    const bool at_node = !(s.cur & LEAF_BIT) || bankable;
    uint32_t nl = (uint32_t)__popcll(__ballot(at_leaf)), nn = (uint32_t)__popcll(__ballot(at_node));
#ifdef RTX_UTIL_STATS
    u.done_lanes += (uint32_t)__popcll(__ballot(s.done()));
#endif
    if (nl * (uint32_t)RTX_LEAF_WEIGHT >= nn * (uint32_t)RTX_NODE_WEIGHT) {
#ifdef RTX_UTIL_STATS
        u.prim_steps++;
        u.prim_lanes += nl;
#endif
        if (at_leaf) prim_step<COUNT, REF, XFC, BANK>(sc, s, st, shadow);
    } else {
        if (bankable) {
            s.bank = s.cur;
            s.cur = st.pop(s.limit_plus);
        }
        const bool run = !(s.cur & LEAF_BIT);
#ifdef RTX_UTIL_STATS
        u.node_steps++;
        u.node_lanes += (uint32_t)__popcll(__ballot(run));
#endif
        if (run) node_step<COUNT, REF, OVH>(sc, s, st, lds_top, u);
    }
}

// Work distribution of the persistent trace kernels.  A wave owns a private run of
// consecutive queue entries [next, end) taken from the global queue head with ONE atomic
// (a single hot counter sustains only ~90 atomics/us), and hands them to its lanes one by
// one: as soon as REFILL_MIN lanes have finished their ray they get new rays, so lanes do
// not idle until the longest ray of a 64-ray batch is done.
#ifndef RTX_REFILL_MIN
#define RTX_REFILL_MIN 32
#endif
// Queue entries a wave reserves with one atomic.  An entry is a primary sample with its whole path
// tree (about 4.6 rays on the headline frame), walked by that wave alone: long runs in a costly
// part of the image (silhouettes: rays of hundreds of steps) make stragglers.
#ifndef RTX_CHUNK_MIN
#define RTX_CHUNK_MIN 32
#endif
#ifndef RTX_CHUNK_MAX
#define RTX_CHUNK_MAX 64
#endif
struct WaveQueue {
    uint32_t *heads;
    uint32_t count, chunk, stripe, nstripes, nwaves, p0, tried;
    uint32_t next, end;
    bool exhausted;
    // The queue of primaries is cut into stripes of `stripe` entries -- one 8x8 pixel block with all its samples
    // (raygen) -- and stripe k belongs to sub-queue k % QPARTS: every sub-queue is a comb over the whole frame, all
    // sub-queues cost the same and run dry together, and all of them are at the same place of the frame at the
    // same time: the whole chip works on the same few tiles, so what one wave brings into the L2s and the
    // Infinity Cache is there for the others.  (Measured, round 3, headline frame: stripes of 4 / 16 / 64 blocks
    // 8.5 / 9.3 / 9.0 ms against 8.3; sub-queues as bands of the frame 9.3 ms; and the sub-queues dealt to the
    // XCDs -- blocks b and b + 8 share one -- so that each XCD's L2 holds an eighth of what the frame sees: L2
    // hit rate 0.55 -> 0.73, fabric reads -45 %, and the kernel 15 % SLOWER, its waves waiting longer for their
    // loads than before.)
    // A position in a sub-queue is a cursor c: stripe c / stripe of the comb, offset c % stripe in it.  A
    // sub-queue's first entries are dealt statically (wave j of the sub-queue owns chunk j: no atomic at all for
    // queues that small); the rest is handed out through the sub-queue's head counter (one hot counter sustains
    // only ~90 atomics/us).  A chunk is 32 or 64 entries and divides the stripe, so a chunk never leaves its stripe.
    // Waves whose first sub-queue is p: wave j * QPARTS + p for j * QPARTS + p < nwaves.
    __device__ __forceinline__ uint32_t waves_of(uint32_t p) const {
        return nwaves > p ? (nwaves - p + (uint32_t)QPARTS - 1u) / (uint32_t)QPARTS : 0u;
    }
    // Statically dealt entries of sub-queue p: one chunk per wave that exists in it -- a chunk without an owner
    // would never be traced.
    __device__ __forceinline__ uint32_t dyn0_of(uint32_t p) const { return waves_of(p) * chunk; }
    // entries of sub-queue p: its stripes p, p + QPARTS, ...; the queue's last stripe may be short
    __device__ __forceinline__ uint32_t len_of(uint32_t p) const {
        if (p >= nstripes) return 0u;
        const uint32_t k = (nstripes - p + (uint32_t)QPARTS - 1u) / (uint32_t)QPARTS;
        const uint32_t cut = (nstripes - 1u) % (uint32_t)QPARTS == p ? nstripes * stripe - count : 0u;
        return k * stripe - cut;
    }
    // the run [c, c + chunk) of sub-queue p (clipped to its length `len`), as queue entries [next, end)
    __device__ __forceinline__ void set_run(uint32_t p, uint32_t c, uint32_t len) {
        const uint32_t si = c / stripe, n = len - c < chunk ? len - c : chunk;
        next = (si * (uint32_t)QPARTS + p) * stripe + (c - si * stripe);
        end = next + n;
    }
    __device__ __forceinline__ void init(uint32_t *h, uint32_t n, uint32_t stripe_entries, uint32_t nw, uint32_t wave) {
        heads = h;
        count = n;
        nwaves = nw;
        stripe = stripe_entries >= 64u ? stripe_entries & ~63u : 64u;
        nstripes = (n + stripe - 1u) / stripe;
#ifdef RTX_CHUNK_FIXED
        chunk = (uint32_t)RTX_CHUNK_FIXED;
#else
#ifdef RTX_CHUNK_FIXED
        chunk = (uint32_t)RTX_CHUNK_FIXED;   // (measured, round 3: 16 / 32 entries per chunk cost the headline frame 28 / 10 %)
#else
        chunk = n / (nwaves * 4u) < 48u ? 32u : 64u;
#endif
#endif
        p0 = wave % (uint32_t)QPARTS;
        tried = 0;
        exhausted = n == 0;
        next = end = 0;
        const uint32_t c = (wave / (uint32_t)QPARTS) * chunk, len = len_of(p0);
        if (c < len) set_run(p0, c, len);
    }
    // Gives every idle lane (ballot `idle`) its queue index; returns false for lanes left empty.
    __device__ __forceinline__ bool take(unsigned long long idle, bool lane_idle, uint32_t &index) {
        while (next >= end) {  // own run used up: this sub-queue's counter, then the others'
            if (tried >= (uint32_t)QPARTS) {
                exhausted = true;
                next = end = 0;
                return false;
            }
            const uint32_t p = (p0 + tried) % (uint32_t)QPARTS;
            const uint32_t len = len_of(p), dyn0 = dyn0_of(p);
            const uint32_t dyn = len > dyn0 ? len - dyn0 : 0u;
            uint32_t *hp = heads + p * (uint32_t)QHEAD_STRIDE;
            uint32_t b = dyn;
            if (dyn != 0u) {
                // look before taking: an exhausted counter costs a load, not a serialised atomic
                uint32_t peek = __builtin_amdgcn_readfirstlane(__hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (peek < dyn) b = wave_fetch(hp, chunk);
            }
            if (b >= dyn) {
                tried++;
                continue;
            }
            set_run(p, dyn0 + b, len);
        }
        uint32_t avail = end - next, nidle = (uint32_t)__popcll(idle);
        uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane_id()) - 1ull));
        index = next + rank;
        bool got = lane_idle && rank < avail;
        next += nidle < avail ? nidle : avail;
        return got;
    }
};

__device__ __forceinline__ void add_visits(uint32_t *ctr, uint32_t nodes, uint32_t prims, int first) {
    // wave reduction, then one 64-bit atomic pair per wave
    for (int d = 32; d > 0; d >>= 1) {
        nodes += __shfl_down(nodes, d, 64);
        prims += __shfl_down(prims, d, 64);
    }
    if (lane_id() == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(ctr + first), (unsigned long long)nodes);
        atomicAdd(reinterpret_cast<unsigned long long *>(ctr + first + 2), (unsigned long long)prims);
    }
}

// ---------------------------------------------------------------- raygen
__device__ __forceinline__ bool slot_to_pixel(const BatchInfo &bi, uint32_t width, uint32_t height, uint32_t lp,
                                              uint32_t &px, uint32_t &py, uint32_t &tile_off) {
    uint32_t T = bi.tile, tt = T * T;
    uint32_t lt = lp / tt, rr = lp % tt;
    uint32_t blk = rr >> 6, in = rr & 63u, bpr = T >> 3;
    uint32_t x_in = (blk % bpr) * 8u + (in & 7u), y_in = (blk / bpr) * 8u + (in >> 3);
    uint32_t gt = lt * bi.nranks + bi.rank;
    tile_off = lt * tt + y_in * T + x_in;
    if (gt >= bi.tiles_x * bi.tiles_y) return false;
    px = (gt % bi.tiles_x) * T + x_in;
    py = (gt / bi.tiles_x) * T + y_in;
    return px < width && py < height;
}

__global__ void __launch_bounds__(BLOCK) raygen_kernel(DeviceScene sc, Frame fr, BatchInfo bi) {
    uint32_t n = bi.npix * bi.ns;
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) fr.counters[CTR_LEVEL_COUNT] = n;
    if (i >= n) return;
    // path node i = ((pixel block * ns) + sample) * 64 + slot in block: the samples of one 8x8 pixel block are
    // neighbours in the queue of primaries: consecutive runs of a sub-queue are the same pixels again (WaveQueue),
    // and the resolve pass finds a pixel's samples 64 nodes apart
    const uint32_t blk = (i >> 6) / bi.ns, sl = (i >> 6) % bi.ns, q = blk * 64u + (i & 63u);
    uint32_t px, py, toff;
    bool valid = slot_to_pixel(bi, sc.width, sc.height, bi.q0 + q, px, py, toff);
    if (!valid) {
        fr.ray_o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        fr.ray_d[i] = make_float4(0.f, 0.f, 0.f, __uint_as_float(INVALID_PRIM));
        return;
    }
    // Camera::generate_ray (camera.rs:32-47)
    float dx = 0.0f, dy = 0.0f;
    if (bi.use_jitter) {
        uint32_t s = bi.s0 + sl;
        dx = ((float)(jitter_u32(bi.seed_lo, bi.seed_hi, px, py, s, 0) >> 8) * (1.0f / 16777216.0f)) / 2.0f;
        dy = ((float)(jitter_u32(bi.seed_lo, bi.seed_hi, px, py, s, 1) >> 8) * (1.0f / 16777216.0f)) / 2.0f;
    }
    float x = ((float)px + dx) / (float)sc.width - 0.5f;
    float y = 0.5f - ((float)py + dy) / (float)sc.height;
    F3 pos = f3(x, y, -1.0f);
    F3 dir = normalize(f3(x, y, -1.0f));
    F3 wo = transform_point(sc.c2w, pos);
    F3 wd = normalize(transform_vector(sc.c2w, dir));
    fr.ray_o[i] = make_float4(wo.x, wo.y, wo.z, 0.0f);
    fr.ray_d[i] = make_float4(wd.x, wd.y, wd.z, __uint_as_float(NO_PRIM));
}

__global__ void __launch_bounds__(BLOCK) load_rays_kernel(Frame fr, const float *org_dir, uint32_t n) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) fr.counters[CTR_LEVEL_COUNT] = n;
    if (i >= n) return;
    const float *p = org_dir + (size_t)i * 6;
    F3 d = normalize(f3(p[3], p[4], p[5]));  // Ray::new (ray.rs:11-16)
    fr.ray_o[i] = make_float4(p[0], p[1], p[2], 0.0f);
    fr.ray_d[i] = make_float4(d.x, d.y, d.z, __uint_as_float(NO_PRIM));
}

// ---------------------------------------------------------------- object.rs helpers for shading
struct PrimView {
    float4 p0, p1, p2, p3;
    uint32_t type, xf, mat;
};
__device__ __forceinline__ PrimView load_prim(const Prim *pp) {
    const float4 *q4 = reinterpret_cast<const float4 *>(pp);
    PrimView v;
    v.p0 = q4[0];
    v.p1 = q4[1];
    v.p2 = q4[2];
    v.p3 = q4[3];
    uint32_t meta = __float_as_uint(v.p3.w);
    v.type = meta & 3u;
    v.xf = (meta >> 2) & 0x1FFFu;
    v.mat = meta >> 16;
    return v;
}
// Object::get_normal (object.rs:225-235)
__device__ __forceinline__ F3 object_normal(const PrimView &v, const Xform *xforms, F3 point) {
    F3 n;
    if (v.type == 0u) {
        F3 p = transform_point(xforms[v.xf].w2o, point);
        n = normalize(p - f3(v.p0.x, v.p0.y, v.p0.z));
    } else {
        n = f3(v.p3.x, v.p3.y, v.p3.z);
    }
    return normalize(transform_vector(xforms[v.xf].o2w, n));
}
// Object::get_uv (object.rs:255-278)
__device__ __forceinline__ void object_uv(const PrimView &v, const Xform *xforms, F3 point, float &u, float &w) {
    F3 p = transform_point(xforms[v.xf].w2o, point);
    if (v.type == 0u) {
        F3 n = normalize(p - f3(v.p0.x, v.p0.y, v.p0.z));
        float theta = acosf(n.z);
        float phi = atan2f(n.y, n.x);
        u = theta / (2.0f * PI_F);
        w = phi / PI_F;
    } else {
        F3 a = f3(v.p0.x, v.p0.y, v.p0.z), b = f3(v.p0.w, v.p1.x, v.p1.y);
        F3 e = (v.type == 2u) ? f3(v.p2.y, v.p2.z, v.p2.w) : f3(v.p1.z, v.p1.w, v.p2.x);
        F3 ba = b - a, ea = e - a;
        u = dot(ba, p - a) / dot(ba, ba);
        w = dot(ea, p - a) / dot(ea, ea);
    }
}

// ---------------------------------------------------------------- path
// World::trace_ray (world.rs:115-131) for the batch's primary samples in ONE persistent launch.
//
// A wave takes primary samples from the batch's queue and owns everything that follows from
// them: when a closest-hit query ends, its lane -- with the hit point in hand -- writes the
// node's record, spawns the bounce rays of the hit's material (material.rs:185-203) as child path
// nodes, and lists one shadow query per light that wants one (ambient lights are lit without a
// query, cone lights only inside their cone, light.rs:99-122).  Children and shadow queries go on
// the WAVE'S OWN work stack (global memory, private to the wave: no atomics, no other wave ever
// waits for it); idle lanes are refilled from the top of that stack first and from the batch's
// queue of primaries when it is empty.  Entries of one kind pushed together are popped together,
// so the lanes refilled in one go trace the shadow rays of neighbouring hit points towards the
// SAME light (or neighbouring primaries, or the bounce rays of neighbouring hits): their first
// node visits are the same addresses.  The stack is LIFO, so a sample's path tree is walked
// depth-first and the stack stays small.  A shadow query's answer is OR-ed into the lit mask of
// its node's record; the colours are resolved bottom-up by the resolve passes afterwards.  There
// is no pass between bounce levels and no global shadow queue: a launch ends with ONE drain.
//
// Work entry / lane tag: x = path node, y = bits 4..0 light, bit 5 light is directional, bit 6
// shadow query (else closest hit), bit 7 (lane tag only) a ray is in flight, bits 11..8 bounce level.
// the cost of one pass between rays in traversal steps, and the range of the waiting-lane threshold
#ifndef RTX_EVENT_STEPS
#define RTX_EVENT_STEPS 8
#endif
#ifndef RTX_REFILL_LO
#define RTX_REFILL_LO 8
#endif
#ifndef RTX_REFILL_HI
#define RTX_REFILL_HI 48
#endif
#ifndef RTX_SHADE_MIN
#define RTX_SHADE_MIN 32
#endif
#ifndef RTX_CHAIN_PRIO
#define RTX_CHAIN_PRIO 1
#endif
constexpr uint32_t WK_LIGHT = 31u, WK_DIR = 1u << 5, WK_SHADOW = 1u << 6, WK_BUSY = 1u << 7, WK_LEVEL_SHIFT = 8;

// light.rs:113-116: is `point` outside the cone?  The reference compares atan2(|a x b|, a . b)
// with the cut-off; the signed squared cosine settles every point that is not within 1e-4 of the
// cut-off (both sides of the comparison carry a few 1e-7 of rounding), the rest -- behind one
// wave-uniform branch -- runs the reference's expression.
__device__ __forceinline__ bool outside_cone(const LightRec &L, F3 point) {
    const F3 a = f3(L.dir[0], L.dir[1], L.dir[2]);
    const F3 b = point - f3(L.pos[0], L.pos[1], L.pos[2]);
    const float dt = dot(a, b), l2 = dot(a, a) * dot(b, b);
    const float q = dt * fabsf(dt), cc = (L.cos_cut * fabsf(L.cos_cut)) * l2, m = 1e-4f * l2;
    bool outside = q < cc - m;
    const bool marginal = !outside && !(q > cc + m);
    if (__ballot(marginal) != 0ull) {
        if (marginal) outside = angle_rad(a, b) > L.angle_deg * DEG_TO_RAD;
    }
    return outside;
}

#ifdef RTX_RAY_HIST
// Diagnostic (-DRTX_RAY_HIST, COUNT launches, RTX_HIST_PRINT=1): records visited per ray, in powers of two, by
// class -- 0 primary, 1 bounce, 2 point / cone shadow query, 3 directional shadow query, +4 for rays with an
// ill-conditioned axis -- behind the counter block: [class][16] ray counts, then [class][16] record sums.
__device__ __forceinline__ void ray_hist(uint32_t *ctr, uint32_t cls, uint32_t records) {
    const uint32_t b = records ? min(15u, 32u - (uint32_t)__clz(records)) : 0u;
    atomicAdd(ctr + CTR_WORDS + cls * 16u + b, 1u);
    atomicAdd(ctr + CTR_WORDS + 128u + cls * 16u + b, records);
}
#endif
template <bool COUNT, bool REF, bool OVH, bool SHADE, bool BANK, bool PRE, bool XFC>
__device__ __forceinline__ void path_body(const DeviceScene &sc, const Frame &fr, uint32_t max_levels) {
    // The block's 30,464 bytes of LDS (five blocks per CU), two ways.  Scenes of long rays (BANK: >= 8 M primitives,
    // trees 30 levels deep): LDS_STACK_DEEP traversal-stack entries per lane, the hits that wait for a shading pass
    // listed in global memory behind the wave's work stack.  All others: LDS_STACK entries per lane and that list in
    // LDS -- one round trip less per shading pass (headline -0.7 %, Suzanne -1 %; the 20 M soup +3 % with the
    // shorter stack, hence the split).
    constexpr bool SHADE_IN_LDS = RTX_SHADE_IN_LDS && !BANK;
    constexpr int NLDS = SHADE_IN_LDS ? LDS_STACK : LDS_STACK_DEEP;
    __shared__ unsigned long long lds_stack[NLDS * TBLOCK];
    __shared__ unsigned long long lds_shade[SHADE_IN_LDS ? (TBLOCK / 64) * SHADE_LDS : 1];
    __shared__ v4f lds_top_store[TOP_NODES ? TOP_NODES * 7 : 1];
    const LdsNodePtr lds_top = (LdsNodePtr)lds_top_store;
    if (TOP_NODES != 0u && !REF) {
        for (uint32_t i = threadIdx.x; i < TOP_NODES * 7u; i += TBLOCK) {
            uint32_t piece = i / (TOP_NODES ? TOP_NODES : 1u), n = i % (TOP_NODES ? TOP_NODES : 1u);
            lds_top_store[i] = n < sc.n_wnodes ? reinterpret_cast<const v4f *>(sc.wnodes + n)[piece] : v4f{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
    }
    const uint32_t n0 = fr.counters[CTR_LEVEL_COUNT];
    const uint32_t nthreads = gridDim.x * TBLOCK;
    const uint32_t gtid = blockIdx.x * TBLOCK + threadIdx.x;
    const uint32_t lane = lane_id();
    Stack st;
    st.lds = (LdsStackPtr)(lds_stack + threadIdx.x);
    st.spill = (GlobalStackPtr)(reinterpret_cast<unsigned long long *>(fr.spill) + gtid);
    st.spill_stride = nthreads;
    st.sp = 0;
    st.nlds = NLDS;
    Trav s;
    s.cur = s.bank = TRAV_DONE;
    s.nodes = s.prims = 0;
    s.t = 0.0f;
    s.prim = NO_PRIM;
    s.any = s.occluded = false;
    uint32_t node = NO_PRIM, tag = 0;
    // the wave's work stack and its run of pool nodes (wave-uniform)
    uint2 *const wk = fr.wstack + (size_t)(gtid / 64u) * (fr.wstack_cap + (uint32_t)SHADE_LIST);
    const uint32_t wk_cap = fr.wstack_cap;
    // hits that wait for the shading pass: (path node, hit primitive | bounce level << 28)
    const LdsStackPtr wsh_lds = (LdsStackPtr)(lds_shade + (SHADE_IN_LDS ? (threadIdx.x / 64u) * (uint32_t)SHADE_LDS : 0u));
    uint2 *const wsh_glb = wk + fr.wstack_cap;
    uint32_t top = 0, shn = 0, run_next = 0, run_end = 0;
    uint32_t n_shadow = 0, n_bounce = 0;
    // entries that one closest-hit query can push when it ends (admission of new primaries)
    uint32_t per_hit = sc.max_bounce;
    for (uint32_t l = 0; l < sc.n_lights; l++) per_hit += RTX_LIGHT(sc, l).type != LIGHT_AMBIENT ? 1u : 0u;
    uint32_t vis_cn = 0, vis_cp = 0, vis_sn = 0, vis_sp = 0, max_steps = 0;  // COUNT
    Util util;
#ifdef RTX_WAVE_TIMES
    // Diagnostic (-DRTX_WAVE_TIMES, RTX_WAVE_TIMES=1): when each wave began, found the queue dry and ended (100 MHz
    // ticks), left in the last two entries of its shading list -- the production kernel otherwise, no other timers.
    const uint32_t wt_begin = (uint32_t)wall_clock64();
    uint32_t wt_dry = 0;
#endif
    // PRE: the queue is the list of primary hits (launch_path, primaries first), nd of them with bounce materials
    const uint32_t nd = PRE ? fr.counters[CTR_PRE_DEEP] : 0u;
    WaveQueue wq;
    wq.init(PRE ? fr.work + WORK_HALF : fr.work, PRE ? nd + fr.counters[CTR_PRE_SHALLOW] : n0, fr.stripe, nthreads / 64u, gtid / 64u);
    uint32_t refill_min = (uint32_t)RTX_REFILL_MIN, steps_run = 0;  // wave-uniform
#if RTX_CHAIN_PRIO
    bool prio_up = false;
#endif
    float end_rate = -1.0f;                                         // rays ending per step, running mean
#ifdef RTX_UTIL_STATS
    const unsigned long long t_begin = wall_clock64();
    unsigned long long t_dry = 0;
    uint32_t dry_events = 0, dry_steps0 = 0, dry_fin = 0, dry_top = 0, dry_fin_c = 0;
    unsigned long long dry_step_ticks0 = 0, dry_sec0[4] = {0, 0, 0, 0};
#endif
    for (;;) {
        const bool idle = s.done();
        const bool fin = idle && (tag & WK_BUSY);
        const unsigned long long idlem = __ballot(idle), finm = __ballot(fin);
        const uint32_t nidle = (uint32_t)__popcll(idlem), nfin = (uint32_t)__popcll(finm), nact = 64u - nidle;
        const uint32_t avail = top + shn + (wq.exhausted ? 0u : 64u);
        if (nact == 0u && nfin == 0u && avail == 0u) break;
#ifdef RTX_WAVE_TIMES
        if (wq.exhausted && wt_dry == 0u) wt_dry = (uint32_t)wall_clock64() | 1u;
#endif
        // lanes between rays are handled together once enough of them wait: a quarter of the
        // wave (of what is left of it once the queue of primaries is dry)
        const uint32_t serve = nidle < avail ? nidle : avail;
        const uint32_t pending = nfin > serve ? nfin : serve;
        uint32_t thr = refill_min;
        if (wq.exhausted) {
            const uint32_t live = nact + pending;
            thr = live >= 4u * refill_min ? refill_min : (live >= 8u ? live / 4u : 1u);
        }
        if (pending < thr && nact != 0u) {
            const uint32_t stop_at = nidle + (thr - pending) < 64u ? nidle + (thr - pending) : 64u;
#ifdef RTX_UTIL_STATS
            const unsigned long long t_s0 = wall_clock64();
#endif
            do {
                wave_step<COUNT, REF, OVH, BANK, XFC>(sc, s, st, (tag & WK_SHADOW) != 0u, lds_top, util);
                steps_run++;
            } while ((uint32_t)__popcll(__ballot(s.done())) < stop_at);
#ifdef RTX_UTIL_STATS
            util.step_ticks += wall_clock64() - t_s0;
#endif
            continue;
        }
        // The number of waiting lanes that pays for one pass through the code below depends on how
        // long this wave's rays are.  With r rays ending per step and a pass that costs E steps,
        // waiting for k lanes wastes about k*k/(2r) lane-steps of idling and 64 E of passing, so the
        // useful share of the wave is (k - k*k/128) / (k + a) with a = E r: largest at
        // k = sqrt(a*a + 128 a) - a.  Long rays (a soup of triangles) want a pass every few lanes,
        // short ones (most primaries miss the model) only once most of the wave waits.
        if (steps_run != 0u) {
#if RTX_FAST_THR
            // (a heuristic: the approximate reciprocal and square root, 2 instructions instead of 30)
            const float r = (float)nfin * __builtin_amdgcn_rcpf((float)steps_run);
            end_rate = end_rate < 0.0f ? r : 0.75f * end_rate + 0.25f * r;
            const float a = (float)RTX_EVENT_STEPS * end_rate;
            const float k = __builtin_amdgcn_sqrtf(a * a + 128.0f * a) - a;
#else
            const float r = (float)nfin / (float)steps_run;
            end_rate = end_rate < 0.0f ? r : 0.75f * end_rate + 0.25f * r;
            const float a = (float)RTX_EVENT_STEPS * end_rate;
            const float k = sqrtf(a * a + 128.0f * a) - a;
#endif
            refill_min = k < (float)RTX_REFILL_LO ? (uint32_t)RTX_REFILL_LO : (k > (float)RTX_REFILL_HI ? (uint32_t)RTX_REFILL_HI : (uint32_t)k);
            steps_run = 0;
        }
#ifdef RTX_UTIL_STATS
        util.events++;
        util.fin_lanes += nfin;
        if (wq.exhausted && t_dry == 0) {
            t_dry = wall_clock64();
            dry_steps0 = util.node_steps + util.prim_steps;
            dry_step_ticks0 = util.step_ticks;
            for (int k = 0; k < 4; k++) dry_sec0[k] = util.sec_ticks[k];
            dry_top = top + shn + nact;
        }
        if (t_dry) {
            dry_events++;
            dry_fin += nfin;
            dry_fin_c += (uint32_t)__popcll(__ballot(fin && !(tag & WK_SHADOW)));
        }
#endif
#ifdef RTX_UTIL_STATS
        unsigned long long t_sec = wall_clock64();
#define RTX_SECTION(k) { const unsigned long long t_now = wall_clock64(); util.sec_ticks[k] += t_now - t_sec; t_sec = t_now; }
#else
#define RTX_SECTION(k)
#endif
        // ---- the rays that have just finished
        if (nfin != 0u) {
            if (SHADE && fin && (tag & WK_SHADOW)) {
                // light.rs:93-96 / 111: point and cone lights reach iff something was hit and nothing
                // in front of the point; a directional light reaches iff nothing was hit at all
                const bool reaches = (tag & WK_DIR) ? !s.any : (s.any && !s.occluded);
                if (reaches) atomicOr(&fr.link[node].x, 1u << (tag & WK_LIGHT));
                if (COUNT) {
#ifdef RTX_RAY_HIST
                    ray_hist(fr.counters, 2u + ((tag & WK_DIR) ? 1u : 0u) + (s.r.ill_any() || s.r.exact_only ? 4u : 0u), s.nodes + s.prims);
#endif
                    vis_sn += s.nodes;
                    vis_sp += s.prims;
                    max_steps = max(max_steps, s.nodes + s.prims);
                    s.nodes = s.prims = 0;
                }
            }
            const bool finc = fin && !(tag & WK_SHADOW);
            if (finc) {
                fr.ray_o[node].w = s.any ? s.t : 0.0f;
                fr.ray_d[node].w = __uint_as_float(s.prim);
                if (COUNT) {
#ifdef RTX_RAY_HIST
                    ray_hist(fr.counters, ((tag >> WK_LEVEL_SHIFT) ? 1u : 0u) + (s.r.ill_any() || s.r.exact_only ? 4u : 0u), s.nodes + s.prims);
#endif
                    vis_cn += s.nodes;
                    vis_cp += s.prims;
                    max_steps = max(max_steps, s.nodes + s.prims);
                    s.nodes = s.prims = 0;
                }
                // a miss is the background (world.rs:127-130): nothing to shade
                if (SHADE && !s.any) fr.link[node] = make_uint2((tag >> WK_LEVEL_SHIFT) << LINK_LEVEL_SHIFT, NO_PRIM);
            }
            if constexpr (SHADE) {
                // hits wait in the wave's shading list until enough of them make a full-width pass
                const bool hitc = finc && s.any;
                const unsigned long long hm = __ballot(hitc);
                if (hitc) {
                    const uint2 e = make_uint2(node, s.prim | ((tag >> WK_LEVEL_SHIFT) << 28));
                    if constexpr (SHADE_IN_LDS) wsh_lds[shn + rank_below(hm)] = (unsigned long long)e.x | ((unsigned long long)e.y << 32);
                    else wsh_glb[shn + rank_below(hm)] = e;
                }
                shn += (uint32_t)__popcll(hm);
            }
            if (fin) tag &= ~WK_BUSY;
        }
        RTX_SECTION(0)
        // ---- shading pass over the listed hits (world.rs:120-125, material.rs:185-203): which lights
        // want a shadow query, the bounce children; run by the idle lanes once SHADE_MIN hits wait
        if constexpr (SHADE) {
            if (shn >= (uint32_t)RTX_SHADE_MIN || shn > (uint32_t)SHADE_LDS - 64u || (shn != 0u && wq.exhausted && top < nidle)) {
                const uint32_t rank = rank_below(idlem);
                const uint32_t n_sh = nidle < shn ? nidle : shn;
#ifdef RTX_UTIL_STATS
                util.shade_events++;
                util.shade_lanes += n_sh;
#endif
                const bool mine = idle && rank < n_sh;
                uint32_t hnode = 0, level = 0, hprim = 0, lit0 = 0, want = 0, nb = 0;
                F3 p = f3(0.f, 0.f, 0.f), d = p;
                if (mine) {
                    uint2 e;
                    if constexpr (SHADE_IN_LDS) {
                        const unsigned long long w = wsh_lds[shn - n_sh + rank];
                        e = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
                    } else {
                        e = wsh_glb[shn - n_sh + rank];
                    }
                    hnode = e.x;
                    level = e.y >> 28;
                    hprim = e.y & 0x0FFFFFFFu;
                    const float4 o4 = fr.ray_o[hnode], d4 = fr.ray_d[hnode];
                    // Does the hit's material spawn bounce rays?  The primitive's record is asked together with the
                    // ray's (the list carries the primitive), the material's only for ids in the range that has such
                    // materials at all; a scene without them (sc.max_bounce == 0) asks nothing.
                    uint32_t mid = 0xFFFFFFFFu;
                    if (sc.max_bounce != 0u && level + 1u < max_levels) mid = sc.prims[hprim].meta >> 16;
                    d = f3(d4.x, d4.y, d4.z);
                    p = f3(o4.x, o4.y, o4.z) + o4.w * d;  // world.rs:120
                    if (mid >= sc.bounce_mat_lo && mid <= sc.bounce_mat_hi) nb = sc.mats[mid].n_bounce;
                }
                shn -= n_sh;
                for (uint32_t l = 0; l < sc.n_lights; l++) {
                    const LightRec &L = RTX_LIGHT(sc, l);
                    if (L.type == LIGHT_AMBIENT) lit0 |= 1u << l;
                    else if (L.type != LIGHT_CONE || !outside_cone(L, p)) want |= 1u << l;
                }
                if (!mine) lit0 = want = 0;
                // shadow queries, light by light (a light's queries are popped together); the bounce children go on
                // top of them: the longest chain of a sample -- closest hit after closest hit down the
                // bounce levels -- is not held up by the shadow queries on the way
                for (uint32_t l = 0; l < sc.n_lights; l++) {
                    const bool wl = (want >> l) & 1u;
                    const unsigned long long m = __ballot(wl);
                    if (m == 0ull) continue;
                    const uint32_t cnt = (uint32_t)__popcll(m);
                    if (top + cnt > wk_cap) {
                        if (lane == 0) fr.counters[CTR_OVERFLOW] = 1;
                        continue;
                    }
                    if (wl)
                        wk[top + rank_below(m)] =
                            make_uint2(hnode, l | (RTX_LIGHT(sc, l).type == LIGHT_DIRECTIONAL ? WK_DIR : 0u) | WK_SHADOW |
                                                  (level << WK_LEVEL_SHIFT));
                    top += cnt;
                    n_shadow += cnt;
                }
                // bounce children: path nodes from this wave's run of the pool
                uint32_t first = NO_PRIM;
                const unsigned long long b0 = __ballot(nb & 1u), b1 = __ballot(nb & 2u), b2 = __ballot(nb & 4u);
                if ((b0 | b1 | b2) != 0ull) {
                    const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
                    const uint32_t before = rank_below(b0) + 2u * rank_below(b1) +
                                            4u * rank_below(b2);
                    if (run_next + total > run_end) {
                        // the rest of the old run becomes holes, a new run is reserved with ONE atomic
                        for (uint32_t h = run_next + lane; h < run_end; h += 64u)
                            fr.link[n0 + h] = make_uint2(LINK_LEVEL_HOLE << LINK_LEVEL_SHIFT, NO_PRIM);
                        const uint32_t wanted = total > (uint32_t)POOL_RUN ? total : (uint32_t)POOL_RUN;
                        run_next = wave_fetch(fr.counters + CTR_POOL, wanted);
                        run_end = run_next + wanted;
                        if ((unsigned long long)n0 + run_end > fr.node_cap) run_end = run_next;  // sizing error
                    }
                    if (run_next != run_end && top + total <= wk_cap) {
                        if (nb) {
                            first = n0 + run_next + before;
                            const PrimView pv = load_prim(sc.prims + hprim);
                            const MaterialRec m = sc.mats[pv.mat];
                            const F3 normal = object_normal(pv, sc.xforms, p);
                            for (uint32_t k = 0; k < m.prog_count; k++) {
                                const MatInstr &ins = sc.prog[m.prog_first + k];
                                if (ins.op != OP_REFLECT && ins.op != OP_REFRACT) continue;
                                F3 nd = (ins.op == OP_REFLECT) ? reflect(d, normal) : refract(d, normal, ins.p0);
                                // Ray::new(p, nd).offset(1e-4) (material.rs:189-191, ray.rs:43-46)
                                F3 rd = normalize(nd);
                                F3 ro = p + 1e-4f * rd;
                                rd = normalize(rd);
                                const uint32_t cn = first + ins.arg;
                                fr.ray_o[cn] = make_float4(ro.x, ro.y, ro.z, 0.0f);
                                fr.ray_d[cn] = make_float4(rd.x, rd.y, rd.z, __uint_as_float(NO_PRIM));
                                wk[top + before + ins.arg] = make_uint2(cn, (level + 1u) << WK_LEVEL_SHIFT);
                            }
                        }
                        run_next += total;
                        top += total;
                        n_bounce += total;
                    } else {  // reported; the children are dropped
                        if (lane == 0) fr.counters[CTR_OVERFLOW] = 1;
                        nb = 0;
                    }
                }
                if (mine) fr.link[hnode] = make_uint2(lit0 | (level << LINK_LEVEL_SHIFT) | (nb << LINK_NB_SHIFT), first);
            }
        }
        RTX_SECTION(1)
        // ---- work for the idle lanes: the top of the wave's stack first, then new primaries
        {
            const uint32_t rank = rank_below(idlem);
            const uint32_t n_pop = nidle < top ? nidle : top;
            bool got = false;
            if (idle && rank < n_pop) {
                const uint2 e = wk[top - n_pop + rank];
                node = e.x;
                tag = e.y | WK_BUSY;
                got = true;
            }
            top -= n_pop;
            if (nidle > n_pop && !wq.exhausted) {
                // every closest-hit query in flight and every listed hit can still push per_hit entries
                uint32_t n_new = nidle - n_pop;
                if (SHADE) {
                    const uint32_t inflight = (uint32_t)__popcll(__ballot((tag & (WK_BUSY | WK_SHADOW)) == WK_BUSY));
                    const uint32_t used = top + (inflight + shn) * per_hit, half = wk_cap / 2u;
#if RTX_ROOM_MUL
                    // (room for all of them, nearly always: one multiplication instead of the division)
                    if (used + n_new * per_hit > half) {
                        const uint32_t room = used < half ? (half - used) / (per_hit ? per_hit : 1u) : 0u;
                        n_new = n_new < room ? n_new : room;
                    }
#else
                    const uint32_t room = used < half ? (half - used) / (per_hit ? per_hit : 1u) : 0u;
                    n_new = n_new < room ? n_new : room;
#endif
                }
                const bool wants = idle && rank >= n_pop && rank < n_pop + n_new;
                const unsigned long long wmask = __ballot(wants);
                if (wmask != 0ull) {
                    uint32_t i = 0;
                    // (a run of the queue may end in the middle: the rest of the lanes take from the next one)
                    bool unserved = wants;
                    for (int round = 0; round < 4 && !wq.exhausted; round++) {
                        const unsigned long long um = __ballot(unserved);
                        if (um == 0ull) break;
                        if (wq.take(um, unserved, i)) {
                            node = PRE ? fr.preq[i < nd ? i : n0 - 1u - (i - nd)] : i;
                            tag = WK_BUSY;  // closest-hit query, level 0
                            got = true;
                            unserved = false;
                        }
                    }
                }
            }
#ifdef RTX_UTIL_STATS
            util.start_lanes += (uint32_t)__popcll(__ballot(got));
#endif
            RTX_SECTION(2)
            // ---- begin the new rays
            if (got) {
                const float4 o4 = fr.ray_o[node], d4 = fr.ray_d[node];
                // (a shadow query's light is asked for before the ray's record is looked at: one round trip, not two)
                F3 lpos = f3(0.f, 0.f, 0.f), ln1 = lpos, ln2 = lpos;
                if (tag & WK_SHADOW) {
                    const LightRec &L = sc.lights[tag & WK_LIGHT];
                    lpos = f3(L.pos[0], L.pos[1], L.pos[2]);
                    ln1 = f3(L.ndir1[0], L.ndir1[1], L.ndir1[2]);
                    ln2 = f3(L.ndir2[0], L.ndir2[1], L.ndir2[2]);
                }
                F3 ro = f3(o4.x, o4.y, o4.z), rd = f3(d4.x, d4.y, d4.z);
                float dist = INFINITY;  // closest hit: nothing found yet; directional light: any hit at all blocks
                if (tag & WK_SHADOW) {
                    const F3 p = ro + o4.w * rd;  // world.rs:120
                    if (tag & WK_DIR) {
                        // Ray::new(point, -direction).offset(1e-4) (light.rs:109-110)
                        // Ray::new(point, -direction).offset(1e-4) (light.rs:109-110): both normalisations are the
                        // flattener's (LightRec::ndir1/2)
                        ro = p + 1e-4f * ln1;
                        rd = ln2;
                    } else {
                        // Light::in_shadow (light.rs:81-97)
                        ro = lpos;
#if RTX_ONE_SQRT
                        // normalize(p - ro) and magnitude(ro - p) (light.rs:82-84) take the square root of the same
                        // number: the components of ro - p are those of p - ro negated, exactly, and their squares equal
                        const F3 v = p - ro;
                        dist = magnitude(v);
                        rd = v * (1.0f / dist);
#else
                        rd = normalize(p - ro);
                        dist = magnitude(ro - p);
#endif
                    }
                }
                if (PRE && tag == WK_BUSY) {
                    // a primary hit of the first launch: nothing to trace, it joins the hits that wait for shading
                    s.cur = s.bank = TRAV_DONE;
                    s.any = true;
                    s.t = o4.w;
                    s.prim = __float_as_uint(d4.w);
                    if (COUNT) s.nodes = s.prims = 0;
                } else if (!(tag & WK_SHADOW) && __float_as_uint(d4.w) == INVALID_PRIM) {  // pixel slot outside the image (padded tile)
                    if (SHADE) fr.link[node] = make_uint2(0u, NO_PRIM);
                    tag = 0;
                } else {
                    init_ray<XFC>(s.r, sc, ro, rd);
                    s.t = dist;
                    trav_begin<REF, OVH>(sc, s, st);  // (a ray that misses the root box is done at once)
                }
            }
            RTX_SECTION(3)
        }
#if RTX_CHAIN_PRIO
        // A frame's launch is as long as its longest sample: closest hit after closest hit down the bounce levels,
        // every step of every ray a dependent load.  A wave that carries such a link -- a closest-hit query of a
        // bounce level -- issues ahead of the waves it shares its SIMD with.  (Measured, round 3: rank 0's 1/8 share
        // of the headline frame 2.44 -> 2.28 ms, configs 1 and 2 -2 %, the whole frame +-0; priority 1 and 3 alike.
        // Passing as soon as such a query ends instead of waiting for the pass threshold: +-0 / +1.5 %.)
        if constexpr (SHADE) {
            const bool chain = __ballot((tag & (WK_BUSY | WK_SHADOW)) == WK_BUSY && (tag >> WK_LEVEL_SHIFT) != 0u) != 0ull;
            if (chain != prio_up) {
                if (chain) __builtin_amdgcn_s_setprio(RTX_CHAIN_PRIO);
                else __builtin_amdgcn_s_setprio(0);
                prio_up = chain;
            }
        }
#endif
    }
#ifdef RTX_WAVE_TIMES
    if (SHADE && lane == 0) {
        uint2 *const wt = wk + fr.wstack_cap;  // (the words behind the wave's work stack)
        wt[SHADE_LIST - 2] = make_uint2(wt_begin, wt_dry);
        wt[SHADE_LIST - 1] = make_uint2((uint32_t)wall_clock64(), 0x57415645u);
    }
#endif
    // what is left of the wave's last run of pool nodes
    if constexpr (SHADE) {
        for (uint32_t h = run_next + lane; h < run_end; h += 64u)
            fr.link[n0 + h] = make_uint2(LINK_LEVEL_HOLE << LINK_LEVEL_SHIFT, NO_PRIM);
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(fr.counters + CTR_BOUNCE_LO), (unsigned long long)n_bounce);
            atomicAdd(reinterpret_cast<unsigned long long *>(fr.counters + CTR_SHADOW_LO), (unsigned long long)n_shadow);
        }
    }
    util.flush(fr.counters);
#ifdef RTX_UTIL_STATS
    if (lane == 0) {  // wave lifetimes in 100 MHz ticks: [10] earliest begin, [11] latest end, [12] sum of lifetimes, [13] sum of time after the queue ran dry
        unsigned long long *c = reinterpret_cast<unsigned long long *>(fr.counters + CTR_UTIL);
        const unsigned long long t_end = wall_clock64();
        atomicMin(c + 10, t_begin);
        atomicMax(c + 11, t_end);
        atomicAdd(c + 12, t_end - t_begin);
        atomicAdd(c + 13, t_dry ? t_end - t_dry : 0ull);
        atomicMax(c + 14, t_dry ? t_end - t_dry : 0ull);
        if (t_dry && t_end - t_dry > 60000ull) {  // > 0.6 ms after the queue ran dry: a straggler, listed behind the counters
            uint32_t *dbg = fr.counters + CTR_WORDS;
            const uint32_t k = atomicAdd(dbg, 1u);
            if (k < 24u) {
                uint32_t *r = dbg + 1 + 10 * k;
                r[6] = (uint32_t)(util.step_ticks - dry_step_ticks0);
                r[7] = (uint32_t)(util.sec_ticks[0] - dry_sec0[0]) + (uint32_t)(util.sec_ticks[2] - dry_sec0[2]);
                r[8] = (uint32_t)(util.sec_ticks[1] - dry_sec0[1]);
                r[9] = (uint32_t)(util.sec_ticks[3] - dry_sec0[3]);
                r[0] = (uint32_t)(t_end - t_dry);
                r[1] = dry_events;
                r[2] = util.node_steps + util.prim_steps - dry_steps0;
                r[3] = dry_fin;
                r[4] = dry_top;
                r[5] = dry_fin_c;
            }
        }
    }
#endif
    if (COUNT) {
        add_visits(fr.counters, vis_cn, vis_cp, CTR_VISIT_NODES_LO);
        add_visits(fr.counters, vis_sn, vis_sp, CTR_SVISIT_NODES_LO);
        atomicMax(fr.counters + CTR_MAX_STEPS, max_steps);
    }
}

template <bool COUNT, bool REF, bool SHADE, bool BANK = false, bool PRE = false, bool XFC = false>
__global__ void __launch_bounds__(TBLOCK, RTX_TRACE_WAVES) path_kernel(DeviceScene sc, Frame fr, uint32_t max_levels) {
    path_body<COUNT, REF, false, SHADE, BANK, PRE, XFC>(sc, fr, max_levels);
}
// The production traversal for scenes whose hits can lie outside their objects' boxes
// (DeviceScene::overhang != 0): the per-slot allowance of node_step is compiled in here only.
template <bool COUNT, bool SHADE, bool PRE = false, bool XFC = false>
__global__ void __launch_bounds__(TBLOCK, RTX_TRACE_WAVES) path_ovh_kernel(DeviceScene sc, Frame fr, uint32_t max_levels) {
    path_body<COUNT, false, true, SHADE, false, PRE, XFC>(sc, fr, max_levels);
}

// ---------------------------------------------------------------- primaries first: the list of primary hits
// Between the two persistent launches of a primaries-first batch (launch_path).  A primary that hit nothing (or
// whose pixel slot lies outside the image) gets the record the path kernel would have written; a hit is classed
// by its material -- 1: spawns bounce rays, 2: does not -- in link.x, which nothing reads before the second
// launch's shading pass overwrites it.  Blocks of PRE_SEG primaries count, one block scans, the same blocks scatter:
// the list keeps the queue's order within each class.
__global__ void __launch_bounds__(BLOCK) pre_count_kernel(DeviceScene sc, Frame fr, uint32_t n0) {
    __shared__ uint32_t cnt[2];
    if (threadIdx.x < 2u) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t seg = blockIdx.x * (uint32_t)PRE_SEG;
    uint32_t c1 = 0, c2 = 0;
    for (uint32_t k = 0; k < (uint32_t)PRE_SEG / BLOCK; k++) {
        const uint32_t i = seg + k * BLOCK + threadIdx.x;
        if (i >= n0) break;
        const uint32_t prim = __float_as_uint(fr.ray_d[i].w);
        uint32_t cls = 0;
        if (prim < INVALID_PRIM) cls = sc.mats[sc.prims[prim].meta >> 16].n_bounce != 0u ? 1u : 2u;
        fr.link[i] = make_uint2(cls, NO_PRIM);   // (cls 0: the final record of a miss, world.rs:127-130)
        c1 += cls == 1u;
        c2 += cls == 2u;
    }
    for (int d = 32; d > 0; d >>= 1) {
        c1 += __shfl_down(c1, d, 64);
        c2 += __shfl_down(c2, d, 64);
    }
    if (lane_id() == 0) {
        atomicAdd(&cnt[0], c1);
        atomicAdd(&cnt[1], c2);
    }
    __syncthreads();
    if (threadIdx.x < 2u) fr.pre_blocks[2u * blockIdx.x + threadIdx.x] = cnt[threadIdx.x];
}
__global__ void __launch_bounds__(1024) pre_scan_kernel(Frame fr, uint32_t nblocks) {
    __shared__ uint32_t part[2][1024];
    const uint32_t t = threadIdx.x, per = (nblocks + 1023u) / 1024u;
    const uint32_t b0 = t * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t b = b0; b < b1; b++) {
        s1 += fr.pre_blocks[2u * b];
        s2 += fr.pre_blocks[2u * b + 1u];
    }
    part[0][t] = s1;
    part[1][t] = s2;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {  // inclusive scan of the 1024 partial sums
        const uint32_t a1 = t >= d ? part[0][t - d] : 0u, a2 = t >= d ? part[1][t - d] : 0u;
        __syncthreads();
        part[0][t] += a1;
        part[1][t] += a2;
        __syncthreads();
    }
    uint32_t o1 = part[0][t] - s1, o2 = part[1][t] - s2;
    for (uint32_t b = b0; b < b1; b++) {
        const uint32_t c1 = fr.pre_blocks[2u * b], c2 = fr.pre_blocks[2u * b + 1u];
        fr.pre_blocks[2u * b] = o1;
        fr.pre_blocks[2u * b + 1u] = o2;
        o1 += c1;
        o2 += c2;
    }
    if (t == 1023u) {
        fr.counters[CTR_PRE_DEEP] = part[0][t];
        fr.counters[CTR_PRE_SHALLOW] = part[1][t];
    }
}
__global__ void __launch_bounds__(BLOCK) pre_scatter_kernel(Frame fr, uint32_t n0) {
    __shared__ uint32_t wcnt[2][BLOCK / 64];
    __shared__ uint32_t base[2];
    const uint32_t seg = blockIdx.x * (uint32_t)PRE_SEG, w = threadIdx.x / 64u;
    const unsigned long long lt = (1ull << lane_id()) - 1ull;
    if (threadIdx.x < 2u) base[threadIdx.x] = fr.pre_blocks[2u * blockIdx.x + threadIdx.x];
    for (uint32_t k = 0; k < (uint32_t)PRE_SEG / BLOCK; k++) {
        const uint32_t i = seg + k * BLOCK + threadIdx.x;
        const uint32_t cls = i < n0 ? fr.link[i].x : 0u;
        const unsigned long long m1 = __ballot(cls == 1u), m2 = __ballot(cls == 2u);
        if (lane_id() == 0) {
            wcnt[0][w] = (uint32_t)__popcll(m1);
            wcnt[1][w] = (uint32_t)__popcll(m2);
        }
        __syncthreads();
        uint32_t o1 = base[0], o2 = base[1], t1 = 0, t2 = 0;
        for (uint32_t v = 0; v < BLOCK / 64u; v++) {
            if (v < w) {
                o1 += wcnt[0][v];
                o2 += wcnt[1][v];
            }
            t1 += wcnt[0][v];
            t2 += wcnt[1][v];
        }
        if (cls == 1u) fr.preq[o1 + (uint32_t)__popcll(m1 & lt)] = i;
        if (cls == 2u) fr.preq[n0 - 1u - (o2 + (uint32_t)__popcll(m2 & lt))] = i;
        const uint32_t nb1 = base[0] + t1, nb2 = base[1] + t2;
        __syncthreads();
        if (threadIdx.x == 0u) {
            base[0] = nb1;
            base[1] = nb2;
        }
    }
}

// ---------------------------------------------------------------- resolve
// Light::get_falloff_at (light.rs:58-79).  The cone factor is clamp((cut-off - angle) / 8 degrees, 0, 1): exactly
// 1 for every point more than 8 degrees inside the cone, which the signed squared cosine settles
// (cos_full carries 0.01 degrees of margin, a thousand times the rounding of either side); only
// the rim runs the reference's atan2, behind a wave-uniform branch.
__device__ __forceinline__ float falloff_at(const LightRec &L, F3 point) {
    F3 lpos = f3(L.pos[0], L.pos[1], L.pos[2]);
    F3 dv = lpos - point;
    float d2 = dot(dv, dv);
    if (L.type == 1u) return 5.0f / (0.001f + d2);
    const F3 a = f3(L.dir[0], L.dir[1], L.dir[2]), b = point - lpos;
    const float dt = dot(a, b), l2 = dot(a, a) * dot(b, b);
    const bool rim = !(dt > 0.0f && dt * dt > (L.cos_full * fabsf(L.cos_full)) * l2 && L.cos_full > 0.0f && L.cos_full <= 1.0f);
    float cone = 1.0f;
    if (__ballot(rim) != 0ull) {
        if (rim) {
            float angle_to_point = angle_rad(a, b) * RAD_TO_DEG;
            float angle_delta = L.angle_deg - angle_to_point;
            cone = clampf(angle_delta / 8.0f, 0.0f, 1.0f);
        }
    }
    return cone * 5.0f / (0.001f + d2);
}
// MaterialType::get_phong_multiple (material.rs:98-113).  ks * spec^shininess is +0 without calling powf when
// ks or spec is +0 and the exponent positive (powf of a number in [0, 1] is finite, powf(+0, y > 0) is +0).
__device__ __forceinline__ float phong_multiple(F3 to_light, F3 normal, F3 incoming, float kd, float ks, float sh) {
    F3 rv = reflect(to_light, normal);
    float spec = clampf(-dot(rv, incoming), 0.0f, 1.0f);
    float diff = clampf(dot(to_light, normal), 0.0f, 1.0f);
    const bool zero = sh > 0.0f && sh < INFINITY && ks >= 0.0f && ks < INFINITY && spec == spec &&
                      (__float_as_uint(ks) == 0u || __float_as_uint(spec) == 0u);
    float sp = 0.0f;
    if (__ballot(!zero) != 0ull) {
        if (!zero) sp = ks * powf(spec, sh);
    }
    return kd * diff + sp;
}
// TextureType::sample (material.rs:53-76)
__device__ __forceinline__ F3 texture_sample(const DeviceScene &sc, const MaterialRec &m, const PrimView &pv, F3 p) {
    if (m.tex_kind == 2u) {
        TextureRec tx = sc.tex[m.tex_id];
        float width = (float)tx.w, height = (float)tx.h;
        float u, v;
        object_uv(pv, sc.xforms, p, u, v);
        float ru = fmodf(u, 1.0f);
        if (ru < 0.0f) ru = ru + 1.0f;
        float rv = fmodf(v, 1.0f);
        if (rv < 0.0f) rv = rv + 1.0f;
        float x = truncf(ru * width), y = truncf(rv * height);
        float xc = clampf(x, 0.0f, width - 1.0f), yc = clampf(y, 0.0f, height - 1.0f);
        uint32_t xi = (xc == xc) ? (uint32_t)xc : 0u, yi = (yc == yc) ? (uint32_t)yc : 0u;
        const uint8_t *px = sc.tex_data + tx.offset + ((size_t)yi * tx.w + xi) * 3;
        return crgb((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
    }
    if (m.tex_kind == 1u) return f3(m.tex_color[0], m.tex_color[1], m.tex_color[2]);
    return crgb(0.5f, 0.5f, 0.5f);
}

// Level 0 = the batch's primary samples, nodes [0, n0); the nodes of a deeper level are those
// pool nodes [n0, n0 + pool) whose link carries that level (runs of the pool belong to waves, so
// the levels are interleaved; unused slots of a run are marked as holes).
// Colour of one path node (its bounce children, if any, have theirs).
__device__ __forceinline__ void resolve_node(const DeviceScene &sc, const Frame &fr, F3 bg, uint32_t node, uint2 lk) {
    float4 o4 = fr.ray_o[node], d4 = fr.ray_d[node];
    uint32_t prim = __float_as_uint(d4.w);
    F3 res;
    if (prim >= INVALID_PRIM) {
        res = prim == INVALID_PRIM ? f3(0.f, 0.f, 0.f) : bg;  // world.rs:127-130
    } else {
        F3 o = f3(o4.x, o4.y, o4.z), d = f3(d4.x, d4.y, d4.z);
        F3 p = o + o4.w * d;
        PrimView pv = load_prim(sc.prims + prim);
        MaterialRec m = sc.mats[pv.mat];
        F3 surface = texture_sample(sc, m, pv, p);  // material.rs:229-230
        F3 normal = f3(0.f, 0.f, 0.f);
        bool have_normal = false;
        // accumulator stack of the Composition fold (material.rs:129-143), depth <= 4, in
        // registers: a dynamically indexed array would live in scratch memory
        F3 acc0 = f3(0.f, 0.f, 0.f), acc1 = acc0, acc2 = acc0, acc3 = acc0;
        int asp = 0;
        F3 val = f3(0.f, 0.f, 0.f);
        for (uint32_t k = 0; k < m.prog_count; k++) {
            MatInstr ins = sc.prog[m.prog_first + k];
            switch (ins.op) {
                case OP_BEGIN:
                    if (asp == 0) acc0 = f3(0.f, 0.f, 0.f);
                    else if (asp == 1) acc1 = f3(0.f, 0.f, 0.f);
                    else if (asp == 2) acc2 = f3(0.f, 0.f, 0.f);
                    else if (asp == 3) acc3 = f3(0.f, 0.f, 0.f);
                    asp++;
                    break;
                case OP_ITEM: {
                    F3 term = cscale(val, ins.p0);
                    if (asp == 1) acc0 = cadd(acc0, term);
                    else if (asp == 2) acc1 = cadd(acc1, term);
                    else if (asp == 3) acc2 = cadd(acc2, term);
                    else if (asp == 4) acc3 = cadd(acc3, term);
                    break;
                }
                case OP_END:
                    asp--;
                    if (asp == 0) val = acc0;
                    else if (asp == 1) val = acc1;
                    else if (asp == 2) val = acc2;
                    else if (asp == 3) val = acc3;
                    break;
                case OP_PHONG: {  // material.rs:144-184
                    if (!have_normal) {
                        normal = object_normal(pv, sc.xforms, p);
                        have_normal = true;
                    }
                    F3 sum = f3(0.f, 0.f, 0.f);
                    for (uint32_t l = 0; l < sc.n_lights; l++) {
                        if (!(lk.x & (1u << l))) continue;
                        const LightRec &L = RTX_LIGHT(sc, l);
                        F3 lcol = f3(L.color[0], L.color[1], L.color[2]);
                        F3 lc;
                        if (L.type == LIGHT_AMBIENT) lc = lcol;
                        else if (L.type == 2u) {
                            float pm = phong_multiple(-f3(L.dir[0], L.dir[1], L.dir[2]), normal, d, ins.p0, ins.p1, ins.p2);
                            lc = cscale(lcol, pm);
                        } else {
                            F3 light_dir = f3(L.pos[0], L.pos[1], L.pos[2]) - p;
                            float fo = falloff_at(L, p);
                            float pm = phong_multiple(normalize(light_dir), normal, d, ins.p0, ins.p1, ins.p2);
                            lc = cscale(cscale(lcol, fo), pm);
                        }
                        sum = cadd(sum, cmul(surface, lc));
                    }
                    val = sum;
                    break;
                }
                case OP_REFLECT:
                case OP_REFRACT: {
                    if (lk.y != NO_PRIM) {
                        float4 c = fr.color[lk.y + ins.arg];
                        val = f3(c.x, c.y, c.z);
                    } else {
                        val = bg;  // trace_ray(.., 0) (world.rs:116-117)
                    }
                    break;
                }
                default: val = crgb(0.5f, 0.5f, 0.5f); break;  // material.rs:204
            }
        }
        res = val;
    }
    fr.color[node] = make_float4(res.x, res.y, res.z, 1.0f);
}

__global__ void __launch_bounds__(BLOCK) resolve_kernel(DeviceScene sc, Frame fr, uint32_t level) {
    const uint32_t n0 = fr.counters[CTR_LEVEL_COUNT];
    const uint32_t nthreads = gridDim.x * BLOCK;
    const F3 bg = f3(sc.background[0], sc.background[1], sc.background[2]);
    if (level == 0u) {
        for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n0; i += nthreads) resolve_node(sc, fr, bg, i, fr.link[i]);
        return;
    }
    // Deeper levels: the pool interleaves the levels (about a third of the nodes of a stretch belong to
    // this launch's), so a block first lists the matching nodes of RESOLVE_SCAN stretches of BLOCK nodes in LDS
    // and then works the list off at full lane width.
    __shared__ uint32_t list[RESOLVE_SCAN * BLOCK];
    __shared__ uint32_t nlist;
    const uint32_t count = min(fr.counters[CTR_POOL], fr.node_cap - n0);
    for (uint32_t t0 = blockIdx.x * (RESOLVE_SCAN * BLOCK); t0 < count; t0 += gridDim.x * (RESOLVE_SCAN * BLOCK)) {
        if (threadIdx.x == 0) nlist = 0;
        __syncthreads();
        for (uint32_t k = 0; k < (uint32_t)RESOLVE_SCAN; k++) {
            const uint32_t i = t0 + k * BLOCK + threadIdx.x;
            const bool match = i < count && ((fr.link[n0 + i].x >> LINK_LEVEL_SHIFT) & 15u) == level;
            const unsigned long long m = __ballot(match);
            uint32_t at = 0;
            if (lane_id() == 0 && m != 0ull) at = atomicAdd(&nlist, (uint32_t)__popcll(m));
            at = __builtin_amdgcn_readfirstlane(at);
            if (match) list[at + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull))] = n0 + i;
        }
        __syncthreads();
        const uint32_t n = nlist;
        for (uint32_t j = threadIdx.x; j < n; j += BLOCK) {
            const uint32_t node = list[j];
            resolve_node(sc, fr, bg, node, fr.link[node]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- accumulate
__global__ void __launch_bounds__(BLOCK) accumulate_kernel(DeviceScene sc, Frame fr, BatchInfo bi, float4 *sum,
                                                           uint8_t *tiles_rgb8, float *tiles_f32) {
    uint32_t q = blockIdx.x * BLOCK + threadIdx.x;
    if (q >= bi.npix) return;
    uint32_t lp = bi.q0 + q;
    float4 a = (bi.s0 == 0) ? make_float4(0.f, 0.f, 0.f, 0.f) : sum[lp];
    for (uint32_t sl = 0; sl < bi.ns; sl++) {  // fold in sample order (world.rs:54-61)
        float4 c = fr.color[((size_t)(q >> 6) * bi.ns + sl) * 64u + (q & 63u)];  // (raygen: node order)
        a.x = a.x + c.x;
        a.y = a.y + c.y;
        a.z = a.z + c.z;
    }
    if (bi.s0 + bi.ns < bi.spp) {
        sum[lp] = a;
        return;
    }
    uint32_t px, py, toff;
    bool valid = slot_to_pixel(bi, sc.width, sc.height, lp, px, py, toff);
    float n = (float)bi.spp;
    F3 res = valid ? crgb(a.x / n, a.y / n, a.z / n) : f3(0.f, 0.f, 0.f);  // world.rs:62-63
    // Color::get_rgb: (c * 255.0) as u8 (color.rs:25-31)
    float v[3] = {res.x * 255.0f, res.y * 255.0f, res.z * 255.0f};
    for (int k = 0; k < 3; k++) {
        float f = v[k];
        uint8_t b = (!(f == f) || f <= 0.0f) ? 0 : (f >= 255.0f ? 255 : (uint8_t)f);
        tiles_rgb8[(size_t)toff * 3 + k] = b;
    }
    if (tiles_f32) {
        tiles_f32[(size_t)toff * 3] = res.x;
        tiles_f32[(size_t)toff * 3 + 1] = res.y;
        tiles_f32[(size_t)toff * 3 + 2] = res.z;
    }
}

__global__ void __launch_bounds__(BLOCK) export_hits_kernel(DeviceScene sc, Frame fr, uint32_t n, uint32_t *out_obj,
                                                            float *out_t) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t prim = __float_as_uint(fr.ray_d[i].w);
    out_obj[i] = prim < INVALID_PRIM ? sc.prim_obj[prim] : 0xFFFFFFFFu;
    out_t[i] = fr.ray_o[i].w;
}
__global__ void __launch_bounds__(BLOCK) export_colors_kernel(Frame fr, uint32_t n, float *out_rgb) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float4 c = fr.color[i];
    out_rgb[(size_t)i * 3] = c.x;
    out_rgb[(size_t)i * 3 + 1] = c.y;
    out_rgb[(size_t)i * 3 + 2] = c.z;
}

// one thread = one pixel of the assembled image: tile gt = tile row * tiles_x + tile column belongs to rank
// gt % nranks, where it is local tile gt / nranks
__global__ void __launch_bounds__(BLOCK) deinterleave_kernel(const uint8_t *all_tiles, size_t rank_stride, uint32_t nranks,
                                                             uint32_t tile, uint32_t width, uint32_t height, uint8_t *image) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t px = i % width, py = i / width, tiles_x = (width + tile - 1) / tile;
    const uint32_t gt = (py / tile) * tiles_x + px / tile;
    const uint8_t *src = all_tiles + (size_t)(gt % nranks) * rank_stride +
                         ((size_t)(gt / nranks) * tile * tile + (size_t)(py % tile) * tile + px % tile) * 3;
    uint8_t *dst = image + (size_t)i * 3;
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}

inline int grid_for(uint32_t n) { return (int)((n + BLOCK - 1) / BLOCK); }

}  // namespace

void launch_raygen(const DeviceScene &sc, const Frame &fr, const BatchInfo &bi, const LaunchCfg &lc) {
    uint32_t n = bi.npix * bi.ns;
    hipLaunchKernelGGL(raygen_kernel, dim3(grid_for(n ? n : 1)), dim3(BLOCK), 0, lc.stream, sc, fr, bi);
}
void launch_load_rays(const Frame &fr, const float *org_dir, uint32_t n, const LaunchCfg &lc) {
    hipLaunchKernelGGL(load_rays_kernel, dim3(grid_for(n ? n : 1)), dim3(BLOCK), 0, lc.stream, fr, org_dir, n);
}
void launch_path(const DeviceScene &sc, const Frame &fr, uint32_t max_levels, bool shade, const LaunchCfg &lc, uint32_t n0) {
    dim3 g(std::max(1, lc.persistent_blocks * BLOCK / TBLOCK)), b(TBLOCK);
#define RTX_LAUNCH_PATH(K) hipLaunchKernelGGL(K, g, b, 0, lc.stream, sc, fr, max_levels)
    // lc.xform_cache: the variant that keeps the ray in the object space of the last transform a test needed (RayCtx);
    // the verbatim and the leaf-banking kernels exist without it only (their scenes are checked / are plain meshes)
#define RTX_LAUNCH_XFC(...)                                           \
    do {                                                              \
        if (lc.xform_cache) RTX_LAUNCH_PATH((__VA_ARGS__, true>));   \
        else RTX_LAUNCH_PATH((__VA_ARGS__, false>));                 \
    } while (0)
    const bool ovh = sc.overhang != 0.0f;
    if (shade && lc.primaries_first && !lc.reference_traversal && !lc.count_visits && !lc.leaf_bank && fr.preq && fr.pre_blocks && n0 != 0u) {
        if (ovh) RTX_LAUNCH_XFC(path_ovh_kernel<false, false, false);
        else RTX_LAUNCH_XFC(path_kernel<false, false, false, false, false);
        const uint32_t nblocks = (n0 + (uint32_t)PRE_SEG - 1u) / (uint32_t)PRE_SEG;
        hipLaunchKernelGGL(pre_count_kernel, dim3(nblocks), dim3(BLOCK), 0, lc.stream, sc, fr, n0);
        hipLaunchKernelGGL(pre_scan_kernel, dim3(1), dim3(1024), 0, lc.stream, fr, nblocks);
        hipLaunchKernelGGL(pre_scatter_kernel, dim3(nblocks), dim3(BLOCK), 0, lc.stream, fr, n0);
        if (ovh) RTX_LAUNCH_XFC(path_ovh_kernel<false, true, true);
        else RTX_LAUNCH_XFC(path_kernel<false, false, true, false, true);
        return;
    }
    if (!shade) {  // closest hits of the primaries only (no visit counters on this path)
        if (lc.reference_traversal) RTX_LAUNCH_PATH((path_kernel<false, true, false>));
        else if (ovh) RTX_LAUNCH_XFC(path_ovh_kernel<false, false, false);
        else RTX_LAUNCH_XFC(path_kernel<false, false, false, false, false);
    } else if (lc.reference_traversal) {
        if (lc.count_visits) RTX_LAUNCH_PATH((path_kernel<true, true, true>));
        else RTX_LAUNCH_PATH((path_kernel<false, true, true>));
    } else if (ovh) {
        if (lc.count_visits) RTX_LAUNCH_XFC(path_ovh_kernel<true, true, false);
        else RTX_LAUNCH_XFC(path_ovh_kernel<false, true, false);
    } else if (lc.leaf_bank) {
        if (lc.count_visits) RTX_LAUNCH_PATH((path_kernel<true, false, true, true>));
        else RTX_LAUNCH_PATH((path_kernel<false, false, true, true>));
    } else {
        if (lc.count_visits) RTX_LAUNCH_XFC(path_kernel<true, false, true, false, false);
        else RTX_LAUNCH_XFC(path_kernel<false, false, true, false, false);
    }
#undef RTX_LAUNCH_XFC
#undef RTX_LAUNCH_PATH
}
void launch_resolve(const DeviceScene &sc, const Frame &fr, uint32_t level, const LaunchCfg &lc) {
    hipLaunchKernelGGL(resolve_kernel, dim3(lc.persistent_blocks), dim3(BLOCK), 0, lc.stream, sc, fr, level);
}
void launch_accumulate(const DeviceScene &sc, const Frame &fr, const BatchInfo &bi, float4 *sum, uint8_t *tiles_rgb8,
                       float *tiles_f32, const LaunchCfg &lc) {
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid_for(bi.npix ? bi.npix : 1)), dim3(BLOCK), 0, lc.stream, sc, fr, bi,
                       sum, tiles_rgb8, tiles_f32);
}
void launch_export_hits(const DeviceScene &sc, const Frame &fr, uint32_t n, uint32_t *out_obj, float *out_t,
                        const LaunchCfg &lc) {
    hipLaunchKernelGGL(export_hits_kernel, dim3(grid_for(n ? n : 1)), dim3(BLOCK), 0, lc.stream, sc, fr, n, out_obj,
                       out_t);
}
void launch_export_colors(const Frame &fr, uint32_t n, float *out_rgb, const LaunchCfg &lc) {
    hipLaunchKernelGGL(export_colors_kernel, dim3(grid_for(n ? n : 1)), dim3(BLOCK), 0, lc.stream, fr, n, out_rgb);
}

void launch_deinterleave(const uint8_t *all_tiles, size_t rank_stride, uint32_t nranks, uint32_t tile, uint32_t width,
                         uint32_t height, uint8_t *image, hipStream_t stream) {
    const uint32_t n = width * height;
    hipLaunchKernelGGL(deinterleave_kernel, dim3(grid_for(n ? n : 1)), dim3(BLOCK), 0, stream, all_tiles, rank_stride, nranks, tile,
                       width, height, image);
}

int query_persistent_blocks(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 256 * 4;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, path_kernel<false, false, true>, TBLOCK, 0) !=
            hipSuccess ||
        per_cu < 1)
        per_cu = 4 * BLOCK / TBLOCK;
    return prop.multiProcessorCount * per_cu * TBLOCK / BLOCK;  // in units of BLOCK threads
}

}  // namespace rtx
