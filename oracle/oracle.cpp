/*
 * oracle/oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the trace path of ellishg/rust-raytracer, written function by
 * function from the reference's source.  Every function cites the reference lines it
 * follows (paths relative to /root/reference).  IEEE f32 throughout; build with
 * -ffp-contract=off so no multiply-add is fused (rustc never fuses).
 *
 * Third-party arithmetic restated from memory (cgmath 0.17.0, Cargo.lock:81-82; the
 * crate source is not in this container):
 *   dot(a,b)        = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   normalize(v)    = v * (1 / sqrt(dot(v,v)))
 *   cross           = (ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx)
 *   M * (x,y,z,w)   = ((c0*x + c1*y) + c2*z) + c3*w      (column-major)
 *   transform_point = (M*(p,1)).xyz * (1 / w);  transform_vector = (M*(v,0)).xyz
 *   Vector3::angle  = atan2(|a x b|, a.b);  Rad->Deg = * f32(180/pi); Deg->Rad = * f32(pi/180)
 *   Point3::centroid([a,b]) = ((0 + a) + b) / 2
 *
 * NOT the thing measured or shipped: only tests/, smoke() and bench.py's
 * cpu_baseline leg may load this.
 */
#include "oracle.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------ vectors (cgmath)
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float magnitude2(V3 a) { return dot(a, a); }
inline float magnitude(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0f / magnitude(a)); }
inline float angle_rad(V3 a, V3 b) { return std::atan2(magnitude(cross(a, b)), dot(a, b)); }
const float RAD_TO_DEG = (float)(180.0 / 3.14159265358979323846);
const float DEG_TO_RAD = (float)(3.14159265358979323846 / 180.0);
const float PI_F = 3.14159265358979323846f;

struct M4 {
    float m[16];  // column-major: m[c*4+r]
};
inline void mul_v4(const M4 &M, float x, float y, float z, float w, float out[4]) {
    for (int r = 0; r < 4; r++)
        out[r] = ((M.m[0 + r] * x + M.m[4 + r] * y) + M.m[8 + r] * z) + M.m[12 + r] * w;
}
inline V3 transform_point(const M4 &M, V3 p) {
    float h[4];
    mul_v4(M, p.x, p.y, p.z, 1.0f, h);
    float iw = 1.0f / h[3];
    return v3(h[0] * iw, h[1] * iw, h[2] * iw);
}
inline V3 transform_vector(const M4 &M, V3 v) {
    float h[4];
    mul_v4(M, v.x, v.y, v.z, 0.0f, h);
    return v3(h[0], h[1], h[2]);
}

// ------------------------------------------------------------------ utils.rs
// utils.rs:5-13
inline float clampf(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
// utils.rs:15-17
inline V3 reflect(V3 v, V3 n) { return normalize(v - 2.0f * dot(v, n) * n); }
// utils.rs:46-59
inline V3 refract(V3 v, V3 normal, float refraction_index) {
    float n = dot(v, normal) <= 0.0f ? 1.0f / refraction_index : refraction_index / 1.0f;
    float cos_in = std::fabs(dot(v, normal));
    float cos_out = std::sqrt(1.0f - std::pow(n, 2.0f) * (1.0f - std::pow(cos_in, 2.0f)));
    return normalize(v * n + (n * cos_in - cos_out) * normal);
}
// utils.rs:20-29
inline void component_wise_range(const V3 *pts, size_t n, V3 &mn, V3 &mx) {
    const float big = std::numeric_limits<float>::max();
    mn = v3(big, big, big);
    mx = v3(-big, -big, -big);
    for (size_t i = 0; i < n; i++) {
        mn = v3(std::fmin(mn.x, pts[i].x), std::fmin(mn.y, pts[i].y), std::fmin(mn.z, pts[i].z));
        mx = v3(std::fmax(mx.x, pts[i].x), std::fmax(mx.y, pts[i].y), std::fmax(mx.z, pts[i].z));
    }
}
// utils.rs:33-44
inline V3 get_axis_scaling(const M4 &M) {
    return v3(magnitude(transform_vector(M, v3(1, 0, 0))),
              magnitude(transform_vector(M, v3(0, 1, 0))),
              magnitude(transform_vector(M, v3(0, 0, 1))));
}

// ------------------------------------------------------------------ color.rs
struct Color {
    float r, g, b;  // alpha is carried by the reference but never reaches the image
};
// color.rs:17-23
inline Color rgb(float r, float g, float b) {
    return Color{clampf(r, 0, 1), clampf(g, 0, 1), clampf(b, 0, 1)};
}
inline Color operator+(Color a, Color b) { return rgb(a.r + b.r, a.g + b.g, a.b + b.b); }  // 67-78
inline Color operator*(Color a, Color b) { return rgb(a.r * b.r, a.g * b.g, a.b * b.b); }  // 80-90
inline Color operator*(Color a, float s) { return rgb(a.r * s, a.g * s, a.b * s); }        // 92-98
inline Color operator*(float s, Color a) { return a * s; }                                 // 108-113
// color.rs:25-31: `as u8` saturates, truncates, NaN -> 0
inline uint8_t to_u8(float c) {
    float v = c * 255.0f;
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

// ------------------------------------------------------------------ ray.rs
struct Ray {
    V3 position, direction;
    static Ray make(V3 p, V3 d) { return Ray{p, normalize(d)}; }            // ray.rs:11-16
    Ray transform_using(const M4 &M) const {                                // ray.rs:18-23
        return Ray::make(transform_point(M, position), transform_vector(M, direction));
    }
    V3 get_point_on_ray(float t) const { return position + t * direction; } // ray.rs:25-28
    float get_t(V3 point) const { return dot(point - position, direction); } // ray.rs:31-33
    Ray offset(float eps) const { return Ray::make(position + eps * direction, direction); }  // 43-46
};

// ------------------------------------------------------------------ jitter
// Counter-based replacement of rand::thread_rng() (camera.rs:36, world.rs:50), shared
// by definition (not by code) with the HIP raygen kernel: PCG-RXS-M-XS output hash.
inline uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
inline uint32_t jitter_u32(uint64_t seed, uint32_t px, uint32_t py, uint32_t sample, uint32_t dim) {
    uint32_t h = pcg((uint32_t)seed ^ 0x9E3779B9u);
    h = pcg(h ^ (uint32_t)(seed >> 32));
    h = pcg(h + px);
    h = pcg(h + py);
    h = pcg(h + sample * 2u + dim);
    return h;
}
// rand 0.7.3 Standard f32: (next_u32() >> 8) * 2^-24
inline float u32_to_f32(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// ------------------------------------------------------------------ scene data
enum { OBJ_SPHERE = 0, OBJ_TRIANGLE = 1, OBJ_QUAD = 2 };
enum { MAT_NONE = 0, MAT_PHONG = 1, MAT_REFLECTIVE = 2, MAT_REFRACTIVE = 3, MAT_COMPOSITION = 4 };
enum { TEX_NONE = 0, TEX_FLAT = 1, TEX_IMAGE = 2 };
enum { LIGHT_AMBIENT = 0, LIGHT_POINT = 1, LIGHT_DIRECTIONAL = 2, LIGHT_CONE = 3 };

struct Xform {
    M4 o2w, w2o;
};
struct Object {
    int type;
    V3 a, b, c, d;  // sphere: a = centre
    float radius;
    const Xform *xf;
    int material;
};
struct MatNode {
    int type;
    float p0, p1, p2;
    int child_first, child_count;
};
struct Material {
    int root, tex_kind, tex_id;
    Color tex_color;
};
struct Texture {
    int w, h;
    const uint8_t *data;
};
struct Light {
    int type;
    V3 pos, dir;
    float angle_deg;
    Color color;
};

// ------------------------------------------------------------------ bvh.rs AABB
struct AABB {
    V3 mn, mx;
};
inline AABB aabb_empty() { return AABB{v3(0, 0, 0), v3(0, 0, 0)}; }  // bvh.rs:54-56

// bvh.rs:61-112
struct Interval {
    int kind;  // 0 Infinite, 1 Closed, 2 Empty
    float a, b;
};
inline Interval interval_new(float a, float b, float x, float slope) {  // bvh.rs:71-86
    if (slope == 0.0f) {
        if (a <= x && x <= b) return Interval{0, 0, 0};
        return Interval{2, 0, 0};
    }
    float ta = (a - x) / slope;
    float tb = (b - x) / slope;
    return Interval{1, std::fmin(ta, tb), std::fmax(ta, tb)};
}
inline Interval interval_intersect(Interval self, Interval other) {  // bvh.rs:89-111
    if (self.kind == 0) return other;
    if (self.kind == 2) return Interval{2, 0, 0};
    if (other.kind == 0) return self;
    if (other.kind == 2) return Interval{2, 0, 0};
    float x = std::fmax(self.a, other.a);
    float y = std::fmin(self.b, other.b);
    if (x <= y) return Interval{1, x, y};
    return Interval{2, 0, 0};
}
// bvh.rs:60-131
inline bool aabb_intersect(const AABB &bb, const Ray &ray, float *t_out) {
    V3 p = ray.get_point_on_ray(0.0f);
    V3 d = ray.direction;
    Interval xi = interval_new(bb.mn.x, bb.mx.x, p.x, d.x);
    Interval yi = interval_new(bb.mn.y, bb.mx.y, p.y, d.y);
    Interval zi = interval_new(bb.mn.z, bb.mx.z, p.z, d.z);
    Interval ti = interval_intersect(xi, interval_intersect(yi, zi));
    if (ti.kind == 1) {
        if (t_out) *t_out = ti.a < 0.0f ? ti.b : ti.a;
        return true;
    }
    // kind 0 is unreachable!() in the reference (a normalised direction has a non-zero
    // component); treat it as a hit with t = 0 rather than aborting the test harness.
    if (ti.kind == 0) {
        if (t_out) *t_out = 0.0f;
        return true;
    }
    return false;
}
// bvh.rs:134-145 (min/max are exact, so folding pairwise equals the reference's one-shot range)
inline AABB aabb_union(const AABB &a, const AABB &b) {
    V3 pts[4] = {a.mn, a.mx, b.mn, b.mx};
    AABB r;
    component_wise_range(pts, 4, r.mn, r.mx);
    return r;
}
// bvh.rs:147-150
inline float surface_area(const AABB &bb) {
    V3 diff = bb.mx - bb.mn;
    return 2.0f * diff.x * diff.y + 2.0f * diff.x * diff.z + 2.0f * diff.z * diff.y;
}

// ------------------------------------------------------------------ object.rs
// object.rs:145-220
inline bool object_intersect(const Object &o, const Ray &ray, float *t_out) {
    Ray osr = ray.transform_using(o.xf->w2o);
    V3 position = osr.get_point_on_ray(0.0f);
    V3 direction = osr.direction;
    float t;
    bool hit = false;
    if (o.type == OBJ_SPHERE) {
        V3 center = o.a;
        float tc = dot(center - position, direction);
        V3 closest = osr.get_point_on_ray(tc);
        float radius_sqrd = std::pow(o.radius, 2.0f);
        float dist_sqrd = magnitude2(center - closest);
        if (dist_sqrd <= radius_sqrd) {
            float delta = std::sqrt(radius_sqrd - dist_sqrd);
            float c0 = tc - delta, c1 = tc + delta;
            // filter(is_sign_positive).min_by(partial_cmp or Equal): first minimum wins
            bool p0 = !std::signbit(c0), p1 = !std::signbit(c1);
            if (p0 && p1) {
                t = (c0 > c1) ? c1 : c0;
                hit = true;
            } else if (p0) {
                t = c0;
                hit = true;
            } else if (p1) {
                t = c1;
                hit = true;
            }
        }
    } else if (o.type == OBJ_QUAD) {
        V3 a = o.a, b = o.b, c = o.c, d = o.d;
        V3 normal = normalize(cross(b - a, d - a));
        if (dot(direction, normal) < 0.0f) {
            float tt = dot(a - position, normal) / dot(direction, normal);
            if (tt > 0.0f) {
                V3 q = osr.get_point_on_ray(tt);
                float s0 = dot(cross(b - a, q - a), normal);
                float s1 = dot(cross(c - b, q - b), normal);
                float s2 = dot(cross(d - c, q - c), normal);
                float s3 = dot(cross(a - d, q - d), normal);
                if (!std::signbit(s0) && !std::signbit(s1) && !std::signbit(s2) && !std::signbit(s3)) {
                    t = tt;
                    hit = true;
                }
            }
        }
    } else {
        V3 a = o.a, b = o.b, c = o.c;
        V3 normal = normalize(cross(b - a, c - a));
        if (dot(direction, normal) < 0.0f) {
            float tt = dot(a - position, normal) / dot(direction, normal);
            if (tt > 0.0f) {
                V3 q = osr.get_point_on_ray(tt);
                float s0 = dot(cross(b - a, q - a), normal);
                float s1 = dot(cross(c - b, q - b), normal);
                float s2 = dot(cross(a - c, q - c), normal);
                if (!std::signbit(s0) && !std::signbit(s1) && !std::signbit(s2)) {
                    t = tt;
                    hit = true;
                }
            }
        }
    }
    if (!hit) return false;
    V3 osp = osr.get_point_on_ray(t);                 // object.rs:214
    V3 wp = transform_point(o.xf->o2w, osp);          // object.rs:216-217
    *t_out = ray.get_t(wp);                           // object.rs:218
    return true;
}
// object.rs:225-235
inline V3 object_normal(const Object &o, V3 point) {
    V3 p = transform_point(o.xf->w2o, point);
    V3 n;
    if (o.type == OBJ_SPHERE) n = normalize(p - o.a);
    else if (o.type == OBJ_QUAD) n = normalize(cross(o.b - o.a, o.d - o.a));
    else n = normalize(cross(o.b - o.a, o.c - o.a));
    return normalize(transform_vector(o.xf->o2w, n));
}
// object.rs:255-278
inline void object_uv(const Object &o, V3 point, float &u, float &v) {
    V3 p = transform_point(o.xf->w2o, point);
    if (o.type == OBJ_SPHERE) {
        V3 n = normalize(p - o.a);
        float theta = std::acos(n.z);
        float phi = std::atan2(n.y, n.x);
        u = theta / (2.0f * PI_F);
        v = phi / PI_F;
    } else {
        V3 e = (o.type == OBJ_QUAD) ? o.d : o.c;
        u = dot(o.b - o.a, p - o.a) / magnitude2(o.b - o.a);
        v = dot(e - o.a, p - o.a) / magnitude2(e - o.a);
    }
}
// object.rs:290-313
inline AABB object_bbox(const Object &o) {
    const M4 &M = o.xf->o2w;
    AABB r;
    if (o.type == OBJ_SPHERE) {
        V3 center = transform_point(M, o.a);
        V3 rad = o.radius * get_axis_scaling(M);
        r.mn = center - rad;
        r.mx = center + rad;
    } else if (o.type == OBJ_QUAD) {
        V3 pts[4] = {transform_point(M, o.a), transform_point(M, o.b), transform_point(M, o.c),
                     transform_point(M, o.d)};
        component_wise_range(pts, 4, r.mn, r.mx);
    } else {
        V3 pts[3] = {transform_point(M, o.a), transform_point(M, o.b), transform_point(M, o.c)};
        component_wise_range(pts, 3, r.mn, r.mx);
    }
    return r;
}

// ------------------------------------------------------------------ bvh.rs tree
struct BvhTree {
    bool leaf;
    AABB aabb;
    std::unique_ptr<BvhTree> left, right;
    std::vector<uint32_t> objects;  // leaf: object indices in Vec order
    size_t size;
};
const int N_BUCKETS = 12;       // bvh.rs:241
const size_t BVH_LEAF_SIZE = 4; // bvh.rs:331

struct Builder {
    const std::vector<Object> &objs;
    std::vector<AABB> boxes;      // Object::get_bounding_box, computed once (pure function)
    std::vector<V3> centroids;    // Point3::centroid(&[min,max]) (bvh.rs:200)
    explicit Builder(const std::vector<Object> &o) : objs(o) {
        boxes.resize(o.size());
        centroids.resize(o.size());
        for (size_t i = 0; i < o.size(); i++) {
            boxes[i] = object_bbox(o[i]);
            V3 sum = (v3(0, 0, 0) + boxes[i].mn) + boxes[i].mx;
            centroids[i] = v3(sum.x / 2.0f, sum.y / 2.0f, sum.z / 2.0f);
        }
    }

    // bvh.rs:195-231; kind 2 = Basic, 3 = SAH
    void split(const std::vector<uint32_t> &objects, int kind, std::vector<uint32_t> &left,
               std::vector<uint32_t> &right) {
        std::vector<V3> cents(objects.size());
        for (size_t i = 0; i < objects.size(); i++) cents[i] = centroids[objects[i]];
        V3 min_c, max_c;
        component_wise_range(cents.data(), cents.size(), min_c, max_c);
        V3 diff = max_c - min_c;
        int maxdim = 0;
        float mx = diff.x;
        if (diff.y > mx) {
            mx = diff.y;
            maxdim = 1;
        }
        if (diff.z > mx) maxdim = 2;
        float midpoint = (max_c[maxdim] + min_c[maxdim]) / 2.0f;
        if (objects.size() <= 4 || kind == 2) {
            for (size_t i = 0; i < objects.size(); i++) {
                if (cents[i][maxdim] < midpoint) left.push_back(objects[i]);
                else right.push_back(objects[i]);
            }
        } else {
            split_by_sah(objects, cents, AABB{min_c, max_c}, maxdim, left, right);
        }
    }

    // bvh.rs:245-321
    void split_by_sah(const std::vector<uint32_t> &objects, const std::vector<V3> &cents,
                      AABB global_bb, int dim, std::vector<uint32_t> &left,
                      std::vector<uint32_t> &right) {
        uint32_t count[N_BUCKETS] = {0};
        AABB bb[N_BUCKETS];
        for (int i = 0; i < N_BUCKETS; i++) bb[i] = aabb_empty();
        std::vector<uint8_t> bucket_of(objects.size());
        float dim_range = global_bb.mx[dim] - global_bb.mn[dim];
        for (size_t i = 0; i < objects.size(); i++) {
            float b = 12.0f * (cents[i][dim] - global_bb.mn[dim]) / dim_range;
            float bt = std::trunc(b);
            int b_ind;  // `as i32`: saturating, NaN -> 0
            if (!(bt == bt)) b_ind = 0;
            else if (bt >= 2147483648.0f) b_ind = 2147483647;
            else if (bt <= -2147483648.0f) b_ind = -2147483647 - 1;
            else b_ind = (int)bt;
            if (b_ind == N_BUCKETS) b_ind -= 1;
            if (b_ind < 0 || b_ind >= N_BUCKETS) {
                std::fprintf(stderr, "oracle: SAH bucket index %d out of range (reference would panic)\n", b_ind);
                std::abort();
            }
            count[b_ind] += 1;
            bb[b_ind] = aabb_union(bb[b_ind], boxes[objects[i]]);
            bucket_of[i] = (uint8_t)b_ind;
        }
        float costs[N_BUCKETS - 1];
        for (int i = 0; i < N_BUCKETS - 1; i++) {
            AABB bl = bb[0];
            float cl = 0.0f;
            for (int j = 0; j <= i; j++) {
                if (j > 0) bl = aabb_union(bl, bb[j]);
                cl += (float)count[j];
            }
            AABB br = bb[i + 1];
            float cr = 0.0f;
            for (int j = i + 1; j < N_BUCKETS; j++) {
                if (j > i + 1) br = aabb_union(br, bb[j]);
                cr += (float)count[j];
            }
            costs[i] = 0.125f + (cl * surface_area(bl) + cr * surface_area(br)) / surface_area(global_bb);
        }
        float min_cost = costs[0];
        int min_i = 0;
        for (int i = 0; i < N_BUCKETS - 1; i++) {
            if (costs[i] < min_cost) {
                min_cost = costs[i];
                min_i = i;
            }
        }
        for (size_t i = 0; i < objects.size(); i++) {
            if (bucket_of[i] <= min_i) left.push_back(objects[i]);
            else right.push_back(objects[i]);
        }
    }

    // bvh.rs:334-357
    std::unique_ptr<BvhTree> build(std::vector<uint32_t> objects, int depth) {
        auto node = std::make_unique<BvhTree>();
        node->size = objects.size();
        if (objects.size() <= BVH_LEAF_SIZE) {
            node->leaf = true;
            if (objects.empty()) node->aabb = aabb_empty();
            else {
                node->aabb = boxes[objects[0]];
                for (size_t i = 1; i < objects.size(); i++) node->aabb = aabb_union(node->aabb, boxes[objects[i]]);
            }
            node->objects = std::move(objects);
            return node;
        }
        if (depth > 4096) {
            std::fprintf(stderr, "oracle: BVH recursion does not terminate (identical centroids; the reference overflows its stack here)\n");
            std::abort();
        }
        std::vector<uint32_t> l, r;
        split(objects, 3, l, r);
        objects.clear();
        objects.shrink_to_fit();
        node->leaf = false;
        node->left = build(std::move(l), depth + 1);
        node->right = build(std::move(r), depth + 1);
        node->aabb = aabb_union(node->left->aabb, node->right->aabb);
        return node;
    }
};

struct Hit {
    uint32_t obj;
    float t;
};

}  // namespace

struct orc_scene {
    std::vector<Xform> xforms;
    std::vector<Object> objects;
    std::vector<MatNode> mat_nodes;
    std::vector<int> child_node;
    std::vector<float> child_coef;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::vector<uint8_t> tex_data;
    std::vector<Light> lights;
    Color background;
    M4 camera_to_world;
    uint32_t width, height;
    std::unique_ptr<BvhTree> bvh;
    double build_seconds;
};

namespace {

struct Counters {
    uint64_t primary = 0, bounce = 0, shadow = 0;
    // visits of the traversal below (bvh.rs:359-395), split by who asked: [0] internal nodes entered
    // (own box hit), [1] objects tested (leaf sizes of the leaves whose box was hit)
    uint64_t closest_vis[2] = {0, 0}, shadow_vis[2] = {0, 0};
};

// bvh.rs:359-395
bool bvh_closest(const orc_scene &s, const BvhTree *n, const Ray &ray, Hit *out, uint64_t *vis) {
    if (!aabb_intersect(n->aabb, ray, nullptr)) return false;
    if (!n->leaf) {
        if (vis) vis[0]++;
        Hit hl, hr;
        bool bl = bvh_closest(s, n->left.get(), ray, &hl, vis);
        bool br = bvh_closest(s, n->right.get(), ray, &hr, vis);
        if (bl && br) {
            *out = (hl.t > hr.t) ? hr : hl;  // min_by keeps the first unless Greater
            return true;
        }
        if (bl) {
            *out = hl;
            return true;
        }
        if (br) {
            *out = hr;
            return true;
        }
        return false;
    }
    bool any = false;
    Hit best{0, 0};
    if (vis) vis[1] += n->objects.size();
    for (uint32_t oi : n->objects) {
        float t;
        if (object_intersect(s.objects[oi], ray, &t)) {
            if (!any) {
                best = Hit{oi, t};
                any = true;
            } else if (best.t > t) {
                best = Hit{oi, t};
            }
        }
    }
    if (any) *out = best;
    return any;
}
// bvh.rs:33-35
inline bool get_closest_intersection(const orc_scene &s, const Ray &ray, Hit *out, uint64_t *vis = nullptr) {
    return bvh_closest(s, s.bvh.get(), ray, out, vis);
}

// light.rs:81-97
bool in_shadow(const orc_scene &s, V3 point, V3 light_position, V3 light_direction, Counters &c) {
    Ray light_ray = Ray::make(light_position, light_direction);
    float light_to_point_t = magnitude(light_position - point);
    c.shadow++;
    Hit h;
    if (get_closest_intersection(s, light_ray, &h, c.shadow_vis)) {
        float epsilon = 1e-4f;
        bool is_in_shadow = h.t + epsilon < light_to_point_t;
        return !is_in_shadow;
    }
    return false;
}
// light.rs:99-122
bool reaches_point(const orc_scene &s, const Light &l, V3 point, Counters &c) {
    switch (l.type) {
        case LIGHT_AMBIENT: return true;
        case LIGHT_POINT: return in_shadow(s, point, l.pos, point - l.pos, c);
        case LIGHT_DIRECTIONAL: {
            Ray r = Ray::make(point, -l.dir).offset(1e-4f);
            c.shadow++;
            Hit h;
            return !get_closest_intersection(s, r, &h, c.shadow_vis);
        }
        default: {
            V3 light_direction = point - l.pos;
            if (angle_rad(l.dir, light_direction) > l.angle_deg * DEG_TO_RAD) return false;
            return in_shadow(s, point, l.pos, light_direction, c);
        }
    }
}
// light.rs:58-79
float get_falloff_at(const Light &l, V3 point) {
    if (l.type == LIGHT_POINT) {
        float d2 = magnitude2(l.pos - point);
        return 5.0f / (0.001f + d2);
    }
    float angle_to_point = angle_rad(l.dir, point - l.pos) * RAD_TO_DEG;
    float angle_delta = l.angle_deg - angle_to_point;
    float cone_falloff = clampf(angle_delta / 8.0f, 0.0f, 1.0f);
    float d2 = magnitude2(l.pos - point);
    return cone_falloff * 5.0f / (0.001f + d2);
}

// material.rs:53-76
Color texture_sample(const orc_scene &s, const Material &m, const Object &o, V3 point) {
    if (m.tex_kind == TEX_IMAGE) {
        const Texture &tx = s.textures[m.tex_id];
        float width = (float)tx.w, height = (float)tx.h;
        float u, v;
        object_uv(o, point, u, v);
        // f32::rem_euclid(1.0): r = x % 1; if r < 0 { r + |1| }
        float ru = std::fmod(u, 1.0f);
        if (ru < 0.0f) ru = ru + 1.0f;
        float rv = std::fmod(v, 1.0f);
        if (rv < 0.0f) rv = rv + 1.0f;
        float x = std::trunc(ru * width), y = std::trunc(rv * height);
        float xc = clampf(x, 0.0f, width - 1.0f), yc = clampf(y, 0.0f, height - 1.0f);
        uint32_t xi = (xc == xc) ? (uint32_t)xc : 0u, yi = (yc == yc) ? (uint32_t)yc : 0u;
        const uint8_t *px = tx.data + ((size_t)yi * tx.w + xi) * 3;
        return rgb((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
    }
    if (m.tex_kind == TEX_FLAT) return m.tex_color;
    return rgb(0.5f, 0.5f, 0.5f);
}

// material.rs:98-113
float phong_multiple(V3 dir_to_light, V3 normal, V3 incoming, float diffuse, float specular,
                     float shininess) {
    V3 r = reflect(dir_to_light, normal);
    float spec_i = clampf(-dot(r, incoming), 0.0f, 1.0f);
    float diff_i = clampf(dot(dir_to_light, normal), 0.0f, 1.0f);
    return diffuse * diff_i + specular * std::pow(spec_i, shininess);
}

Color trace_ray(const orc_scene &s, const Ray &ray, uint32_t max_depth, bool primary, Counters &c);

// material.rs:118-206
Color material_color(const orc_scene &s, int node, Color surface, const Ray &ray, float t,
                     const Object &o, const std::vector<const Light *> &lights, uint32_t max_depth,
                     Counters &c) {
    const MatNode &m = s.mat_nodes[node];
    switch (m.type) {
        case MAT_COMPOSITION: {
            Color acc{0, 0, 0};
            for (int k = 0; k < m.child_count; k++) {
                int cn = s.child_node[m.child_first + k];
                float coef = s.child_coef[m.child_first + k];
                Color sub = material_color(s, cn, surface, ray, t, o, lights, max_depth, c);
                acc = acc + coef * sub;
            }
            return acc;
        }
        case MAT_PHONG: {
            V3 p = ray.get_point_on_ray(t);
            V3 normal = object_normal(o, p);
            Color acc{0, 0, 0};
            for (const Light *l : lights) {
                Color lc;
                if (l->type == LIGHT_AMBIENT) lc = l->color;
                else if (l->type == LIGHT_DIRECTIONAL) {
                    float pm = phong_multiple(-l->dir, normal, ray.direction, m.p0, m.p1, m.p2);
                    lc = pm * l->color;
                } else {
                    V3 light_dir = l->pos - p;
                    float falloff = get_falloff_at(*l, p);
                    float pm = phong_multiple(normalize(light_dir), normal, ray.direction, m.p0, m.p1, m.p2);
                    lc = pm * (falloff * l->color);
                }
                acc = acc + surface * lc;
            }
            return acc;
        }
        case MAT_REFLECTIVE: {
            V3 p = ray.get_point_on_ray(t);
            V3 normal = object_normal(o, p);
            Ray r = Ray::make(p, reflect(ray.direction, normal)).offset(1e-4f);
            return trace_ray(s, r, max_depth, false, c);
        }
        case MAT_REFRACTIVE: {
            V3 p = ray.get_point_on_ray(t);
            V3 normal = object_normal(o, p);
            Ray r = Ray::make(p, refract(ray.direction, normal, m.p0)).offset(1e-4f);
            return trace_ray(s, r, max_depth, false, c);
        }
        default: return rgb(0.5f, 0.5f, 0.5f);
    }
}

// world.rs:115-131
Color trace_ray(const orc_scene &s, const Ray &ray, uint32_t max_depth, bool primary, Counters &c) {
    if (max_depth == 0) return s.background;
    if (primary) c.primary++;
    else c.bounce++;
    Hit h;
    if (!get_closest_intersection(s, ray, &h, c.closest_vis)) return s.background;
    const Object &o = s.objects[h.obj];
    V3 p = ray.get_point_on_ray(h.t);
    std::vector<const Light *> lit;
    for (const Light &l : s.lights)
        if (reaches_point(s, l, p, c)) lit.push_back(&l);
    // object.rs:240-250 -> material.rs:220-240
    const Material &m = s.materials[o.material];
    Color surface = texture_sample(s, m, o, ray.get_point_on_ray(h.t));
    return material_color(s, m.root, surface, ray, h.t, o, lit, max_depth - 1, c);
}

// camera.rs:32-47
Ray generate_ray(const orc_scene &s, uint32_t px, uint32_t py, bool use_jitter, uint64_t seed,
                 uint32_t sample) {
    float dx = 0.0f, dy = 0.0f;
    if (use_jitter) {
        dx = u32_to_f32(jitter_u32(seed, px, py, sample, 0)) / 2.0f;
        dy = u32_to_f32(jitter_u32(seed, px, py, sample, 1)) / 2.0f;
    }
    float x = ((float)px + dx) / (float)s.width - 0.5f;
    float y = 0.5f - ((float)py + dy) / (float)s.height;
    float dist = -1.0f;
    Ray r = Ray::make(v3(x, y, dist), v3(x, y, dist));
    return r.transform_using(s.camera_to_world);
}

template <class F>
void parallel_jobs(uint32_t njobs, int threads, F f) {
    if (threads < 1) threads = 1;
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t]() {
            for (;;) {
                uint32_t j = next.fetch_add(1);
                if (j >= njobs) break;
                f(j, t);
            }
        });
    for (auto &th : pool) th.join();
}

uint32_t tree_depth(const BvhTree *n) {  // bvh.rs:404-409
    if (n->leaf) return 0;
    uint32_t a = tree_depth(n->left.get()), b = tree_depth(n->right.get());
    return 1 + (a > b ? a : b);
}
float tree_total_sa(const BvhTree *n) {  // bvh.rs:419-426
    if (n->leaf) return surface_area(n->aabb);
    return surface_area(n->aabb) + tree_total_sa(n->left.get()) + tree_total_sa(n->right.get());
}
void tree_digest(const BvhTree *n, uint32_t *leaf_order, uint32_t &nl, uint32_t *pre, uint32_t &np) {
    if (pre) pre[np] = (uint32_t)n->size;
    np++;
    if (n->leaf) {
        for (uint32_t o : n->objects) {
            if (leaf_order) leaf_order[nl] = o;
            nl++;
        }
    } else {
        tree_digest(n->left.get(), leaf_order, nl, pre, np);
        tree_digest(n->right.get(), leaf_order, nl, pre, np);
    }
}

M4 load_m4(const float *p) {
    M4 m;
    std::memcpy(m.m, p, sizeof(m.m));
    return m;
}
V3 load_v3(const float *p) { return v3(p[0], p[1], p[2]); }

}  // namespace

extern "C" {

orc_scene *orc_scene_create(const orc_scene_desc *d) {
    auto s = new orc_scene();
    s->xforms.resize(d->n_xforms);
    for (uint32_t i = 0; i < d->n_xforms; i++) {
        s->xforms[i].o2w = load_m4(d->xforms + (size_t)i * 32);
        s->xforms[i].w2o = load_m4(d->xforms + (size_t)i * 32 + 16);
    }
    s->objects.resize(d->n_objects);
    for (uint32_t i = 0; i < d->n_objects; i++) {
        const float *v = d->obj_verts + (size_t)i * 12;
        Object &o = s->objects[i];
        o.type = d->obj_type[i];
        o.a = load_v3(v);
        o.b = load_v3(v + 3);
        o.c = load_v3(v + 6);
        o.d = load_v3(v + 9);
        o.radius = v[3];
        o.xf = &s->xforms[d->obj_xform[i]];
        o.material = d->obj_material[i];
    }
    s->mat_nodes.resize(d->n_mat_nodes);
    for (uint32_t i = 0; i < d->n_mat_nodes; i++)
        s->mat_nodes[i] = MatNode{d->node_type[i], d->node_params[i * 3], d->node_params[i * 3 + 1],
                                  d->node_params[i * 3 + 2], d->node_child_first[i], d->node_child_count[i]};
    s->child_node.assign(d->child_node, d->child_node + d->n_mat_children);
    s->child_coef.assign(d->child_coef, d->child_coef + d->n_mat_children);
    s->materials.resize(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; i++)
        s->materials[i] = Material{d->mat_root[i], d->mat_tex_kind[i], d->mat_tex_id[i],
                                   rgb(d->mat_tex_color[i * 3], d->mat_tex_color[i * 3 + 1], d->mat_tex_color[i * 3 + 2])};
    size_t tex_bytes = 0;
    for (uint32_t i = 0; i < d->n_textures; i++) {
        size_t end = d->tex_offset[i] + (size_t)d->tex_w[i] * d->tex_h[i] * 3;
        if (end > tex_bytes) tex_bytes = end;
    }
    if (tex_bytes) s->tex_data.assign(d->tex_data, d->tex_data + tex_bytes);
    s->textures.resize(d->n_textures);
    for (uint32_t i = 0; i < d->n_textures; i++)
        s->textures[i] = Texture{d->tex_w[i], d->tex_h[i], s->tex_data.data() + d->tex_offset[i]};
    s->lights.resize(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; i++)
        s->lights[i] = Light{d->light_type[i], load_v3(d->light_pos + i * 3), load_v3(d->light_dir + i * 3),
                             d->light_angle_deg[i],
                             rgb(d->light_color[i * 3], d->light_color[i * 3 + 1], d->light_color[i * 3 + 2])};
    s->background = rgb(d->background[0], d->background[1], d->background[2]);
    s->camera_to_world = load_m4(d->camera_to_world);
    s->width = d->width;
    s->height = d->height;
    // world.rs:104 -> bvh.rs:13-26
    auto t0 = std::chrono::steady_clock::now();
    Builder b(s->objects);
    std::vector<uint32_t> all(s->objects.size());
    for (size_t i = 0; i < all.size(); i++) all[i] = (uint32_t)i;
    s->bvh = b.build(std::move(all), 0);
    s->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return s;
}

void orc_scene_destroy(orc_scene *s) { delete s; }

int orc_render(orc_scene *s, uint32_t spp, uint32_t bounces, uint64_t seed, int threads,
               uint32_t x0, uint32_t x1, uint32_t xs, uint32_t y0, uint32_t y1, uint32_t ys,
               uint8_t *out_rgb8, float *out_rgb_f32, orc_stats *stats) {
    if (spp == 0 || xs == 0 || ys == 0) return -1;
    if (x1 > s->width) x1 = s->width;
    if (y1 > s->height) y1 = s->height;
    std::vector<uint32_t> cols;
    for (uint32_t x = x0; x < x1; x += xs) cols.push_back(x);
    std::vector<Counters> ctr(threads < 1 ? 1 : threads);
    auto t0 = std::chrono::steady_clock::now();
    // world.rs:38-68: one job per column
    parallel_jobs((uint32_t)cols.size(), threads, [&](uint32_t j, int tid) {
        uint32_t x = cols[j];
        Counters &c = ctr[tid];
        for (uint32_t y = y0; y < y1; y += ys) {
            float sum[3] = {0, 0, 0};
            for (uint32_t sidx = 0; sidx < spp; sidx++) {
                Ray ray = generate_ray(*s, x, y, spp != 1, seed, sidx);
                Color col = trace_ray(*s, ray, bounces, true, c);
                sum[0] = sum[0] + col.r;
                sum[1] = sum[1] + col.g;
                sum[2] = sum[2] + col.b;
            }
            float n = (float)spp;
            Color res = rgb(sum[0] / n, sum[1] / n, sum[2] / n);  // world.rs:62-63
            size_t idx = ((size_t)y * s->width + x) * 3;
            if (out_rgb_f32) {
                out_rgb_f32[idx] = res.r;
                out_rgb_f32[idx + 1] = res.g;
                out_rgb_f32[idx + 2] = res.b;
            }
            if (out_rgb8) {
                out_rgb8[idx] = to_u8(res.r);
                out_rgb8[idx + 1] = to_u8(res.g);
                out_rgb8[idx + 2] = to_u8(res.b);
            }
        }
    });
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (auto &c : ctr) {
            stats->primary_rays += c.primary;
            stats->bounce_rays += c.bounce;
            stats->shadow_rays += c.shadow;
            stats->closest_nodes += c.closest_vis[0];
            stats->closest_prims += c.closest_vis[1];
            stats->shadow_nodes += c.shadow_vis[0];
            stats->shadow_prims += c.shadow_vis[1];
        }
        stats->build_seconds = s->build_seconds;
        stats->trace_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

int orc_trace_primary(orc_scene *s, uint32_t sample, int use_jitter, uint64_t seed, int threads,
                      uint32_t *out_obj, float *out_t) {
    parallel_jobs(s->width, threads, [&](uint32_t x, int) {
        for (uint32_t y = 0; y < s->height; y++) {
            Ray ray = generate_ray(*s, x, y, use_jitter != 0, seed, sample);
            Hit h;
            size_t idx = (size_t)y * s->width + x;
            if (get_closest_intersection(*s, ray, &h)) {
                out_obj[idx] = h.obj;
                out_t[idx] = h.t;
            } else {
                out_obj[idx] = 0xFFFFFFFFu;
                out_t[idx] = 0.0f;
            }
        }
    });
    return 0;
}

int orc_trace_rays(orc_scene *s, const float *od, uint64_t n, int threads, uint32_t *out_obj,
                   float *out_t) {
    const uint32_t chunk = 1024;
    uint32_t njobs = (uint32_t)((n + chunk - 1) / chunk);
    parallel_jobs(njobs, threads, [&](uint32_t j, int) {
        uint64_t e = (uint64_t)(j + 1) * chunk;
        if (e > n) e = n;
        for (uint64_t i = (uint64_t)j * chunk; i < e; i++) {
            Ray ray = Ray::make(load_v3(od + i * 6), load_v3(od + i * 6 + 3));
            Hit h;
            if (get_closest_intersection(*s, ray, &h)) {
                out_obj[i] = h.obj;
                out_t[i] = h.t;
            } else {
                out_obj[i] = 0xFFFFFFFFu;
                out_t[i] = 0.0f;
            }
        }
    });
    return 0;
}

int orc_shade_rays(orc_scene *s, const float *od, uint64_t n, uint32_t depth, float *out_rgb) {
    Counters c;
    for (uint64_t i = 0; i < n; i++) {
        Ray ray = Ray::make(load_v3(od + i * 6), load_v3(od + i * 6 + 3));
        Color col = trace_ray(*s, ray, depth, true, c);
        out_rgb[i * 3] = col.r;
        out_rgb[i * 3 + 1] = col.g;
        out_rgb[i * 3 + 2] = col.b;
    }
    return 0;
}

uint32_t orc_bvh_digest(orc_scene *s, uint32_t *leaf_order, uint32_t *preorder_sizes,
                        uint32_t *depth, float *total_sa) {
    uint32_t nl = 0, np = 0;
    tree_digest(s->bvh.get(), leaf_order, nl, preorder_sizes, np);
    if (depth) *depth = tree_depth(s->bvh.get());
    if (total_sa) *total_sa = tree_total_sa(s->bvh.get());
    return np;
}

void orc_generate_ray(orc_scene *s, uint32_t px, uint32_t py, uint32_t sample, int use_jitter,
                      uint64_t seed, float *out) {
    Ray r = generate_ray(*s, px, py, use_jitter != 0, seed, sample);
    out[0] = r.position.x;
    out[1] = r.position.y;
    out[2] = r.position.z;
    out[3] = r.direction.x;
    out[4] = r.direction.y;
    out[5] = r.direction.z;
}

float orc_aabb_surface_area(const float *mn, const float *mx) {
    return surface_area(AABB{load_v3(mn), load_v3(mx)});
}
int orc_aabb_intersect(const float *mn, const float *mx, const float *org, const float *dir, float *t) {
    Ray r = Ray::make(load_v3(org), load_v3(dir));
    return aabb_intersect(AABB{load_v3(mn), load_v3(mx)}, r, t) ? 1 : 0;
}
float orc_ray_get_t(const float *org, const float *dir, const float *point) {
    return Ray::make(load_v3(org), load_v3(dir)).get_t(load_v3(point));
}
int orc_object_intersect(orc_scene *s, uint32_t obj, const float *org, const float *dir, float *t) {
    Ray r = Ray::make(load_v3(org), load_v3(dir));
    return object_intersect(s->objects[obj], r, t) ? 1 : 0;
}
void orc_object_bbox(orc_scene *s, uint32_t obj, float *mn, float *mx) {
    AABB b = object_bbox(s->objects[obj]);
    mn[0] = b.mn.x; mn[1] = b.mn.y; mn[2] = b.mn.z;
    mx[0] = b.mx.x; mx[1] = b.mx.y; mx[2] = b.mx.z;
}
void orc_bvh_split(orc_scene *s, int kind, uint32_t *n_left, uint32_t *n_right) {
    size_t n = s->objects.size();
    if (kind == 0 || kind == 1) {  // bvh.rs:160-178: both halve at len/2 (by_x sorts first)
        *n_left = (uint32_t)(n / 2);
        *n_right = (uint32_t)(n - n / 2);
        return;
    }
    Builder b(s->objects);
    std::vector<uint32_t> all(n), l, r;
    for (size_t i = 0; i < n; i++) all[i] = (uint32_t)i;
    b.split(all, kind, l, r);
    *n_left = (uint32_t)l.size();
    *n_right = (uint32_t)r.size();
}
float orc_clamp(float x, float lo, float hi) { return clampf(x, lo, hi); }
void orc_reflect(const float *v, const float *n, float *out) {
    V3 r = reflect(load_v3(v), load_v3(n));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_refract(const float *v, const float *n, float idx, float *out) {
    V3 r = refract(load_v3(v), load_v3(n), idx);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_component_wise_range(const float *pts, uint32_t n, float *mn, float *mx) {
    std::vector<V3> p(n);
    for (uint32_t i = 0; i < n; i++) p[i] = load_v3(pts + i * 3);
    V3 a, b;
    component_wise_range(p.data(), n, a, b);
    mn[0] = a.x; mn[1] = a.y; mn[2] = a.z;
    mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
}
void orc_axis_scaling(const float *m16, float *out3) {
    V3 r = get_axis_scaling(load_m4(m16));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
uint32_t orc_jitter_u32(uint64_t seed, uint32_t px, uint32_t py, uint32_t sample, uint32_t dim) {
    return jitter_u32(seed, px, py, sample, dim);
}

}  // extern "C"
