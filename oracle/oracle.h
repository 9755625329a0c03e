/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C entry points of the CPU restatement of ellishg/rust-raytracer's trace path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (rust-raytracer_amd/) never links or calls it.
 *
 * Parity status: the reference is Rust and cannot be built here (no cargo/rustc,
 * crates not vendored), so this restatement is pinned by the reference's own 13
 * inline unit tests (src/bvh.rs:437-577, src/object.rs:323-407,
 * src/utils.rs:66-108) restated in tests/test_oracle_kat.py.  Everything those
 * tests do not cover (camera, lights, materials, colour algebra, trace recursion)
 * is "parity unpinned": written directly from the cited lines, see DESIGN.md.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Scene description: plain arrays, the same layout the product's rtx_scene_desc
 * uses (include/rtx.h) so tests can hand identical inputs to both sides. */
typedef struct orc_scene_desc {
    uint32_t n_objects;
    const int32_t *obj_type;      /* 0 sphere, 1 triangle, 2 quad  (object.rs:14-18) */
    const float *obj_verts;       /* n*12: sphere = centre xyz + radius; tri = a,b,c; quad = a,b,c,d */
    const int32_t *obj_xform;     /* n: index into xforms */
    const int32_t *obj_material;  /* n: index into materials */
    uint32_t n_xforms;
    const float *xforms;          /* n_xforms*32: object_to_world (16, column-major) then world_to_object (16) */
    uint32_t n_materials;
    const int32_t *mat_root;      /* root material node */
    const int32_t *mat_tex_kind;  /* 0 None, 1 Flat, 2 Image (material.rs:14-18) */
    const float *mat_tex_color;   /* n*3 */
    const int32_t *mat_tex_id;
    uint32_t n_mat_nodes;
    const int32_t *node_type;     /* 0 None, 1 Phong, 2 Reflective, 3 Refractive, 4 Composition */
    const float *node_params;     /* n*3: Phong diffuse,specular,shininess | Refractive index */
    const int32_t *node_child_first;
    const int32_t *node_child_count;
    uint32_t n_mat_children;
    const int32_t *child_node;
    const float *child_coef;
    uint32_t n_textures;
    const int32_t *tex_w;
    const int32_t *tex_h;
    const uint64_t *tex_offset;   /* byte offset of each RGB8 image in tex_data */
    const uint8_t *tex_data;
    uint32_t n_lights;
    const int32_t *light_type;    /* 0 Ambient, 1 Point, 2 Directional, 3 Cone (light.rs:12-17) */
    const float *light_pos;       /* n*3 */
    const float *light_dir;       /* n*3, already normalised by the constructor (light.rs:42,54) */
    const float *light_angle_deg; /* n */
    const float *light_color;     /* n*3 */
    float background[3];
    float camera_to_world[16];    /* column-major (camera.rs:20) */
    uint32_t width, height;
} orc_scene_desc;

typedef struct orc_scene orc_scene;

typedef struct orc_stats {
    uint64_t primary_rays, bounce_rays, shadow_rays;  /* one ray = one Bvh::get_closest_intersection */
    double build_seconds, trace_seconds;
    /* visits of BvhTree::get_closest_intersection (bvh.rs:359-395) over those rays: internal nodes whose own
     * box the ray passes, and objects tested (the Vec of every leaf whose box the ray passes); closest = the
     * queries of World::trace_ray (world.rs:118), shadow = those of Light::reaches_point (light.rs:90,111) */
    uint64_t closest_nodes, closest_prims, shadow_nodes, shadow_prims;
} orc_stats;

orc_scene *orc_scene_create(const orc_scene_desc *desc);
void orc_scene_destroy(orc_scene *s);

/* world.rs:17-88 with a column-per-job pool.  Renders pixels x in [x0,x1) step xs,
 * y in [y0,y1) step ys into full-size W*H buffers (other pixels untouched).
 * jitter: spp==1 uses none (world.rs:47-51); spp>1 uses the counter-based jitter. */
int orc_render(orc_scene *s, uint32_t spp, uint32_t bounces, uint64_t seed, int threads,
               uint32_t x0, uint32_t x1, uint32_t xs, uint32_t y0, uint32_t y1, uint32_t ys,
               uint8_t *out_rgb8, float *out_rgb_f32, orc_stats *stats);

/* Closest hit of the primary ray of sample `sample` of every pixel (object index in
 * input order, 0xFFFFFFFF for a miss). use_jitter=0 reproduces spp==1. */
int orc_trace_primary(orc_scene *s, uint32_t sample, int use_jitter, uint64_t seed, int threads,
                      uint32_t *out_obj, float *out_t);
/* Closest hit of n arbitrary rays (origin xyz, direction xyz; Ray::new normalises). */
int orc_trace_rays(orc_scene *s, const float *org_dir, uint64_t n, int threads,
                   uint32_t *out_obj, float *out_t);
/* Colour of trace_ray(ray, depth) for n rays (world.rs:115-131), rgb f32. */
int orc_shade_rays(orc_scene *s, const float *org_dir, uint64_t n, uint32_t depth, float *out_rgb);

/* BVH structure digest: leaf-order permutation of object indices (n_objects entries),
 * preorder node sizes (2*n entries max, returns count), depth, total surface area. */
uint32_t orc_bvh_digest(orc_scene *s, uint32_t *leaf_order, uint32_t *preorder_sizes,
                        uint32_t *depth, float *total_sa);

/* Primary ray of (px,py,sample) exactly as camera.rs:32-47. */
void orc_generate_ray(orc_scene *s, uint32_t px, uint32_t py, uint32_t sample, int use_jitter,
                      uint64_t seed, float *out_org_dir);

/* ---- known-answer hooks for the reference's 13 unit tests ---- */
float orc_aabb_surface_area(const float *mn, const float *mx);                     /* bvh.rs:147-150 */
int orc_aabb_intersect(const float *mn, const float *mx, const float *org, const float *dir,
                       float *t);                                                  /* bvh.rs:60-131 */
float orc_ray_get_t(const float *org, const float *dir, const float *point);       /* ray.rs:31-33 */
int orc_object_intersect(orc_scene *s, uint32_t obj, const float *org, const float *dir,
                         float *t);                                                /* object.rs:145-220 */
void orc_object_bbox(orc_scene *s, uint32_t obj, float *mn, float *mx);            /* object.rs:290-313 */
/* split kinds: 0 naive, 1 by_x, 2 Basic, 3 SAH (bvh.rs:160-231) over all objects in input order */
void orc_bvh_split(orc_scene *s, int kind, uint32_t *n_left, uint32_t *n_right);
float orc_clamp(float x, float lo, float hi);                                      /* utils.rs:5-13 */
void orc_reflect(const float *v, const float *n, float *out);                      /* utils.rs:15-17 */
void orc_refract(const float *v, const float *n, float idx, float *out);           /* utils.rs:46-59 */
void orc_component_wise_range(const float *pts, uint32_t n, float *mn, float *mx); /* utils.rs:20-29 */
void orc_axis_scaling(const float *m16, float *out3);                              /* utils.rs:33-44 */
uint32_t orc_jitter_u32(uint64_t seed, uint32_t px, uint32_t py, uint32_t sample, uint32_t dim);

#ifdef __cplusplus
}
#endif
#endif
