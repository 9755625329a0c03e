/*
 * rtx_host.h -- host-side callers of the trace path, exported as C so that tests, the
 * bench and the CLI share one implementation.  These stand in for the parts of the
 * reference that sit either side of the hot path and stay on the host:
 *   scene construction   src/scene.rs:11-226  (load_basic, load_random_spheres, load_suzanne,
 *                                              default_camera) + the synthetic bench scenes
 *   mesh / texture input src/object.rs:30-91 (OBJ), src/material.rs:40-47 (PNG decode)
 *   image output         src/world.rs:76-81 (PNG encode)
 * (The C++ mirror of the reference's own API -- Object::new_sphere, Light::new_cone,
 * render(...) -- is rust-raytracer_amd/csrc/host_scene.h.)
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H
#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtx_host_scene rtx_host_scene;

/* Built-in scenes by name.  `width` = -w/--pixel_width (square viewport, scene.rs:218-226).
 * media_dir holds Suzanne.obj and texture.png (the reference's media/).  Names:
 *   "basic"            load_basic()                                    scene.rs:11-108
 *   "random_spheres"   load_random_spheres(param or 1000)              scene.rs:111-186
 *   "suzanne"          load_suzanne()                                  scene.rs:188-214
 *   "basic_suzanne"    BASELINE config 2: load_basic + load_suzanne + directional light
 *   "scan"             BASELINE config 3: load_basic + 4 synthetic scan-like meshes,
 *                      param = total triangles (0 = 2,374,525).  With param = 0, each of
 *                      dragon_vrip.ply / bun_zipper.ply / Armadillo.ply / happy_vrip.ply found in
 *                      media_dir replaces its stand-in (same place, size and material)
 *   "scan_glass"       BASELINE config 4: same geometry, mirror+glass+Phong compositions
 *   "soup"             BASELINE config 5: param random triangles (0 = 20,000,000), 4 point lights
 *   "file:<path>"      a scene description file (camera, materials, objects, meshes, lights; the format
 *                      is documented at the top of csrc/scene_file.cpp, example: assets/scenes/basic.scene)
 *   "mesh:<path>"      load_basic + the OBJ or PLY (ascii / binary) file at <path>, scaled to a largest
 *                      extent of 1.6 and set on the floor (Object::new_mesh, object.rs:30-91, reads OBJ
 *                      only; the headline render's Stanford scans are PLY); param = 1: gold half-mirror
 * Returns NULL on error (rtx_host_last_error()). */
rtx_host_scene *rtx_scene_builtin(const char *name, uint32_t width, const char *media_dir, uint64_t param);
const rtx_scene_desc *rtx_scene_get_desc(const rtx_host_scene *s);
void rtx_scene_free(rtx_host_scene *s);
const char *rtx_host_last_error(void);

/* PNG (8-bit, non-interlaced) over zlib.  read: *out_rgb is malloc'ed RGB8 (alpha dropped
 * as image::DynamicImage::to_rgb does, material.rs:45); free with rtx_host_free. */
int rtx_png_read_rgb8(const char *path, uint32_t *w, uint32_t *h, uint8_t **out_rgb);
int rtx_png_write_rgb8(const char *path, uint32_t w, uint32_t h, const uint8_t *rgb);
void rtx_host_free(void *p);

/* world::render (world.rs:17-29) on a built-in scene: upload, trace on `device`, save PNG.
 * num_threads drives the host BVH build (rtx_set_build_threads; 0 = default), the trace runs on the GPU.
 * The PNG has the camera's size: width x width for the built-in scenes, the `size W H` of a scene file. */
int rtx_render_builtin_to_png(const char *scene, uint32_t width, const char *media_dir, uint64_t param,
                              uint32_t samples_per_pixel, uint32_t max_ray_bounces, const char *path,
                              uint32_t num_threads, int device);

/* Host half of rtx_upload_scene without a GPU: builds the reference-topology BVH
 * (bvh.rs:334-357) and flattens it; reports the leaf-order permutation (n_objects entries),
 * the preorder node sizes (up to 2*n_objects entries), depth and total surface area
 * (bvh.rs:404-409,419-426).  Returns the node count or a negative rtx_status. */
int rtx_host_bvh_digest(const rtx_scene_desc *desc, int threads, uint32_t *leaf_order, uint32_t *preorder_sizes,
                        uint32_t *depth, float *total_sa, uint32_t *n_internal);

/* How far a hit can lie outside the box the reference bounds its object with, in world units
 * (largest value over the scene; no counterpart in the reference, which never asks): 0 for
 * triangles, planar quads and spheres under rotation + uniform scale, positive for non-planar
 * quads (object.rs:167-189 vs 284-299) and sheared spheres (utils.rs:33-44), infinity for a
 * transform that is not affine, negative rtx_status (as a float) for a bad descriptor.  Scenes
 * with a non-zero value are traced by the `_ovh` kernels (exactness). */
float rtx_scene_hit_overhang(const rtx_scene_desc *desc);

/* The host's cgmath-order f32 matrix arithmetic (csrc/hostmath.h) behind one door, so that tests can pin it against
 * the reference's own vectors (utils.rs:91-108 test_get_scaling, object.rs:379-407) and against float64: it builds
 * every transform, camera and box that BOTH sides of a GPU-vs-oracle parity test are fed with.  Matrices are 16
 * column-major f32.  `in` / `out` per op:
 *   FROM_ANGLE_X/Y/Z        in: degrees (1)            out: Matrix4::from_angle_*(Deg(a))                 (16)
 *   FROM_TRANSLATION        in: x y z                  out: Matrix4::from_translation                     (16)
 *   FROM_NONUNIFORM_SCALE   in: x y z                  out: Matrix4::from_nonuniform_scale                (16)
 *   MUL                     in: lhs (16), rhs (16)     out: lhs * rhs                                     (16)
 *   INVERT                  in: m (16)                 out: m.invert() / inverse_transform()              (16); RTX_ERR_DEGENERATE if singular
 *   LOOK_AT                 in: eye, center, up (9)    out: Matrix4::look_at (right-handed, camera.rs:19) (16)
 *   AXIS_SCALING            in: m (16)                 out: utils::get_axis_scaling (utils.rs:33-44)      (3)
 *   TRANSFORM_POINT         in: m (16), p (3)          out: Transform::transform_point                    (3)
 *   TRANSFORM_VECTOR        in: m (16), v (3)          out: Transform::transform_vector                   (3)
 *   OBJECT_BOX              in: type (as float: RTX_OBJ_*), 12 vertex floats, object_to_world (16)
 *                                                      out: Object::get_bounding_box (object.rs:290-313): min (3), max (3) */
enum { RTX_M4_FROM_ANGLE_X = 0, RTX_M4_FROM_ANGLE_Y, RTX_M4_FROM_ANGLE_Z, RTX_M4_FROM_TRANSLATION, RTX_M4_FROM_NONUNIFORM_SCALE,
       RTX_M4_MUL, RTX_M4_INVERT, RTX_M4_LOOK_AT, RTX_M4_AXIS_SCALING, RTX_M4_TRANSFORM_POINT, RTX_M4_TRANSFORM_VECTOR,
       RTX_M4_OBJECT_BOX };
int rtx_host_m4(int op, const float *in, float *out);

#ifdef __cplusplus
}
#endif
#endif
