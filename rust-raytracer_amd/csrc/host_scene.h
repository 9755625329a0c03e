// C++ mirror of the reference's host API around the trace path: same type and
// function names, argument meaning and error behaviour as the Rust crate, so that
// host code written against ellishg/rust-raytracer ports one to one.
//   Color      src/color.rs        Material / MaterialType / TextureType  src/material.rs
//   Object     src/object.rs       Light  src/light.rs      Camera  src/camera.rs
//   load_*     src/scene.rs        render src/world.rs:17-29
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtx.h"
#include "hostmath.h"

namespace rtx {

// color.rs:5-65 -- clamped on construction
struct Color {
    float r, g, b, a;
    static Color rgb(float r, float g, float b) { return rgba(r, g, b, 1.0f); }
    static Color rgba(float r, float g, float b, float a);
    static Color white() { return rgb(1, 1, 1); }
    static Color black() { return rgb(0, 0, 0); }
    static Color red() { return rgb(1, 0, 0); }
    static Color green() { return rgb(0, 1, 0); }
    static Color blue() { return rgb(0, 0, 1); }
    static Color yellow() { return rgb(1, 1, 0); }
    static Color grayscale(float v) { return rgb(v, v, v); }
};
Color operator*(Color a, Color b);  // color.rs:80-90

struct Image {
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> rgb;
};

// material.rs:14-18
struct TextureType {
    int kind = RTX_TEX_NONE;
    Color color = Color::black();
    std::shared_ptr<const Image> image;
    static TextureType new_texture(const std::string &path);  // throws std::runtime_error
    static TextureType new_flat(Color c);
    static TextureType None() { return TextureType(); }
};
// material.rs:21-31
struct MaterialType {
    int kind = RTX_MAT_NONE;
    float p0 = 0, p1 = 0, p2 = 0;
    std::vector<std::pair<MaterialType, float>> parts;
    static MaterialType new_phong(float diffuse, float specular, float shininess);
    static MaterialType Reflective();
    static MaterialType Refractive(float index);
    static MaterialType Composition(std::vector<std::pair<MaterialType, float>> parts);
    static MaterialType None() { return MaterialType(); }
};
// material.rs:33-37; shared so that a mesh's polygons intern to one material
struct MaterialData {
    MaterialType material_type;
    TextureType texture_type;
};
struct Material {
    std::shared_ptr<const MaterialData> d;
    static Material make(MaterialType m, TextureType t);  // Material::new
};

// object.rs:14-26
struct Object {
    int type = RTX_OBJ_SPHERE;
    V3 v[4] = {};
    float radius = 0;
    M4 object_to_world = m4_identity(), world_to_object = m4_identity();
    Material material;
    static std::vector<Object> new_mesh(const std::string &path, const M4 &object_to_world, Material m);  // 30-91
    static std::vector<Object> from_mesh(const struct ObjMesh &mesh, const M4 &object_to_world, Material m);  // the object.rs:62-88 half of new_mesh
    static Object new_sphere(V3 center, float radius, Material m);                                      // 93-100
    static Object new_quad(V3 a, V3 b, V3 c, V3 d, Material m);                                          // 102-115
    static Object new_triangle(V3 a, V3 b, V3 c, Material m);                                            // 117-129
    Object transform(const M4 &object_to_world) const;                                                   // 131-139
};

// light.rs:12-56
struct Light {
    int type = RTX_LIGHT_AMBIENT;
    V3 position = {0, 0, 0}, direction = {0, 0, 0};
    float angle_deg = 0;
    Color color = Color::black();
    static Light new_point(V3 position, Color c);
    static Light new_ambient(Color c);
    static Light new_directional(V3 direction, Color c);
    static Light new_cone(V3 position, V3 direction, float angle_deg, Color c);
};

// camera.rs:5-26
struct Camera {
    M4 camera_to_world;
    uint32_t width, height;
    static Camera make(uint32_t width, uint32_t height, V3 eye, V3 at, V3 up);  // Camera::new
};

// scene.rs
std::pair<std::vector<Object>, std::vector<Light>> load_basic(const std::string &media_dir);
std::pair<std::vector<Object>, std::vector<Light>> load_random_spheres(uint16_t num_spheres);
std::pair<std::vector<Object>, std::vector<Light>> load_suzanne(const std::string &media_dir);
Camera default_camera(uint32_t pixel_width);

// Flat scene arrays behind an rtx_scene_desc (what crosses the C ABI).
struct SceneArrays {
    std::vector<int32_t> obj_type, obj_xform, obj_material;
    std::vector<float> obj_verts, xforms;
    std::vector<int32_t> mat_root, mat_tex_kind, mat_tex_id;
    std::vector<float> mat_tex_color;
    std::vector<int32_t> node_type, node_child_first, node_child_count, child_node;
    std::vector<float> node_params, child_coef;
    std::vector<int32_t> tex_w, tex_h;
    std::vector<uint64_t> tex_offset;
    std::vector<uint8_t> tex_data;
    std::vector<int32_t> light_type;
    std::vector<float> light_pos, light_dir, light_angle, light_color;
    float background[3] = {0, 0, 0};
    M4 camera_to_world = m4_identity();
    uint32_t width = 0, height = 0;
    // interning by identity: the caches OWN a reference, so an address cannot be reused by
    // a later allocation while its entry is alive
    std::vector<std::pair<std::shared_ptr<const MaterialData>, int>> mat_cache;
    std::vector<std::pair<std::shared_ptr<const Image>, int>> tex_cache;

    int add_xform(const M4 &o2w, const M4 &w2o);   // interned
    int add_material(const Material &m);           // interned by identity
    void add_object(const Object &o);
    void add_triangle(const float a[3], const float b[3], const float c[3], int xform, int material);
    void add_mesh(const struct ObjMesh &mesh, const M4 &o2w, const Material &m);  // one transform + material entry for all polygons
    void add_light(const Light &l);
    void set_camera(const Camera &c);
    void set_background(Color c);
    rtx_scene_desc desc() const;  // borrowed view; valid while *this is unchanged
};

// world::render (world.rs:17-29).  Returns RTX_OK or a negative rtx_status; `error`
// receives the message.  The trace runs on HIP device `device`.
int render(const Camera &camera, const std::vector<Object> &objects, const std::vector<Light> &lights,
           Color background_color, uint16_t samples_per_pixel, uint16_t max_ray_bounces, const std::string &path,
           size_t num_threads, int device, std::string *error);

// OBJ subset the reference consumes through obj 0.9.1 (object.rs:39-75): positions and
// polygon faces of 3 or 4 vertices; vt/vn indices are parsed and ignored.
struct ObjMesh {
    std::vector<float> positions;                // xyz
    std::vector<std::vector<uint32_t>> polys;    // 0-based position indices
};
bool load_obj(const std::string &path, ObjMesh &out, std::string &err);
// Stanford PLY (ascii / binary, either endianness): vertex x y z + face index lists.
bool load_ply(const std::string &path, ObjMesh &out, std::string &err);
bool load_mesh(const std::string &path, ObjMesh &out, std::string &err);  // by extension: .ply, else OBJ
// Scene description file (scene_file.cpp documents the format): fills camera, background,
// objects and lights; `width` is the -w default when the file has no `size` line.
bool load_scene_file(SceneArrays &s, const std::string &path, uint32_t width, const std::string &media_dir, std::string &err);
bool png_read_rgb8(const std::string &path, Image &out, std::string &err);
bool png_write_rgb8(const std::string &path, uint32_t w, uint32_t h, const uint8_t *rgb, std::string &err);

}  // namespace rtx
