// Host-side mirror of the reference's scene API (see host_scene.h) plus the synthetic
// scenes of the benchmark configurations, and the C exports of include/rtx_host.h.
#include "host_scene.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include "../../include/rtx_host.h"

namespace rtx {

// ------------------------------------------------------------------ color.rs
static float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
Color Color::rgba(float r, float g, float b, float a) { return Color{clamp01(r), clamp01(g), clamp01(b), clamp01(a)}; }
Color operator*(Color a, Color b) { return Color::rgba(a.r * b.r, a.g * b.g, a.b * b.b, a.a * b.a); }

// ------------------------------------------------------------------ material.rs
TextureType TextureType::new_texture(const std::string &path) {
    auto img = std::make_shared<Image>();
    std::string err;
    if (!png_read_rgb8(path, *img, err)) throw std::runtime_error(err);
    TextureType t;
    t.kind = RTX_TEX_IMAGE;
    t.image = img;
    return t;
}
TextureType TextureType::new_flat(Color c) {
    TextureType t;
    t.kind = RTX_TEX_FLAT;
    t.color = c;
    return t;
}
MaterialType MaterialType::new_phong(float diffuse, float specular, float shininess) {
    MaterialType m;
    m.kind = RTX_MAT_PHONG;
    m.p0 = diffuse;
    m.p1 = specular;
    m.p2 = shininess;
    return m;
}
MaterialType MaterialType::Reflective() {
    MaterialType m;
    m.kind = RTX_MAT_REFLECTIVE;
    return m;
}
MaterialType MaterialType::Refractive(float index) {
    MaterialType m;
    m.kind = RTX_MAT_REFRACTIVE;
    m.p0 = index;
    return m;
}
MaterialType MaterialType::Composition(std::vector<std::pair<MaterialType, float>> parts) {
    MaterialType m;
    m.kind = RTX_MAT_COMPOSITION;
    m.parts = std::move(parts);
    return m;
}
Material Material::make(MaterialType m, TextureType t) {
    auto d = std::make_shared<MaterialData>();
    d->material_type = std::move(m);
    d->texture_type = std::move(t);
    return Material{d};
}

// ------------------------------------------------------------------ object.rs
Object Object::new_sphere(V3 center, float radius, Material m) {
    Object o;
    o.type = RTX_OBJ_SPHERE;
    o.v[0] = center;
    o.radius = radius;
    o.material = std::move(m);
    return o;
}
Object Object::new_quad(V3 a, V3 b, V3 c, V3 d, Material m) {
    Object o;
    o.type = RTX_OBJ_QUAD;
    o.v[0] = a;
    o.v[1] = b;
    o.v[2] = c;
    o.v[3] = d;
    o.material = std::move(m);
    return o;
}
Object Object::new_triangle(V3 a, V3 b, V3 c, Material m) {
    Object o;
    o.type = RTX_OBJ_TRIANGLE;
    o.v[0] = a;
    o.v[1] = b;
    o.v[2] = c;
    o.material = std::move(m);
    return o;
}
Object Object::transform(const M4 &o2w) const {
    Object o = *this;
    o.object_to_world = o2w * object_to_world;
    if (!m4_invert(o.object_to_world, o.world_to_object)) throw std::runtime_error("transform is not invertible");
    return o;
}
std::vector<Object> Object::new_mesh(const std::string &path, const M4 &object_to_world, Material m) {
    ObjMesh mesh;
    std::string err;
    if (!load_mesh(path, mesh, err)) throw std::runtime_error(err);  // OBJ as in the reference, or PLY
    return from_mesh(mesh, object_to_world, m);
}
std::vector<Object> Object::from_mesh(const ObjMesh &mesh, const M4 &object_to_world, Material m) {
    M4 inv;
    if (!m4_invert(object_to_world, inv)) throw std::runtime_error("mesh transform is not invertible");
    std::vector<Object> out;
    out.reserve(mesh.polys.size());
    for (const auto &poly : mesh.polys) {
        if (poly.size() != 3 && poly.size() != 4)  // object.rs:73-75 panics
            throw std::runtime_error("Mesh has a polygon with " + std::to_string(poly.size()) + " vertices which is not supported.");
        Object o;
        o.type = poly.size() == 3 ? RTX_OBJ_TRIANGLE : RTX_OBJ_QUAD;
        for (size_t k = 0; k < poly.size(); k++) {
            const float *p = &mesh.positions[(size_t)poly[k] * 3];
            o.v[k] = v3(p[0], p[1], p[2]);
        }
        o.object_to_world = object_to_world;
        o.world_to_object = inv;
        o.material = m;
        out.push_back(std::move(o));
    }
    return out;
}

// ------------------------------------------------------------------ light.rs
Light Light::new_point(V3 position, Color c) {
    Light l;
    l.type = RTX_LIGHT_POINT;
    l.position = position;
    l.color = c;
    return l;
}
Light Light::new_ambient(Color c) {
    Light l;
    l.type = RTX_LIGHT_AMBIENT;
    l.color = c;
    return l;
}
Light Light::new_directional(V3 direction, Color c) {
    Light l;
    l.type = RTX_LIGHT_DIRECTIONAL;
    l.direction = normalize(direction);
    l.color = c;
    return l;
}
Light Light::new_cone(V3 position, V3 direction, float angle_deg, Color c) {
    Light l;
    l.type = RTX_LIGHT_CONE;
    l.position = position;
    l.direction = normalize(direction);
    l.angle_deg = angle_deg;
    l.color = c;
    return l;
}

// ------------------------------------------------------------------ camera.rs
Camera Camera::make(uint32_t width, uint32_t height, V3 eye, V3 at, V3 up) {
    Camera c;
    c.width = width;
    c.height = height;
    M4 w2c = m4_look_at(eye, at, up);
    if (!m4_invert(w2c, c.camera_to_world)) throw std::runtime_error("camera matrix is not invertible");
    return c;
}

// ------------------------------------------------------------------ scene.rs
Camera default_camera(uint32_t w) { return Camera::make(w, w, v3(0.0f, 1.5f, 5.0f), v3(0, 0, 0), v3(0, 1, 0)); }

std::pair<std::vector<Object>, std::vector<Light>> load_basic(const std::string &media_dir) {
    std::vector<Object> objects;
    std::vector<Light> lights;
    TextureType texture = TextureType::new_texture(media_dir + "/texture.png");
    // textured plane (scene.rs:17-26)
    objects.push_back(Object::new_quad(v3(-5, -1, 5), v3(5, -1, 5), v3(5, -1, -5), v3(-5, -1, -5),
                                       Material::make(MaterialType::new_phong(1.0f, 0.0f, 1.0f), texture)));
    // textured sphere (scene.rs:28-39)
    {
        Object o = Object::new_sphere(v3(1.0f, 0.5f, -2.0f), 1.5f, Material::make(MaterialType::new_phong(0.4f, 0.6f, 1.8f), texture));
        M4 rotate = m4_from_angle_x(90.0f);
        M4 translate = m4_from_translation(v3(1.0f, 0.5f, -2.0f));
        M4 tinv;
        m4_invert(translate, tinv);
        objects.push_back(o.transform(translate * rotate * tinv));
    }
    // mirror sphere (scene.rs:41-51)
    objects.push_back(Object::new_sphere(
        v3(-2, 0, -2), 1.0f,
        Material::make(MaterialType::Composition({{MaterialType::Reflective(), 0.4f}, {MaterialType::new_phong(0.4f, 0.6f, 1.8f), 0.6f}}),
                       TextureType::new_flat(Color::blue()))));
    // transparent sphere (scene.rs:53-64)
    objects.push_back(Object::new_sphere(
        v3(1.0f, -0.25f, 1.0f), 0.75f,
        Material::make(MaterialType::Composition({{MaterialType::Refractive(1.3f), 0.8f}, {MaterialType::new_phong(0.4f, 0.6f, 1.8f), 0.2f}}),
                       TextureType::new_flat(Color::green()))));
    // yellow triangle (scene.rs:66-75)
    objects.push_back(Object::new_triangle(v3(-2, 0, 1), v3(-1, 0, 1), v3(-2, 2, 1),
                                           Material::make(MaterialType::new_phong(1.0f, 0.0f, 1.0f), TextureType::new_flat(Color::yellow()))));
    lights.push_back(Light::new_point(v3(2.0f, 3.0f, 0.5f), Color::rgb(1, 1, 1)));
    lights.push_back(Light::new_cone(v3(-1.5f, 2.0f, 0.0f), v3(1, -1, 0), 50.0f, Color::red()));
    lights.push_back(Light::new_cone(v3(1.5f, 2.0f, 0.0f), v3(-1, -1, 0), 50.0f, Color::blue()));
    lights.push_back(Light::new_cone(v3(0.0f, 2.0f, 1.5f), v3(0, -1, -1), 50.0f, Color::green()));
    lights.push_back(Light::new_ambient(Color::grayscale(0.2f)));
    return {std::move(objects), std::move(lights)};
}

// Small deterministic generator for the synthetic scenes (PCG32).  The reference seeds
// rand 0.7.3's StdRng (ChaCha20) with 248 (scene.rs:141); that stream is not reproduced
// here, so sphere positions differ from the Rust binary's ("parity unpinned").
struct Pcg32 {
    uint64_t state, inc;
    explicit Pcg32(uint64_t seed) : state(0), inc((seed << 1) | 1) {
        next();
        state += 0x853c49e6748fea9bULL + seed;
        next();
    }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((-rot) & 31));
    }
    float unit() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
    float range(float lo, float hi) { return lo + (hi - lo) * unit(); }
};

std::pair<std::vector<Object>, std::vector<Light>> load_random_spheres(uint16_t num_spheres) {
    std::vector<Object> objects;
    std::vector<Light> lights;
    // ground plane (scene.rs:115-125)
    objects.push_back(Object::new_quad(v3(-100, 0, 10), v3(10, 0, 10), v3(10, 0, -10), v3(-100, 0, -10),
                                       Material::make(MaterialType::new_phong(1.0f, 0.0f, 0.2f), TextureType::new_flat(Color::grayscale(0.3f)))));
    // back plane (scene.rs:127-139)
    float back_pos = -8.0f, height = 4.0f;
    objects.push_back(Object::new_quad(v3(10, -1, back_pos), v3(10, height, back_pos), v3(-10, height, back_pos), v3(-10, -1, back_pos),
                                       Material::make(MaterialType::new_phong(1.0f, 0.0f, 1.0f),
                                                      TextureType::new_flat(Color::blue() * Color::grayscale(0.5f)))));
    Pcg32 rng(248);
    for (uint16_t i = 0; i < num_spheres; i++) {  // scene.rs:142-156
        float r = rng.range(0.5f, 1.0f), g = rng.range(0.5f, 1.0f), b = rng.range(0.5f, 1.0f);
        float x = rng.range(-4.0f, 4.0f), y = rng.range(0.1f, 2.5f), z = rng.range(-6.0f, 3.5f);
        float rad = rng.range(0.05f, 0.15f);
        objects.push_back(Object::new_sphere(v3(x, y, z), rad,
                                             Material::make(MaterialType::new_phong(1.0f, 0.0f, 1.0f), TextureType::new_flat(Color::rgb(r, g, b)))));
    }
    auto add_sphere = [&](V3 p, Color c) {  // scene.rs:159-174
        objects.push_back(Object::new_sphere(
            p, 0.5f,
            Material::make(MaterialType::Composition({{MaterialType::Reflective(), 0.6f}, {MaterialType::new_phong(1.0f, 0.6f, 1.0f), 0.4f}}),
                           TextureType::new_flat(c))));
    };
    add_sphere(v3(0.0f, 0.5f, 0.0f), Color::red());
    add_sphere(v3(1.5f, 0.5f, 0.5f), Color::blue());
    add_sphere(v3(-1.5f, 0.5f, -0.5f), Color::yellow());
    lights.push_back(Light::new_point(v3(1.0f, 2.0f, 2.5f), Color::white()));
    lights.push_back(Light::new_point(v3(-2.0f, 2.0f, 1.0f), Color::white()));
    lights.push_back(Light::new_ambient(Color::grayscale(0.2f)));
    lights.push_back(Light::new_directional(v3(-0.2f, -1.0f, -0.9f), Color::grayscale(0.3f)));
    return {std::move(objects), std::move(lights)};
}

std::pair<std::vector<Object>, std::vector<Light>> load_suzanne(const std::string &media_dir) {
    // scene.rs:188-214
    M4 translation = m4_from_translation(v3(0, 0, 0));
    M4 scale = m4_from_nonuniform_scale(1.0f, 1.0f, 1.0f);
    M4 rotation = m4_from_angle_y(20.0f) * m4_from_angle_z(35.0f) * m4_from_angle_x(-20.0f);
    M4 o2w = translation * scale * rotation;
    Material m = Material::make(MaterialType::Composition({{MaterialType::Reflective(), 0.6f}, {MaterialType::new_phong(0.2f, 0.8f, 1.0f), 0.4f}}),
                                TextureType::new_flat(Color::red()));
    return {Object::new_mesh(media_dir + "/Suzanne.obj", o2w, m), {}};
}

// ------------------------------------------------------------------ SceneArrays
int SceneArrays::add_xform(const M4 &o2w, const M4 &w2o) {
    size_t n = xforms.size() / 32;
    // look at the most recent entries first: meshes add thousands of identical matrices
    for (size_t k = 0; k < n && k < 64; k++) {
        size_t i = n - 1 - k;
        if (!std::memcmp(&xforms[i * 32], o2w.m, 64) && !std::memcmp(&xforms[i * 32 + 16], w2o.m, 64)) return (int)i;
    }
    xforms.insert(xforms.end(), o2w.m, o2w.m + 16);
    xforms.insert(xforms.end(), w2o.m, w2o.m + 16);
    return (int)n;
}

static int add_mat_node(SceneArrays &s, const MaterialType &m) {
    int idx = (int)s.node_type.size();
    s.node_type.push_back(m.kind);
    s.node_params.push_back(m.p0);
    s.node_params.push_back(m.p1);
    s.node_params.push_back(m.p2);
    s.node_child_first.push_back(0);
    s.node_child_count.push_back(0);
    if (m.kind == RTX_MAT_COMPOSITION) {
        std::vector<int> kids;
        for (const auto &p : m.parts) kids.push_back(add_mat_node(s, p.first));
        s.node_child_first[idx] = (int)s.child_node.size();
        s.node_child_count[idx] = (int)kids.size();
        for (size_t k = 0; k < kids.size(); k++) {
            s.child_node.push_back(kids[k]);
            s.child_coef.push_back(m.parts[k].second);
        }
    }
    return idx;
}

int SceneArrays::add_material(const Material &m) {
    static const Material none = Material::make(MaterialType::None(), TextureType::None());
    const std::shared_ptr<const MaterialData> &d = m.d ? m.d : none.d;
    for (size_t k = 0; k < mat_cache.size() && k < 64; k++) {
        const auto &e = mat_cache[mat_cache.size() - 1 - k];
        if (e.first == d) return e.second;
    }
    int idx = (int)mat_root.size();
    mat_root.push_back(add_mat_node(*this, d->material_type));
    const TextureType &t = d->texture_type;
    mat_tex_kind.push_back(t.kind);
    mat_tex_color.push_back(t.color.r);
    mat_tex_color.push_back(t.color.g);
    mat_tex_color.push_back(t.color.b);
    int tid = 0;
    if (t.kind == RTX_TEX_IMAGE) {
        tid = -1;
        for (auto &e : tex_cache)
            if (e.first == t.image) tid = e.second;
        if (tid < 0) {
            tid = (int)tex_w.size();
            tex_w.push_back((int)t.image->w);
            tex_h.push_back((int)t.image->h);
            tex_offset.push_back(tex_data.size());
            tex_data.insert(tex_data.end(), t.image->rgb.begin(), t.image->rgb.end());
            tex_cache.push_back({t.image, tid});
        }
    }
    mat_tex_id.push_back(tid);
    mat_cache.push_back({d, idx});
    return idx;
}

void SceneArrays::add_object(const Object &o) {
    obj_type.push_back(o.type);
    float v[12] = {0};
    if (o.type == RTX_OBJ_SPHERE) {
        v[0] = o.v[0].x;
        v[1] = o.v[0].y;
        v[2] = o.v[0].z;
        v[3] = o.radius;
    } else {
        int nv = o.type == RTX_OBJ_QUAD ? 4 : 3;
        for (int k = 0; k < nv; k++) {
            v[k * 3] = o.v[k].x;
            v[k * 3 + 1] = o.v[k].y;
            v[k * 3 + 2] = o.v[k].z;
        }
    }
    obj_verts.insert(obj_verts.end(), v, v + 12);
    obj_xform.push_back(add_xform(o.object_to_world, o.world_to_object));
    obj_material.push_back(add_material(o.material));
}

void SceneArrays::add_triangle(const float a[3], const float b[3], const float c[3], int xform, int material) {
    obj_type.push_back(RTX_OBJ_TRIANGLE);
    float v[12] = {a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2], 0, 0, 0};
    obj_verts.insert(obj_verts.end(), v, v + 12);
    obj_xform.push_back(xform);
    obj_material.push_back(material);
}

// A loaded mesh straight into the flat arrays: one transform and one material entry for the
// whole mesh, no per-polygon Object with two matrices and a material clone (object.rs:62-88
// makes those).
void SceneArrays::add_mesh(const ObjMesh &mesh, const M4 &o2w, const Material &m) {
    SceneArrays &s = *this;
    M4 w2o;
    if (!m4_invert(o2w, w2o)) throw std::runtime_error("mesh transform is not invertible");
    const int xf = s.add_xform(o2w, w2o), mat = s.add_material(m);
    const size_t n = mesh.polys.size();
    s.obj_type.reserve(s.obj_type.size() + n);
    s.obj_xform.reserve(s.obj_xform.size() + n);
    s.obj_material.reserve(s.obj_material.size() + n);
    s.obj_verts.reserve(s.obj_verts.size() + n * 12);
    for (const auto &poly : mesh.polys) {
        if (poly.size() != 3 && poly.size() != 4)  // object.rs:73-75 panics
            throw std::runtime_error("Mesh has a polygon with " + std::to_string(poly.size()) + " vertices which is not supported.");
        float v[12] = {0};
        for (size_t k = 0; k < poly.size(); k++) std::memcpy(v + 3 * k, &mesh.positions[(size_t)poly[k] * 3], 12);
        s.obj_type.push_back(poly.size() == 3 ? RTX_OBJ_TRIANGLE : RTX_OBJ_QUAD);
        s.obj_verts.insert(s.obj_verts.end(), v, v + 12);
        s.obj_xform.push_back(xf);
        s.obj_material.push_back(mat);
    }
}

void SceneArrays::add_light(const Light &l) {
    light_type.push_back(l.type);
    light_pos.insert(light_pos.end(), {l.position.x, l.position.y, l.position.z});
    light_dir.insert(light_dir.end(), {l.direction.x, l.direction.y, l.direction.z});
    light_angle.push_back(l.angle_deg);
    light_color.insert(light_color.end(), {l.color.r, l.color.g, l.color.b});
}
void SceneArrays::set_camera(const Camera &c) {
    camera_to_world = c.camera_to_world;
    width = c.width;
    height = c.height;
}
void SceneArrays::set_background(Color c) {
    background[0] = c.r;
    background[1] = c.g;
    background[2] = c.b;
}

rtx_scene_desc SceneArrays::desc() const {
    rtx_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_objects = (uint32_t)obj_type.size();
    d.obj_type = obj_type.data();
    d.obj_verts = obj_verts.data();
    d.obj_xform = obj_xform.data();
    d.obj_material = obj_material.data();
    d.n_xforms = (uint32_t)(xforms.size() / 32);
    d.xforms = xforms.data();
    d.n_materials = (uint32_t)mat_root.size();
    d.mat_root = mat_root.data();
    d.mat_tex_kind = mat_tex_kind.data();
    d.mat_tex_color = mat_tex_color.data();
    d.mat_tex_id = mat_tex_id.data();
    d.n_mat_nodes = (uint32_t)node_type.size();
    d.node_type = node_type.data();
    d.node_params = node_params.data();
    d.node_child_first = node_child_first.data();
    d.node_child_count = node_child_count.data();
    d.n_mat_children = (uint32_t)child_node.size();
    d.child_node = child_node.data();
    d.child_coef = child_coef.data();
    d.n_textures = (uint32_t)tex_w.size();
    d.tex_w = tex_w.data();
    d.tex_h = tex_h.data();
    d.tex_offset = tex_offset.data();
    d.tex_data = tex_data.data();
    d.n_lights = (uint32_t)light_type.size();
    d.light_type = light_type.data();
    d.light_pos = light_pos.data();
    d.light_dir = light_dir.data();
    d.light_angle_deg = light_angle.data();
    d.light_color = light_color.data();
    std::memcpy(d.background, background, sizeof(background));
    std::memcpy(d.camera_to_world, camera_to_world.m, sizeof(d.camera_to_world));
    d.width = width;
    d.height = height;
    return d;
}

// ------------------------------------------------------------------ synthetic bench scenes
namespace {

// A closed, scan-like mesh: a cube-sphere of 12*n^2 near-uniform triangles displaced by a
// few low-frequency lobes and a fine ripple, truncated to exactly `ntris` triangles (the
// missing patch sits on the -y face, as scans have holes underneath).  Stands in for the
// Stanford models of the README headline scene, which are not redistributable here.
void add_scan_mesh(SceneArrays &s, uint64_t ntris, uint32_t seed, V3 center, float radius, int material, int xform) {
    if (ntris == 0) return;
    uint32_t n = (uint32_t)std::ceil(std::sqrt((double)ntris / 12.0));
    if (n < 1) n = 1;
    Pcg32 rng(seed);
    float ph[6], fr[6][3], am[6];
    for (int k = 0; k < 6; k++) {
        ph[k] = rng.range(0.0f, 6.2831853f);
        for (int j = 0; j < 3; j++) fr[k][j] = rng.range(-1.0f, 1.0f) * (k < 3 ? 3.0f : 9.0f);
        am[k] = k < 3 ? 0.16f : 0.04f;
    }
    float fine = 40.0f + 10.0f * rng.unit();
    auto vertex = [&](int face, uint32_t i, uint32_t j, float out[3]) {
        float u = -1.0f + 2.0f * (float)i / (float)n, v = -1.0f + 2.0f * (float)j / (float)n;
        float p[3];
        switch (face) {  // outward-facing parametrisations
            case 0: p[0] = 1; p[1] = v; p[2] = -u; break;   // +x
            case 1: p[0] = -1; p[1] = v; p[2] = u; break;   // -x
            case 2: p[0] = u; p[1] = 1; p[2] = -v; break;   // +y
            case 3: p[0] = u; p[1] = v; p[2] = 1; break;    // +z
            case 4: p[0] = -u; p[1] = v; p[2] = -1; break;  // -z
            default: p[0] = u; p[1] = -1; p[2] = v; break;  // -y
        }
        float len = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        float d[3] = {p[0] / len, p[1] / len, p[2] / len};
        float r = 1.0f;
        for (int k = 0; k < 6; k++) r += am[k] * std::sin(fr[k][0] * d[0] + fr[k][1] * d[1] + fr[k][2] * d[2] + ph[k]);
        r += 0.004f * std::sin(fine * d[0]) * std::sin(fine * d[1] + 1.0f) * std::sin(fine * d[2] + 2.0f);
        out[0] = center.x + radius * r * d[0];
        out[1] = center.y + radius * r * d[1];
        out[2] = center.z + radius * r * d[2];
    };
    uint64_t emitted = 0;
    std::vector<float> row0((size_t)(n + 1) * 3), row1((size_t)(n + 1) * 3);
    for (int face = 0; face < 6 && emitted < ntris; face++) {
        for (uint32_t i = 0; i <= n; i++) vertex(face, i, 0, &row0[(size_t)i * 3]);
        for (uint32_t j = 0; j < n && emitted < ntris; j++) {
            for (uint32_t i = 0; i <= n; i++) vertex(face, i, j + 1, &row1[(size_t)i * 3]);
            for (uint32_t i = 0; i < n && emitted < ntris; i++) {
                const float *p00 = &row0[(size_t)i * 3], *p10 = &row0[(size_t)(i + 1) * 3];
                const float *p01 = &row1[(size_t)i * 3], *p11 = &row1[(size_t)(i + 1) * 3];
                // counter-clockwise seen from outside: normal (b-a)x(c-a) points outward
                s.add_triangle(p00, p10, p11, xform, material);
                emitted++;
                if (emitted < ntris) {
                    s.add_triangle(p00, p11, p01, xform, material);
                    emitted++;
                }
            }
            row0.swap(row1);
        }
    }
}

void reserve_tris(SceneArrays &s, uint64_t n) {
    s.obj_type.reserve(s.obj_type.size() + n);
    s.obj_xform.reserve(s.obj_xform.size() + n);
    s.obj_material.reserve(s.obj_material.size() + n);
    s.obj_verts.reserve(s.obj_verts.size() + n * 12);
}

}  // namespace

}  // namespace rtx

// ------------------------------------------------------------------ C exports
using namespace rtx;

struct rtx_host_scene {
    SceneArrays arrays;
    rtx_scene_desc desc;
};

namespace {
thread_local std::string g_host_err;

void add_all(SceneArrays &s, const std::pair<std::vector<Object>, std::vector<Light>> &ol) {
    for (const auto &o : ol.first) s.add_object(o);
    for (const auto &l : ol.second) s.add_light(l);
}

// Uniform scale + translation that gives the mesh a largest extent of `extent`, centres it
// over (cx, cz) and sets its lowest point on y = floor_y.
M4 fit_on_floor(const ObjMesh &mesh, float cx, float floor_y, float cz, float extent, const std::string &what) {
    if (mesh.positions.empty() || mesh.polys.empty()) throw std::runtime_error(what + ": no faces");
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < mesh.positions.size(); i++) {
        lo[i % 3] = std::min(lo[i % 3], mesh.positions[i]);
        hi[i % 3] = std::max(hi[i % 3], mesh.positions[i]);
    }
    float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(ext > 0.0f) || !std::isfinite(ext)) throw std::runtime_error(what + ": degenerate bounding box");
    float k = extent / ext;
    return m4_from_translation(v3(cx - 0.5f * (lo[0] + hi[0]) * k, floor_y - lo[1] * k, cz - 0.5f * (lo[2] + hi[2]) * k)) *
           m4_from_nonuniform_scale(k, k, k);
}

// Materials of the four stand-in meshes as they appear in media/scene.jpg: green, red and
// green Phong, and a gold half-mirror.
const char *const SCAN_FILES[4] = {"dragon_vrip.ply", "bun_zipper.ply", "Armadillo.ply", "happy_vrip.ply"};
void add_scan_scene(SceneArrays &s, const std::string &media, uint64_t total, bool glass) {
    add_all(s, load_basic(media));
    if (total == 0) total = 2374525ull;
    const uint64_t ref_counts[4] = {871414ull, 69451ull, 345944ull, 1087716ull};  // dragon, bunny, armadillo, Buddha
    uint64_t counts[4], acc = 0;
    for (int k = 0; k < 4; k++) {
        counts[k] = k == 3 ? total - acc : total * ref_counts[k] / 2374525ull;
        acc += counts[k];
    }
    Material mats[4];
    if (glass) {
        for (int k = 0; k < 4; k++)
            mats[k] = Material::make(
                MaterialType::Composition({{MaterialType::Reflective(), 0.4f}, {MaterialType::Refractive(1.3f), 0.4f}, {MaterialType::new_phong(0.6f, 0.4f, 4.0f), 0.2f}}),
                TextureType::new_flat(k == 1 ? Color::rgb(0.9f, 0.2f, 0.2f) : (k == 3 ? Color::rgb(0.9f, 0.75f, 0.3f) : Color::rgb(0.2f, 0.8f, 0.3f))));
    } else {
        mats[0] = Material::make(MaterialType::new_phong(0.7f, 0.3f, 8.0f), TextureType::new_flat(Color::rgb(0.2f, 0.8f, 0.3f)));
        mats[1] = Material::make(MaterialType::new_phong(0.8f, 0.2f, 4.0f), TextureType::new_flat(Color::rgb(0.9f, 0.2f, 0.2f)));
        mats[2] = Material::make(MaterialType::new_phong(0.7f, 0.3f, 8.0f), TextureType::new_flat(Color::rgb(0.3f, 0.7f, 0.2f)));
        mats[3] = Material::make(MaterialType::Composition({{MaterialType::Reflective(), 0.5f}, {MaterialType::new_phong(0.5f, 0.5f, 6.0f), 0.5f}}),
                                 TextureType::new_flat(Color::rgb(0.9f, 0.75f, 0.3f)));
    }
    const V3 centers[4] = {v3(-0.9f, -0.45f, 0.3f), v3(0.0f, -0.68f, 1.6f), v3(2.5f, -0.5f, 0.2f), v3(-0.5f, -0.3f, -1.3f)};
    const float radii[4] = {0.55f, 0.3f, 0.45f, 0.65f};
    M4 id = m4_identity();
    int xf = s.add_xform(id, id);
    // the Stanford scans themselves when they lie in the media directory
    // (they ship with neither the reference nor this repository); each one found replaces its
    // stand-in, same place, size and material.  Only at the default triangle count.
    ObjMesh real[4];
    bool have[4] = {false, false, false, false};
    uint64_t synthetic = 0;
    for (int k = 0; k < 4; k++) {
        const std::string path = media + "/" + SCAN_FILES[k];
        std::string err;
        if (total == 2374525ull && std::ifstream(path).good()) {
            if (!load_mesh(path, real[k], err)) throw std::runtime_error(err);
            have[k] = true;
        } else {
            synthetic += counts[k];
        }
    }
    reserve_tris(s, synthetic);
    for (int k = 0; k < 4; k++) {
        if (have[k])
            s.add_mesh(real[k], fit_on_floor(real[k], centers[k].x, -1.0f, centers[k].z, 2.0f * radii[k], SCAN_FILES[k]), mats[k]);
        else
            add_scan_mesh(s, counts[k], (uint32_t)(k + 1), centers[k], radii[k], s.add_material(mats[k]), xf);
    }
}

// A mesh file under the lights and floor of load_basic: scaled so that its largest extent is
// 1.6, centred over (0, 0.3) in x/z and set on the floor plane y = -1.
void add_mesh_scene(SceneArrays &s, const std::string &media, const std::string &path, uint64_t material) {
    ObjMesh mesh;
    std::string err;
    if (!load_mesh(path, mesh, err)) throw std::runtime_error(err);
    M4 o2w = fit_on_floor(mesh, 0.0f, -1.0f, 0.3f, 1.6f, path);
    Material m = material == 1
                     ? Material::make(MaterialType::Composition({{MaterialType::Reflective(), 0.5f}, {MaterialType::new_phong(0.5f, 0.5f, 6.0f), 0.5f}}),
                                      TextureType::new_flat(Color::rgb(0.9f, 0.75f, 0.3f)))
                     : Material::make(MaterialType::new_phong(0.7f, 0.3f, 8.0f), TextureType::new_flat(Color::grayscale(0.8f)));
    add_all(s, load_basic(media));
    s.add_mesh(mesh, o2w, m);
}

void add_soup_scene(SceneArrays &s, uint64_t n) {
    if (n == 0) n = 20000000ull;
    M4 id = m4_identity();
    int xf = s.add_xform(id, id);
    int mat = s.add_material(Material::make(MaterialType::new_phong(1.0f, 0.0f, 1.0f), TextureType::new_flat(Color::grayscale(0.8f))));
    reserve_tris(s, n);
    Pcg32 rng(5);
    for (uint64_t i = 0; i < n; i++) {
        float c[3], e1[3], e2[3];
        for (int k = 0; k < 3; k++) c[k] = rng.range(-10.0f, 10.0f);
        for (int k = 0; k < 3; k++) e1[k] = rng.range(-0.05f, 0.05f);
        for (int k = 0; k < 3; k++) e2[k] = rng.range(-0.05f, 0.05f);
        float a[3] = {c[0], c[1], c[2]}, b[3] = {c[0] + e1[0], c[1] + e1[1], c[2] + e1[2]}, d[3] = {c[0] + e2[0], c[1] + e2[1], c[2] + e2[2]};
        s.add_triangle(a, b, d, xf, mat);
    }
    const float L = 8.0f;
    s.add_light(Light::new_point(v3(L, 12.0f, L), Color::white()));
    s.add_light(Light::new_point(v3(-L, 12.0f, L), Color::white()));
    s.add_light(Light::new_point(v3(L, 12.0f, -L), Color::white()));
    s.add_light(Light::new_point(v3(-L, 12.0f, -L), Color::white()));
    s.add_light(Light::new_ambient(Color::grayscale(0.2f)));
}

bool build_builtin(SceneArrays &s, const std::string &name, uint32_t width, const std::string &media, uint64_t param) {
    s.set_background(Color::grayscale(0.2f));  // main.rs:111
    s.set_camera(default_camera(width));       // main.rs:108
    if (name == "basic") add_all(s, load_basic(media));
    else if (name == "random_spheres") add_all(s, load_random_spheres((uint16_t)(param ? param : 1000)));
    else if (name == "suzanne") add_all(s, load_suzanne(media));
    else if (name == "basic_suzanne") {
        add_all(s, load_basic(media));
        add_all(s, load_suzanne(media));
        s.add_light(Light::new_directional(v3(-0.2f, -1.0f, -0.9f), Color::grayscale(0.3f)));
    } else if (name == "scan") add_scan_scene(s, media, param, false);
    else if (name == "scan_glass") add_scan_scene(s, media, param, true);
    else if (name == "soup") {
        add_soup_scene(s, param);
        // look into the cloud from outside it
        s.set_camera(Camera::make(width, width, v3(0.0f, 6.0f, 24.0f), v3(0, 0, 0), v3(0, 1, 0)));
    } else if (name.rfind("file:", 0) == 0) {
        // a scene description file (scene_file.cpp): what the reference does by editing main.rs:82-111
        std::string err;
        if (!load_scene_file(s, name.substr(5), width, media, err)) throw std::runtime_error(err);
    } else if (name.rfind("mesh:", 0) == 0) {
        // any OBJ / PLY file under the lights and floor of load_basic: mesh:<path>, scaled to
        // fit a unit box and set on the floor (`param` = material: 0 grey Phong, 1 mirror mix)
        add_mesh_scene(s, media, name.substr(5), param);
    } else {
        g_host_err = "unknown scene '" + name + "'";
        return false;
    }
    return true;
}
}  // namespace

namespace rtx {
int render(const Camera &camera, const std::vector<Object> &objects, const std::vector<Light> &lights, Color background_color,
           uint16_t samples_per_pixel, uint16_t max_ray_bounces, const std::string &path, size_t num_threads, int device,
           std::string *error) {
    SceneArrays s;
    s.set_camera(camera);
    s.set_background(background_color);
    for (const auto &o : objects) s.add_object(o);
    for (const auto &l : lights) s.add_light(l);
    if (s.xforms.empty()) {
        M4 id = m4_identity();
        s.add_xform(id, id);
    }
    if (s.mat_root.empty()) s.add_material(Material());
    rtx_scene_desc d = s.desc();
    rtx_ctx *ctx = nullptr;
    int rc = rtx_init(device, &ctx);
    if (rc != RTX_OK) {
        if (error) *error = rtx_last_error(nullptr);
        return rc;
    }
    if (num_threads) (void)rtx_set_build_threads(ctx, (int)num_threads);
    rc = rtx_upload_scene(ctx, &d);
    std::vector<uint8_t> img((size_t)camera.width * camera.height * 3);
    if (rc == RTX_OK) rc = rtx_render(ctx, samples_per_pixel, max_ray_bounces, 0, nullptr, 0, img.data(), nullptr, nullptr);
    if (rc != RTX_OK && error) *error = rtx_last_error(ctx);
    rtx_destroy(ctx);
    if (rc != RTX_OK) return rc;
    std::string err;
    if (!png_write_rgb8(path, camera.width, camera.height, img.data(), err)) {
        if (error) *error = err;
        return RTX_ERR_IO;
    }
    return RTX_OK;
}
}  // namespace rtx

extern "C" {

rtx_host_scene *rtx_scene_builtin(const char *name, uint32_t width, const char *media_dir, uint64_t param) {
    if (!name || width == 0) {
        g_host_err = "scene name and a non-zero width are required";
        return nullptr;
    }
    auto *hs = new rtx_host_scene();
    try {
        if (!build_builtin(hs->arrays, name, width, media_dir ? media_dir : "assets", param)) {
            delete hs;
            return nullptr;
        }
    } catch (const std::exception &e) {
        g_host_err = e.what();
        delete hs;
        return nullptr;
    }
    if (hs->arrays.xforms.empty()) {
        M4 id = m4_identity();
        hs->arrays.add_xform(id, id);
    }
    if (hs->arrays.mat_root.empty()) hs->arrays.add_material(Material());
    hs->desc = hs->arrays.desc();
    return hs;
}
const rtx_scene_desc *rtx_scene_get_desc(const rtx_host_scene *s) { return s ? &s->desc : nullptr; }
void rtx_scene_free(rtx_host_scene *s) { delete s; }
const char *rtx_host_last_error(void) { return g_host_err.c_str(); }

int rtx_png_read_rgb8(const char *path, uint32_t *w, uint32_t *h, uint8_t **out_rgb) {
    if (!path || !w || !h || !out_rgb) return RTX_ERR_INVALID;
    Image img;
    if (!png_read_rgb8(path, img, g_host_err)) return RTX_ERR_IO;
    *w = img.w;
    *h = img.h;
    *out_rgb = (uint8_t *)std::malloc(img.rgb.size());
    if (!*out_rgb) return RTX_ERR_IO;
    std::memcpy(*out_rgb, img.rgb.data(), img.rgb.size());
    return RTX_OK;
}
int rtx_png_write_rgb8(const char *path, uint32_t w, uint32_t h, const uint8_t *rgb) {
    if (!path || !rgb || !w || !h) return RTX_ERR_INVALID;
    return png_write_rgb8(path, w, h, rgb, g_host_err) ? RTX_OK : RTX_ERR_IO;
}
void rtx_host_free(void *p) { std::free(p); }

int rtx_render_builtin_to_png(const char *scene, uint32_t width, const char *media_dir, uint64_t param, uint32_t spp,
                              uint32_t bounces, const char *path, uint32_t num_threads, int device) {
    if (!path) return RTX_ERR_INVALID;
    rtx_host_scene *hs = rtx_scene_builtin(scene, width, media_dir, param);
    if (!hs) return RTX_ERR_INVALID;
    rtx_ctx *ctx = nullptr;
    int rc = rtx_init(device, &ctx);
    if (rc != RTX_OK) {
        g_host_err = rtx_last_error(nullptr);
        rtx_scene_free(hs);
        return rc;
    }
    if (num_threads) (void)rtx_set_build_threads(ctx, (int)num_threads);
    rc = rtx_upload_scene(ctx, &hs->desc);
    // a scene file may carry its own `size W H` (scene_file.cpp): the image is the camera's, not width x width
    const uint32_t img_w = hs->desc.width, img_h = hs->desc.height;
    std::vector<uint8_t> img((size_t)img_w * img_h * 3);
    if (rc == RTX_OK) rc = rtx_render(ctx, spp, bounces, 0, nullptr, 0, img.data(), nullptr, nullptr);
    if (rc != RTX_OK) g_host_err = rtx_last_error(ctx);
    rtx_destroy(ctx);
    rtx_scene_free(hs);
    if (rc != RTX_OK) return rc;
    return png_write_rgb8(path, img_w, img_h, img.data(), g_host_err) ? RTX_OK : RTX_ERR_IO;
}

}  // extern "C"
