// Scene description files: `--scene file:<path>` / rtx_scene_builtin("file:<path>", ...).
// The reference picks and edits its scene in source (main.rs:82-111, scene.rs); this is the
// same vocabulary -- Camera::new, Color, MaterialType, TextureType, Material, Object::new_*,
// Object::transform, Light::new_* -- as one statement per line:
//
//   # comment                          (blank lines and anything after '#' are ignored)
//   camera  EX EY EZ  AX AY AZ  UX UY UZ      Camera::new(w, h, eye, at, up)   default: scene.rs:218-226
//   size W H                                  viewport; default: -w squared
//   background R G B                          default: 0.2 grey (main.rs:111)
//   shader NAME phong KD KS SHININESS         MaterialType::new_phong
//   shader NAME reflective | none
//   shader NAME refractive INDEX
//   shader NAME mix W SHADER [W SHADER ...]   MaterialType::Composition (earlier shaders by name)
//   texture NAME flat R G B                   TextureType::new_flat
//   texture NAME image PATH                   TextureType::new_texture (PNG)
//   texture NAME none
//   material NAME SHADER TEXTURE              Material::new; one material entry per statement
//   sphere MATERIAL CX CY CZ RADIUS [OPS]
//   triangle MATERIAL A(3) B(3) C(3) [OPS]
//   quad MATERIAL A(3) B(3) C(3) D(3) [OPS]
//   mesh MATERIAL PATH [OPS]                  Object::new_mesh: OBJ or PLY, 3- and 4-gons
//   light point X Y Z R G B
//   light cone X Y Z DX DY DZ ANGLE_DEG R G B
//   light directional DX DY DZ R G B
//   light ambient R G B
//
// OPS build the object_to_world matrix as the product, left to right, of
//   translate X Y Z | scale SX SY SZ | rotate_x DEG | rotate_y DEG | rotate_z DEG
// e.g. `translate 1 .5 -2 rotate_x 90 translate -1 -.5 2` rotates about (1, .5, -2).
// Paths are relative to the scene file's directory; `media:NAME` means NAME in the media
// directory (--media).  Errors carry file:line.
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "host_scene.h"

namespace rtx {

namespace {

struct Line {
    std::vector<std::string> tok;
    size_t pos = 0;
    bool more() const { return pos < tok.size(); }
    const std::string &word(const char *what) {
        if (!more()) throw std::runtime_error(std::string("missing ") + what);
        return tok[pos++];
    }
    float num(const char *what) {
        const std::string &w = word(what);
        char *end = nullptr;
        float v = std::strtof(w.c_str(), &end);
        if (end == w.c_str() || *end != '\0') throw std::runtime_error("'" + w + "' is not a number (" + what + ")");
        return v;
    }
    V3 vec(const char *what) {
        float x = num(what), y = num(what), z = num(what);
        return v3(x, y, z);
    }
    Color color() {
        float r = num("red"), g = num("green"), b = num("blue");
        return Color::rgb(r, g, b);
    }
    void end() {
        if (more()) throw std::runtime_error("unexpected '" + tok[pos] + "'");
    }
};

template <class T>
const T &lookup(const std::map<std::string, T> &m, const std::string &name, const char *kind) {
    auto it = m.find(name);
    if (it == m.end()) throw std::runtime_error(std::string("unknown ") + kind + " '" + name + "'");
    return it->second;
}

M4 parse_ops(Line &l) {
    M4 m = m4_identity();
    while (l.more()) {
        const std::string op = l.word("transform");
        if (op == "translate") m = m * m4_from_translation(l.vec("translation"));
        else if (op == "scale") {
            float x = l.num("scale"), y = l.num("scale"), z = l.num("scale");
            m = m * m4_from_nonuniform_scale(x, y, z);
        } else if (op == "rotate_x") m = m * m4_from_angle_x(l.num("angle"));
        else if (op == "rotate_y") m = m * m4_from_angle_y(l.num("angle"));
        else if (op == "rotate_z") m = m * m4_from_angle_z(l.num("angle"));
        else throw std::runtime_error("unknown transform '" + op + "'");
    }
    return m;
}

}  // namespace

bool load_scene_file(SceneArrays &s, const std::string &path, uint32_t width, const std::string &media_dir, std::string &err) {
    std::ifstream f(path);
    if (!f) {
        err = "cannot open " + path;
        return false;
    }
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : path.substr(0, slash);
    auto resolve = [&](const std::string &p) {
        if (p.rfind("media:", 0) == 0) return media_dir + "/" + p.substr(6);
        return (!p.empty() && p[0] == '/') ? p : dir + "/" + p;
    };
    std::map<std::string, MaterialType> shaders;
    std::map<std::string, TextureType> textures;
    std::map<std::string, Material> materials;
    bool have_camera = false;
    V3 eye = v3(0, 0, 0), at = v3(0, 0, -1), up = v3(0, 1, 0);
    uint32_t w = width, h = width;
    std::string text;
    size_t lineno = 0;
    try {
        while (std::getline(f, text)) {
            lineno++;
            const size_t hash = text.find('#');
            if (hash != std::string::npos) text.resize(hash);
            Line l;
            std::istringstream ss(text);
            for (std::string t; ss >> t;) l.tok.push_back(t);
            if (!l.more()) continue;
            const std::string kw = l.word("statement");
            if (kw == "camera") {
                eye = l.vec("eye");
                at = l.vec("at");
                up = l.vec("up");
                have_camera = true;
                l.end();
            } else if (kw == "size") {
                float fw = l.num("width"), fh = l.num("height");
                if (!(fw >= 1.0f && fw <= 65536.0f && fh >= 1.0f && fh <= 65536.0f)) throw std::runtime_error("size out of range");
                w = (uint32_t)fw;
                h = (uint32_t)fh;
                l.end();
            } else if (kw == "background") {
                s.set_background(l.color());
                l.end();
            } else if (kw == "shader") {
                const std::string name = l.word("shader name"), kind = l.word("shader kind");
                MaterialType m;
                if (kind == "phong") {
                    float kd = l.num("diffuse"), ks = l.num("specular"), sh = l.num("shininess");
                    m = MaterialType::new_phong(kd, ks, sh);
                } else if (kind == "reflective") m = MaterialType::Reflective();
                else if (kind == "refractive") m = MaterialType::Refractive(l.num("index"));
                else if (kind == "none") m = MaterialType::None();
                else if (kind == "mix") {
                    std::vector<std::pair<MaterialType, float>> parts;
                    while (l.more()) {
                        float wgt = l.num("weight");
                        parts.push_back({lookup(shaders, l.word("shader"), "shader"), wgt});
                    }
                    if (parts.empty()) throw std::runtime_error("mix needs at least one 'WEIGHT SHADER' pair");
                    m = MaterialType::Composition(std::move(parts));
                } else throw std::runtime_error("unknown shader kind '" + kind + "'");
                l.end();
                shaders[name] = m;
            } else if (kw == "texture") {
                const std::string name = l.word("texture name"), kind = l.word("texture kind");
                TextureType t;
                if (kind == "flat") t = TextureType::new_flat(l.color());
                else if (kind == "image") t = TextureType::new_texture(resolve(l.word("image path")));
                else if (kind == "none") t = TextureType::None();
                else throw std::runtime_error("unknown texture kind '" + kind + "'");
                l.end();
                textures[name] = t;
            } else if (kw == "material") {
                const std::string name = l.word("material name");
                const MaterialType &m = lookup(shaders, l.word("shader"), "shader");
                const TextureType &t = lookup(textures, l.word("texture"), "texture");
                l.end();
                materials[name] = Material::make(m, t);
            } else if (kw == "sphere") {
                const Material &m = lookup(materials, l.word("material"), "material");
                V3 c = l.vec("centre");
                float r = l.num("radius");
                s.add_object(Object::new_sphere(c, r, m).transform(parse_ops(l)));
            } else if (kw == "triangle") {
                const Material &m = lookup(materials, l.word("material"), "material");
                V3 a = l.vec("a"), b = l.vec("b"), c = l.vec("c");
                s.add_object(Object::new_triangle(a, b, c, m).transform(parse_ops(l)));
            } else if (kw == "quad") {
                const Material &m = lookup(materials, l.word("material"), "material");
                V3 a = l.vec("a"), b = l.vec("b"), c = l.vec("c"), d = l.vec("d");
                s.add_object(Object::new_quad(a, b, c, d, m).transform(parse_ops(l)));
            } else if (kw == "mesh") {
                const Material &m = lookup(materials, l.word("material"), "material");
                const std::string file = resolve(l.word("mesh path"));
                ObjMesh mesh;
                std::string merr;
                if (!load_mesh(file, mesh, merr)) throw std::runtime_error(merr);
                s.add_mesh(mesh, parse_ops(l), m);
            } else if (kw == "light") {
                const std::string kind = l.word("light kind");
                if (kind == "point") {
                    V3 p = l.vec("position");
                    s.add_light(Light::new_point(p, l.color()));
                } else if (kind == "cone") {
                    V3 p = l.vec("position"), d = l.vec("direction");
                    float a = l.num("angle");
                    s.add_light(Light::new_cone(p, d, a, l.color()));
                } else if (kind == "directional") {
                    V3 d = l.vec("direction");
                    s.add_light(Light::new_directional(d, l.color()));
                } else if (kind == "ambient") s.add_light(Light::new_ambient(l.color()));
                else throw std::runtime_error("unknown light kind '" + kind + "'");
                l.end();
            } else {
                throw std::runtime_error("unknown statement '" + kw + "'");
            }
        }
        if (have_camera) s.set_camera(Camera::make(w, h, eye, at, up));
        else if (w != width || h != width)  // scene.rs:218-226's camera with the file's size
            s.set_camera(Camera::make(w, h, v3(0.0f, 1.5f, 5.0f), v3(0, 0, 0), v3(0, 1, 0)));
    } catch (const std::exception &e) {
        err = path + ":" + std::to_string(lineno) + ": " + e.what();
        return false;
    }
    return true;
}

}  // namespace rtx
