// On-disk formats either side of the trace path: Wavefront OBJ in (object.rs:30-91 via the
// obj crate), PNG in (material.rs:40-47 via the image crate), PNG out (world.rs:76-81).
// libpng is not available in the image, so PNG is hand-written over zlib.
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "host_scene.h"

namespace rtx {

// ------------------------------------------------------------------ OBJ
bool load_obj(const std::string &path, ObjMesh &out, std::string &err) {
    std::ifstream f(path);
    if (!f) {
        err = "cannot open " + path;
        return false;
    }
    out.positions.clear();
    out.polys.clear();
    std::string line;
    size_t lineno = 0;
    while (std::getline(f, line)) {
        lineno++;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v") {
            float x, y, z;
            if (!(ss >> x >> y >> z)) {
                err = path + ":" + std::to_string(lineno) + ": bad vertex";
                return false;
            }
            out.positions.push_back(x);
            out.positions.push_back(y);
            out.positions.push_back(z);
        } else if (tag == "f") {
            std::vector<uint32_t> poly;
            std::string tok;
            while (ss >> tok) {
                // position index is the part before the first '/'
                long idx = std::strtol(tok.c_str(), nullptr, 10);
                long nverts = (long)(out.positions.size() / 3);
                if (idx < 0) idx = nverts + idx + 1;  // relative index
                if (idx < 1 || idx > nverts) {
                    err = path + ":" + std::to_string(lineno) + ": face index out of range";
                    return false;
                }
                poly.push_back((uint32_t)(idx - 1));
            }
            if (poly.size() < 3) {
                err = path + ":" + std::to_string(lineno) + ": face with fewer than 3 vertices";
                return false;
            }
            out.polys.push_back(std::move(poly));
        }
        // vt, vn, o, g, s, usemtl, mtllib: ignored like object.rs:55-57 does
    }
    return true;
}

// ------------------------------------------------------------------ PLY
// The format of the Stanford scans behind the reference's headline render (README.md:7;
// bun_zipper.ply, dragon_vrip.ply, ...): ascii and binary (little / big endian), a `vertex`
// element with x, y, z among its scalar properties, a `face` element with one index list.
// Other elements and properties are skipped.
namespace {
struct PlyProp {
    std::string name;
    int type = 0;       // scalar type (value type of a list)
    int count_type = -1;  // >= 0: list property
};
struct PlyElem {
    std::string name;
    size_t count = 0;
    std::vector<PlyProp> props;
};
// 0 int8 1 uint8 2 int16 3 uint16 4 int32 5 uint32 6 float32 7 float64
int ply_type(const std::string &t) {
    static const char *names[8][2] = {{"char", "int8"},   {"uchar", "uint8"}, {"short", "int16"},  {"ushort", "uint16"},
                                      {"int", "int32"},   {"uint", "uint32"}, {"float", "float32"}, {"double", "float64"}};
    for (int i = 0; i < 8; i++)
        if (t == names[i][0] || t == names[i][1]) return i;
    return -1;
}
const int PLY_SIZE[8] = {1, 1, 2, 2, 4, 4, 4, 8};
double ply_decode(const uint8_t *p, int type, bool big) {
    uint8_t b[8];
    int n = PLY_SIZE[type];
    for (int i = 0; i < n; i++) b[i] = big ? p[n - 1 - i] : p[i];
    switch (type) {
        case 0: { int8_t v; std::memcpy(&v, b, 1); return v; }
        case 1: return b[0];
        case 2: { int16_t v; std::memcpy(&v, b, 2); return v; }
        case 3: { uint16_t v; std::memcpy(&v, b, 2); return v; }
        case 4: { int32_t v; std::memcpy(&v, b, 4); return v; }
        case 5: { uint32_t v; std::memcpy(&v, b, 4); return v; }
        case 6: { float v; std::memcpy(&v, b, 4); return v; }
        default: { double v; std::memcpy(&v, b, 8); return v; }
    }
}
}  // namespace

bool load_ply(const std::string &path, ObjMesh &out, std::string &err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        err = "cannot open " + path;
        return false;
    }
    out.positions.clear();
    out.polys.clear();
    std::string line;
    if (!std::getline(f, line) || line.substr(0, 3) != "ply") {
        err = path + ": not a PLY file";
        return false;
    }
    int format = -1;  // 0 ascii, 1 binary little endian, 2 binary big endian
    std::vector<PlyElem> elems;
    bool header_done = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "format") {
            std::string fm;
            ss >> fm;
            format = fm == "ascii" ? 0 : fm == "binary_little_endian" ? 1 : fm == "binary_big_endian" ? 2 : -1;
        } else if (tag == "element") {
            PlyElem e;
            ss >> e.name >> e.count;
            elems.push_back(e);
        } else if (tag == "property") {
            if (elems.empty()) {
                err = path + ": property before any element";
                return false;
            }
            PlyProp pr;
            std::string t;
            ss >> t;
            if (t == "list") {
                std::string ct, vt;
                ss >> ct >> vt >> pr.name;
                pr.count_type = ply_type(ct);
                pr.type = ply_type(vt);
                if (pr.count_type < 0 || pr.count_type > 5) pr.type = -1;
            } else {
                pr.type = ply_type(t);
                ss >> pr.name;
            }
            if (pr.type < 0) {
                err = path + ": unsupported property type in '" + line + "'";
                return false;
            }
            elems.back().props.push_back(pr);
        } else if (tag == "end_header") {
            header_done = true;
            break;
        }  // comment, obj_info: ignored
    }
    if (!header_done || format < 0) {
        err = path + ": bad PLY header";
        return false;
    }
    const bool big = format == 2;
    std::vector<uint8_t> buf(8);
    // reads one scalar of `type` (ascii: next token)
    auto scalar = [&](int type, double &v) -> bool {
        if (format == 0) return (bool)(f >> v);
        if (!f.read(reinterpret_cast<char *>(buf.data()), PLY_SIZE[type])) return false;
        v = ply_decode(buf.data(), type, big);
        return true;
    };
    for (const PlyElem &e : elems) {
        const bool is_vertex = e.name == "vertex", is_face = e.name == "face";
        int ix = -1, iy = -1, iz = -1, ilist = -1;
        for (size_t k = 0; k < e.props.size(); k++) {
            const PlyProp &pr = e.props[k];
            if (is_vertex && pr.count_type < 0) {
                if (pr.name == "x") ix = (int)k;
                if (pr.name == "y") iy = (int)k;
                if (pr.name == "z") iz = (int)k;
            }
            if (is_face && pr.count_type >= 0 && ilist < 0 && (pr.name == "vertex_indices" || pr.name == "vertex_index")) ilist = (int)k;
        }
        if (is_vertex && (ix < 0 || iy < 0 || iz < 0)) {
            err = path + ": vertex element without x, y, z";
            return false;
        }
        if (is_face && ilist < 0) {
            err = path + ": face element without a vertex_indices list";
            return false;
        }
        const size_t plausible = std::min<size_t>(e.count, (size_t)1 << 26);  // a lying header must not reserve the world
        if (is_vertex) out.positions.reserve(out.positions.size() + plausible * 3);
        if (is_face) out.polys.reserve(out.polys.size() + plausible);
        for (size_t i = 0; i < e.count; i++) {
            float xyz[3] = {0, 0, 0};
            std::vector<uint32_t> poly;
            for (size_t k = 0; k < e.props.size(); k++) {
                const PlyProp &pr = e.props[k];
                double v;
                if (pr.count_type < 0) {
                    if (!scalar(pr.type, v)) goto truncated;
                    if ((int)k == ix) xyz[0] = (float)v;
                    if ((int)k == iy) xyz[1] = (float)v;
                    if ((int)k == iz) xyz[2] = (float)v;
                } else {
                    if (!scalar(pr.count_type, v) || v < 0 || v > 1e6) goto truncated;
                    size_t n = (size_t)v;
                    for (size_t j = 0; j < n; j++) {
                        if (!scalar(pr.type, v)) goto truncated;
                        if ((int)k == ilist) {
                            if (v < 0 || v >= (double)(out.positions.size() / 3)) {
                                err = path + ": face index out of range";
                                return false;
                            }
                            poly.push_back((uint32_t)v);
                        }
                    }
                }
            }
            if (is_vertex) out.positions.insert(out.positions.end(), xyz, xyz + 3);
            if (is_face) {
                if (poly.size() < 3) {
                    err = path + ": face with fewer than 3 vertices";
                    return false;
                }
                out.polys.push_back(std::move(poly));
            }
        }
    }
    return true;
truncated:
    err = path + ": truncated or malformed PLY body";
    return false;
}

// By file extension: .ply -> load_ply, anything else -> load_obj.
bool load_mesh(const std::string &path, ObjMesh &out, std::string &err) {
    std::string ext = path.size() >= 4 ? path.substr(path.size() - 4) : "";
    for (auto &c : ext) c = (char)std::tolower((unsigned char)c);
    return ext == ".ply" ? load_ply(path, out, err) : load_obj(path, out, err);
}

// ------------------------------------------------------------------ PNG
namespace {
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(x >> 24);
    v.push_back(x >> 16);
    v.push_back(x >> 8);
    v.push_back(x);
}
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}
}  // namespace

bool png_read_rgb8(const std::string &path, Image &out, std::string &err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        err = "cannot open " + path;
        return false;
    }
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (buf.size() < 8 || std::memcmp(buf.data(), sig, 8) != 0) {
        err = path + ": not a PNG";
        return false;
    }
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    size_t pos = 8;
    bool have_ihdr = false;
    while (pos + 12 <= buf.size()) {
        uint32_t len = be32(&buf[pos]);
        const uint8_t *type = &buf[pos + 4];
        const uint8_t *data = &buf[pos + 8];
        if (pos + 12 + (size_t)len > buf.size()) {
            err = path + ": truncated chunk";
            return false;
        }
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0) {
        err = path + ": missing IHDR";
        return false;
    }
    if (w > 65536u || h > 65536u || (uint64_t)w * h > (1ull << 28)) {
        err = path + ": image dimensions out of range";
        return false;
    }
    if (depth != 8 || interlace != 0) {
        err = path + ": only 8-bit non-interlaced PNG is supported";
        return false;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) {
        err = path + ": unsupported colour type";
        return false;
    }
    size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) {
        err = path + ": zlib inflate failed";
        return false;
    }
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; y++) {
        int ft = raw[(stride + 1) * y];
        const uint8_t *src = &raw[(stride + 1) * y + 1];
        uint8_t *dst = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= (size_t)ch ? dst[i - ch] : 0;
            int b = up ? up[i] : 0;
            int c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int x = src[i];
            switch (ft) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) / 2; break;
                case 4: x += paeth(a, b, c); break;
                default: err = path + ": bad filter type"; return false;
            }
            dst[i] = (uint8_t)x;
        }
    }
    out.w = w;
    out.h = h;
    out.rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t *p = &img[i * ch];
        uint8_t r, g, b;
        if (ctype == 2 || ctype == 6) {
            r = p[0];
            g = p[1];
            b = p[2];
        } else if (ctype == 3) {
            size_t k = (size_t)p[0] * 3;
            if (k + 3 > plte.size()) {
                err = path + ": palette index out of range";
                return false;
            }
            r = plte[k];
            g = plte[k + 1];
            b = plte[k + 2];
        } else {
            r = g = b = p[0];
        }
        out.rgb[i * 3] = r;
        out.rgb[i * 3 + 1] = g;
        out.rgb[i * 3 + 2] = b;
    }
    return true;
}

bool png_write_rgb8(const std::string &path, uint32_t w, uint32_t h, const uint8_t *rgb, std::string &err) {
    size_t stride = (size_t)w * 3;
    std::vector<uint8_t> raw((stride + 1) * h);
    for (uint32_t y = 0; y < h; y++) {
        raw[(stride + 1) * y] = 0;
        std::memcpy(&raw[(stride + 1) * y + 1], rgb + stride * y, stride);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
        err = "zlib deflate failed";
        return false;
    }
    comp.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto chunk = [&](const char *type, const std::vector<uint8_t> &data) {
        put32(out, (uint32_t)data.size());
        size_t start = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        uint32_t crc = (uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start));
        put32(out, crc);
    };
    std::vector<uint8_t> ihdr;
    put32(ihdr, w);
    put32(ihdr, h);
    ihdr.push_back(8);
    ihdr.push_back(2);
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    chunk("IHDR", ihdr);
    chunk("IDAT", comp);
    chunk("IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f) {
        err = "cannot write " + path;
        return false;
    }
    f.write((const char *)out.data(), (std::streamsize)out.size());
    if (!f) {
        err = "write failed: " + path;
        return false;
    }
    return true;
}

}  // namespace rtx
