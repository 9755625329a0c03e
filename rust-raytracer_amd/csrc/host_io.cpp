// On-disk formats either side of the trace path: Wavefront OBJ in (object.rs:30-91 via the
// obj crate), PNG in (material.rs:40-47 via the image crate), PNG out (world.rs:76-81).
// libpng is not available in the image, so PNG is hand-written over zlib.
#include <zlib.h>

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
