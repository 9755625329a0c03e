#include "flatten.h"

#include <cmath>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <functional>
#include <thread>

namespace rtx {

namespace {

struct Emit {
    const BuiltBvh &bvh;
    FlatScene &out;

    static float sa(const BuildNode &n) {  // AABB::surface_area (bvh.rs:147-150)
        float dx = n.mx[0] - n.mn[0], dy = n.mx[1] - n.mn[1], dz = n.mx[2] - n.mn[2];
        return 2.0f * dx * dy + 2.0f * dx * dz + 2.0f * dz * dy;
    }

    // Returns the child reference of build node `bn`; accumulates total_sa in the
    // reference's order: sa(node) + total(left) + total(right).
    uint32_t emit(int32_t bn, float &total) {
        const BuildNode &n = bvh.nodes[bn];
        out.preorder_sizes.push_back(n.count);
        if (n.left < 0) {
            out.n_leaves++;
            total = sa(n);
            return LEAF_BIT | (n.first << 3) | n.count;
        }
        uint32_t idx = (uint32_t)out.nodes.size();
        out.nodes.emplace_back();
        float tl, tr;
        uint32_t l = emit(n.left, tl);
        uint32_t r = emit(n.right, tr);
        const BuildNode &L = bvh.nodes[n.left], &R = bvh.nodes[n.right];
        BvhNode &o = out.nodes[idx];
        for (int k = 0; k < 3; k++) {
            o.lmin[k] = L.mn[k];
            o.lmax[k] = L.mx[k];
            o.rmin[k] = R.mn[k];
            o.rmax[k] = R.mx[k];
        }
        o.left = l;
        o.right = r;
        o.pad[0] = o.pad[1] = 0;
        total = sa(n) + tl + tr;
        return idx;
    }
};

// Wide (up to 4 slots) view of the same tree; see WideNode in device_types.h.
struct EmitWide {
    const BuiltBvh &bvh;
    FlatScene &out;
    const std::vector<float> &node_ovh;  // per reference node: largest hit overhang in its subtree

    static float sa(const BuildNode &n) {
        float dx = n.mx[0] - n.mn[0], dy = n.mx[1] - n.mn[1], dz = n.mx[2] - n.mn[2];
        return 2.0f * dx * dy + 2.0f * dx * dz + 2.0f * dz * dy;
    }
    static uint32_t leaf_ref(const BuildNode &n) { return LEAF_BIT | (n.first << 3) | n.count; }

    // Iterative (explicit work list): the tree can be deep.  The first TOP_BFS nodes are
    // numbered breadth-first (the kernels keep the top of the tree in LDS, indexed by node
    // number), the rest depth-first for locality.
    void run(int32_t root) {
        struct Item {
            int32_t bn;
            uint32_t widx, level;
        };
        constexpr size_t TOP_BFS = 1024;
        std::deque<Item> work;
        out.wnodes.emplace_back();
        work.push_back({root, 0, 1});
        while (!work.empty()) {
            Item it;
            if (out.wnodes.size() < TOP_BFS) {  // nodes are numbered when their parent is processed
                it = work.front();
                work.pop_front();
            } else {
                it = work.back();
                work.pop_back();
            }
            if (it.level > out.wide_depth) out.wide_depth = it.level;
            const BuildNode &n = bvh.nodes[it.bn];
            int32_t slots[4] = {n.left, n.right, -1, -1};
            int ns = 2;
            while (ns < 4) {  // expand the internal slot of largest surface area
                int best = -1;
                float best_sa = -1.0f;
                for (int k = 0; k < ns; k++) {
                    const BuildNode &c = bvh.nodes[slots[k]];
                    if (c.left < 0) continue;
                    float a = sa(c);
                    if (!(a == a)) a = INFINITY;
                    if (best < 0 || a > best_sa) {
                        best = k;
                        best_sa = a;
                    }
                }
                if (best < 0) break;
                const BuildNode &c = bvh.nodes[slots[best]];
                // keep left-to-right order: the two children take the place of their parent
                for (int k = ns; k > best + 1; k--) slots[k] = slots[k - 1];
                slots[best] = c.left;
                slots[best + 1] = c.right;
                ns++;
            }
            WideNode w;
            std::memset(&w, 0, sizeof(w));
            // an empty slot holds a NaN box: the kernels' slab test can never report it hit (node_step)
            const float qnan = std::numeric_limits<float>::quiet_NaN();
            for (int k = 0; k < 4; k++) {
                w.ref[k] = WIDE_EMPTY;
                w.mnx[k] = w.mny[k] = w.mnz[k] = w.mxx[k] = w.mxy[k] = w.mxz[k] = qnan;
            }
            for (int k = 0; k < ns; k++) {
                const BuildNode &c = bvh.nodes[slots[k]];
                for (int a = 0; a < 3; a++)
                    if (!std::isfinite(c.mn[a]) || !std::isfinite(c.mx[a])) out.nonfinite_boxes = true;
                w.mnx[k] = c.mn[0];
                w.mny[k] = c.mn[1];
                w.mnz[k] = c.mn[2];
                w.mxx[k] = c.mx[0];
                w.mxy[k] = c.mx[1];
                w.mxz[k] = c.mx[2];
                w.ovh[k] = node_ovh[slots[k]];
                if (c.left < 0) {
                    w.ref[k] = leaf_ref(c);
                } else {
                    uint32_t ci = (uint32_t)out.wnodes.size();
                    out.wnodes.emplace_back();
                    w.ref[k] = ci;
                    work.push_back({slots[k], ci, it.level + 1});
                }
            }
            out.wnodes[it.widx] = w;
        }
    }
};

// Material tree -> post-order program (material.rs:118-206).
int emit_material(const rtx_scene_desc &d, int node, int depth, std::vector<MatInstr> &prog, uint32_t &n_bounce,
                  std::string &err) {
    if (node < 0 || (uint32_t)node >= d.n_mat_nodes) {
        err = "material node index out of range";
        return RTX_ERR_INVALID;
    }
    MatInstr ins;
    std::memset(&ins, 0, sizeof(ins));
    int type = d.node_type[node];
    const float *p = d.node_params + (size_t)node * 3;
    switch (type) {
        case RTX_MAT_NONE:
            ins.op = OP_NONE;
            prog.push_back(ins);
            return RTX_OK;
        case RTX_MAT_PHONG:
            ins.op = OP_PHONG;
            ins.p0 = p[0];
            ins.p1 = p[1];
            ins.p2 = p[2];
            prog.push_back(ins);
            return RTX_OK;
        case RTX_MAT_REFLECTIVE:
        case RTX_MAT_REFRACTIVE:
            ins.op = type == RTX_MAT_REFLECTIVE ? OP_REFLECT : OP_REFRACT;
            ins.arg = n_bounce++;
            ins.p0 = p[0];
            prog.push_back(ins);
            return RTX_OK;
        case RTX_MAT_COMPOSITION: {
            if (depth >= MAX_COMP_DEPTH) {
                err = "Composition nested deeper than 4";
                return RTX_ERR_CAPACITY;
            }
            ins.op = OP_BEGIN;
            prog.push_back(ins);
            int first = d.node_child_first[node], cnt = d.node_child_count[node];
            if (first < 0 || cnt < 0 || (uint32_t)(first + cnt) > d.n_mat_children) {
                err = "material child range out of bounds";
                return RTX_ERR_INVALID;
            }
            for (int k = 0; k < cnt; k++) {
                int rc = emit_material(d, d.child_node[first + k], depth + 1, prog, n_bounce, err);
                if (rc != RTX_OK) return rc;
                MatInstr it;
                std::memset(&it, 0, sizeof(it));
                it.op = OP_ITEM;
                it.p0 = d.child_coef[first + k];
                prog.push_back(it);
            }
            MatInstr en;
            std::memset(&en, 0, sizeof(en));
            en.op = OP_END;
            prog.push_back(en);
            return RTX_OK;
        }
        default: err = "unknown material node type"; return RTX_ERR_INVALID;
    }
}

}  // namespace

// The reference tests an object only when the ray passes its bounding box, but two kinds of
// object can be HIT outside that box, and the production traversal's pruning (which compares
// box entry / exit t with hit t) has to know by how much:
//  * a quad is bounded by its four vertices (object.rs:284-299) but intersected on the plane
//    through a with normal (b-a)x(d-a), inside the four edges seen along that normal
//    (object.rs:167-189): a non-planar quad's region reaches to the projection c' of c onto
//    that plane, which can lie outside the vertices' box;
//  * a sphere is bounded by centre +- radius * |M e_k| (column norms, utils.rs:33-44), while the
//    transformed sphere extends radius * |row_k(M)| along world axis k: different as soon as
//    a rotation meets a non-uniform scale.
// Returns the largest such excess over the objects, in world units (0 for triangles, planar
// quads, spheres under rotation + uniform scale: every scene of scene.rs except Suzanne's
// slightly non-planar quads); infinity when a transform is not affine.
float hit_overhang(const rtx_scene_desc &d, std::vector<float> &per_object) {
    double worst = 0.0;
    per_object.assign(d.n_objects, 0.0f);
    for (uint32_t o = 0; o < d.n_objects; o++) {
        const int type = d.obj_type[o];
        if (type == RTX_OBJ_TRIANGLE) continue;
        const float *v = d.obj_verts + (size_t)o * 12;
        const float *M = d.xforms + (size_t)d.obj_xform[o] * 32;  // column-major object_to_world
        if (M[3] != 0.0f || M[7] != 0.0f || M[11] != 0.0f || M[15] != 1.0f) {
            per_object.assign(d.n_objects, INFINITY);
            return INFINITY;
        }
        auto world = [&](const double p[3], double w[3]) {
            for (int k = 0; k < 3; k++) w[k] = (double)M[k] * p[0] + (double)M[4 + k] * p[1] + (double)M[8 + k] * p[2] + (double)M[12 + k];
        };
        float mn[3], mx[3];
        object_bounding_box(d, o, mn, mx);
        double excess = 0.0;
        if (type == RTX_OBJ_SPHERE) {
            for (int k = 0; k < 3; k++) {
                double row = std::sqrt((double)M[k] * M[k] + (double)M[4 + k] * M[4 + k] + (double)M[8 + k] * M[8 + k]);
                double c[3] = {v[0], v[1], v[2]}, wc[3];
                world(c, wc);
                excess = std::max(excess, std::max(wc[k] + std::fabs((double)v[3]) * row - (double)mx[k],
                                                   (double)mn[k] - (wc[k] - std::fabs((double)v[3]) * row)));
            }
            // rounding of the box itself is not an overhang
            if (excess < 4e-7 * (std::fabs((double)v[3]) + std::fabs((double)mx[0]) + std::fabs((double)mx[1]) + std::fabs((double)mx[2]))) excess = 0.0;
        } else {
            double a[3], b[3], c[3], e[3], n[3];
            for (int k = 0; k < 3; k++) {
                a[k] = v[k];
                b[k] = v[3 + k];
                c[k] = v[6 + k];
                e[k] = v[9 + k];
            }
            const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {e[0] - a[0], e[1] - a[1], e[2] - a[2]};
            n[0] = u[1] * w[2] - u[2] * w[1];
            n[1] = u[2] * w[0] - u[0] * w[2];
            n[2] = u[0] * w[1] - u[1] * w[0];
            const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (!(len > 0.0)) continue;  // degenerate normal: the NaN normal never reports a hit
            const double off = ((c[0] - a[0]) * n[0] + (c[1] - a[1]) * n[1] + (c[2] - a[2]) * n[2]) / len;
            double cp[3] = {c[0] - off * n[0] / len, c[1] - off * n[1] / len, c[2] - off * n[2] / len}, wcp[3];
            world(cp, wcp);
            for (int k = 0; k < 3; k++) excess = std::max(excess, std::max(wcp[k] - (double)mx[k], (double)mn[k] - wcp[k]));
            double size = 0.0;
            for (int k = 0; k < 3; k++) size = std::max(size, std::max(std::fabs((double)mn[k]), std::fabs((double)mx[k])));
            if (excess < 4e-7 * size) excess = 0.0;  // a planar quad: f32 rounding of the vertices only
        }
        per_object[o] = !(excess == excess) ? INFINITY : (excess == 0.0 ? 0.0f : (float)(excess * 1.0001 + 1e-7));
        if (!(excess <= worst)) worst = excess;  // NaN propagates to "no pruning"
    }
    if (!(worst == worst)) return INFINITY;
    return worst == 0.0 ? 0.0f : (float)(worst * 1.0001 + 1e-7);
}

int flatten_scene(const rtx_scene_desc &d, const BuiltBvh &bvh, FlatScene &out, std::string &err, int threads) {
    out = FlatScene();
    const bool timing = std::getenv("RTX_FLATTEN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "flatten %-12s %.3f s\n", what, std::chrono::duration<double>(t - t_last).count());
        t_last = t;
    };
    if (d.n_xforms == 0 || d.n_xforms > 8192) {
        err = "scene needs between 1 and 8192 transforms";
        return d.n_xforms == 0 ? RTX_ERR_INVALID : RTX_ERR_CAPACITY;
    }
    if (d.n_materials == 0 || d.n_materials > 65536) {
        err = "scene needs between 1 and 65536 materials";
        return d.n_materials == 0 ? RTX_ERR_INVALID : RTX_ERR_CAPACITY;
    }
    if (d.n_lights > (uint32_t)MAX_LIGHTS) {
        err = "more than 16 lights";
        return RTX_ERR_CAPACITY;
    }
    if (d.n_objects >= (1u << 28)) {
        err = "more than 2^28 objects";
        return RTX_ERR_CAPACITY;
    }
    out.xforms.resize(d.n_xforms);
    std::vector<uint8_t> xf_identity(d.n_xforms);
    for (uint32_t i = 0; i < d.n_xforms; i++) {
        std::memcpy(out.xforms[i].o2w, d.xforms + (size_t)i * 32, 64);
        std::memcpy(out.xforms[i].w2o, d.xforms + (size_t)i * 32 + 16, 64);
        bool id = true;
        for (int k = 0; k < 32; k++) {
            float want = ((k & 15) % 5 == 0) ? 1.0f : 0.0f;
            if (d.xforms[(size_t)i * 32 + k] != want || std::signbit(d.xforms[(size_t)i * 32 + k])) id = false;
        }
        xf_identity[i] = id;
    }
    out.xf0_identity = xf_identity[0];
    // Do many objects carry a transform?  The path kernels come in two variants (kernels.hip, RayCtx): the one without
    // the cache computes the object-space ray of such an object per test, which costs nothing when they are the five
    // objects of load_basic beside 2.4 M mesh triangles (-3 % on the headline frame) and 4 % when they are the mesh.
    {
        uint64_t with = 0;
        for (uint32_t i = 0; i < d.n_objects; i++) {
            const int xf = d.obj_xform[i];
            if (xf >= 0 && (uint32_t)xf < d.n_xforms && !xf_identity[xf]) with++;
        }
        out.any_xform = with * 8 > d.n_objects;
    }
    // materials
    out.mats.resize(d.n_materials);
    std::vector<uint32_t> mat_bounce(d.n_materials);
    for (uint32_t i = 0; i < d.n_materials; i++) {
        MaterialRec &m = out.mats[i];
        m.prog_first = (uint32_t)out.prog.size();
        uint32_t nb = 0;
        int rc = emit_material(d, d.mat_root[i], 0, out.prog, nb, err);
        if (rc != RTX_OK) return rc;
        if (nb > (uint32_t)MAX_BOUNCE_LEAVES) {
            err = "material with more than 4 reflective/refractive leaves";
            return RTX_ERR_CAPACITY;
        }
        m.prog_count = (uint32_t)out.prog.size() - m.prog_first;
        m.n_bounce = nb;
        mat_bounce[i] = nb;
        if (nb != 0u) {
            if (out.bounce_mat_lo > out.bounce_mat_hi) out.bounce_mat_lo = out.bounce_mat_hi = i;
            out.bounce_mat_hi = i;
        }
        if (d.mat_tex_kind[i] < RTX_TEX_NONE || d.mat_tex_kind[i] > RTX_TEX_IMAGE) {
            err = "unknown texture kind";
            return RTX_ERR_INVALID;
        }
        m.tex_kind = (uint32_t)d.mat_tex_kind[i];
        m.tex_id = (uint32_t)d.mat_tex_id[i];
        if (m.tex_kind == RTX_TEX_IMAGE && (d.mat_tex_id[i] < 0 || m.tex_id >= d.n_textures)) {
            err = "texture index out of range";
            return RTX_ERR_INVALID;
        }
        for (int k = 0; k < 3; k++) {  // Color::rgb clamps on construction (color.rs:17-23)
            float c = d.mat_tex_color[i * 3 + k];
            m.tex_color[k] = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
        }
    }
    // textures
    size_t tex_bytes = 0;
    out.tex.resize(d.n_textures);
    for (uint32_t i = 0; i < d.n_textures; i++) {
        if (d.tex_w[i] <= 0 || d.tex_h[i] <= 0) {
            err = "empty texture";
            return RTX_ERR_INVALID;
        }
        out.tex[i].w = (uint32_t)d.tex_w[i];
        out.tex[i].h = (uint32_t)d.tex_h[i];
        out.tex[i].offset = d.tex_offset[i];
        size_t end = d.tex_offset[i] + (size_t)d.tex_w[i] * d.tex_h[i] * 3;
        if (end > tex_bytes) tex_bytes = end;
    }
    if (tex_bytes) out.tex_data.assign(d.tex_data, d.tex_data + tex_bytes);
    // lights
    out.lights.resize(d.n_lights);
    for (uint32_t i = 0; i < d.n_lights; i++) {
        LightRec &l = out.lights[i];
        if (d.light_type[i] < RTX_LIGHT_AMBIENT || d.light_type[i] > RTX_LIGHT_CONE) {
            err = "unknown light type";
            return RTX_ERR_INVALID;
        }
        l.type = (uint32_t)d.light_type[i];
        for (int k = 0; k < 3; k++) {
            l.pos[k] = d.light_pos[i * 3 + k];
            l.dir[k] = d.light_dir[i * 3 + k];
            float c = d.light_color[i * 3 + k];
            l.color[k] = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
        }
        l.angle_deg = d.light_angle_deg[i];
        {   // the kernels compare atan2(...) > angle_deg * (pi/180) in f32; cos of that same f32 value
            const float cut = l.angle_deg * (float)(3.14159265358979323846 / 180.0);
            if (!(cut == cut)) l.cos_cut = -2.0f;         // NaN: `angle > NaN` is false for every point (light.rs:115): never outside
            else if (cut < 0.0f) l.cos_cut = 2.0f;        // angle >= 0 > cut: always outside
            else if (cut >= 3.14159265f) l.cos_cut = -2.0f;  // angle <= pi <= cut: never outside
            else l.cos_cut = (float)std::cos((double)cut);
            const double full = (double)l.angle_deg - 8.01;  // degrees
            l.cos_full = (full == full && full > 1.0 && full < 89.0) ? (float)std::cos(full * 3.14159265358979323846 / 180.0) : 2.0f;
        }
        {   // normalize(v) = v * (1 / sqrt((x x + y y) + z z)), as cgmath and the kernels evaluate it
            auto norm3 = [](const float *v, float *o) {
                const float m = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]), s = 1.0f / m;
                o[0] = v[0] * s;
                o[1] = v[1] * s;
                o[2] = v[2] * s;
            };
            const float neg[3] = {-l.dir[0], -l.dir[1], -l.dir[2]};
            norm3(neg, l.ndir1);
            norm3(l.ndir1, l.ndir2);
        }
    }
    lap("tables");
    // primitives in leaf order
    uint32_t n = d.n_objects;
    out.prims.resize((size_t)n + 3);  // 3 zero records of padding: a leaf visit touches first..first+3
    std::memset(out.prims.data() + n, 0, 3 * sizeof(Prim));
    out.prim_obj = bvh.leaf_order;
    // (one gather of the objects' rows into leaf order: independent per primitive, spread over the host threads)
    std::atomic<bool> bad{false};
    auto fill = [&](uint32_t k0, uint32_t k1, uint32_t &max_bounce) {
        for (uint32_t k = k0; k < k1; k++) {
            uint32_t o = bvh.leaf_order[k];
            const float *v = d.obj_verts + (size_t)o * 12;
            Prim &p = out.prims[k];
            std::memset(&p, 0, sizeof(p));
            int type = d.obj_type[o];
            int xf = d.obj_xform[o], mat = d.obj_material[o];
            if (type < 0 || type > 2 || xf < 0 || (uint32_t)xf >= d.n_xforms || mat < 0 || (uint32_t)mat >= d.n_materials) {
                bad = true;
                return;
            }
            if (type == RTX_OBJ_SPHERE) {
                p.v[0] = v[0];
                p.v[1] = v[1];
                p.v[2] = v[2];
                p.v[3] = v[3];
                p.v[4] = v[3] * v[3];  // radius.powf(2.0) (object.rs:154)
            } else {
                int nv = type == RTX_OBJ_QUAD ? 12 : 9;
                for (int j = 0; j < nv; j++) p.v[j] = v[j];
                V3 a = v3(v[0], v[1], v[2]), b = v3(v[3], v[4], v[5]);
                V3 e = type == RTX_OBJ_QUAD ? v3(v[9], v[10], v[11]) : v3(v[6], v[7], v[8]);
                V3 nn = normalize(cross(b - a, e - a));  // object.rs:168,191
                p.n[0] = nn.x;
                p.n[1] = nn.y;
                p.n[2] = nn.z;
            }
            p.meta = (uint32_t)type | ((uint32_t)xf << 2) | (xf_identity[xf] ? 0x8000u : 0u) | ((uint32_t)mat << 16);
            if (mat_bounce[mat] > max_bounce) max_bounce = mat_bounce[mat];
        }
    };
    {
        const uint32_t nt = n < (1u << 16) ? 1u : (uint32_t)std::max(1, std::min(threads, 64));
        std::vector<uint32_t> mb(nt, 0u);
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < nt; t++)
            pool.emplace_back(fill, (uint32_t)((uint64_t)n * t / nt), (uint32_t)((uint64_t)n * (t + 1) / nt), std::ref(mb[t]));
        fill(0, (uint32_t)((uint64_t)n / nt), mb[0]);
        for (auto &th : pool) th.join();
        for (uint32_t t = 0; t < nt; t++) out.max_bounce = std::max(out.max_bounce, mb[t]);
        if (bad) {
            err = "object with bad type / transform / material index";
            return RTX_ERR_INVALID;
        }
    }
    lap("primitives");
    std::vector<float> obj_ovh;
    out.overhang = hit_overhang(d, obj_ovh);
    lap("overhang");
    // tree
    if (!bvh.nodes.empty()) {
        // the binary view (reference traversal, digest) and the wide view write disjoint parts of `out`:
        // the binary one runs on a thread of its own beside the wide one
        Emit e{bvh, out};
        out.nodes.reserve(bvh.nodes.size() / 2 + 1);
        out.preorder_sizes.reserve(bvh.nodes.size());
        std::thread binary([&] {
            float total = 0.0f;
            out.root_ref = e.emit(0, total);
            out.total_sa = total;
        });
        struct Joiner {
            std::thread &t;
            ~Joiner() { if (t.joinable()) t.join(); }
        } joiner{binary};
        if (bvh.nodes[0].left < 0) {
            binary.join();
            out.wide_root_ref = out.root_ref;
        } else {
            // children are numbered after their parent (bvh_build.cpp), so one descending pass
            // folds the per-object overhangs up the tree
            std::vector<float> node_ovh(bvh.nodes.size(), 0.0f);
            if (out.overhang != 0.0f)
                for (size_t i = bvh.nodes.size(); i-- > 0;) {
                    const BuildNode &bn = bvh.nodes[i];
                    float m = 0.0f;
                    if (bn.left < 0)
                        for (uint32_t k = bn.first; k < bn.first + bn.count; k++) m = std::max(m, obj_ovh[bvh.leaf_order[k]]);
                    else
                        m = std::max(node_ovh[bn.left], node_ovh[bn.right]);
                    node_ovh[i] = m;
                }
            EmitWide w{bvh, out, node_ovh};
            out.wnodes.reserve(bvh.nodes.size() / 4 + 1);
            w.run(0);
            out.wide_root_ref = 0;
            binary.join();
            lap("tree views");
        }
        for (int k = 0; k < 3; k++) {
            out.root_mn[k] = bvh.nodes[0].mn[k];
            out.root_mx[k] = bvh.nodes[0].mx[k];
        }
    }
    return RTX_OK;
}

}  // namespace rtx
