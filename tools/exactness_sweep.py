"""Sweep at scale: production traversal == on-device reference traversal (hit id and t bit for
bit) over adversarial ray families on a scene.  Usage:
    python tools/exactness_sweep.py SCENE PARAM SEED REPS [RAYS_PER_REP]
Families (equal shares of every repetition):
  0 toward vertices from near and far origins          1 along primitive edges
  2 nearly axis-parallel through vertices (component ratio 1e-3 .. 1e-8)
  3 exactly axis-parallel through vertices (zero direction components)
  4 origins ON primitives (random barycentric point), random / vertex-aimed directions
  5 origins on primitives offset by 1e-4 along random directions (secondary-ray like)
  6 rays inside the plane of a primitive, through one of its vertices
  7 tiny and huge direction scales of family 0 (normalisation, 1/d range)
Spheres contribute their six axis-extreme points as "vertices"."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
rt = importlib.import_module("rust-raytracer_amd")

def geometry(s):
    """World-space vertex data of a HostScene for the ray families."""
    d = s.desc.contents
    N = d.n_objects
    typ = np.ctypeslib.as_array(d.obj_type, (N,))
    verts = np.ctypeslib.as_array(d.obj_verts, (N, 12)).astype(np.float64)
    xf = np.ctypeslib.as_array(d.obj_xform, (N,))
    mats = np.ctypeslib.as_array(d.xforms, (d.n_xforms, 32)).astype(np.float64)[:, :16].reshape(-1, 4, 4).transpose(0, 2, 1)


    def to_world(p, idx):
        m = mats[xf[idx]]
        w = np.einsum("nij,nj->ni", m[:, :3, :3], p) + m[:, :3, 3]
        return w


    poly = np.nonzero(typ != 0)[0]
    sph = np.nonzero(typ == 0)[0]
    tri_a = to_world(verts[poly, 0:3], poly)
    tri_b = to_world(verts[poly, 3:6], poly)
    tri_c = to_world(verts[poly, 6:9], poly)
    vlist = [tri_a, tri_b, tri_c]
    quads = poly[typ[poly] == 2]
    if len(quads):
        vlist.append(to_world(verts[quads, 9:12], quads))
    if len(sph):
        c, r = verts[sph, 0:3], verts[sph, 3:4]
        for k in range(3):
            for sg in (-1.0, 1.0):
                e = np.zeros(3)
                e[k] = sg
                vlist.append(to_world(c + r * e, sph))
    vw = np.concatenate(vlist)
    lo, hi = vw.min(0), vw.max(0)
    ext = float((hi - lo).max())
    return dict(N=N, vw=vw, lo=lo, hi=hi, ext=ext, poly=poly, tri_a=tri_a, tri_b=tri_b, tri_c=tri_c)


def make_rays(rng, geo, n):
    """One repetition: n/8 rays of each family; returns (rays float32 [8m, 6], family ids, m)."""
    vw, lo, hi, ext, poly = geo['vw'], geo['lo'], geo['hi'], geo['ext'], geo['poly']
    tri_a, tri_b, tri_c = geo['tri_a'], geo['tri_b'], geo['tri_c']
    m = n // 8
    fam = np.repeat(np.arange(8), m)
    org = np.zeros((8 * m, 3))
    dirv = np.zeros((8 * m, 3))
    pick = lambda k: vw[rng.integers(0, len(vw), k)]
    # 0: toward vertices
    t0 = pick(m)
    o0 = rng.uniform(lo - 0.5 * ext, hi + 0.5 * ext, (m, 3))
    o0[m // 2:] = (o0[m // 2:] - t0[m // 2:]) * 10.0 ** rng.uniform(1, 4, (m - m // 2, 1)) + t0[m // 2:]
    org[0:m], dirv[0:m] = o0, t0 - o0
    # 1: along edges
    if len(poly):
        i = rng.integers(0, len(poly), m)
        a, b = tri_a[i], tri_b[i]
        use_c = rng.random(m) < 0.5
        b = np.where(use_c[:, None], tri_c[i], b)
        o1 = a - (b - a) * 10.0 ** rng.uniform(-0.3, 3.6, (m, 1))
        org[m:2 * m], dirv[m:2 * m] = o1, b - o1
    else:
        org[m:2 * m], dirv[m:2 * m] = o0, t0 - o0
    # 2: nearly axis-parallel
    t2 = pick(m)
    off = rng.uniform(-0.5 * ext, 0.5 * ext, (m, 3))
    ax = rng.integers(0, 3, m)
    off[np.arange(m), ax] *= 10.0 ** rng.uniform(-8, -3, m)
    two = rng.random(m) < 0.3   # two small components
    ax2 = (ax + 1 + rng.integers(0, 2, m)) % 3
    off[np.arange(m)[two], ax2[two]] *= 10.0 ** rng.uniform(-8, -3, int(two.sum()))
    org[2 * m:3 * m], dirv[2 * m:3 * m] = t2 - off, off
    # 3: exactly axis-parallel
    t3 = pick(m).astype(np.float32).astype(np.float64)
    off = rng.uniform(-0.5 * ext, 0.5 * ext, (m, 3))
    ax = rng.integers(0, 3, m)
    off[np.arange(m), ax] = 0.0
    two = rng.random(m) < 0.5
    off[np.arange(m)[two], ax2[two]] = 0.0
    off[(off == 0).all(1), 0] = 1.0
    org[3 * m:4 * m], dirv[3 * m:4 * m] = t3 - off, off
    # 4 / 5: origins on primitives
    for f, eps in ((4, 0.0), (5, 1e-4)):
        if len(poly):
            i = rng.integers(0, len(poly), m)
            u, v = rng.random((m, 1)), rng.random((m, 1))
            fl = (u + v) > 1
            u, v = np.where(fl, 1 - u, u), np.where(fl, 1 - v, v)
            p = tri_a[i] + u * (tri_b[i] - tri_a[i]) + v * (tri_c[i] - tri_a[i])
        else:
            p = pick(m)
        dd = rng.standard_normal((m, 3))
        aim = rng.random(m) < 0.4
        dd[aim] = pick(int(aim.sum())) - p[aim]
        dd[(dd == 0).all(1)] = 1.0
        dn = dd / np.linalg.norm(dd, axis=1, keepdims=True)
        org[f * m:(f + 1) * m], dirv[f * m:(f + 1) * m] = p + eps * dn, dd
    # 6: in-plane rays
    if len(poly):
        i = rng.integers(0, len(poly), m)
        e1, e2 = tri_b[i] - tri_a[i], tri_c[i] - tri_a[i]
        w1, w2 = rng.standard_normal((m, 1)), rng.standard_normal((m, 1))
        dd = w1 * e1 + w2 * e2
        dd[(dd == 0).all(1)] = 1.0
        o6 = tri_a[i] - dd * 10.0 ** rng.uniform(-1, 3, (m, 1))
        org[6 * m:7 * m], dirv[6 * m:7 * m] = o6, dd
    else:
        org[6 * m:7 * m], dirv[6 * m:7 * m] = o0, t0 - o0
    # 7: direction scales
    t7 = pick(m)
    o7 = rng.uniform(lo - ext, hi + ext, (m, 3))
    org[7 * m:], dirv[7 * m:] = o7, (t7 - o7) * 10.0 ** rng.uniform(-30, 30, (m, 1))
    rays = np.concatenate([org, dirv], axis=1).astype(np.float32)
    return rays, fam, m


def main():
    scene, param, seed, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 20
    gpu = rt.Context(0)
    s = rt.HostScene(scene, 64, param)
    gpu.upload(s)
    # shading check: World::trace_ray colours (bounces and any-hit shadow queries included) from a
    # second context that runs the verbatim traversal everywhere; same shading code, so bit-equal
    SHADE = os.environ.get("SWEEP_SHADE", "1") != "0"
    if SHADE:
        os.environ["RTX_REFERENCE_TRAVERSAL"] = "1"
        gref = rt.Context(0)
        del os.environ["RTX_REFERENCE_TRAVERSAL"]
        gref.upload(s)
    geo = geometry(s)
    print(f"{scene} param {param}: {geo['N']} objects, {len(geo['vw'])} vertices, extent {geo['ext']:.3g}", flush=True)
    rng = np.random.default_rng(seed)
    bad = 0
    total = 0
    for rep in range(reps):
        rays, fam, m = make_rays(rng, geo, n)
        obj_g, t_g = gpu.trace_rays(rays)
        obj_r, t_r = gpu.trace_rays(rays, flags=rt.FLAG_REFERENCE_TRAVERSAL)
        diff = (obj_g != obj_r) | (t_g.view(np.uint32) != t_r.view(np.uint32))
        if SHADE:
            k = len(rays) // 4            # a quarter of every family (the verbatim traversal shades slowly)
            sub = rays.reshape(8, m, 6)[:, : m // 4].reshape(-1, 6)
            c_p = gpu.shade_rays(sub, depth=3)
            c_r = gref.shade_rays(sub, depth=3)
            cd = (c_p.view(np.uint32) != c_r.view(np.uint32)).any(axis=1)
            sdiff = np.zeros(len(rays), bool)
            sdiff.reshape(8, m)[:, : m // 4] = cd.reshape(8, m // 4)
            diff = diff | sdiff
            shade_note = f" shaded {len(sub)} colour mismatches {int(cd.sum())}"
        else:
            shade_note = ""
        bad += int(diff.sum())
        total += len(rays)
        hits = obj_r != 0xFFFFFFFF
        per = " ".join(f"{k}:{int(diff[fam == k].sum())}/{hits[fam == k].mean():.2f}" for k in range(8))
        print(f"rep {rep}: {len(rays)} rays,{shade_note} mismatches {int(diff.sum())}   family:mismatches/hit-fraction {per}", flush=True)
        if diff.any():
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"sweep_mismatch_{scene.replace('/', '_').replace(':', '_')}.txt"), "a") as f:
                for i in np.nonzero(diff)[0][:32]:
                    f.write(f"{int(fam[i])} " + " ".join(repr(float(x)) for x in rays[i]) +
                            f"  prod {int(obj_g[i])} {float(t_g[i])!r} ref {int(obj_r[i])} {float(t_r[i])!r}\n")
    print(f"TOTAL {total} rays, MISMATCHES {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
