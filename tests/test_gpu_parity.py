"""Parity of the HIP path against the CPU oracle, through the C ABI (needs an MI355X).

Bars (north_star): primary hit ids equal and t bit-identical (the kernels repeat the
reference's f32 operations in order, so not even epsilon-ties are excused here); images
within 1/255 per channel on >= 99.9 % of pixels (libm calls -- atan2, acos, pow -- differ
in the last ulp between glibc and the device library, which can flip a quantisation step).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES = [("basic", 96, 0), ("random_spheres", 96, 300), ("suzanne", 96, 0), ("basic_suzanne", 96, 0),
          ("scan", 96, 20000), ("scan_glass", 64, 8000), ("soup", 96, 50000)]


def _pair(rt, orc, gpu, name, width, param):
    s = rt.HostScene(name, width, param)
    gpu.upload(s)
    return s, orc.Scene(s)


def _image_close(a, b, frac=0.999):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=2)
    return (d <= 1).mean() >= frac, (d <= 1).mean(), int(d.max())


@pytest.mark.parametrize("name,width,param", SCENES)
def test_bvh_topology_matches_oracle(rt, orc, gpu, name, width, param):
    s, o = _pair(rt, orc, gpu, name, width, param)
    leaf_g, pre_g = gpu.bvh_digest()
    leaf_o, pre_o, depth_o, sa_o = o.bvh_digest()
    assert np.array_equal(leaf_g, leaf_o)
    assert np.array_equal(pre_g, pre_o)


@pytest.mark.parametrize("name,width,param", SCENES)
def test_primary_hits_bit_exact(rt, orc, gpu, name, width, param):
    s, o = _pair(rt, orc, gpu, name, width, param)
    obj_o, t_o = o.trace_primary()
    for flags in (0, rt.FLAG_REFERENCE_TRAVERSAL):
        obj_g, t_g = gpu.trace_primary(flags=flags)
        assert np.array_equal(obj_g, obj_o), f"{(obj_g != obj_o).sum()} hit ids differ (flags={flags})"
        assert np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32)), "t not bit-identical"
    # jittered sample 3 of a 4-spp render
    obj_o, t_o = o.trace_primary(sample=3, use_jitter=True, seed=7)
    obj_g, t_g = gpu.trace_primary(sample=3, use_jitter=True, seed=7)
    assert np.array_equal(obj_g, obj_o)
    assert np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))


@pytest.mark.parametrize("name,width,param", SCENES)
@pytest.mark.parametrize("spp", [1, 4])
def test_image_matches_oracle(rt, orc, gpu, name, width, param, spp):
    s, o = _pair(rt, orc, gpu, name, width, param)
    img_o, st_o = o.render(spp=spp, bounces=4, seed=1)
    img_g, st_g = gpu.render(spp=spp, bounces=4, seed=1)
    ok, frac, worst = _image_close(img_g, img_o)
    assert ok, f"only {frac:.5f} of pixels within 1/255 (worst {worst})"
    # one ray = one Bvh::get_closest_intersection; the cone pre-test uses atan2, so allow
    # a handful of boundary flips
    assert st_g.primary_rays == st_o.primary_rays
    assert abs(int(st_g.bounce_rays) - int(st_o.bounce_rays)) <= max(2, st_o.bounce_rays // 10000)
    assert abs(int(st_g.shadow_rays) - int(st_o.shadow_rays)) <= max(4, st_o.shadow_rays // 10000)


def test_visit_counters(rt, gpu):
    """RTX_FLAG_COUNT_VISITS: same image, non-zero counters; reference traversal visits more."""
    s = rt.HostScene("basic_suzanne", 64)
    gpu.upload(s)
    img0, st0 = gpu.render(spp=2, bounces=4, seed=3)
    img1, st1 = gpu.render(spp=2, bounces=4, seed=3, flags=rt.FLAG_COUNT_VISITS)
    img2, st2 = gpu.render(spp=2, bounces=4, seed=3, flags=rt.FLAG_COUNT_VISITS | rt.FLAG_REFERENCE_TRAVERSAL)
    assert np.array_equal(img0, img1) and np.array_equal(img0, img2)
    assert st0.closest_nodes == 0 and st1.closest_nodes > 0 and st1.shadow_nodes > 0 and st1.closest_prims > 0
    assert st2.closest_nodes > st1.closest_nodes and st2.shadow_nodes > st1.shadow_nodes
    assert st1.rays == st0.rays == st2.rays


@pytest.mark.parametrize("name,width,param", [("basic_suzanne", 96, 0), ("scan", 96, 20000), ("random_spheres", 64, 300)])
def test_visit_counters_match_the_oracle(rt, orc, gpu, name, width, param):
    """The counters behind the roofline's byte count: the device's verbatim traversal (RTX_FLAG_REFERENCE_TRAVERSAL |
    RTX_FLAG_COUNT_VISITS) must visit exactly the nodes and objects that the oracle's bvh.rs:359-395 restatement
    counts on the same rays.  Closest-hit queries are bit-identical rays: exact.  Shadow queries: exact whenever both
    sides ask the same number of them (a cone cut-off decided by atan2's last ulp can add or drop a query)."""
    s, o = _pair(rt, orc, gpu, name, width, param)
    _, st_o = o.render(spp=2, bounces=3, seed=4)
    _, st_g = gpu.render(spp=2, bounces=3, seed=4, flags=rt.FLAG_COUNT_VISITS | rt.FLAG_REFERENCE_TRAVERSAL)
    assert st_o.closest_nodes > 0 and st_o.closest_prims > 0
    assert (st_g.primary_rays, st_g.bounce_rays) == (st_o.primary_rays, st_o.bounce_rays)
    assert (st_g.closest_nodes, st_g.closest_prims) == (st_o.closest_nodes, st_o.closest_prims)
    if st_g.shadow_rays == st_o.shadow_rays:
        assert (st_g.shadow_nodes, st_g.shadow_prims) == (st_o.shadow_nodes, st_o.shadow_prims)
    else:
        assert abs(st_g.shadow_rays - st_o.shadow_rays) <= 1e-4 * st_o.shadow_rays
        assert abs(st_g.shadow_nodes - st_o.shadow_nodes) <= 1e-3 * st_o.shadow_nodes
        assert abs(st_g.shadow_prims - st_o.shadow_prims) <= 1e-3 * st_o.shadow_prims
    # the shipped traversal finds the same hits with fewer visits
    _, st_p = gpu.render(spp=2, bounces=3, seed=4, flags=rt.FLAG_COUNT_VISITS)
    assert st_p.rays == st_g.rays
    assert 0 < st_p.closest_prims <= st_g.closest_prims and 0 < st_p.shadow_prims <= st_g.shadow_prims


def test_bounce_depths(rt, orc, gpu):
    s, o = _pair(rt, orc, gpu, "basic", 64, 0)
    for b in (0, 1, 2, 6):
        img_o, _ = o.render(spp=1, bounces=b)
        img_g, _ = gpu.render(spp=1, bounces=b)
        ok, frac, worst = _image_close(img_g, img_o)
        assert ok, (b, frac, worst)


def test_secondary_rays_bit_exact(rt, orc, gpu):
    """Arbitrary (incoherent) rays incl. axis-parallel and origin-inside-box cases."""
    s, o = _pair(rt, orc, gpu, "basic_suzanne", 32, 0)
    rng = np.random.default_rng(3)
    n = 20000
    org = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:200, 0] = 0.0
    d[200:400, 1] = 0.0
    d[400:500, :2] = 0.0
    d[400:500, 2] = 1.0
    rays = np.concatenate([org, d], axis=1)
    obj_o, t_o = o.trace_rays(rays)
    obj_g, t_g = gpu.trace_rays(rays)
    assert np.array_equal(obj_g, obj_o)
    assert np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
    obj_r, t_r = gpu.trace_rays(rays, flags=rt.FLAG_REFERENCE_TRAVERSAL)
    assert np.array_equal(obj_r, obj_o)


def _world_vertices(s):
    """World-space vertices of the scene's triangles and quads (float64 product of the f32 data)."""
    d = s.desc.contents
    n = d.n_objects
    typ = np.ctypeslib.as_array(d.obj_type, (n,))
    verts = np.ctypeslib.as_array(d.obj_verts, (n, 12)).astype(np.float64)
    xf = np.ctypeslib.as_array(d.obj_xform, (n,))
    mats = np.ctypeslib.as_array(d.xforms, (d.n_xforms, 32)).astype(np.float64)
    out = []
    for i in np.nonzero(typ != 0)[0]:
        m = mats[xf[i], :16].reshape(4, 4).T  # column-major object_to_world
        nv = 4 if typ[i] == 2 else 3
        for k in range(nv):
            v = m @ np.append(verts[i, 3 * k:3 * k + 3], 1.0)
            out.append(v[:3] / v[3])
    return np.array(out)


@pytest.mark.parametrize("name,width,param", [("basic_suzanne", 32, 0), ("scan", 32, 20000)])
def test_grazing_rays_bit_exact(rt, orc, gpu, name, width, param):
    """Rays aimed at mesh vertices and along triangle edges, from near and from far away: they
    pass through corners and faces of leaf boxes (the slab test's error band, box_exact) and
    through edges shared by triangles (sign-of-zero edge tests, ties in t)."""
    s, o = _pair(rt, orc, gpu, name, width, param)
    vw = _world_vertices(s)
    rng = np.random.default_rng(11)
    n = 24000
    tgt = vw[rng.integers(0, len(vw), n)]
    org = rng.uniform(-6, 6, (n, 3))
    org[n // 3:2 * n // 3] *= 300.0          # far origins: large o * (1/d) in the slab planes
    k = 2 * n // 3                            # rays along an edge: through two vertices of one primitive
    a = vw[(rng.integers(0, len(vw) // 3, n - k)) * 3]
    b = vw[(rng.integers(0, len(vw) // 3, n - k)) * 3 + 1]
    org[k:] = a - (b - a) * rng.uniform(0.5, 40.0, (n - k, 1))
    tgt[k:] = b
    # nearly axis-parallel rays through vertices: one direction component ~1e-4 of the others,
    # so o * (1/d) is in the thousands and the slab t values carry a large absolute error
    m = n // 6
    ax = rng.integers(0, 3, m)
    off = rng.uniform(-5, 5, (m, 3))
    off[np.arange(m), ax] = rng.uniform(-1e-3, 1e-3, m)
    org[:m] = tgt[:m] - off
    rays = np.concatenate([org, tgt - org], axis=1).astype(np.float32)
    obj_o, t_o = o.trace_rays(rays)
    for flags in (0, rt.FLAG_REFERENCE_TRAVERSAL):
        obj_g, t_g = gpu.trace_rays(rays, flags=flags)
        assert np.array_equal(obj_g, obj_o)
        assert np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
    assert (obj_o != 0xFFFFFFFF).mean() > 0.3


def test_grazing_rays_shaded_identically_by_both_traversals(rt, gpu, monkeypatch):
    """World::trace_ray for rays aimed at mesh vertices (bounces and shadow rays start at
    vertices and edges): the pruned traversal and the verbatim one (a second context forced to
    it by RTX_REFERENCE_TRAVERSAL=1) share all shading code, so the colours must be identical
    bit for bit -- this covers the any-hit shadow queries, which no hit-id API exposes."""
    s = rt.HostScene("scan_glass", 32, 8000)
    gpu.upload(s)
    vw = _world_vertices(s)
    rng = np.random.default_rng(23)
    n = 12000
    tgt = vw[rng.integers(0, len(vw), n)]
    org = rng.uniform(-6, 6, (n, 3))
    m = n // 3  # nearly axis-parallel
    ax = rng.integers(0, 3, m)
    off = rng.uniform(-5, 5, (m, 3))
    off[np.arange(m), ax] = rng.uniform(-1e-3, 1e-3, m)
    org[:m] = tgt[:m] - off
    rays = np.concatenate([org, tgt - org], axis=1).astype(np.float32)
    c_p = gpu.shade_rays(rays, depth=4)
    monkeypatch.setenv("RTX_REFERENCE_TRAVERSAL", "1")
    ref = rt.Context(0)
    try:
        ref.upload(s)
        c_r = ref.shade_rays(rays, depth=4)
    finally:
        ref.close()
    assert np.array_equal(c_p.view(np.uint32), c_r.view(np.uint32))
    assert (c_p.max(axis=1) > 0).mean() > 0.5


def test_degenerate_rays_match_oracle(rt, orc, gpu):
    """Zero, NaN and infinite components (utils.rs refract yields NaN directions on total
    internal reflection): both traversals and the oracle agree bit for bit."""
    s, o = _pair(rt, orc, gpu, "basic_suzanne", 32, 0)
    nan, inf = np.nan, np.inf
    base = [0.3, 1.2, 4.0]
    rays = np.array([
        base + [0.0, 0.0, 0.0],            # zero direction: normalisation gives NaN
        base + [nan, 0.0, -1.0], base + [0.0, nan, -1.0], base + [nan, nan, nan],
        base + [inf, 0.0, -1.0], base + [0.0, -inf, -1.0], base + [inf, inf, inf],
        [nan, 1.0, 4.0, 0.0, 0.0, -1.0], [inf, 1.0, 4.0, 0.0, 0.0, -1.0], [0.0, -inf, 4.0, 0.0, 0.1, -1.0],
        base + [1e-30, 0.0, -1.0], base + [1e-38, 1e-38, -1.0], base + [-0.0, 0.0, -1.0],
        base + [1e38, 1e38, -1e38], base + [1e-45, -1.0, 1e-45],   # overflow in the norm / denormals
        [1e30, 1.0, 4.0, -1.0, 0.0, 0.0], [0.0, 1e20, 0.0, 0.0, -1.0, 0.0],
        base + [0.0, -1.0, 0.0], base + [0.0, 0.0, -1.0], base + [1.0, 0.0, 0.0],
    ], np.float32)
    obj_o, t_o = o.trace_rays(rays)
    for flags in (0, rt.FLAG_REFERENCE_TRAVERSAL):
        obj_g, t_g = gpu.trace_rays(rays, flags=flags)
        assert np.array_equal(obj_g, obj_o), (flags, obj_g, obj_o)
        assert np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32)), (flags, t_g, t_o)
    assert (obj_o != 0xFFFFFFFF).any() and (obj_o == 0xFFFFFFFF).any()


@pytest.mark.parametrize("name,width,param", [("basic_suzanne", 32, 0), ("scan", 32, 20000)])
def test_direction_scales_bit_exact(rt, orc, gpu, name, width, param):
    """Directions of every magnitude from 1e-30 to 1e30 aimed at vertices.  Around 1e-23 the
    squared length is a denormal, Ray::new's normalisation returns a direction that is not
    unit length, and the slab parameter stops being a distance: such rays must not prune."""
    s, o = _pair(rt, orc, gpu, name, width, param)
    vw = _world_vertices(s)
    rng = np.random.default_rng(23)
    n = 60000
    tgt = vw[rng.integers(0, len(vw), n)]
    org = rng.uniform(-12, 12, (n, 3))
    scale = 10.0 ** rng.uniform(-30, 30, (n, 1))
    scale[: n // 2] = 10.0 ** rng.uniform(-24.5, -18.5, (n // 2, 1))     # the denormal band of the squared length
    rays = np.concatenate([org, (tgt - org) * scale], axis=1).astype(np.float32)
    obj_o, t_o = o.trace_rays(rays)
    for flags in (0, rt.FLAG_REFERENCE_TRAVERSAL):
        obj_g, t_g = gpu.trace_rays(rays, flags=flags)
        assert np.array_equal(obj_g, obj_o), f"flags {flags}: {(obj_g != obj_o).sum()} hit ids differ"
        assert np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
    assert (obj_o != 0xFFFFFFFF).mean() > 0.3


def test_hits_outside_their_boxes_bit_exact(rt, orc, gpu):
    """Objects that the reference can hit OUTSIDE the box it bounds them with: non-planar quads
    (hit region = the quad seen along (b-a)x(d-a), reaching the projection of c) and spheres
    under rotation + non-uniform scale (box from column norms).  Pruning by box entry t has to
    allow for that overhang; found by a sweep: a ray aimed at a Suzanne vertex returned the
    second-nearest polygon."""
    rng = np.random.default_rng(5)
    objs = []
    g = 14
    grid = np.stack(np.meshgrid(np.arange(g + 1), np.arange(g + 1), indexing="ij"), -1).astype(np.float64)
    h = rng.uniform(-0.45, 0.45, (g + 1, g + 1))           # strongly non-planar quads of a height field
    P = np.concatenate([(grid - g / 2) * 0.5, h[..., None]], -1)[..., [0, 2, 1]]     # y up
    for layer in range(3):                                  # stacked sheets: competing hits close in t
        for i in range(g):
            for j in range(g):
                a, b, c, d = P[i, j], P[i, j + 1], P[i + 1, j + 1], P[i + 1, j]
                off = np.array([0, 0.35 * layer, 0])
                objs.append(("quad", a + off, d + off, c + off, b + off, 0, layer % 2))
    c30, s30 = np.cos(0.6), np.sin(0.6)
    rot = np.array([[c30, -s30, 0, 0], [s30, c30, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    shear = rot @ np.diag([3.0, 0.4, 1.0, 1.0]) @ rot.T @ np.array([[1, 0, 0, 0.2], [0, 1, 0, 0.1], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    for k in range(40):
        objs.append(("sphere", rng.uniform(-1.0, 1.0, 3), rng.uniform(0.05, 0.3), 0, 1))
    xforms = [(np.eye(4), np.eye(4)), (shear, np.linalg.inv(shear))]
    s = rt.ArrayScene(objs, width=48, height=48, xforms=xforms)
    gpu.upload(s)
    o = orc.Scene(s)
    n = 120000
    tgt = rng.uniform(-3.5, 3.5, (n, 3)) * np.array([1, 0.4, 1])
    org = rng.uniform(-9, 9, (n, 3))
    org[: n // 2, 1] = np.abs(org[: n // 2, 1]) + 1.0      # from above: quads face up or down, both are covered
    rays = np.concatenate([org, tgt - org], axis=1).astype(np.float32)
    obj_o, t_o = o.trace_rays(rays)
    for flags in (0, rt.FLAG_REFERENCE_TRAVERSAL):
        obj_g, t_g = gpu.trace_rays(rays, flags=flags)
        assert np.array_equal(obj_g, obj_o), f"flags {flags}: {(obj_g != obj_o).sum()} hit ids differ"
        assert np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
    hit = obj_o != 0xFFFFFFFF
    assert hit.mean() > 0.2 and (obj_o[hit] >= 3 * g * g).any() and (obj_o[hit] < 3 * g * g).any()
    img_o, _ = o.render(spp=1, bounces=2, seed=1)
    img_g, _ = gpu.render(spp=1, bounces=2, seed=1)
    assert _image_close(img_g, img_o)[0]


def test_tangent_rays_on_spheres_bit_exact(rt, orc, gpu):
    """Rays tangent to spheres (the d^2 <= r^2 boundary of object.rs:150-166) and through the
    faces of their bounding boxes, near and far, incl. nearly axis-parallel ones."""
    s, o = _pair(rt, orc, gpu, "random_spheres", 32, 300)
    d = s.desc.contents
    n_obj = d.n_objects
    typ = np.ctypeslib.as_array(d.obj_type, (n_obj,))
    verts = np.ctypeslib.as_array(d.obj_verts, (n_obj, 12)).astype(np.float64)
    xf = np.ctypeslib.as_array(d.obj_xform, (n_obj,))
    mats = np.ctypeslib.as_array(d.xforms, (d.n_xforms, 32)).astype(np.float64)
    cen, rad = [], []
    for i in np.nonzero(typ == 0)[0]:
        m = mats[xf[i], :16].reshape(4, 4).T
        c = m @ np.append(verts[i, :3], 1.0)
        cen.append(c[:3] / c[3])
        rad.append(verts[i, 3] * np.linalg.norm(m[:3, 0]))
    cen, rad = np.array(cen), np.array(rad)
    rng = np.random.default_rng(31)
    n = 24000
    k = rng.integers(0, len(cen), n)
    org = rng.uniform(-8, 8, (n, 3))
    org[:, 1] = np.abs(org[:, 1]) + 0.2
    org[n // 2:] *= 50.0
    m = n // 6  # nearly axis-parallel tangents
    ax = rng.integers(0, 3, m)
    off = rng.uniform(-6, 6, (m, 3))
    off[np.arange(m), ax] = rng.uniform(-1e-3, 1e-3, m)
    org[:m] = cen[k[:m]] - off
    to_c = cen[k] - org
    perp = np.cross(to_c, rng.normal(size=(n, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    # tangent point as seen from org: offset by r * (1 +- tiny) perpendicular to the line of sight
    scale = 1.0 + rng.choice([-1e-6, -1e-7, 0.0, 1e-7, 1e-6], n)
    tgt = cen[k] + perp * (rad[k] * scale)[:, None]
    rays = np.concatenate([org, tgt - org], axis=1).astype(np.float32)
    obj_o, t_o = o.trace_rays(rays)
    for flags in (0, rt.FLAG_REFERENCE_TRAVERSAL):
        obj_g, t_g = gpu.trace_rays(rays, flags=flags)
        assert np.array_equal(obj_g, obj_o)
        assert np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
    assert (obj_o != 0xFFFFFFFF).mean() > 0.3


def test_reference_bvh_known_answers_on_device(rt, gpu):
    """bvh.rs:494-540 (test_bvh_intersect) on the device."""
    from test_oracle_kat import BVH_OBJECTS, BVH_RAYS
    gpu.upload(rt.ArrayScene(BVH_OBJECTS))
    rays = np.array([list(o) + list(d) for o, d, _ in BVH_RAYS], np.float32)
    obj, t = gpu.trace_rays(rays)
    for (o, d, want), got in zip(BVH_RAYS, obj):
        assert (got != 0xFFFFFFFF) == want, (o, d)


def test_shade_rays_matches_oracle(rt, orc, gpu):
    s, o = _pair(rt, orc, gpu, "basic", 32, 0)
    rng = np.random.default_rng(5)
    n = 4000
    org = np.tile(np.array([0, 1.5, 5], np.float32), (n, 1))
    d = np.stack([rng.uniform(-0.6, 0.6, n), rng.uniform(-0.8, 0.1, n), -np.ones(n)], axis=1).astype(np.float32)
    rays = np.concatenate([org, d], axis=1)
    c_o = o.shade_rays(rays, depth=4)
    c_g = gpu.shade_rays(rays, depth=4)
    close = np.abs(c_o - c_g).max(axis=1) <= 1.0 / 255.0
    assert close.mean() >= 0.999


def test_tile_partition_covers_image(rt, orc, gpu):
    """Two ranks' interleaved tiles assemble into the single-rank image (non-multiple size)."""
    s = rt.HostScene("basic", 100)
    gpu.upload(s)
    full, _ = gpu.render(spp=1, bounces=3)
    a, _ = gpu.render(spp=1, bounces=3, part=(32, 0, 2))
    b, _ = gpu.render(spp=1, bounces=3, part=(32, 1, 2))
    assert np.array_equal(np.maximum(a, b), full)
    assert not (a.astype(bool) & b.astype(bool)).any()


def test_sample_batching_is_exact(rt, gpu, monkeypatch):
    """Small node budget forces sample / pixel chunking; the image must not change."""
    s = rt.HostScene("basic", 64)
    gpu.upload(s)
    ref, _ = gpu.render(spp=5, bounces=4, seed=9)
    monkeypatch.setenv("RTX_NODE_BUDGET_MB", "1")
    # the batches alternate between two lanes (streams, wavefront buffers) by default; one lane: all on the slot's stream
    for lanes in ("2", "1"):
        monkeypatch.setenv("RTX_BATCH_LANES", lanes)
        c2 = rt.Context(0)
        try:
            c2.upload(s)
            img, st = c2.render(spp=5, bounces=4, seed=9)
            assert st.batches > 1
            assert np.array_equal(img, ref), f"lanes {lanes}"
            # a second frame right behind the first, asynchronously, then a synchronous one in the same slot
            buf = rt.DeviceBuffer(c2.partition_pixels(None) * 3)
            c2.render_device(buf.ptr, 5, 4, seed=9, stats=False)
            st2 = c2.render_device(buf.ptr, 5, 4, seed=9)
            assert st2.batches == st.batches and st2.rays == st.rays
            assert np.array_equal(rt.untile(buf.to_host(), 64, 64, 32), ref), f"lanes {lanes}, device tiles"
        finally:
            c2.close()


def test_batched_glass_frame_two_lanes(rt, monkeypatch):
    """A frame of many batches with bounce levels (pool nodes, resolve launches per level) is byte-identical
    whether its batches run on one lane or alternate between two."""
    s = rt.HostScene("scan_glass", 256, 30000)
    frames = []
    monkeypatch.setenv("RTX_MIN_BATCHES", "5")
    for lanes in ("1", "2"):
        monkeypatch.setenv("RTX_BATCH_LANES", lanes)
        c = rt.Context(0)
        try:
            c.upload(s)
            buf = rt.DeviceBuffer(c.partition_pixels(None) * 3)
            st = c.render_device(buf.ptr, 16, 4, seed=2)
            assert st.batches >= 4
            frames.append((buf.to_host(), st.rays))
        finally:
            c.close()
    assert frames[0][0].any() and frames[0][1] == frames[1][1]
    assert np.array_equal(frames[0][0], frames[1][0])


def test_slots_overlap_and_agree(rt, gpu, monkeypatch):
    """Frames in flight in two render slots at once (different batchings, too) must not differ in a byte
    from one frame rendered alone."""
    s = rt.HostScene("scan_glass", 512, 30000)
    monkeypatch.setenv("RTX_MIN_BATCHES", "3")
    c = rt.Context(0)
    try:
        c.upload(s)
        assert c.slots >= 2
        n = c.partition_pixels(None) * 3
        ref = rt.DeviceBuffer(n)
        st0 = c.render_device(ref.ptr, 8, 4, seed=5)               # alone, synchronous
        assert st0.batches >= 3
        want = ref.to_host()
        assert want.any()
        for nslots in (2, c.slots):                                     # frames queued without waiting, rotating over the slots
            bufs = [rt.DeviceBuffer(n, fill=k + 1) for k in range(2 * nslots)]
            for k, b in enumerate(bufs):
                c.render_device(b.ptr, 8, 4, seed=5, stats=False, slot=k % nslots)
            c.synchronize()
            for k, b in enumerate(bufs):
                got = b.to_host()
                assert np.array_equal(got, want), f"frame {k} (slot {k % nslots} of {nslots}): {(got != want).sum()} bytes differ"
        st1 = c.render_device(bufs[0].ptr, 8, 4, seed=5, slot=1)   # counters of a slot-1 render
        assert st1.rays == st0.rays and st1.batches == st0.batches
    finally:
        c.close()
    monkeypatch.setenv("RTX_MIN_BATCHES", "1")
    c = rt.Context(0)
    try:
        c.upload(s)
        one = rt.DeviceBuffer(n)
        st = c.render_device(one.ptr, 8, 4, seed=5)
        assert st.batches == 1 and st.rays == st0.rays
        assert np.array_equal(one.to_host(), want)
    finally:
        c.close()


@pytest.mark.parametrize("name,width,param,spp", [("soup", 256, 200000, 4), ("scan_glass", 256, 30000, 4), ("basic_suzanne", 192, 0, 2),
                                                  ("random_spheres", 128, 300, 2)])
def test_leaf_banking_variant_is_exact(rt, orc, monkeypatch, name, width, param, spp):
    """The path kernel variant that sets leaves aside during node steps (chosen by scene size; forced here with
    RTX_LEAF_BANK) visits the tree in another order: frames, ray counts and primary hits must not differ in a bit
    from the default variant's, and the frame must match the oracle like the default's."""
    s = rt.HostScene(name, width, param)
    frames, stats = [], []
    for bank in ("0", "1"):
        monkeypatch.setenv("RTX_LEAF_BANK", bank)
        c = rt.Context(0)
        try:
            c.upload(s)
            buf = rt.DeviceBuffer(c.partition_pixels(None) * 3)
            st = c.render_device(buf.ptr, spp, 4, seed=3)
            frames.append(buf.to_host())
            stats.append(st)
        finally:
            c.close()
    assert frames[0].any()
    assert np.array_equal(frames[0], frames[1]), f"{(frames[0] != frames[1]).sum()} bytes differ"
    assert (stats[0].primary_rays, stats[0].bounce_rays, stats[0].shadow_rays) == (stats[1].primary_rays, stats[1].bounce_rays, stats[1].shadow_rays)
    o = orc.Scene(s)
    img_o, st_o = o.render(spp=spp, bounces=4, seed=3)
    img_g = rt.untile(frames[1], width, width, 32)
    ok, frac, worst = _image_close(img_g, img_o)
    assert ok, f"only {frac:.5f} of pixels within 1/255 (worst {worst})"


@pytest.mark.parametrize("name,width,param,spp,part", [("scan_glass", 256, 30000, 4, None), ("basic_suzanne", 200, 0, 3, (32, 1, 3)),
                                                       ("random_spheres", 128, 300, 1, None), ("basic", 768, 0, 2, None)])
def test_primaries_first_launch_is_exact(rt, monkeypatch, name, width, param, spp, part):
    """RTX_PRIMARIES_FIRST=1: the primaries' closest hits in a launch of their own, the hits listed with the bounce
    materials first (count / scan / scatter), the rest of every path tree from that list in a second persistent
    launch (kernels.h, launch_path).  Another order of the same work: tile bytes and ray counts must not differ in
    a bit from the single launch's -- padded tiles (width 200), a partition, two batches and a scene whose hits lie
    outside their boxes (path_ovh_kernel) included."""
    s = rt.HostScene(name, width, param)
    frames, stats = [], []
    for pf in ("0", "1"):
        monkeypatch.setenv("RTX_PRIMARIES_FIRST", pf)
        if name == "basic": monkeypatch.setenv("RTX_MIN_BATCHES", "2")
        c = rt.Context(0)
        try:
            c.upload(s)
            buf = rt.DeviceBuffer(c.partition_pixels(part) * 3)
            st = c.render_device(buf.ptr, spp, 4, seed=5, part=part)
            frames.append(buf.to_host())
            stats.append(st)
        finally:
            c.close()
    assert frames[0].any()
    assert np.array_equal(frames[0], frames[1]), f"{(frames[0] != frames[1]).sum()} bytes differ"
    assert (stats[0].primary_rays, stats[0].bounce_rays, stats[0].shadow_rays) == (stats[1].primary_rays, stats[1].bounce_rays, stats[1].shadow_rays)
    if name == "basic": assert stats[1].batches >= 2


@pytest.mark.parametrize("name,width,param,spp", [("basic_suzanne", 160, 0, 2), ("scan", 128, 30000, 2), ("random_spheres", 128, 300, 2),
                                                  ("basic", 96, 0, 1)])
def test_transform_cache_variants_are_exact(rt, monkeypatch, name, width, param, spp):
    """The path kernels come with and without a cache of the ray in the last transform's object space (kernels.hip,
    RayCtx; the host picks by whether any object has a transform).  Forced either way (RTX_XFORM_CACHE), scenes with
    and without transforms must give the same bytes, ray counts and primary hits."""
    s = rt.HostScene(name, width, param)
    out = []
    for xc in ("0", "1"):
        monkeypatch.setenv("RTX_XFORM_CACHE", xc)
        c = rt.Context(0)
        try:
            c.upload(s)
            img, st = c.render(spp=spp, bounces=4, seed=2)
            obj, t = c.trace_primary()
            out.append((img, st, obj, t))
        finally:
            c.close()
    (i0, s0, o0, t0), (i1, s1, o1, t1) = out
    assert i0.any() and np.array_equal(i0, i1)
    assert (s0.primary_rays, s0.bounce_rays, s0.shadow_rays) == (s1.primary_rays, s1.bounce_rays, s1.shadow_rays)
    assert np.array_equal(o0, o1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))


def test_errors(rt, gpu):
    c = rt.Context(0)
    with pytest.raises(rt.RtxError) as e:
        c.width = c.height = 4
        c.render()
    assert e.value.status == rt.RTX_ERR_NO_SCENE
    c.close()
    # more than 4 objects with identical centroids: the reference recurses forever (bvh.rs:221-230)
    objs = [("sphere", (0, 0, 0), 1.0 + 0.1 * i) for i in range(6)]
    with pytest.raises(rt.RtxError) as e:
        gpu.upload(rt.ArrayScene(objs))
    assert e.value.status == rt.RTX_ERR_DEGENERATE
    # empty scene renders the background
    gpu.upload(rt.ArrayScene([], width=16, height=16, background=(0.2, 0.2, 0.2)))
    img, st = gpu.render()
    assert (img == 51).all() and st.rays == 256


def test_cli_renders_the_same_png(rt, gpu, tmp_path):
    """rtx-render with the reference's flags (main.rs:26-77) writes the PNG that the library
    renders for the same scene, size, samples and bounces (spp 1: no jitter involved)."""
    import subprocess
    exe = os.path.join(os.path.dirname(rt.__file__), "rtx-render")
    out = str(tmp_path / "cli.png")
    subprocess.check_call([exe, "-f", out, "-w", "96", "-s", "1", "-b", "3", "-t", "2", "--scene", "basic_suzanne",
                           "--media", rt.MEDIA_DIR])
    img_cli = rt.png_read(out)
    s = rt.HostScene("basic_suzanne", 96)
    gpu.upload(s)
    img, _ = gpu.render(spp=1, bounces=3)
    assert img_cli.shape == img.shape and np.array_equal(img_cli, img)
    assert subprocess.call([exe, "--no-such-flag"], stderr=subprocess.DEVNULL) == 2


def test_cli_non_square_scene_file(rt, gpu, tmp_path):
    """A scene file with its own `size W H`: the CLI's PNG has the camera's size, not -w squared
    (rtx_render_builtin_to_png once sized its buffer width x width: heap overflow / scrambled rows)."""
    import subprocess
    src = open(os.path.join(rt.MEDIA_DIR, "scenes", "basic.scene")).read()
    scene = tmp_path / "wide.scene"
    scene.write_text(src + "\nsize 120 56\n")   # (its texture is named media:texture.png: found through --media)
    exe = os.path.join(os.path.dirname(rt.__file__), "rtx-render")
    out = str(tmp_path / "wide.png")
    subprocess.check_call([exe, "-f", out, "-w", "40", "-s", "1", "-b", "2", "--scene", "file:" + str(scene),
                           "--media", rt.MEDIA_DIR])
    img_cli = rt.png_read(out)
    s = rt.HostScene("file:" + str(scene), 40)
    assert (s.width, s.height) == (120, 56)
    gpu.upload(s)
    img, _ = gpu.render(spp=1, bounces=2)
    assert img.shape == (56, 120, 3)
    assert img_cli.shape == img.shape and np.array_equal(img_cli, img)


def test_odd_persistent_grids_trace_every_ray(rt, gpu, monkeypatch):
    """Grids whose wave count is not a multiple of the 16 sub-queues (37 or 3 blocks): every
    statically dealt queue chunk must still have an owner -- same bytes, same ray counts."""
    s = rt.HostScene("basic_suzanne", 160)
    gpu.upload(s)
    ref, st0 = gpu.render(spp=3, bounces=4, seed=11)
    obj0, t0 = gpu.trace_primary()
    for blocks in ("37", "3", "1"):
        monkeypatch.setenv("RTX_PERSISTENT_BLOCKS", blocks)
        c = rt.Context(0)
        try:
            c.upload(s)
            img, st = c.render(spp=3, bounces=4, seed=11)
            obj, t = c.trace_primary()
        finally:
            c.close()
        assert st.rays == st0.rays, f"{blocks} blocks: {st.rays} rays instead of {st0.rays}"
        assert np.array_equal(img, ref), f"{blocks} blocks: {(img != ref).sum()} bytes differ"
        assert np.array_equal(obj, obj0) and np.array_equal(t.view(np.uint32), t0.view(np.uint32))


def test_ply_mesh_scene_matches_oracle(rt, orc, gpu, tmp_path):
    """A binary PLY (the Stanford scans' format, README.md:7) through mesh:<path>: a displaced
    UV sphere of triangles with quad caps rows, traced and shaded like any other scene."""
    import struct
    nu, nv = 96, 48
    rng = np.random.default_rng(11)
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0.15, np.pi - 0.15, nv)
    r = 1.0 + 0.08 * rng.standard_normal((nv, nu))
    pos = np.stack([r * np.sin(v)[:, None] * np.cos(u), r * np.cos(v)[:, None] + 0 * u, r * np.sin(v)[:, None] * np.sin(u)], -1)
    pos = pos.reshape(-1, 3).astype(np.float32) * np.float32(17.0) + np.float32(40.0)   # the loader rescales and centres
    faces = []
    for j in range(nv - 1):
        for i in range(nu):
            a, b, c, d = j * nu + i, j * nu + (i + 1) % nu, (j + 1) * nu + (i + 1) % nu, (j + 1) * nu + i
            faces += [[a, b, c, d]] if j % 8 == 0 else [[a, b, c], [a, c, d]]
    path = tmp_path / "blob.ply"
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {len(pos)}\nproperty float x\nproperty float y\n"
                 f"property float z\nelement face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        f.write(pos.tobytes())
        for fc in faces:
            f.write(struct.pack("<B" + "i" * len(fc), len(fc), *fc))
    for material in (0, 1):
        s, o = _pair(rt, orc, gpu, f"mesh:{path}", 96, material)
        assert s.n_objects == 5 + len(faces)
        obj_o, t_o = o.trace_primary()
        obj_g, t_g = gpu.trace_primary()
        assert np.array_equal(obj_g, obj_o) and np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
        assert (obj_o >= 5).mean() > 0.05          # the mesh is in view
        img_o, _ = o.render(spp=2, bounces=4, seed=1)
        img_g, _ = gpu.render(spp=2, bounces=4, seed=1)
        ok, frac, worst = _image_close(img_g, img_o)
        assert ok, f"only {frac:.5f} of pixels within 1/255 (worst {worst})"


def test_scene_file_renders_like_builtin(rt, gpu):
    """assets/scenes/basic.scene describes load_basic; both ways of building the scene render
    the same image, and hit ids / t agree bit for bit."""
    a = rt.HostScene("basic", 96)
    b = rt.HostScene("file:" + os.path.join(rt.MEDIA_DIR, "scenes", "basic.scene"), 96)
    out = []
    for s in (a, b):
        gpu.upload(s)
        obj, t = gpu.trace_primary()
        img, _ = gpu.render(spp=2, bounces=4, seed=3)
        out.append((obj, t, img))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    assert np.array_equal(out[0][2], out[1][2])


def test_adversarial_lights_scene_matches_oracle(rt, orc, gpu):
    """assets/scenes/adversarial_lights.scene: a directional light exactly along -y and one along
    +x (shadow rays with two zero direction components), a point light ON a mesh vertex, a cone
    light lying in the floor plane; mirror + glass + Phong mix on two Suzannes and a sphere."""
    s, o = _pair(rt, orc, gpu, "file:" + os.path.join(rt.MEDIA_DIR, "scenes", "adversarial_lights.scene"), 80, 0)
    obj_o, t_o = o.trace_primary()
    obj_g, t_g = gpu.trace_primary()
    assert np.array_equal(obj_g, obj_o) and np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
    for spp in (1, 3):
        img_o, st_o = o.render(spp=spp, bounces=4, seed=2)
        img_g, st_g = gpu.render(spp=spp, bounces=4, seed=2)
        ok, frac, worst = _image_close(img_g, img_o)
        assert ok, f"only {frac:.5f} of pixels within 1/255 (worst {worst})"
        assert abs(int(st_g.shadow_rays) - int(st_o.shadow_rays)) <= max(4, st_o.shadow_rays // 10000)


def _lights_scene(path, n_lights):
    """A glass / mirror / Phong scene under `n_lights` lights of all four kinds in turn (ambient, point, directional,
    cone), the last one directional: the light mask of a path node has 16 bits, the lights sit in the kernel
    arguments (DeviceScene::lights_k[16]) and in a table for per-lane look-ups."""
    lines = ["shader s phong 0.7 0.3 4", "shader r reflective", "shader g refractive 1.3", "shader k mix 0.4 r 0.4 g 0.2 s",
             "texture t flat 0.8 0.7 0.3", "texture c image media:texture.png", "material m k t", "material f s c",
             "quad f -5 -1 5   5 -1 5   5 -1 -5   -5 -1 -5", "mesh f media:Suzanne.obj", "sphere m -2.2 0 -0.5 1"]
    for i in range(n_lights):
        a = 2.0 * np.pi * i / max(n_lights, 1)
        x, z = 3.0 * np.cos(a), 3.0 * np.sin(a)
        c = "%.3f %.3f %.3f" % (0.15 + 0.05 * (i % 3), 0.2, 0.1 + 0.05 * (i % 4))
        kind = 2 if i == n_lights - 1 else i % 4
        if kind == 0:
            lines.append("light ambient 0.02 0.02 0.02")
        elif kind == 1:
            lines.append("light point %.4f 2.5 %.4f   %s" % (x, z, c))
        elif kind == 2:
            lines.append("light directional %.4f -1 %.4f   %s" % (-0.3 * np.cos(a), -0.3 * np.sin(a), c))
        else:
            lines.append("light cone %.4f 3 %.4f   %.4f -3 %.4f   35   %s" % (x, z, -x, -z, c))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def test_sixteen_lights_none_and_too_many(rt, orc, gpu, tmp_path):
    """The limits of the light table: 16 lights (every bit of a node's light mask, every entry of the copy in the
    kernel arguments) against the oracle; no light at all; a 17th is refused at upload."""
    p16 = str(tmp_path / "l16.scene")
    _lights_scene(p16, 16)
    s, o = _pair(rt, orc, gpu, "file:" + p16, 72, 0)
    assert s.desc.contents.n_lights == 16
    obj_o, t_o = o.trace_primary()
    obj_g, t_g = gpu.trace_primary()
    assert np.array_equal(obj_g, obj_o) and np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
    for spp in (1, 2):
        img_o, st_o = o.render(spp=spp, bounces=3, seed=5)
        img_g, st_g = gpu.render(spp=spp, bounces=3, seed=5)
        ok, frac, worst = _image_close(img_g, img_o)
        assert ok, f"only {frac:.5f} of pixels within 1/255 (worst {worst})"
        assert st_g.shadow_rays > 4 * st_g.primary_rays // 2   # (twelve lights ask for shadow queries)
        assert abs(int(st_g.shadow_rays) - int(st_o.shadow_rays)) <= max(4, st_o.shadow_rays // 10000)
    img_ref, _ = gpu.render(spp=2, bounces=3, seed=5, flags=rt.FLAG_REFERENCE_TRAVERSAL)
    img_g, _ = gpu.render(spp=2, bounces=3, seed=5)
    assert np.array_equal(img_g, img_ref)
    p0 = str(tmp_path / "l0.scene")
    _lights_scene(p0, 0)
    s, o = _pair(rt, orc, gpu, "file:" + p0, 48, 0)
    img_o, st_o = o.render(spp=1, bounces=3, seed=5)
    img_g, st_g = gpu.render(spp=1, bounces=3, seed=5)
    assert st_g.shadow_rays == 0 and st_o.shadow_rays == 0 and _image_close(img_g, img_o)[0]
    p17 = str(tmp_path / "l17.scene")
    _lights_scene(p17, 17)
    with pytest.raises(rt.RtxError) as e:
        gpu.upload(rt.HostScene("file:" + p17, 48, 0))
    assert e.value.status == rt.RTX_ERR_CAPACITY


def test_bounce_materials_between_plain_ones(rt, orc, gpu, tmp_path):
    """The shading pass asks for a hit's material record only when its id lies in the range of the materials that
    spawn bounce rays (DeviceScene::bounce_mat_lo / hi); here plain Phong materials sit below, between and above the
    mirror and the glass mix, so the range has holes and objects on both sides of it."""
    lines = ["shader p phong 0.7 0.3 4", "shader q phong 0.5 0.5 16", "shader r reflective", "shader g refractive 1.3",
             "shader km mix 0.5 r 0.5 p", "shader kg mix 0.6 g 0.4 q",
             "texture t1 flat 0.8 0.3 0.3", "texture t2 flat 0.3 0.8 0.3", "texture t3 flat 0.3 0.3 0.8", "texture c image media:texture.png",
             "material a p c", "material b km t1", "material c q t2", "material d kg t3", "material e p t1",
             "quad a -5 -1 5   5 -1 5   5 -1 -5   -5 -1 -5",
             "sphere b -2.4 0 -1 1", "sphere c -0.8 -0.3 0.6 0.7", "sphere d 0.9 0 -0.6 1", "sphere e 2.6 -0.4 0.4 0.6",
             "light point 2 3 0.5   1 1 1", "light cone -1.5 2 0   1 -1 0   50   1 0.4 0.4", "light ambient 0.2 0.2 0.2"]
    path = str(tmp_path / "holes.scene")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    s, o = _pair(rt, orc, gpu, "file:" + path, 96, 0)
    obj_o, t_o = o.trace_primary()
    obj_g, t_g = gpu.trace_primary()
    assert np.array_equal(obj_g, obj_o) and np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
    for bounces in (1, 4):
        img_o, st_o = o.render(spp=2, bounces=bounces, seed=7)
        img_g, st_g = gpu.render(spp=2, bounces=bounces, seed=7)
        ok, frac, worst = _image_close(img_g, img_o)
        assert ok, f"only {frac:.5f} of pixels within 1/255 (worst {worst})"
        assert st_g.bounce_rays == st_o.bounce_rays and (bounces == 1) == (st_g.bounce_rays == 0)
        assert abs(int(st_g.shadow_rays) - int(st_o.shadow_rays)) <= max(4, st_o.shadow_rays // 10000)
