"""Random scenes through the scene-file front end, rendered by the HIP path and by the oracle:
nested material compositions, every texture and light kind, transformed spheres / triangles /
quads / meshes, random cameras.  Same bars as test_gpu_parity.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shader_tree(rng, L, name, depth, budget):
    """Appends shader statements for a random material tree; returns the bounce leaves used.
    flatten.cpp limits: nesting <= 4 compositions, <= 4 reflective/refractive leaves."""
    kinds = ["phong", "phong", "reflective", "refractive", "none"] + (["mix", "mix"] if depth < 3 else [])
    kind = kinds[rng.integers(0, len(kinds))]
    if kind in ("reflective", "refractive") and budget[0] <= 0:
        kind = "phong"
    if kind == "phong":
        L.append(f"shader {name} phong {rng.uniform(0, 1):.3f} {rng.uniform(0, 1):.3f} {rng.uniform(0.5, 20):.3f}")
    elif kind == "reflective":
        budget[0] -= 1
        L.append(f"shader {name} reflective")
    elif kind == "refractive":
        budget[0] -= 1
        L.append(f"shader {name} refractive {rng.uniform(1.0, 1.8):.3f}")
    elif kind == "none":
        L.append(f"shader {name} none")
    else:
        parts = []
        for k in range(int(rng.integers(1, 4))):
            child = f"{name}_{k}"
            _shader_tree(rng, L, child, depth + 1, budget)
            parts.append(f"{rng.uniform(0.1, 0.9):.3f} {child}")
        L.append(f"shader {name} mix " + " ".join(parts))


def _ops(rng, strong):
    ops = []
    for _ in range(int(rng.integers(0, 4))):
        k = rng.integers(0, 5)
        if k == 0:
            ops.append("translate " + " ".join(f"{x:.3f}" for x in rng.uniform(-1.5, 1.5, 3)))
        elif k == 1:
            sc = rng.uniform(0.5, 1.8, 3) if strong else np.repeat(rng.uniform(0.5, 1.8), 3)
            ops.append("scale " + " ".join(f"{x:.3f}" for x in sc))
        else:
            ops.append(f"rotate_{'xyz'[k - 2]} {rng.uniform(-180, 180):.2f}")
    return " ".join(ops)


def _scene_text(seed):
    rng = np.random.default_rng(seed)
    L = [f"camera {rng.uniform(-3, 3):.3f} {rng.uniform(0.5, 4):.3f} {rng.uniform(4, 7):.3f}  "
         f"{rng.uniform(-1, 1):.3f} {rng.uniform(-0.5, 0.5):.3f} {rng.uniform(-1, 1):.3f}  0 1 0",
         f"size {int(rng.integers(40, 72))} {int(rng.integers(40, 72))}",
         "background " + " ".join(f"{x:.3f}" for x in rng.uniform(0, 0.5, 3))]
    n_mat = 6
    for i in range(n_mat):
        _shader_tree(rng, L, f"sh{i}", 0, [4])
        tk = rng.integers(0, 4)
        if tk == 0:
            L.append(f"texture tx{i} image media:texture.png")
        elif tk == 1:
            L.append(f"texture tx{i} none")
        else:
            L.append(f"texture tx{i} flat " + " ".join(f"{x:.3f}" for x in rng.uniform(0, 1, 3)))
        L.append(f"material m{i} sh{i} tx{i}")
    mat = lambda: f"m{int(rng.integers(0, n_mat))}"
    L.append(f"quad {mat()} -6 -1 6  6 -1 6  6 -1 -6  -6 -1 -6")
    for _ in range(int(rng.integers(6, 14))):
        c = rng.uniform(-2.5, 2.5, 3)
        L.append(f"sphere {mat()} {c[0]:.3f} {abs(c[1]) * 0.6:.3f} {c[2]:.3f} {rng.uniform(0.2, 0.9):.3f} {_ops(rng, True)}")
    for _ in range(int(rng.integers(4, 10))):
        p = rng.uniform(-2.5, 2.5, (3, 3))
        L.append(f"triangle {mat()} " + " ".join(f"{x:.3f}" for x in p.reshape(-1)) + " " + _ops(rng, True))
    for _ in range(int(rng.integers(3, 8))):
        o = rng.uniform(-2.5, 2.5, 3)
        u, v = rng.uniform(-1.5, 1.5, 3), rng.uniform(-1.5, 1.5, 3)
        warp = rng.uniform(-0.3, 0.3, 3) * (rng.random() < 0.5)      # half of the quads are not planar parallelograms
        q = np.array([o, o + u, o + u + v + warp, o + v])
        L.append(f"quad {mat()} " + " ".join(f"{x:.3f}" for x in q.reshape(-1)) + " " + _ops(rng, True))
    if rng.random() < 0.7:
        L.append(f"mesh {mat()} media:Suzanne.obj translate {rng.uniform(-2, 2):.3f} {rng.uniform(0, 1):.3f} {rng.uniform(-2, 1):.3f} {_ops(rng, False)}")
    col = lambda: " ".join(f"{x:.3f}" for x in rng.uniform(0.2, 1, 3))
    for _ in range(int(rng.integers(1, 4))):
        L.append("light point " + " ".join(f"{x:.3f}" for x in rng.uniform(-4, 4, 3) * [1, 0, 1] + [0, rng.uniform(1, 6), 0]) + " " + col())
    for _ in range(int(rng.integers(0, 3))):
        p = rng.uniform(-3, 3, 3) * [1, 0, 1] + [0, rng.uniform(1.5, 5), 0]
        L.append("light cone " + " ".join(f"{x:.3f}" for x in p) + " " + " ".join(f"{x:.3f}" for x in -p + rng.uniform(-1, 1, 3)) +
                 f" {rng.uniform(10, 80):.2f} " + col())
    if rng.random() < 0.6:
        L.append("light directional " + " ".join(f"{x:.3f}" for x in rng.uniform(-1, 1, 3) * [1, 0, 1] + [0, -1, 0]) + " " + col())
    if rng.random() < 0.8:
        L.append("light ambient " + " ".join(f"{x:.3f}" for x in rng.uniform(0, 0.3, 3)))
    return "\n".join(L) + "\n"


@pytest.mark.parametrize("seed", range(int(os.environ.get("RTX_FUZZ_SEEDS", "24"))))   # RTX_FUZZ_SEEDS=500 for a long hunt
def test_random_scene_matches_oracle(rt, orc, gpu, tmp_path, seed):
    path = tmp_path / f"fuzz{seed}.scene"
    path.write_text(_scene_text(seed))
    s = rt.HostScene(f"file:{path}", 64)
    gpu.upload(s)
    o = orc.Scene(s)
    obj_o, t_o = o.trace_primary()
    for flags in (0, rt.FLAG_REFERENCE_TRAVERSAL):
        obj_g, t_g = gpu.trace_primary(flags=flags)
        assert np.array_equal(obj_g, obj_o), f"seed {seed} flags {flags}: {(obj_g != obj_o).sum()} hit ids differ"
        assert np.array_equal(t_g.view(np.uint32), t_o.view(np.uint32))
    for spp, bounces in ((1, 4), (3, 3)):
        img_o, st_o = o.render(spp=spp, bounces=bounces, seed=seed)
        img_g, st_g = gpu.render(spp=spp, bounces=bounces, seed=seed)
        d = np.abs(img_g.astype(np.int32) - img_o.astype(np.int32)).max(axis=2)
        frac = (d <= 1).mean()
        assert frac >= 0.999, f"seed {seed} spp {spp}: only {frac:.5f} of pixels within 1/255 (worst {int(d.max())})"
        assert st_g.primary_rays == st_o.primary_rays
        assert abs(int(st_g.bounce_rays) - int(st_o.bounce_rays)) <= max(2, st_o.bounce_rays // 5000)
        assert abs(int(st_g.shadow_rays) - int(st_o.shadow_rays)) <= max(4, st_o.shadow_rays // 5000)
