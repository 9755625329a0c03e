"""Host-side logic of the backend, no GPU needed: the C-ABI library loads and exports what
include/*.h declares, the BVH builder reproduces the oracle's (= the reference's) topology,
the on-disk formats decode, and the oracle still matches its committed golden vectors."""
import ctypes
import hashlib
import json
import math
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "golden.json")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_library_exports_every_declared_symbol(rt):
    lib = rt.lib()
    names = []
    for h in ("rtx.h", "rtx_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", text)
    names = sorted(set(names))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"


def test_scene_desc_layout_matches_oracle(rt):
    """rtx_scene_desc and orc_scene_desc are declared separately; field order must agree."""
    def fields(path, struct):
        text = open(path).read()
        body = text[text.index("typedef struct " + struct):]
        body = body[:body.index("} " + struct)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split("{", 1)[1].split(";"):
            decl = decl.strip()
            if decl:
                out.append(re.sub(r"\s+", " ", decl))
        return out
    a = fields(os.path.join(ROOT, "include", "rtx.h"), "rtx_scene_desc")
    b = fields(os.path.join(ROOT, "oracle", "oracle.h"), "orc_scene_desc")
    assert a == b


def test_no_gpu_means_loud_failure(rt):
    """Without a usable HIP device rtx_init must fail (no CPU fallback in the product)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtxError) as e:
        rt.Context(0)
    assert e.value.status == rt.RTX_ERR_HIP


@pytest.mark.parametrize("name,param", [("basic", 0), ("random_spheres", 1000), ("suzanne", 0), ("basic_suzanne", 0),
                                        ("scan", 60000), ("scan_glass", 5000), ("soup", 30000)])
def test_bvh_builder_topology_equals_oracle(rt, orc, name, param):
    s = rt.HostScene(name, 32, param)
    leaf_h, pre_h, depth_h, sa_h, n_int = rt.host_bvh_digest(s, threads=4)
    leaf_o, pre_o, depth_o, sa_o = orc.Scene(s).bvh_digest()
    assert np.array_equal(leaf_h, leaf_o)
    assert np.array_equal(pre_h, pre_o)
    assert depth_h == depth_o and sa_h == sa_o
    assert n_int == (len(pre_o) - 1) // 2


def test_bvh_builder_is_thread_count_invariant(rt):
    s = rt.HostScene("scan", 32, 150000)
    ref = rt.host_bvh_digest(s, threads=1)
    for t in (2, 8):
        got = rt.host_bvh_digest(s, threads=t)
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]) and ref[2:] == got[2:]


@pytest.mark.parametrize("name,param", [("soup", 60000), ("scan", 30000), ("random_spheres", 5000), ("scan_glass", 700)])
def test_bvh_top_phase_through_the_splitter_interface(rt, monkeypatch, name, param):
    """build_bvh with a TopSplitter (bvh_build.h): the nodes at the top are split breadth first from the splitter's
    per-node ranges, bucket counts / boxes and stable partitions, the pieces by the host threads.  With the host
    implementation of the interface (RTX_TOP_SPLITTER=host) the tree must be the plain recursion's, bit for bit --
    the same driver runs the GPU implementation in rtx_upload_scene (tests/test_gpu_build.py)."""
    s = rt.HostScene(name, 32, param)
    ref = rt.host_bvh_digest(s, threads=4)
    monkeypatch.setenv("RTX_TOP_SPLITTER", "host")
    for t in (1, 5):
        got = rt.host_bvh_digest(s, threads=t)
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]) and ref[2:] == got[2:]


def test_bvh_top_phase_reports_degenerate_input(rt, monkeypatch):
    """600 objects with one centroid cannot be separated at the root: the top phase must say so like the recursion."""
    monkeypatch.setenv("RTX_TOP_SPLITTER", "host")
    objs = [("sphere", (1, 2, 3), 0.5 + 0.001 * i) for i in range(600)]
    with pytest.raises(rt.RtxError) as e:
        rt.host_bvh_digest(rt.ArrayScene(objs))
    assert e.value.status == rt.RTX_ERR_DEGENERATE
    # ... and further down, in a piece that the host threads build: a cluster of 40 inside 2000 separable objects
    objs = [("sphere", (float(i % 50), float(i // 50), 0.0), 0.3) for i in range(2000)] + [("sphere", (7.25, 3.25, 1.0), 0.1 + 0.001 * i) for i in range(40)]
    with pytest.raises(rt.RtxError) as e:
        rt.host_bvh_digest(rt.ArrayScene(objs))
    assert e.value.status == rt.RTX_ERR_DEGENERATE


def test_bvh_digests_match_golden(rt):
    for key, (name, param) in {"bvh_suzanne_0": ("suzanne", 0), "bvh_soup_10000": ("soup", 10000),
                               "bvh_scan_100000": ("scan", 100000)}.items():
        leaf, pre, depth, sa, _ = rt.host_bvh_digest(rt.HostScene(name, 32, param))
        g = META[key]
        assert _sha(leaf) == g["leaf_order_sha256"] and _sha(pre) == g["preorder_sizes_sha256"]
        assert depth == g["bvh_depth"] and len(pre) == g["bvh_nodes"] and sa == np.float32(g["total_sa"])


def test_degenerate_build_is_reported(rt):
    objs = [("sphere", (1, 2, 3), 0.5 + 0.1 * i) for i in range(7)]   # identical centroids
    with pytest.raises(rt.RtxError) as e:
        rt.host_bvh_digest(rt.ArrayScene(objs))
    assert e.value.status == rt.RTX_ERR_DEGENERATE
    # <= 4 such objects are a single leaf, as in the reference (bvh.rs:336)
    leaf, pre, depth, sa, n_int = rt.host_bvh_digest(rt.ArrayScene(objs[:4]))
    assert list(pre) == [4] and depth == 0 and n_int == 0


def test_bad_object_indices_are_rejected(rt):
    """A descriptor whose objects point outside the transform / material tables or carry an
    unknown type is refused on the host, before anything is indexed with it."""
    objs = [("sphere", (float(i), 0, 0), 0.4) for i in range(6)]
    for field, bad in (("obj_xform", 7), ("obj_xform", -1), ("obj_type", 3), ("obj_type", -2), ("obj_material", 9)):
        sc = rt.ArrayScene(objs)
        getattr(sc, field)[3] = bad
        with pytest.raises(rt.RtxError) as e:
            rt.host_bvh_digest(sc)
        assert e.value.status == rt.RTX_ERR_INVALID, field


def test_hit_overhang(rt, tmp_path):
    """Only non-planar quads and sheared spheres can be hit outside their boxes.  The headline
    scene must report exactly 0: a non-zero value selects the kernels with the per-slot allowance."""
    for name, param in (("basic", 0), ("random_spheres", 300), ("scan", 30000), ("scan_glass", 5000), ("soup", 20000)):
        assert rt.hit_overhang(rt.HostScene(name, 16, param)) == 0.0, name
    assert rt.hit_overhang(rt.HostScene("scan", 16, 0)) == 0.0
    assert 0.085 < rt.hit_overhang(rt.HostScene("suzanne", 16)) < 0.095          # some of its quads are far from planar
    flat = ("quad", (0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1))
    assert rt.hit_overhang(rt.ArrayScene([flat])) == 0.0
    # c lifted by 1 above the plane of a, b, d and hanging outside the vertices' box in x: the hit
    # region reaches its projection (1.5, 0, 1)
    warped = ("quad", (0, 0, 0), (1, 0, 0), (1.5, 1, 1), (0, 0, 1))
    assert abs(rt.hit_overhang(rt.ArrayScene([warped])) - 0.0) < 1e-6             # projection (1.5,0,1) is inside x <= 1.5
    warped = ("quad", (0, 0, 0), (1, 0, 0), (1, 1, 1), (0, 0.5, 1))               # tilted plane: projection of c leaves the box
    assert rt.hit_overhang(rt.ArrayScene([warped])) > 0.05
    c, s_ = np.cos(0.5), np.sin(0.5)
    rot = np.array([[c, -s_, 0, 0], [s_, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    shear = rot @ np.diag([3.0, 0.5, 1.0, 1.0])
    sph = [("sphere", (0, 0, 0), 1.0, 0, 1)]
    ov = rt.hit_overhang(rt.ArrayScene(sph, xforms=[(np.eye(4), np.eye(4)), (shear, np.linalg.inv(shear))]))
    rows, cols = np.linalg.norm(shear[:3, :3], axis=1), np.linalg.norm(shear[:3, :3], axis=0)
    assert abs(ov - max(0.0, (rows - cols).max())) < 1e-3 and ov > 0.1
    assert rt.hit_overhang(rt.ArrayScene(sph, xforms=[(np.eye(4), np.eye(4)), (rot, rot.T)])) == 0.0
    persp = np.eye(4); persp[3, 2] = 0.1
    assert rt.hit_overhang(rt.ArrayScene(sph, xforms=[(np.eye(4), np.eye(4)), (persp, np.linalg.inv(persp))])) == float("inf")


def test_empty_and_tiny_scenes(rt, orc):
    leaf, pre, depth, sa, n_int = rt.host_bvh_digest(rt.ArrayScene([]))
    assert len(leaf) == 0 and len(pre) == 0
    s = rt.ArrayScene([("triangle", (0, 0, 1), (0, 1, 1), (1, 0, 1))])
    leaf, pre, depth, sa, n_int = rt.host_bvh_digest(s)
    assert list(leaf) == [0] and list(pre) == [1]


def test_oracle_matches_golden_vectors(rt, orc):
    for key, g in META.items():
        if "_spp" not in key:
            continue
        name, width, param, spp = re.match(r"(.+)_(\d+)_(\d+)_spp(\d+)", key).groups()
        width, param, spp = int(width), int(param), int(spp)
        gold = np.load(os.path.join(GOLD, key + ".npz"))
        o = orc.Scene(rt.HostScene(name, width, param))
        img, st = o.render(spp=spp, bounces=4, seed=1, threads=4)
        obj, t = o.trace_primary(sample=spp - 1, use_jitter=spp > 1, seed=1)
        assert np.array_equal(obj, gold["obj"])
        assert np.array_equal(t.view(np.uint32), gold["t"].view(np.uint32))
        d = np.abs(img.astype(int) - gold["img"].astype(int)).max(axis=2)
        assert (d <= 1).mean() >= 0.999        # libm (acos/atan2/pow) may differ between glibc builds
        assert st.primary_rays == g["primary"]
        assert abs(st.shadow_rays - g["shadow"]) <= 4 and abs(st.bounce_rays - g["bounce"]) <= 2
        leaf, pre, depth, sa = o.bvh_digest()
        assert _sha(leaf) == g["leaf_order_sha256"] and depth == g["bvh_depth"]


def test_jitter_hash_golden(orc):
    for k, v in META["jitter_u32"].items():
        seed, px, py, s, d = (int(x) for x in k.split(","))
        assert orc.jitter_u32(seed, px, py, s, d) == v


def test_png_decoder_matches_pil_fixture(rt):
    tex = rt.png_read(os.path.join(ROOT, "assets", "texture.png"))
    gold = np.load(os.path.join(GOLD, "texture_rgb.npz"))["rgb"]
    assert tex.shape == (256, 256, 3) and np.array_equal(tex, gold)


def test_png_roundtrip(rt, tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    rt.png_write(p, img)
    assert np.array_equal(rt.png_read(p), img)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(p)), img)
    with pytest.raises(rt.RtxError):
        rt.png_read(str(tmp_path / "missing.png"))


def test_suzanne_obj_ingestion(rt):
    """object.rs:30-91: 500 polygons, 468 quads + 32 triangles, one shared transform and material."""
    s = rt.HostScene("suzanne", 16)
    d = s.desc.contents
    types = np.ctypeslib.as_array(d.obj_type, shape=(d.n_objects,))
    assert d.n_objects == 500 and (types == 2).sum() == 468 and (types == 1).sum() == 32
    assert d.n_xforms == 1 and d.n_materials == 1 and d.n_lights == 0
    m = np.ctypeslib.as_array(d.xforms, shape=(32,)).reshape(2, 4, 4)
    np.testing.assert_allclose(m[0].T @ m[1].T, np.eye(4), atol=1e-6)   # world_to_object is the inverse


def _mesh_fixture():
    """A small closed mesh with triangles and quads (a cube with two faces split)."""
    rng = np.random.default_rng(3)
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * np.float32(0.37)
    v += rng.uniform(-0.01, 0.01, v.shape).astype(np.float32)
    faces = [[0, 1, 3, 2], [4, 6, 7, 5], [0, 4, 5, 1], [2, 3, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]]
    return v, faces


def _write_ply(path, v, faces, fmt, extra=False):
    """ascii / binary_little_endian / binary_big_endian; `extra` adds properties and an element to skip."""
    import struct
    e = {"binary_little_endian": "<", "binary_big_endian": ">"}.get(fmt)
    head = ["ply", f"format {fmt} 1.0", "comment written by the test", f"element vertex {len(v)}"]
    if extra:
        head += ["property uchar red", "property float x", "property double y", "property float z", "property short q"]
    else:
        head += ["property float x", "property float y", "property float z"]
    if extra:
        head += ["element edge 2", "property int a", "property int b"]
    head += [f"element face {len(faces)}"]
    if extra:
        head += ["property uchar flags"]
    head += ["property list uchar int vertex_indices" if not extra else "property list ushort uint vertex_index", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        if e is None:
            for p in v:
                xyz = [repr(float(c)) for c in p]
                f.write((" ".join((["200"] if extra else []) + xyz + (["-3"] if extra else [])) + "\n").encode())
            if extra:
                f.write(b"0 1\n1 2\n")
            for fc in faces:
                f.write((" ".join((["7"] if extra else []) + [str(len(fc))] + [str(i) for i in fc]) + "\n").encode())
        else:
            for p in v:
                if extra:
                    f.write(struct.pack(e + "Bfdfh", 200, p[0], float(p[1]), p[2], -3))
                else:
                    f.write(struct.pack(e + "fff", *p))
            if extra:
                f.write(struct.pack(e + "iiii", 0, 1, 1, 2))
            for fc in faces:
                if extra:
                    f.write(struct.pack(e + "BH" + "I" * len(fc), 7, len(fc), *fc))
                else:
                    f.write(struct.pack(e + "B" + "i" * len(fc), len(fc), *fc))


def _scene_arrays(scene):
    d = scene.desc.contents   # `scene` stays referenced while its arrays are copied
    n = d.n_objects
    return (np.ctypeslib.as_array(d.obj_type, shape=(n,)).copy(), np.ctypeslib.as_array(d.obj_verts, shape=(n * 12,)).copy(),
            np.ctypeslib.as_array(d.xforms, shape=(d.n_xforms * 32,)).copy())


def test_ply_and_obj_ingestion_agree(rt, tmp_path):
    """The same mesh as OBJ, ascii PLY and both binary PLY byte orders gives identical scene arrays
    (README.md:7: the reference's headline scene is made of the Stanford .ply scans)."""
    v, faces = _mesh_fixture()
    obj = tmp_path / "m.obj"
    with open(obj, "w") as f:
        for p in v:
            f.write("v " + " ".join(repr(float(c)) for c in p) + "\n")
        for fc in faces:
            f.write("f " + " ".join(str(i + 1) for i in fc) + "\n")
    want = _scene_arrays(rt.HostScene(f"mesh:{obj}", 16))
    assert (want[0] == 2).sum() == 4 + 1 and (want[0] == 1).sum() == 4 + 1 and (want[0] == 0).sum() == 3   # + load_basic
    for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
        for extra in (False, True):
            ply = tmp_path / f"m_{fmt}_{int(extra)}.PLY"
            _write_ply(ply, v, faces, fmt, extra)
            got = _scene_arrays(rt.HostScene(f"mesh:{ply}", 16))
            for a, b in zip(want, got):
                assert a.shape == b.shape and a.tobytes() == b.tobytes(), (fmt, extra)
    # set on the floor y = -1 and scaled to a largest extent of 1.6: the world-space box of the mesh objects
    s = rt.HostScene(f"mesh:{obj}", 16, 1)
    d = s.desc.contents
    assert d.n_materials == 6 and d.n_lights == 5
    xf = np.ctypeslib.as_array(d.obj_xform, shape=(d.n_objects,))[-8:]
    assert (xf == xf[0]).all()
    m = np.ctypeslib.as_array(d.xforms, shape=(d.n_xforms, 2, 4, 4))[xf[0], 0].T.astype(np.float64)
    ov = np.ctypeslib.as_array(d.obj_verts, shape=(d.n_objects, 4, 3))[-8:]
    ty = np.ctypeslib.as_array(d.obj_type, shape=(d.n_objects,))[-8:]
    pts = np.concatenate([ov[i, :(4 if ty[i] == 2 else 3)] for i in range(8)]).astype(np.float64)
    w = pts @ m[:3, :3].T + m[:3, 3]
    lo, hi = w.min(0), w.max(0)
    assert abs(lo[1] + 1.0) < 1e-5 and abs((hi - lo).max() - 1.6) < 1e-5
    assert abs(lo[0] + hi[0]) < 1e-5 and abs(0.5 * (lo[2] + hi[2]) - 0.3) < 1e-5


def test_scan_scene_uses_stanford_files_when_present(rt, tmp_path):
    """real scans lying in the media directory replace their stand-ins
    (same place and size); here four small PLY files under the scans' names."""
    import shutil
    shutil.copy(os.path.join(rt.MEDIA_DIR, "texture.png"), tmp_path / "texture.png")
    v, faces = _mesh_fixture()
    for i, name in enumerate(rt.SCAN_FILES):
        _write_ply(tmp_path / name, v * np.float32(1 + i), faces, ("ascii", "binary_little_endian", "binary_big_endian", "ascii")[i])
    s = rt.HostScene("scan", 16, 0, media_dir=str(tmp_path))
    d = s.desc.contents
    assert d.n_objects == 5 + 4 * len(faces) and d.n_materials == 5 + 4
    ov = np.ctypeslib.as_array(d.obj_verts, shape=(d.n_objects, 4, 3))
    xf = np.ctypeslib.as_array(d.obj_xform, shape=(d.n_objects,))
    xforms = np.ctypeslib.as_array(d.xforms, shape=(d.n_xforms, 2, 4, 4))
    want = [((-0.9, 0.3), 0.55), ((0.0, 1.6), 0.3), ((2.5, 0.2), 0.45), ((-0.5, -1.3), 0.65)]   # stand-ins' centres / radii
    for k, ((cx, cz), radius) in enumerate(want):
        sl = slice(5 + k * len(faces), 5 + (k + 1) * len(faces))
        m = xforms[xf[sl][0], 0].T.astype(np.float64)
        pts = ov[sl].reshape(-1, 3).astype(np.float64)
        pts = pts[np.abs(pts).sum(1) > 0]                      # the unused 4th vertex of triangles
        w = pts @ m[:3, :3].T + m[:3, 3]
        lo, hi = w.min(0), w.max(0)
        assert abs(lo[1] + 1.0) < 1e-5 and abs((hi - lo).max() - 2 * radius) < 1e-5
        assert abs(0.5 * (lo[0] + hi[0]) - cx) < 1e-5 and abs(0.5 * (lo[2] + hi[2]) - cz) < 1e-5
    # a non-default triangle count always means the synthetic meshes
    assert rt.HostScene("scan", 16, 2000, media_dir=str(tmp_path)).n_objects == 2005


def _desc_arrays(scene):
    """Every array of an rtx_scene_desc as numpy copies, keyed by field name."""
    d = scene.desc.contents
    n = {"obj_type": d.n_objects, "obj_verts": d.n_objects * 12, "obj_xform": d.n_objects, "obj_material": d.n_objects,
         "xforms": d.n_xforms * 32, "mat_root": d.n_materials, "mat_tex_kind": d.n_materials, "mat_tex_id": d.n_materials,
         "mat_tex_color": d.n_materials * 3, "node_type": d.n_mat_nodes, "node_params": d.n_mat_nodes * 3,
         "node_child_first": d.n_mat_nodes, "node_child_count": d.n_mat_nodes, "child_node": d.n_mat_children,
         "child_coef": d.n_mat_children, "tex_w": d.n_textures, "tex_h": d.n_textures, "light_type": d.n_lights,
         "light_pos": d.n_lights * 3, "light_dir": d.n_lights * 3, "light_angle_deg": d.n_lights, "light_color": d.n_lights * 3}
    out = {k: np.ctypeslib.as_array(getattr(d, k), shape=(v,)).copy() for k, v in n.items() if v}
    out["camera_to_world"] = np.array(list(d.camera_to_world), np.float32)
    out["background"] = np.array(list(d.background), np.float32)
    out["size"] = np.array([d.width, d.height])
    return out


def test_scene_file_reproduces_load_basic(rt):
    """assets/scenes/basic.scene states scene.rs:11-108 + main.rs:108-111 in the scene-file
    vocabulary; it must give the arrays of the built-in scene (+0 / -0 aside in the one
    rotation-about-a-point matrix, which the built-in forms with an explicit inverse)."""
    a = _desc_arrays(rt.HostScene("basic", 48))
    b = _desc_arrays(rt.HostScene("file:" + os.path.join(rt.MEDIA_DIR, "scenes", "basic.scene"), 48))
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k


def test_scene_file_statements_and_errors(rt, tmp_path):
    v, faces = _mesh_fixture()
    _write_ply(tmp_path / "m.ply", v, faces, "binary_little_endian")
    ok = tmp_path / "ok.scene"
    ok.write_text("""
        size 40 24          # not square
        shader s phong 0.5 0.5 2
        texture t none
        material m s t
        mesh m m.ply  translate 0 1 0 scale 2 2 2 rotate_y 30
        sphere m 0 0 0 1
        light directional 0 -1 0  1 1 1
    """)
    s = rt.HostScene(f"file:{ok}", 64)
    d = s.desc.contents
    assert (d.width, d.height, d.n_objects, d.n_lights, d.n_materials, d.n_xforms) == (40, 24, len(faces) + 1, 1, 1, 2)
    m = np.ctypeslib.as_array(d.xforms, shape=(d.n_xforms, 2, 4, 4))
    np.testing.assert_allclose(m[0, 0].T @ m[0, 1].T, np.eye(4), atol=1e-6)
    c, sn = np.cos(np.radians(30)), np.sin(np.radians(30))
    want = np.array([[2 * c, 0, 2 * sn, 0], [0, 2, 0, 1], [-2 * sn, 0, 2 * c, 0], [0, 0, 0, 1]])
    np.testing.assert_allclose(m[0, 0].T, want, atol=1e-6)
    bad = {"frobnicate 1 2 3": "unknown statement", "sphere nope 0 0 0 1": "unknown material",
           "shader a phong 1 x 1": "not a number", "light point 0 0 0 1 1": "missing", "texture t image missing.png": "",
           "shader a mix 0.5 b": "unknown shader", "background 0 0 0 0": "unexpected",
           "shader s none\ntexture t none\nmaterial m s t\nsphere m 0 0 0 1 spin 3": "unknown transform"}
    for i, (text, msg) in enumerate(bad.items()):
        p = tmp_path / f"bad{i}.scene"
        p.write_text("# header\n" + text + "\n")
        with pytest.raises(rt.RtxError) as e:
            rt.HostScene(f"file:{p}", 16)
        nline = 2 + text.count("\n")
        assert f"bad{i}.scene:{nline}:" in str(e.value) and msg in str(e.value), str(e.value)
    with pytest.raises(rt.RtxError):
        rt.HostScene(f"file:{tmp_path / 'absent.scene'}", 16)


def test_ply_errors_are_reported(rt, tmp_path):
    v, faces = _mesh_fixture()
    good = tmp_path / "g.ply"
    _write_ply(good, v, faces, "binary_little_endian")
    raw = good.read_bytes()
    cases = {
        "truncated.ply": raw[:-5],
        "nomagic.ply": b"plx" + raw[3:],
        "noheader.ply": raw[:40],
        "badtype.ply": raw.replace(b"property float x", b"property half x"),
        "nox.ply": raw.replace(b"property float x", b"property float u"),
        "index.ply": raw[:-4] + (99).to_bytes(4, "little"),
        "pentagon.ply": None,
    }
    for name, data in cases.items():
        p = tmp_path / name
        if data is None:
            _write_ply(p, v, [[0, 1, 3, 2, 4]], "ascii")
        else:
            p.write_bytes(data)
        with pytest.raises(rt.RtxError):
            rt.HostScene(f"mesh:{p}", 16)
    with pytest.raises(rt.RtxError):
        rt.HostScene(f"mesh:{tmp_path / 'missing.ply'}", 16)


def test_builtin_scene_inventory(rt):
    b = rt.HostScene("basic", 16).desc.contents
    assert (b.n_objects, b.n_lights, b.n_textures) == (5, 5, 1)
    r = rt.HostScene("random_spheres", 16).desc.contents
    assert (r.n_objects, r.n_lights) == (1005, 4)           # scene.rs:111-186
    c2 = rt.HostScene("basic_suzanne", 16).desc.contents
    assert (c2.n_objects, c2.n_lights) == (505, 6)
    # every distinct Material is its own entry (interning must not match a freed material's address)
    for _ in range(5):
        assert rt.HostScene("basic_suzanne", 16).desc.contents.n_materials == 6
        assert rt.HostScene("scan", 16, 2000).desc.contents.n_materials == 9
    sc = rt.HostScene("scan", 16, 0).desc.contents
    assert sc.n_objects == 2374525 + 5
    with pytest.raises(rt.RtxError):
        rt.HostScene("nope", 16)


def test_default_camera_matrix(rt, orc):
    """camera.rs:12-26 / scene.rs:218-226: eye (0,1.5,5) looking at the origin."""
    s = rt.HostScene("basic", 64)
    c2w = np.array(list(s.desc.contents.camera_to_world), np.float64).reshape(4, 4).T
    np.testing.assert_allclose(c2w[:3, 3], (0, 1.5, 5), atol=1e-6)
    fwd = -c2w[:3, 2]
    np.testing.assert_allclose(fwd, np.array([0, -1.5, -5]) / np.linalg.norm([0, 1.5, 5]), atol=1e-6)
    np.testing.assert_allclose(c2w[:3, :3].T @ c2w[:3, :3], np.eye(3), atol=1e-6)
    # centre pixel ray of the oracle points along the camera's forward axis
    ray = orc.Scene(s).generate_ray(32, 32)
    np.testing.assert_allclose(ray[3:], fwd, atol=1e-6)


# ---------------------------------------------------------------- the Rust side of the boundary (integration/)
_RUST_OF_CTYPES = {"c_uint": "u32", "c_uint32": "u32", "c_int": "i32", "c_int32": "i32", "c_ulong": "u64", "c_uint64": "u64",
                   "c_float": "f32", "c_double": "f64", "c_ubyte": "u8", "c_uint8": "u8"}


def _rust_structs(text):
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", text, flags=re.S):
        out[m.group(1)] = [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub (\w+): ([^,\n]+),", m.group(2))]
    return out


def _ctypes_field_type(t):
    import ctypes as C
    if hasattr(t, "_length_"):      # array
        return f"[{_RUST_OF_CTYPES[t._type_.__name__]}; {t._length_}]"
    if hasattr(t, "contents") or t.__name__.startswith("LP_"):   # pointer: the ABI's arrays are all borrowed inputs
        return "*const " + _RUST_OF_CTYPES[t._type_.__name__]
    return _RUST_OF_CTYPES[t.__name__]


def test_rtx_sys_is_generated_from_the_header():
    """integration/rtx-sys/src/lib.rs is what tools/gen_rtx_sys.py makes of include/rtx.h today."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rtx_sys
    have = open(os.path.join(ROOT, "integration", "rtx-sys", "src", "lib.rs")).read()
    assert have == gen_rtx_sys.generate(), "include/rtx.h changed: run tools/gen_rtx_sys.py"


def test_rtx_sys_agrees_with_the_binding_the_gpu_tests_use(rt):
    """Independent of the generator: the #[repr(C)] structs of lib.rs have the fields, order and types of the
    ctypes Structures that every GPU test passes through the ABI, and lib.rs declares every function of rtx.h
    with as many arguments as the header."""
    text = open(os.path.join(ROOT, "integration", "rtx-sys", "src", "lib.rs")).read()
    rs = _rust_structs(text)
    for name, cls in (("rtx_scene_desc", rt.SceneDesc), ("rtx_stats", rt.Stats), ("rtx_bvh_info", rt.BvhInfo),
                      ("rtx_partition", rt.Partition)):
        want = [(f, _ctypes_field_type(t)) for f, t in cls._fields_]
        assert rs[name] == want, f"{name}: {[a for a, b in zip(rs[name], want) if a != b][:3]}"
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rtx.h")).read(), flags=re.S)
    protos = {m.group(1): m.group(2) for m in re.finditer(r"\b(rtx_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", header, flags=re.S)}
    rfns = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (rtx_\w+)\((.*?)\)(?: -> [^;]+)?;", text)}
    assert set(protos) == set(rfns), set(protos) ^ set(rfns)
    for name, args in protos.items():
        n_c = 0 if args.strip() in ("", "void") else args.count(",") + 1
        n_r = 0 if not rfns[name].strip() else rfns[name].count(",") + 1
        assert n_c == n_r, name
    # pointer mutability of the two calls the reference's render() patch makes
    assert "pub fn rtx_upload_scene(ctx: *mut rtx_ctx, desc: *const rtx_scene_desc) -> c_int;" in text
    assert "out_rgb8: *mut u8" in text and "out: *mut *mut rtx_ctx" in text


def test_flatten_rs_fills_every_field_of_the_descriptor(rt):
    text = open(os.path.join(ROOT, "integration", "flatten.rs")).read()
    body = text[text.index("rtx_scene_desc {"):]
    body = body[:body.index("\n        }")]
    filled = re.findall(r"^\s+(\w+):", body, flags=re.M)
    assert filled == [f for f, _ in rt.SceneDesc._fields_]
    # and uses the vocabulary constants of the header, not literals
    for const in ("RTX_OBJ_SPHERE", "RTX_OBJ_TRIANGLE", "RTX_OBJ_QUAD", "RTX_MAT_COMPOSITION", "RTX_TEX_IMAGE", "RTX_LIGHT_CONE"):
        assert const in text


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference checkout exists in the authoring container only")
def test_reference_patch_applies(tmp_path):
    """integration/reference.patch (pub(crate) accessors, mod flatten, the new body of world::render) applies to the
    reference's sources as they are."""
    import shutil
    import subprocess
    work = tmp_path / "ref"
    work.mkdir()
    shutil.copytree("/root/reference/src", work / "src")
    shutil.copy("/root/reference/Cargo.toml", work / "Cargo.toml")
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "integration", "reference.patch")], cwd=work,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "integration", "reference.patch")], cwd=work)
    world = (work / "src" / "world.rs").read_text()
    assert "rtx_upload_scene(ctx, &desc)" in world and "ThreadPool::new" not in world
    assert "pub(crate) object_type" in (work / "src" / "object.rs").read_text()


# ---------------------------------------------------------------- host matrix arithmetic (csrc/hostmath.h via rtx_host_m4)
# Every transform, camera and box that a GPU-vs-oracle parity test feeds to BOTH sides is built by this code, so it is
# pinned on its own: against the reference's own vectors, restated through the host's builders (not numpy), and
# against float64.

def _m4(rt, op, *a):
    return rt.host_m4(op, *a)


def test_hostmath_reference_get_scaling(rt):  # utils.rs:91-108, test_get_scaling, through the host's own builders
    rotate = _m4(rt, "mul", _m4(rt, "mul", _m4(rt, "from_angle_x", [120.0]), _m4(rt, "from_angle_y", [90.0])),
                 _m4(rt, "from_angle_z", [45.0]))
    translate = _m4(rt, "from_translation", [1.0, 0.5, -2.0])
    np.testing.assert_allclose(_m4(rt, "axis_scaling", _m4(rt, "mul", rotate, translate)), (1, 1, 1), atol=1e-6)
    scale = _m4(rt, "from_nonuniform_scale", [1.0, 2.0, 3.0])
    inv_t = _m4(rt, "invert", translate)                       # translate.inverse_transform().unwrap()
    m = _m4(rt, "mul", _m4(rt, "mul", _m4(rt, "mul", translate, rotate), scale), inv_t)
    np.testing.assert_allclose(_m4(rt, "axis_scaling", m), (1, 2, 3), atol=1e-5)   # approx's default epsilon is f32::EPSILON-relative; 1e-5 abs is tighter than the values' own rounding allows to fail
    np.testing.assert_allclose(_m4(rt, "axis_scaling", _m4(rt, "from_nonuniform_scale", [10.0, -0.5, 1.0])), (10, 0.5, 1), atol=0)


def test_hostmath_reference_bounding_boxes(rt):  # object.rs:379-407 (identity transform), epsilon 1e-4 as there
    eye = np.eye(4, dtype=np.float32)
    box = _m4(rt, "object_box", [0.0], [2.0, 0.0, 1.0, 1.5] + [0.0] * 8, eye)     # 0 = RTX_OBJ_SPHERE: centre, radius
    assert np.linalg.norm(box[:3] - (0.5, -1.5, -0.5)) < 1e-4 and np.linalg.norm(box[3:] - (3.5, 1.5, 2.5)) < 1e-4
    quad = [1.0, 1.0, 1.0, 3.0, 1.0, 1.0, 3.0, 1.0, 0.0, 1.0, 1.0, 0.0]
    box = _m4(rt, "object_box", [2.0], quad, eye)                                   # 2 = RTX_OBJ_QUAD
    assert np.linalg.norm(box[:3] - (1.0, 1.0, 0.0)) < 1e-4 and np.linalg.norm(box[3:] - (3.0, 1.0, 1.0)) < 1e-4


def _rot64(axis, deg):
    c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    m = np.eye(4)
    i, j = {"x": (1, 2), "y": (2, 0), "z": (0, 1)}[axis]
    m[i, i], m[j, j], m[j, i], m[i, j] = c, c, s, -s          # right-handed rotation about `axis`
    return m


def _look_at64(eye, center, up):  # cgmath Matrix4::look_at (RH): rows (s, u, -f), translation (-eye.s, -eye.u, eye.f)
    eye, center, up = (np.asarray(v, np.float64) for v in (eye, center, up))
    f = (center - eye) / np.linalg.norm(center - eye)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[:3, 3] = (-eye @ s, -eye @ u, eye @ f)
    return m


def test_hostmath_against_float64(rt):
    """The builders against float64 numpy at 1e-6: rotations, the three scene transforms (scene.rs:35-38, 194-199),
    default_camera's look_at and its inverse (scene.rs:218-226, camera.rs:19-20), a general affine inverse."""
    for axis in "xyz":
        for deg in (0.0, 20.0, 35.0, -20.0, 90.0, 120.0, 45.0, 180.0, 270.0):
            np.testing.assert_allclose(_m4(rt, "from_angle_" + axis, [deg]), _rot64(axis, deg), atol=1e-6)
    # scene.rs:35-38: translate * rotate_x(90) * translate.invert()
    t = _m4(rt, "from_translation", [1.0, 0.5, -2.0])
    o2w = _m4(rt, "mul", _m4(rt, "mul", t, _m4(rt, "from_angle_x", [90.0])), _m4(rt, "invert", t))
    t64 = np.eye(4)
    t64[:3, 3] = (1.0, 0.5, -2.0)
    np.testing.assert_allclose(o2w, t64 @ _rot64("x", 90) @ np.linalg.inv(t64), atol=1e-6)
    # scene.rs:194-199: translation(0) * scale(1) * (Ry(20) * Rz(35) * Rx(-20))
    rot = _m4(rt, "mul", _m4(rt, "mul", _m4(rt, "from_angle_y", [20.0]), _m4(rt, "from_angle_z", [35.0])), _m4(rt, "from_angle_x", [-20.0]))
    suz = _m4(rt, "mul", _m4(rt, "mul", _m4(rt, "from_translation", [0.0, 0.0, 0.0]), _m4(rt, "from_nonuniform_scale", [1.0, 1.0, 1.0])), rot)
    suz64 = _rot64("y", 20) @ _rot64("z", 35) @ _rot64("x", -20)
    np.testing.assert_allclose(suz, suz64, atol=1e-6)
    np.testing.assert_allclose(_m4(rt, "invert", suz), np.linalg.inv(suz64), atol=1e-6)
    # default_camera: look_at((0,1.5,5), origin, +y) and camera_to_world = its inverse
    la = _m4(rt, "look_at", [0.0, 1.5, 5.0], [0.0, 0.0, 0.0], [0.0, 1.0, 0.0])
    la64 = _look_at64((0, 1.5, 5), (0, 0, 0), (0, 1, 0))
    np.testing.assert_allclose(la, la64, atol=1e-6)
    np.testing.assert_allclose(_m4(rt, "invert", la), np.linalg.inv(la64), atol=2e-6)
    # ... which is the matrix the scene descriptions carry
    s = rt.HostScene("basic", 16)
    c2w = np.array(list(s.desc.contents.camera_to_world), np.float32).reshape(4, 4).T
    np.testing.assert_array_equal(c2w, _m4(rt, "invert", la))
    # a general affine matrix (rotation, non-uniform scale incl. a negative one, shear through the product, translation)
    rng = np.random.default_rng(5)
    for _ in range(20):
        a = np.eye(4)
        a[:3, :3] = rng.uniform(-2, 2, (3, 3)) + 2.5 * np.diag(rng.choice([-1.0, 1.0], 3))
        a[:3, 3] = rng.uniform(-5, 5, 3)
        a32 = a.astype(np.float32)
        inv = _m4(rt, "invert", a32)
        np.testing.assert_allclose(inv.astype(np.float64) @ a32.astype(np.float64), np.eye(4), atol=2e-5)
        p = rng.uniform(-3, 3, 3).astype(np.float32)
        np.testing.assert_allclose(_m4(rt, "transform_point", a32, p), (a32.astype(np.float64) @ np.append(p, 1.0))[:3], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(_m4(rt, "transform_vector", a32, p), a32[:3, :3].astype(np.float64) @ p, rtol=1e-5, atol=1e-5)
    with pytest.raises(rt.RtxError) as e:
        _m4(rt, "invert", np.zeros((4, 4), np.float32))
    assert e.value.status == rt.RTX_ERR_DEGENERATE
