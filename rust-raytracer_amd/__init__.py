"""ctypes binding of the MI355X trace backend (include/rtx.h, include/rtx_host.h).

The product is the C-ABI shared library ``librtx_hip.so`` built from ``csrc/``; this module
only loads it and wraps the entry points for the tests, ``bench.py`` and
``__graft_entry__.py``.  There is no CPU fallback: if the library is missing, or no HIP
device is usable, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_LIB") or os.path.join(_HERE, "librtx_hip.so")
REPO_ROOT = os.path.dirname(_HERE)
MEDIA_DIR = os.path.join(REPO_ROOT, "assets")

RTX_OK = 0
RTX_ERR_INVALID, RTX_ERR_HIP, RTX_ERR_NO_SCENE, RTX_ERR_CAPACITY, RTX_ERR_DEGENERATE, RTX_ERR_IO, RTX_ERR_COMM = -1, -2, -3, -4, -5, -6, -7
FLAG_COUNT_VISITS = 1
FLAG_REFERENCE_TRAVERSAL = 2

_f32p, _i32p, _u8p, _u32p, _u64p = (C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_uint64))


class SceneDesc(C.Structure):
    """rtx_scene_desc (identical layout to the oracle's orc_scene_desc)."""
    _fields_ = [
        ("n_objects", C.c_uint32), ("obj_type", _i32p), ("obj_verts", _f32p), ("obj_xform", _i32p),
        ("obj_material", _i32p),
        ("n_xforms", C.c_uint32), ("xforms", _f32p),
        ("n_materials", C.c_uint32), ("mat_root", _i32p), ("mat_tex_kind", _i32p), ("mat_tex_color", _f32p),
        ("mat_tex_id", _i32p),
        ("n_mat_nodes", C.c_uint32), ("node_type", _i32p), ("node_params", _f32p), ("node_child_first", _i32p),
        ("node_child_count", _i32p),
        ("n_mat_children", C.c_uint32), ("child_node", _i32p), ("child_coef", _f32p),
        ("n_textures", C.c_uint32), ("tex_w", _i32p), ("tex_h", _i32p), ("tex_offset", _u64p), ("tex_data", _u8p),
        ("n_lights", C.c_uint32), ("light_type", _i32p), ("light_pos", _f32p), ("light_dir", _f32p),
        ("light_angle_deg", _f32p), ("light_color", _f32p),
        ("background", C.c_float * 3), ("camera_to_world", C.c_float * 16),
        ("width", C.c_uint32), ("height", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("bounce_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("closest_nodes", C.c_uint64), ("closest_prims", C.c_uint64),
                ("shadow_nodes", C.c_uint64), ("shadow_prims", C.c_uint64),
                ("trace_ms", C.c_double), ("path_kernel_ms", C.c_double), ("resolve_ms", C.c_double),
                ("path_launches", C.c_uint32), ("batches", C.c_uint32), ("max_ray_records", C.c_uint32),
                ("reserved", C.c_uint32)]

    @property
    def rays(self):
        return self.primary_rays + self.bounce_rays + self.shadow_rays


class BvhInfo(C.Structure):
    _fields_ = [("n_objects", C.c_uint32), ("n_internal", C.c_uint32), ("n_leaves", C.c_uint32), ("depth", C.c_uint32),
                ("total_sa", C.c_float), ("build_seconds", C.c_double), ("flatten_seconds", C.c_double),
                ("upload_seconds", C.c_double), ("device_bytes", C.c_uint64)]


class Partition(C.Structure):
    _fields_ = [("tile", C.c_uint32), ("rank", C.c_uint32), ("nranks", C.c_uint32)]


class RtxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"rtx status {status}: {msg}")
        self.status = status


_lib = None


def lib():
    """Loads librtx_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `make -C rust-raytracer_amd/csrc` (or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rtx_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.rtx_destroy.argtypes = [vp]
    L.rtx_destroy.restype = None
    L.rtx_last_error.argtypes = [vp]
    L.rtx_last_error.restype = C.c_char_p
    L.rtx_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    L.rtx_bvh_info_get.argtypes = [vp, C.POINTER(BvhInfo)]
    L.rtx_bvh_digest.argtypes = [vp, _u32p, _u32p]
    L.rtx_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(Partition), C.c_uint32, _u8p, _f32p,
                             C.POINTER(Stats)]
    L.rtx_render_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(Partition), C.c_uint32, vp,
                                    C.POINTER(Stats)]
    L.rtx_render_device_noalloc.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(Partition), C.c_uint32, vp]
    L.rtx_set_build_threads.argtypes = [vp, C.c_int]
    L.rtx_render_device_slot.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(Partition), C.c_uint32,
                                         vp, C.POINTER(Stats)]
    L.rtx_comm_unique_id.argtypes = [C.c_char_p]
    L.rtx_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.rtx_comm_rank.argtypes = [vp]
    L.rtx_comm_size.argtypes = [vp]
    L.rtx_gather_bytes_per_rank.argtypes = [vp, C.c_uint32]
    L.rtx_gather_bytes_per_rank.restype = C.c_uint64
    L.rtx_gather_image.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.rtx_assemble_image.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.rtx_slot_count.argtypes = [vp]
    L.rtx_slot_stream.argtypes = [vp, C.c_uint32]
    L.rtx_slot_stream.restype = vp
    L.rtx_slot_synchronize.argtypes = [vp, C.c_uint32]
    L.rtx_slot_stats.argtypes = [vp, C.c_uint32, C.POINTER(Stats)]
    L.rtx_partition_pixels.argtypes = [vp, C.POINTER(Partition)]
    L.rtx_partition_pixels.restype = C.c_uint64
    L.rtx_stream.argtypes = [vp]
    L.rtx_stream.restype = vp
    L.rtx_synchronize.argtypes = [vp]
    L.rtx_trace_primary.argtypes = [vp, C.c_uint32, C.c_int, C.c_uint64, C.c_uint32, _u32p, _f32p]
    L.rtx_trace_rays.argtypes = [vp, _f32p, C.c_uint64, C.c_uint32, _u32p, _f32p]
    L.rtx_shade_rays.argtypes = [vp, _f32p, C.c_uint64, C.c_uint32, _f32p]
    L.rtx_scene_builtin.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint64]
    L.rtx_scene_builtin.restype = vp
    L.rtx_scene_get_desc.argtypes = [vp]
    L.rtx_scene_get_desc.restype = C.POINTER(SceneDesc)
    L.rtx_scene_free.argtypes = [vp]
    L.rtx_scene_free.restype = None
    L.rtx_host_last_error.restype = C.c_char_p
    L.rtx_png_read_rgb8.argtypes = [C.c_char_p, _u32p, _u32p, C.POINTER(_u8p)]
    L.rtx_png_write_rgb8.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, _u8p]
    L.rtx_host_free.argtypes = [vp]
    L.rtx_host_free.restype = None
    L.rtx_render_builtin_to_png.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                            C.c_char_p, C.c_uint32, C.c_int]
    L.rtx_host_bvh_digest.argtypes = [C.POINTER(SceneDesc), C.c_int, _u32p, _u32p, _u32p, _f32p, _u32p]
    L.rtx_scene_hit_overhang.argtypes = [C.POINTER(SceneDesc)]
    L.rtx_scene_hit_overhang.restype = C.c_float
    L.rtx_host_m4.argtypes = [C.c_int, _f32p, _f32p]
    _lib = L
    return L


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


# config 3's real inputs: Stanford scans that "scan" / "scan_glass" pick up from the media directory
# when present (dragon, bunny, armadillo, Buddha -- the stand-ins' order)
SCAN_FILES = ("dragon_vrip.ply", "bun_zipper.ply", "Armadillo.ply", "happy_vrip.ply")


class HostScene:
    """A built-in scene (rtx_scene_builtin); .desc is the rtx_scene_desc pointer."""

    def __init__(self, name, width, param=0, media_dir=None):
        L = lib()
        self._h = L.rtx_scene_builtin(name.encode(), width, (media_dir or MEDIA_DIR).encode(), param)
        if not self._h:
            raise RtxError(RTX_ERR_INVALID, L.rtx_host_last_error().decode())
        self.desc = L.rtx_scene_get_desc(self._h)
        self.name = name

    @property
    def width(self):
        return self.desc.contents.width

    @property
    def height(self):
        return self.desc.contents.height

    @property
    def n_objects(self):
        return self.desc.contents.n_objects

    def close(self):
        if self._h:
            lib().rtx_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ArrayScene:
    """A scene described by numpy arrays (for tests that build their own geometry).

    objects: list of ("sphere", (cx,cy,cz), r, material, xform=0) / ("triangle", a, b, c, material, xform=0) /
             ("quad", a, b, c, d, material, xform=0)
    materials: list of dicts {"root": node, "tex_kind": 0|1|2, "color": (r,g,b), "tex_id": 0}
    mat_nodes: list of (type, (p0,p1,p2), [(child_node, coef), ...])
    lights: list of (type, pos, dir, angle_deg, color)
    """

    def __init__(self, objects, width=8, height=8, materials=None, mat_nodes=None, lights=(), xforms=None,
                 background=(0.2, 0.2, 0.2), camera_to_world=None, textures=()):
        n = len(objects)
        self.obj_type = np.zeros(max(n, 1), np.int32)
        self.obj_verts = np.zeros((max(n, 1), 12), np.float32)
        self.obj_xform = np.zeros(max(n, 1), np.int32)
        self.obj_material = np.zeros(max(n, 1), np.int32)
        for i, o in enumerate(objects):
            kind = o[0]
            if kind == "sphere":
                self.obj_type[i] = 0
                self.obj_verts[i, 0:3] = o[1]
                self.obj_verts[i, 3] = o[2]
                rest = o[3:]
            elif kind == "triangle":
                self.obj_type[i] = 1
                self.obj_verts[i, 0:9] = np.concatenate([np.asarray(v, np.float32) for v in o[1:4]])
                rest = o[4:]
            else:
                self.obj_type[i] = 2
                self.obj_verts[i, 0:12] = np.concatenate([np.asarray(v, np.float32) for v in o[1:5]])
                rest = o[5:]
            self.obj_material[i] = rest[0] if len(rest) > 0 else 0
            self.obj_xform[i] = rest[1] if len(rest) > 1 else 0
        ident = np.eye(4, dtype=np.float32)
        if xforms is None:
            xforms = [(ident, ident)]
        # column-major storage: flatten the transpose of the row-major numpy matrix
        self.xforms = np.concatenate([np.concatenate([np.asarray(a, np.float32).T.reshape(-1),
                                                      np.asarray(b, np.float32).T.reshape(-1)]) for a, b in xforms])
        self.n_xforms = len(xforms)
        if mat_nodes is None:
            mat_nodes = [(0, (0, 0, 0), [])]
        if materials is None:
            materials = [{"root": 0, "tex_kind": 0, "color": (0, 0, 0), "tex_id": 0}]
        self.node_type = np.array([m[0] for m in mat_nodes], np.int32)
        self.node_params = np.array([m[1] for m in mat_nodes], np.float32).reshape(-1)
        first, count, cn, cc = [], [], [], []
        for m in mat_nodes:
            first.append(len(cn))
            count.append(len(m[2]))
            for node, coef in m[2]:
                cn.append(node)
                cc.append(coef)
        self.node_child_first = np.array(first, np.int32)
        self.node_child_count = np.array(count, np.int32)
        self.child_node = np.array(cn + [0], np.int32)
        self.child_coef = np.array(cc + [0], np.float32)
        self.n_children = len(cn)
        self.mat_root = np.array([m["root"] for m in materials], np.int32)
        self.mat_tex_kind = np.array([m.get("tex_kind", 0) for m in materials], np.int32)
        self.mat_tex_color = np.array([m.get("color", (0, 0, 0)) for m in materials], np.float32).reshape(-1)
        self.mat_tex_id = np.array([m.get("tex_id", 0) for m in materials], np.int32)
        self.tex_w = np.array([t.shape[1] for t in textures] + [0], np.int32)
        self.tex_h = np.array([t.shape[0] for t in textures] + [0], np.int32)
        offs, blob, pos = [], [], 0
        for t in textures:
            offs.append(pos)
            b = np.ascontiguousarray(t, np.uint8).reshape(-1)
            blob.append(b)
            pos += b.size
        self.tex_offset = np.array(offs + [0], np.uint64)
        self.tex_data = np.concatenate(blob) if blob else np.zeros(4, np.uint8)
        nl = len(lights)
        self.light_type = np.array([l[0] for l in lights] + [0], np.int32)
        self.light_pos = np.array([l[1] for l in lights] + [(0, 0, 0)], np.float32).reshape(-1)
        self.light_dir = np.array([l[2] for l in lights] + [(0, 0, 0)], np.float32).reshape(-1)
        self.light_angle = np.array([l[3] for l in lights] + [0], np.float32)
        self.light_color = np.array([l[4] for l in lights] + [(0, 0, 0)], np.float32).reshape(-1)
        d = SceneDesc()
        d.n_objects = n
        d.obj_type = _ptr(self.obj_type, _i32p)
        d.obj_verts = _ptr(self.obj_verts, _f32p)
        d.obj_xform = _ptr(self.obj_xform, _i32p)
        d.obj_material = _ptr(self.obj_material, _i32p)
        d.n_xforms = self.n_xforms
        d.xforms = _ptr(self.xforms, _f32p)
        d.n_materials = len(materials)
        d.mat_root = _ptr(self.mat_root, _i32p)
        d.mat_tex_kind = _ptr(self.mat_tex_kind, _i32p)
        d.mat_tex_color = _ptr(self.mat_tex_color, _f32p)
        d.mat_tex_id = _ptr(self.mat_tex_id, _i32p)
        d.n_mat_nodes = len(mat_nodes)
        d.node_type = _ptr(self.node_type, _i32p)
        d.node_params = _ptr(self.node_params, _f32p)
        d.node_child_first = _ptr(self.node_child_first, _i32p)
        d.node_child_count = _ptr(self.node_child_count, _i32p)
        d.n_mat_children = self.n_children
        d.child_node = _ptr(self.child_node, _i32p)
        d.child_coef = _ptr(self.child_coef, _f32p)
        d.n_textures = len(textures)
        d.tex_w = _ptr(self.tex_w, _i32p)
        d.tex_h = _ptr(self.tex_h, _i32p)
        d.tex_offset = _ptr(self.tex_offset, _u64p)
        d.tex_data = _ptr(self.tex_data, _u8p)
        d.n_lights = nl
        d.light_type = _ptr(self.light_type, _i32p)
        d.light_pos = _ptr(self.light_pos, _f32p)
        d.light_dir = _ptr(self.light_dir, _f32p)
        d.light_angle_deg = _ptr(self.light_angle, _f32p)
        d.light_color = _ptr(self.light_color, _f32p)
        d.background[:] = background
        c2w = np.eye(4, dtype=np.float32) if camera_to_world is None else np.asarray(camera_to_world, np.float32)
        d.camera_to_world[:] = list(c2w.T.reshape(-1))
        d.width = width
        d.height = height
        self._desc = d
        self.desc = C.pointer(d)
        self.width, self.height, self.n_objects = width, height, n


def hit_overhang(scene):
    """rtx_scene_hit_overhang: how far a hit can lie outside its object's reference box (0 for most scenes)."""
    return float(lib().rtx_scene_hit_overhang(scene.desc))


M4_OPS = {"from_angle_x": (0, 1, 16), "from_angle_y": (1, 1, 16), "from_angle_z": (2, 1, 16), "from_translation": (3, 3, 16),
          "from_nonuniform_scale": (4, 3, 16), "mul": (5, 32, 16), "invert": (6, 16, 16), "look_at": (7, 9, 16),
          "axis_scaling": (8, 16, 3), "transform_point": (9, 19, 3), "transform_vector": (10, 19, 3), "object_box": (11, 29, 6)}


def host_m4(op, *args):
    """rtx_host_m4: the host's cgmath-order f32 matrix arithmetic (csrc/hostmath.h).  Matrices go in and come out as
    4x4 numpy arrays in mathematical layout (m[row, col]); the ABI's 16 floats are column-major."""
    code, n_in, n_out = M4_OPS[op]
    flat = []
    for a in args:
        a = np.asarray(a, np.float32)
        flat.append(a.T.reshape(-1) if a.shape == (4, 4) else a.reshape(-1))
    buf = np.ascontiguousarray(np.concatenate(flat), np.float32)
    assert buf.size == n_in, (op, buf.size, n_in)
    out = np.zeros(n_out, np.float32)
    rc = lib().rtx_host_m4(code, _ptr(buf, _f32p), _ptr(out, _f32p))
    if rc < 0:
        raise RtxError(rc, f"rtx_host_m4({op})")
    return out.reshape(4, 4).T.copy() if n_out == 16 else out


def host_bvh_digest(scene, threads=8):
    """rtx_host_bvh_digest: (leaf_order, preorder_sizes, depth, total_sa, n_internal); no GPU needed."""
    L = lib()
    n = scene.n_objects
    leaf = np.zeros(max(n, 1), np.uint32)
    pre = np.zeros(2 * max(n, 1) + 1, np.uint32)
    depth, ni, sa = C.c_uint32(0), C.c_uint32(0), C.c_float(0)
    rc = L.rtx_host_bvh_digest(scene.desc, threads, _ptr(leaf, _u32p), _ptr(pre, _u32p), C.byref(depth), C.byref(sa),
                               C.byref(ni))
    if rc < 0:
        raise RtxError(rc, L.rtx_last_error(None).decode())
    return leaf[:n], pre[:rc], depth.value, sa.value, ni.value


class Context:
    """rtx_ctx on one HIP device."""

    def __init__(self, device=0):
        L = lib()
        h = C.c_void_p()
        rc = L.rtx_init(device, C.byref(h))
        if rc != RTX_OK:
            raise RtxError(rc, L.rtx_last_error(None).decode())
        self._h = h
        self.width = self.height = 0

    def _check(self, rc):
        if rc < 0:
            raise RtxError(rc, lib().rtx_last_error(self._h).decode())
        return rc

    def upload(self, scene):
        self._check(lib().rtx_upload_scene(self._h, scene.desc))
        self.width, self.height = scene.desc.contents.width, scene.desc.contents.height
        info = BvhInfo()
        self._check(lib().rtx_bvh_info_get(self._h, C.byref(info)))
        self.n_objects = info.n_objects
        return info

    def bvh_digest(self):
        n = self.n_objects
        leaf = np.zeros(max(n, 1), np.uint32)
        pre = np.zeros(2 * max(n, 1) + 1, np.uint32)
        k = self._check(lib().rtx_bvh_digest(self._h, _ptr(leaf, _u32p), _ptr(pre, _u32p)))
        return leaf[:n], pre[:k]

    def render(self, spp=1, bounces=4, seed=0, part=None, flags=0, want_f32=False):
        img = np.zeros((self.height, self.width, 3), np.uint8)
        f32 = np.zeros((self.height, self.width, 3), np.float32) if want_f32 else None
        st = Stats()
        p = C.byref(Partition(*part)) if part else None
        self._check(lib().rtx_render(self._h, spp, bounces, seed, p, flags, _ptr(img, _u8p),
                                     _ptr(f32, _f32p) if want_f32 else None, C.byref(st)))
        return (img, f32, st) if want_f32 else (img, st)

    def partition_pixels(self, part=None):
        p = C.byref(Partition(*part)) if part else None
        return lib().rtx_partition_pixels(self._h, p)

    def render_device(self, d_ptr, spp=1, bounces=4, seed=0, part=None, flags=0, stats=True, slot=0):
        st = Stats()
        p = C.byref(Partition(*part)) if part else None
        self._check(lib().rtx_render_device_slot(self._h, slot, spp, bounces, seed, p, flags, C.c_void_p(d_ptr),
                                                 C.byref(st) if stats else None))
        return st

    def comm_init(self, comm_id, rank, nranks):
        """rtx_comm_init: joins the RCCL communicator that `comm_id` (comm_unique_id() of rank 0) names."""
        self._check(lib().rtx_comm_init(self._h, comm_id, rank, nranks))

    def gather_bytes_per_rank(self, tile=32):
        return lib().rtx_gather_bytes_per_rank(self._h, tile)

    def gather_image(self, d_tiles, d_all, d_image=0, slot=0, tile=32):
        """rtx_gather_image: all-gather of the ranks' tile buffers on the slot's stream, then the row-major image."""
        self._check(lib().rtx_gather_image(self._h, slot, tile, C.c_void_p(d_tiles), C.c_void_p(d_all),
                                           C.c_void_p(d_image) if d_image else None))

    def assemble_image(self, d_all, d_image, nranks, slot=0, tile=32):
        """rtx_assemble_image: the reordering half of gather_image alone (all-gathered tile buffers of nranks ranks -> image)."""
        self._check(lib().rtx_assemble_image(self._h, slot, tile, nranks, C.c_void_p(d_all), C.c_void_p(d_image)))

    @property
    def slots(self):
        return self._check(lib().rtx_slot_count(self._h))

    def slot_stream(self, slot):
        return int(lib().rtx_slot_stream(self._h, slot) or 0)

    def slot_stats(self, slot):
        """Waits for the slot's last render; its rtx_stats."""
        st = Stats()
        self._check(lib().rtx_slot_stats(self._h, slot, C.byref(st)))
        return st

    def slot_synchronize(self, slot):
        self._check(lib().rtx_slot_synchronize(self._h, slot))

    def render_device_noalloc(self, d_ptr, spp=1, bounces=4, seed=0, part=None, flags=0):
        """rtx_render_device_noalloc: wavefront state in the library's static device array (asynchronous)."""
        p = C.byref(Partition(*part)) if part else None
        self._check(lib().rtx_render_device_noalloc(self._h, spp, bounces, seed, p, flags, C.c_void_p(d_ptr)))

    def set_build_threads(self, n):
        self._check(lib().rtx_set_build_threads(self._h, n))

    @property
    def stream(self):
        """The hipStream_t (as an integer) that this context's renders are ordered on (rtx_stream)."""
        return int(lib().rtx_stream(self._h) or 0)

    def synchronize(self):
        self._check(lib().rtx_synchronize(self._h))

    def trace_primary(self, sample=0, use_jitter=False, seed=0, flags=0):
        obj = np.full((self.height, self.width), 0xFFFFFFFF, np.uint32)
        t = np.zeros((self.height, self.width), np.float32)
        self._check(lib().rtx_trace_primary(self._h, sample, int(use_jitter), seed, flags, _ptr(obj, _u32p),
                                            _ptr(t, _f32p)))
        return obj, t

    def trace_rays(self, org_dir, flags=0):
        od = np.ascontiguousarray(org_dir, np.float32).reshape(-1, 6)
        n = od.shape[0]
        obj = np.zeros(max(n, 1), np.uint32)
        t = np.zeros(max(n, 1), np.float32)
        self._check(lib().rtx_trace_rays(self._h, _ptr(od, _f32p), n, flags, _ptr(obj, _u32p), _ptr(t, _f32p)))
        return obj[:n], t[:n]

    def shade_rays(self, org_dir, depth=4):
        od = np.ascontiguousarray(org_dir, np.float32).reshape(-1, 6)
        n = od.shape[0]
        rgb = np.zeros((max(n, 1), 3), np.float32)
        self._check(lib().rtx_shade_rays(self._h, _ptr(od, _f32p), n, depth, _ptr(rgb, _f32p)))
        return rgb[:n]

    def close(self):
        if self._h:
            lib().rtx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_hip = None


def _hiprt():
    """The HIP runtime that librtx_hip.so itself is linked against (same soname: the copy already
    mapped), for the few tests and bench paths that need a caller-owned device buffer."""
    global _hip
    if _hip is None:
        lib()
        h = C.CDLL("libamdhip64.so.7")
        h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        h.hipFree.argtypes = [C.c_void_p]
        h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        h.hipDeviceSynchronize.argtypes = []
        _hip = h
    return _hip


class DeviceBuffer:
    """`nbytes` of device memory on the current HIP device (hipMalloc), filled with `fill`."""

    def __init__(self, nbytes, fill=0):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        rc = _hiprt().hipMalloc(C.byref(p), max(self.nbytes, 256))
        if rc != 0 or not p.value:
            raise RtxError(RTX_ERR_HIP, f"hipMalloc({nbytes}) failed with {rc}")
        self.ptr = p.value
        _hiprt().hipMemset(C.c_void_p(self.ptr), fill, max(self.nbytes, 256))
        _hiprt().hipDeviceSynchronize()

    def to_host(self):
        """Synchronous copy to a numpy uint8 array (after all work of the device has finished)."""
        out = np.empty(self.nbytes, np.uint8)
        _hiprt().hipDeviceSynchronize()
        rc = _hiprt().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), self.nbytes, 2)  # hipMemcpyDeviceToHost
        if rc != 0:
            raise RtxError(RTX_ERR_HIP, f"hipMemcpy failed with {rc}")
        return out

    def free(self):
        if self.ptr:
            _hiprt().hipFree(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def comm_info():
    """rtx_comm_info: the RCCL and HIP runtime this process has bound (version and resolved path of each)."""
    buf = C.create_string_buffer(1024)
    L = lib()
    L.rtx_comm_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    rc = L.rtx_comm_info(None, buf, 1024)
    if rc < 0:
        raise RtxError(rc, "rtx_comm_info")
    return buf.value.decode()


def comm_unique_id():
    """rtx_comm_unique_id: 128 bytes that name a new RCCL communicator (rank 0 makes them, every rank gets them)."""
    buf = C.create_string_buffer(128)
    rc = lib().rtx_comm_unique_id(buf)
    if rc != RTX_OK:
        raise RtxError(rc, lib().rtx_last_error(None).decode())
    return buf.raw


def kernel_source_hash():
    """Identifies the kernel sources a PMC measurement belongs to (profiles/pmc.json; bench.py drops stale ones)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "kernels.h", "device_types.h"):
        h.update(open(os.path.join(_HERE, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def untile(buf, width, height, tile=32):
    """Packed tiles of a whole frame ([tile][row in tile][col in tile][rgb], what rtx_render_device writes for
    part = (tile, 0, 1)) -> row-major (height, width, 3) image."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    a = np.asarray(buf, np.uint8)[:tx * ty * tile * tile * 3].reshape(ty, tx, tile, tile, 3)
    return np.ascontiguousarray(a.transpose(0, 2, 1, 3, 4).reshape(ty * tile, tx * tile, 3)[:height, :width])


def png_read(path):
    L = lib()
    w, h, p = C.c_uint32(), C.c_uint32(), _u8p()
    rc = L.rtx_png_read_rgb8(path.encode(), C.byref(w), C.byref(h), C.byref(p))
    if rc != RTX_OK:
        raise RtxError(rc, L.rtx_host_last_error().decode())
    a = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    L.rtx_host_free(p)
    return a


def png_write(path, img):
    L = lib()
    a = np.ascontiguousarray(img, np.uint8)
    rc = L.rtx_png_write_rgb8(path.encode(), a.shape[1], a.shape[0], _ptr(a, _u8p))
    if rc != RTX_OK:
        raise RtxError(rc, L.rtx_host_last_error().decode())
