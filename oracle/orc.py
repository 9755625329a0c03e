"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORC_LIB") or os.path.join(_HERE, "liboracle.so")
_f32p, _u8p, _u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("bounce_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("build_seconds", C.c_double), ("trace_seconds", C.c_double),
                ("closest_nodes", C.c_uint64), ("closest_prims", C.c_uint64), ("shadow_nodes", C.c_uint64),
                ("shadow_prims", C.c_uint64)]

    @property
    def rays(self):
        return self.primary_rays + self.bounce_rays + self.shadow_rays


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `make -C oracle`")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.orc_scene_create.argtypes = [vp]
    L.orc_scene_create.restype = vp
    L.orc_scene_destroy.argtypes = [vp]
    L.orc_scene_destroy.restype = None
    L.orc_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int] + [C.c_uint32] * 6 + [_u8p, _f32p,
                                                                                                    C.POINTER(Stats)]
    L.orc_trace_primary.argtypes = [vp, C.c_uint32, C.c_int, C.c_uint64, C.c_int, _u32p, _f32p]
    L.orc_trace_rays.argtypes = [vp, _f32p, C.c_uint64, C.c_int, _u32p, _f32p]
    L.orc_shade_rays.argtypes = [vp, _f32p, C.c_uint64, C.c_uint32, _f32p]
    L.orc_bvh_digest.argtypes = [vp, _u32p, _u32p, _u32p, _f32p]
    L.orc_bvh_digest.restype = C.c_uint32
    L.orc_generate_ray.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, _f32p]
    L.orc_generate_ray.restype = None
    L.orc_aabb_surface_area.argtypes = [_f32p, _f32p]
    L.orc_aabb_surface_area.restype = C.c_float
    L.orc_aabb_intersect.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p]
    L.orc_ray_get_t.argtypes = [_f32p, _f32p, _f32p]
    L.orc_ray_get_t.restype = C.c_float
    L.orc_object_intersect.argtypes = [vp, C.c_uint32, _f32p, _f32p, _f32p]
    L.orc_object_bbox.argtypes = [vp, C.c_uint32, _f32p, _f32p]
    L.orc_object_bbox.restype = None
    L.orc_bvh_split.argtypes = [vp, C.c_int, _u32p, _u32p]
    L.orc_bvh_split.restype = None
    L.orc_clamp.argtypes = [C.c_float] * 3
    L.orc_clamp.restype = C.c_float
    L.orc_reflect.argtypes = [_f32p, _f32p, _f32p]
    L.orc_reflect.restype = None
    L.orc_refract.argtypes = [_f32p, _f32p, C.c_float, _f32p]
    L.orc_refract.restype = None
    L.orc_component_wise_range.argtypes = [_f32p, C.c_uint32, _f32p, _f32p]
    L.orc_component_wise_range.restype = None
    L.orc_axis_scaling.argtypes = [_f32p, _f32p]
    L.orc_axis_scaling.restype = None
    L.orc_jitter_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_jitter_u32.restype = C.c_uint32
    _lib = L
    return L


def _v(a):
    arr = np.ascontiguousarray(a, np.float32)
    return arr, arr.ctypes.data_as(_f32p)


class Scene:
    """Oracle scene built from a scene-description pointer (same layout as rtx_scene_desc)."""

    def __init__(self, scene):
        self._keep = scene  # the arrays behind the pointers must outlive the call
        d = scene.desc.contents
        self.width, self.height, self.n_objects = d.width, d.height, d.n_objects
        self._h = lib().orc_scene_create(C.cast(scene.desc, C.c_void_p))

    def render(self, spp=1, bounces=4, seed=0, threads=8, region=None, want_f32=False):
        x0, x1, xs, y0, y1, ys = region or (0, self.width, 1, 0, self.height, 1)
        img = np.zeros((self.height, self.width, 3), np.uint8)
        f32 = np.zeros((self.height, self.width, 3), np.float32) if want_f32 else None
        st = Stats()
        rc = lib().orc_render(self._h, spp, bounces, seed, threads, x0, x1, xs, y0, y1, ys,
                              img.ctypes.data_as(_u8p), f32.ctypes.data_as(_f32p) if want_f32 else None, C.byref(st))
        assert rc == 0
        return (img, f32, st) if want_f32 else (img, st)

    def trace_primary(self, sample=0, use_jitter=False, seed=0, threads=8):
        obj = np.zeros((self.height, self.width), np.uint32)
        t = np.zeros((self.height, self.width), np.float32)
        lib().orc_trace_primary(self._h, sample, int(use_jitter), seed, threads, obj.ctypes.data_as(_u32p),
                                t.ctypes.data_as(_f32p))
        return obj, t

    def trace_rays(self, org_dir, threads=8):
        od = np.ascontiguousarray(org_dir, np.float32).reshape(-1, 6)
        n = od.shape[0]
        obj = np.zeros(max(n, 1), np.uint32)
        t = np.zeros(max(n, 1), np.float32)
        lib().orc_trace_rays(self._h, od.ctypes.data_as(_f32p), n, threads, obj.ctypes.data_as(_u32p),
                             t.ctypes.data_as(_f32p))
        return obj[:n], t[:n]

    def shade_rays(self, org_dir, depth=4):
        od = np.ascontiguousarray(org_dir, np.float32).reshape(-1, 6)
        n = od.shape[0]
        rgb = np.zeros((max(n, 1), 3), np.float32)
        lib().orc_shade_rays(self._h, od.ctypes.data_as(_f32p), n, depth, rgb.ctypes.data_as(_f32p))
        return rgb[:n]

    def bvh_digest(self):
        n = self.n_objects
        leaf = np.zeros(max(n, 1), np.uint32)
        pre = np.zeros(2 * max(n, 1) + 1, np.uint32)
        depth, sa = C.c_uint32(0), C.c_float(0)
        k = lib().orc_bvh_digest(self._h, leaf.ctypes.data_as(_u32p), pre.ctypes.data_as(_u32p), C.byref(depth),
                                 C.byref(sa))
        return leaf[:n], pre[:k], depth.value, sa.value

    def generate_ray(self, px, py, sample=0, use_jitter=False, seed=0):
        out = np.zeros(6, np.float32)
        lib().orc_generate_ray(self._h, px, py, sample, int(use_jitter), seed, out.ctypes.data_as(_f32p))
        return out

    def object_intersect(self, obj, org, dir):
        (_, o), (_, d) = _v(org), _v(dir)
        t = C.c_float(0)
        hit = lib().orc_object_intersect(self._h, obj, o, d, C.byref(t))
        return t.value if hit else None

    def object_bbox(self, obj):
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        lib().orc_object_bbox(self._h, obj, mn.ctypes.data_as(_f32p), mx.ctypes.data_as(_f32p))
        return mn, mx

    def bvh_split(self, kind):
        l, r = C.c_uint32(0), C.c_uint32(0)
        lib().orc_bvh_split(self._h, kind, C.byref(l), C.byref(r))
        return l.value, r.value

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def aabb_intersect(mn, mx, org, dir):
    a, pa = _v(mn)
    b, pb = _v(mx)
    o, po = _v(org)
    d, pd = _v(dir)
    t = C.c_float(0)
    hit = lib().orc_aabb_intersect(pa, pb, po, pd, C.byref(t))
    return t.value if hit else None


def aabb_surface_area(mn, mx):
    a, pa = _v(mn)
    b, pb = _v(mx)
    return lib().orc_aabb_surface_area(pa, pb)


def ray_get_t(org, dir, point):
    o, po = _v(org)
    d, pd = _v(dir)
    p, pp = _v(point)
    return lib().orc_ray_get_t(po, pd, pp)


def reflect(v, n):
    a, pa = _v(v)
    b, pb = _v(n)
    out = np.zeros(3, np.float32)
    lib().orc_reflect(pa, pb, out.ctypes.data_as(_f32p))
    return out


def refract(v, n, idx):
    a, pa = _v(v)
    b, pb = _v(n)
    out = np.zeros(3, np.float32)
    lib().orc_refract(pa, pb, idx, out.ctypes.data_as(_f32p))
    return out


def component_wise_range(pts):
    p, pp = _v(pts)
    mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_component_wise_range(pp, p.reshape(-1, 3).shape[0], mn.ctypes.data_as(_f32p), mx.ctypes.data_as(_f32p))
    return mn, mx


def axis_scaling(m_rowmajor):
    m = np.ascontiguousarray(np.asarray(m_rowmajor, np.float32).T)  # column-major storage
    out = np.zeros(3, np.float32)
    lib().orc_axis_scaling(m.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p))
    return out


def clamp(x, lo, hi):
    return lib().orc_clamp(x, lo, hi)


def jitter_u32(seed, px, py, sample, dim):
    return lib().orc_jitter_u32(seed, px, py, sample, dim)
