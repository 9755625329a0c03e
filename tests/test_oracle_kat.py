"""The reference's own 13 unit tests, restated 1:1 against the CPU oracle.

These are the only golden vectors the reference holds for the trace path
(src/bvh.rs:437-577, src/object.rs:323-407, src/utils.rs:66-108); they pin the oracle.
"""
import math

import numpy as np
import pytest


def _none_scene(rt, objects, xforms=None):
    return rt.ArrayScene(objects, xforms=xforms)


# ---------------------------------------------------------------- src/bvh.rs
def test_aabb_surface_area(orc):  # bvh.rs:437-443
    assert orc.aabb_surface_area((0, 0, 0), (1, 1, 2)) == 10.0


def test_aabb_intersect(orc):  # bvh.rs:445-492
    mn, mx = (-1.0, -1.0, 10.0), (1.0, 1.0, 11.0)
    assert orc.aabb_intersect(mn, mx, (0, 0, 0), (0, 0, 1)) is not None
    assert orc.aabb_intersect(mn, mx, (0.5, 0, 0), (0, 0, 1)) is not None
    assert orc.aabb_intersect(mn, mx, (-10.1, -10.0, 0.0), (10.1, 10.2, 10.3)) is not None
    assert orc.aabb_intersect(mn, mx, (0, 0, 10.5), (0, 0, 1)) is not None
    assert orc.aabb_intersect(mn, mx, (0, 0, 10.5), (1, 1, 1)) is not None
    assert orc.aabb_intersect(mn, mx, (1.5, 0, 0), (1.5, 0, 1)) is None
    mn, mx = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    assert orc.aabb_intersect(mn, mx, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)) is not None
    assert orc.aabb_intersect(mn, mx, (-0.5, -0.5, -0.5), (-0.5, 0.5, 0.5)) is None
    assert orc.aabb_intersect(mn, mx, (-1.0, 0.5, 0.5), (1, 0, 0)) == 1.0
    assert orc.aabb_intersect(mn, mx, (-0.5, -0.5, 0.5), (0.5, 0.5, 0.0)) == np.float32(1.0) / np.sqrt(np.float32(2.0))
    # ray grazes a corner
    assert orc.aabb_intersect(mn, mx, (-1.0, -1.0, 0.0), (1.0, 0.5, 0.0)) == orc.ray_get_t((-1, -1, 0), (1, 0.5, 0),
                                                                                        (1, 0, 0))
    # ray starts in the middle and shoots out
    assert orc.aabb_intersect(mn, mx, (0.5, 0.5, 0.5), (1, 0, 0)) == 0.5


BVH_OBJECTS = [("triangle", (0, 0, 1), (0, 1, 1), (1, 0, 1)),
               ("sphere", (0, 5, 1), 0.5),
               ("quad", (0, -5, 0), (1, -5, 0), (1, -5, -1), (0, -5, -1))]
BVH_RAYS = [((-1, 0, 0), (-1, 0, 1), False), ((0.1, 0.1, 0), (0, 0, 1), True), ((-0.1, 0.1, 0), (0, 0, 1), False),
            ((0, 5.25, 0), (0, 0, 1), True), ((0, 5.55, 0), (0, 0, 1), False), ((0, 0, 0), (0, -1, 0), True),
            ((0, 0, 0), (0.1, -5.0, -0.1), True), ((2, 0, 0), (0, -1, 0), False)]


def test_bvh_intersect(rt, orc):  # bvh.rs:494-540
    s = orc.Scene(_none_scene(rt, BVH_OBJECTS))
    rays = np.array([list(o) + list(d) for o, d, _ in BVH_RAYS], np.float32)
    obj, t = s.trace_rays(rays)
    for (o, d, want), got in zip(BVH_RAYS, obj):
        assert (got != 0xFFFFFFFF) == want, (o, d)


def test_bvh_split(rt, orc):  # bvh.rs:542-577
    objs = [("sphere", c, 1.0) for c in [(0, 0, 0), (1, 6, 1), (1, 7, 1), (1, 8, 1), (1, 9, 1)]]
    s = orc.Scene(_none_scene(rt, objs))
    assert s.bvh_split(0) == (2, 3)   # naive
    assert s.bvh_split(1) == (2, 3)   # by x axis
    assert s.bvh_split(2) == (1, 4)   # Basic
    assert s.bvh_split(3) == (1, 4)   # SAH


# ---------------------------------------------------------------- src/object.rs
def test_sphere(rt, orc):  # object.rs:323-331
    s = orc.Scene(_none_scene(rt, [("sphere", (1, 2, 3), 0.25)]))
    assert s.object_intersect(0, (0, 0, 0), (-1, 0, 0)) is None
    assert s.object_intersect(0, (0, 1, 0), (1, 1, 3)) is not None


def test_quad(rt, orc):  # object.rs:333-351
    s = orc.Scene(_none_scene(rt, [("quad", (-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1))]))
    assert s.object_intersect(0, (0, 1, 0), (0, -1, 0.5)) is not None
    assert s.object_intersect(0, (0, 1, 0), (0, 1, 0)) is None
    assert s.object_intersect(0, (0, 1, 0), (0, 0, 1)) is None
    assert s.object_intersect(0, (0, -1, 0), (0, 1, 0)) is None   # back face


def test_triangle(rt, orc):  # object.rs:353-377
    s = orc.Scene(_none_scene(rt, [("triangle", (0, 0, 0), (1, 0, 0), (0, 1, 0))]))
    assert s.object_intersect(0, (0.1, 0.1, 1.0), (0, 0, -1)) is not None
    assert s.object_intersect(0, (1, 1, -1), (0, 0, 1)) is None
    s = orc.Scene(_none_scene(rt, [("triangle", (0, 0, 0), (0, 1, 0), (1, 0, 0))]))
    assert s.object_intersect(0, (0.1, 0.1, 1.0), (0, 0, -1)) is None
    assert s.object_intersect(0, (1, 1, -1), (0, 0, 1)) is None


def test_sphere_bounding_box(rt, orc):  # object.rs:379-389
    s = orc.Scene(_none_scene(rt, [("sphere", (2, 0, 1), 1.5)]))
    a, b = s.object_bbox(0)
    assert np.linalg.norm(a - np.array([0.5, -1.5, -0.5])) < 1e-4
    assert np.linalg.norm(b - np.array([3.5, 1.5, 2.5])) < 1e-4


def test_quad_bounding_box(rt, orc):  # object.rs:391-407
    s = orc.Scene(_none_scene(rt, [("quad", (1, 1, 1), (3, 1, 1), (3, 1, 0), (1, 1, 0))]))
    a, b = s.object_bbox(0)
    assert np.linalg.norm(a - np.array([1.0, 1.0, 0.0])) < 1e-4
    assert np.linalg.norm(b - np.array([3.0, 1.0, 1.0])) < 1e-4


# ---------------------------------------------------------------- src/utils.rs
def test_clamp(orc):  # utils.rs:66-71
    assert orc.clamp(100.0, 0.0, 1.0) == 1.0
    assert orc.clamp(-100.0, 0.0, 1.0) == 0.0
    assert orc.clamp(0.5, 0.0, 1.0) == 0.5


def test_reflect(orc):  # utils.rs:73-78
    n = np.float32(math.sqrt(2.0) / 2.0) * np.array([-1.0, 1.0, 0.0], np.float32)
    r = orc.reflect((1, 0, 0), n)
    assert np.linalg.norm(r - np.array([0, 1, 0])) < 1e-5


def test_component_wise_range(orc):  # utils.rs:80-89
    mn, mx = orc.component_wise_range([(-1, 0, 1), (0, -1, 0), (1, 0, 0)])
    assert tuple(mn) == (-1, -1, 0) and tuple(mx) == (1, 0, 1)


def _rot(axis, deg):
    c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    m = np.eye(4)
    i, j = {"x": (1, 2), "y": (2, 0), "z": (0, 1)}[axis]
    m[i, i], m[j, j], m[i, j], m[j, i] = c, c, -s, s
    return m


def test_get_scaling(orc):  # utils.rs:91-108
    rotate = _rot("x", 120) @ _rot("y", 90) @ _rot("z", 45)
    translate = np.eye(4)
    translate[:3, 3] = (1.0, 0.5, -2.0)
    np.testing.assert_allclose(orc.axis_scaling(rotate @ translate), (1, 1, 1), atol=1e-6)
    scale = np.diag([1.0, 2.0, 3.0, 1.0])
    m = translate @ rotate @ scale @ np.linalg.inv(translate)
    np.testing.assert_allclose(orc.axis_scaling(m), (1, 2, 3), atol=1e-5)
    np.testing.assert_allclose(orc.axis_scaling(np.diag([10.0, -0.5, 1.0, 1.0])), (10, 0.5, 1), atol=1e-6)
