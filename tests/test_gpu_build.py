"""GPU-assisted binning for the top of the BVH build (csrc/gpu_build.hip): rtx_upload_scene splits the
nodes at the top of a big tree with kernels -- range of the centroids, bucket counts and boxes, stable partition --
and leaves the pieces to the host threads.  The tree must be the host build's, bit for bit: same leaf order, same
preorder node sizes, same frame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _digest(rt, monkeypatch, scene, mode):
    monkeypatch.setenv("RTX_GPU_BINNING", mode)
    c = rt.Context(0)
    try:
        info = c.upload(scene)
        leaf, pre = c.bvh_digest()
        img, st = c.render(spp=1, bounces=2)
        return leaf.copy(), pre.copy(), info, img, st
    finally:
        c.close()


@pytest.mark.parametrize("name,width,param", [("soup", 64, 300000), ("scan", 64, 200000), ("random_spheres", 64, 5000),
                                              ("scan_glass", 64, 4100), ("basic_suzanne", 64, 0)])
def test_gpu_binning_builds_the_host_tree(rt, monkeypatch, name, width, param):
    """RTX_GPU_BINNING=1 (kernels from 4096 objects on) against =0 (host only) and against the host digest
    (rtx_host_bvh_digest, no GPU involved); basic_suzanne is below the threshold and must not care."""
    s = rt.HostScene(name, width, param)
    leaf0, pre0, info0, img0, st0 = _digest(rt, monkeypatch, s, "0")
    leaf1, pre1, info1, img1, st1 = _digest(rt, monkeypatch, s, "1")
    assert np.array_equal(leaf0, leaf1), f"{(leaf0 != leaf1).sum()} leaf-order entries differ"
    assert np.array_equal(pre0, pre1)
    assert (info0.n_internal, info0.n_leaves, info0.depth, info0.total_sa) == (info1.n_internal, info1.n_leaves, info1.depth, info1.total_sa)
    assert np.array_equal(img0, img1) and st0.rays == st1.rays
    leaf_h, pre_h, depth_h, _, _ = rt.host_bvh_digest(s, threads=4)
    assert np.array_equal(leaf_h, leaf1) and np.array_equal(pre_h, pre1) and depth_h == info1.depth


def test_gpu_binning_at_the_headline_size(rt, monkeypatch):
    """2,374,530 objects: the default (kernels from 2^20 objects on) against the host-only build."""
    s = rt.HostScene("scan", 256, 0)
    leaf0, pre0, info0, img0, _ = _digest(rt, monkeypatch, s, "0")
    monkeypatch.delenv("RTX_GPU_BINNING")
    c = rt.Context(0)
    try:
        info = c.upload(s)
        leaf, pre = c.bvh_digest()
        img, _ = c.render(spp=1, bounces=2)
    finally:
        c.close()
    assert np.array_equal(leaf0, leaf) and np.array_equal(pre0, pre) and info0.depth == info.depth
    assert np.array_equal(img0, img)


def test_gpu_binning_reports_degenerate_input(rt, monkeypatch):
    """Objects that no split separates: at the root (kernels) and inside a piece (host threads)."""
    monkeypatch.setenv("RTX_GPU_BINNING", "1")
    c = rt.Context(0)
    try:
        objs = [("sphere", (1, 2, 3), 0.5 + 0.0001 * i) for i in range(5000)]
        with pytest.raises(rt.RtxError) as e:
            c.upload(rt.ArrayScene(objs))
        assert e.value.status == rt.RTX_ERR_DEGENERATE
        objs = [("sphere", (float(i % 100), float(i // 100), 0.0), 0.3) for i in range(6000)] + \
               [("sphere", (7.25, 3.25, 1.0), 0.1 + 0.001 * i) for i in range(40)]
        with pytest.raises(rt.RtxError) as e:
            c.upload(rt.ArrayScene(objs))
        assert e.value.status == rt.RTX_ERR_DEGENERATE
        # the context is still good for a scene that builds
        c.upload(rt.HostScene("basic", 32, 0))
        img, _ = c.render(spp=1, bounces=1)
        assert img.any()
    finally:
        c.close()
