"""BASELINE.json's full-size configurations on the MI355X, through the C ABI.

Each test combines
  * size-independent properties: the production traversal (wide nodes, ordered, pruned,
    per-lane refill) must give the same hit ids / t bits / image as the on-device verbatim
    restatement of bvh.rs:359-395 (RTX_FLAG_REFERENCE_TRAVERSAL); tile partitions and
    sample batching must not change a byte; ray counts obey the scene's light arithmetic;
  * the oracle on the same full-size frame where it finishes in seconds (configs 3 and 4
    at 2 spp, the soup at 2 M triangles); the 20 M-triangle soup is beyond the oracle's
    sequential builder and is covered by the properties alone.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config3_scan_2p37M_primary_hits(rt, orc, gpu):
    """Config 3 geometry (2,374,525 triangles + load_basic), 1024^2 primary rays."""
    s = rt.HostScene("scan", 1024, 0)
    info = gpu.upload(s)
    assert info.n_objects == 2374530
    obj, t = gpu.trace_primary()
    obj_r, t_r = gpu.trace_primary(flags=rt.FLAG_REFERENCE_TRAVERSAL)
    assert np.array_equal(obj, obj_r)
    assert np.array_equal(t.view(np.uint32), t_r.view(np.uint32))
    assert (obj != 0xFFFFFFFF).mean() > 0.5
    # the oracle on the same full frame (its reference traversal does ~13 Mrays/s on 16 threads)
    o = orc.Scene(s)
    leaf_g, pre_g = gpu.bvh_digest()
    leaf_o, pre_o, _, _ = o.bvh_digest()
    assert np.array_equal(leaf_g, leaf_o) and np.array_equal(pre_g, pre_o)
    obj_o, t_o = o.trace_primary(threads=16)
    assert np.array_equal(obj, obj_o)
    assert np.array_equal(t.view(np.uint32), t_o.view(np.uint32))
    # jittered sample 5 of the 16-spp frame
    obj_o, t_o = o.trace_primary(sample=5, use_jitter=True, seed=1, threads=16)
    obj_j, t_j = gpu.trace_primary(sample=5, use_jitter=True, seed=1)
    assert np.array_equal(obj_j, obj_o)
    assert np.array_equal(t_j.view(np.uint32), t_o.view(np.uint32))
    o.close()


def test_config3_scan_2p37M_image(rt, orc, gpu):
    """Config 3 at 1024^2, 4 bounces: full image and ray counts vs the oracle, partition
    invariance, reference traversal gives the same bytes."""
    s = rt.HostScene("scan", 1024, 0)
    gpu.upload(s)
    img, st = gpu.render(spp=2, bounces=4, seed=1)
    assert st.primary_rays == 1024 * 1024 * 2
    o = orc.Scene(s)
    img_o, st_o = o.render(spp=2, bounces=4, seed=1, threads=16)
    close = (np.abs(img.astype(int) - img_o.astype(int)).max(axis=2) <= 1).mean()
    assert close >= 0.999, close
    assert st.primary_rays == st_o.primary_rays
    assert abs(int(st.bounce_rays) - int(st_o.bounce_rays)) <= st_o.bounce_rays // 10000 + 2
    assert abs(int(st.shadow_rays) - int(st_o.shadow_rays)) <= st_o.shadow_rays // 10000 + 4
    o.close()
    # two ranks' tiles assemble to the same frame
    a, sa = gpu.render(spp=2, bounces=4, seed=1, part=(32, 0, 2))
    b, sb = gpu.render(spp=2, bounces=4, seed=1, part=(32, 1, 2))
    assert np.array_equal(np.maximum(a, b), img)
    assert sa.rays + sb.rays == st.rays
    # the verbatim traversal renders the same bytes and counts the same rays
    img_r, st_r = gpu.render(spp=2, bounces=4, seed=1, flags=rt.FLAG_REFERENCE_TRAVERSAL)
    assert np.array_equal(img_r, img) and st_r.rays == st.rays


def test_config3_headline_16spp_batching(rt, gpu, monkeypatch):
    """The headline frame (1024^2, 16 spp) in its default batches (one per pipe) and in forced smaller chunks."""
    s = rt.HostScene("scan", 1024, 0)
    gpu.upload(s)
    ref, st = gpu.render(spp=16, bounces=4, seed=1)
    assert st.batches <= 4 and st.primary_rays == 16 * 1024 * 1024
    monkeypatch.setenv("RTX_NODE_BUDGET_MB", "2048")
    c2 = rt.Context(0)
    try:
        c2.upload(s)
        img, st2 = c2.render(spp=16, bounces=4, seed=1)
        assert st2.batches > st.batches
        assert np.array_equal(img, ref) and st2.rays == st.rays
    finally:
        c2.close()


def test_config4_scan_glass_deep_bounces(rt, orc, gpu):
    """Config 4 materials (mirror + glass + Phong compositions: binary bounce tree) on the
    full geometry; 512^2 so the verbatim traversal finishes in seconds."""
    s = rt.HostScene("scan_glass", 512, 0)
    gpu.upload(s)
    img, st = gpu.render(spp=2, bounces=4, seed=3)
    img_r, st_r = gpu.render(spp=2, bounces=4, seed=3, flags=rt.FLAG_REFERENCE_TRAVERSAL)
    assert np.array_equal(img, img_r) and st.rays == st_r.rays
    assert st.bounce_rays > 0.25 * st.primary_rays   # mesh hits branch into mirror + glass rays
    o = orc.Scene(s)
    img_o, _ = o.render(spp=2, bounces=4, seed=3, threads=16)
    close = (np.abs(img.astype(int) - img_o.astype(int)).max(axis=2) <= 1).mean()
    assert close >= 0.999, close
    o.close()


def test_config5_soup_20M(rt, gpu):
    """Config 5: 20,000,000 random triangles (BVH far beyond the Infinity Cache), 2048^2,
    4 point lights + ambient: every hit spawns exactly 4 shadow queries, none bounces."""
    s = rt.HostScene("soup", 2048, 0)
    info = gpu.upload(s)
    assert info.n_objects == 20000000
    obj, t = gpu.trace_primary()
    obj_r, t_r = gpu.trace_primary(flags=rt.FLAG_REFERENCE_TRAVERSAL)
    assert np.array_equal(obj, obj_r)
    assert np.array_equal(t.view(np.uint32), t_r.view(np.uint32))
    hits = int((obj != 0xFFFFFFFF).sum())
    assert hits > 0.3 * obj.size
    img, st = gpu.render(spp=1, bounces=4)
    assert st.primary_rays == 2048 * 2048 and st.bounce_rays == 0
    assert st.shadow_rays == 4 * hits
    a, _ = gpu.render(spp=1, bounces=4, part=(64, 0, 2))
    b, _ = gpu.render(spp=1, bounces=4, part=(64, 1, 2))
    assert np.array_equal(np.maximum(a, b), img)


def test_soup_2M_vs_oracle(rt, orc, gpu):
    """The soup generator at a size the oracle still builds in seconds: hit ids + image."""
    s = rt.HostScene("soup", 512, 2000000)
    gpu.upload(s)
    o = orc.Scene(s)
    obj, t = gpu.trace_primary()
    obj_o, t_o = o.trace_primary(threads=16)
    assert np.array_equal(obj, obj_o)
    assert np.array_equal(t.view(np.uint32), t_o.view(np.uint32))
    img, st = gpu.render(spp=1, bounces=4)
    img_o, st_o = o.render(spp=1, bounces=4, threads=16)
    close = (np.abs(img.astype(int) - img_o.astype(int)).max(axis=2) <= 1).mean()
    assert close >= 0.999, close
    assert st.rays == st_o.primary_rays + st_o.bounce_rays + st_o.shadow_rays
    o.close()
