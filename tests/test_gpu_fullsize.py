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


def test_config3_headline_primaries_first(rt, gpu, monkeypatch):
    """The headline frame as a primaries-first launch (RTX_PRIMARIES_FIRST=1: 16.8 M primaries, 4,096 blocks of the
    classification passes, four of them per thread of the scan): the same bytes and ray counts as the single launch."""
    s = rt.HostScene("scan", 1024, 0)
    gpu.upload(s)
    ref, st = gpu.render(spp=16, bounces=4, seed=1)
    monkeypatch.setenv("RTX_PRIMARIES_FIRST", "1")
    c2 = rt.Context(0)
    try:
        c2.upload(s)
        img, st2 = c2.render(spp=16, bounces=4, seed=1)
        assert np.array_equal(img, ref), f"{(img != ref).any(axis=2).sum()} pixels differ"
        assert (st2.primary_rays, st2.bounce_rays, st2.shadow_rays) == (st.primary_rays, st.bounce_rays, st.shadow_rays)
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


def test_config4_at_its_stated_size(rt, orc, gpu):
    """BASELINE configs[3] as stated: the 2.37 M-triangle geometry with mirror + glass + Phong compositions, 4096^2,
    64 spp, 4 bounces -- 64 batches alternating between the slot's two lanes.  The production traversal and the
    on-device verbatim bvh.rs:359-395 must produce the same bytes and count the same rays; the oracle traces a
    stated subset of the frame (six 32x32 tiles: silhouettes of the meshes, the floor, the background) at all 64 samples."""
    W = 4096
    s = rt.HostScene("scan_glass", W, 0)
    info = gpu.upload(s)
    assert info.n_objects == 2374530
    img, st = gpu.render(spp=64, bounces=4, seed=1)
    assert st.batches == 64 and st.primary_rays == W * W * 64
    assert st.bounce_rays > 0.25 * st.primary_rays and st.shadow_rays > st.primary_rays
    img_r, st_r = gpu.render(spp=64, bounces=4, seed=1, flags=rt.FLAG_REFERENCE_TRAVERSAL)
    assert np.array_equal(img, img_r), f"{(img != img_r).any(axis=2).sum()} pixels differ between the production and the verbatim traversal"
    assert (st.primary_rays, st.bounce_rays, st.shadow_rays) == (st_r.primary_rays, st_r.bounce_rays, st_r.shadow_rays)
    del img_r
    o = orc.Scene(s)
    checked = 0
    for tx, ty in [(40, 60), (64, 70), (75, 66), (88, 72), (64, 120), (3, 3)]:   # tile column, tile row of the 128 x 128 tiles
        x0, y0 = tx * 32, ty * 32
        img_o, _ = o.render(spp=64, bounces=4, seed=1, threads=16, region=(x0, x0 + 32, 1, y0, y0 + 32, 1))
        a, b = img[y0:y0 + 32, x0:x0 + 32].astype(int), img_o[y0:y0 + 32, x0:x0 + 32].astype(int)
        close = (np.abs(a - b).max(axis=2) <= 1).mean()
        assert close >= 0.999, (tx, ty, close)
        checked += 1
    assert checked == 6
    o.close()


def test_config5_at_its_stated_size(rt, gpu, monkeypatch):
    """BASELINE configs[4] as stated: 20,000,000 random triangles, 2048^2, 16 spp, 4 point lights + ambient.  Size-independent
    properties (the oracle's sequential builder does not get this far): every primary hit spawns exactly 4 shadow queries
    and nothing bounces; the leaf-banking kernel the host picks for scenes this size and the default kernel produce the
    same bytes; the verbatim traversal produces the same bytes on rank 3's tiles of a 16-way partition."""
    W = 2048
    s = rt.HostScene("soup", W, 0)
    info = gpu.upload(s)
    assert info.n_objects == 20000000
    img, st = gpu.render(spp=16, bounces=4, seed=1)
    assert st.primary_rays == W * W * 16 and st.bounce_rays == 0
    hits = 0
    for k in range(16):   # the primary rays of the 16 jittered samples, one by one
        obj, _ = gpu.trace_primary(sample=k, use_jitter=True, seed=1)
        hits += int((obj != 0xFFFFFFFF).sum())
    assert st.shadow_rays == 4 * hits, (st.shadow_rays, hits)
    part = (64, 3, 16)
    a, sa = gpu.render(spp=16, bounces=4, seed=1, part=part)
    b, sb = gpu.render(spp=16, bounces=4, seed=1, part=part, flags=rt.FLAG_REFERENCE_TRAVERSAL)
    assert np.array_equal(a, b) and sa.rays == sb.rays
    mine = a.any(axis=2)
    assert mine.mean() > 0.03 and np.array_equal(a[mine], img[mine])   # the share is the frame's own pixels
    del a, b
    monkeypatch.setenv("RTX_LEAF_BANK", "0")    # read by rtx_init: this context runs the kernel without leaf banking
    c2 = rt.Context(0)
    try:
        c2.upload(s)
        img2, st2 = c2.render(spp=16, bounces=4, seed=1)
        assert np.array_equal(img2, img) and st2.rays == st.rays
    finally:
        c2.close()
