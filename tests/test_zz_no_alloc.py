"""rtx_render_device_noalloc (include/rtx.h): the entry point that allocates nothing.

Its wavefront state lives in the library's static device array instead of hipMalloc'ed buffers;
the result must be the bytes that rtx_render_device produces (bit-exact: RGB8 tiles), on two
calls in a row through the C ABI, the second with a smaller input -- the array keeps the first
call's contents, so anything that must start at zero has to be cleared by the call itself.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(rt, gpu, spp, bounces, part, seed=3):
    n = gpu.partition_pixels(part) * 3
    a, b = rt.DeviceBuffer(n, fill=7), rt.DeviceBuffer(n, fill=9)
    gpu.render_device(a.ptr, spp, bounces, seed=seed, part=part)  # with stats: synchronous
    gpu.render_device_noalloc(b.ptr, spp, bounces, seed=seed, part=part)
    gpu.synchronize()
    ha, hb = a.to_host(), b.to_host()
    a.free()
    b.free()
    return ha, hb


def test_noalloc_matches_render_device_twice(rt, gpu):
    s = rt.HostScene("basic_suzanne", 96)
    gpu.upload(s)
    # first call: the whole 96x96 frame, 4 spp, mirror + glass bounces
    a, b = _both(rt, gpu, 4, 4, (32, 0, 1))
    assert a.any(), "reference render is empty"
    assert np.array_equal(a, b), f"{(a != b).sum()} bytes differ on the first call"
    # second call, smaller input: one rank's share of a 2-way split, 1 spp (no jitter), 2 bounces
    a, b = _both(rt, gpu, 1, 2, (16, 1, 2))
    assert a.any()
    assert np.array_equal(a, b), f"{(a != b).sum()} bytes differ on the second (smaller) call"
    # and a third on another scene (other lights / materials / tree depth) after a re-upload
    s2 = rt.HostScene("scan_glass", 64, 8000)
    gpu.upload(s2)
    a, b = _both(rt, gpu, 2, 4, (32, 0, 1))
    assert np.array_equal(a, b), f"{(a != b).sum()} bytes differ after a scene change"


def test_noalloc_rejects_what_does_not_fit(rt, gpu):
    s = rt.HostScene("basic", 1024)
    gpu.upload(s)
    out = rt.DeviceBuffer(gpu.partition_pixels((32, 0, 1)) * 3)
    with pytest.raises(rt.RtxError) as e:
        gpu.render_device_noalloc(out.ptr, 16, 4, seed=1)
    assert e.value.status == rt.RTX_ERR_CAPACITY
    assert "static scratch" in str(e.value)


def test_noalloc_reports_a_pool_overflow(rt, monkeypatch):
    """The call itself cannot read the overflow flag back (it is asynchronous and allocates nothing):
    the flag travels through the pinned word of rtx_init and rtx_synchronize reports it.  Forced here by
    a context whose pool of bounce children is shrunk to one run (RTX_TEST_POOL_NODES, read by rtx_init)."""
    monkeypatch.setenv("RTX_TEST_POOL_NODES", "128")
    ctx = rt.Context(0)
    try:
        ctx.upload(rt.HostScene("scan_glass", 64, 8000))   # every mesh hit spawns a mirror and a glass ray
        out = rt.DeviceBuffer(ctx.partition_pixels((32, 0, 1)) * 3)
        ctx.render_device_noalloc(out.ptr, 2, 4, seed=3)    # returns RTX_OK: nothing is known yet
        with pytest.raises(rt.RtxError) as e:
            ctx.synchronize()
        assert e.value.status == rt.RTX_ERR_CAPACITY
        # the flag is consumed: the next synchronize is clean, and a render that fits reports nothing
        ctx.synchronize()
        ctx.render_device_noalloc(out.ptr, 1, 1, seed=3)    # one level: no bounce children at all
        ctx.synchronize()
        out.free()
    finally:
        ctx.close()
