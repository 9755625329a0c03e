"""The multi-GPU exchange step behind the C ABI (rtx_comm_init / rtx_gather_image, include/rtx.h): RCCL all-gather
of the ranks' RGB8 tile buffers on the render slot's stream, then the de-interleave kernel.  A one-GPU box can only
run it with a communicator of ONE rank: that checks the linkage against librccl, the stream ordering behind an
asynchronous render, and the reordering kernel (tile edges that do not divide the image).  The reordering for N > 1
ranks runs here too, through rtx_assemble_image on a gathered buffer put together on one GPU (2..16 ranks); the
packing logic is also covered on the CPU by tests/test_distributed.py (gloo, world size 2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("width,tile", [(96, 32), (100, 16)])
def test_gather_image_one_rank(rt, width, tile):
    s = rt.HostScene("basic_suzanne", width)
    c = rt.Context(0)
    try:
        c.upload(s)
        c.comm_init(rt.comm_unique_id(), 0, 1)
        img, _ = c.render(spp=2, bounces=3, seed=6, part=(tile, 0, 1))
        nbytes = c.gather_bytes_per_rank(tile)
        assert nbytes == c.partition_pixels((tile, 0, 1)) * 3
        d_tiles, d_all, d_img = rt.DeviceBuffer(nbytes, fill=1), rt.DeviceBuffer(nbytes, fill=2), rt.DeviceBuffer(width * width * 3, fill=3)
        for slot in (0, 1):
            c.render_device(d_tiles.ptr, 2, 3, seed=6, part=(tile, 0, 1), stats=False, slot=slot)   # asynchronous
            c.gather_image(d_tiles.ptr, d_all.ptr, d_img.ptr, slot=slot, tile=tile)                 # behind it, same stream
            c.slot_synchronize(slot)
            got = d_img.to_host().reshape(width, width, 3)
            assert np.array_equal(got, img), f"slot {slot}: {(got != img).sum()} bytes differ"
            assert np.array_equal(d_all.to_host(), d_tiles.to_host())
    finally:
        c.close()


@pytest.mark.parametrize("width,tile,nranks", [(128, 32, 2), (100, 16, 3), (160, 32, 8), (96, 32, 16)])
def test_assemble_image_of_many_ranks(rt, width, tile, nranks):
    """What N ranks' all-gather leaves behind, produced on one GPU: every rank's share rendered into its stretch of
    the gathered buffer; the de-interleave kernel must turn it into the image a single rank renders (ragged tile
    counts, more ranks than tiles in a row, tile edges that do not divide the image)."""
    s = rt.HostScene("basic_suzanne", width)
    c = rt.Context(0)
    try:
        c.upload(s)
        img, _ = c.render(spp=2, bounces=3, seed=6)
        stride = c.partition_pixels((tile, 0, nranks)) * 3
        d_all = rt.DeviceBuffer(stride * nranks, fill=7)
        for r in range(nranks):
            share = c.partition_pixels((tile, r, nranks)) * 3
            assert share <= stride
            if share:
                c.render_device(d_all.ptr + r * stride, 2, 3, seed=6, part=(tile, r, nranks), stats=False, slot=r % c.slots)
        c.synchronize()
        d_img = rt.DeviceBuffer(width * width * 3, fill=3)
        c.assemble_image(d_all.ptr, d_img.ptr, nranks, slot=0, tile=tile)
        c.slot_synchronize(0)
        got = d_img.to_host().reshape(width, width, 3)
        assert np.array_equal(got, img), f"{(got != img).sum()} bytes differ"
    finally:
        c.close()


def test_gather_without_communicator_fails(rt, gpu):
    s = rt.HostScene("basic", 32)
    gpu.upload(s)
    b = rt.DeviceBuffer(32 * 32 * 3)
    with pytest.raises(rt.RtxError) as e:
        gpu.gather_image(b.ptr, b.ptr)
    assert e.value.status == rt.RTX_ERR_COMM


def _run_bench(args, env_extra=None, timeout=170):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(env_extra or {}))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_comm_info_names_the_linked_runtime(rt):
    """The library's RCCL and HIP runtime are the ones it is linked against (/opt/rocm), in this process."""
    info = rt.comm_info()
    assert info.startswith("RCCL ") and "HIP runtime" in info
    assert info.count("/opt/rocm") == 2, info


def test_bench_n_rank_path_runs_on_the_linked_rccl():
    """bench.py's own N-rank code path (rendezvous, rtx_comm_init, render + rtx_gather_image per step, the reductions)
    driven with a communicator of one rank: the process must not have torch's bundled runtime underneath it."""
    rc, line, err = _run_bench(["--gpus", "1", "--force-comm", "--scene", "basic_suzanne", "--width", "256", "--spp", "2", "--steps", "3",
                                "--warmup", "1", "--no-cpu-baseline", "--no-reference-visits"])
    assert rc == 0, err[-2000:]
    g = line["config"]["gather"]
    assert g.startswith("rtx_gather_image (library RCCL communicator)"), g
    assert g.count("/opt/rocm") == 2 and "torch" not in g, g
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["frames_in_flight"] == 1


def test_bench_fails_loudly_without_a_communicator():
    """Two ranks on ONE device: RCCL refuses the duplicate GPU.  The default is to fail (a host gather must not pass for an
    RCCL curve); --allow-host-gather runs, and the line says what it measured."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(29600 + os.getpid() % 300), os.path.join(root, "bench.py"), "--gpus", "2", "--one-device", "--scene", "basic",
            "--width", "128", "--spp", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-reference-visits"]
    p = subprocess.run(base, capture_output=True, text=True, timeout=170)
    assert p.returncode != 0, p.stdout[-500:]
    assert "RCCL communicator unavailable" in p.stderr
    p = subprocess.run(base + ["--allow-host-gather"], capture_output=True, text=True, timeout=170)
    assert p.returncode == 0, p.stderr[-2000:]
    import json
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["gather"].startswith("FALLBACK"), line["config"]["gather"]
    assert line["n_gpus"] == 2 and line["config"]["one_frame_in_flight"]["value"] > 0
