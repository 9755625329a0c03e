"""Probe: frame time of rank 0's share when the frame is split over N ranks (one GPU): one frame at a time
(synchronous), frames queued back to back in one render slot, and frames alternating between the two slots."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
s = rt.HostScene("scan", 1024, 0)
ctx.upload(s)
ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
K = 20
full = None
for n in ns:
    part = (32, 0, n)
    tiles = [rt.DeviceBuffer(ctx.partition_pixels(part) * 3) for _ in range(2)]
    for k in range(4):
        st = ctx.render_device(tiles[k & 1].ptr, 16, 4, seed=1, part=part, slot=k & 1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        st = ctx.render_device(tiles[0].ptr, 16, 4, seed=1, part=part)
    dt = (time.perf_counter() - t0) / K * 1e3
    t0 = time.perf_counter()
    for _ in range(K):  # without statistics a render returns as soon as its launches are queued
        ctx.render_device(tiles[0].ptr, 16, 4, seed=1, part=part, stats=False)
    ctx.synchronize()
    dq = (time.perf_counter() - t0) / K * 1e3
    t0 = time.perf_counter()
    for k in range(K):
        ctx.render_device(tiles[k & 1].ptr, 16, 4, seed=1, part=part, stats=False, slot=k & 1)
    ctx.synchronize()
    d2 = (time.perf_counter() - t0) / K * 1e3
    more = ""
    for S in range(3, ctx.slots + 1):  # RTX_PIPES=3/4: more frames in flight
        bufs = [rt.DeviceBuffer(ctx.partition_pixels(part) * 3) for _ in range(S)]
        for k in range(S):
            ctx.render_device(bufs[k].ptr, 16, 4, seed=1, part=part, stats=False, slot=k)
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            ctx.render_device(bufs[k % S].ptr, 16, 4, seed=1, part=part, stats=False, slot=k % S)
        ctx.synchronize()
        more += f", {S} slots {(time.perf_counter() - t0) / K * 1e3:.3f}"
    full = full or d2 * n
    print(f"1/{n} of the frame: {dt:.3f} ms/frame alone ({st.rays/dt/1e3:.0f} Mrays/s), queued in one slot {dq:.3f}, "
          f"alternating slots {d2:.3f} ms/frame = {full / n / d2 * 100:.0f} % of 1/{n} of the whole frame's time{more}", flush=True)
