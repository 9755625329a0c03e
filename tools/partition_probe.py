"""Probe: frame time of rank 0's share when the frame is split over N ranks (one GPU)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
s = rt.HostScene("scan", 1024, 0)
ctx.upload(s)
ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
for n in ns:
    part = (32, 0, n)
    tiles = rt.DeviceBuffer(ctx.partition_pixels(part) * 3)
    for _ in range(2):
        ctx.render_device(tiles.ptr, 16, 4, seed=1, part=part)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        st = ctx.render_device(tiles.ptr, 16, 4, seed=1, part=part)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    t0 = time.perf_counter()
    for _ in range(10):  # without statistics a render returns as soon as its launches are queued
        ctx.render_device(tiles.ptr, 16, 4, seed=1, part=part, stats=False)
    ctx.synchronize()
    dq = (time.perf_counter() - t0) / 10 * 1e3
    print(f"1/{n} of the frame: {dt:.3f} ms/frame, {st.rays/dt/1e3:.0f} Mrays/s per GPU, ideal {15.3/n:.3f}; queued back to back {dq:.3f} ms/frame")
