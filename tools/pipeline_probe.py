"""Do frames overlap usefully?  Two contexts on one GPU render alternate frames without waiting for each other
(each frame complete: raygen, path, resolve, accumulate), against one context rendering them back to back.
Usage: pipeline_probe.py [nranks ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
a, b = rt.Context(0), rt.Context(0)
s = rt.HostScene("scan", 1024, 0)
a.upload(s)
b.upload(s)
K = 20
for n in [int(x) for x in sys.argv[1:]] or [1, 8]:
    part = (32, 0, n)
    ta, tb = rt.DeviceBuffer(a.partition_pixels(part) * 3), rt.DeviceBuffer(a.partition_pixels(part) * 3)
    for c, t in ((a, ta), (b, tb)):
        for _ in range(2):
            c.render_device(t.ptr, 16, 4, seed=1, part=part)
    t0 = time.perf_counter()
    for _ in range(K):
        a.render_device(ta.ptr, 16, 4, seed=1, part=part, stats=False)
    a.synchronize()
    one = (time.perf_counter() - t0) / K * 1e3
    t0 = time.perf_counter()
    for k in range(K):
        c, t = ((a, ta), (b, tb))[k & 1]
        c.render_device(t.ptr, 16, 4, seed=1, part=part, stats=False)
    a.synchronize()
    b.synchronize()
    two = (time.perf_counter() - t0) / K * 1e3
    print(f"1/{n} of the frame: one stream {one:.3f} ms/frame, two alternating {two:.3f} ms/frame", flush=True)
