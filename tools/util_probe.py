"""Lane-utilisation / wave-lifetime census of a diagnostic build (-DRTX_UTIL_STATS): RTX_LIB=<variant> RTX_UTIL_PRINT=1.
Usage: util_probe.py [nranks ...]   (rank 0's share of the headline frame split over nranks)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
s = rt.HostScene("scan", 1024, 0)
ctx.upload(s)
for n in [int(a) for a in sys.argv[1:]] or [1]:
    part = (32, 0, n)
    tiles = rt.DeviceBuffer(ctx.partition_pixels(part) * 3)
    ctx.render_device(tiles.ptr, 16, 4, seed=1, part=part)
    print(f"--- 1/{n}", file=sys.stderr, flush=True)
    st = ctx.render_device(tiles.ptr, 16, 4, seed=1, part=part)
    print(f"1/{n}: path {st.path_kernel_ms:.3f} ms, resolve {st.resolve_ms:.3f} ms", file=sys.stderr, flush=True)
