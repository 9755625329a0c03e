"""Lane-utilisation / wave-lifetime census of a diagnostic build (-DRTX_UTIL_STATS): RTX_LIB=<variant> RTX_UTIL_PRINT=1.
Usage: util_probe.py [--scene NAME --tris N --width W --spp S --bounces B] [nranks ...]
(rank 0's share of the frame split over nranks; default: the headline frame, whole)"""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="scan")
ap.add_argument("--tris", type=int, default=0)
ap.add_argument("--width", type=int, default=1024)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--bounces", type=int, default=4)
ap.add_argument("nranks", type=int, nargs="*")
a = ap.parse_args()
ctx = rt.Context(0)
s = rt.HostScene(a.scene, a.width, a.tris)
ctx.upload(s)
for n in a.nranks or [1]:
    part = (32, 0, n)
    tiles = rt.DeviceBuffer(ctx.partition_pixels(part) * 3)
    ctx.render_device(tiles.ptr, a.spp, a.bounces, seed=1, part=part)
    print(f"--- {a.scene} 1/{n}", file=sys.stderr, flush=True)
    st = ctx.render_device(tiles.ptr, a.spp, a.bounces, seed=1, part=part)
    print(f"1/{n}: path {st.path_kernel_ms:.3f} ms, resolve {st.resolve_ms:.3f} ms", file=sys.stderr, flush=True)
