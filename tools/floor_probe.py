"""What does not shrink with the share: rank 0's part of the headline frame for small shares and bounce depths.
Usage: floor_probe.py [scene [width [param]]]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "scan"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
param = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = rt.Context(0)
s = rt.HostScene(name, width, param)
ctx.upload(s)
for n in (8, 64, 512):
    part = (32, 0, n)
    tiles = rt.DeviceBuffer(ctx.partition_pixels(part) * 3)
    for b in (1, 2, 4):
        for _ in range(2):
            ctx.render_device(tiles.ptr, 16, b, seed=1, part=part)
        ts = []
        for _ in range(5):
            st = ctx.render_device(tiles.ptr, 16, b, seed=1, part=part, flags=0)
            ts.append((st.path_kernel_ms, st.resolve_ms))
        st = ctx.render_device(tiles.ptr, 16, b, seed=1, part=part, flags=rt.FLAG_COUNT_VISITS)
        p = sorted(t[0] for t in ts)[2]
        print(f"1/{n} share, {b} bounces: path {p:.3f} ms, resolve {sorted(t[1] for t in ts)[2]:.3f} ms, {st.rays} rays, "
              f"longest ray {st.max_ray_records} records", flush=True)
