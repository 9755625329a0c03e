"""One warm frame of rank 0's share with the frame split over N ranks (for kernel traces)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
s = rt.HostScene("scan", 1024, 0)
ctx.upload(s)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
part = (32, 0, n)
tiles = rt.DeviceBuffer(ctx.partition_pixels(part) * 3)
for _ in range(3):
    ctx.render_device(tiles.ptr, 16, 4, seed=1, part=part)
ctx.synchronize()
