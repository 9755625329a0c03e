"""Probe: lane-utilisation census of the trace kernels' scheduling steps.  Needs a diagnostic
build of the library (tools/build_variant.sh util -DRTX_UTIL_STATS) named by RTX_LIB, and
RTX_UTIL_PRINT=1; the library prints the census of the frame to stderr."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
s = rt.HostScene(sys.argv[1] if len(sys.argv) > 1 else "scan", 1024, 0)
ctx.upload(s)
tiles = torch.zeros(ctx.partition_pixels(None) * 3, dtype=torch.uint8, device="cuda")
st = ctx.render_device(tiles.data_ptr(), 16, 4, seed=1)
ctx.synchronize()
print("rays", st.rays)
