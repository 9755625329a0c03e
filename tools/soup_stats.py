"""Visit statistics of one soup frame (COUNT pass): totals and the longest ray."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
s = rt.HostScene("soup", 2048, 20000000)
ctx.upload(s)
tiles = rt.DeviceBuffer(ctx.partition_pixels(None) * 3)
st = ctx.render_device(tiles.ptr, 16, 4, seed=1, flags=rt.FLAG_COUNT_VISITS)
print("rays", st.rays, "closest n/p", st.closest_nodes, st.closest_prims, "shadow n/p", st.shadow_nodes, st.shadow_prims,
      "max records by one ray", st.max_ray_records, "path ms", round(st.path_kernel_ms, 1), "resolve ms", round(st.resolve_ms, 1))
