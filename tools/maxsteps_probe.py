import importlib, sys
sys.path.insert(0, "/root/repo")
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
s = rt.HostScene("scan", 1024, 0)
ctx.upload(s)
for part in (None, (32, 0, 8)):
    img, st = ctx.render(spp=16, bounces=4, seed=1, part=part, flags=rt.FLAG_COUNT_VISITS)
    print(part, "rays", st.rays, "max records by one ray", st.max_ray_records, "avg", (st.closest_nodes+st.closest_prims+st.shadow_nodes+st.shadow_prims)/st.rays)
