import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
s = rt.HostScene("soup", 2048, 20000000)
ctx.upload(s)
tiles = torch.zeros(ctx.partition_pixels(None) * 3, dtype=torch.uint8, device="cuda")
ctx.render_device(tiles.data_ptr(), 16, 4, seed=1)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    st = ctx.render_device(tiles.data_ptr(), 16, 4, seed=1)
ctx.synchronize()
dt = (time.perf_counter() - t0) / 2 * 1e3
print(f"soup: {dt:.1f} ms/frame, {st.rays/dt/1e3:.0f} Mrays/s")
