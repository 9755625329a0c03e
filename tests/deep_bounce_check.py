"""Not collected: random scenes (test_gpu_fuzz) at 6-8 bounces, HIP path vs oracle; exercises the
wavefront capacity / batching logic with wide bounce trees."""
import importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
import test_gpu_fuzz as fz
rt = importlib.import_module("rust-raytracer_amd")
gpu = rt.Context(0)
d = tempfile.mkdtemp()
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    p = os.path.join(d, f"f{seed}.scene")
    open(p, "w").write(fz._scene_text(seed))
    s = rt.HostScene(f"file:{p}", 64)
    gpu.upload(s)
    o = orc.Scene(s)
    for bounces in (6, 8):
        img_o, st_o = o.render(spp=1, bounces=bounces, seed=seed)
        img_g, st_g = gpu.render(spp=1, bounces=bounces, seed=seed)
        dd = np.abs(img_g.astype(np.int32) - img_o.astype(np.int32)).max(axis=2)
        ok = (dd <= 1).mean() >= 0.999 and abs(int(st_g.rays) - int(st_o.rays)) <= max(8, st_o.rays // 2000)
        bad += not ok
        print(seed, bounces, "rays", st_g.rays, st_o.rays, "batches", st_g.batches, "within", round(float((dd <= 1).mean()), 5), "OK" if ok else "MISMATCH", flush=True)
print("BAD", bad)
sys.exit(1 if bad else 0)
