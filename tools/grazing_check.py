"""One-off check at full size: production traversal == on-device reference traversal (bit for
bit) for millions of rays aimed at vertices / along edges of the 2.37 M-triangle scene."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
rt = importlib.import_module("rust-raytracer_amd")
from test_gpu_parity import _world_vertices
gpu = rt.Context(0)
s = rt.HostScene("scan", 64, 0)
gpu.upload(s)
vw = _world_vertices(s)
print("vertices", len(vw))
rng = np.random.default_rng(17)
bad = 0
for rep in range(8):
    n = 1000000
    tgt = vw[rng.integers(0, len(vw), n)]
    org = rng.uniform(-6, 6, (n, 3))
    org[n // 3:2 * n // 3] *= 10.0 ** rng.uniform(1, 4)
    k = 2 * n // 3
    a = vw[(rng.integers(0, len(vw) // 3, n - k)) * 3]
    b = vw[(rng.integers(0, len(vw) // 3, n - k)) * 3 + 1]
    org[k:] = a - (b - a) * rng.uniform(0.5, 4000.0, (n - k, 1))
    tgt[k:] = b
    m = n // 6  # nearly axis-parallel rays through vertices
    ax = rng.integers(0, 3, m)
    off = rng.uniform(-5, 5, (m, 3))
    off[np.arange(m), ax] = rng.uniform(-1e-3, 1e-3, m)
    org[:m] = tgt[:m] - off
    rays = np.concatenate([org, tgt - org], axis=1).astype(np.float32)
    obj_g, t_g = gpu.trace_rays(rays)
    obj_r, t_r = gpu.trace_rays(rays, flags=rt.FLAG_REFERENCE_TRAVERSAL)
    diff = (obj_g != obj_r) | (t_g.view(np.uint32) != t_r.view(np.uint32))
    bad += int(diff.sum())
    for i in np.nonzero(diff)[0][:8]:
        print("  ray", repr(rays[i].tolist()), "prod", int(obj_g[i]), float(t_g[i]), hex(int(t_g[i:i+1].view(np.uint32)[0])),
              "ref", int(obj_r[i]), float(t_r[i]), hex(int(t_r[i:i+1].view(np.uint32)[0])))
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/grazing_mismatch.txt", "a") as f:
            f.write(" ".join(repr(float(x)) for x in rays[i]) + "\n")
    print(rep, "rays", n, "hits", int((obj_r != 0xFFFFFFFF).sum()), "mismatches", int(diff.sum()), flush=True)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
