"""Debug aid: the grazing-ray set of tests/test_gpu_parity.py on the small scan scene; prints
every ray on which production traversal, on-device reference traversal and oracle disagree."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # lives under tests/: it uses the oracle
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
rt = importlib.import_module("rust-raytracer_amd")
import orc
from test_gpu_parity import _world_vertices
gpu = rt.Context(0)
s = rt.HostScene("scan", 32, 20000)
gpu.upload(s)
o = orc.Scene(s)
vw = _world_vertices(s)
rng = np.random.default_rng(11)
n = 24000
tgt = vw[rng.integers(0, len(vw), n)]
org = rng.uniform(-6, 6, (n, 3))
org[n // 3:2 * n // 3] *= 300.0
k = 2 * n // 3
a = vw[(rng.integers(0, len(vw) // 3, n - k)) * 3]
b = vw[(rng.integers(0, len(vw) // 3, n - k)) * 3 + 1]
org[k:] = a - (b - a) * rng.uniform(0.5, 40.0, (n - k, 1))
tgt[k:] = b
m = n // 6
ax = rng.integers(0, 3, m)
off = rng.uniform(-5, 5, (m, 3))
off[np.arange(m), ax] = rng.uniform(-1e-3, 1e-3, m)
org[:m] = tgt[:m] - off
rays = np.concatenate([org, tgt - org], axis=1).astype(np.float32)
obj_o, t_o = o.trace_rays(rays)
obj_g, t_g = gpu.trace_rays(rays)
obj_r, t_r = gpu.trace_rays(rays, flags=rt.FLAG_REFERENCE_TRAVERSAL)
bad = np.nonzero((obj_g != obj_o) | (t_g.view(np.uint32) != t_o.view(np.uint32)) | (obj_r != obj_o))[0]
print("mismatching rays", len(bad))
for i in bad[:12]:
    print(i, rays[i].tolist(), "oracle", int(obj_o[i]), float(t_o[i]), "prod", int(obj_g[i]), float(t_g[i]), "ref", int(obj_r[i]), float(t_r[i]))
