"""One-off check at full size: colours of World::trace_ray for rays aimed at vertices of the
2.37 M-triangle glass/mirror scene, pruned traversal vs verbatim traversal (second context
under RTX_REFERENCE_TRAVERSAL=1): must be bit-identical (covers shadow and bounce rays)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
rt = importlib.import_module("rust-raytracer_amd")
from test_gpu_parity import _world_vertices
name = sys.argv[1] if len(sys.argv) > 1 else "scan_glass"
s = rt.HostScene(name, 64, 0)
prod = rt.Context(0)
prod.upload(s)
os.environ["RTX_REFERENCE_TRAVERSAL"] = "1"
ref = rt.Context(0)
ref.upload(s)
vw = _world_vertices(s)
rng = np.random.default_rng(29)
bad = 0
for rep in range(4):
    n = 500000
    tgt = vw[rng.integers(0, len(vw), n)]
    org = rng.uniform(-6, 6, (n, 3))
    m = n // 3
    ax = rng.integers(0, 3, m)
    off = rng.uniform(-5, 5, (m, 3))
    off[np.arange(m), ax] = rng.uniform(-1e-3, 1e-3, m)
    org[:m] = tgt[:m] - off
    rays = np.concatenate([org, tgt - org], axis=1).astype(np.float32)
    a = prod.shade_rays(rays, depth=4)
    b = ref.shade_rays(rays, depth=4)
    diff = (a.view(np.uint32) != b.view(np.uint32)).any(axis=1)
    bad += int(diff.sum())
    print(rep, "rays", n, "lit", int((a.max(axis=1) > 0).sum()), "mismatches", int(diff.sum()), flush=True)
    for i in np.nonzero(diff)[0][:4]:
        print("  ray", rays[i].tolist(), "prod", a[i].tolist(), "ref", b[i].tolist())
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
