"""Production vs on-device reference traversal for single rays (edit the list)."""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
rt = importlib.import_module("rust-raytracer_amd")
gpu = rt.Context(0)
s = rt.HostScene(sys.argv[1] if len(sys.argv) > 1 else "basic_suzanne", 32, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
gpu.upload(s)
base = np.array([3.764397621154785, 2.7982633113861084, 3.878384590148926, -1.036612507291374e-18, -6.29887024617226e-19, -9.127597881495748e-19])
for k in (1.0, 1e18, 1e10, 1e-10, 3.7e17):
    r = base.copy(); r[3:] *= k
    rays = r[None].astype(np.float32)
    d = rays[0, 3:]
    l2 = np.float32(d[0] * d[0]) + np.float32(d[1] * d[1]) + np.float32(d[2] * d[2])
    inv = np.float32(1.0) / np.sqrt(np.float32(l2))
    dn = d * inv
    print("scale", k, "l2", repr(float(l2)), "normalised", [repr(float(x)) for x in dn], "|dn|^2-1", float(np.dot(dn.astype(np.float64), dn.astype(np.float64)) - 1),
          "prod", gpu.trace_rays(rays), "ref", gpu.trace_rays(rays, flags=rt.FLAG_REFERENCE_TRAVERSAL))
