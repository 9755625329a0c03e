import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
rt = importlib.import_module("rust-raytracer_amd")
gpu = rt.Context(0)
s = rt.HostScene("scan", 64, 0)
gpu.upload(s)
rays = np.array([[-0.9290392398834229, -1.6776655912399292, -4.674737453460693, 0.00048351840814575553, 1.0938538312911987, 3.170393228530884]], np.float32)
print("prod", gpu.trace_rays(rays), "ref", gpu.trace_rays(rays, flags=rt.FLAG_REFERENCE_TRAVERSAL))
