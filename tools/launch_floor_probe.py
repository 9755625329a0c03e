"""Probe (run under rocprofv3 --kernel-trace): trace_closest launches over n = 1 .. 1M rays."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
ctx.upload(rt.HostScene("scan", 64, 0))
rng = np.random.default_rng(1)
for n in (1, 64, 4096, 65536, 1 << 20):
    org = np.tile(np.array([0, 1.5, 5], np.float32), (n, 1))
    d = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.7, 0.1, n), -np.ones(n)], axis=1).astype(np.float32)
    for _ in range(3):
        ctx.trace_rays(np.concatenate([org, d], axis=1))
    print("n", n)
