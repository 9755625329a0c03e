"""Probe: BVH build time (rtx_bvh_info.build_seconds) with and without GPU-assisted binning, alternating, same process."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("rust-raytracer_amd")
name, param, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 3
s = rt.HostScene(name, 64, param)
times = {"0": [], "1": []}
for r in range(reps):
    for mode in ("0", "1"):
        os.environ["RTX_GPU_BINNING"] = mode
        c = rt.Context(0)
        info = c.upload(s)
        times[mode].append(info.build_seconds)
        c.close()
n = info.n_objects
for mode in ("0", "1"):
    t = sorted(times[mode])
    print(f"{name} {n} objects, RTX_GPU_BINNING={mode}: build {min(t):.3f} s best, {t[len(t)//2]:.3f} s median of {reps}  ({' '.join(f'{x:.3f}' for x in times[mode])})", flush=True)
