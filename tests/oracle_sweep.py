"""Not collected by pytest: the adversarial ray families of tools/exactness_sweep.py against the
CPU ORACLE (the sweep tool itself compares the two device traversals).  Hit ids and t of the
production traversal, of the on-device verbatim traversal and of oracle/ must be bit-identical.
    python tests/oracle_sweep.py SCENE PARAM SEED REPS [RAYS_PER_REP]
The oracle visits both children of every node, so keep scenes and counts moderate."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import orc
import exactness_sweep as sw

rt = importlib.import_module("rust-raytracer_amd")


def main():
    scene, param, seed, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 18
    gpu = rt.Context(0)
    s = rt.HostScene(scene, 64, param)
    gpu.upload(s)
    o = orc.Scene(s)
    geo = sw.geometry(s)
    print(f"{scene} param {param}: {geo['N']} objects, extent {geo['ext']:.3g}", flush=True)
    rng = np.random.default_rng(seed)
    bad = total = 0
    for rep in range(reps):
        rays, fam, m = sw.make_rays(rng, geo, n)
        obj_o, t_o = o.trace_rays(rays)
        note = []
        for name, flags in (("production", 0), ("verbatim", rt.FLAG_REFERENCE_TRAVERSAL)):
            obj_g, t_g = gpu.trace_rays(rays, flags=flags)
            diff = (obj_g != obj_o) | (t_g.view(np.uint32) != t_o.view(np.uint32))
            bad += int(diff.sum())
            note.append(f"{name} != oracle: {int(diff.sum())}")
            for i in np.nonzero(diff)[0][:8]:
                print("   family", int(fam[i]), [repr(float(x)) for x in rays[i]], name, int(obj_g[i]), repr(float(t_g[i])),
                      "oracle", int(obj_o[i]), repr(float(t_o[i])))
        total += len(rays)
        print(f"rep {rep}: {len(rays)} rays, hits {float((obj_o != 0xFFFFFFFF).mean()):.2f}, " + ", ".join(note), flush=True)
    print(f"TOTAL {total} rays, MISMATCHES {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
