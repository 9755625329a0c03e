"""Generates the committed golden vectors from the CPU oracle (run once in the authoring
container: `python tests/golden/make_golden.py`).  The reference itself is Rust and cannot
be run here, and its own tests hold no image or hit-id fixtures, so these vectors are the
oracle's outputs (the oracle is pinned by the reference's 13 unit tests, test_oracle_kat.py).
They guard against drift of the oracle and give the GPU tests size-independent anchors.

Also writes the decoded reference texture (media/texture.png -> RGB8) with PIL as an
independent check of the hand-written PNG decoder."""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
rt = importlib.import_module("rust-raytracer_amd")
import orc  # noqa: E402

CASES = [("basic", 64, 0, 1), ("basic_suzanne", 64, 0, 1), ("basic_suzanne", 64, 0, 4), ("random_spheres", 64, 200, 1),
         ("scan", 64, 10000, 2)]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    meta = {}
    for name, width, param, spp in CASES:
        s = rt.HostScene(name, width, param)
        o = orc.Scene(s)
        key = f"{name}_{width}_{param}_spp{spp}"
        img, st = o.render(spp=spp, bounces=4, seed=1, threads=4)
        obj, t = o.trace_primary(sample=spp - 1, use_jitter=spp > 1, seed=1)
        leaf, pre, depth, sa = o.bvh_digest()
        np.savez_compressed(os.path.join(HERE, key + ".npz"), img=img, obj=obj, t=t)
        meta[key] = {"primary": int(st.primary_rays), "bounce": int(st.bounce_rays), "shadow": int(st.shadow_rays),
                     "bvh_depth": int(depth), "bvh_nodes": int(len(pre)), "leaf_order_sha256": digest(leaf),
                     "preorder_sizes_sha256": digest(pre), "total_sa": float(sa)}
    # BVH digests of larger inputs (structure only)
    for name, param in [("suzanne", 0), ("soup", 10000), ("scan", 100000)]:
        s = rt.HostScene(name, 32, param)
        leaf, pre, depth, sa = orc.Scene(s).bvh_digest()
        meta[f"bvh_{name}_{param}"] = {"bvh_depth": int(depth), "bvh_nodes": int(len(pre)), "leaf_order_sha256": digest(leaf),
                                       "preorder_sizes_sha256": digest(pre), "total_sa": float(sa)}
    meta["jitter_u32"] = {f"{seed},{px},{py},{s},{d}": int(orc.jitter_u32(seed, px, py, s, d))
                          for seed, px, py, s, d in [(0, 0, 0, 0, 0), (1, 2, 3, 4, 1), (2**40 + 7, 1023, 511, 15, 0), (1, 0, 0, 0, 1)]}
    json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    from PIL import Image
    tex = np.asarray(Image.open(os.path.join(ROOT, "assets", "texture.png")).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "texture_rgb.npz"), rgb=tex)
    print("wrote", len(meta), "entries")


if __name__ == "__main__":
    main()
