#!/usr/bin/env python3
"""Feeds truncated and corrupted PNG / OBJ / PLY / scene files to the host parsers through the C ABI
(rtx_png_read_rgb8, rtx_scene_builtin("mesh:..."), rtx_scene_builtin("file:...")): every input must come back
as a clean error or a scene -- never a crash.  Meant to run under the sanitizer build (tools/run_sanitizers.sh);
no GPU call is made.  Usage: fuzz_parsers.py [cases per seed file, default 300]"""
import ctypes as C
import importlib
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("rust-raytracer_amd")
L = rt.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(20261111)


def ply_ascii():
    return (b"ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
            b"element face 2\nproperty list uchar int vertex_indices\nend_header\n"
            b"0 0 0\n1 0 0\n1 1 0\n0 1 0.3\n3 0 1 2\n4 0 1 2 3\n")


def ply_binary():
    head = (b"ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
            b"property uchar red\nelement face 1\nproperty list uchar uint vertex_indices\nend_header\n")
    body = b"".join(struct.pack("<fffB", *v) for v in ((0, 0, 0, 1), (1, 0, 0, 2), (0, 1, 0, 3))) + struct.pack("<BIII", 3, 0, 1, 2)
    return head + body


def obj_small():
    v = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    t = "".join(f"v {x} {y} {z}\n" for x, y, z in v) + "vt 0 0\nvt 1 0\nvn 0 0 1\n"
    t += "f 1/1/1 2/2/1 3/1/1 4/2/1\nf 5/1/1 8/1/1 7/2/1 6/2/1\nf 1/1/1 5/2/1 6/1/1\nf 2//1 6//1 7//1 3//1\nf 4 3 7 8\n"
    return t.encode()


def mutations(data, n):
    yield data
    yield b""
    for _ in range(n):
        kind = rng.integers(0, 5)
        b = bytearray(data)
        if kind == 0 and len(b) > 1:                       # truncate
            del b[int(rng.integers(0, len(b))):]
        elif kind == 1 and b:                              # flip bytes
            for _ in range(int(rng.integers(1, 9))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 2 and b:                              # splice garbage
            at = int(rng.integers(0, len(b)))
            b[at:at] = bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
        elif kind == 3 and b:                              # blow a number up (counts, sizes, indices)
            text = bytes(b)
            digits = [i for i, ch in enumerate(text) if 48 <= ch <= 57]
            if digits:
                at = digits[int(rng.integers(0, len(digits)))]
                b[at:at + 1] = b"4294967295" if rng.integers(0, 2) else b"-7"
        else:                                              # duplicate a chunk
            if len(b) > 4:
                a0 = int(rng.integers(0, len(b) - 2))
                a1 = int(rng.integers(a0 + 1, len(b)))
                b[a1:a1] = b[a0:a1]
        yield bytes(b)


def run(kind, name, data, tmp):
    path = os.path.join(tmp, name)
    total = ok = 0
    for m in mutations(data, N):
        open(path, "wb").write(m)
        total += 1
        if kind == "png":
            w, h, p = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
            if L.rtx_png_read_rgb8(path.encode(), C.byref(w), C.byref(h), C.byref(p)) == 0:
                ok += 1
                L.rtx_host_free(p)
        else:
            h = L.rtx_scene_builtin((kind + ":" + path).encode(), 16, rt.MEDIA_DIR.encode(), 0)
            if h:
                ok += 1
                d = L.rtx_scene_get_desc(h)
                assert d.contents.n_objects < (1 << 28)
                # the host half of an upload: BVH build + flatten of whatever the parser accepted
                if d.contents.n_objects:
                    leaf = np.zeros(d.contents.n_objects, np.uint32)
                    pre = np.zeros(2 * d.contents.n_objects + 1, np.uint32)
                    L.rtx_host_bvh_digest(d, 2, leaf.ctypes.data_as(C.POINTER(C.c_uint32)), pre.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          None, None, None)
                L.rtx_scene_free(h)
    print(f"{kind:5s} {name:24s} {total:5d} inputs, {ok:5d} accepted, {total - ok:5d} rejected cleanly", flush=True)


with tempfile.TemporaryDirectory() as tmp:
    run("png", "texture.png", open(os.path.join(rt.MEDIA_DIR, "texture.png"), "rb").read(), tmp)
    run("mesh", "small.obj", obj_small(), tmp)
    run("mesh", "ascii.ply", ply_ascii(), tmp)
    run("mesh", "binary.ply", ply_binary(), tmp)
    for f in sorted(os.listdir(os.path.join(rt.MEDIA_DIR, "scenes"))):
        run("file", f, open(os.path.join(rt.MEDIA_DIR, "scenes", f), "rb").read(), tmp)
print("fuzz done: no crash")
