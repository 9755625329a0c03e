"""Writes a scene file of objects the reference can hit outside their boxes (height-field quads
that are far from planar, spheres under rotation + non-uniform scale) for the exactness sweep:
    python tools/make_warped_scene.py OUT.scene [GRID] [SEED]"""
import sys
import numpy as np
out = sys.argv[1]
g = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 9)
h = rng.uniform(-0.4, 0.4, (g + 1, g + 1))
L = ["shader s phong 0.7 0.3 4", "texture t flat 0.8 0.7 0.3", "material m s t",
     "shader k mix 0.5 s 0.5 r", ] 
L = ["shader s phong 0.7 0.3 4", "shader r reflective", "shader k mix 0.6 s 0.4 r", "texture t flat 0.8 0.7 0.3",
     "texture u flat 0.3 0.5 0.9", "material m s t", "material q k u"]
sc = 8.0 / g
for layer in range(3):
    for i in range(g):
        for j in range(g):
            P = lambda a, b: ((a - g / 2) * sc, h[a, b] + 0.3 * layer - 0.5, (b - g / 2) * sc)
            a, b, c, d = P(i, j), P(i, j + 1), P(i + 1, j + 1), P(i + 1, j)
            coords = " ".join(f"{x:.7g}" for p in (a, d, c, b) for x in p)
            ops = "" if layer != 1 else " rotate_y 17 scale 1 1.3 0.8"
            L.append(f"quad {'m' if (i + j) % 3 else 'q'} {coords}{ops}")
for k in range(200):
    c = rng.uniform(-3, 3, 3)
    L.append(f"sphere q {c[0]:.6g} {c[1] * 0.3 + 1:.6g} {c[2]:.6g} {rng.uniform(0.05, 0.3):.5g} "
             f"rotate_z {rng.uniform(0, 90):.4g} scale {rng.uniform(0.3, 3):.4g} {rng.uniform(0.3, 3):.4g} 1 rotate_x {rng.uniform(0, 90):.4g}")
L += ["light point 2 5 1 1 1 1", "light directional -0.3 -1 -0.2 0.5 0.5 0.5", "light cone -2 4 0 0.4 -1 0 40 1 0.8 0.6", "light ambient 0.2 0.2 0.2"]
open(out, "w").write("\n".join(L) + "\n")
print(out, len(L), "lines")
