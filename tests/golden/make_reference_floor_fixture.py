"""Cuts the fixture tests/golden/reference_scene_floor_band.npy out of the reference's own render media/scene.jpg
(README.md:6 -- the only output of the real Rust binary that the reference holds).  Run where /root/reference exists:
    python tests/golden/make_reference_floor_fixture.py
The picture shows an older arrangement of the objects than scene.rs:11-108 builds today, but the camera
(default_camera), the floor quad, its texture and the lights are the same, so the checker pattern and the plus marks of
the floor land on the same pixels.  The fixture is the luminance (u8) of image rows 880..1023, the band in front of
every object; tests/test_reference_image.py correlates its high-passed structure with our renders of load_basic.
It also writes reference_scene_floor_patches.json: the mean RGB of five 16x16 patches of the floor (row 1000) that no
object of either arrangement shadows differently -- lit by the same coloured lights in both pictures."""
import json
import os
import numpy as np
from PIL import Image

ROWS = (880, 1024)
src = "/root/reference/media/scene.jpg"
a = np.asarray(Image.open(src).convert("RGB")).astype(np.float64)
lum = np.clip(np.rint(a @ np.array([0.299, 0.587, 0.114])), 0, 255).astype(np.uint8)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_scene_floor_band.npy")
np.save(out, lum[ROWS[0]:ROWS[1]])
print(out, lum[ROWS[0]:ROWS[1]].shape)

PATCHES = [(1000, 160), (1000, 208), (1000, 352), (1000, 400), (1000, 448)]  # (row, column) of the patch centres
means = [[float(v) for v in a[y - 8:y + 8, x - 8:x + 8].reshape(-1, 3).mean(0)] for y, x in PATCHES]
pj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_scene_floor_patches.json")
json.dump({"source": "media/scene.jpg", "half": 8, "centres": PATCHES, "mean_rgb": means}, open(pj, "w"), indent=1)
print(pj, means)
