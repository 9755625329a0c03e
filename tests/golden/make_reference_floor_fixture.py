"""Cuts the fixture tests/golden/reference_scene_floor_band.npy out of the reference's own render media/scene.jpg
(README.md:6 -- the only output of the real Rust binary that the reference holds).  Run where /root/reference exists:
    python tests/golden/make_reference_floor_fixture.py
The picture shows an older arrangement of the objects than scene.rs:11-108 builds today, but the camera
(default_camera), the floor quad, its texture and the lights are the same, so the checker pattern and the plus marks of
the floor land on the same pixels.  The fixture is the luminance (u8) of image rows 880..1023, the band in front of
every object; tests/test_reference_image.py correlates its high-passed structure with our renders of load_basic."""
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
