"""The one output of the real Rust binary that the reference holds: media/scene.jpg (README.md:6).

It shows an older arrangement of the objects than scene.rs:11-108 builds today, but default_camera, the floor quad, its
texture (media/texture.png) and the lights are the same, so the floor's checker pattern and plus marks must land on the
same pixels in our render of load_basic.  tests/golden/reference_scene_floor_band.npy is the luminance of image rows
880..1023 of that picture (cut by tests/golden/make_reference_floor_fixture.py); here its high-passed structure is
correlated with ours: the correlation has to be high and has to PEAK at zero shift (a camera, look_at, quad
intersection or texture addressing that is off by a pixel or two moves the peak).  This pins camera.rs:32-47,
object.rs:167-189 and the uv / texture lookup of material.rs:53-76 against reference output -- loosely (JPEG, other
shadows), but against the real thing rather than against our own restatement."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = (880, 1024)


def _box(a, k):
    c = np.cumsum(np.cumsum(np.pad(a, ((k, k), (k, k)), mode="edge"), 0), 1)
    c = np.pad(c, ((1, 0), (1, 0)))
    n = 2 * k + 1
    return (c[n:, n:] - c[:-n, n:] - c[n:, :-n] + c[:-n, :-n]) / (n * n)


def _structure(lum):
    l = np.log1p(lum.astype(np.float64))
    return l - _box(l, 6)


def _ncc(a, b):
    a = a - a.mean()
    b = b - b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


def _check(img):
    ref = _structure(np.load(os.path.join(HERE, "golden", "reference_scene_floor_band.npy")))
    lum = np.rint(img.astype(np.float64) @ np.array([0.299, 0.587, 0.114]))
    ours = _structure(lum)  # high-pass on the whole image, then the band (so that a shift brings in real rows)
    m = 8
    at0 = _ncc(ref[:, m:-m], ours[ROWS[0]:ROWS[1], m:-m])
    assert at0 >= 0.72, f"floor structure correlates only {at0:.3f} with the reference's render"
    for dy, dx in [(0, 2), (0, -2), (2, 0), (-2, 0), (0, 4), (4, 0)]:
        shifted = np.roll(ours, (dy, dx), (0, 1))[ROWS[0]:ROWS[1], m:-m]
        s = _ncc(ref[:, m:-m], shifted)
        assert s < at0 - 0.15, f"shift {(dy, dx)} correlates {s:.3f}, zero shift {at0:.3f}: the peak is not at zero"
    # the light on the floor: mean colour of five patches that both arrangements of the objects leave alike (several
    # coloured lights with falloff on the textured floor, Phong: light.rs:58-79, material.rs:144-184, texture.png);
    # the JPEG and an ambient level that differs by 2-3 steps between the two versions of the scene are inside the bar
    import json
    pj = json.load(open(os.path.join(HERE, "golden", "reference_scene_floor_patches.json")))
    h = pj["half"]
    for (y, x), want in zip(pj["centres"], pj["mean_rgb"]):
        got = img[y - h:y + h, x - h:x + h].reshape(-1, 3).astype(np.float64).mean(0)
        assert np.abs(got - np.array(want)).max() <= 6.0, f"patch {(y, x)}: {np.round(got, 1)} against the reference's {np.round(want, 1)}"
    return at0


def test_oracle_floor_lands_on_the_reference_renders_pixels(rt, orc):
    s = rt.HostScene("basic", 1024, 0)
    o = orc.Scene(s)
    try:
        img, _ = o.render(spp=1, bounces=4, seed=1, threads=min(8, os.cpu_count() or 1), region=(0, 1024, 1, ROWS[0] - 16, 1024, 1))
    finally:
        o.close()
    _check(img)


@pytest.mark.gpu
def test_gpu_floor_lands_on_the_reference_renders_pixels(rt, gpu):
    s = rt.HostScene("basic", 1024, 0)
    gpu.upload(s)
    img, _ = gpu.render(spp=4, bounces=4, seed=1)
    _check(img)
