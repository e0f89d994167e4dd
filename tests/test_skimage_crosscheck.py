"""Cross-check of the oracle against scikit-image — an implementation independent of this repository and of OpenCV — for the parts of the
path it also implements: the FAST-9/16 segment test (corner set), cv::FAST's corner score (by its definition: the largest threshold at which
the pixel is still a corner), the intensity-centroid orientation over ORB's 31-px disc, the disc's half-width table and the 256 rBRIEF test
pairs.  scikit-image only exists in the image's conda python 3.9 (`/opt/conda/bin/python3.9`), so `tools/skimage_reference.py` runs there as
a subprocess; the test skips where that interpreter or the package is missing.  It does not pin cv::resize or cv::GaussianBlur (scikit-image's
versions use different arithmetic): for those `tests/test_opencv_crosscheck.py` needs an OpenCV."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from hyslam_amd.synth import synth_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY39 = "/opt/conda/bin/python3.9"


def _reference(tmp_path, img, kps, threshold=20, angles=None):
    if not os.path.exists(PY39):
        pytest.skip("no conda python 3.9 (scikit-image) in this image")
    src, dst = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    extra = {} if angles is None else {"kps_angle_deg": np.asarray(angles, np.float64)}
    np.savez(src, img=img, threshold=threshold, kps=np.asarray(kps, np.int64).reshape(-1, 2), **extra)
    r = subprocess.run([PY39, os.path.join(ROOT, "tools", "skimage_reference.py"), src, dst], capture_output=True, text=True)
    if r.returncode != 0:
        if "No module named" in r.stderr:
            pytest.skip("scikit-image not importable: " + r.stderr.strip().splitlines()[-1])
        raise AssertionError(r.stderr[-2000:])
    return np.load(dst)


@pytest.mark.parametrize("kind", ["scene", "noise", "blocks"])
def test_fast_corner_set_and_score_equal_skimage(tmp_path, kind):
    rng = np.random.default_rng(7)
    if kind == "scene":
        img = synth_image(5, 320, 240)
    elif kind == "noise":
        img = rng.integers(0, 256, (160, 200), dtype=np.uint8)
    else:                                        # flat blocks with exact-threshold steps: the strict inequalities of the segment test decide
        img = np.full((120, 160), 100, np.uint8)
        for k in range(60):
            x, y, s = int(rng.integers(4, 150)), int(rng.integers(4, 110)), int(rng.integers(2, 9))
            img[y:y + s, x:x + s] = 100 + int(rng.choice([19, 20, 21, -19, -20, -21, 40, -40]))
    ref = _reference(tmp_path, img, np.zeros((0, 2)))
    want = ref["fast_score"]
    got = np.zeros(want.shape, bool)
    c = oracle.fast(img, 20, nonmax=False)       # every pixel that passes the segment test (cv::FAST computes no score without NMS)
    got[c[:, 1], c[:, 0]] = True
    assert got.sum() > (50 if kind != "blocks" else 10)
    assert np.array_equal(got, want > 0), "corner sets differ at %d pixels" % int((got != (want > 0)).sum())
    # 3x3 strict non-max suppression on scikit-image's score map is a three-line numpy statement: it must give the oracle's nonmax=True
    # output, positions AND scores (a wrong score anywhere would change a survivor or its response)
    s = np.pad(want.astype(np.int32), 1)
    nb = np.stack([s[1 + dy:1 + dy + want.shape[0], 1 + dx:1 + dx + want.shape[1]] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dy or dx])
    keep = (want > 0) & (want > nb.max(0))
    k = oracle.fast(img, 20, nonmax=True)
    nm = np.zeros(want.shape, np.int32)
    nm[k[:, 1], k[:, 0]] = k[:, 2]
    assert np.array_equal(nm > 0, keep), "NMS survivors differ at %d pixels" % int(((nm > 0) != keep).sum())
    assert np.array_equal(nm[keep], want[keep].astype(np.int32)), "scores differ"


def test_orientation_disc_and_pattern_equal_skimage(tmp_path):
    img = synth_image(9, 640, 480)
    p = oracle.default_params(1500)
    kps, _ = oracle.extract(p, img)
    lvl0 = kps[kps["octave"] == 0]
    assert len(lvl0) > 250
    xy = np.stack([lvl0["x"], lvl0["y"]], 1).astype(np.int64)
    blurred = oracle.gaussian_blur7(img)         # the reference measures the angle on the blurred level (ORBExtractor.cpp:536-541)
    ref = _reference(tmp_path, blurred, xy, angles=lvl0["angle"])
    assert np.array_equal(ref["umax"], oracle.umax())
    assert np.array_equal(ref["pattern"].astype(np.int32).reshape(-1), np.asarray(oracle.pattern(), np.int32).reshape(-1))
    # scikit-image takes atan2 of the same moments in double; cv::fastAtan2 is a polynomial with 0.3 degrees of error
    d = np.abs(lvl0["angle"].astype(np.float64) - ref["angle_deg"])
    d = np.minimum(d, 360.0 - d)
    assert d.max() < 0.35, d.max()
    # steered BRIEF: scikit-image rotates the same pattern in double and rounds half away from zero, the reference in float with cvRound;
    # a sample position can differ by a pixel only when a rotated coordinate lies within ~1e-6 of a .5 boundary, so nearly every descriptor
    # must be bit-identical (same pattern order, same rotation sense, same comparison direction, bit i of byte j = test 8j + i)
    want = ref["brief_bits"].astype(np.uint8)
    _, desc = oracle.extract(p, img)
    got = np.unpackbits(desc[kps["octave"] == 0], axis=1, bitorder="little")
    assert got.shape == want.shape
    same = (got == want).all(axis=1)
    assert same.mean() > 0.97 and (got != want).mean() < 5e-4, (same.mean(), (got != want).mean())


def test_resize_and_blur_geometry_against_torch():
    """cv::resize(INTER_LINEAR) samples at half-pixel centres with edge clamping and cv::GaussianBlur(7x7, sigma 2) pads with
    BORDER_REFLECT_101: PyTorch's bilinear interpolation (align_corners=False) and reflect padding do the same in float.  OpenCV's 8-bit
    paths round through fixed point, so the comparison allows one grey level — it pins the geometry (which source pixels, which weights,
    which border rule), not the rounding."""
    import torch
    import torch.nn.functional as F
    img = synth_image(11, 322, 243)
    t = torch.from_numpy(img.astype(np.float32))[None, None]
    for dw, dh in ((268, 203), (224, 169), (161, 122)):      # scale 1.2, 1.44, 2.0 (the area-free bilinear range the reference uses)
        got = oracle.resize_linear(img, dw, dh).astype(np.int32)
        ref = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0, 0].numpy()
        assert np.abs(got - ref).max() <= 1.0 + 1e-3, (dw, dh, np.abs(got - ref).max())
        assert np.abs(got - ref).mean() < 0.3
    # blur: separable 7-tap filter, reflect-101 border, one rounding at the end.  With the oracle's own 8.8 fixed-point taps (deviation D3:
    # {18,34,49,55,49,34,18}/256, the ufixedpoint16 rounding of OpenCV 3.4.1-3.4.8) as float weights the float result rounds to the oracle's
    # bytes; against the ideal Gaussian cv::getGaussianKernel(7, 2) the same taps are a documented +0.4 % gain (they sum to 257/256)
    got = oracle.gaussian_blur7(img).astype(np.int32)
    p = F.pad(t.double(), (3, 3, 3, 3), mode="reflect")      # reflect = BORDER_REFLECT_101 (the edge pixel is not repeated)
    k = torch.tensor([18, 34, 49, 55, 49, 34, 18], dtype=torch.float64) / 256.0
    ref = F.conv2d(F.conv2d(p, k.view(1, 1, 1, 7)), k.view(1, 1, 7, 1))[0, 0].numpy()
    assert np.array_equal(got, np.minimum(np.floor(ref + 0.5), 255).astype(np.int32))
    g = np.exp(-((np.arange(7) - 3.0) ** 2) / 8.0)
    g = torch.from_numpy(g / g.sum())
    ideal = F.conv2d(F.conv2d(p, g.view(1, 1, 1, 7)), g.view(1, 1, 7, 1))[0, 0].numpy()
    assert np.abs(got - ideal).max() < 3.0 and np.abs(got - ideal).mean() < 1.5      # mean 1.0 on this frame: the 257/256 gain, twice
