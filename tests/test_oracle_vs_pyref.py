"""The C++ oracle against an independent numpy restatement of every primitive (tests/pyref.py)."""
import numpy as np
import pytest

import oracle
import pyref
from hyslam_amd.synth import synth_image


@pytest.fixture(scope="module")
def img():
    return synth_image(11, 320, 240)


def test_resize(img):
    for (dw, dh) in ((267, 200), (229, 171), (320, 240), (160, 121), (400, 300)):
        assert np.array_equal(oracle.resize_linear(img, dw, dh), pyref.resize_linear(img, dw, dh)), (dw, dh)


def test_fast_whole_image(img):
    for t in (20, 7, 60):
        for nms in (True, False):
            a = oracle.fast(img, t, nms)
            b = pyref.fast(img, t, nms)
            assert len(a) > 0 or t == 60
            if not nms:      # cv::FAST only scores corners when it suppresses; response stays 0 otherwise
                a, b = a[:, :2], b[:, :2]
            assert np.array_equal(a, b), (t, nms, len(a), len(b))


def test_fast_small_views():
    rng = np.random.default_rng(3)
    for (h, w) in ((7, 7), (6, 40), (9, 8), (37, 37), (40, 12)):
        v = rng.integers(0, 256, (h, w), dtype=np.uint8)
        assert np.array_equal(oracle.fast(v, 20, True), pyref.fast(v, 20, True)), (h, w)


def test_gaussian_blur(img):
    assert np.array_equal(oracle.gaussian_blur7(img), pyref.gaussian_blur7(img))
    small = img[:9, :11]
    assert np.array_equal(oracle.gaussian_blur7(small), pyref.gaussian_blur7(small))
    taps = [16, 34, 50, 56, 50, 34, 16]      # a 256-sum variant (later OpenCV releases)
    assert np.array_equal(oracle.gaussian_blur7(img, taps), pyref.gaussian_blur7(img, taps))
    flat = np.full((40, 40), 255, np.uint8)
    assert (oracle.gaussian_blur7(flat) == 255).all()       # 257/256 gain saturates, never wraps


def test_angle_and_descriptor(img):
    bl = oracle.gaussian_blur7(img)
    pat = oracle.pattern()
    rng = np.random.default_rng(5)
    for _ in range(60):
        x, y = int(rng.integers(19, 320 - 19)), int(rng.integers(19, 240 - 19))
        a = oracle.ic_angle(bl, x, y)
        assert np.float32(a) == pyref.ic_angle(bl, x, y)
        assert np.array_equal(oracle.orb_descriptor(bl, x, y, a), pyref.orb_descriptor(bl, x, y, a, pat))


def test_fast_atan2_bit_exact():
    rng = np.random.default_rng(6)
    for _ in range(3000):
        y, x = (np.float32(v) for v in rng.integers(-300000, 300000, 2))
        assert np.float32(oracle.lib().hso_fast_atan2(float(y), float(x))) == pyref.fast_atan2(y, x)


def test_octtree_against_python_restatement():
    rng = np.random.default_rng(7)
    for trial in range(40):
        W, Hh = int(rng.integers(60, 700)), int(rng.integers(60, 400))
        if W / Hh < 0.5:
            continue
        n = int(rng.integers(1, 900))
        clustered = trial % 3 == 0
        xs = rng.integers(3, W - 3, n) if not clustered else np.clip(rng.normal(W / 2, W / 12, n).astype(int), 3, W - 4)
        ys = rng.integers(3, Hh - 3, n) if not clustered else np.clip(rng.normal(Hh / 2, Hh / 12, n).astype(int), 3, Hh - 4)
        pts = np.unique(np.stack([xs, ys], 1), axis=0)
        rng.shuffle(pts)
        resp = rng.integers(19, 60, len(pts))               # few distinct values: exercises response ties
        c = np.concatenate([pts, resp[:, None]], 1).astype(np.float32)
        N = int(rng.integers(1, 400))
        a = oracle.distribute_octtree(c, 16, 16 + W, 16, 16 + Hh, N).tolist()
        b = pyref.distribute_octtree(c.tolist(), 16, 16 + W, 16, 16 + Hh, N)
        assert a == b, (trial, W, Hh, len(pts), N)
