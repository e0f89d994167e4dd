"""Opportunistic cross-check of the oracle's OpenCV restatements against a real OpenCV, where one is installed (SURVEY.md §8c last row).
OpenCV is NOT available in the build container nor on the GPU box, so these tests normally SKIP; they are the only lever that can move
the oracle off "parity unpinned": run `python -m pytest tests/test_opencv_crosscheck.py` on any machine with `cv2` (ideally 3.4.x, the
reference's version — CMakeLists.txt:32-35), or diff the dump of tools/dump_boundaries.py with tools/check_with_opencv.py there.

Call sites in the reference: cv::FAST src/features/low_level/ORBFinder.cpp:67, cv::resize src/features/ORBExtractor.cpp:577,
cv::GaussianBlur ORBExtractor.cpp:537, cv::fastAtan2 ORBFinder.cpp:42."""
import numpy as np
import pytest

import oracle
from hyslam_amd.synth import synth_image

cv2 = pytest.importorskip("cv2")


@pytest.fixture(scope="module")
def img():
    return synth_image(11, 640, 480)


def test_fast_matches_cv2(img):
    det = cv2.FastFeatureDetector_create(threshold=20, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
    for view in (img, img[100:137, 200:237], img[:7, :7], img[40:80, 50:62]):
        view = np.ascontiguousarray(view)
        kps = det.detect(view, None)
        got = np.array([[int(k.pt[0]), int(k.pt[1]), int(k.response)] for k in kps], np.int32).reshape(-1, 3)
        assert np.array_equal(got, oracle.fast(view, 20, True)), view.shape


def test_resize_matches_cv2(img):
    for (dw, dh) in ((533, 400), (444, 333), (457, 343), (320, 240)):
        if (dw, dh) == (320, 240):
            continue                                            # exact 2x: OpenCV switches to INTER_AREA; not on the path
        assert np.array_equal(cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LINEAR), oracle.resize_linear(img, dw, dh)), (dw, dh)


def test_gaussian_blur_matches_cv2(img):
    """The 8-bit Gaussian is OpenCV-version dependent (deviation D3): report WHICH tap set this OpenCV uses."""
    ref = cv2.GaussianBlur(img, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    candidates = {"3.4.1-3.4.x ufixedpoint16 (default)": None, "256-sum taps": [16, 34, 50, 56, 50, 34, 16]}
    hits = [name for name, taps in candidates.items() if np.array_equal(ref, oracle.gaussian_blur7(img, taps))]
    assert hits, "cv2 %s GaussianBlur matches none of the oracle's tap sets (IPP build?)" % cv2.__version__
    print("cv2", cv2.__version__, "GaussianBlur ==", hits)


def test_fast_atan2_matches_cv2():
    rng = np.random.default_rng(6)
    y = rng.integers(-200000, 200000, 20000).astype(np.float32)
    x = rng.integers(-200000, 200000, 20000).astype(np.float32)
    ref = np.array([cv2.fastAtan2(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    got = np.array([oracle.lib().hso_fast_atan2(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    assert np.array_equal(ref, got)


def test_preprocess_matches_cv2():
    """ImageProcessing::PreProcessImg (src/main/ImageProcessing.cpp:118-138): cv::resize(img, img, Size(), s, s) on the colour frame, then cvtColor to grey"""
    col = np.ascontiguousarray(np.stack([synth_image(21 + 7 * k, 322, 241) for k in range(3)], axis=2))
    for rgb in (True, False):
        for s in (1.0, 0.5, 0.75, 0.4):
            f = cv2.resize(col, None, fx=s, fy=s)
            ref = cv2.cvtColor(f, cv2.COLOR_RGB2GRAY if rgb else cv2.COLOR_BGR2GRAY)
            assert np.array_equal(ref, oracle.preprocess(col, rgb, s)), (rgb, s)
