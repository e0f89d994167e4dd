"""The oracle (and the synthetic generator) against the committed golden vectors in tests/golden/ (made by tools/make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

import oracle
from hyslam_amd.synth import synth_image, synth_stereo_pair

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", ["c1_mono_640x480_1000", "imaging_800x600_1500_s14"])
def test_mono_golden(name):
    g = np.load(os.path.join(G, name + ".npz"))
    img = synth_image(int(g["seed"]), int(g["w"]), int(g["h"]))
    assert sha(img) == str(g["image_sha256"]), "synthetic generator drifted"
    k, d, dbg = oracle.extract(oracle.default_params(int(g["nfeat"]), float(g["scale"])), img, debug=True)
    assert k.tobytes() == g["keypoints"].tobytes()
    assert np.array_equal(d, g["descriptors"])
    assert dbg["n_candidates"].tolist() == g["n_candidates"].tolist() and dbg["n_selected"].tolist() == g["n_selected"].tolist()
    assert [sha(l) for l in dbg["pyramid"]] == g["pyramid_sha256"].tolist()
    assert [sha(l) for l in dbg["blurred"]] == g["blurred_sha256"].tolist()


def test_stereo_golden():
    g = np.load(os.path.join(G, "stereo_640x480_1000.npz"))
    L, R = synth_stereo_pair(int(g["seed"]), int(g["w"]), int(g["h"]))
    assert sha(L) == str(g["left_sha256"]) and sha(R) == str(g["right_sha256"])
    p = oracle.default_params(int(g["nfeat"]))
    sp = oracle.stereo_params(fx=float(g["fx"]), mbf=float(g["fx"]) * 0.12, n_rows=int(g["h"]))
    kL, dL, kR, dR, uR, depth = oracle.stereo_frontend(p, sp, L, R)           # the threaded harness path
    assert kL.tobytes() == g["kL"].tobytes() and kR.tobytes() == g["kR"].tobytes()
    assert np.array_equal(dL, g["dL"]) and np.array_equal(dR, g["dR"])
    assert np.array_equal(uR, g["uRight"]) and np.array_equal(depth, g["depth"])
    assert int((depth > 0).sum()) > 50
    # depth is mbf/disparity wherever a match survived
    m = depth > 0
    assert np.array_equal(depth[m], np.float32(float(g["fx"]) * 0.12) / (kL["x"][m] - uR[m]))


def test_committed_boundary_dump_is_what_the_oracle_computes():
    """tests/golden/oracle_boundaries.npz (inputs + the oracle's outputs at the four cv:: call sites, for tools/check_with_opencv.py on a machine that
    has OpenCV 3.4) must stay in step with the oracle: every array's sha256 is committed beside it and recomputed here"""
    import hashlib
    import json
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(g, "oracle_boundaries.sha256.json")))
    d = np.load(os.path.join(g, "oracle_boundaries.npz"))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert set(d.files) == set(meta["arrays_sha256"])
    for k in d.files:
        assert sha(d[k]) == meta["arrays_sha256"][k], k
    img, cell = d["image"], d["cell"]
    assert np.array_equal(oracle.fast(img, 20, True), d["fast_image"]) and np.array_equal(oracle.fast(cell, 20, True), d["fast_cell"])
    assert np.array_equal(oracle.resize_linear(img, 533, 400), d["resize_533x400"]) and np.array_equal(oracle.resize_linear(img, 457, 343), d["resize_457x343"])
    assert np.array_equal(oracle.gaussian_blur7(img), d["blur_default_taps"])
    assert np.array_equal(oracle.gaussian_blur7(img, [16, 34, 50, 56, 50, 34, 16]), d["blur_256sum_taps"])
    col = d["colour"]          # round 6: ImageProcessing::PreProcessImg
    assert np.array_equal(oracle.preprocess(col, True, 1.0), d["pre_rgb_1_0"]) and np.array_equal(oracle.preprocess(col, False, 0.5), d["pre_bgr_0_5"])
    assert np.array_equal(oracle.preprocess(col, True, 0.75), d["pre_rgb_0_75"]) and np.array_equal(oracle.preprocess(img, True, 0.5), d["pre_grey_0_5"])
    at = np.array([oracle.lib().hso_fast_atan2(float(a), float(b)) for a, b in zip(d["atan_y"][:2000], d["atan_x"][:2000])], np.float32)
    assert np.array_equal(at, d["atan_deg"][:2000])
