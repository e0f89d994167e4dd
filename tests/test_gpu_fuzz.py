"""Fixed-seed slices of the randomised parity sweeps (tools/fuzz_parity.py, tools/fuzz_matchers.py): random frame sizes, strides, feature
counts, scale factors, level counts and image statistics for the extractor and the stereo matcher; random scenes, radii, thresholds and
ratios for the FeatureMatcher cores.  Every case is compared bit-exactly with the oracle.  The tools run thousands of cases on demand."""
import os
import sys

import numpy as np
import pytest

import hyslam_amd as HS

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


def test_fuzz_extraction_and_stereo(gpu):
    import fuzz_parity
    rng = np.random.default_rng(20261003)
    compared = 0
    for i in range(120):
        msg, good = fuzz_parity.one_case(rng, i)
        assert good, msg
        compared += "keypoints ok" in msg
    assert compared >= 80                        # the rest are configurations the library refuses cleanly (aspect ratio, quota limit)


def test_fuzz_matchers(gpu):
    import fuzz_matchers
    rng = np.random.default_rng(20261004)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=500))
    for i in range(40):
        msg, good = fuzz_matchers.one_case(rng, i, ex)
        assert good, msg


def test_fuzz_stress_slice(gpu):
    """a slice of the stress mode (round 4): big / saturated / banded frames, tiny and huge quotas, cell edge, FAST threshold and blur taps drawn at random,
    several frames per call, the stereo front end through the ingest tickets on 1-17 pairs, the same handle reused on other sizes and content"""
    import fuzz_parity
    fuzz_parity.STRESS = True
    try:
        rng = np.random.default_rng(20261005)
        compared = extras = 0
        for i in range(60):
            msg, good = fuzz_parity.one_case(rng, i)
            assert good, msg
            compared += "keypoints ok" in msg
            extras += ("batch of" in msg) + ("front end" in msg) + ("handle reused" in msg)
        assert compared >= 35 and extras >= 10
    finally:
        fuzz_parity.STRESS = False


def test_fuzz_frame_records(gpu):
    """config 5's device path on random shapes: 1-8 frame records of random capacity and counts, 2-NN and vocabulary-grouped match of a random rank vs every peer"""
    import fuzz_matchers
    rng = np.random.default_rng(20261006)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=500))
    for i in range(40):
        msg, good = fuzz_matchers.records_case(rng, i, ex)
        assert good, msg
