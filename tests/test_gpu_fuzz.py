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
