"""ImageProcessing::PreProcessImg (src/main/ImageProcessing.cpp:118-138) in the oracle (oracle/hs_oracle.cpp: preprocessImg) against the independent numpy
restatement (tests/pyref.py: preprocess) and against hand-computed known answers from OpenCV 3.4's published constants: cvtColor's fixed-point weights
(R2Y 4899, G2Y 9617, B2Y 1868 at 14 bits), the 2x2 rounded mean INTER_LINEAR silently becomes at scale 0.5, cvRound for the output size."""
import numpy as np
import pytest

import oracle
import pyref


def test_known_answers():
    # grey weights: pure channels and white (4899 + 9617 + 1868 = 16384: white stays 255, the three weights round to 76 / 150 / 29)
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 20, 30]]], np.uint8)
    assert oracle.preprocess(px, True, 1.0).tolist() == [[76, 150, 29, 255, (10 * 4899 + 20 * 9617 + 30 * 1868 + 8192) >> 14]]
    assert oracle.preprocess(px, False, 1.0).tolist() == [[29, 150, 76, 255, (30 * 4899 + 20 * 9617 + 10 * 1868 + 8192) >> 14]]
    # alpha is ignored
    rgba = np.concatenate([px, np.full((1, 5, 1), 77, np.uint8)], axis=2)
    assert np.array_equal(oracle.preprocess(rgba, True, 1.0), oracle.preprocess(px, True, 1.0))
    # scale 0.5 = INTER_AREA's fast path: (a + b + c + d + 2) >> 2 (a bilinear at the block centre would give the same mean WITHOUT the upward rounding of .5:
    # 0 + 0 + 1 + 1 -> 1 here, and 1 + 0 + 0 + 0 -> 0 but 1 + 1 + 0 + 0 -> 1)
    g = np.array([[0, 0, 1, 1, 3, 0], [1, 1, 0, 0, 0, 0]], np.uint8)
    assert oracle.preprocess(g, True, 0.5).tolist() == [[1, 1, 1]]
    # output size = cvRound(w * scale): half to even
    assert oracle.preprocess_size(2704, 2028, 0.5) == (1352, 1014)
    assert oracle.preprocess_size(5, 3, 0.5) == (2, 2) and oracle.preprocess_size(7, 7, 0.5) == (4, 4)
    # an odd width at 0.5 whose last block has one column: the float mean of what exists (5 / 2 = 2.5 -> 2, half to even), not (sum + 1) >> 1
    g7 = np.zeros((2, 7), np.uint8); g7[0, 6], g7[1, 6] = 2, 3
    assert oracle.preprocess(g7, True, 0.5)[0, 3] == 2
    # scale 1: the frame itself
    rng = np.random.default_rng(5)
    f = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(oracle.preprocess(f, True, 1.0), f)
    with pytest.raises(ValueError):
        oracle.preprocess(np.zeros((4, 4, 2), np.uint8), True, 1.0)
    with pytest.raises(ValueError):
        oracle.preprocess(np.zeros((1, 1), np.uint8), True, 0.25)          # cvRound(0.25) = 0: OpenCV asserts on an empty size


@pytest.mark.parametrize("cn", [1, 3, 4])
@pytest.mark.parametrize("scale", [1.0, 0.5, 0.75, 0.4, 0.25, 1.5, 0.3333])
def test_oracle_agrees_with_the_numpy_restatement(cn, scale):
    rng = np.random.default_rng(int(scale * 1000) + cn)
    for (w, h) in ((64, 48), (61, 47), (97, 33), (2, 2), (9, 4)):
        ow, oh = oracle.preprocess_size(w, h, scale)
        if ow < 1 or oh < 1:
            continue
        src = rng.integers(0, 256, (h, w) if cn == 1 else (h, w, cn), dtype=np.uint8)
        for rgb in (True, False):
            a, b = oracle.preprocess(src, rgb, scale), pyref.preprocess(src, rgb, scale)
            assert a.shape == b.shape == (oh, ow), (w, h, scale)
            assert np.array_equal(a, b), (w, h, cn, scale, rgb)
