#!/usr/bin/env python3
"""Dumps the oracle's inputs and outputs at the four OpenCV boundaries of the path (cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2;
reference call sites ORBFinder.cpp:67,42, ORBExtractor.cpp:577,537) into one .npz, so that anyone with OpenCV 3.4 can diff them with
tools/check_with_opencv.py — the oracle is otherwise pinned only by source-derived known-answer tests ("parity unpinned").
usage: python tools/dump_boundaries.py [out.npz]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from hyslam_amd.synth import synth_image  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "oracle_boundaries.npz"
img = synth_image(11, 640, 480)
cell = np.ascontiguousarray(img[100:137, 200:237])
rng = np.random.default_rng(6)
ay = rng.integers(-200000, 200000, 20000).astype(np.float32)
ax = rng.integers(-200000, 200000, 20000).astype(np.float32)
# round 6: ImageProcessing::PreProcessImg (src/main/ImageProcessing.cpp:118-138): cv::resize by a scale factor on a colour frame + cvtColor to grey
col = np.ascontiguousarray(np.stack([synth_image(21 + 7 * k, 322, 241) for k in range(3)], axis=2))          # odd size: the 0.5 path has trailing partial blocks
np.savez_compressed(
    out, image=img, cell=cell, colour=col,
    pre_rgb_1_0=oracle.preprocess(col, True, 1.0), pre_bgr_0_5=oracle.preprocess(col, False, 0.5), pre_rgb_0_75=oracle.preprocess(col, True, 0.75),
    pre_grey_0_5=oracle.preprocess(img, True, 0.5),
    fast_image=oracle.fast(img, 20, True), fast_cell=oracle.fast(cell, 20, True),
    resize_533x400=oracle.resize_linear(img, 533, 400), resize_457x343=oracle.resize_linear(img, 457, 343),
    blur_default_taps=oracle.gaussian_blur7(img), blur_256sum_taps=oracle.gaussian_blur7(img, [16, 34, 50, 56, 50, 34, 16]),
    atan_y=ay, atan_x=ax, atan_deg=np.array([oracle.lib().hso_fast_atan2(float(a), float(b)) for a, b in zip(ay, ax)], np.float32))
print("wrote", out)
