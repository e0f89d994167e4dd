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
np.savez_compressed(
    out, image=img, cell=cell,
    fast_image=oracle.fast(img, 20, True), fast_cell=oracle.fast(cell, 20, True),
    resize_533x400=oracle.resize_linear(img, 533, 400), resize_457x343=oracle.resize_linear(img, 457, 343),
    blur_default_taps=oracle.gaussian_blur7(img), blur_256sum_taps=oracle.gaussian_blur7(img, [16, 34, 50, 56, 50, 34, 16]),
    atan_y=ay, atan_x=ax, atan_deg=np.array([oracle.lib().hso_fast_atan2(float(a), float(b)) for a, b in zip(ay, ax)], np.float32))
print("wrote", out)
