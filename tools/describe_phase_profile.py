#!/usr/bin/env python3
"""Per-phase cycle totals of k_describe (library built with `make -C hyslam_amd/csrc EXTRA=-DHS_DESC_PROFILE`)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import hyslam_amd as HS
from hyslam_amd.synth import synth_stereo_pair

L, R = synth_stereo_pair(1, 1920, 1080)
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000, fScaleFactor=1.2, nLevels=8))
imgs = [L, R] * 16
ex.extract_batch(imgs)
lib = ex._lib
W = 1 << 17
out = np.zeros(W * 8, np.uint32)
lib.hs_debug_describe_profile(out.ctypes.data_as(C.c_void_p), W)
ex.extract_batch(imgs)
lib.hs_debug_describe_profile(out.ctypes.data_as(C.c_void_p), W)
o = out.reshape(W, 8).astype(np.float64)
o = o[o[:, 7] > 0]
v = o.sum(0)
names = ["header: which keypoint, level, record (scalar loads)", "patch fetch -> LDS", "row pass", "row + column pass", "moments + angle", "sincos + 256 tests + stores", "wave lifetime"]
n = v[7]
print("keypoints %d; ticks per keypoint (avg) / share of wave lifetime" % n)
for i in range(7):
    print("%-55s %9.0f  %5.1f %%" % (names[i], v[i] / n, 100 * v[i] / v[6]))
