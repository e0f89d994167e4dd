#!/usr/bin/env python3
"""Per-phase cycle totals of k_fast_rows (library built with `make -C hyslam_amd/csrc EXTRA=-DHS_FAST_PROFILE`)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import hyslam_amd as HS
from hyslam_amd.synth import synth_stereo_pair

L, R = synth_stereo_pair(1, 1920, 1080)
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000, fScaleFactor=1.2, nLevels=8))
imgs = [L, R] * (int(sys.argv[1]) if len(sys.argv) > 1 else 8)
ex.extract_batch(imgs)
lib = ex._lib
out = (C.c_ulonglong * 16)()
lib.hs_debug_fast_profile(out)
ex.extract_batch(imgs)
lib.hs_debug_fast_profile(out)
v = np.array(list(out), dtype=np.float64)
names = ["stage(wait loads + LDS writes)", "geometry + prefetch issue", "scan A total (incl. expand, corners)", "-", "  corners: segment test", "  corners: score",
         "NMS + emit", "counts + zero score", "items", "wave lifetime", "waves"]
items, waves = v[8], v[10]
print("items %d waves %d, cycles per item (avg) / share of wave lifetime" % (items, waves))
for i in (0, 1, 2, 4, 5, 6, 7):
    print("%-40s %9.0f  %5.1f %%" % (names[i], v[i] / items, 100 * v[i] / v[9]))
print("%-40s %9.0f" % ("wave lifetime per item", v[9] / items))
print("items that spilled corners (list overflow): %d of %d" % (v[3], items))
print("scan A row steps + expand (derived)      %9.0f  %5.1f %%" % ((v[2] - v[4] - v[5]) / items, 100 * (v[2] - v[4] - v[5]) / v[9]))
