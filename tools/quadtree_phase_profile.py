#!/usr/bin/env python3
"""Cycle stamps of the level-0 quadtree workgroup of image 0 (library built with `make -C hyslam_amd/csrc EXTRA=-DHS_QT_PROFILE`).
Tags: 1 gather done, 2 roots done, 10/110 processing order chosen (phase 1 / phase 2), 11 children counted, 12 cut chosen,
13 next list laid out, 14 points relabelled, 3 loop left, 4 selection written."""
import ctypes as C, sys
sys.path.insert(0, ".")
import hyslam_amd as HS
from hyslam_amd.synth import synth_stereo_pair
W, H, NF, SF = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 2000, 1.2)
L, R = synth_stereo_pair(1, W, H)
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF, fScaleFactor=SF, nLevels=8))
ex.extract_batch([L, R]); ex.extract_batch([L, R])
out = (C.c_ulonglong * 128)()
ex._lib.hs_debug_qt_profile(out)
k = int(out[127]); t0 = out[0]
prev = t0
for i in range(0, k, 2):
    print("tag %3d  +%7d  (total %7d)" % (out[i + 1], out[i] - prev, out[i] - t0)); prev = out[i]
