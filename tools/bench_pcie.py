#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (never bench.py's `value`): frames start and end in host memory.
hs_orb_extract_batch (H2D of the frames, kernels, D2H of keypoints + descriptors, synchronous) + hs_stereo_match per pair."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hyslam_amd as HS  # noqa: E402
from hyslam_amd.synth import synth_stereo_pair  # noqa: E402

W, H, B = 1920, 1080, 8
pairs = [synth_stereo_pair(1000 + i, W, H) for i in range(4)]
frames = [pairs[i % 4][0] for i in range(B)] + [pairs[i % 4][1] for i in range(B)]
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000))
cam = HS.Camera(1050.0, 126.0, 1080.0)


def step():
    ks, ds = ex.extract_batch(frames)
    for i in range(B):
        sm = HS.Stereomatcher(ks[i], ks[B + i], ds[i], ds[B + i], cam, extractor=ex)
        sm.computeStereoMatches()


for _ in range(3):
    step()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    step()
dt = time.perf_counter() - t0
print(json.dumps({"pcie_inclusive_pairs_per_s": round(n * B / dt, 1), "pairs_per_call": B, "ms_per_pair": round(dt / (n * B) * 1e3, 3),
                  "note": "pageable host buffers, synchronous host API, python binding overhead included"}))
