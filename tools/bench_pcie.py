#!/usr/bin/env python3
"""PCIe-inclusive rates (never bench.py's `value`): frames start in host memory and keypoints / descriptors / uRight / depth end there.
  synchronous   hs_orb_extract_batch + hs_stereo_match per pair, pageable frames (what round 2 measured)
  pipelined     hs_orb_submit_batch / hs_orb_wait, two tickets in flight: H2D of batch i+1 under the kernels of batch i, D2H on a third stream;
                frames in page-locked memory (hs_host_alloc) and in pageable memory
The same measurement is part of bench.py's JSON line (`pcie_inclusive`)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hyslam_amd as HS  # noqa: E402
from hyslam_amd.synth import synth_stereo_pair  # noqa: E402

W, H = 1920, 1080


def pipelined(ex, sp, frames, seconds):
    """frames: [2B, H, W] (left frames first); returns (pairs/s, H2D GB/s).  Two tickets in flight."""
    B2 = len(frames)
    imgs = [frames[i] for i in range(B2)]
    outs = [None, None]
    t = [ex.submit_batch(imgs, sp), ex.submit_batch(imgs, sp)]
    outs[0] = ex.wait(t[0]); outs[1] = ex.wait(t[1])                       # warm-up: workspace, staging slots, result arrays
    t = [ex.submit_batch(imgs, sp), ex.submit_batch(imgs, sp)]
    n, k, t0 = 0, 0, time.perf_counter()
    while True:
        outs[k] = ex.wait(t[k], outs[k])
        n += 1
        if time.perf_counter() - t0 >= seconds and n >= 4:
            break
        t[k] = ex.submit_batch(imgs, sp)
        k ^= 1
    ex.wait(t[k ^ 1], outs[k ^ 1])
    n += 1
    dt = time.perf_counter() - t0
    return n * (B2 // 2) / dt, n * B2 * W * H / dt / 1e9, outs[0]


def main(pairs_per_call=16, seconds=1.5):
    B = pairs_per_call
    src = [synth_stereo_pair(1000 + i, W, H) for i in range(4)]
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000))
    cam = HS.Camera(1050.0, 126.0, 1080.0)
    sp = HS.stereo_params(cam)
    res = {"pairs_per_call": B}
    pin = ex.pinned_frames(2 * B, H, W)
    page = np.empty((2 * B, H, W), np.uint8)
    for i in range(B):
        pin[i], pin[B + i] = src[i % 4]
        page[i], page[B + i] = src[i % 4]
    for name, fr in (("pipelined_pinned", pin), ("pipelined_pageable", page)):
        v, gbs, out = pipelined(ex, sp, fr, seconds)
        res[name] = {"pairs_per_s": round(v, 1), "h2d_GBps": round(gbs, 2), "keypoints_left0": int(out[0][0]), "stereo_matches0": int((out[4][0, :out[0][0]] > 0).sum())}
    # the synchronous host API, as round 2 measured it
    frames = [page[i] for i in range(2 * B)]

    def step():
        ks, ds = ex.extract_batch(frames)
        for i in range(B):
            sm = HS.Stereomatcher(ks[i], ks[B + i], ds[i], ds[B + i], cam, extractor=ex)
            sm.computeStereoMatches()

    step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds or n < 2:
        step()
        n += 1
    dt = time.perf_counter() - t0
    res["synchronous_pageable"] = {"pairs_per_s": round(n * B / dt, 1), "h2d_GBps": round(n * 2 * B * W * H / dt / 1e9, 2)}
    res["note"] = "frames in host memory -> keypoints, descriptors, uRight, depth in host memory; python binding overhead included"
    return res


if __name__ == "__main__":
    print(json.dumps(main(int(sys.argv[1]) if len(sys.argv) > 1 else 16)))
