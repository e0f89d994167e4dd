#!/usr/bin/env python3
"""Run on a machine WITH OpenCV (ideally 3.4.x): compares a dump of tools/dump_boundaries.py with cv2 at the four boundaries.
usage: python tools/check_with_opencv.py oracle_boundaries.npz        (needs only numpy + cv2)"""
import sys

import cv2
import numpy as np

d = np.load(sys.argv[1])
img, cell = d["image"], d["cell"]
det = cv2.FastFeatureDetector_create(threshold=20, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
fast = lambda v: np.array([[int(k.pt[0]), int(k.pt[1]), int(k.response)] for k in det.detect(v, None)], np.int32).reshape(-1, 3)
res = {
    "FAST image": np.array_equal(fast(img), d["fast_image"]), "FAST 37x37 cell": np.array_equal(fast(cell), d["fast_cell"]),
    "resize 533x400": np.array_equal(cv2.resize(img, (533, 400), interpolation=cv2.INTER_LINEAR), d["resize_533x400"]),
    "resize 457x343": np.array_equal(cv2.resize(img, (457, 343), interpolation=cv2.INTER_LINEAR), d["resize_457x343"]),
    "GaussianBlur == default taps {18,34,49,55,49,34,18}": np.array_equal(cv2.GaussianBlur(img, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101), d["blur_default_taps"]),
    "GaussianBlur == 256-sum taps {16,34,50,56,50,34,16}": np.array_equal(cv2.GaussianBlur(img, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101), d["blur_256sum_taps"]),
    "fastAtan2": np.array_equal(np.array([cv2.fastAtan2(float(a), float(b)) for a, b in zip(d["atan_y"], d["atan_x"])], np.float32), d["atan_deg"]),
}
if "colour" in d:       # round 6: PreProcessImg = cv::resize(img, img, Size(), s, s) then cvtColor (src/main/ImageProcessing.cpp:118-138)
    col = d["colour"]
    def pre(f, rgb, s):
        f = cv2.resize(f, None, fx=s, fy=s)
        return f if f.ndim == 2 else cv2.cvtColor(f, cv2.COLOR_RGB2GRAY if rgb else cv2.COLOR_BGR2GRAY)
    res["PreProcessImg RGB, scale 1.0 (cvtColor weights)"] = np.array_equal(pre(col, True, 1.0), d["pre_rgb_1_0"])
    res["PreProcessImg BGR, scale 0.5 (INTER_AREA fast path, odd size)"] = np.array_equal(pre(col, False, 0.5), d["pre_bgr_0_5"])
    res["PreProcessImg RGB, scale 0.75 (bilinear per channel)"] = np.array_equal(pre(col, True, 0.75), d["pre_rgb_0_75"])
    res["PreProcessImg grey, scale 0.5"] = np.array_equal(pre(img, True, 0.5), d["pre_grey_0_5"])
print("OpenCV", cv2.__version__)
for k, v in res.items():
    print("%-55s %s" % (k, "MATCH" if v else "differs"))
