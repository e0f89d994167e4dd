#!/usr/bin/env python3
"""Batch-1 anatomy of k_fast_rows (library built with `make -C hyslam_amd/csrc EXTRA=-DHS_FAST_WAVES`): at one stereo pair per call every
persistent workgroup gets at most one or two work items, so the launch lasts as long as its slowest wave.  Prints, per pyramid level, the
workgroups' start stamps and the phase durations of their FIRST item (tile staged / next prefetch issued / scan A / corners scored / NMS),
and the slowest waves.  usage: fast_b1_timeline.py [frames=2]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import hyslam_amd as HS
from hyslam_amd.synth import synth_stereo_pair

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L, R = synth_stereo_pair(1, 1920, 1080)
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000, fScaleFactor=1.2, nLevels=8))
imgs = [L, R] * (frames // 2)
ex.extract_batch(imgs)
ex.extract_batch(imgs)
out = (C.c_ulonglong * (4096 * 16))()
ex._lib.hs_debug_fast_waves(out)
raw = np.array(list(out), dtype=np.uint64).reshape(4096, 16)
keep = raw[:, 3] > 0
raw = raw[keep]
t0 = raw[:, 0].min()
us = lambda c: (c.astype(np.float64) - float(t0)) / 100.0
start, end, items = us(raw[:, 0]), us(raw[:, 2]), raw[:, 3].astype(int)
level = (raw[:, 8] >> np.uint64(32)).astype(int)
codes = ((raw[:, 8] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(int)
corners = (raw[:, 8] & np.uint64(0xFFFF)).astype(int)
ph = np.stack([us(raw[:, 8 + i]) for i in range(1, 6)], axis=1)      # staged, prefetched, scan A, scored, done
print("workgroups with work %d, items %d (max %d per workgroup), span %.1f us" % (len(raw), items.sum(), items.max(), end.max()))
print("start stamps: median %.2f, 90 %% %.2f, max %.2f us" % (np.median(start), np.percentile(start, 90), start.max()))
print("level  wgs  start(med/max)  stage  prefetch  scanA  corners  nms+emit  first item total (med / 90 %% / max)  codes(med/max)  corners(med/max)")
for l in sorted(set(level)):
    m = level == l
    d = np.diff(np.concatenate([start[m, None], ph[m]], axis=1), axis=1)
    tot = ph[m, 4] - start[m]
    print("%5d %4d  %5.1f / %5.1f   %5.1f  %5.1f   %5.1f   %5.1f   %5.1f     %5.1f / %5.1f / %5.1f      %4d / %4d   %4d / %4d" % (
        l, m.sum(), np.median(start[m]), start[m].max(), *np.median(d, axis=0), np.median(tot), np.percentile(tot, 90), tot.max(),
        np.median(codes[m]), codes[m].max(), np.median(corners[m]), corners[m].max()))
o = np.argsort(-end)[:12]
print("slowest waves: (end us, start us, items, level of first item, codes, corners, phases of the first item)")
for i in o:
    d = np.diff(np.concatenate([[start[i]], ph[i]]))
    print("  end %.1f start %.1f items %d level %d codes %d corners %d phases %s" % (end[i], start[i], items[i], level[i], codes[i], corners[i], np.round(d, 1)))
