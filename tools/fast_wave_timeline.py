#!/usr/bin/env python3
"""Start / end stamps of every persistent workgroup of k_fast_rows (library built with `make -C hyslam_amd/csrc EXTRA=-DHS_FAST_WAVES`):
how long the launch ramps up, how the workgroups' ends spread (the tail), items per workgroup.  usage: fast_wave_timeline.py [frames]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import hyslam_amd as HS
from hyslam_amd.synth import synth_stereo_pair

L, R = synth_stereo_pair(1, 1920, 1080)
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000, fScaleFactor=1.2, nLevels=8))
imgs = [L, R] * ((int(sys.argv[1]) if len(sys.argv) > 1 else 32) // 2)
ex.extract_batch(imgs)
ex.extract_batch(imgs)
lib = ex._lib
out = (C.c_ulonglong * (4096 * 16))()       # 16 slots per workgroup since round 4 (the first item's phase stamps follow the six summary slots)
lib.hs_debug_fast_waves(out)
raw = np.array(list(out), dtype=np.uint64).reshape(4096, 16)[:, :8]
v = raw[:, :5].astype(np.float64)
long_w = raw[:, 5]
blk = np.arange(4096)
keep = v[:, 3] > 0
v, blk, long_w = v[keep], blk[keep], long_w[keep]
v[:, :3] -= v[:, 0].min()                     # s_memrealtime: 100 MHz, the same counter on every CU
v[:, :3] /= 100.0                             # -> microseconds
v[:, 4] /= 100.0
start, first, end, items = v[:, 0], v[:, 1], v[:, 2], v[:, 3]
T = end.max()
print("workgroups with work: %d, items %d (%.1f per workgroup, min %d max %d); launch span %.1f us" % (len(v), items.sum(), items.mean(), items.min(), items.max(), T))
print("start stamps: median %.2f us, 99 %% %.2f us (%.1f %% of the span); first item done: median %.1f us (%.1f %% of the span), 99 %% %.1f us" % (np.median(start), np.percentile(start, 99), 100 * np.percentile(start, 99) / T, np.median(first), 100 * np.median(first) / T, np.percentile(first, 99)))
for q in (1, 10, 25, 50, 75, 90, 99, 100):
    print("  %3d %% of the workgroups have finished at %.1f %% of the span" % (q, 100 * np.percentile(end, q) / T))
busy = np.zeros(100)
for s_, e_ in zip(start, end):
    a, b = int(100 * s_ / T), min(99, int(100 * e_ / T))
    busy[a:b + 1] += 1
print("resident workgroups per 5 % of the span:", " ".join("%d" % x for x in busy[::5]))
per = (end - start) / items
print("us per item: mean %.1f, median %.1f, 10 %% %.1f, 90 %% %.1f" % (per.mean(), np.median(per), np.percentile(per, 10), np.percentile(per, 90)))
print("per home queue (block index % 8): workgroups, items, median / max end (% of the span)")
for c in range(8):
    m = (blk % 8) == c
    print("  queue %d: %4d workgroups, %6d items, end median %.1f %% max %.1f %%, items per workgroup %.1f (min %d max %d)" % (c, m.sum(), items[m].sum(), 100 * np.median(end[m]) / T, 100 * end[m].max() / T, items[m].mean(), items[m].min(), items[m].max()))
order = np.argsort(end)
print("earliest finishers: items", items[order[:12]].astype(int), "end %", np.round(100 * end[order[:12]] / T, 1))
print("latest finishers:   items", items[order[-12:]].astype(int), "end %", np.round(100 * end[order[-12:]] / T, 1))
print("corr(items, end) = %.2f" % np.corrcoef(items, end)[0, 1])
longest = v[:, 4]
o2 = np.argsort(-longest)[:16]
items_per_img = int(lib.hs_orb_debug_items_per_image(ex._h)) if hasattr(lib, "hs_orb_debug_items_per_image") else 0
print("longest items (us, work index, spilled):", [(round(float(longest[i]), 1), int(long_w[i] & 0xFFFFFFFF), int(long_w[i] >> 32)) for i in o2])
print("longest item per workgroup: median %.1f us, 90 %% %.1f, 99 %% %.1f, max %.1f; launch span %.1f us" % (np.median(longest), np.percentile(longest, 90), np.percentile(longest, 99), longest.max(), T))
