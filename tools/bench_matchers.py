#!/usr/bin/env python3
"""Device timings of the matcher kernels (BASELINE configs 4/5 shapes), all data resident in HBM; prints one JSON line.
  projection : 1920x1080 frame, 2000 keypoints, 50 000-landmark local map, th=5 (SearchByProjection local-map variant)
  knn2       : 2000 x 2000 brute-force Hamming 2-NN
Run on the GPU box: python tools/bench_matchers.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyslam_amd as HS  # noqa: E402
from hyslam_amd import _native as N  # noqa: E402
import oracle  # noqa: E402
import scenes  # noqa: E402

dev = torch.device("cuda", 0)
sc = scenes.projection_scene(33, 1920, 1080, nfeat=2000, copies=25, fx=1050.0)
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000))
fa = sc["frame_args"]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
d_kps, d_desc, d_uR, d_obs = t(fa["kps"]), t(fa["desc"]), t(fa["uR"].astype(np.float32)), t(fa["kp_lm_obs"].astype(np.int32))
Fh, keep = oracle.make_frame_view(N.FrameView, **fa)
Fd = N.FrameView.from_buffer_copy(Fh)
Fd.kps, Fd.desc, Fd.uR, Fd.kp_lm_obs = d_kps.data_ptr(), d_desc.data_ptr(), d_uR.data_ptr(), d_obs.data_ptr()
lms = sc["lms"]; L = len(lms)
d_lms = t(lms)
midx = torch.empty(L, dtype=torch.int32, device=dev); mdist = torch.empty(L, dtype=torch.float32, device=dev); nm = torch.zeros(1, dtype=torch.int32, device=dev)
pp = N.ProjParams(5.0, 100.0, 0.8, 0.5, 1.5, 1, 1, 0)
ts = torch.cuda.Stream()                       # a real stream handle: 0 (torch's default stream) would mean "the handle's own stream" to the C ABI
st = ts.cuda_stream


def run_proj():
    N.check(ex._h, ex._lib.hs_search_by_projection_device(ex._h, C.byref(Fd), d_lms.data_ptr(), L, C.byref(pp), midx.data_ptr(), mdist.data_ptr(), nm.data_ptr(), st))


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ts)
    for _ in range(n):
        fn()
    e1.record(ts); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ms_proj = timeit(run_proj)
oi, od, on = oracle.search_by_projection(oracle.make_frame_view(oracle.FrameView, **fa)[0], lms, oracle.ProjParams(5.0, 100.0, 0.8, 0.5, 1.5, 1, 1, 0))
assert int(nm.item()) == on and np.array_equal(midx.cpu().numpy(), oi), "device projection search differs from the oracle"
t0 = time.perf_counter(); oracle.search_by_projection(oracle.make_frame_view(oracle.FrameView, **fa)[0], lms, oracle.ProjParams(5.0, 100.0, 0.8, 0.5, 1.5, 1, 1, 0)); cpu_proj = (time.perf_counter() - t0) * 1e3

q = torch.randint(0, 256, (2000, 32), dtype=torch.uint8, device=dev); tr = torch.randint(0, 256, (2000, 32), dtype=torch.uint8, device=dev)
bi, bd, sd = (torch.empty(2000, dtype=torch.int32, device=dev) for _ in range(3))
ms_knn = timeit(lambda: N.check(ex._h, ex._lib.hs_hamming_knn2_device(ex._h, q.data_ptr(), 2000, tr.data_ptr(), 2000, bi.data_ptr(), bd.data_ptr(), sd.data_ptr(), st)))
print(json.dumps({"projection_50k_landmarks_ms": round(ms_proj, 4), "landmarks": L, "keypoints": int(Fd.n), "matches": on,
                  "projection_landmarks_per_s": round(L / ms_proj * 1e3), "projection_oracle_cpu_ms_1core": round(cpu_proj, 1),
                  "knn2_2000x2000_ms": round(ms_knn, 4), "knn2_Gpairs_per_s": round(4e6 / ms_knn / 1e6, 2)}))
