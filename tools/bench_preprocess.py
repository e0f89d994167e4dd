#!/usr/bin/env python3
"""ImageProcessing::PreProcessImg on the device (round 6): what the reference's "Imaging" camera frame costs on its way to the extractor.
  python3 tools/bench_preprocess.py          (GPU box)   prints one JSON object
    kernel     k_preprocess alone on device-resident frames: us per frame, GB/s of algorithmic bytes (w h CN read + ow oh written) against the HBM peak
    call       hs_orb_extract_camera_batch (colour frame in host memory -> features in host memory) against
               [the oracle's PreProcessImg on one host core + hs_orb_extract of the grey frame] — the reference does the former part with OpenCV on the CPU
The oracle is used as the CPU stand-in and as the checker (tools/ is not product code)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import hyslam_amd as HS  # noqa: E402
import oracle  # noqa: E402
from hyslam_amd.synth import synth_image  # noqa: E402


def frame(seed, w, h, cn):
    if cn == 1:
        return synth_image(seed, w, h)
    ch = [synth_image(seed + 7 * k, w, h) for k in range(3)] + ([np.full((h, w), 255, np.uint8)] if cn == 4 else [])
    return np.ascontiguousarray(np.stack(ch, axis=2))


out = {}
dev = torch.device("cuda", 0)
for name, (w, h, cn, rgb, scale, nfeat, fs) in {"imaging_2704x2028x3_scale0.5": (2704, 2028, 3, True, 0.5, 3000, 1.4), "stereo_1280x720x3_scale1.0": (1280, 720, 3, True, 1.0, 1000, 1.2),
                                                 "bilinear_1920x1080x3_scale0.75": (1920, 1080, 3, True, 0.75, 2000, 1.2)}.items():
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=nfeat, fScaleFactor=fs))
    f = frame(5, w, h, cn)
    ow, oh = oracle.preprocess_size(w, h, scale)
    B = max(8, int(700e6 // (w * h * cn)))            # > 256 MiB of frames per pass: the Infinity Cache must not serve the re-reads
    d_src = torch.from_numpy(np.stack([f] * B)).to(dev)
    gp = (ow + 63) & ~63
    d_grey = torch.zeros((B, oh, gp), dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    run = lambda: ex.preprocess_device(d_src.data_ptr(), w, h, w * cn, w * cn * h, B, cn, rgb, scale, d_grey.data_ptr(), gp, gp * oh, st.cuda_stream)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record(st)
    for _ in range(reps):
        run()
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * B)
    algo = w * h * cn + ow * oh
    assert np.array_equal(d_grey[0, :, :ow].cpu().numpy(), oracle.preprocess(f, rgb, scale))
    # the call surface, host memory to host memory
    ex.extract_camera_batch([f], rgb, scale)
    t0 = time.perf_counter()
    for _ in range(20):
        k, d = ex.extract_camera_batch([f], rgb, scale)
    call_ms = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    for _ in range(5):
        g = oracle.preprocess(f, rgb, scale)
    cpu_pre_ms = (time.perf_counter() - t0) / 5 * 1e3
    ex(g)
    t0 = time.perf_counter()
    for _ in range(20):
        k2, d2 = ex(g)
    grey_call_ms = (time.perf_counter() - t0) / 20 * 1e3
    assert k[0].tobytes() == k2.tobytes()
    out[name] = {"kernel_us_per_frame": round(us, 2), "kernel_GBps_algorithmic": round(algo / us / 1e3, 1), "kernel_frac_of_8TBps": round(algo / us / 1e3 / 8000, 4),
                 "bytes_per_frame": algo, "camera_call_ms": round(call_ms, 3), "cpu_PreProcessImg_ms_one_core_oracle": round(cpu_pre_ms, 3), "grey_call_ms": round(grey_call_ms, 3),
                 "host_path_ms": round(cpu_pre_ms + grey_call_ms, 3), "keypoints": int(len(k[0]))}
print(json.dumps(out))
