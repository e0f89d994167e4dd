"""Run by test_gpu_parity.py in its own process: torch first (its bundled HIP runtime), then the library.
Checks hs_stereo_frontend_batch_device / hs_orb_extract_batch_device on torch tensors against the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyslam_amd as HS  # noqa: E402
from hyslam_amd import _native as N  # noqa: E402
from hyslam_amd.synth import synth_stereo_pair  # noqa: E402
import oracle  # noqa: E402

dev = torch.device("cuda", 0)
W, H, B, NF = 640, 480, 3, 1000
pairs = [synth_stereo_pair(80 + i, W, H) for i in range(B)]
left = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
right = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF))
cap = ex.max_keypoints()
kb = N.KP_DTYPE.itemsize
mk = lambda n, dt=torch.uint8: torch.zeros(n, dtype=dt, device=dev)
kL, kR, dL, dR = mk(B * cap * kb), mk(B * cap * kb), mk(B * cap * 32), mk(B * cap * 32)
nL, nR = mk(B, torch.int32), mk(B, torch.int32)
uR, depth = mk(B * cap, torch.float32), mk(B * cap, torch.float32)
cam = HS.Camera(500.0, 60.0, float(H))
sp = HS.stereo_params(cam)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ex.stereo_frontend_batch_device(left.data_ptr(), right.data_ptr(), B, W, H, W, W * H, kL.data_ptr(), dL.data_ptr(), nL.data_ptr(),
                                    kR.data_ptr(), dR.data_ptr(), nR.data_ptr(), cap, sp, uR.data_ptr(), depth.data_ptr(), s.cuda_stream)
s.synchronize()
p = oracle.default_params(NF)
osp = oracle.stereo_params(fx=500.0, mbf=60.0, n_rows=H)
for i in range(B):
    okL, odL = oracle.extract(p, pairs[i][0])
    okR, odR = oracle.extract(p, pairs[i][1])
    ouR, odepth, _, _ = oracle.stereo_match(okL, odL, okR, odR, osp)
    n_l, n_r = int(nL[i]), int(nR[i])
    assert (n_l, n_r) == (len(okL), len(okR)), (i, n_l, n_r, len(okL), len(okR))
    gkL = kL.view(B, cap * kb)[i].cpu().numpy().view(N.KP_DTYPE)[:n_l]
    gkR = kR.view(B, cap * kb)[i].cpu().numpy().view(N.KP_DTYPE)[:n_r]
    assert gkL.tobytes() == okL.tobytes() and gkR.tobytes() == okR.tobytes(), i
    assert np.array_equal(dL.view(B, cap, 32)[i, :n_l].cpu().numpy(), odL) and np.array_equal(dR.view(B, cap, 32)[i, :n_r].cpu().numpy(), odR), i
    assert np.array_equal(uR.view(B, cap)[i, :n_l].cpu().numpy(), ouR) and np.array_equal(depth.view(B, cap)[i, :n_l].cpu().numpy(), odepth), i
# mono device batch on the current stream + separate stereo call
kL.zero_(); nL.zero_()
ex.extract_batch_device(left.data_ptr(), B, W, H, W, W * H, kL.data_ptr(), dL.data_ptr(), nL.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
okL, odL = oracle.extract(p, pairs[1][0])
assert kL.view(B, cap * kb)[1].cpu().numpy().view(N.KP_DTYPE)[:int(nL[1])].tobytes() == okL.tobytes()
# two lanes: the batch is split over two streams inside the handle; results must not change
ex.set_lanes(2)
kL.zero_(); kR.zero_(); dL.zero_(); dR.zero_(); nL.zero_(); nR.zero_(); uR.zero_(); depth.zero_()
ex.stereo_frontend_batch_device(left.data_ptr(), right.data_ptr(), B, W, H, W, W * H, kL.data_ptr(), dL.data_ptr(), nL.data_ptr(),
                                kR.data_ptr(), dR.data_ptr(), nR.data_ptr(), cap, sp, uR.data_ptr(), depth.data_ptr(), s.cuda_stream)
s.synchronize()
for i in range(B):
    okL, odL = oracle.extract(p, pairs[i][0])
    okR, odR = oracle.extract(p, pairs[i][1])
    ouR, odepth, _, _ = oracle.stereo_match(okL, odL, okR, odR, osp)
    n_l = int(nL[i])
    assert n_l == len(okL) and int(nR[i]) == len(okR), i
    assert kL.view(B, cap * kb)[i].cpu().numpy().view(N.KP_DTYPE)[:n_l].tobytes() == okL.tobytes(), i
    assert np.array_equal(dR.view(B, cap, 32)[i, :len(okR)].cpu().numpy(), odR), i
    assert np.array_equal(uR.view(B, cap)[i, :n_l].cpu().numpy(), ouR) and np.array_equal(depth.view(B, cap)[i, :n_l].cpu().numpy(), odepth), i
kL.zero_(); nL.zero_()
ex.extract_batch_device(left.data_ptr(), B, W, H, W, W * H, kL.data_ptr(), dL.data_ptr(), nL.data_ptr(), cap, s.cuda_stream)
s.synchronize()
for i in range(B):
    okL, odL = oracle.extract(p, pairs[i][0])
    assert kL.view(B, cap * kb)[i].cpu().numpy().view(N.KP_DTYPE)[:int(nL[i])].tobytes() == okL.tobytes(), i
# padded device buffers: a 16-byte aligned row stride with spare rows between images, and an odd stride (unaligned loads everywhere)
ex.set_lanes(1)
for S, extra in ((704, 3), (645, 1)):
    buf = torch.zeros(B, H + extra, S, dtype=torch.uint8, device=dev)
    buf[:, :H, :W] = left
    kL.zero_(); nL.zero_()
    ex.extract_batch_device(buf.data_ptr(), B, W, H, S, (H + extra) * S, kL.data_ptr(), dL.data_ptr(), nL.data_ptr(), cap, s.cuda_stream)
    s.synchronize()
    for i in range(B):
        okL, odL = oracle.extract(p, pairs[i][0])
        n_l = int(nL[i])
        assert n_l == len(okL), (S, i, n_l, len(okL))
        assert kL.view(B, cap * kb)[i].cpu().numpy().view(N.KP_DTYPE)[:n_l].tobytes() == okL.tobytes(), (S, i)
        assert np.array_equal(dL.view(B, cap, 32)[i, :n_l].cpu().numpy(), odL), (S, i)
print("device api ok")
