"""Synthetic tracking scenes for the matcher tests (BASELINE config 4 in miniature): a frame with extracted features and a
local map of landmarks back-projected from those features at seeded depths under a slightly different pose."""
import numpy as np

import oracle
from hyslam_amd.synth import synth_stereo_pair


def small_rotation(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


def projection_scene(seed, w=640, h=480, nfeat=1000, copies=3, sensor=1, fx=500.0):
    """-> dict(frame_args=..., lms=LM array, kps, desc, uR).  frame_args feed oracle.make_frame_view(cls, **frame_args)."""
    rng = np.random.default_rng(seed)
    L, R = synth_stereo_pair(seed, w, h)
    p = oracle.default_params(nfeat)
    sp = oracle.stereo_params(fx=fx, mbf=fx * 0.12, n_rows=h)
    kps, desc, kR, dR, uR, depth = oracle.stereo_frontend(p, sp, L, R)
    n = len(kps)
    cx, cy = w / 2.0 - 0.5, h / 2.0 - 0.5
    Rcw = small_rotation(*(rng.normal(0, 0.004, 3)))
    tcw = rng.normal(0, 0.02, 3).astype(np.float32)
    Rwc = Rcw.T
    Ow = -Rwc @ tcw
    lms = np.zeros(n * copies + 40, oracle.LM_DTYPE)
    k = 0
    for c in range(copies):
        d = np.where(depth > 0, depth, rng.uniform(2.0, 25.0, n)).astype(np.float32) * rng.uniform(0.97, 1.03, n).astype(np.float32)
        px = kps["x"] + rng.normal(0, 1.5 if c else 0.3, n)
        py = kps["y"] + rng.normal(0, 1.5 if c else 0.3, n)
        Pc = np.stack([(px - cx) * d / fx, (py - cy) * d / fx, d], 1)
        Pw = (Rwc @ (Pc - tcw).T).T
        sl = slice(k, k + n)
        lms["pos"][sl] = Pw.astype(np.float32)
        lms["size"][sl] = (kps["size"] * d / fx * rng.uniform(0.7, 1.4, n)).astype(np.float32)
        dist = np.linalg.norm(Pw - Ow, axis=1)
        lms["min_dist"][sl] = (dist * rng.uniform(0.3, 1.1, n)).astype(np.float32)       # some fail the 0.8*min test
        lms["max_dist"][sl] = (dist * rng.uniform(0.9, 3.0, n)).astype(np.float32)       # some fail the 1.2*max test
        nrm = (Pw - Ow) / dist[:, None]                                 # mean viewing direction: camera -> point (MapPoint::UpdateNormalAndDepth)
        lms["normal"][sl] = nrm.astype(np.float32)
        dd = desc.copy()
        flips = rng.integers(0, 30 if c else 8, n)
        for i in range(n):
            bits = rng.integers(0, 256, flips[i])
            for b in bits:
                dd[i, b >> 3] ^= 1 << (b & 7)
        lms["desc"][sl] = dd
        lms["assoc_kp"][sl] = -1
        lms["prev_angle"][sl] = (kps["angle"] + rng.normal(0, 4, n) + (rng.random(n) < 0.1) * rng.uniform(0, 360, n)) % 360
        k += n
    # outliers: behind the camera, far outside the image, null entries
    t = lms[k:]
    t["pos"] = rng.normal(0, 30, (len(t), 3)).astype(np.float32)
    t["pos"][:10, 2] = -np.abs(t["pos"][:10, 2]) - 1
    t["size"], t["min_dist"], t["max_dist"], t["assoc_kp"] = 0.2, 0.1, 1e3, -1
    t["desc"] = rng.integers(0, 256, (len(t), 32), dtype=np.uint8)
    t["skip"][-5:] = 1
    # a few landmarks already associated with a keypoint of this frame (landMarkSizePixels uses the keypoint size then)
    pick = rng.choice(n, min(25, n), replace=False)
    lms["assoc_kp"][pick] = pick
    kp_lm_obs = np.full(n, -1, np.int32)
    kp_lm_obs[rng.choice(n, min(60, n), replace=False)] = rng.integers(0, 4, min(60, n))                  # 0 observations must NOT block a keypoint
    perm = rng.permutation(len(lms))
    frame_args = dict(Rcw=Rcw, tcw=tcw, fx=fx, fy=fx, cx=cx, cy=cy, mbf=fx * 0.12, sensor=sensor, bounds=(0.0, float(w), 0.0, float(h)),
                      kps=kps, desc=desc, uR=uR, kp_lm_obs=kp_lm_obs)
    return dict(frame_args=frame_args, lms=lms[perm].copy(), kps=kps, desc=desc, uR=uR)


def synthetic_featvec(desc, n_nodes, seed):
    """A seeded stand-in for DBoW2's FeatureVector (node id -> ascending keypoint indices): descriptors are hashed onto
    `n_nodes` vocabulary nodes by their first bits, so two views of the same point mostly share a node."""
    rng = np.random.default_rng(seed)
    sel = rng.choice(256, 12, replace=False)
    bits = np.unpackbits(desc, axis=1, bitorder="little")[:, sel]
    node = (bits.astype(np.int64) * (1 << np.arange(12))).sum(1) % n_nodes
    ids = np.unique(node)
    idx = np.concatenate([np.nonzero(node == i)[0] for i in ids]).astype(np.int32)
    ptr = np.concatenate([[0], np.cumsum([(node == i).sum() for i in ids])]).astype(np.int32)
    return ids.astype(np.int32) * 7 + 3, ptr, idx          # non-contiguous node ids, like real vocabulary node ids
