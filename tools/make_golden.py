#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded synthetic inputs' identities (sha256) and the ORACLE's outputs for them.

The reference holds no golden vectors for this path and cannot be run here (OpenCV absent), so these
fixtures pin the oracle against regressions and give the GPU tests a second, frozen comparison point; they do not pin
the oracle to the reference ("parity unpinned", see oracle/hs_oracle.h).  Run from the repo root."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from hyslam_amd.synth import synth_image, synth_stereo_pair  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def mono(name, seed, w, h, nfeat, scale):
    img = synth_image(seed, w, h)
    p = oracle.default_params(nfeat, scale)
    k, d, dbg = oracle.extract(p, img, debug=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, w=w, h=h, nfeat=nfeat, scale=np.float32(scale),
                        image_sha256=sha(img), keypoints=k, descriptors=d,
                        n_candidates=dbg["n_candidates"], n_selected=dbg["n_selected"],
                        pyramid_sha256=np.array([sha(l) for l in dbg["pyramid"]]),
                        blurred_sha256=np.array([sha(l) for l in dbg["blurred"]]))
    print(name, len(k), dbg["n_candidates"].tolist())


def stereo(name, seed, w, h, nfeat, fx):
    L, R = synth_stereo_pair(seed, w, h)
    p = oracle.default_params(nfeat)
    kL, dL = oracle.extract(p, L)
    kR, dR = oracle.extract(p, R)
    sp = oracle.stereo_params(fx=fx, mbf=fx * 0.12, n_rows=h)
    uR, depth, bi, bd = oracle.stereo_match(kL, dL, kR, dR, sp)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), seed=seed, w=w, h=h, nfeat=nfeat, fx=np.float32(fx),
                        left_sha256=sha(L), right_sha256=sha(R), kL=kL, dL=dL, kR=kR, dR=dR, uRight=uR, depth=depth,
                        best_idx=bi, best_dist=bd)
    print(name, len(kL), len(kR), int((depth > 0).sum()))


def bench_c2(name, ranks=8, distinct=4, w=1920, h=1080, nfeat=2000):
    """bench.py's own workload (BASELINE C2): sha256 of the oracle's outputs for EVERY distinct pair of every rank 0..7 (pair i of rank r =
    `synth_stereo_pair(1000 + 97 * r + i, 1920, 1080)`, i = 0..3, fx = 1050, mbf = fx * 0.12).  bench.py hashes what ITS timed loop left in the
    output buffers of all `--pairs` pairs the same way (pair j of a step is a copy of distinct pair j % 4) and prints `parity_checksum_ok` +
    `pairs_checked` — data only: the oracle never travels into the bench's measured path.  The rank's top-level fields are pair 0's (round 1-5 layout)."""
    import json
    p = oracle.default_params(nfeat)
    sp = oracle.stereo_params(fx=1050.0, mbf=1050.0 * 0.12, n_rows=h)
    out = {"workload": "bench.py C2, pair i of rank r = synth_stereo_pair(1000 + 97 r + i, %d, %d), i = 0..%d, %d features, fx 1050, mbf 126" % (w, h, distinct - 1, nfeat),
           "hash": "sha256 over kL[:nL].tobytes() + dL[:nL] + kR[:nR] + dR[:nR] + uRight[:nL] + depth[:nL] (hs_keypoint records, uint8 descriptors, float32)",
           "ranks": {}}
    for r in range(ranks):
        per_pair = []
        for i in range(distinct):
            L, R = synth_stereo_pair(1000 + 97 * r + i, w, h)
            kL, dL, kR, dR, uR, depth = oracle.stereo_frontend(p, sp, L, R)
            hsh = hashlib.sha256()
            for a in (kL, dL, kR, dR, uR, depth):
                hsh.update(np.ascontiguousarray(a).tobytes())
            per_pair.append({"seed": 1000 + 97 * r + i, "left_sha256": sha(L), "nL": len(kL), "nR": len(kR), "stereo_matches": int((depth > 0).sum()),
                             "outputs_sha256": hsh.hexdigest()})
            print(name, r, i, len(kL), len(kR), int((depth > 0).sum()))
        out["ranks"][str(r)] = dict(per_pair[0], pairs=per_pair)
    json.dump(out, open(os.path.join(OUT, name + ".json"), "w"), indent=1)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "bench":
        bench_c2("bench_c2_seed1000")
        sys.exit(0)
    mono("c1_mono_640x480_1000", 1, 640, 480, 1000, 1.2)          # BASELINE.json configs[0]
    mono("imaging_800x600_1500_s14", 4, 800, 600, 1500, 1.4)      # "Imaging" profile: scale 1.4 (config/slam_feature_config.yaml:22-29)
    stereo("stereo_640x480_1000", 3, 640, 480, 1000, 500.0)
    bench_c2("bench_c2_seed1000")
