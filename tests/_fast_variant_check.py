"""Run under different HS_FAST_* environments by test_gpu_parity.py::test_fast_kernel_variants_in_subprocess:
stage-wise parity of a structured frame, full parity of a saturated (noise) frame, a frame with an odd width and the large-cell grid."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle
import hyslam_amd as HS
from hyslam_amd.synth import synth_image
from test_gpu_parity import stage_parity, settings, assert_same_features


def main():
    stage_parity(synth_image(1, 640, 480), 1000)
    stage_parity(synth_image(683, 643, 481), 700)
    stage_parity(synth_image(12, 1920, 1080), 2000)
    rng = np.random.default_rng(9)
    for shape in ((240, 320), (480, 900)):
        noise = rng.integers(0, 256, shape, dtype=np.uint8)
        ok, od = oracle.extract(oracle.default_params(1500), noise)
        gk, gd = HS.ORBExtractor(settings(1500))(noise)
        assert_same_features(gk, gd, ok, od)
    img = synth_image(77, 640, 480)
    for cells in (40, 24, 12):
        p = oracle.default_params(800)
        p.cell_px = cells
        ok, od = oracle.extract(p, img)
        s = settings(800)
        s.N_CELLS = cells
        gk, gd = HS.ORBExtractor(s)(img)
        assert_same_features(gk, gd, ok, od)
    # the stereo matcher on the same handle: golden pair, repeated calls (nothing may be left over between calls), identical views (median 0
    # rejects everything)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stereo_640x480_1000.npz"))
    ex = HS.ORBExtractor(settings(1000))
    cam = HS.Camera(float(g["fx"]), float(g["fx"]) * 0.12, float(g["h"]))
    sp = oracle.stereo_params(fx=float(g["fx"]), mbf=float(g["fx"]) * 0.12, n_rows=int(g["h"]))
    for rep in range(3):
        sm = HS.Stereomatcher(g["kL"], g["kR"], g["dL"], g["dR"], cam, extractor=ex)
        sm.computeStereoMatches()
        assert np.array_equal(sm.getData()[0], g["uRight"]) and np.array_equal(sm.getData()[1], g["depth"]), rep
        sm = HS.Stereomatcher(g["kL"], g["kL"], g["dL"], g["dL"], cam, extractor=ex)
        sm.computeStereoMatches()
        ouR, odepth, _, _ = oracle.stereo_match(g["kL"], g["dL"], g["kL"], g["dL"], sp)
        assert np.array_equal(sm.getData()[0], ouR) and np.array_equal(sm.getData()[1], odepth), rep
    # a batch of 8 frames in one call: the FAST work queues own whole images and walk them item-major (HS_FAST_IMAGE_MAJOR=1: image-major)
    frames = [synth_image(40 + i, 640, 480) for i in range(8)]
    ex8 = HS.ORBExtractor(settings(600))
    ref = [oracle.extract(oracle.default_params(600), f) for f in frames]
    for nb in (8, 3, 5):                                       # 8: an image is dealt to several queues; 3, 5: the item-major list of all images round-robin
        kl, dl = ex8.extract_batch(frames[:nb])
        for (ok, od), gk, gd in zip(ref, kl, dl):
            assert_same_features(gk, gd, ok, od)
    print("FAST_VARIANT_OK", {k: v for k, v in os.environ.items() if k.startswith("HS_")})


if __name__ == "__main__":
    main()
