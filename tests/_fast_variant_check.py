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
    print("FAST_VARIANT_OK", {k: v for k, v in os.environ.items() if k.startswith("HS_FAST")})


if __name__ == "__main__":
    main()
