#!/opt/conda/bin/python3.9
"""Reference outputs from scikit-image (an implementation independent of this repository and of OpenCV) for the parts of the path it also
implements.  Runs under the image's conda python 3.9, the only interpreter here that has scikit-image:
    /opt/conda/bin/python3.9 tools/skimage_reference.py IN.npz OUT.npz
IN: img (uint8, 2-D), threshold, kps (n,2) int (x, y) for the orientation check.
OUT: fast_score  (h,w) int16: the largest threshold t' >= threshold at which skimage.feature.corner_fast(n=9) still calls the pixel a corner
                 (0 where it is no corner at `threshold`) — by definition cv::FAST's cornerScore;
     angle_deg   (n,)  float64: skimage.feature.corner_orientations with ORB's 31-px disc (skimage.feature.orb.OFAST_MASK), degrees in [0,360);
     umax, pattern: skimage's copies of ORB's disc half-widths and of the 256 test pairs;
     brief_bits  (n,256) uint8: skimage.feature.orb_cy._orb_loop (steered BRIEF) at kps with `kps_angle_deg` on the same image."""
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage.feature import corner_fast, corner_orientations          # noqa: E402
from skimage.feature import orb as sk_orb                              # noqa: E402
from skimage.feature.orb import OFAST_MASK, OFAST_UMAX                 # noqa: E402
from skimage.feature.orb_cy import _orb_loop                           # noqa: E402


def main():
    d = np.load(sys.argv[1])
    img = d["img"].astype(np.float64)          # integer-valued doubles: img_as_float leaves floats alone, so thresholds stay in grey levels
    t0 = int(d["threshold"])
    h, w = img.shape
    score = np.zeros((h, w), np.int16)
    alive = np.ones((h, w), bool)
    for t in range(t0, 256):
        resp = corner_fast(img, n=9, threshold=float(t)) > 0
        alive &= resp
        if not alive.any():
            break
        score[alive] = t
    kps = d["kps"]
    ang = np.zeros(len(kps))
    if len(kps):
        rc = np.stack([kps[:, 1], kps[:, 0]], 1).astype(np.intp)       # (row, col)
        ang = np.rad2deg(corner_orientations(img, rc, OFAST_MASK)) % 360.0
    bits = np.zeros((0, 256), np.uint8)
    if len(kps) and "kps_angle_deg" in d.files:
        rc = np.ascontiguousarray(np.stack([kps[:, 1], kps[:, 0]], 1).astype(np.intp))
        bits = np.asarray(_orb_loop(np.ascontiguousarray(img), rc, np.deg2rad(d["kps_angle_deg"].astype(np.float64))))
    pos = np.loadtxt(sk_orb.__file__.replace("orb.py", "orb_descriptor_positions.txt"), dtype=np.int8)
    np.savez(sys.argv[2], fast_score=score, angle_deg=ang, umax=np.asarray(OFAST_UMAX, np.int32), pattern=pos, brief_bits=bits)


if __name__ == "__main__":
    main()
