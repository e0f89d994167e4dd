"""Deterministic synthetic grey images for tests and bench.

Pure-noise images saturate FAST, so frames are *structured*: a smooth gradient background,
a few thousand filled rectangles / discs with random grey levels (each with its own stereo
disparity, painted far-to-near) and sigma~2 pixel noise.  Everything is integer arithmetic on a
counter-based SplitMix64 stream, so the same seed gives the same bytes on every machine
(no dependence on numpy's own generators).
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n, stream=0):
    """n 64-bit words of the SplitMix64 sequence started at `seed` (+ an independent stream id)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream) * np.uint64(0xDA942042E4DD58B5)
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * _GAMMA)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _noise(seed, stream, h, w):
    """integer noise ~ N(0, 2): centred sum of 8 uniform bytes, scaled."""
    r = splitmix64(seed, h * w, stream).view(np.uint8).reshape(h * w, 8).astype(np.int32).sum(axis=1)
    return np.floor_divide((r - 1020) * 2 + 104, 209).reshape(h, w)


def _scene(seed, w, h, n_shapes):
    r = splitmix64(seed, n_shapes * 8, stream=1).reshape(n_shapes, 8)
    x0 = (r[:, 0] % np.uint64(w + 64)).astype(np.int64) - 32
    y0 = (r[:, 1] % np.uint64(h + 64)).astype(np.int64) - 32
    sw = 8 + (r[:, 2] % np.uint64(112)).astype(np.int64)
    sh = 8 + (r[:, 3] % np.uint64(112)).astype(np.int64)
    grey = (r[:, 4] % np.uint64(256)).astype(np.int64)
    disc = (r[:, 5] % np.uint64(4)) == 0
    disp = 4 + (r[:, 6] % np.uint64(117)).astype(np.int64)      # stereo disparity in [4,120]
    order = np.argsort(disp, kind="stable")                     # far first, near painted last
    return [(int(x0[i]), int(y0[i]), int(sw[i]), int(sh[i]), int(grey[i]), bool(disc[i]), int(disp[i])) for i in order]


def _render(scene, w, h, shift_sign, bg_disp):
    xs = np.arange(w, dtype=np.int64)[None, :] + (bg_disp if shift_sign else 0)
    ys = np.arange(h, dtype=np.int64)[:, None]
    img = 56 + (xs * 96) // max(w, 1) + (ys * 48) // max(h, 1) + (((xs // 97) + (ys // 61)) % 2) * 6
    img = img.astype(np.int32)
    for (x0, y0, sw, sh, grey, disc, disp) in scene:
        xa = x0 - (disp if shift_sign else 0)
        xb, ya, yb = xa + sw, y0, y0 + sh
        cxa, cxb, cya, cyb = max(xa, 0), min(xb, w), max(ya, 0), min(yb, h)
        if cxa >= cxb or cya >= cyb:
            continue
        if not disc:
            img[cya:cyb, cxa:cxb] = grey
        else:
            yy = np.arange(cya, cyb, dtype=np.int64)[:, None] * 2 - (ya + yb - 1)
            xx = np.arange(cxa, cxb, dtype=np.int64)[None, :] * 2 - (xa + xb - 1)
            m = (xx * xx) * (sh * sh) + (yy * yy) * (sw * sw) <= (sw * sw) * (sh * sh)
            sub = img[cya:cyb, cxa:cxb]
            sub[m] = grey
    return img


def synth_image(seed, w, h, n_shapes=None):
    """One structured grey frame (uint8, h x w)."""
    if n_shapes is None:
        n_shapes = max(40, (w * h) // 800)
    img = _render(_scene(seed, w, h, n_shapes), w, h, False, 0) + _noise(seed, 2, h, w)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_rig(seed, n_cams, w, h, step=None, cams=None):
    """A rig of `n_cams` cameras looking at one wide scene (BASELINE config 5: "overlapping content between neighbours"): camera i sees the
    columns [i*step, i*step + w) of a (w + (n_cams-1)*step)-wide panorama (default step = w/4: neighbours share three quarters of their
    view), each with its own pixel noise.  `cams` selects which cameras to render (default all); returns a list of uint8 h x w frames."""
    if step is None:
        step = w // 4
    pw = w + (n_cams - 1) * step
    pano = _render(_scene(seed, pw, h, max(40, (pw * h) // 800)), pw, h, False, 0)
    out = []
    for i in (range(n_cams) if cams is None else cams):
        out.append(np.clip(pano[:, i * step:i * step + w] + _noise(seed, 20 + i, h, w), 0, 255).astype(np.uint8))
    return out


def synth_stereo_pair(seed, w, h, n_shapes=None):
    """Rectified stereo pair: the right frame renders the same scene with every shape moved left by
    its own disparity (background: 4 px) and independent noise."""
    if n_shapes is None:
        n_shapes = max(40, (w * h) // 800)
    scene = _scene(seed, w, h, n_shapes)
    left = _render(scene, w, h, False, 4) + _noise(seed, 2, h, w)
    right = _render(scene, w, h, True, 4) + _noise(seed, 3, h, w)
    return np.clip(left, 0, 255).astype(np.uint8), np.clip(right, 0, 255).astype(np.uint8)


def synth_local_map(kps, desc, depth, n_landmarks, seed, fx, fy, cx, cy, Rcw=None, tcw=None):
    """A synthetic local map for the projection matchers (BASELINE config 4): `n_landmarks` landmarks made by back-projecting the
    frame's own keypoints (cycled) at their stereo depth (a seeded depth where there is none), jittered by ~1.5 px and 3 % depth, with a
    few descriptor bits flipped; `size` / `min_dist` / `max_dist` / `normal` filled consistently with the pose (Rcw, tcw; identity if None).
    Returns a numpy array with the layout of hs_landmark (include/hyslam_amd.h)."""
    from ._native import LM_DTYPE
    n = len(kps)
    r = splitmix64(seed, n_landmarks * 8, stream=7).reshape(n_landmarks, 8)
    u = lambda c: (r[:, c] >> np.uint64(11)).astype(np.float64) / float(1 << 53)          # uniform [0,1)
    src = np.arange(n_landmarks) % max(n, 1)
    Rcw = np.eye(3, dtype=np.float32) if Rcw is None else np.asarray(Rcw, np.float32)
    tcw = np.zeros(3, np.float32) if tcw is None else np.asarray(tcw, np.float32)
    Rwc = Rcw.T.astype(np.float64)
    Ow = -Rwc @ tcw.astype(np.float64)
    d0 = np.asarray(depth, np.float64)[src]
    d = np.where(d0 > 0, d0, 2.0 + 23.0 * u(0)) * (0.97 + 0.06 * u(1))
    px = kps["x"][src].astype(np.float64) + 3.0 * (u(2) - 0.5)
    py = kps["y"][src].astype(np.float64) + 3.0 * (u(3) - 0.5)
    Pc = np.stack([(px - cx) * d / fx, (py - cy) * d / fy, d], 1)
    Pw = (Rwc @ (Pc - tcw.astype(np.float64)).T).T
    lms = np.zeros(n_landmarks, LM_DTYPE)
    lms["pos"] = Pw.astype(np.float32)
    lms["size"] = (kps["size"][src] * d / fx * (0.8 + 0.4 * u(4))).astype(np.float32)
    dist = np.linalg.norm(Pw - Ow, axis=1)
    lms["min_dist"] = (dist * 0.5).astype(np.float32)
    lms["max_dist"] = (dist * 2.0).astype(np.float32)
    lms["normal"] = ((Pw - Ow) / dist[:, None]).astype(np.float32)
    dd = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)[src].copy()
    flip = (r[:, 5] % np.uint64(256)).astype(np.int64)                                    # one flipped bit per landmark, two for the later copies
    dd[np.arange(n_landmarks), flip >> 3] ^= (1 << (flip & 7)).astype(np.uint8)
    late = np.arange(n_landmarks) >= n
    flip2 = (r[:, 6] % np.uint64(256)).astype(np.int64)
    dd[late, flip2[late] >> 3] ^= (1 << (flip2[late] & 7)).astype(np.uint8)
    lms["desc"] = dd
    lms["assoc_kp"] = -1
    lms["prev_angle"] = kps["angle"][src]
    return lms


def synth_vocab_tree(k=10, levels=4, seed=17):
    """A seeded synthetic k-ary vocabulary with `levels` levels below the root, as the flat tree of the C ABI (hs_vocab_tree): random node
    descriptors, consecutive word ids and idf-like weights at the leaves.  ORBvoc (the reference's vocabulary) is a missing blob, so bench.py's
    BoW variant of config 5 and the tests run on this stand-in.  Returns (VocabTree, keepalive arrays, number of words)."""
    from ._native import VocabTree
    r = splitmix64(seed, 8, stream=11)
    rng = np.random.default_rng(int(r[0] % np.uint64(1 << 62)))
    n_nodes = sum(k ** l for l in range(levels + 1))
    first_leaf = sum(k ** l for l in range(levels))
    cb = np.zeros(n_nodes, np.int32); cc = np.zeros(n_nodes, np.int32)
    nxt = 1
    for i in range(first_leaf):
        cb[i], cc[i] = nxt, k
        nxt += k
    desc = rng.integers(0, 256, (n_nodes, 32), dtype=np.uint8)
    word = np.full(n_nodes, -1, np.int32); word[first_leaf:] = np.arange(n_nodes - first_leaf)
    weight = np.zeros(n_nodes, np.float32); weight[first_leaf:] = rng.uniform(0.5, 9.0, n_nodes - first_leaf)
    T = VocabTree(n_nodes, levels, cb.ctypes.data, cc.ctypes.data, desc.ctypes.data, word.ctypes.data, weight.ctypes.data, None)
    return T, [cb, cc, desc, word, weight], n_nodes - first_leaf
