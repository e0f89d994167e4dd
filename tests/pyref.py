"""Independent numpy restatement of the primitives the oracle implements in C++ (test infrastructure).

Written from the specification text, vectorised over whole images, so that it
shares no code structure with oracle/hs_oracle.cpp: an indexing or rounding slip in either shows up as a
disagreement.  Slow; used on small inputs only.
"""
import numpy as np

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
TAPS = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def cv_round(v):
    return np.rint(v).astype(np.int64)          # round half to even, like cvtss2si


def resize_linear(src, dw, dh):
    sh, sw = src.shape
    S = src.astype(np.int64)

    def table(dn, sn, clamp_weights):
        scale = 1.0 / (np.float64(dn) / np.float64(sn))
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp_weights:
            low = s < 0
            f[low] = 0
            s[low] = 0
            high = s >= sn - 1
            f[high] = 0
            s[high] = sn - 1
        a0 = np.clip(cv_round((np.float32(1.0) - f) * np.float32(2048)), -32768, 32767)
        a1 = np.clip(cv_round(f * np.float32(2048)), -32768, 32767)
        return s, a0, a1

    sx, a0, a1 = table(dw, sw, True)
    sy, b0, b1 = table(dh, sh, False)
    sx1 = np.minimum(sx + 1, sw - 1)
    Hrows = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]            # (sh, dw)
    r0 = np.clip(sy, 0, sh - 1)
    r1 = np.clip(sy + 1, 0, sh - 1)
    H0, H1 = Hrows[r0], Hrows[r1]
    out = (((b0[:, None] * (H0 >> 4)) >> 16) + ((b1[:, None] * (H1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def preprocess(src, rgb, fscale):
    """ImageProcessing::PreProcessImg (src/main/ImageProcessing.cpp:118-138) from the text of OpenCV 3.4's resize / cvtColor: vectorised over the frame
    (the oracle walks pixels).  src (h, w) or (h, w, cn) uint8."""
    if src.ndim == 2:
        src = src[:, :, None]
    sh, sw, cn = src.shape
    inv = np.float64(np.float32(fscale))
    dw, dh = int(np.rint(sw * inv)), int(np.rint(sh * inv))
    scale = 1.0 / inv
    S = src.astype(np.int64)
    if (dw, dh) == (sw, sh):
        col = S
    elif abs(scale - np.rint(scale)) < np.finfo(np.float64).eps and int(np.rint(scale)) == 2:
        # INTER_LINEAR at exactly 1/2 is INTER_AREA's fast path: rounded 2x2 means for the full blocks, float means of what exists for trailing partial ones
        pad = np.zeros((2 * dh + 2, 2 * dw + 2, cn), np.int64)
        msk = np.zeros((2 * dh + 2, 2 * dw + 2, 1), np.int64)
        hh, ww = min(sh, 2 * dh), min(sw, 2 * dw)
        pad[:hh, :ww] = S[:hh, :ww]
        msk[:hh, :ww] = 1
        blk = lambda a: a[0:2 * dh:2, 0:2 * dw:2] + a[0:2 * dh:2, 1:2 * dw:2] + a[1:2 * dh:2, 0:2 * dw:2] + a[1:2 * dh:2, 1:2 * dw:2]
        ssum, cnt = blk(pad), blk(msk)
        full = (np.arange(dw)[None, :, None] < sw // 2) & (2 * np.arange(dh)[:, None, None] + 1 < sh)
        fast = (ssum + 2) >> 2
        with np.errstate(divide="ignore", invalid="ignore"):
            slow = np.where(cnt > 0, np.rint(ssum.astype(np.float32) / np.maximum(cnt, 1).astype(np.float32)), 0).astype(np.int64)
        col = np.where(full, fast, np.clip(slow, 0, 255))
    else:
        def table(dn, sn, clamp_weights):
            d = np.arange(dn, dtype=np.float64)
            f = ((d + 0.5) * scale - 0.5).astype(np.float32)
            s_ = np.floor(f).astype(np.int64)
            f = (f - s_.astype(np.float32)).astype(np.float32)
            if clamp_weights:
                low = s_ < 0
                f[low] = 0; s_[low] = 0
                high = s_ >= sn - 1
                f[high] = 0; s_[high] = sn - 1
            a0 = np.clip(cv_round((np.float32(1.0) - f) * np.float32(2048)), -32768, 32767)
            a1 = np.clip(cv_round(f * np.float32(2048)), -32768, 32767)
            return s_, a0, a1
        sx, a0, a1 = table(dw, sw, True)
        sy, b0, b1 = table(dh, sh, False)
        sx1 = np.minimum(sx + 1, sw - 1)
        H = S[:, sx] * a0[None, :, None] + S[:, sx1] * a1[None, :, None]
        H0, H1 = H[np.clip(sy, 0, sh - 1)], H[np.clip(sy + 1, 0, sh - 1)]
        col = (((b0[:, None, None] * (H0 >> 4)) >> 16) + ((b1[:, None, None] * (H1 >> 4)) >> 16) + 2) >> 2
    if cn == 1:
        return col[:, :, 0].astype(np.uint8)
    r, g, b = (col[:, :, 0], col[:, :, 1], col[:, :, 2]) if rgb else (col[:, :, 2], col[:, :, 1], col[:, :, 0])
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def fast_scores(img, t=20):
    """Score map (0 = not a corner) over the whole view; only [3,h-3) x [3,w-3) can be non-zero."""
    h, w = img.shape
    I = img.astype(np.int64)
    c = I[3:h - 3, 3:w - 3]
    ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])      # (16, H, W)
    d = c[None] - ring
    dark = d > t                 # ring < v - t
    bright = d < -t              # ring > v + t

    def arc9(m):
        mm = np.concatenate([m, m[:8]], 0)
        ok = np.zeros(m.shape[1:], bool)
        for k in range(16):
            ok |= mm[k:k + 9].all(0)
        return ok

    corner = arc9(dark) | arc9(bright)
    dd = np.concatenate([d, d[:8]], 0)
    amin = np.max(np.stack([dd[k:k + 9].min(0) for k in range(16)]), 0)
    amax = np.max(np.stack([(-dd[k:k + 9]).min(0) for k in range(16)]), 0)
    score = np.maximum(np.maximum(amin, amax), t) - 1
    out = np.zeros((h, w), np.int64)
    out[3:h - 3, 3:w - 3] = np.where(corner, score, 0)
    return out


def fast(img, t=20, nonmax=True):
    """-> rows (x, y, score) in scan order."""
    h, w = img.shape
    if h < 7 or w < 7:
        return np.zeros((0, 3), np.int64)
    s = fast_scores(img, t)
    keep = s > 0
    if nonmax:
        p = np.pad(s, 1)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx or dy:
                    keep &= s > p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys, s[ys, xs]], 1)


def gaussian_blur7(img, taps=TAPS):
    taps = np.asarray(taps, np.int64)
    p = np.pad(img.astype(np.int64), 3, mode="reflect")        # numpy 'reflect' == BORDER_REFLECT_101
    h, w = img.shape
    hor = np.zeros((h + 6, w), np.int64)
    for k in range(7):
        hor = np.minimum(hor + np.minimum(taps[k] * p[:, k:k + w], 0xFFFF), 0xFFFF)
    ver = np.zeros((h, w), np.int64)
    for k in range(7):
        ver = np.minimum(ver + taps[k] * hor[k:k + h], 0xFFFFFFFF)
    return np.minimum((ver + 0x8000) >> 16, 255).astype(np.uint8)


def fast_atan2(y, x):
    f = np.float32
    k = f(180.0 / np.pi)
    p1, p3, p5, p7 = (f(0.9997878412794807) * k, f(-0.3258083974640975) * k, f(0.1555786518463281) * k, f(-0.04432655554792128) * k)
    x, y = f(x), f(y)
    ax, ay = f(abs(x)), f(abs(y))
    eps = f(2.2204460492503131e-16)
    if ax >= ay:
        c = f(ay / f(ax + eps))
        c2 = f(c * c)
        a = f(f(f(f(f(f(f(p7 * c2) + p5) * c2) + p3) * c2) + p1) * c)
    else:
        c = f(ax / f(ay + eps))
        c2 = f(c * c)
        a = f(f(90.0) - f(f(f(f(f(f(f(p7 * c2) + p5) * c2) + p3) * c2) + p1) * c))
    if x < 0:
        a = f(f(180.0) - a)
    if y < 0:
        a = f(f(360.0) - a)
    return a


def ic_angle(blurred, x, y):
    cx, cy = int(np.rint(np.float32(x))), int(np.rint(np.float32(y)))
    m10 = m01 = 0
    for v in range(-15, 16):
        d = UMAX[abs(v)]
        row = blurred[cy + v, cx - d:cx + d + 1].astype(np.int64)
        u = np.arange(-d, d + 1)
        m10 += int((u * row).sum())
        m01 += v * int(row.sum())
    return fast_atan2(np.float32(m01), np.float32(m10))


def orb_descriptor(blurred, x, y, angle, pattern):
    f = np.float32
    cx, cy = int(np.rint(f(x))), int(np.rint(f(y)))
    theta = f(f(angle) * f(np.pi / f(180.0)))
    a, b = f(np.cos(np.float64(theta))), f(np.sin(np.float64(theta)))
    pat = np.asarray(pattern, np.int64).reshape(512, 2).astype(np.float32)
    px, py = pat[:, 0], pat[:, 1]
    dy = np.rint((px * b).astype(f) + (py * a).astype(f)).astype(np.int64)
    dx = np.rint((px * a).astype(f) - (py * b).astype(f)).astype(np.int64)
    vals = blurred[cy + dy, cx + dx].astype(np.int64).reshape(256, 2)
    bits = (vals[:, 0] < vals[:, 1]).astype(np.uint8)
    return np.packbits(bits.reshape(32, 8), axis=1, bitorder="little").ravel()


def hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def distribute_octtree(cands, min_x, max_x, min_y, max_y, n_target):
    """SURVEY.md §8a.1 R1 step 3, with tie-break D1 (creation order stands in for the node address).
    cands: rows (x, y, response).  Returns indices of the kept candidates in final list order."""
    import math
    W, Hh = max_x - min_x, max_y - min_y
    n_ini = int(math.floor(np.float32(W) / np.float32(Hh) + np.float32(0.5)))      # round() of a positive float
    if n_ini < 1:
        return []
    hx = np.float32(W) / np.float32(n_ini)
    seq = [0]

    def node(x0, x1, y0, y1, pts):
        seq[0] += 1
        return dict(x0=x0, x1=x1, y0=y0, y1=y1, pts=pts, seq=seq[0])

    def split(nd):
        mx = nd["x0"] + int(math.ceil((nd["x1"] - nd["x0"]) / 2.0))
        my = nd["y0"] + int(math.ceil((nd["y1"] - nd["y0"]) / 2.0))
        quads = [[], [], [], []]
        for i in nd["pts"]:
            quads[(0 if cands[i][0] < mx else 1) + (0 if cands[i][1] < my else 2)].append(i)
        b = [(nd["x0"], mx, nd["y0"], my), (mx, nd["x1"], nd["y0"], my), (nd["x0"], mx, my, nd["y1"]), (mx, nd["x1"], my, nd["y1"])]
        return [node(*b[q], quads[q]) for q in range(4) if quads[q]]

    roots = [node(int(hx * np.float32(i)), int(hx * np.float32(i + 1)), 0, Hh, []) for i in range(n_ini)]
    for i, c in enumerate(cands):
        roots[int(np.float32(c[0]) / hx)]["pts"].append(i)
    nodes = [r for r in roots if r["pts"]]                  # front ... back
    fresh = []
    done = False
    while not done:
        prev = len(nodes)
        new_front, keep, fresh = [], [], []
        for nd in nodes:
            if len(nd["pts"]) == 1:
                keep.append(nd)
                continue
            ch = split(nd)
            new_front = ch[::-1] + new_front               # each child is pushed to the front in turn
            fresh += [c for c in ch if len(c["pts"]) > 1]
        nodes = new_front + keep
        if len(nodes) >= n_target or len(nodes) == prev:
            done = True
        elif len(nodes) + 3 * len(fresh) > n_target:
            while not done:
                prev = len(nodes)
                order = sorted(fresh, key=lambda n: (len(n["pts"]), n["seq"]))
                fresh = []
                for nd in reversed(order):
                    ch = split(nd)
                    idx = next(k for k, m in enumerate(nodes) if m is nd)
                    nodes.pop(idx)
                    nodes = ch[::-1] + nodes
                    fresh += [c for c in ch if len(c["pts"]) > 1]
                    if len(nodes) >= n_target:
                        break
                if len(nodes) >= n_target or len(nodes) == prev:
                    done = True
    out = []
    for nd in nodes:
        best = nd["pts"][0]
        for i in nd["pts"][1:]:
            if cands[i][2] > cands[best][2]:
                best = i
        out.append(best)
    return out
