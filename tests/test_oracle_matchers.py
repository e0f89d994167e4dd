"""CPU checks of the matcher half of the oracle: an independent python restatement of the projection search on a small
scene, known answers for ComputeThreeMaxima / the rotation histogram quirk, and brute-force 2-NN against numpy."""
import numpy as np

import oracle
import pyref
import scenes


def three_maxima(counts):
    """ComputeThreeMaxima (MatchCriteria.cpp:727-767) on a list of bin sizes."""
    m1 = m2 = m3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(counts):
        if s > m1:
            m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
        elif s > m2:
            m3, m2, i3, i2 = m2, s, i2, i
        elif s > m3:
            m3, i3 = s, i
    if m2 < np.float32(0.1) * np.float32(m1):
        i2 = i3 = -1
    elif m3 < np.float32(0.1) * np.float32(m1):
        i3 = -1
    return [i for i in (i1, i2, i3) if i >= 0]


def test_rotation_histogram_uses_only_13_bins():
    # factor = 1/30 and bin = round(rot*factor): rot in [0,360) lands in bins 0..12
    a = np.zeros(360, np.float32)
    b = np.arange(360, dtype=np.float32)
    keep = oracle.rotation_consistency(a, b)          # rot = b - a
    bins = np.floor(b * np.float32(1.0 / 30) + np.float32(0.5)).astype(int)      # round half away from zero, positive values
    assert bins.max() == 12
    counts = np.bincount(bins, minlength=30)
    top3 = three_maxima(counts)
    assert np.array_equal(keep, np.isin(bins, top3))


def test_three_maxima_ten_percent_rule():
    # 100 matches at rot 0, 9 at rot 90, 9 at rot 180: second/third < 10 % of the first -> only the first bin survives
    a = np.zeros(118, np.float32)
    b = np.concatenate([np.zeros(100), np.full(9, 90), np.full(9, 180)]).astype(np.float32)
    keep = oracle.rotation_consistency(a, b)
    assert keep[:100].all() and not keep[100:].any()
    b = np.concatenate([np.zeros(100), np.full(10, 90), np.full(9, 180)]).astype(np.float32)
    keep = oracle.rotation_consistency(np.zeros(119, np.float32), b)
    assert keep[:110].all() and not keep[110:].any()
    # negative rotations wrap by +360
    keep = oracle.rotation_consistency(np.full(4, 350, np.float32), np.array([340, 341, 100, 342], np.float32))
    assert keep.tolist() == [True, True, True, True]


def test_knn2_against_numpy():
    rng = np.random.default_rng(2)
    q = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (130, 32), dtype=np.uint8)
    t[5] = t[90] = q[3]                                  # exact duplicate: first index must win, second distance = 0
    bi, bd, sd = oracle.hamming_knn2(q, t)
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    assert np.array_equal(bi, D.argmin(1)) and np.array_equal(bd, D.min(1))
    assert np.array_equal(sd, np.sort(D, 1)[:, 1]) and bi[3] == 5 and sd[3] == 0
    bi, bd, sd = oracle.hamming_knn2(q[:2], t[:1])
    assert bi.tolist() == [0, 0] and sd.tolist() == [-1, -1]


def py_search_by_projection(fa, lms, pp):
    """Independent restatement (per landmark, brute force over keypoints) of SURVEY.md §8a.1 R3."""
    f32 = np.float32
    Rcw, tcw = np.asarray(fa["Rcw"], f32), np.asarray(fa["tcw"], f32)
    Ow = (-(Rcw.T @ tcw)).astype(f32)
    kps, desc, uR, obs = fa["kps"], fa["desc"], fa["uR"], fa["kp_lm_obs"]
    minx, maxx, miny, maxy = (f32(v) for v in fa["bounds"])
    invW, invH = f32(64) / (maxx - minx), f32(48) / (maxy - miny)
    gx = np.round((kps["x"] - minx) * invW).astype(int)            # numpy rounds half to even; coordinates never sit on .5 cells here
    gy = np.round((kps["y"] - miny) * invH).astype(int)
    ingrid = (gx >= 0) & (gx < 64) & (gy >= 0) & (gy < 48)

    def proj(P):
        Pc = (Rcw.astype(np.float64) @ P.astype(np.float64) + tcw.astype(np.float64)).astype(f32)
        z = Pc[2]
        xh, yh = f32(Pc[0] / z), f32(Pc[1] / z)
        u = f32(np.float64(fa["fx"]) * np.float64(xh) + np.float64(fa["cx"]) * np.float64(f32(z / z)))
        v = f32(np.float64(fa["fy"]) * np.float64(yh) + np.float64(fa["cy"]) * np.float64(f32(z / z)))
        ur = f32(u - f32(f32(fa["mbf"]) * f32(f32(1) / z))) if fa["sensor"] == 1 else f32(-1)
        return u, v, ur, bool(z > 0 and minx <= u <= maxx and miny <= v <= maxy)

    out = {}
    for li, lm in enumerate(lms):
        if lm["skip"]:
            continue
        with np.errstate(all="ignore"):
            u, v, ur, ok = proj(lm["pos"])
        if not ok:
            continue
        if pp.use_distance:
            dist = f32(np.sqrt(((lm["pos"] - Ow).astype(np.float64) ** 2).sum()))
            if dist < f32(0.8) * lm["min_dist"] or dist > f32(1.2) * lm["max_dist"]:
                continue
        if pp.use_viewing_angle:
            po = (lm["pos"] - Ow).astype(f32)
            dist = f32(np.sqrt((po.astype(np.float64) ** 2).sum()))
            pon = (po.astype(np.float64) * (1.0 / np.float64(dist))).astype(f32)
            if not (float((pon.astype(np.float64) * lm["normal"].astype(np.float64)).sum()) > float(f32(np.cos(f32(pp.max_view_angle))))):
                continue
        if lm["assoc_kp"] >= 0:
            size = kps["size"][lm["assoc_kp"]]
        else:
            half = f32(lm["size"] / f32(2))
            size = f32(proj(lm["pos"] + np.array([half, 0, 0], f32))[0] - proj(lm["pos"] - np.array([half, 0, 0], f32))[0])
        r = f32(f32(f32(pp.th) * size) / f32(31))
        x0 = max(0, int(np.floor(f32(f32(f32(u - minx) - r) * invW)))); x1 = min(63, int(np.ceil(f32(f32(f32(u - minx) + r) * invW))))
        y0 = max(0, int(np.floor(f32(f32(f32(v - miny) - r) * invH)))); y1 = min(47, int(np.ceil(f32(f32(f32(v - miny) + r) * invH))))
        m = ingrid & (gx >= x0) & (gx <= x1) & (gy >= y0) & (gy <= y1) & (np.abs(kps["x"] - u) < r) & (np.abs(kps["y"] - v) < r)
        if pp.use_prev_matched:
            m &= ~(obs > 0)
        if pp.use_reprojection:
            ex, ey = (u - kps["x"]).astype(f32), (v - kps["y"]).astype(f32)
            er = np.where(uR >= 0, (ur - uR).astype(f32), f32(0)).astype(f32)
            err = ((ex * ex).astype(f32) + (ey * ey).astype(f32)).astype(f32) + (er * er).astype(f32)
            sf = (kps["size"] / f32(31)).astype(f32)
            sigma = (f32(pp.sigma_ref) * (sf * sf).astype(f32)).astype(f32)
            factor = np.where(uR > 0, f32(1.30), f32(1.00)).astype(f32)
            m &= (err / sigma).astype(f32) < (factor * f32(pp.reproj_threshold)).astype(f32)
        m &= (kps["size"] > f32(pp.frac_smaller) * size) & (kps["size"] < f32(pp.frac_larger) * size)
        if pp.use_stereo and fa["sensor"] != 0:
            m &= (np.abs(ur - uR) < r) & (uR > 0)
        cand = np.nonzero(m)[0]
        if len(cand) == 0:
            continue
        cand = cand[np.lexsort((cand, gy[cand], gx[cand]))]           # grid column, then row, then insertion order
        d = np.unpackbits(desc[cand] ^ lm["desc"][None, :], axis=1).sum(1)
        b = int(np.argmin(d))
        second = np.sort(d)[1] if len(d) > 1 else np.finfo(np.float32).max
        if d[b] <= pp.score_threshold and not (f32(d[b]) > f32(pp.second_best_ratio) * f32(second)):
            out[li] = (int(cand[b]), float(d[b]))
    if pp.first_wins:
        seen, keep = set(), {}
        for li in sorted(out):
            if out[li][0] not in seen:
                seen.add(out[li][0]); keep[li] = out[li]
        out = keep
    return out


def test_projection_search_against_python_restatement():
    sc = scenes.projection_scene(21, 320, 240, nfeat=400, copies=2, fx=260.0)
    F, keep = oracle.make_frame_view(oracle.FrameView, **sc["frame_args"])
    for (ud, us, rot, thr, ratio, th) in ((1, 1, 0, 100.0, 0.8, 5.0), (1, 0, 0, 60.0, 1.0, 8.0), (0, 1, 0, 100.0, 0.9, 3.0)):
        pp = oracle.ProjParams(th, thr, ratio, 0.5, 1.5, ud, us, rot)
        midx, mdist, n = oracle.search_by_projection(F, sc["lms"], pp)
        ref = py_search_by_projection(sc["frame_args"], sc["lms"], pp)
        got = {int(i): (int(midx[i]), float(mdist[i])) for i in np.nonzero(midx >= 0)[0]}
        assert n == len(got) > 50
        assert got == ref


def test_fuse_against_python_restatement():
    sc = scenes.projection_scene(24, 320, 240, nfeat=400, copies=3, fx=260.0)
    F, keep = oracle.make_frame_view(oracle.FrameView, **sc["frame_args"])
    pp = oracle.ProjParams(3.0, 50.0, 1.0, 0.5, 1.5, use_distance=1, use_stereo=0, check_rotation=0, use_prev_matched=0,
                           use_viewing_angle=1, max_view_angle=1.047, use_reprojection=1, reproj_threshold=5.99, sigma_ref=1.0, first_wins=1)
    lms = sc["lms"].copy()
    lms["normal"][::7] *= -1                                   # seen from behind: fails the 60 degree viewing-angle test
    midx, mdist, n = oracle.search_by_projection(F, lms, pp)
    ref = py_search_by_projection(sc["frame_args"], lms, pp)
    got = {int(i): (int(midx[i]), float(mdist[i])) for i in np.nonzero(midx >= 0)[0]}
    assert n == len(got) > 50 and got == ref
    assert len(set(v[0] for v in got.values())) == len(got)     # one landmark per keypoint
    assert (midx[::7] < 0).all()


def test_projection_rotation_check_drops_duplicates_and_outlier_bins():
    sc = scenes.projection_scene(22, 320, 240, nfeat=400, copies=3, fx=260.0)
    F, keep = oracle.make_frame_view(oracle.FrameView, **sc["frame_args"])
    base = oracle.search_by_projection(F, sc["lms"], oracle.ProjParams(5.0, 100.0, 0.9, 0.5, 1.5, 0, 1, 0))
    rot = oracle.search_by_projection(F, sc["lms"], oracle.ProjParams(5.0, 100.0, 0.9, 0.5, 1.5, 0, 1, 1))
    m0, m1 = base[0], rot[0]
    assert (m1[m0 < 0] < 0).all() and rot[2] < base[2]
    kept = m1[m1 >= 0]
    assert len(np.unique(kept)) == len(kept) == rot[2]                    # one landmark per keypoint survives ...
    for kp in np.unique(m0[m0 >= 0]):                                      # ... and it is the LAST landmark that matched it
        owners = np.nonzero(m0 == kp)[0]
        assert (m1[owners[:-1]] < 0).all()


def test_bow_core_properties():
    sc = scenes.projection_scene(23, 320, 240, nfeat=400, copies=1, fx=260.0)
    k1, d1 = sc["kps"], sc["desc"]
    rng = np.random.default_rng(4)
    perm = rng.permutation(len(k1))
    k2, d2 = k1[perm].copy(), d1[perm].copy()
    d2[::3, 5] ^= 0x11
    fv1 = scenes.synthetic_featvec(d1, 37, 9)
    fv2 = scenes.synthetic_featvec(d2, 37, 9)
    m, n = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, None, 50.0, 0.9, False)
    inv = np.argsort(perm)
    ok = m >= 0
    assert n == ok.sum() > len(k1) // 2
    assert (m[ok] == inv[ok]).mean() > 0.95
    keep1 = (np.arange(len(k1)) % 2 == 0).astype(np.uint8)
    m2, n2 = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, keep1, 50.0, 0.9, True)
    assert (m2[keep1 == 0] < 0).all() and 0 < n2 <= (keep1 == 1).sum()


def test_legacy_key_frame_bow_matches_each_side2_feature_once():
    """the legacy SearchByBoW(KF1, KF2) (FeatureMatcher.cc:938-1077) against a pure-python restatement written from the reference text: exclusive
    use of key-frame-2 features (vbMatched2), strict < on both thresholds, orientation histogram on angle1 - angle2 with ComputeThreeMaxima"""
    sc = scenes.projection_scene(29, 320, 240, nfeat=500, copies=1, fx=260.0)
    k1, d1 = sc["kps"], sc["desc"]
    rng = np.random.default_rng(6)
    # key frame 2: every feature of key frame 1 twice (a near copy and a noisier one), shuffled: the two copies compete for the same partners
    k2 = np.concatenate([k1, k1]); d2 = np.concatenate([d1, d1]).copy()
    d2[len(k1):, 3] ^= 0x05
    perm = rng.permutation(len(k2)); k2, d2 = k2[perm].copy(), d2[perm].copy()
    k2["angle"] = (k2["angle"] + rng.normal(0, 4, len(k2)) + (rng.random(len(k2)) < 0.2) * 90) % 360
    fv1, fv2 = scenes.synthetic_featvec(d1, 23, 4), scenes.synthetic_featvec(d2, 23, 4)
    keep1 = (rng.random(len(k1)) < 0.9).astype(np.uint8); keep2 = (rng.random(len(k2)) < 0.9).astype(np.uint8)
    for ori in (0, 1):
        m, n = oracle.search_by_bow_legacy(k1, d1, fv1, k2, d2, fv2, keep1, keep2, 50.0, 0.9, ori)
        # ---- restatement
        want = np.full(len(k1), -1, np.int64); taken = np.zeros(len(k2), bool); hist = [[] for _ in range(30)]
        ids2 = {int(v): j for j, v in enumerate(fv2[0])}
        for a, node in enumerate(fv1[0]):
            b = ids2.get(int(node))
            if b is None:
                continue
            for i1 in fv1[2][fv1[1][a]:fv1[1][a + 1]]:
                if not keep1[i1]:
                    continue
                best1 = best2 = float("inf"); bi = -1
                for i2 in fv2[2][fv2[1][b]:fv2[1][b + 1]]:
                    if taken[i2] or not keep2[i2]:
                        continue
                    dist = float(pyref.hamming(d1[i1], d2[i2]))
                    if dist < best1:
                        best2, best1, bi = best1, dist, i2
                    elif dist < best2:
                        best2 = dist
                if best1 < 50.0 and np.float32(best1) < np.float32(0.9) * np.float32(best2 if best2 != float("inf") else 3.4e38):
                    want[i1] = bi; taken[bi] = True
                    if ori:
                        rot = np.float32(k1["angle"][i1]) - np.float32(k2["angle"][bi])
                        if rot < 0:
                            rot = np.float32(rot + np.float32(360.0))
                        bn = int(np.floor(np.float32(rot * np.float32(1.0 / 30)) + 0.5))
                        hist[0 if bn == 30 else bn].append(i1)
        if ori:
            sizes = [len(h) for h in hist]
            order = sorted(range(30), key=lambda i: (-sizes[i], i))
            i1_, i2_, i3_ = order[0], order[1], order[2]
            keepb = {i1_}
            lim = np.float32(0.1) * np.float32(sizes[i1_])                      # `max2 < 0.1f * (float)max1` in float
            if not (np.float32(sizes[i2_]) < lim) and sizes[i2_] > 0:
                keepb.add(i2_)
                if not (np.float32(sizes[i3_]) < lim) and sizes[i3_] > 0:
                    keepb.add(i3_)
            for bn in range(30):
                if bn not in keepb:
                    for i1 in hist[bn]:
                        want[i1] = -1
        assert n == (want >= 0).sum() > 100 and np.array_equal(m, want), ori
        got = m[m >= 0]
        assert len(np.unique(got)) == len(got)                                   # a key-frame-2 feature is used at most once


def test_triangulation_core_epipolar_gate():
    """SearchForTriangulation's core on a rectified pair: F12 of a pure x-translation makes the epipolar distance |y2 - y1|."""
    import hyslam_amd  # noqa: F401  (synthetic stereo generator only)
    from hyslam_amd.synth import synth_stereo_pair
    Limg, Rimg = synth_stereo_pair(25, 320, 240)
    p = oracle.default_params(400)
    k1, d1 = oracle.extract(p, Limg)
    k2, d2 = oracle.extract(p, Rimg)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    fv1, fv2 = scenes.synthetic_featvec(d1, 5, 3), scenes.synthetic_featvec(d2, 5, 3)
    rng = np.random.default_rng(1)
    keep1 = (rng.random(len(k1)) < 0.9).astype(np.uint8)
    keep2 = (rng.random(len(k2)) < 0.9).astype(np.uint8)
    m, n = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, keep1, 90.0, 1.0, False, keep2=keep2, F12=F12)
    ok = m >= 0
    assert n == ok.sum() > 15
    dy = k2["y"][m[ok]] - k1["y"][ok]
    sig = (k2["size"][m[ok]] / np.float32(31)) ** 2
    assert (dy * dy < 3.84 * sig).all() and keep1[ok].all() and keep2[m[ok]].all()
    # brute force restatement per shared node
    node1 = {int(i): fv1[2][fv1[1][k]:fv1[1][k + 1]] for k, i in enumerate(fv1[0])}
    node2 = {int(i): fv2[2][fv2[1][k]:fv2[1][k + 1]] for k, i in enumerate(fv2[0])}
    ref = np.full(len(k1), -1, np.int32)
    for nid in node1:
        if nid not in node2:
            continue
        for i1 in node1[nid]:
            if not keep1[i1]:
                continue
            c = [i2 for i2 in node2[nid] if keep2[i2] and
                 np.float32((k2["y"][i2] - k1["y"][i1]) ** 2) < 3.84 * np.float32((k2["size"][i2] / np.float32(31)) ** 2)]
            if not c:
                continue
            d = np.unpackbits(d2[c] ^ d1[i1][None, :], axis=1).sum(1)
            b = int(np.argmin(d))
            second = np.sort(d)[1] if len(d) > 1 else np.finfo(np.float32).max
            if d[b] < 90 and np.float32(d[b]) < np.float32(second):
                ref[i1] = c[b]
    assert np.array_equal(m, ref)
    m_rot, n_rot = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, keep1, 90.0, 1.0, True, keep2=keep2, F12=F12)
    assert n_rot <= n and ((m_rot == m) | (m_rot == -1)).all()


def test_vocabulary_transform_descends_to_nearest_children():
    """DBoW2 transform on a synthetic 6-ary, 3-level vocabulary against a direct python walk (first minimum wins)."""
    T, keep, n_words = oracle.make_vocab_tree(oracle.VocabTree, 6, 3, 5)
    cb, cc, nd, word, weight = keep
    rng = np.random.default_rng(3)
    desc = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    desc[:50] = nd[rng.integers(1, len(nd), 50)]                      # exact node descriptors: distance 0 somewhere on the way
    for levelsup in (1, 2, 3, 5):
        w, wt, node = oracle.bow_transform(T, desc, levelsup)
        for i in range(len(desc)):
            cur, lvl, nid = 0, 0, 0
            while cc[cur]:
                lvl += 1
                ch = np.arange(cb[cur], cb[cur] + cc[cur])
                d = np.unpackbits(nd[ch] ^ desc[i][None, :], axis=1).sum(1)
                cur = int(ch[np.argmin(d)])
                if lvl == 3 - levelsup:
                    nid = cur
            assert (w[i], node[i]) == (word[cur], nid) and wt[i] == weight[cur]
        assert 0 <= w.min() and w.max() < n_words
    assert (oracle.bow_transform(T, desc, 3)[2] == 0).all() and (oracle.bow_transform(T, desc, 9)[2] == 0).all()     # root when levelsup >= L


def mono_init_scene(seed, n_keep=400):
    """Two mono frames of the same points: frame 2 = frame 1 moved by a few pixels, descriptors with some flipped bits, shuffled."""
    sc = scenes.projection_scene(seed, 320, 240, nfeat=n_keep, copies=1, fx=260.0)
    k1, d1 = sc["kps"], sc["desc"]
    rng = np.random.default_rng(seed)
    perm = rng.permutation(len(k1))
    k2, d2 = k1[perm].copy(), d1[perm].copy()
    k2["x"] += rng.normal(6, 3, len(k2)).astype(np.float32); k2["y"] += rng.normal(-4, 3, len(k2)).astype(np.float32)
    k2["angle"] = (k2["angle"] + rng.normal(0, 5, len(k2)) + (rng.random(len(k2)) < 0.1) * 150) % 360
    for i in range(len(d2)):
        for b in rng.integers(0, 256, rng.integers(0, 25)):
            d2[i, b >> 3] ^= 1 << (b & 7)
    fa = dict(sc["frame_args"]); fa.update(kps=k2, desc=d2, uR=None, kp_lm_obs=None, sensor=0)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    return k1, d1, fa, prev


def test_search_for_initialization_against_python_restatement():
    k1, d1, fa, prev = mono_init_scene(26)
    F2, keep = oracle.make_frame_view(oracle.FrameView, **fa)
    m, prev_out, n = oracle.search_for_initialization(k1, d1, F2, prev, 30, 50.0, 0.9)
    # python restatement: sequential loop, brute-force window (grid cell range of the query + |dx|,|dy| < r), steal-only-if-better
    k2, d2 = fa["kps"], fa["desc"]
    f32 = np.float32
    invW, invH = f32(64) / f32(320), f32(48) / f32(240)
    gx = np.floor((k2["x"] * invW) + f32(0.5)).astype(int); gy = np.floor((k2["y"] * invH) + f32(0.5)).astype(int)
    ingrid = (gx >= 0) & (gx < 64) & (gy >= 0) & (gy < 48)
    owner, odist = {}, {}
    r = f32(30)
    for i1 in range(len(k1)):
        x, y = prev[i1]
        x0 = max(0, int(np.floor(f32(f32(x - r) * invW)))); x1 = min(63, int(np.ceil(f32(f32(x + r) * invW))))
        y0 = max(0, int(np.floor(f32(f32(y - r) * invH)))); y1 = min(47, int(np.ceil(f32(f32(y + r) * invH))))
        c = np.nonzero(ingrid & (gx >= x0) & (gx <= x1) & (gy >= y0) & (gy <= y1) & (np.abs(k2["x"] - x) < r) & (np.abs(k2["y"] - y) < r))[0]
        if len(c) == 0:
            continue
        c = c[np.lexsort((c, gy[c], gx[c]))]
        d = np.unpackbits(d2[c] ^ d1[i1][None, :], axis=1).sum(1)
        ok = np.array([(i2 not in odist) or (dd < odist[i2]) for i2, dd in zip(c.tolist(), d.tolist())], bool)
        c, d = c[ok], d[ok]
        if len(c) == 0:
            continue
        b = int(np.argmin(d)); second = np.sort(d)[1] if len(d) > 1 else np.finfo(np.float32).max
        if d[b] <= 50 and f32(d[b]) < f32(second) * f32(0.9):
            owner[int(c[b])] = i1; odist[int(c[b])] = int(d[b])
    idx2 = np.array(sorted(owner)); i1s = np.array([owner[i] for i in idx2])
    keepm = oracle.rotation_consistency(k2["angle"][idx2], k1["angle"][i1s])           # rot = prev(views1) - curr(views2)
    ref = np.full(len(k1), -1, np.int32); ref[i1s[keepm]] = idx2[keepm]
    assert np.array_equal(m, ref) and n == keepm.sum() > 100
    moved = m >= 0
    assert np.array_equal(prev_out[moved], np.stack([k2["x"][m[moved]], k2["y"][m[moved]]], 1)) and np.array_equal(prev_out[~moved], prev[~moved])
