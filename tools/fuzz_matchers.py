#!/usr/bin/env python3
"""Randomised matcher parity sweep (runs on the GPU box): projection search in its three Frame variants and Fuse, the BoW-grouped search
with and without the epipolar gate, the legacy exclusive key-frame search, mono initialisation and brute-force 2-NN, on random scenes, sensors, radii, thresholds and ratios,
against the oracle.   python3 tools/fuzz_matchers.py --cases 60 --seed 1     (exit code 1 on the first mismatch)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle                                   # noqa: E402  (tests/oracle.py: the checker)
import scenes                                   # noqa: E402
import hyslam_amd as HS                         # noqa: E402
from hyslam_amd import _native as N             # noqa: E402


STRESS = False                                  # --stress: frames up to 4000 x 3000, up to 8000 keypoints per frame, up to 20 landmark copies (160 000 landmarks)


def one_case(rng, i, ex):
    seed = int(rng.integers(0, 1 << 30))
    if STRESS:
        w, h = int(rng.integers(600, 4001)), int(rng.integers(400, 3001))
        h = min(h, 2 * w - 1)
        nfeat = int(rng.integers(1000, 8000))
        copies = int(rng.integers(3, 21))
    else:
        w, h = int(rng.integers(200, 900)), int(rng.integers(160, 640))
        h = min(h, 2 * w - 1)
        nfeat = int(rng.integers(50, 1500))
        copies = int(rng.integers(1, 6))
    sensor = int(rng.integers(0, 2))
    fx = float(rng.uniform(200, 900))
    nnratio = float(np.float32(rng.choice([0.6, 0.7, 0.8, 0.9, 1.0])))
    th_high, th_low = float(rng.choice([100.0, 80.0, 120.0])), float(rng.choice([50.0, 40.0, 70.0]))
    desc = "case %d: seed %d %dx%d nfeat %d copies %d sensor %d fx %.0f nnratio %.1f" % (i, seed, w, h, nfeat, copies, sensor, fx, nnratio)
    sc = scenes.projection_scene(seed, w, h, nfeat=nfeat, copies=copies, sensor=sensor, fx=fx)
    if len(sc["kps"]) == 0:
        return desc + " -> no keypoints", True
    m = HS.FeatureMatcher(HS.FeatureMatcherSettings(nnratio=nnratio, TH_HIGH=th_high, TH_LOW=th_low), ex)
    Fo, k1 = oracle.make_frame_view(oracle.FrameView, **sc["frame_args"])
    Fg, k2 = oracle.make_frame_view(N.FrameView, **sc["frame_args"])
    lms = sc["lms"].copy()
    if rng.random() < 0.3:
        lms["skip"][rng.random(len(lms)) < 0.2] = 1
    th = float(rng.choice([1.0, 3.0, 5.0, 7.0, 15.0]))
    res = []
    gi, gd, gn = m.SearchByProjection(Fg, lms, th)
    oi, od, on = oracle.search_by_projection(Fo, lms, oracle.ProjParams(th, th_high, nnratio, 0.5, 1.5, 1, 1, 0))
    res.append(("map %d" % on, gn == on and np.array_equal(gi, oi) and np.array_equal(gd, od)))
    gi, gd, gn = m.SearchByProjectionLastFrame(Fg, lms, th)
    oi, od, on = oracle.search_by_projection(Fo, lms, oracle.ProjParams(th, th_high, nnratio, 0.5, 1.5, 0, 1, 1))
    res.append(("last %d" % on, gn == on and np.array_equal(gi, oi) and np.array_equal(gd, od)))
    orb_dist = int(rng.choice([50, 70, 100]))
    gi, gd, gn = m.SearchByProjectionKeyFrame(Fg, lms, th, orb_dist)
    oi, od, on = oracle.search_by_projection(Fo, lms, oracle.ProjParams(th, float(orb_dist), 1.0, 0.5, 1.5, 1, 0, 0))
    res.append(("kf %d" % on, gn == on and np.array_equal(gi, oi) and np.array_equal(gd, od)))
    lf = lms.copy()
    lf["normal"][:: int(rng.integers(2, 9))] *= -1
    rth = float(rng.choice([5.99, 7.8, 2.0]))
    gi, gd, gn = m.Fuse(Fg, lf, th, rth)
    pp = oracle.ProjParams(th, th_low, 1.0, 0.5, 1.5, use_distance=1, use_stereo=0, check_rotation=0, use_prev_matched=0,
                           use_viewing_angle=1, max_view_angle=1.047, use_reprojection=1, reproj_threshold=rth, sigma_ref=1.0, first_wins=1)
    oi, od, on = oracle.search_by_projection(Fo, lf, pp)
    res.append(("fuse %d" % on, gn == on and np.array_equal(gi, oi) and np.array_equal(gd, od)))
    # BoW-grouped search between the frame and a shuffled, perturbed copy
    ka, da = sc["kps"], sc["desc"]
    perm = rng.permutation(len(ka))
    kb, db = ka[perm].copy(), da[perm].copy()
    db[:: int(rng.integers(2, 6)), int(rng.integers(0, 32))] ^= int(rng.integers(1, 256))
    kb["angle"] = (kb["angle"] + rng.normal(0, 3, len(kb)) + (rng.random(len(kb)) < 0.15) * 120) % 360
    nodes = int(rng.choice([1, 7, 61, 500, 3000]))
    fva, fvb = scenes.synthetic_featvec(da, nodes, seed & 0xFFFF), scenes.synthetic_featvec(db, nodes, seed & 0xFFFF)
    keep = (rng.random(len(ka)) < 0.8).astype(np.uint8) if rng.random() < 0.5 else None
    rot = bool(rng.integers(0, 2))
    gm, gn = m.SearchByBoW(ka, da, fva, kb, db, fvb, keep, rot)
    om, on = oracle.search_by_bow(ka, da, fva, kb, db, fvb, keep, th_low, nnratio, rot)
    res.append(("bow %d" % on, gn == on and np.array_equal(gm, om)))
    if rng.random() < 0.5:                      # pure sideways translation: epipolar lines are the image rows, the copy's points lie on them
        F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    else:
        F12 = (rng.normal(0, 1, (3, 3)) * np.array([[1e-5, 1e-5, 1e-2], [1e-5, 1e-5, 1e-2], [1e-2, 1e-2, 1.0]])).astype(np.float32)
    keep2 = (rng.random(len(kb)) < 0.85).astype(np.uint8)
    gm, gn = m.SearchForTriangulation(ka, da, fva, kb, db, fvb, F12, keep, keep2)
    om, on = oracle.search_by_bow(ka, da, fva, kb, db, fvb, keep, th_low, 1.0, True, keep2=keep2, F12=F12)
    res.append(("tri %d" % on, gn == on and np.array_equal(gm, om)))
    # the legacy key-frame / key-frame search: side 2 holds every feature twice, so the copies compete for the same partners (exclusive matching)
    kc = np.concatenate([kb, kb]); dc = np.concatenate([db, db]).copy()
    dc[len(kb):, int(rng.integers(0, 32))] ^= int(rng.integers(1, 16))
    pc = rng.permutation(len(kc)); kc, dc = kc[pc].copy(), dc[pc].copy()
    fvc = scenes.synthetic_featvec(dc, nodes, seed & 0xFFFF)
    keepc = (rng.random(len(kc)) < 0.85).astype(np.uint8) if rng.random() < 0.7 else None
    ml = HS.FeatureMatcher(HS.FeatureMatcherSettings(nnratio=nnratio, TH_HIGH=th_high, TH_LOW=th_low, checkOri=rot), ex)
    gm, gn = ml.SearchByBoWLegacy(ka, da, fva, kc, dc, fvc, keep, keepc)
    om, on = oracle.search_by_bow_legacy(ka, da, fva, kc, dc, fvc, keep, keepc, th_low, nnratio, rot)
    res.append(("legacy %d" % on, gn == on and np.array_equal(gm, om)))
    # mono initialisation window search
    prev = np.stack([ka["x"] + rng.normal(0, 6, len(ka)), ka["y"] + rng.normal(0, 6, len(ka))], 1).astype(np.float32)
    window = int(rng.choice([10, 20, 50, 100]))
    gm, gprev, gn = m.SearchForInitialization(ka, da, Fg, prev, window)
    om, oprev, on = oracle.search_for_initialization(ka, da, Fo, prev, window, th_low, nnratio)
    res.append(("init %d" % on, gn == on and np.array_equal(gm, om) and np.array_equal(gprev, oprev)))
    # brute-force 2-NN, ragged sizes including 0 and 1 targets
    nq, nt = int(rng.integers(0, 1200)), int(rng.choice([0, 1, 2, int(rng.integers(3, 2500))]))
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if nq and nt > 2:
        t[: min(nt, nq) // 2] = q[: min(nt, nq) // 2]
    g, o = m.HammingKnn2(q, t), oracle.hamming_knn2(q, t)
    res.append(("knn2 %dx%d" % (nq, nt), all(np.array_equal(a, b) for a, b in zip(g, o))))
    good = all(r[1] for r in res)
    return desc + " -> " + ", ".join(n + ("" if g else " MISMATCH") for n, g in res), good


def records_case(rng, i, ex):
    """BASELINE config 5's device path on random shapes: `world` frame records (odd and even capacities: the padded descriptor offset), random counts
    incl. empty records, descriptors with planted near-duplicates across the ranks; the cross-camera 2-NN of a random rank against every peer and the
    vocabulary-grouped match on a random small vocabulary, both against the oracle"""
    import hipmem
    from hyslam_amd import distributed as D
    from hyslam_amd.synth import synth_vocab_tree
    world = int(rng.integers(1, 9))
    cap = int(rng.integers(40, 2600 if STRESS else 700))
    rb = D.record_bytes(cap)
    base = rng.integers(0, 256, (int(rng.integers(1, 400)), 32), dtype=np.uint8)          # a pool the ranks draw from: real matches across cameras
    feats, host = [], np.zeros((world, rb), np.uint8)
    for r in range(world):
        n = int(rng.choice([0, 1, cap, int(rng.integers(0, cap + 1))]))
        k = np.zeros(n, N.KP_DTYPE)
        k["x"], k["y"], k["angle"], k["octave"] = rng.uniform(0, 1900, n), rng.uniform(0, 1000, n), rng.uniform(0, 360, n), rng.integers(0, 8, n)
        d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        take = rng.random(n) < 0.6
        d[take] = base[rng.integers(0, len(base), int(take.sum()))]
        flip = rng.random(n) < 0.5                                                         # a few bits off: distances around the thresholds
        d[flip, rng.integers(0, 32)] ^= np.uint8(rng.integers(1, 256))
        host[r] = D.pack_record(k, d, cap)
        feats.append((k, d))
    recs = hipmem.DevBuf.from_numpy(host.reshape(-1))
    rank = int(rng.integers(0, world))
    desc = "case %d: records world %d cap %d rank %d counts %s" % (i, world, cap, rank, [len(f[0]) for f in feats])
    res = []
    outs = [hipmem.DevBuf(world * cap * 4) for _ in range(3)]
    D.records_knn2_device(ex, recs.ptr, rb, world, rank, cap, outs[0].ptr, outs[1].ptr, outs[2].ptr, 0)
    ex.synchronize()
    bi, bd, sd = (o.to_numpy(np.int32, world * cap).reshape(world, cap) for o in outs)
    nq, good = len(feats[rank][0]), True
    for peer in range(world):
        if peer == rank or nq == 0 or len(feats[peer][0]) == 0:
            continue
        obi, obd, osd = oracle.hamming_knn2(feats[rank][1], feats[peer][1])
        same = np.array_equal(bi[peer, :nq], obi) and np.array_equal(bd[peer, :nq], obd) and np.array_equal(sd[peer, :nq], osd)
        if not same:                                            # which query, what both sides say (and what a plain numpy scan says)
            q = int(np.nonzero((bi[peer, :nq] != obi) | (bd[peer, :nq] != obd) | (sd[peer, :nq] != osd))[0][0])
            dist = np.unpackbits(feats[rank][1][q][None, :] ^ feats[peer][1], axis=1).sum(1)
            order = np.argsort(dist, kind="stable")
            desc += " [peer %d query %d: gpu (%d, %d, %d) oracle (%d, %d, %d) numpy best %d at %d second %d; %d differing queries]" % (
                peer, q, bi[peer, q], bd[peer, q], sd[peer, q], obi[q], obd[q], osd[q], dist[order[0]], order[0], dist[order[1]] if len(order) > 1 else -1,
                int(((bi[peer, :nq] != obi) | (bd[peer, :nq] != obd) | (sd[peer, :nq] != osd)).sum()))
        good = good and same
    res.append(("knn2", good))
    kk, LL = int(rng.choice([2, 4, 10])), int(rng.choice([2, 3, 4]))
    up = int(rng.integers(0, LL))
    Tg, keep, n_words = synth_vocab_tree(kk, LL, int(rng.integers(0, 1 << 20)))
    To = oracle.VocabTree(Tg.n_nodes, Tg.levels, Tg.child_begin, Tg.child_count, Tg.desc, Tg.word_id, Tg.weight, None)
    try:
        voc = D.DeviceVocabulary(ex, Tg, up, keep)
    except N.HsError as e:                                  # more feature-vector groups than the device search keeps counters for: refused cleanly
        res.append(("bow k%d L%d up%d refused (%s)" % (kk, LL, up, str(e)[-60:]), True))
        return desc + " -> " + ", ".join("%s%s" % (n, "" if g else " MISMATCH") for n, g in res), all(g for _, g in res)
    fvs = [HS.ORBVocabulary.containers(*oracle.bow_transform(To, feats[r][1], up))[1] if len(feats[r][0]) else {} for r in range(world)]
    th_low, ratio, ori = float(rng.choice([50.0, 70.0, 30.0])), float(np.float32(rng.choice([0.6, 0.8, 1.0]))), bool(rng.integers(0, 2))
    d_m, d_nm = hipmem.DevBuf(world * cap * 4), hipmem.DevBuf(world * 4)
    voc.records_bow_match_device(recs.ptr, rb, world, rank, cap, th_low, ratio, ori, d_m.ptr, d_nm.ptr, 0)
    ex.synchronize()
    gm, gn = d_m.to_numpy(np.int32, world * cap).reshape(world, cap), d_nm.to_numpy(np.int32, world)
    good = True
    k1, d1 = feats[rank]
    for peer in range(world):
        if peer == rank:
            good = good and gn[peer] == 0 and bool((gm[peer] == -1).all())
            continue
        k2, d2 = feats[peer]
        if len(k1) == 0 or len(k2) == 0:
            good = good and gn[peer] == 0
            continue
        om, on = oracle.search_by_bow(k1, d1, fvs[rank], k2, d2, fvs[peer], None, th_low, ratio, ori)
        good = good and gn[peer] == on and np.array_equal(gm[peer, :len(k1)], om)
    res.append(("bow k%d L%d up%d" % (kk, LL, up), good))
    voc.close()
    ok = all(g for _, g in res)
    return desc + " -> " + ", ".join("%s%s" % (n, "" if g else " MISMATCH") for n, g in res), ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=1e9)
    ap.add_argument("--stress", action="store_true", help="large frames, thousands of keypoints per frame, up to 160 000 landmarks")
    ap.add_argument("--records", action="store_true", help="only the frame-record cases (BASELINE config 5's device path); otherwise every fifth case is one")
    a = ap.parse_args()
    global STRESS
    STRESS = a.stress
    rng = np.random.default_rng(a.seed)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=500))
    t0, bad = time.time(), 0
    for i in range(a.cases):
        msg, good = records_case(rng, i, ex) if (a.records or i % 5 == 4) else one_case(rng, i, ex)
        print(msg, flush=True)
        bad += not good
        if not good or time.time() - t0 > a.seconds:
            break
    print("fuzz matchers: %d cases, %d mismatches, %.0f s" % (i + 1, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
