"""GPU parity of the FeatureMatcher cores (projection search, BoW-grouped search, brute-force 2-NN) against the oracle."""
import numpy as np
import pytest

import oracle
import scenes
import hyslam_amd as HS
from hyslam_amd import _native as N

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher(gpu):
    return HS.FeatureMatcher(HS.FeatureMatcherSettings(nnratio=0.8), HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=500)))


def both_views(sc):
    Fo, k1 = oracle.make_frame_view(oracle.FrameView, **sc["frame_args"])
    Fg, k2 = oracle.make_frame_view(N.FrameView, **sc["frame_args"])
    return Fo, Fg, (k1, k2)


@pytest.mark.parametrize("sensor", [1, 0])
def test_projection_variants_small(matcher, sensor):
    sc = scenes.projection_scene(31, 640, 480, nfeat=1000, copies=3, sensor=sensor)
    Fo, Fg, keep = both_views(sc)
    lms = sc["lms"]
    for name, call, pp in (
        ("local map", lambda: matcher.SearchByProjection(Fg, lms, 5.0), oracle.ProjParams(5.0, 100.0, 0.8, 0.5, 1.5, 1, 1, 0)),
        ("last frame", lambda: matcher.SearchByProjectionLastFrame(Fg, lms, 7.0), oracle.ProjParams(7.0, 100.0, 0.8, 0.5, 1.5, 0, 1, 1)),
        ("keyframe", lambda: matcher.SearchByProjectionKeyFrame(Fg, lms, 4.0, 70), oracle.ProjParams(4.0, 70.0, 1.0, 0.5, 1.5, 1, 0, 0)),
    ):
        gi, gd, gn = call()
        oi, od, on = oracle.search_by_projection(Fo, lms, pp)
        assert on > 100, name
        assert np.array_equal(gi, oi) and np.array_equal(gd, od) and gn == on, name


def test_c4_projection_50k_landmarks_1080p(matcher):
    """BASELINE config 4 shape: one 1920x1080 / 2000-feature frame against a 50 000-point local map, th = 5, nnratio 0.8."""
    sc = scenes.projection_scene(33, 1920, 1080, nfeat=2000, copies=25, fx=1050.0)
    assert len(sc["lms"]) > 50000
    Fo, Fg, keep = both_views(sc)
    gi, gd, gn = matcher.SearchByProjection(Fg, sc["lms"], 5.0)
    oi, od, on = oracle.search_by_projection(Fo, sc["lms"], oracle.ProjParams(5.0, 100.0, 0.8, 0.5, 1.5, 1, 1, 0))
    assert on > 5000 and gn == on and np.array_equal(gi, oi) and np.array_equal(gd, od)
    gi, gd, gn = matcher.SearchByProjectionLastFrame(Fg, sc["lms"], 5.0)
    oi, od, on = oracle.search_by_projection(Fo, sc["lms"], oracle.ProjParams(5.0, 100.0, 0.8, 0.5, 1.5, 0, 1, 1))
    assert gn == on and np.array_equal(gi, oi) and np.array_equal(gd, od)


def test_fuse(matcher):
    sc = scenes.projection_scene(39, 640, 480, nfeat=1000, copies=4)
    Fo, Fg, keep = both_views(sc)
    lms = sc["lms"].copy()
    lms["normal"][::5] *= -1
    lms["skip"][::11] = 1                                      # bad / already in the keyframe / protected (FeatureMatcher.cc:480-485)
    gi, gd, gn = matcher.Fuse(Fg, lms, 3.0, 5.99)
    pp = oracle.ProjParams(3.0, 50.0, 1.0, 0.5, 1.5, use_distance=1, use_stereo=0, check_rotation=0, use_prev_matched=0,
                           use_viewing_angle=1, max_view_angle=1.047, use_reprojection=1, reproj_threshold=5.99, sigma_ref=1.0, first_wins=1)
    oi, od, on = oracle.search_by_projection(Fo, lms, pp)
    assert on > 200 and gn == on and np.array_equal(gi, oi) and np.array_equal(gd, od)
    kept = gi[gi >= 0]
    assert len(np.unique(kept)) == len(kept)


def test_projection_edge_cases(matcher):
    sc = scenes.projection_scene(35, 320, 240, nfeat=300, copies=1, fx=260.0)
    Fo, Fg, keep = both_views(sc)
    gi, gd, gn = matcher.SearchByProjection(Fg, sc["lms"][:0], 5.0)          # no landmarks
    assert len(gi) == 0 and gn == 0
    lms = sc["lms"].copy()
    lms["skip"] = 1                                                            # all nullptr
    gi, gd, gn = matcher.SearchByProjection(Fg, lms, 5.0)
    assert (gi == -1).all() and gn == 0
    fa = dict(sc["frame_args"]); fa["kps"] = fa["kps"][:0]; fa["desc"] = fa["desc"][:0]; fa["uR"] = fa["uR"][:0]; fa["kp_lm_obs"] = fa["kp_lm_obs"][:0]
    Fg0, k0 = oracle.make_frame_view(N.FrameView, **fa)                         # frame without keypoints
    gi, gd, gn = matcher.SearchByProjection(Fg0, sc["lms"], 5.0)
    assert (gi == -1).all() and gn == 0


def test_bow_grouped_search(matcher):
    sc = scenes.projection_scene(37, 640, 480, nfeat=1000, copies=1)
    k1, d1 = sc["kps"], sc["desc"]
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(k1))
    k2, d2 = k1[perm].copy(), d1[perm].copy()
    d2[::2, 7] ^= 0x3C
    k2["angle"] = (k2["angle"] + rng.normal(0, 3, len(k2)) + (rng.random(len(k2)) < 0.15) * 120) % 360
    for nodes in (61, 500):
        fv1, fv2 = scenes.synthetic_featvec(d1, nodes, 11), scenes.synthetic_featvec(d2, nodes, 11)
        keep1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
        for kp, rot in ((None, False), (keep1, True)):
            gm, gn = matcher.SearchByBoW(k1, d1, fv1, k2, d2, fv2, kp, rot)
            om, on = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, kp, 50.0, 0.8, rot)
            assert on > 100 and gn == on and np.array_equal(gm, om), (nodes, rot)
    # disjoint vocabularies: nothing matches
    fv2 = (fv2[0] + 1, fv2[1], fv2[2])
    gm, gn = matcher.SearchByBoW(k1, d1, fv1, k2, d2, fv2, None, True)
    assert gn == 0 and (gm == -1).all()


def test_legacy_key_frame_bow(matcher):
    """hs_search_by_bow_legacy — the legacy SearchByBoW(KF1, KF2) (FeatureMatcher.cc:938-1077): every key-frame-2 feature matched at most once
    (sequential inside a node), orientation histogram on angle1 - angle2"""
    sc = scenes.projection_scene(39, 640, 480, nfeat=1000, copies=1)
    k1, d1 = sc["kps"], sc["desc"]
    rng = np.random.default_rng(7)
    k2 = np.concatenate([k1, k1]); d2 = np.concatenate([d1, d1]).copy()
    d2[len(k1):, 3] ^= 0x05                                                    # two candidates per feature compete for the same partners
    perm = rng.permutation(len(k2)); k2, d2 = k2[perm].copy(), d2[perm].copy()
    k2["angle"] = (k2["angle"] + rng.normal(0, 4, len(k2)) + (rng.random(len(k2)) < 0.2) * 90) % 360
    for nodes in (17, 300):
        fv1, fv2 = scenes.synthetic_featvec(d1, nodes, 21), scenes.synthetic_featvec(d2, nodes, 21)
        keep1 = (rng.random(len(k1)) < 0.9).astype(np.uint8); keep2 = (rng.random(len(k2)) < 0.9).astype(np.uint8)
        for ori, kp1, kp2 in ((True, keep1, keep2), (False, None, None)):
            m = HS.FeatureMatcher(HS.FeatureMatcherSettings(nnratio=0.9, checkOri=ori), matcher._ex)
            gm, gn = m.SearchByBoWLegacy(k1, d1, fv1, k2, d2, fv2, kp1, kp2)
            om, on = oracle.search_by_bow_legacy(k1, d1, fv1, k2, d2, fv2, kp1, kp2, 50.0, 0.9, ori)
            assert on > 200 and gn == on and np.array_equal(gm, om), (nodes, ori)
            got = gm[gm >= 0]
            assert len(np.unique(got)) == len(got)


def test_search_for_triangulation_core(matcher):
    from hyslam_amd.synth import synth_stereo_pair
    matcher = HS.FeatureMatcher(HS.FeatureMatcherSettings(nnratio=0.8, TH_LOW=90.0), matcher._ex)
    Limg, Rimg = synth_stereo_pair(41, 640, 480)
    p = oracle.default_params(1000)
    k1, d1 = oracle.extract(p, Limg)
    k2, d2 = oracle.extract(p, Rimg)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    rng = np.random.default_rng(8)
    for nodes in (7, 90):
        fv1, fv2 = scenes.synthetic_featvec(d1, nodes, 13), scenes.synthetic_featvec(d2, nodes, 13)
        keep1 = (rng.random(len(k1)) < 0.85).astype(np.uint8)
        keep2 = (rng.random(len(k2)) < 0.85).astype(np.uint8)
        gm, gn = matcher.SearchForTriangulation(k1, d1, fv1, k2, d2, fv2, F12, keep1, keep2)
        om, on = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, keep1, 90.0, 1.0, True, keep2=keep2, F12=F12)
        assert on > 15 and gn == on and np.array_equal(gm, om), nodes
    # a general (non-degenerate) fundamental matrix
    F12 = (rng.normal(0, 1, (3, 3)) * np.array([[1e-5, 1e-5, 1e-2], [1e-5, 1e-5, 1e-2], [1e-2, 1e-2, 1.0]])).astype(np.float32)
    gm, gn = matcher.SearchForTriangulation(k1, d1, fv1, k2, d2, fv2, F12, None, None, 31.0, 4.0)
    om, on = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, None, 90.0, 1.0, True, keep2=None, F12=F12, sigma_ref=4.0)
    assert gn == on and np.array_equal(gm, om)


def test_vocabulary_transform_and_bow_pipeline(matcher):
    """Frame::ComputeBoW + SearchByBoW end to end on a synthetic 10-ary vocabulary: transform on the GPU == oracle, then the
    feature vectors it produces drive the BoW-grouped matcher."""
    To, ko, n_words = oracle.make_vocab_tree(oracle.VocabTree, 10, 4, 17)
    Tg, kg, _ = oracle.make_vocab_tree(N.VocabTree, 10, 4, 17)
    sc = scenes.projection_scene(43, 640, 480, nfeat=1000, copies=1)
    k1, d1 = sc["kps"], sc["desc"]
    rng = np.random.default_rng(10)
    perm = rng.permutation(len(k1))
    k2, d2 = k1[perm].copy(), d1[perm].copy()
    d2[::4, 3] ^= 0x81
    voc = HS.ORBVocabulary(Tg, matcher._ex)
    for levelsup in (2, 4):
        bow1, fv1, raw1 = voc.transform(d1, levelsup)
        bow2, fv2, raw2 = voc.transform(d2, levelsup)
        for raw, d in ((raw1, d1), (raw2, d2)):
            ow, owt, ond = oracle.bow_transform(To, d, levelsup)
            assert np.array_equal(raw[0], ow) and np.array_equal(raw[1], owt) and np.array_equal(raw[2], ond)
        assert abs(sum(bow1.values()) - 1.0) < 1e-5 and len(fv1[0]) == len(np.unique(raw1[2]))
        gm, gn = matcher.SearchByBoW(k1, d1, fv1, k2, d2, fv2, None, True)
        om, on = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, None, 50.0, 0.8, True)
        assert gn == on and np.array_equal(gm, om)
    assert gn > 300                                                     # levelsup 4 = root node: one big group, most points re-found


def test_search_for_initialization(matcher):
    from test_oracle_matchers import mono_init_scene
    for seed, window in ((51, 100), (52, 20)):
        k1, d1, fa, prev = mono_init_scene(seed, 1000)
        Fo, ko = oracle.make_frame_view(oracle.FrameView, **fa)
        Fg, kg = oracle.make_frame_view(N.FrameView, **fa)
        gm, gprev, gn = matcher.SearchForInitialization(k1, d1, Fg, prev, window)
        om, oprev, on = oracle.search_for_initialization(k1, d1, Fo, prev, window, 50.0, 0.8)
        assert on > 100 and gn == on and np.array_equal(gm, om) and np.array_equal(gprev, oprev), (seed, window)


def test_knn2(matcher):
    rng = np.random.default_rng(6)
    q = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (2003, 32), dtype=np.uint8)
    t[100:600] = q[500:1000]
    t[700] = t[100]                                                              # duplicate: first index wins, second distance 0
    g = matcher.HammingKnn2(q, t)
    o = oracle.hamming_knn2(q, t)
    for a, b in zip(g, o):
        assert np.array_equal(a, b)
    assert g[0][500] == 100 and g[1][500] == 0 and g[2][500] == 0
    g = matcher.HammingKnn2(q[:3], t[:1])
    assert g[0].tolist() == [0, 0, 0] and g[2].tolist() == [-1, -1, -1]
    g = matcher.HammingKnn2(q[:3], t[:0])
    assert g[0].tolist() == [-1, -1, -1]


def sim3_scene(seed):
    """two keyframes looking at the same landmarks: KF2 = KF1 moved by a small Sim3 (scale 1.1); every keypoint of each keyframe owns a landmark"""
    sc = scenes.projection_scene(seed, 640, 480, nfeat=800, copies=1)
    fa = sc["frame_args"]
    n = len(fa["kps"])
    rng = np.random.default_rng(seed)
    lms = sc["lms"][:0].copy()
    # landmarks of KF1's keypoints: back-project at seeded depths with the keyframe's own pose
    fx, cx, cy = fa["fx"], fa["cx"], fa["cy"]
    Rcw, tcw = np.asarray(fa["Rcw"], np.float64), np.asarray(fa["tcw"], np.float64)
    d = rng.uniform(3.0, 20.0, n)
    Pc = np.stack([(fa["kps"]["x"] - cx) * d / fx, (fa["kps"]["y"] - cy) * d / fx, d], 1)
    Pw = (Rcw.T @ (Pc - tcw).T).T
    l1 = np.zeros(n, oracle.LM_DTYPE)
    l1["pos"] = Pw.astype(np.float32); l1["size"] = (fa["kps"]["size"] * d / fx).astype(np.float32)
    dist = np.linalg.norm(Pw + (Rcw.T @ tcw), axis=1)
    l1["min_dist"] = (0.8 * dist * rng.uniform(0.4, 1.15, n)).astype(np.float32)        # invariance range; some landmarks fall outside
    l1["max_dist"] = (1.2 * dist * rng.uniform(0.9, 2.5, n)).astype(np.float32)
    l1["normal"] = ((Pw + (Rcw.T @ tcw)) / dist[:, None]).astype(np.float32)
    l1["desc"] = fa["desc"]; l1["assoc_kp"] = -1
    l1["skip"][rng.choice(n, n // 12, replace=False)] = 1
    return sc, fa, l1


def test_sim3_projection_and_search(matcher):
    sc, fa, l1 = sim3_scene(81)
    n = len(fa["kps"])
    KFo, k1 = oracle.make_frame_view(oracle.FrameView, **fa)
    KFg, k2 = oracle.make_frame_view(N.FrameView, **fa)
    rng = np.random.default_rng(3)
    # ---- SearchByProjection(pKF, Scw, ...): Scw = 1.07 * [R|t] close to the keyframe's pose; landmarks in shuffled order so that several compete
    #      for the same keypoint (sequential semantics), some keypoints already matched
    s = np.float32(1.07)
    Scw = np.eye(4, dtype=np.float32)
    Scw[:3, :3] = s * scenes.small_rotation(0.002, -0.003, 0.001) @ np.asarray(fa["Rcw"], np.float32)
    Scw[:3, 3] = s * (np.asarray(fa["tcw"], np.float32) + np.array([0.01, -0.02, 0.015], np.float32))
    lms = np.concatenate([l1, l1[rng.choice(n, n // 2, replace=False)]])               # duplicates: the second copy must lose its keypoint
    lms = lms[rng.permutation(len(lms))]
    taken = (rng.random(n) < 0.1).astype(np.uint8)
    gi, gt, gn = matcher.SearchByProjectionSim3(KFg, Scw, lms, taken, 4)
    oi, ot, on = oracle.search_by_projection_sim3(KFo, Scw, lms, 4, 50.0, taken)
    assert on > 150 and gn == on and np.array_equal(gi, oi) and np.array_equal(gt, ot)
    m = gi[gi >= 0]
    assert len(np.unique(m)) == len(m) and not taken[m].any()                            # one landmark per keypoint, never a pre-matched one
    # ---- SearchBySim3: KF2 sees the same landmarks from a pose related by [s12 R12 | t12]
    fb = dict(fa)
    perm = rng.permutation(n)
    fb["kps"] = fa["kps"][perm].copy(); fb["desc"] = fa["desc"][perm].copy(); fb["uR"] = fa["uR"][perm].copy(); fb["kp_lm_obs"] = fa["kp_lm_obs"][perm].copy()
    fb["kps"]["x"] += rng.normal(0, 0.7, n).astype(np.float32); fb["kps"]["y"] += rng.normal(0, 0.7, n).astype(np.float32)
    fb["desc"][::3, 11] ^= 0x18
    s12 = 1.0 / 1.1
    R12 = scenes.small_rotation(0.001, 0.002, -0.001)
    t12 = np.array([0.02, -0.01, 0.03], np.float32)
    # KF2 pose: x_c1 = s12 R12 x_c2 + t12  =>  T2w = (1/s12) R12^T (T1w - t12)
    R2w = (R12.T @ np.asarray(fa["Rcw"], np.float32)).astype(np.float32)
    t2w = ((R12.T @ (np.asarray(fa["tcw"], np.float32) - t12)) / np.float32(s12)).astype(np.float32)
    fb["Rcw"], fb["tcw"] = R2w, t2w
    K2o, k3 = oracle.make_frame_view(oracle.FrameView, **fb)
    K2g, k4 = oracle.make_frame_view(N.FrameView, **fb)
    l2 = l1[perm].copy()
    l2["pos"] = (l2["pos"] * np.float32(1.0))                                           # the same world points, owned by KF2's keypoints
    l2["skip"] = 0; l2["skip"][rng.choice(n, n // 10, replace=False)] = 1
    l1b = l1.copy(); l1b["assoc_kp"] = -1
    gm, gn = matcher.SearchBySim3(KFg, l1b, K2g, l2, s12, R12, t12, 7.5)
    om, on = oracle.search_by_sim3(KFo, l1b, K2o, l2, s12, R12, t12, 7.5, 100.0)
    assert on > 100 and gn == on and np.array_equal(gm, om)


def test_frame_grid(matcher):
    """Frame::AssignFeaturesToGrid / PosInGrid (row M7): cells of every keypoint, including keypoints on the borders (round, not floor: the last
    column / row rounds to 64 / 48 = outside) and outside the bounds"""
    import ctypes as C
    sc = scenes.projection_scene(45, 640, 480, nfeat=1000, copies=1)
    fa = dict(sc["frame_args"])
    k = fa["kps"].copy()
    k["x"][:6] = [0.0, 639.9, 4.99, 5.0, 634.9, 700.0]          # 4.99 * 0.1 rounds to 0, 5.0 * 0.1 = 0.5 rounds away from zero to 1 (std::round)
    k["y"][:6] = [0.0, 479.9, 475.1, 5.0, -3.0, 10.0]
    fa["kps"] = k
    Fo, k1 = oracle.make_frame_view(oracle.FrameView, **fa)
    Fg, k2 = oracle.make_frame_view(N.FrameView, **fa)
    want = oracle.frame_grid(Fo)
    got = np.zeros((len(k), 2), np.int8)
    ex = matcher._ex
    N.check(ex._h, ex._lib.hs_frame_grid(ex._h, C.byref(Fg), got.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(got.astype(np.int32), want)
    assert (want[:6] == -1).any() and (want >= -1).all() and want[:, 0].max() <= 63 and want[:, 1].max() <= 47


def test_device_resident_frames_through_the_python_mirror(gpu):
    """include/hyslam_amd.h "device-resident frames"  as the Python mirror uses it: an extraction published to the device's frame cache is
    found again by its keypoint array, the stereo matcher and the projection search then run on the device copies — same bits as the host-pointer
    calls and as the oracle; a released frame, or one whose keypoints differ in one bit, is not found and the calls fall back to the host arrays."""
    from hyslam_amd.synth import synth_stereo_pair
    sc = scenes.projection_scene(31, 640, 480, nfeat=1000, copies=3)
    L, R = synth_stereo_pair(31, 640, 480)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=1000))
    (kl, kr), (dl, dr) = ex.extract_batch([L, R], publish=True)
    tl, tr = ex.last_frame_tokens
    assert tl and tr and tl != tr
    assert np.array_equal(kl, sc["kps"]) and np.array_equal(dl, sc["desc"])          # the premise: extraction is bit-exact
    assert ex.find_frame(sc["kps"]) == tl and ex.find_frame(kr) == tr
    off = kl.copy(); off["angle"][7] = np.nextafter(off["angle"][7], np.float32(400))
    assert ex.find_frame(off) == 0 and ex.find_frame(kl[:-1]) == 0
    # stereo matcher: device copies == host arrays
    cam = HS.Camera(500.0, 500.0 * 0.12, 480.0)
    sm = HS.Stereomatcher(kl, kr, dl, dr, cam, extractor=ex)
    sm.computeStereoMatches()
    assert sm.frames_on_device
    u_dev, z_dev = [a.copy() for a in sm.getData()]
    sm_host = HS.Stereomatcher(off, kr, dl, dr, cam, extractor=ex)                   # one bit off: not found -> host path (the angle is not read by the stereo matcher)
    sm_host.computeStereoMatches()
    assert not sm_host.frames_on_device
    assert np.array_equal(u_dev, sm_host.getData()[0]) and np.array_equal(z_dev, sm_host.getData()[1]) and (u_dev >= 0).sum() > 50
    # projection search on the cached frame == host-pointer call == oracle
    m = HS.FeatureMatcher(HS.FeatureMatcherSettings(nnratio=0.8), ex)
    Fo, keep_o = oracle.make_frame_view(oracle.FrameView, **sc["frame_args"])
    Fg, keep_g = oracle.make_frame_view(N.FrameView, **sc["frame_args"])
    oi, od, on = oracle.search_by_projection(Fo, sc["lms"], oracle.ProjParams(5.0, 100.0, 0.8, 0.5, 1.5, 1, 1, 0))
    gi, gd, gn = m.SearchByProjection(Fg, sc["lms"], 5.0)
    assert m.frame_on_device and on > 100 and gn == on and np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert ex.release_frame(tl) and not ex.release_frame(tl) and ex.find_frame(kl) == 0
    gi, gd, gn = m.SearchByProjection(Fg, sc["lms"], 5.0)
    assert not m.frame_on_device and gn == on and np.array_equal(gi, oi) and np.array_equal(gd, od)
    sm.computeStereoMatches()                                                         # the left view is gone from the cache: host arrays
    assert not sm.frames_on_device and np.array_equal(sm.getData()[0], u_dev)
    ex.release_frame(tr)
