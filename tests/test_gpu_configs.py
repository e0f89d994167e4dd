"""BASELINE configurations C3 and C5 and every `*_device` matcher entry point through the C ABI on device-resident buffers
(hipMalloc'ed by tests/hipmem.py, no torch), against the oracle."""
import ctypes as C

import numpy as np
import pytest

import hipmem
import oracle
import scenes
import hyslam_amd as HS
from hyslam_amd import _native as N
from hyslam_amd import distributed as D
from hyslam_amd.synth import synth_image, synth_stereo_pair

pytestmark = pytest.mark.gpu
KB = N.KP_DTYPE.itemsize


def test_c3_batch_64_frames_1080p(gpu):
    """BASELINE config 3: 64 distinct 1920x1080 frames (seeds 100..163) through ONE hs_orb_extract_batch_device call.
    Size-independent properties AND bit-exact oracle parity on all 64 frames."""
    W, H, B, NF = 1920, 1080, 64, 2000
    frames = np.stack([synth_image(100 + i, W, H) for i in range(B)])
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF))
    ex.reserve(W, H, B)
    cap = ex.max_keypoints()
    d_img = hipmem.DevBuf.from_numpy(frames)
    d_k, d_d, d_n = hipmem.DevBuf(B * cap * KB), hipmem.DevBuf(B * cap * 32), hipmem.DevBuf(B * 4)
    ex.extract_batch_device(d_img.ptr, B, W, H, W, W * H, d_k.ptr, d_d.ptr, d_n.ptr, cap, 0)
    ex.synchronize()
    n = d_n.to_numpy(np.int32, B)
    kps = d_k.to_numpy(N.KP_DTYPE, B * cap).reshape(B, cap)
    desc = d_d.to_numpy(np.uint8, B * cap * 32).reshape(B, cap, 32)
    quota, sc = ex.GetFeaturesPerLevel(), ex.GetScaleFactors()
    seen = set()
    for i in range(B):
        k = kps[i, :n[i]]
        assert NF <= n[i] <= NF + 2 * 8, (i, n[i])
        assert (np.diff(k["octave"]) >= 0).all()
        for l in range(8):
            m = k["octave"] == l
            assert quota[l] <= m.sum() <= quota[l] + 2, (i, l)
            lw, lh = np.rint(np.float32(W) / sc[l]), np.rint(np.float32(H) / sc[l])
            x, y = k["x"][m] / sc[l], k["y"][m] / sc[l]
            assert (x > 18.5).all() and (x < lw - 19.5).all() and (y > 18.5).all() and (y < lh - 19.5).all(), (i, l)
        assert (k["response"] >= 19).all() and (k["angle"] >= 0).all() and (k["angle"] < 360).all()
        assert desc[i, :n[i]].any(axis=1).all()
        seen.add(k.tobytes())
    assert len(seen) == B                                      # 64 distinct frames -> 64 distinct results (no slot aliasing inside the batch)
    p = oracle.default_params(NF)
    from concurrent.futures import ThreadPoolExecutor          # ALL 64 frames against the oracle (it holds no GIL: ~3 s on 8 threads)
    with ThreadPoolExecutor(8) as pool:
        ref = list(pool.map(lambda f: oracle.extract(p, f), frames))
    for i, (ok, od) in enumerate(ref):
        assert n[i] == len(ok), i
        assert kps[i, :n[i]].tobytes() == ok.tobytes() and np.array_equal(desc[i, :n[i]], od), i
    # the same batch again through the same workspace: deterministic
    d_k2, d_d2, d_n2 = hipmem.DevBuf(B * cap * KB), hipmem.DevBuf(B * cap * 32), hipmem.DevBuf(B * 4)
    ex.extract_batch_device(d_img.ptr, B, W, H, W, W * H, d_k2.ptr, d_d2.ptr, d_n2.ptr, cap, 0)
    ex.synchronize()
    n2 = d_n2.to_numpy(np.int32, B)
    k2 = d_k2.to_numpy(N.KP_DTYPE, B * cap).reshape(B, cap)
    assert np.array_equal(n, n2) and all(k2[i, :n[i]].tobytes() == kps[i, :n[i]].tobytes() for i in range(B))


def test_c5_records_and_cross_camera_knn2_world1(gpu):
    """BASELINE config 5 at world size 1 with four locally built records: every frame is extracted STRAIGHT into the all-gather record
    layout (count / keypoints / descriptors pointers into one buffer), then hs_records_knn2_device matches one record against the others
    with the counts read on the device.  Checked against unpack_record + oracle.extract + oracle.hamming_knn2."""
    W, H, NF, world = 640, 480, 1000, 4
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF))
    ex.reserve(W, H, 1)
    cap = ex.max_keypoints()
    rb = D.record_bytes(cap)
    assert rb == N.lib().hs_record_bytes(cap)
    o_n, o_k, o_d = D.record_offsets(cap)
    offs = (C.c_size_t(), C.c_size_t(), C.c_size_t())
    N.lib().hs_record_offsets(cap, *(C.byref(o) for o in offs))
    assert (o_n, o_k, o_d) == tuple(o.value for o in offs)
    frames = [synth_image(200 + i, W, H) for i in range(world)]
    frames[2] = frames[0][:, ::-1].copy()                       # a mirrored view: a different count, few cross matches
    recs = hipmem.DevBuf(world * rb)
    for i, f in enumerate(frames):
        d_f = hipmem.DevBuf.from_numpy(f)
        base = recs.ptr + i * rb
        ex.extract_batch_device(d_f.ptr, 1, W, H, W, W * H, base + o_k, base + o_d, base + o_n, cap, 0)
        ex.synchronize()
    host = recs.to_numpy(np.uint8, world * rb).reshape(world, rb)
    p = oracle.default_params(NF)
    feats = []
    for i, f in enumerate(frames):
        k, d = D.unpack_record(host[i], cap)
        ok, od = oracle.extract(p, f)
        assert k.tobytes() == ok.tobytes() and np.array_equal(d, od), i
        assert D.pack_record(k, d, cap)[:16 + 0].tobytes()[:4] == host[i, :4].tobytes()
        feats.append((k, d))
    for rank in (0, 2):
        outs = [hipmem.DevBuf(world * cap * 4) for _ in range(3)]
        for o in outs:
            o.fill(0xEE)
        D.records_knn2_device(ex, recs.ptr, rb, world, rank, cap, outs[0].ptr, outs[1].ptr, outs[2].ptr, 0)
        ex.synchronize()
        bi, bd, sd = (o.to_numpy(np.int32, world * cap).reshape(world, cap) for o in outs)
        nq = len(feats[rank][0])
        for peer in range(world):
            if peer == rank:
                assert (bi[peer] == np.int32(-286331154)).all()        # 0xEEEEEEEE: the own row is left untouched
                continue
            obi, obd, osd = oracle.hamming_knn2(feats[rank][1], feats[peer][1])
            assert np.array_equal(bi[peer, :nq], obi) and np.array_equal(bd[peer, :nq], obd) and np.array_equal(sd[peer, :nq], osd), (rank, peer)
            assert (bi[peer, nq:] == np.int32(-286331154)).all()
    # frame 1 vs frame 0 are different scenes, frame 2 is frame 0 mirrored: descriptors rarely agree; frame 3 vs itself would be all zeros
    # a corrupt count (larger than cap) is clamped on the device, never read out of bounds
    bad = host.copy()
    bad[1, :4] = np.frombuffer(np.int32(10 ** 6).tobytes(), np.uint8)
    d_bad = hipmem.DevBuf.from_numpy(bad)
    outs = [hipmem.DevBuf(world * cap * 4) for _ in range(3)]
    D.records_knn2_device(ex, d_bad.ptr, rb, world, 0, cap, outs[0].ptr, outs[1].ptr, outs[2].ptr, 0)
    ex.synchronize()
    bi = outs[0].to_numpy(np.int32, world * cap).reshape(world, cap)
    assert bi[1].max() < cap


def test_c5_real_shape_8_records_1080p(gpu):
    """BASELINE config 5 at the shape BASELINE names: 8 cameras x 1920x1080, 2000 features, cap = hs_orb_max_keypoints (2012), neighbouring
    cameras overlapping (synth_rig).  Every frame is extracted straight into its all-gather record; then, FOR EVERY RANK 0..7, the cross-camera
    brute-force 2-NN (hs_records_knn2_device) and the vocabulary-grouped BoW match (hs_records_bow_match_device, a synthetic vocabulary of
    ORBvoc's shape: k = 10, L = 6, feature vectors 4 levels up) against all seven peers == oracle, bit for bit."""
    from hyslam_amd.synth import synth_rig, synth_vocab_tree
    W, H, NF, world = 1920, 1080, 2000, 8
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF))
    ex.reserve(W, H, 1)
    cap = ex.max_keypoints()
    rb = D.record_bytes(cap)
    assert cap == 2032 and rb == 113808                                   # 16 + 2032 * (24 + 32): the all-gather message of DESIGN.md §6
    o_n, o_k, o_d = D.record_offsets(cap)
    frames = synth_rig(200, world, W, H)
    recs = hipmem.DevBuf(world * rb)
    for i, f in enumerate(frames):
        d_f = hipmem.DevBuf.from_numpy(f)
        b = recs.ptr + i * rb
        ex.extract_batch_device(d_f.ptr, 1, W, H, W, W * H, b + o_k, b + o_d, b + o_n, cap, 0)
        ex.synchronize()
    host = recs.to_numpy(np.uint8, world * rb).reshape(world, rb)
    p = oracle.default_params(NF)
    feats = []
    for i, f in enumerate(frames):
        k, d = D.unpack_record(host[i], cap)
        ok, od = oracle.extract(p, f)
        assert k.tobytes() == ok.tobytes() and np.array_equal(d, od), i
        feats.append((k, d))
    # ---- brute-force 2-NN, every rank against every peer
    good = np.zeros((world, world), np.int64)
    for rank in range(world):
        outs = [hipmem.DevBuf(world * cap * 4) for _ in range(3)]
        D.records_knn2_device(ex, recs.ptr, rb, world, rank, cap, outs[0].ptr, outs[1].ptr, outs[2].ptr, 0)
        ex.synchronize()
        bi, bd, sd = (o.to_numpy(np.int32, world * cap).reshape(world, cap) for o in outs)
        nq = len(feats[rank][0])
        for peer in range(world):
            if peer == rank:
                continue
            obi, obd, osd = oracle.hamming_knn2(feats[rank][1], feats[peer][1])
            assert np.array_equal(bi[peer, :nq], obi) and np.array_equal(bd[peer, :nq], obd) and np.array_equal(sd[peer, :nq], osd), (rank, peer)
            good[rank, peer] = int(((obd < 50) & (obd < 0.8 * osd)).sum())
    assert all(good[r, r + 1] > 200 for r in range(world - 1)), good          # neighbours share three quarters of their view
    assert good[0, 7] < good[0, 1] // 4                                          # cameras 0 and 7 share nothing
    # ---- vocabulary-grouped match, every rank against every peer
    Tg, keep, n_words = synth_vocab_tree(10, 6, 23)
    To = oracle.VocabTree(Tg.n_nodes, Tg.levels, Tg.child_begin, Tg.child_count, Tg.desc, Tg.word_id, Tg.weight, None)
    voc = D.DeviceVocabulary(ex, Tg, 4, keep)
    assert voc.groups == 100
    fvs = [HS.ORBVocabulary.containers(*oracle.bow_transform(To, feats[i][1], 4))[1] for i in range(world)]
    total = 0
    for rank in range(world):
        d_m, d_nm = hipmem.DevBuf(world * cap * 4), hipmem.DevBuf(world * 4)
        voc.records_bow_match_device(recs.ptr, rb, world, rank, cap, 50.0, 0.8, True, d_m.ptr, d_nm.ptr, 0)
        ex.synchronize()
        gm = d_m.to_numpy(np.int32, world * cap).reshape(world, cap)
        gn = d_nm.to_numpy(np.int32, world)
        k1, d1 = feats[rank]
        for peer in range(world):
            if peer == rank:
                assert gn[peer] == 0 and (gm[peer] == -1).all()
                continue
            k2, d2 = feats[peer]
            om, on = oracle.search_by_bow(k1, d1, fvs[rank], k2, d2, fvs[peer], None, 50.0, 0.8, True)
            assert gn[peer] == on and np.array_equal(gm[peer, :len(k1)], om), (rank, peer)
            total += on
    assert total > 1000
    voc.close()


def _device_frame(fa):
    """FrameView whose pointers are device addresses (+ the buffers that keep them alive)"""
    Fh, keep = oracle.make_frame_view(N.FrameView, **fa)
    bufs = [hipmem.DevBuf.from_numpy(np.ascontiguousarray(fa["kps"], N.KP_DTYPE)), hipmem.DevBuf.from_numpy(np.ascontiguousarray(fa["desc"], np.uint8)),
            hipmem.DevBuf.from_numpy(np.ascontiguousarray(fa["uR"], np.float32)), hipmem.DevBuf.from_numpy(np.ascontiguousarray(fa["kp_lm_obs"], np.int32))]
    Fd = N.FrameView.from_buffer_copy(Fh)
    Fd.kps, Fd.desc, Fd.uR, Fd.kp_lm_obs = (b.ptr for b in bufs)
    return Fd, bufs


@pytest.mark.parametrize("variant", ["local_map", "last_frame", "fuse"])
def test_search_by_projection_device(gpu, variant):
    """hs_search_by_projection_device (FeatureViews resident in HBM) on a caller stream == oracle, for the three criteria sets"""
    sc = scenes.projection_scene(61, 640, 480, nfeat=1000, copies=4)
    Fo, _ = oracle.make_frame_view(oracle.FrameView, **sc["frame_args"])
    Fd, keep = _device_frame(sc["frame_args"])
    lms = sc["lms"]
    L = len(lms)
    args = {"local_map": (5.0, 100.0, 0.8, 0.5, 1.5, 1, 1, 0), "last_frame": (7.0, 100.0, 0.8, 0.5, 1.5, 0, 1, 1)}
    if variant == "fuse":
        kw = dict(use_distance=1, use_stereo=0, check_rotation=0, use_prev_matched=0, use_viewing_angle=1, max_view_angle=1.047,
                  use_reprojection=1, reproj_threshold=5.99, sigma_ref=1.0, first_wins=1)
        pg, po = N.ProjParams(3.0, 50.0, 1.0, 0.5, 1.5, **kw), oracle.ProjParams(3.0, 50.0, 1.0, 0.5, 1.5, **kw)
    else:
        pg, po = N.ProjParams(*args[variant]), oracle.ProjParams(*args[variant])
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=500))
    d_lms = hipmem.DevBuf.from_numpy(lms)
    d_i, d_d, d_n = hipmem.DevBuf(L * 4), hipmem.DevBuf(L * 4), hipmem.DevBuf(4)
    st = hipmem.Stream()
    for _ in range(2):                                           # twice: the per-handle scratch is reused
        N.check(ex._h, ex._lib.hs_search_by_projection_device(ex._h, C.byref(Fd), d_lms.ptr, L, C.byref(pg), d_i.ptr, d_d.ptr, d_n.ptr, st.ptr))
    st.synchronize()
    gi, gd, gn = d_i.to_numpy(np.int32, L), d_d.to_numpy(np.float32, L), int(d_n.to_numpy(np.int32, 1)[0])
    oi, od, on = oracle.search_by_projection(Fo, lms, po)
    assert on > 100 and gn == on and np.array_equal(gi, oi)
    assert np.array_equal(gd[gi >= 0], od[oi >= 0])             # distances of dropped entries are only normalised by the host-pointer wrapper


def test_hamming_knn2_device(gpu):
    rng = np.random.default_rng(16)
    q = rng.integers(0, 256, (1777, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (2001, 32), dtype=np.uint8)
    t[5:900] = q[100:995]
    t[1000] = t[5]
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=500))
    dq, dt = hipmem.DevBuf.from_numpy(q), hipmem.DevBuf.from_numpy(t)
    outs = [hipmem.DevBuf(len(q) * 4) for _ in range(3)]
    st = hipmem.Stream()
    N.check(ex._h, ex._lib.hs_hamming_knn2_device(ex._h, dq.ptr, len(q), dt.ptr, len(t), outs[0].ptr, outs[1].ptr, outs[2].ptr, st.ptr))
    st.synchronize()
    g = [o.to_numpy(np.int32, len(q)) for o in outs]
    o = oracle.hamming_knn2(q, t)
    for a, b in zip(g, o):
        assert np.array_equal(a, b)
    assert g[0][100] == 5 and g[1][100] == 0 and g[2][100] == 0


def test_stereo_match_batch_device(gpu):
    """hs_stereo_match_batch_device on three pairs with different counts laid out like the extractor's outputs (stride cap), then a second
    call with MORE pairs but a smaller cap on the same handle (the strip counters and strip lists grow independently), then batches large enough
    for the two-keypoints-per-wavefront matcher."""
    p = oracle.default_params(1000)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=1000))
    feats = []
    for i, (w, h) in enumerate(((640, 480), (640, 480), (800, 600))):
        L, R = synth_stereo_pair(90 + i, w, h)
        feats.append((oracle.extract(p, L), oracle.extract(p, R), h))

    def run(sel, cap, n_rows):
        P = len(sel)
        kL, kR = np.zeros((P, cap), N.KP_DTYPE), np.zeros((P, cap), N.KP_DTYPE)
        dL, dR = np.zeros((P, cap, 32), np.uint8), np.zeros((P, cap, 32), np.uint8)
        nL, nR = np.zeros(P, np.int32), np.zeros(P, np.int32)
        for j, i in enumerate(sel):
            (a, b), (c, d), _ = feats[i]
            a, b, c, d = a[:cap], b[:cap], c[:cap], d[:cap]
            nL[j], nR[j] = len(a), len(c)
            kL[j, :len(a)], dL[j, :len(a)], kR[j, :len(c)], dR[j, :len(c)] = a, b, c, d
        bufs = [hipmem.DevBuf.from_numpy(x) for x in (kL, dL, nL, kR, dR, nR)]
        d_u, d_z = hipmem.DevBuf(P * cap * 4), hipmem.DevBuf(P * cap * 4)
        sp = N.StereoParams(500.0, 60.0, n_rows, 100.0, 50.0, 31.0)
        ex.stereo_match_batch_device(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, bufs[5].ptr, P, cap, sp, d_u.ptr, d_z.ptr, 0)
        ex.synchronize()
        u, z = d_u.to_numpy(np.float32, P * cap).reshape(P, cap), d_z.to_numpy(np.float32, P * cap).reshape(P, cap)
        osp = oracle.stereo_params(fx=500.0, mbf=60.0, n_rows=n_rows)
        for j in range(P):
            ou, oz, _, _ = oracle.stereo_match(kL[j, :nL[j]], dL[j, :nL[j]], kR[j, :nR[j]], dR[j, :nR[j]], osp)
            assert np.array_equal(u[j, :nL[j]], ou) and np.array_equal(z[j, :nL[j]], oz), (sel, cap, j)
            assert (z[j, :nL[j]] > 0).sum() > 50

    run([2], 1100, 600)                   # 1 pair, large cap, 19 strips
    run([0, 1, 2, 0, 1, 2, 0, 1], 400, 600)   # 8 pairs, small cap: pairs*strips grows while pairs*strips*cap shrinks
    run([0, 1], 1100, 480)
    # pairs * cap >= 16384: the matcher gives TWO left keypoints to a wavefront (k_stereo_match<2>, the bench's 16-pair path); an odd cap and odd
    # counts put a wavefront's second keypoint past the end
    run([0, 1, 2] * 5 + [0], 1100, 600)
    run([2, 0, 1] * 3, 1999, 600)


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_stereo_front_end_nine_pairs_one_call(gpu, fuse, monkeypatch):
    """hs_stereo_frontend_batch_device as the bench drives it — several pairs in ONE call: 9 distinct 640x480 pairs at 2000 features put
    pairs * cap over the matcher's two-keypoints-per-wavefront threshold; every pair's keypoints, descriptors, uRight and depth vs the oracle.
    fuse = 1 (the default): the right keypoints are binned into the strips by an extra workgroup of the describe launch (two stereo launches);
    fuse = 0: a separate k_stereo_strips launch (three).  The call is repeated on the same handle (nothing may be left over between calls), then
    run with one pair (one left keypoint per wavefront) and with identical views (median 0: everything is rejected)."""
    monkeypatch.setenv("HS_STEREO_FUSE", fuse)
    W, H, P = 640, 480, 9
    p = oracle.default_params(2000)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=2000))
    pairs = [synth_stereo_pair(300 + i, W, H) for i in range(P)]
    left = np.stack([a for a, _ in pairs]); right = np.stack([b for _, b in pairs])
    ex.reserve(W, H, 2 * P)
    cap = ex.max_keypoints()
    assert P * cap >= 16384
    dl, dr = hipmem.DevBuf.from_numpy(left), hipmem.DevBuf.from_numpy(right)
    kb, db = cap * N.KP_DTYPE.itemsize, cap * 32
    dk = [hipmem.DevBuf(P * kb) for _ in range(2)]; dd = [hipmem.DevBuf(P * db) for _ in range(2)]; dn = [hipmem.DevBuf(P * 4) for _ in range(2)]
    du, dz = hipmem.DevBuf(P * cap * 4), hipmem.DevBuf(P * cap * 4)
    sp = N.StereoParams(500.0, 60.0, H, 100.0, 50.0, 31.0)
    osp = oracle.stereo_params(fx=500.0, mbf=60.0, n_rows=H)
    ref = [oracle.stereo_frontend(p, osp, L, R) for L, R in pairs]
    ref_same = oracle.stereo_frontend(p, osp, pairs[0][0], pairs[0][0])

    def run(lp, rp, npairs, expect):
        ex.stereo_frontend_batch_device(lp, rp, npairs, W, H, W, W * H, dk[0].ptr, dd[0].ptr, dn[0].ptr, dk[1].ptr, dd[1].ptr, dn[1].ptr, cap, sp, du.ptr, dz.ptr, 0)
        ex.synchronize()
        nL, nR = dn[0].to_numpy(np.int32, P), dn[1].to_numpy(np.int32, P)
        kL = dk[0].to_numpy(N.KP_DTYPE, P * cap).reshape(P, cap); kR = dk[1].to_numpy(N.KP_DTYPE, P * cap).reshape(P, cap)
        dL = dd[0].to_numpy(np.uint8, P * db).reshape(P, cap, 32); dR = dd[1].to_numpy(np.uint8, P * db).reshape(P, cap, 32)
        u, z = du.to_numpy(np.float32, P * cap).reshape(P, cap), dz.to_numpy(np.float32, P * cap).reshape(P, cap)
        matched = 0
        for i, (okL, odL, okR, odR, ou, oz) in enumerate(expect):
            assert nL[i] == len(okL) and nR[i] == len(okR), i
            assert kL[i, :nL[i]].tobytes() == okL.tobytes() and kR[i, :nR[i]].tobytes() == okR.tobytes(), i
            assert np.array_equal(dL[i, :nL[i]], odL) and np.array_equal(dR[i, :nR[i]], odR), i
            assert np.array_equal(u[i, :nL[i]], ou) and np.array_equal(z[i, :nL[i]], oz), i
            matched += int((oz > 0).sum())
        return matched

    assert run(dl.ptr, dr.ptr, P, ref) > 50 * P
    assert run(dl.ptr, dr.ptr, P, ref) > 50 * P                       # again on the same handle
    assert run(dl.ptr, dr.ptr, 1, ref[:1]) > 50                       # one pair: one left keypoint per wavefront
    assert run(dl.ptr, dl.ptr, 1, [ref_same]) == 0                    # identical views: median 0, every match rejected (:146-155)
    assert run(dl.ptr + 3 * W * H, dr.ptr + 3 * W * H, 2, ref[3:5]) > 100


def test_bench_launch_shape_64_pairs_1080p(gpu):
    """bench.py's own default step, in full: 64 stereo pairs (128 frames of 1920x1080, separate copies in HBM of the 4 distinct pairs of rank 0 —
    synth_stereo_pair(1000 + i), pair j = distinct pair j % 4) through ONE hs_stereo_frontend_batch_device call at 2000 features, fx 1050, mbf 126.
    Every one of the 64 pairs' keypoints, descriptors, uRight and depth against oracle.stereo_frontend (4 oracle runs; the other 60 must equal their
    copy's), twice on the same handle, and the committed checksums bench.py verifies itself against (tests/golden/bench_c2_seed1000.json)."""
    import hashlib, json, os
    W, H, B, ND, NF = 1920, 1080, 64, 4, 2000
    pairs = [synth_stereo_pair(1000 + i, W, H) for i in range(ND)]
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF))
    ex.reserve(W, H, 2 * B)
    cap = ex.max_keypoints()
    dl = hipmem.DevBuf.from_numpy(np.stack([pairs[j % ND][0] for j in range(B)]))
    dr = hipmem.DevBuf.from_numpy(np.stack([pairs[j % ND][1] for j in range(B)]))
    kb, db = cap * KB, cap * 32
    dk = [hipmem.DevBuf(B * kb) for _ in range(2)]; dd = [hipmem.DevBuf(B * db) for _ in range(2)]; dn = [hipmem.DevBuf(B * 4) for _ in range(2)]
    du, dz = hipmem.DevBuf(B * cap * 4), hipmem.DevBuf(B * cap * 4)
    sp = HS.stereo_params(HS.Camera(fx=1050.0, mbf=1050.0 * 0.12, mnMaxY=float(H)))
    p = oracle.default_params(NF)
    osp = oracle.stereo_params(fx=1050.0, mbf=1050.0 * 0.12, n_rows=H)
    ref = [oracle.stereo_frontend(p, osp, L, R) for L, R in pairs]
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bench_c2_seed1000.json")))["ranks"]["0"]["pairs"]
    for rep in range(2):
        ex.stereo_frontend_batch_device(dl.ptr, dr.ptr, B, W, H, W, W * H, dk[0].ptr, dd[0].ptr, dn[0].ptr, dk[1].ptr, dd[1].ptr, dn[1].ptr, cap, sp, du.ptr, dz.ptr, 0)
        ex.synchronize()
        nL, nR = dn[0].to_numpy(np.int32, B), dn[1].to_numpy(np.int32, B)
        kL = dk[0].to_numpy(N.KP_DTYPE, B * cap).reshape(B, cap); kR = dk[1].to_numpy(N.KP_DTYPE, B * cap).reshape(B, cap)
        dL = dd[0].to_numpy(np.uint8, B * db).reshape(B, cap, 32); dR = dd[1].to_numpy(np.uint8, B * db).reshape(B, cap, 32)
        u, z = du.to_numpy(np.float32, B * cap).reshape(B, cap), dz.to_numpy(np.float32, B * cap).reshape(B, cap)
        for j in range(B):
            okL, odL, okR, odR, ou, oz = ref[j % ND]
            assert nL[j] == len(okL) and nR[j] == len(okR), (rep, j)
            assert kL[j, :nL[j]].tobytes() == okL.tobytes() and kR[j, :nR[j]].tobytes() == okR.tobytes(), (rep, j)
            assert np.array_equal(dL[j, :nL[j]], odL) and np.array_equal(dR[j, :nR[j]], odR), (rep, j)
            assert np.array_equal(u[j, :nL[j]], ou) and np.array_equal(z[j, :nL[j]], oz), (rep, j)
            assert (oz > 0).sum() > 300, j
            h = hashlib.sha256()
            for a in (kL[j, :nL[j]], dL[j, :nL[j]], kR[j, :nR[j]], dR[j, :nR[j]], u[j, :nL[j]], z[j, :nL[j]]):
                h.update(np.ascontiguousarray(a).tobytes())
            g = gold[j % ND]
            assert h.hexdigest() == g["outputs_sha256"] and nL[j] == g["nL"] and nR[j] == g["nR"], (rep, j)


def test_reserve_failure_leaves_handle_usable(gpu):
    """a failed configure() (absurd batch: allocation failure, or an unsupported geometry) must not leave stale geometry behind: the next
    small extract on the same handle still matches the oracle"""
    img = synth_image(5, 640, 480)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=1000))
    ok, od = oracle.extract(oracle.default_params(1000), img)
    gk, gd = ex(img)
    assert gk.tobytes() == ok.tobytes()
    with pytest.raises(HS.HsError):
        ex.reserve(16384, 16384, 65535)                          # 16 TB of pyramid levels: the first hipMalloc fails at once
    gk, gd = ex(img)                                             # the handle reconfigures instead of reusing freed buffers
    assert gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
    with pytest.raises(HS.HsError):
        ex.reserve(100, 4000, 1)                                 # aspect ratio < 0.5: rejected after the old geometry was released
    gk, gd = ex(img)                                             # SAME frame size as before the failure: must not take the early exit
    assert gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)
    with pytest.raises(HS.HsError):
        ex.reserve(16384, 16384, 65535)
    ex.reserve(640, 480, 3)                                      # a larger batch of the old size after a failure
    ks, ds = ex.extract_batch([img, img[::-1].copy(), img])
    assert ks[0].tobytes() == ok.tobytes() and ks[2].tobytes() == ok.tobytes() and np.array_equal(ds[2], od)


def test_c5_device_vocabulary_and_cross_camera_bow(gpu):
    """BASELINE config 5, BoW variant at world size 1 with four locally built records: vocabulary transform on device-resident descriptors ==
    oracle transform; hs_records_bow_match_device == (oracle transform -> DBoW2 feature vectors -> oracle.search_by_bow) per peer."""
    from hyslam_amd.synth import synth_vocab_tree
    W, H, NF, world = 640, 480, 1000, 4
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF))
    ex.reserve(W, H, 1)
    cap = ex.max_keypoints()
    rb = D.record_bytes(cap)
    o_n, o_k, o_d = D.record_offsets(cap)
    base = synth_image(300, W, H)
    frames = [base, synth_image(301, W, H), base.copy(), base[:, ::-1].copy()]
    rng = np.random.default_rng(4)
    noisy = frames[2].astype(np.int16) + rng.integers(-3, 4, frames[2].shape)          # a second view of frame 0: most features re-found
    frames[2] = np.clip(noisy, 0, 255).astype(np.uint8)
    recs = hipmem.DevBuf(world * rb)
    for i, f in enumerate(frames):
        d_f = hipmem.DevBuf.from_numpy(f)
        b = recs.ptr + i * rb
        ex.extract_batch_device(d_f.ptr, 1, W, H, W, W * H, b + o_k, b + o_d, b + o_n, cap, 0)
        ex.synchronize()
    host = recs.to_numpy(np.uint8, world * rb).reshape(world, rb)
    feats = [D.unpack_record(host[i], cap) for i in range(world)]
    Tg, keep, n_words = synth_vocab_tree(10, 4, 17)
    To = oracle.VocabTree(Tg.n_nodes, Tg.levels, Tg.child_begin, Tg.child_count, Tg.desc, Tg.word_id, Tg.weight, None)
    for levelsup in (2, 3, 4, 6):
        voc = D.DeviceVocabulary(ex, Tg, levelsup, keep)
        assert voc.groups == (10 ** max(4 - levelsup, 0))
        # ---- transform of record 0's descriptors straight from the record
        n0 = len(feats[0][0])
        d_w, d_wt, d_nd = hipmem.DevBuf(cap * 4), hipmem.DevBuf(cap * 4), hipmem.DevBuf(cap * 4)
        voc.transform_device(recs.ptr + o_d, recs.ptr + o_n, cap, d_w.ptr, d_wt.ptr, d_nd.ptr, 0)
        ex.synchronize()
        ow, owt, ond = oracle.bow_transform(To, feats[0][1], levelsup)
        assert np.array_equal(d_w.to_numpy(np.int32, n0), ow) and np.array_equal(d_wt.to_numpy(np.float32, n0), owt) and np.array_equal(d_nd.to_numpy(np.int32, n0), ond)
        # ---- cross-camera BoW match of record `rank` against the others
        for rank, rot in ((0, True), (2, False)):
            d_m, d_nm = hipmem.DevBuf(world * cap * 4), hipmem.DevBuf(world * 4)
            voc.records_bow_match_device(recs.ptr, rb, world, rank, cap, 50.0, 0.8, rot, d_m.ptr, d_nm.ptr, 0)
            ex.synchronize()
            gm = d_m.to_numpy(np.int32, world * cap).reshape(world, cap)
            gn = d_nm.to_numpy(np.int32, world)
            k1, d1 = feats[rank]
            fv1 = HS.ORBVocabulary.containers(*oracle.bow_transform(To, d1, levelsup))[1]
            for peer in range(world):
                if peer == rank:
                    assert gn[peer] == 0 and (gm[peer] == -1).all()
                    continue
                k2, d2 = feats[peer]
                fv2 = HS.ORBVocabulary.containers(*oracle.bow_transform(To, d2, levelsup))[1]
                om, on = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, None, 50.0, 0.8, rot)
                assert gn[peer] == on and np.array_equal(gm[peer, :len(k1)], om), (levelsup, rank, peer)
                assert (gm[peer, len(k1):] == -1).all()
            if levelsup >= 3 and rank == 0:
                assert gn[2] > 300                                                   # the noisy second view of frame 0
        voc.close()
