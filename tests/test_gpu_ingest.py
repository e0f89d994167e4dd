"""Pipelined host ingest of the C ABI (hs_orb_submit_batch / hs_orb_wait / hs_host_alloc): two tickets in flight, the H2D copy of the second
under the kernels of the first — results bit-identical to the oracle whatever overlaps (ImageProcessing.cpp:69-116, System.cc:194-196)."""
import ctypes as C

import numpy as np
import pytest

import oracle
import hyslam_amd as HS
from hyslam_amd import _native as N
from hyslam_amd.synth import synth_image, synth_stereo_pair

pytestmark = pytest.mark.gpu


def test_two_stereo_tickets_in_flight_pinned_and_pageable(gpu):
    W, H, NF, P = 640, 480, 1000, 3
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF))
    cam = HS.Camera(fx=500.0, mbf=60.0, mnMaxY=float(H))
    sp = HS.stereo_params(cam)
    osp = oracle.stereo_params(fx=500.0, mbf=60.0, n_rows=H)
    p = oracle.default_params(NF)
    pin = ex.pinned_frames(2 * P, H, W)
    page = np.zeros((2 * P, H, W + 24), np.uint8)[:, :, 7:7 + W]           # pageable, row-strided, unaligned views
    pairs_a = [synth_stereo_pair(40 + i, W, H) for i in range(P)]
    pairs_b = [synth_stereo_pair(50 + i, W, H) for i in range(P)]
    for i in range(P):
        pin[i], pin[P + i] = pairs_a[i]
        page[i], page[P + i] = pairs_b[i]
    ta = ex.submit_batch([pin[i] for i in range(2 * P)], sp)
    tb = ex.submit_batch([page[i] for i in range(2 * P)], sp)              # enqueued while ticket a is still running
    assert ta != tb and ta > 0 and tb > 0
    with pytest.raises(HS.HsError):                                         # both staging slots are in flight
        ex.submit_batch([pin[i] for i in range(2 * P)], sp)
    # waiting out of order is allowed
    rb = ex.wait(tb)
    ra = ex.wait(ta)
    for (n, k, d, uR, depth), pairs in ((ra, pairs_a), (rb, pairs_b)):
        for i, (L, R) in enumerate(pairs):
            okL, odL = oracle.extract(p, L); okR, odR = oracle.extract(p, R)
            assert n[i] == len(okL) and n[P + i] == len(okR)
            assert k[i, :n[i]].tobytes() == okL.tobytes() and np.array_equal(d[i, :n[i]], odL)
            assert k[P + i, :n[P + i]].tobytes() == okR.tobytes() and np.array_equal(d[P + i, :n[P + i]], odR)
            ouR, odepth, _, _ = oracle.stereo_match(okL, odL, okR, odR, osp)
            assert np.array_equal(uR[i, :n[i]], ouR) and np.array_equal(depth[i, :n[i]], odepth)
            assert (odepth > 0).sum() > 50
    # a waited ticket is gone; slots are free again
    dummy = [np.zeros(1, np.int32), np.zeros((1, 1), N.KP_DTYPE), np.zeros((1, 1, 32), np.uint8)]
    assert ex._lib.hs_orb_wait(ex._h, ta, dummy[1].ctypes.data_as(C.c_void_p), dummy[2].ctypes.data_as(C.c_void_p), dummy[0].ctypes.data_as(C.c_void_p), 1, None, None) == N.HS_ERR_INVALID
    tc = ex.submit_batch([pin[i] for i in range(2 * P)], sp)
    rc = ex.wait(tc)
    assert np.array_equal(rc[0], ra[0]) and rc[1].tobytes() == ra[1].tobytes()


def test_mono_tickets_change_of_geometry_and_capacity_check(gpu):
    NF = 800
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=NF))
    p = oracle.default_params(NF)
    a = [synth_image(60 + i, 512, 384) for i in range(3)]
    b = [synth_image(70 + i, 800, 600) for i in range(2)]
    t1 = ex.submit_batch(a)
    t2 = ex.submit_batch(b)                                                # another frame size while the first ticket is in flight
    r2 = ex.wait(t2); r1 = ex.wait(t1)
    for (n, k, d, uR, depth), frames in ((r1, a), (r2, b)):
        assert uR is None and depth is None
        for i, f in enumerate(frames):
            ok, od = oracle.extract(p, f)
            assert n[i] == len(ok) and k[i, :n[i]].tobytes() == ok.tobytes() and np.array_equal(d[i, :n[i]], od)
    # an odd batch cannot be a stereo batch; a too small output capacity is refused and the ticket stays waitable
    with pytest.raises(HS.HsError):
        ex.submit_batch(a, HS.stereo_params(HS.Camera(400.0, 48.0, 384.0)))
    t3 = ex.submit_batch(a)
    small = (np.zeros(3, np.int32), np.zeros((3, 10), N.KP_DTYPE), np.zeros((3, 10, 32), np.uint8))
    st = ex._lib.hs_orb_wait(ex._h, t3, small[1].ctypes.data_as(C.c_void_p), small[2].ctypes.data_as(C.c_void_p), small[0].ctypes.data_as(C.c_void_p), 10, None, None)
    assert st == N.HS_ERR_CAPACITY
    r3 = ex.wait(t3)
    assert np.array_equal(r3[0], r1[0])


def test_ticket_survives_a_failed_wait_and_can_be_cancelled(gpu):
    """the Python binding keeps a ticket (and the frames it reads from) until hs_orb_wait has SUCCEEDED: a wait with too small reused arrays
    raises and can be repeated; a ticket waited for after a later submit of ANOTHER geometry still gets arrays of its own capacity;
    cancel() frees the slot; frames_copied() turns true"""
    import time
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=800))
    p = oracle.default_params(800)
    a = [synth_image(60 + i, 512, 384) for i in range(2)]
    wide = [synth_image(90, 1500, 220)]                                   # 7 root nodes per level: a larger max_keypoints() than the first geometry's
    t1 = ex.submit_batch(a)
    cap1 = ex._tickets[t1][3]
    t2 = ex.submit_batch(wide)
    assert ex._tickets[t2][3] >= cap1
    small = (np.zeros(2, np.int32), np.zeros((2, 10), N.KP_DTYPE), np.zeros((2, 10, 32), np.uint8), None, None)
    with pytest.raises(ValueError):
        ex.wait(t1, small)
    assert t1 in ex._tickets
    # reused arrays whose shapes disagree with each other (hs_orb_wait writes ALL of them in one [batch][cap] layout), or of the wrong type or batch:
    # refused before the C side can write out of bounds
    big = cap1 + 8
    for bad in ((np.zeros(2, np.int32), np.zeros((2, big), N.KP_DTYPE), np.zeros((2, 10, 32), np.uint8), None, None),            # descriptors too short
                (np.zeros(2, np.int32), np.zeros((1, big), N.KP_DTYPE), np.zeros((1, big, 32), np.uint8), None, None),           # one frame short
                (np.zeros(2, np.int64), np.zeros((2, big), N.KP_DTYPE), np.zeros((2, big, 32), np.uint8), None, None),           # counts of the wrong type
                (np.zeros(2, np.int32), np.zeros((2, big), N.KP_DTYPE), np.zeros((2, big, 32), np.uint8)[:, ::1, ::-1], None, None)):   # not contiguous
        with pytest.raises(ValueError):
            ex.wait(t1, bad)
    assert t1 in ex._tickets
    r1 = ex.wait(t1)                                                       # sized by the ticket's own capacity, not by the current geometry's
    assert r1[1].shape[1] == cap1
    for i, f in enumerate(a):
        ok, od = oracle.extract(p, f)
        assert r1[0][i] == len(ok) and r1[1][i, :r1[0][i]].tobytes() == ok.tobytes() and np.array_equal(r1[2][i, :r1[0][i]], od)
    assert t1 not in ex._tickets
    with pytest.raises(KeyError):
        ex.wait(t1)
    for _ in range(2000):
        if ex.frames_copied(t2):
            break
        time.sleep(0.001)
    assert ex.frames_copied(t2)
    ex.cancel(t2)
    assert t2 not in ex._tickets and ex._lib.hs_ticket_frames_copied(ex._h, t2) == -1
    t3, t4 = ex.submit_batch(a), ex.submit_batch(a)                        # both slots are free again
    r3, r4 = ex.wait(t3), ex.wait(t4)
    assert np.array_equal(r3[0], r1[0]) and r3[1][0, :r3[0][0]].tobytes() == r4[1][0, :r4[0][0]].tobytes()


def test_concurrent_handles_random_geometry(gpu):
    """distinct handles are concurrent (the reference runs its two extractors in two threads, ImageProcessing.cpp:82-84): three host threads, each with
    its own handles, call with changing geometry, quota and batch size at the same time — every result must be the oracle's, whatever the other
    threads' kernels are doing on the GPU"""
    import threading
    rng = np.random.default_rng(11)
    cases = []
    for i in range(8):
        w, h = int(rng.integers(200, 1300)), int(rng.integers(160, 800))
        h = min(h, w)
        nf = int(rng.integers(100, 2500))
        img = synth_image(100 + i, w, h)
        ok, od = oracle.extract(oracle.default_params(nf, 1.2, 8), img, cap=4 * nf + 2000)
        cases.append((img, nf, ok, od))
    bad = []

    def worker(t):
        try:
            ex, r = {}, np.random.default_rng(t)
            for it in range(25):
                j = int(r.integers(0, len(cases)))
                img, nf, ok, od = cases[j]
                if nf not in ex:
                    ex[nf] = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=nf, fScaleFactor=1.2, nLevels=8))
                if r.random() < 0.3:
                    ks, ds = ex[nf].extract_batch([img, img, img])
                    res = list(zip(ks, ds))
                else:
                    res = [ex[nf](img)]
                for gk, gd in res:
                    if not (len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)):
                        bad.append((t, it, j))
        except Exception as e:                                             # a dead thread must fail the test, not shorten it
            bad.append((t, "exception", repr(e)[:200]))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad[:5]
