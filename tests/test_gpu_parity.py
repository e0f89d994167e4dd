"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs,
bit-exact for every stage (pyramid bytes, FAST candidates, quadtree selection, keypoint fields, descriptor bytes,
uRight/depth), against the committed golden vectors, and through size-independent properties at full size."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import hyslam_amd as HS
from hyslam_amd import _native as N
from hyslam_amd.synth import synth_image, synth_stereo_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def settings(n, scale=1.2, levels=8):
    return HS.FeatureExtractorSettings(nFeatures=n, fScaleFactor=scale, nLevels=levels)


def assert_same_features(gk, gd, ok, od):
    assert len(gk) == len(ok), (len(gk), len(ok))
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(gk[f], ok[f]), f
    assert gk.tobytes() == ok.tobytes()
    assert np.array_equal(gd, od)


def stage_parity(img, nfeat, scale=1.2):
    p = oracle.default_params(nfeat, scale)
    ok, od, dbg = oracle.extract(p, img, debug=True)
    ex = HS.ORBExtractor(settings(nfeat, scale))
    # the product path first (the quadtree stage works from the key histogram the FAST kernel leaves and never lists the candidates), then the same
    # frame in debug mode (hs_orb_set_debug: the candidates are gathered into dense lists as well): same features, and the candidate sets per level
    gk0, gd0 = ex(img)
    assert_same_features(gk0, gd0, ok, od)
    ex.set_debug(True)
    gk, gd = ex(img)
    sc = oracle.scale_tables(p)[0]
    for l in range(p.nlevels):
        assert np.array_equal(ex.debug_level(0, l), dbg["pyramid"][l]), "pyramid level %d" % l
        gc = ex.debug_candidates(0, l)
        oc = dbg["candidates"][l].astype(np.int32)
        assert len(gc) == len(oc), "candidate count level %d" % l
        if len(gc):
            assert np.array_equal(gc[np.lexsort((gc[:, 0], gc[:, 1]))], oc[np.lexsort((oc[:, 0], oc[:, 1]))]), "candidates level %d" % l
        gs = ex.debug_selected(0, l)
        m = ok["octave"] == l
        assert len(gs) == int(m.sum()) == int(dbg["n_selected"][l]), "selection count level %d" % l
        mul = np.float32(sc[l]) if l else np.float32(1)
        assert np.array_equal(gs[:, 0].astype(np.float32) * mul, ok["x"][m]) and np.array_equal(gs[:, 1].astype(np.float32) * mul, ok["y"][m])
        assert np.array_equal(gs[:, 2].astype(np.float32), ok["response"][m])
    assert_same_features(gk, gd, ok, od)
    ex.set_debug(False)
    return ex, gk, gd


def test_c1_mono_640x480_stagewise(gpu):
    stage_parity(synth_image(1, 640, 480), 1000)


def test_c1_matches_committed_golden(gpu):
    for name in ("c1_mono_640x480_1000", "imaging_800x600_1500_s14"):
        g = np.load(os.path.join(G, name + ".npz"))
        img = synth_image(int(g["seed"]), int(g["w"]), int(g["h"]))
        ex = HS.ORBExtractor(settings(int(g["nfeat"]), float(g["scale"])))
        gk, gd = ex(img)
        assert gk.tobytes() == g["keypoints"].tobytes() and np.array_equal(gd, g["descriptors"]), name
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        assert [sha(ex.debug_level(0, l)) for l in range(8)] == g["pyramid_sha256"].tolist()


def test_c2_stereo_1080p_extract_and_match(gpu):
    L, R = synth_stereo_pair(2, 1920, 1080)
    ex, gkL, gdL = stage_parity(L, 2000)
    p = oracle.default_params(2000)
    okR, odR = oracle.extract(p, R)
    gkR, gdR = ex(R)
    assert_same_features(gkR, gdR, okR, odR)
    sp = oracle.stereo_params(fx=1050.0, mbf=1050.0 * 0.12, n_rows=1080)
    ouR, odepth, _, _ = oracle.stereo_match(gkL, gdL, gkR, gdR, sp)
    sm = HS.Stereomatcher(gkL, gkR, gdL, gdR, HS.Camera(1050.0, 1050.0 * 0.12, 1080.0), extractor=ex)
    sm.computeStereoMatches()
    guR, gdepth = sm.getData()
    assert np.array_equal(guR, ouR) and np.array_equal(gdepth, odepth)
    assert int((gdepth > 0).sum()) > 200


def test_stereo_golden_and_identical_views(gpu):
    g = np.load(os.path.join(G, "stereo_640x480_1000.npz"))
    ex = HS.ORBExtractor(settings(1000))
    cam = HS.Camera(float(g["fx"]), float(g["fx"]) * 0.12, float(g["h"]))
    sm = HS.Stereomatcher(g["kL"], g["kR"], g["dL"], g["dR"], cam, extractor=ex)
    sm.computeStereoMatches()
    assert np.array_equal(sm.getData()[0], g["uRight"]) and np.array_equal(sm.getData()[1], g["depth"])
    # left == right: every keypoint matches itself at distance 0 with disparity 0 (the 0.01 px clamp, Stereomatcher.cpp:128-132);
    # the median distance is then 0, thDist = 0, and the reference's `dist < thDist` filter (:146-155) rejects every match.
    sm = HS.Stereomatcher(g["kL"], g["kL"], g["dL"], g["dL"], cam, extractor=ex)
    sm.computeStereoMatches()
    uR, depth = sm.getData()
    sp = oracle.stereo_params(fx=float(g["fx"]), mbf=float(g["fx"]) * 0.12, n_rows=int(g["h"]))
    ouR, odepth, _, _ = oracle.stereo_match(g["kL"], g["dL"], g["kL"], g["dL"], sp)
    assert np.array_equal(uR, ouR) and np.array_equal(depth, odepth)
    assert (depth == -1).all() and (uR == -1).all()
    # one differing right descriptor lifts the median above 0 only if it is the median element; flip bits in most of them instead
    dR = g["dL"].copy()
    dR[: len(dR) * 3 // 4, 0] ^= 0x0F                                           # distance 4 for 3/4 of the keypoints
    sm = HS.Stereomatcher(g["kL"], g["kL"], g["dL"], dR, cam, extractor=ex)
    sm.computeStereoMatches()
    uR, depth = sm.getData()
    ouR, odepth, _, _ = oracle.stereo_match(g["kL"], g["dL"], g["kL"], dR, sp)
    assert np.array_equal(uR, ouR) and np.array_equal(depth, odepth) and (depth > 0).sum() > len(dR) // 2
    m = depth > 0
    assert np.array_equal(uR[m], (g["kL"]["x"][m].astype(np.float64) - 0.01).astype(np.float32))
    assert np.array_equal(depth[m], np.full(int(m.sum()), np.float32(float(g["fx"]) * 0.12) / np.float32(0.01), np.float32))


@pytest.mark.parametrize("w,h,nfeat,scale", [(643, 481, 700, 1.2), (320, 200, 500, 1.2), (800, 600, 1500, 1.4), (1024, 400, 6000, 1.2),
                                              (97, 83, 300, 1.2), (2562, 1441, 3000, 1.2), (3840, 2160, 3000, 1.2), (4000, 3000, 3000, 1.4),
                                              (4096, 64, 500, 1.2), (4000, 3000, 2000, 1.2), (1280, 720, 1000, 1.2), (1352, 1014, 3000, 1.4)])
def test_ragged_sizes_and_profiles(gpu, w, h, nfeat, scale):
    """odd widths (byte-wise tile loads at level 0), levels too small for a FAST cell, the 1.4 'Imaging' profile,
    a wide frame with three root nodes and a quota above 1300, frames with ~3900 and ~8800 cells at level 0 (one and several gather rounds in the
    quadtree kernel), the 4000x3000 documentation camera of BASELINE config 4 with the reference's 'Imaging' settings (3000 features, 1.4), a 64:1 strip (127 root nodes, capacity beyond the generic bound); the same 4000x3000 camera at 1.2 / 2000 features
    ("also run at 1.2"), and the reference's OWN camera geometries with their feature profiles: the ZED-mini SLAM camera 1280x720 / 1000 @1.2 and
    the GoPro Imaging camera 2704x2028 at scale 0.5 = 1352x1014 / 3000 @1.4 (config/sample_primary_config_file.yaml:35-38,63-66, config/slam_feature_config.yaml:8-29)."""
    stage_parity(synth_image(40 + w, w, h), nfeat, scale)


@pytest.mark.parametrize("w,h,nfeat,scale,density", [(1352, 1014, 9000, 1.4, 1), (1352, 1014, 9000, 1.4, 4), (1280, 720, 3000, 1.2, 1),
                                                      (4000, 3000, 9000, 1.4, 1), (1920, 1080, 10800, 1.4, 4), (1920, 1080, 15000, 1.2, 4)])
def test_init_extractors_quota_above_2040(gpu, w, h, nfeat, scale, density):
    """The reference builds, per camera, a third extractor with THREE TIMES the camera's feature count for the frames it sees while initialising
    (ImageProcessing.cpp:34-36, :51-53).  For its own "Imaging" camera (3000 features @1.4, config/slam_feature_config.yaml:22-29; 2704x2028 at scale 0.5)
    that is 9000 features @1.4: a level-0 quota of 2758, above the 2040 the quadtree kernel's general instance lists in LDS — until round 6 the library
    REFUSED to create that extractor.  The large-list instance (<3328, no points in LDS, rectangles in global scratch>) takes quotas up to 3320 per level.
    Stage-wise parity on the reference's own camera sizes, on a scene dense enough to FILL the quota (lists of more than 2048 nodes), at the SLAM camera's
    init profile (3000 @1.2: the general instance), on the 4000x3000 camera, and at the largest quotas the instance holds (@1.4 and @1.2)."""
    img = synth_image(60 + w + nfeat, w, h, density * max(40, (w * h) // 800))
    ex, gk, gd = stage_parity(img, nfeat, scale)
    q = ex.GetFeaturesPerLevel()
    per_level = np.bincount(gk["octave"], minlength=8)
    if density > 1:
        assert per_level[0] >= q[0] and (q[0] <= 2040 or per_level[0] > 2040), (q, per_level)      # the quota was reached: the list really grew past the general instance's capacity
    # the same handle on two frames in one call (both workgroups of a level index their own scratch)
    img2 = synth_image(61 + w + nfeat, w, h, density * max(40, (w * h) // 800))
    (k1, k2), (d1, d2) = ex.extract_batch([img, img2])
    p = oracle.default_params(nfeat, scale)
    ok2, od2 = oracle.extract(p, img2)
    assert_same_features(k1, d1, gk, gd)
    assert_same_features(k2, d2, ok2, od2)


def test_quota_beyond_the_large_instance_is_refused_cleanly(gpu):
    """a level quota above 3320 (here 20 000 features @1.2: 4342 on level 0) is still refused at create — with a status, not a crash"""
    with pytest.raises(Exception):
        HS.ORBExtractor(settings(20000, 1.2))


def test_random_geometries(gpu):
    """seeded sweep over frame sizes, scale factors, level counts, cell sizes and quotas (cell rows with a single-cell last group, levels
    without cells, tiles of every height class, 4-level pyramids): full feature parity for each"""
    rng = np.random.default_rng(20261003)
    for case in range(24):
        w, h = int(rng.integers(90, 900)), int(rng.integers(80, 600))
        if w < h // 2 + 1:
            w = h                                                 # aspect ratios below 0.5 are rejected (nIni == 0 in the reference)
        scale = float(rng.choice([1.2, 1.25, 1.4, 1.7]))
        levels = int(rng.choice([3, 5, 8]))
        cells = int(rng.choice([16, 24, 30, 30, 37, 48]))
        nfeat = int(rng.integers(100, 2500))
        img = synth_image(1000 + case, w, h) if case % 5 else rng.integers(0, 256, (h, w), dtype=np.uint8)
        p = oracle.default_params(nfeat, scale, levels)
        p.cell_px = cells
        st = settings(nfeat, scale, levels)
        st.N_CELLS = cells
        try:
            ex = HS.ORBExtractor(st)
            gk, gd = ex(img)
        except HS.HsError as e:                                   # configurations the library rejects up front are fine; wrong bits are not
            assert "not supported" in str(e) or "undefined" in str(e) or "collapses" in str(e), (case, w, h, scale, levels, cells, str(e))
            continue
        ok, od = oracle.extract(p, img)
        assert len(gk) == len(ok), (case, w, h, scale, levels, cells, nfeat, len(gk), len(ok))
        assert gk.tobytes() == ok.tobytes() and np.array_equal(gd, od), (case, w, h, scale, levels, cells, nfeat)


def test_custom_blur_taps_and_large_cells(gpu):
    """non-default parameters: a 256-sum tap set and taps that overflow a byte (generic blur path), and N_CELLS = 40
    (cells wider than 37 px: the general FAST tile variant), N_CELLS = 60 (cells up to 119 px: tall tiles)."""
    img = synth_image(77, 640, 480)
    for taps in ([16, 34, 50, 56, 50, 34, 16], [300, 20, 10, 5, 10, 20, 300]):
        p = oracle.default_params(800)
        for i, t in enumerate(taps):
            p.blur_taps[i] = t
        ok, od = oracle.extract(p, img)
        ex = HS.ORBExtractor(settings(800), blur_taps=taps)
        gk, gd = ex(img)
        assert_same_features(gk, gd, ok, od)
    for cells in (40, 60):                                    # cells up to 2*N_CELLS-1 px: the taller FAST tile variants (up to 125 + 6 rows)
        p = oracle.default_params(800)
        p.cell_px = cells
        ok, od = oracle.extract(p, img)
        s = settings(800)
        s.N_CELLS = cells
        gk, gd = HS.ORBExtractor(s)(img)
        assert_same_features(gk, gd, ok, od)
    s.N_CELLS = 300                                           # one cell wider than 247 px: rejected, never mis-computed
    with pytest.raises(HS.HsError):
        HS.ORBExtractor(s)(img)


def test_empty_flat_and_noise_frames(gpu):
    ex = HS.ORBExtractor(settings(500))
    k, d = ex(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)                                   # silent return, ORBExtractor.cpp:499-500
    k, d = ex(np.full((240, 320), 77, np.uint8))
    assert len(k) == 0                                                          # no corners anywhere
    rng = np.random.default_rng(9)
    noise = rng.integers(0, 256, (240, 320), dtype=np.uint8)                    # FAST saturates: thousands of candidates per level
    ok, od = oracle.extract(oracle.default_params(500), noise)
    gk, gd = ex(noise)
    assert_same_features(gk, gd, ok, od)


def test_blur_saturation_white_blocks(gpu):
    """the default 7-tap kernel sums to 257/256: over an all-white (254, 255) neighbourhood the blurred value leaves the byte range and must
    saturate to 255, never wrap (the column pass relies on the clamp bit of v_dot2 for that); also a custom kernel of byte taps summing to 257
    with a heavy centre, and a 9-level configuration (the generic level search of k_describe's header)"""
    rng = np.random.default_rng(77)
    img = rng.integers(0, 40, (480, 640)).astype(np.int32)
    for _ in range(120):
        x, y, bw, bh = int(rng.integers(0, 600)), int(rng.integers(0, 440)), int(rng.integers(8, 40)), int(rng.integers(8, 40))
        img[y:y + bh, x:x + bw] = int(rng.choice([253, 254, 255, 255]))
    img = np.clip(img, 0, 255).astype(np.uint8)
    for nlev in (8, 9):
        ok, od = oracle.extract(oracle.default_params(800, 1.2, nlev), img)
        gk, gd = HS.ORBExtractor(settings(800, 1.2, nlev))(img)
        assert len(ok) > 200
        assert_same_features(gk, gd, ok, od)


@pytest.mark.parametrize("env", [{}, {"HS_QT_POINT_DOMAIN": "1"}, {"HS_EXTRACT_SPLIT": "1"}, {"HS_EXTRACT_SPLIT": "0"}, {"HS_PYRAMID_NO_FUSE": "1"}, {"HS_FAST_TEST_SMALL_LISTS": "1"}, {"HS_FAST_TEST_SCAN_B": "1"}, {"HS_FAST_COLS": "32"},
                                 {"HS_FAST_COLS": "32", "HS_FAST_TEST_SMALL_LISTS": "1"}, {"HS_FAST_COLS": "32", "HS_FAST_TEST_SCAN_B": "1"},
                                 {"HS_PYRAMID_CHAIN": "2"}, {"HS_PYRAMID_CHAIN": "0"}, {"HS_PYRAMID_DEEP_MAX": "0"}, {"HS_PYRAMID_DEEP_MAX": "100000"}, {"HS_PYRAMID_DEEP_MAX": "100000", "HS_PYRAMID_NW8": "0"}, {"HS_PYRAMID_NW8": "100000"}, {"HS_PYRAMID_NW8": "0"},
                                 # explicit launch plans (chain lengths from level 1): four- and five-level chains, a chain that starts on an even level, single levels
                                 {"HS_PYRAMID_PLAN": "3,4", "HS_PYRAMID_DEEP_MAX": "0"}, {"HS_PYRAMID_PLAN": "2,5", "HS_PYRAMID_DEEP_MAX": "0"}, {"HS_PYRAMID_PLAN": "1,2,3,1", "HS_PYRAMID_DEEP_MAX": "0"},
                                 {"HS_FAST_ORDER": "0"}, {"HS_FAST_ORDER": "2"}, {"HS_FAST_IMAGE_MAJOR": "1"},
                                 # the work queues (what every launch used until round 3; now the launches of > 2 units per workgroup) on wide and on narrow items
                                 {"HS_FAST_NO_FOLD": "1", "HS_FAST_COLS": "64"}, {"HS_FAST_NO_FOLD": "1", "HS_FAST_COLS": "64", "HS_FAST_NQ": "8"},
                                 {"HS_FAST_NO_FOLD": "1", "HS_FAST_COLS": "64", "HS_FAST_NQ": "16"}, {"HS_FAST_NO_FOLD": "1", "HS_FAST_COLS": "64", "HS_FAST_ORDER": "0"},
                                 {"HS_FAST_NO_FOLD": "1", "HS_FAST_COLS": "32"}, {"HS_FAST_NO_FOLD": "1"},
                                 # the folded static schedule on wide items; narrow items never (threshold 1 item)
                                 {"HS_FAST_COLS": "64"}, {"HS_FAST_COLS": "64", "HS_FAST_TEST_SMALL_LISTS": "1"}, {"HS_FAST_NARROW_MAX": "1"},
                                 # the quadtree stage gathering the candidates and computing their keys itself (the scheme until round 3)
                                 {"HS_FAST_KEYS": "0"}, {"HS_FAST_KEYS": "0", "HS_QT_POINT_DOMAIN": "1"},
                                 # ... the keys for every batch size / for the first two levels only / with the point-domain passes forced (the candidates are fetched after all)
                                 {"HS_FAST_KEYS_MAX_BATCH": "100000"}, {"HS_FAST_KEYS_LEVELS": "2"}, {"HS_FAST_KEYS_MAX_BATCH": "100000", "HS_QT_POINT_DOMAIN": "1"},
                                 {"HS_FAST_KEYS_MAX_BATCH": "100000", "HS_FAST_COLS": "64", "HS_FAST_TEST_SMALL_LISTS": "1"}, {"HS_FAST_KEYS_MAX_BATCH": "100000", "HS_FAST_TEST_SCAN_B": "1"}])
def test_fast_kernel_variants_in_subprocess(gpu, env):
    """the FAST kernel's tile-width variants (narrow / wide work items forced whatever the batch), its two schedules (folded static for small
    launches, work queues) forced on both widths, its list-overflow (flush) paths forced by a tiny LDS list, NMS driven from the score
    tile instead of the corner list, the quadtree's point-domain passes, the pyramid's chain kernel for every fused group / for none, its deep
    chains (small batches: as many levels per launch as the LDS holds — all seven at 1080p) never / for every batch, and its
    8-wave / 4-wave workgroups forced, and the split launch sequence (level 0's FAST + quadtree on a second
    stream beside the pyramid) forced on / off: same bits as the oracle"""
    e = dict(os.environ)
    e.update(env)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fast_variant_check.py")], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FAST_VARIANT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("env", [{"HS_QT_POINT_DOMAIN": "1"}, {"HS_FAST_KEYS": "0"}, {"HS_FAST_KEYS": "0", "HS_QT_POINT_DOMAIN": "1"}])
def test_large_list_quadtree_variants(gpu, monkeypatch, env):
    """k_quadtree<3328, 0, rectangles in global scratch> (round 6) on its other paths: the point-domain passes from the start (HS_QT_POINT_DOMAIN=1: every node rectangle
    is read and written through the workgroup's piece of global scratch between barriers) and without the FAST kernel's keys (the gather); a scene dense enough to
    fill a level-0 quota of 2 758 (the reference's 9000-feature init extractor @1.4), one frame and two frames per call."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w, h, nfeat, scale = 1352, 1014, 9000, 1.4
    p = oracle.default_params(nfeat, scale)
    imgs = [synth_image(90 + i, w, h, 4 * max(40, (w * h) // 800)) for i in range(2)]
    ref = [oracle.extract(p, im) for im in imgs]
    ex = HS.ORBExtractor(settings(nfeat, scale))
    assert max(ex.GetFeaturesPerLevel()) > 2040
    gk, gd = ex(imgs[0])
    assert_same_features(gk, gd, *ref[0])
    assert int((gk["octave"] == 0).sum()) > 2040                       # the list really grew past the general instance's capacity
    ks, ds = ex.extract_batch(imgs)
    for i in range(2):
        assert_same_features(ks[i], ds[i], *ref[i])


@pytest.mark.parametrize("env", [{"HS_QT_SMALL": "1"}, {"HS_QT_SMALL": "1", "HS_QT_POINT_DOMAIN": "1"}, {"HS_QT_SMALL": "1", "HS_FAST_KEYS_MAX_BATCH": "100000"}])
def test_quadtree_two_per_cu_instance(gpu, monkeypatch, env):
    """k_quadtree<1024, 0> — the instance without points in LDS, two workgroups per CU, for launches of more than 256 workgroups (HS_QT_SMALL=1; off by default:
    measured slower) — on 40 frames x 8 levels = 320 workgroups: count domain after a gather, the point-domain passes forced, and with the FAST kernel's keys"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    frames = [synth_image(300 + i, 416, 320) if i % 7 else np.random.default_rng(i).integers(0, 256, (320, 416), dtype=np.uint8) for i in range(40)]
    ex = HS.ORBExtractor(settings(900))
    kl, dl = ex.extract_batch(frames)
    p = oracle.default_params(900)
    for i, f in enumerate(frames):
        ok, od = oracle.extract(p, f)
        assert_same_features(kl[i], dl[i], ok, od)


def test_fast_thresholds_other_than_the_references_20(gpu):
    """hs_orb_params::fast_threshold accepts 0..255 (the reference is stuck at 20, ORBFinder.cpp:58-60): the quick reject's reduced-precision bound
    k = (t + 1) / 4, the one-polarity choice and 'score >= t is the segment test' must hold for every t, not only for 20"""
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:480, 0:640]
    frames = [synth_image(5, 640, 480), rng.integers(0, 256, (480, 640), dtype=np.uint8),
              np.clip((((yy // 5) + (xx // 5)) % 2 * 150).astype(np.int32) + rng.integers(0, 30, (480, 640)), 0, 255).astype(np.uint8),
              ((xx * 255) // 639).astype(np.uint8)]
    for t in (0, 1, 3, 4, 7, 19, 21, 40, 63, 64, 127, 128, 200, 255):
        p = oracle.default_params(800, 1.2, 8)
        p.fast_threshold = t
        ex = HS.ORBExtractor(settings(800), fast_threshold=t)
        for img in frames:
            ok, od = oracle.extract(p, img, cap=8000)
            gk, gd = ex(img)
            assert len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and np.array_equal(gd, od), "threshold %d" % t


def test_quadtree_nodes_above_65535_points_rank_by_size(gpu):
    """a saturated 4 Mpx frame (checkerboard of pitch 4 + noise) with a small quota: level 1 holds 264 572 candidates in ONE root, whose four children
    hold more than 65 535 points each when the size-ordered passes start.  Their sort keys carried the count in 16 bits (clamped), so they tied and
    went by list index instead of by size: 5 of 59 keypoints differed from the oracle (seed 2019 of the long fuzz campaign of round 4).  The
    reference sorts by the real size (`ORBExtractor.cpp:321-325`)."""
    h, w = 1679, 2421
    yy, xx = np.mgrid[0:h, 0:w]
    base = (((yy // 4) + (xx // 4)) % 2 * 121).astype(np.int32)
    p = oracle.default_params(50, 1.2, 8)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=50, fScaleFactor=1.2, nLevels=8))
    for seed in (5, 9, 14, 7):                                    # noise seeds 5, 9, 14 differed with the clamped keys (7 happened to agree)
        img = np.clip(base + np.random.default_rng(seed).integers(0, 30, (h, w)), 0, 255).astype(np.uint8)
        ok, od, dbg = oracle.extract(p, img, cap=4 * 50 + 64 * 8 + 1024, debug=True, cand_cap=1 << 21)
        assert int(dbg["n_candidates"].max()) > 4 * 65535, "the frame must put more than 65 535 points into each child of one root"
        buf = np.zeros((h, w + 16), np.uint8)                     # the row stride of the fuzz case (odd: the unaligned load path)
        buf[:, :w] = img
        gk, gd = ex(buf[:, :w])
        assert_same_features(gk, gd, ok, od)


def test_quadtree_count_domain_and_its_fallbacks(gpu):
    """DistributeOctTree in the count domain (histogram pyramid of the points' geometric keys, phase 1 in closed form) and every way out of it:
    clustered corners that need nodes deeper than the pyramid (switch to the point-domain passes in the middle of the distribution), 3..8 root
    nodes (pyramid depth 5), more than 8 root nodes (point domain from the start), quotas the first pass already exceeds, single-point roots"""
    rng = np.random.default_rng(21)
    frames = []
    flat = np.full((480, 640), 90, np.uint8)
    a = flat.copy(); a[200:290, 300:420] = rng.integers(0, 256, (90, 120), dtype=np.uint8)                      # one dense cluster
    b = flat.copy()
    for (y, x) in ((40, 50), (60, 500), (380, 90), (300, 330)):
        b[y:y + 40, x:x + 56] = rng.integers(0, 256, (40, 56), dtype=np.uint8)                                   # four clusters, far apart
    c = synth_image(5, 640, 480); c[100:160, 100:200] = rng.integers(0, 256, (60, 100), dtype=np.uint8)         # a scene with a hot spot
    frames += [(a, 3000, 8), (a, 300, 8), (b, 2500, 8), (b, 40, 4), (c, 4000, 8), (c, 60, 8)]
    frames += [(synth_image(31, 1200, 300), 1500, 6), (synth_image(32, 1500, 220), 900, 4), (synth_image(33, 1900, 200), 1200, 3)]   # 4, 7 and 11 roots
    d = flat.copy(); d[100:104, 100:104] = 255; d[300:304, 500:504] = 0                                          # a handful of corners: single-point roots
    frames += [(d, 500, 3), (rng.integers(0, 256, (300, 420), dtype=np.uint8), 5000, 5)]
    for img, nf, nl in frames:
        p = oracle.default_params(nf); p.nlevels = nl
        s = settings(nf); s.nLevels = nl
        ex = HS.ORBExtractor(s)
        ok, od = oracle.extract(p, img)
        gk, gd = ex(img)
        assert_same_features(gk, gd, ok, od)
        for l in range(nl):                                        # the selection of every level, in list order
            assert len(ex.debug_selected(0, l)) == (ok["octave"] == l).sum(), (img.shape, nf, l)
    assert len(ok) > 0


def test_batch_equals_singles_and_is_deterministic(gpu):
    imgs = [synth_image(60 + i, 640, 480) for i in range(5)]
    ex = HS.ORBExtractor(settings(1000))
    ks, ds = ex.extract_batch(imgs)
    ks2, ds2 = ex.extract_batch(imgs)
    for i, im in enumerate(imgs):
        k1, d1 = ex(im)
        assert k1.tobytes() == ks[i].tobytes() == ks2[i].tobytes() and np.array_equal(d1, ds[i]) and np.array_equal(ds[i], ds2[i])
    ok, od = oracle.extract(oracle.default_params(1000), imgs[3])
    assert_same_features(ks[3], ds[3], ok, od)


def test_strided_input_and_capacity_error(gpu):
    big = synth_image(70, 700, 500)
    view = big[10:490, 20:660]                                                  # non-contiguous rows are re-packed by the binding
    ex = HS.ORBExtractor(settings(800))
    ok, od = oracle.extract(oracle.default_params(800), view)
    gk, gd = ex(view)
    assert_same_features(gk, gd, ok, od)
    img = np.ascontiguousarray(view)
    kps = np.zeros(10, N.KP_DTYPE)
    desc = np.zeros((10, 32), np.uint8)
    n = C.c_int32(-5)
    st = ex._lib.hs_orb_extract(ex._h, img.ctypes.data_as(C.c_void_p), 640, 480, 640, kps.ctypes.data_as(C.c_void_p),
                                desc.ctypes.data_as(C.c_void_p), 10, C.byref(n))
    assert st == N.HS_ERR_CAPACITY and b"cap" in ex._lib.hs_orb_last_error(ex._h)
    assert not kps["x"].any()                                                   # nothing partial written


def test_properties_at_full_size(gpu):
    """BASELINE sizes, checked through size-independent properties (no oracle run): level quotas, bounds, response ordering
    inside the quadtree guarantee, scale bookkeeping, determinism, stereo consistency."""
    L, R = synth_stereo_pair(5, 1920, 1080)
    ex = HS.ORBExtractor(settings(2000))
    (kL, kR), (dL, dR) = ex.extract_batch([L, R])
    (kL2, _), _ = ex.extract_batch([L, R])
    assert kL.tobytes() == kL2.tobytes()
    quota = ex.GetFeaturesPerLevel()
    sc = ex.GetScaleFactors()
    for k in (kL, kR):
        assert (np.diff(k["octave"]) >= 0).all()                                 # levels concatenated in order
        for l in range(8):
            m = k["octave"] == l
            assert quota[l] <= m.sum() <= quota[l] + 2                           # DistributeOctTree overshoots by at most 2
            lw, lh = np.rint(np.float32(1920) / sc[l]), np.rint(np.float32(1080) / sc[l])
            x, y = k["x"][m] / sc[l], k["y"][m] / sc[l]
            assert (np.abs(x - np.rint(x)) < 1e-3).all() and (x > 18.5).all() and (x < lw - 19.5).all() and (y > 18.5).all() and (y < lh - 19.5).all()
            assert (k["size"][m] == np.float32(int(np.float32(31) * sc[l]))).all()
        assert (k["response"] >= 19).all() and (k["angle"] >= 0).all() and (k["angle"] < 360).all()
        assert len(np.unique(np.stack([k["x"], k["y"], k["octave"]], 1), axis=0)) == len(k)
    sm = HS.Stereomatcher(kL, kR, dL, dR, HS.Camera(), extractor=ex)
    sm.computeStereoMatches()
    uR, depth = sm.getData()
    m = depth > 0
    assert m.sum() > 200 and (uR[~m] == -1).all() and (depth[~m] == -1).all()
    disp = kL["x"][m] - uR[m]
    assert (disp > 0).all() and (disp < 1050).all() and np.array_equal(depth[m], np.float32(1050.0 * 0.12) / disp)


def test_device_api_bench_and_torch_interop_in_subprocess(gpu):
    """Device-resident batch entry points with torch-owned HBM buffers and torch's stream (separate process: torch brings its
    own HIP runtime and must be imported before the library)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_device_api_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--pairs", "2", "--cpu-seconds", "0"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] > 0 and line["unit"] == "stereo_pairs/s" and line["roofline"]["frac"] > 0 and line["n_gpus"] == 1
