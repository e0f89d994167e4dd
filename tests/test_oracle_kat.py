"""Known-answer tests that pin the oracle to tables derived from the reference SOURCE alone:
per-level quotas and scale factors (ORBExtractor.cpp:76-119), pyramid sizes (:568-569), FAST cell grids (:413-428),
umax (ORBFinder.cpp:131-149), the rBRIEF pattern checksum (ORBFinder.cpp:151-409), KeyPoint.size per octave
(ORBExtractor.cpp:478), and the ORBDistance bit-hack (DescriptorDistance.cpp:9-25)."""
import hashlib
import struct

import numpy as np
import pytest

import oracle


@pytest.mark.parametrize("n,scale,expect", [
    (1000, 1.2, [217, 181, 151, 126, 105, 87, 73, 60]),
    (2000, 1.2, [434, 362, 302, 251, 209, 175, 145, 122]),
    (3000, 1.2, [652, 543, 452, 377, 314, 262, 218, 182]),
    (6000, 1.2, [1303, 1086, 905, 754, 628, 524, 436, 364]),
    (3000, 1.4, [919, 657, 469, 335, 239, 171, 122, 88]),
])
def test_t1_quotas(n, scale, expect):
    assert oracle.scale_tables(oracle.default_params(n, scale))[4].tolist() == expect


def test_t1_scale_factors():
    sc, isc, s2, is2, _ = oracle.scale_tables(oracle.default_params(1000))
    expect = np.array([1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196], np.float32)
    assert np.array_equal(sc, expect)
    assert np.array_equal(isc, np.float32(1) / sc) and np.array_equal(s2, sc * sc) and np.array_equal(is2, np.float32(1) / (sc * sc))


@pytest.mark.parametrize("w,h,scale,expect", [
    (640, 480, 1.2, [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]),
    (1920, 1080, 1.2, [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362), (536, 301)]),
    (4000, 3000, 1.2, [(4000, 3000), (3333, 2500), (2778, 2083), (2315, 1736), (1929, 1447), (1608, 1206), (1340, 1005), (1116, 837)]),
    (4000, 3000, 1.4, [(4000, 3000), (2857, 2143), (2041, 1531), (1458, 1093), (1041, 781), (744, 558), (531, 398), (379, 285)]),
])
def test_t2_pyramid_sizes(w, h, scale, expect):
    p = oracle.default_params(1000, scale)
    assert [oracle.pyramid_size(p, w, h, l) for l in range(8)] == expect


def test_t3_cell_grids():
    p = oracle.default_params(2000)
    grids = [oracle.cell_grid(p, *oracle.pyramid_size(p, 1920, 1080, l)) for l in range(8)]
    assert [(g[0], g[1]) for g in grids] == [(62, 34), (52, 28), (43, 23), (35, 19), (29, 16), (24, 13), (20, 11), (16, 8)]
    assert grids[0][2:] == (31, 31) and grids[2][2:] == (31, 32) and grids[6][2:] == (31, 30) and grids[7][2:] == (32, 34)
    assert sum(g[0] * g[1] for g in grids) == 6342
    grids = [oracle.cell_grid(p, *oracle.pyramid_size(p, 640, 480, l)) for l in range(8)]
    assert [(g[0], g[1]) for g in grids] == [(20, 14), (16, 12), (13, 10), (11, 8), (9, 6), (7, 5), (6, 4), (4, 3)]
    assert grids[0][2:] == (31, 32) and grids[7][2:] == (37, 34)
    assert sum(g[0] * g[1] for g in grids) == 815


def test_t4_umax_and_pattern():
    assert oracle.umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    pat = oracle.pattern()
    assert pat[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12] and pat[-8:].tolist() == [7, 0, 12, -2, -1, -6, 0, -11]
    assert pat.min() == -13 and pat.max() == 12 and int(pat.sum()) == -406
    assert hashlib.sha256(struct.pack("<1024i", *pat.tolist())).hexdigest() == \
        "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"
    # rotated reach stays inside the 37x37 window the kernels stage
    r = np.hypot(pat[0::2].astype(float), pat[1::2].astype(float)).max()
    assert 18.0 < r < 18.5


def test_t5_keypoint_size_per_octave():
    img = np.zeros((480, 640), np.uint8)
    for scale, expect in ((1.2, [31, 37, 44, 53, 64, 77, 92, 111]), (1.4, [31, 43, 60, 85, 119, 166, 233, 326])):
        sc = oracle.scale_tables(oracle.default_params(1000, scale))[0]
        assert [int(np.float32(31) * s) for s in sc] == expect
    assert len(oracle.extract(oracle.default_params(1000), img)[0]) == 0       # flat frame: no corners, no crash


def test_cvround_half_even():
    assert [oracle.lib().hso_cv_round_f(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]
    assert [oracle.lib().hso_cv_round_d(v) for v in (0.5, 1.5, 2.5, -2.5)] == [0, 2, 2, -2]


def test_hamming_bithack_equals_popcount():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    for i in range(200):
        assert oracle.hamming(a[i], b[i]) == int(np.unpackbits(a[i] ^ b[i]).sum())
    assert oracle.hamming(a[0], a[0]) == 0 and oracle.hamming(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


def test_fast_atan2_tracks_atan2():
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = (float(v) for v in rng.integers(-200000, 200000, 2))
        a = oracle.lib().hso_fast_atan2(y, x)
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.3, (y, x, a, ref)
    assert oracle.lib().hso_fast_atan2(0.0, 0.0) == 0.0
    assert oracle.lib().hso_fast_atan2(0.0, 5.0) == 0.0 and oracle.lib().hso_fast_atan2(5.0, 0.0) == 90.0
