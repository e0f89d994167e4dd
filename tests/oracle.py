"""ctypes binding of the CPU oracle (oracle/_build/libhs_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by hyslam_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "libhs_oracle.so")


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4")])


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("cell_px", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("fast_threshold", C.c_int32),
                ("blur_taps", C.c_uint16 * 7), ("_pad", C.c_uint16)]


class StereoParams(C.Structure):
    _fields_ = [("fx", C.c_float), ("mbf", C.c_float), ("n_rows", C.c_int32), ("th_high", C.c_float),
                ("th_low", C.c_float), ("size_ref", C.c_float)]


class ExtractDebug(C.Structure):
    _fields_ = [("pyramid", C.POINTER(C.c_void_p)), ("blurred", C.POINTER(C.c_void_p)),
                ("n_candidates", C.POINTER(C.c_int32)), ("n_selected", C.POINTER(C.c_int32)),
                ("candidates", C.POINTER(C.c_void_p)), ("cand_cap", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.hso_fast_atan2.restype = C.c_float
        L.hso_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.hso_cv_round_f.argtypes = [C.c_float]
        L.hso_cv_round_d.argtypes = [C.c_double]
        L.hso_ic_angle.restype = C.c_float
        L.hso_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.hso_orb_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.hso_pattern.restype = C.POINTER(C.c_int32)
        _lib = L
    return _lib


def default_params(nfeatures=1000, scale=1.2, nlevels=8):
    p = OrbParams()
    lib().hso_default_params(C.byref(p))
    p.nfeatures, p.scale_factor, p.nlevels = nfeatures, scale, nlevels
    return p


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def scale_tables(p):
    n = p.nlevels
    sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
    q = np.zeros(n, np.int32)
    lib().hso_scale_tables(C.byref(p), *(a.ctypes.data_as(C.c_void_p) for a in (sc, isc, s2, is2, q)))
    return sc, isc, s2, is2, q


def pyramid_size(p, w, h, level):
    lw, lh = C.c_int32(), C.c_int32()
    lib().hso_pyramid_size(C.byref(p), w, h, level, C.byref(lw), C.byref(lh))
    return lw.value, lh.value


def cell_grid(p, lw, lh):
    v = [C.c_int32() for _ in range(4)]
    lib().hso_cell_grid(C.byref(p), lw, lh, *(C.byref(x) for x in v))
    return tuple(x.value for x in v)


def umax():
    u = np.zeros(16, np.int32)
    lib().hso_umax(u.ctypes.data_as(C.c_void_p))
    return u


def pattern():
    return np.ctypeslib.as_array(lib().hso_pattern(), shape=(1024,)).copy()


def resize_linear(src, dw, dh):
    src, ps = _u8(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().hso_resize_linear_u8(ps, src.shape[1], src.shape[0], src.strides[0], dst.ctypes.data_as(C.c_void_p), dw, dh, dw)
    return dst


def preprocess_size(w, h, scale):
    ow, oh = C.c_int32(), C.c_int32()
    lib().hso_preprocess_size(int(w), int(h), C.c_float(scale), C.byref(ow), C.byref(oh))
    return ow.value, oh.value


def preprocess(src, rgb, scale):
    """ImageProcessing::PreProcessImg (ImageProcessing.cpp:118-138): src = (h, w) or (h, w, 3 | 4) uint8; returns the grey frame of the scaled size"""
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    ow, oh = preprocess_size(w, h, scale)
    dst = np.zeros((max(oh, 1), max(ow, 1)), np.uint8)
    rc = lib().hso_preprocess(src.ctypes.data_as(C.c_void_p), w, h, src.strides[0], cn, int(bool(rgb)), C.c_float(scale), dst.ctypes.data_as(C.c_void_p), dst.strides[0])
    if rc != 0:
        raise ValueError("preprocess: empty result or unsupported channel count")
    return dst


def fast(img, threshold=20, nonmax=True, cap=1 << 20):
    img, pi = _u8(img)
    out = np.zeros((cap, 3), np.int32)
    n = lib().hso_fast9_16(pi, img.shape[1], img.shape[0], img.strides[0], threshold, int(nonmax), out.ctypes.data_as(C.c_void_p), cap)
    assert n <= cap
    return out[:n].copy()


def gaussian_blur7(img, taps=None):
    img, pi = _u8(img)
    dst = np.zeros_like(img)
    t = None if taps is None else np.ascontiguousarray(taps, np.uint16)
    lib().hso_gaussian_blur7(pi, img.shape[1], img.shape[0], img.strides[0], dst.ctypes.data_as(C.c_void_p), dst.strides[0],
                             None if t is None else t.ctypes.data_as(C.c_void_p))
    return dst


def ic_angle(img, x, y):
    img, pi = _u8(img)
    return float(lib().hso_ic_angle(pi, img.strides[0], float(x), float(y)))


def orb_descriptor(img, x, y, angle):
    img, pi = _u8(img)
    d = np.zeros(32, np.uint8)
    lib().hso_orb_descriptor(pi, img.strides[0], float(x), float(y), float(angle), d.ctypes.data_as(C.c_void_p))
    return d


def hamming(a, b):
    a, pa = _u8(a)
    b, pb = _u8(b)
    return lib().hso_hamming256(pa, pb)


def distribute_octtree(cands_xyr, minX, maxX, minY, maxY, N):
    c = np.ascontiguousarray(cands_xyr, np.float32).reshape(-1, 3)
    cap = N + 64 + 4 * 16
    out = np.zeros(cap, np.int32)
    n = lib().hso_distribute_octtree(c.ctypes.data_as(C.c_void_p), len(c), minX, maxX, minY, maxY, N, out.ctypes.data_as(C.c_void_p), cap)
    assert n <= cap
    return out[:n].copy()


def level_candidates(p, level_img, cap=1 << 20):
    img, pi = _u8(level_img)
    out = np.zeros((cap, 3), np.float32)
    n = lib().hso_level_candidates(C.byref(p), pi, img.shape[1], img.shape[0], img.strides[0], out.ctypes.data_as(C.c_void_p), cap)
    assert n <= cap
    return out[:n].copy()


def extract(p, img, cap=None, debug=False, cand_cap=1 << 18):
    """Returns (kps structured array, desc (n,32) u8[, dbg dict])."""
    img, pi = _u8(img)
    h, w = img.shape
    if cap is None:
        cap = p.nfeatures + 4 * p.nlevels + 64
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    dbg = None
    keep = []
    if debug:
        L = p.nlevels
        sizes = [pyramid_size(p, w, h, l) for l in range(L)]
        pyr = [np.zeros((lh, lw), np.uint8) for (lw, lh) in sizes]
        blur = [np.zeros((lh, lw), np.uint8) for (lw, lh) in sizes]
        cands = [np.zeros((cand_cap, 3), np.float32) for _ in range(L)]
        ncand = np.zeros(L, np.int32)
        nsel = np.zeros(L, np.int32)
        arr = lambda lst: (C.c_void_p * L)(*[a.ctypes.data for a in lst])
        a1, a2, a3 = arr(pyr), arr(blur), arr(cands)
        keep = [a1, a2, a3]
        dbg = ExtractDebug(C.cast(a1, C.POINTER(C.c_void_p)), C.cast(a2, C.POINTER(C.c_void_p)),
                           ncand.ctypes.data_as(C.POINTER(C.c_int32)), nsel.ctypes.data_as(C.POINTER(C.c_int32)),
                           C.cast(a3, C.POINTER(C.c_void_p)), cand_cap)
    n = lib().hso_orb_extract(C.byref(p), pi, w, h, img.strides[0], kps.ctypes.data_as(C.c_void_p),
                              desc.ctypes.data_as(C.c_void_p), cap, C.byref(dbg) if dbg is not None else None)
    assert 0 <= n <= cap, n
    if debug:
        assert (ncand <= cand_cap).all()
        return kps[:n].copy(), desc[:n].copy(), dict(pyramid=pyr, blurred=blur, n_candidates=ncand, n_selected=nsel,
                                                     candidates=[c[:k].copy() for c, k in zip(cands, ncand)])
    return kps[:n].copy(), desc[:n].copy()


def stereo_params(fx=1050.0, mbf=1050.0 * 0.12, n_rows=1080, th_high=100.0, th_low=50.0, size_ref=31.0):
    return StereoParams(fx, mbf, n_rows, th_high, th_low, size_ref)


def stereo_match(kpsL, descL, kpsR, descR, sp):
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE)
    kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    descL = np.ascontiguousarray(descL, np.uint8)
    descR = np.ascontiguousarray(descR, np.uint8)
    nL, nR = len(kpsL), len(kpsR)
    uR = np.zeros(nL, np.float32)
    depth = np.zeros(nL, np.float32)
    bidx = np.zeros(nL, np.int32)
    bdist = np.zeros(nL, np.int32)
    lib().hso_stereo_match(kpsL.ctypes.data_as(C.c_void_p), descL.ctypes.data_as(C.c_void_p), nL,
                           kpsR.ctypes.data_as(C.c_void_p), descR.ctypes.data_as(C.c_void_p), nR, C.byref(sp),
                           uR.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p),
                           bidx.ctypes.data_as(C.c_void_p), bdist.ctypes.data_as(C.c_void_p))
    return uR, depth, bidx, bdist


def stereo_frontend(p, sp, imgL, imgR, cap=None):
    imgL, pl = _u8(imgL)
    imgR, pr = _u8(imgR)
    h, w = imgL.shape
    if cap is None:
        cap = p.nfeatures + 4 * p.nlevels + 64
    kL, kR = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
    dL, dR = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    nL, nR = C.c_int32(), C.c_int32()
    uR, depth = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    lib().hso_stereo_frontend(C.byref(p), C.byref(sp), pl, pr, w, h, imgL.strides[0],
                              kL.ctypes.data_as(C.c_void_p), dL.ctypes.data_as(C.c_void_p), C.byref(nL),
                              kR.ctypes.data_as(C.c_void_p), dR.ctypes.data_as(C.c_void_p), C.byref(nR), cap,
                              uR.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p))
    return kL[:nL.value], dL[:nL.value], kR[:nR.value], dR[:nR.value], uR[:nL.value], depth[:nL.value]


# ---------------------------------------------------------------- matchers (oracle/hs_oracle_match.cpp)
LM_DTYPE = np.dtype([("pos", "<f4", 3), ("size", "<f4"), ("min_dist", "<f4"), ("max_dist", "<f4"), ("normal", "<f4", 3),
                     ("assoc_kp", "<i4"), ("prev_angle", "<f4"), ("skip", "<i4"), ("desc", "u1", 32)])
assert LM_DTYPE.itemsize == 80


class FrameView(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mbf", C.c_float),
                ("sensor", C.c_int32), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("size_ref", C.c_float), ("n", C.c_int32),
                ("kps", C.c_void_p), ("desc", C.c_void_p), ("uR", C.c_void_p), ("kp_lm_obs", C.c_void_p)]


class VocabTree(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("levels", C.c_int32), ("child_begin", C.c_void_p), ("child_count", C.c_void_p),
                ("desc", C.c_void_p), ("word_id", C.c_void_p), ("weight", C.c_void_p), ("orig_id", C.c_void_p)]


class ProjParams(C.Structure):
    _fields_ = [("th", C.c_float), ("score_threshold", C.c_float), ("second_best_ratio", C.c_float),
                ("frac_smaller", C.c_float), ("frac_larger", C.c_float),
                ("use_distance", C.c_int32), ("use_stereo", C.c_int32), ("check_rotation", C.c_int32),
                ("use_prev_matched", C.c_int32), ("use_viewing_angle", C.c_int32), ("max_view_angle", C.c_float),
                ("use_reprojection", C.c_int32), ("reproj_threshold", C.c_float), ("sigma_ref", C.c_float), ("first_wins", C.c_int32),
                ("dist_is_invariance_range", C.c_int32)]

    def __init__(self, th=3.0, score_threshold=100.0, second_best_ratio=0.6, frac_smaller=0.5, frac_larger=1.5, use_distance=1, use_stereo=1,
                 check_rotation=0, use_prev_matched=1, use_viewing_angle=0, max_view_angle=1.047, use_reprojection=0, reproj_threshold=5.99,
                 sigma_ref=1.0, first_wins=0, dist_is_invariance_range=0):
        super().__init__(th, score_threshold, second_best_ratio, frac_smaller, frac_larger, use_distance, use_stereo, check_rotation,
                         use_prev_matched, use_viewing_angle, max_view_angle, use_reprojection, reproj_threshold, sigma_ref, first_wins, dist_is_invariance_range)


def make_frame_view(cls, Rcw, tcw, fx, fy, cx, cy, mbf, sensor, bounds, kps, desc, uR=None, kp_lm_obs=None, size_ref=31.0):
    """Builds a FrameView-like ctypes struct of class `cls`; returns (struct, keepalive list)."""
    Rcw = np.asarray(Rcw, np.float32).reshape(3, 3)
    tcw = np.asarray(tcw, np.float32).reshape(3)
    Ow = (-(Rcw.T.astype(np.float32) @ tcw)).astype(np.float32)          # mOw = -mRcw.t()*mtcw (Frame.cc:160-166)
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    keep = [kps, desc]
    F = cls()
    F.Rcw[:] = Rcw.ravel().tolist(); F.tcw[:] = tcw.tolist(); F.Ow[:] = Ow.tolist()
    F.fx, F.fy, F.cx, F.cy, F.mbf, F.sensor = fx, fy, cx, cy, mbf, sensor
    F.min_x, F.max_x, F.min_y, F.max_y = bounds
    F.size_ref, F.n = size_ref, len(kps)
    F.kps, F.desc = kps.ctypes.data, desc.ctypes.data
    if uR is not None:
        uR = np.ascontiguousarray(uR, np.float32); keep.append(uR); F.uR = uR.ctypes.data
    if kp_lm_obs is not None:
        o = np.ascontiguousarray(kp_lm_obs, np.int32); keep.append(o); F.kp_lm_obs = o.ctypes.data
    return F, keep


def search_by_projection(F, lms, pp):
    lms = np.ascontiguousarray(lms, LM_DTYPE)
    L = len(lms)
    midx = np.full(L, -1, np.int32)
    mdist = np.full(L, -1, np.float32)
    n = lib().hso_search_by_projection(C.byref(F), lms.ctypes.data_as(C.c_void_p), L, C.byref(pp),
                                       midx.ctypes.data_as(C.c_void_p), mdist.ctypes.data_as(C.c_void_p))
    return midx, mdist, n


def search_by_projection_sim3(KF, Scw, lms, th, th_low, kp_matched):
    """-> (match_idx[L], kp_matched' [n], nmatches); lms carry the invariance range in min_dist / max_dist"""
    lms = np.ascontiguousarray(lms, LM_DTYPE)
    S = np.ascontiguousarray(Scw, np.float32).reshape(16)
    taken = np.ascontiguousarray(kp_matched, np.uint8).copy()
    midx = np.full(len(lms), -1, np.int32)
    n = lib().hso_search_by_projection_sim3(C.byref(KF), S.ctypes.data_as(C.c_void_p), lms.ctypes.data_as(C.c_void_p), len(lms), int(th), C.c_float(th_low),
                                            taken.ctypes.data_as(C.c_void_p), midx.ctypes.data_as(C.c_void_p))
    return midx, taken, n


def search_by_sim3(KF1, lms1, KF2, lms2, s12, R12, t12, th, th_high):
    l1 = np.ascontiguousarray(lms1, LM_DTYPE); l2 = np.ascontiguousarray(lms2, LM_DTYPE)
    R = np.ascontiguousarray(R12, np.float32).reshape(9); t = np.ascontiguousarray(t12, np.float32).reshape(3)
    m = np.full(KF1.n, -1, np.int32)
    n = lib().hso_search_by_sim3(C.byref(KF1), l1.ctypes.data_as(C.c_void_p), C.byref(KF2), l2.ctypes.data_as(C.c_void_p), C.c_float(s12),
                                 R.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), C.c_float(th), C.c_float(th_high), m.ctypes.data_as(C.c_void_p))
    return m, n


def frame_grid(F):
    out = np.zeros((F.n, 2), np.int32)
    lib().hso_frame_grid(C.byref(F), out.ctypes.data_as(C.c_void_p))
    return out


def search_by_bow(k1, d1, fv1, k2, d2, fv2, keep1, thr, ratio, check_rotation, keep2=None, F12=None, size_ref=31.0, sigma_ref=1.0):
    """fv = (node_id, node_ptr, idx) int32 arrays (CSR feature vector)."""
    k1 = np.ascontiguousarray(k1, KP_DTYPE); k2 = np.ascontiguousarray(k2, KP_DTYPE)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    a = [np.ascontiguousarray(x, np.int32) for x in fv1]
    b = [np.ascontiguousarray(x, np.int32) for x in fv2]
    keep = None if keep1 is None else np.ascontiguousarray(keep1, np.uint8)
    kp2 = None if keep2 is None else np.ascontiguousarray(keep2, np.uint8)
    Fm = None if F12 is None else np.ascontiguousarray(F12, np.float32).reshape(9)
    m = np.full(len(k1), -1, np.int32)
    p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    n = lib().hso_search_by_bow_ex(p(k1), p(d1), len(k1), p(a[0]), p(a[1]), p(a[2]), len(a[0]),
                                   p(k2), p(d2), len(k2), p(b[0]), p(b[1]), p(b[2]), len(b[0]),
                                   p(keep), p(kp2), p(Fm), C.c_float(size_ref), C.c_float(sigma_ref),
                                   C.c_float(thr), C.c_float(ratio), int(check_rotation), p(m))
    return m, n


def search_by_bow_legacy(k1, d1, fv1, k2, d2, fv2, keep1, keep2, th_low, nnratio, check_orientation):
    """the legacy SearchByBoW(pKF1, pKF2, vpMatches12), FeatureMatcher.cc:938-1077"""
    k1 = np.ascontiguousarray(k1, KP_DTYPE); k2 = np.ascontiguousarray(k2, KP_DTYPE)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    a = [np.ascontiguousarray(x, np.int32) for x in fv1]
    b = [np.ascontiguousarray(x, np.int32) for x in fv2]
    kp1 = None if keep1 is None else np.ascontiguousarray(keep1, np.uint8)
    kp2 = None if keep2 is None else np.ascontiguousarray(keep2, np.uint8)
    m = np.full(len(k1), -1, np.int32)
    p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    n = lib().hso_search_by_bow_legacy(p(k1), p(d1), len(k1), p(a[0]), p(a[1]), p(a[2]), len(a[0]),
                                       p(k2), p(d2), len(k2), p(b[0]), p(b[1]), p(b[2]), len(b[0]),
                                       p(kp1), p(kp2), C.c_float(th_low), C.c_float(nnratio), int(check_orientation), p(m))
    return m, n


def hamming_knn2(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    bi, bd, sd = (np.zeros(len(q), np.int32) for _ in range(3))
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    lib().hso_hamming_knn2(p(q), len(q), p(t), len(t), p(bi), p(bd), p(sd))
    return bi, bd, sd


def rotation_consistency(angle_a, angle_b):
    a = np.ascontiguousarray(angle_a, np.float32); b = np.ascontiguousarray(angle_b, np.float32)
    keep = np.zeros(len(a), np.uint8)
    lib().hso_rotation_consistency(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), len(a), keep.ctypes.data_as(C.c_void_p))
    return keep.astype(bool)


def make_vocab_tree(cls, k, levels, seed):
    """A seeded synthetic k-ary vocabulary of `levels` levels below the root (stands in for ORBvoc, which is not available):
    node descriptors are random, leaves get consecutive word ids and idf-like weights.  Returns (struct, keepalive, n_words)."""
    rng = np.random.default_rng(seed)
    n_nodes = sum(k ** l for l in range(levels + 1))
    cb = np.zeros(n_nodes, np.int32); cc = np.zeros(n_nodes, np.int32)
    nxt = 1
    first_leaf = sum(k ** l for l in range(levels))
    for i in range(first_leaf):
        cb[i], cc[i] = nxt, k
        nxt += k
    desc = rng.integers(0, 256, (n_nodes, 32), dtype=np.uint8)
    desc[rng.integers(1, n_nodes, n_nodes // 20)] = desc[rng.integers(1, n_nodes, n_nodes // 20)]     # duplicate node descriptors: ties
    word = np.full(n_nodes, -1, np.int32); word[first_leaf:] = np.arange(n_nodes - first_leaf)
    weight = np.zeros(n_nodes, np.float32); weight[first_leaf:] = rng.uniform(0.5, 9.0, n_nodes - first_leaf)
    T = cls(n_nodes, levels, cb.ctypes.data, cc.ctypes.data, desc.ctypes.data, word.ctypes.data, weight.ctypes.data)
    return T, [cb, cc, desc, word, weight], n_nodes - first_leaf


def bow_transform(T, desc, levelsup):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(desc)
    w = np.zeros(n, np.int32); wt = np.zeros(n, np.float32); nd = np.zeros(n, np.int32)
    lib().hso_bow_transform(C.byref(T), desc.ctypes.data_as(C.c_void_p), n, levelsup, w.ctypes.data_as(C.c_void_p),
                            wt.ctypes.data_as(C.c_void_p), nd.ctypes.data_as(C.c_void_p))
    return w, wt, nd


def search_for_initialization(k1, d1, F2, prev_xy, window, th_low, nnratio):
    k1 = np.ascontiguousarray(k1, KP_DTYPE); d1 = np.ascontiguousarray(d1, np.uint8)
    prev = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2).copy()
    m = np.full(len(k1), -1, np.int32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    n = lib().hso_search_for_initialization(p(k1), p(d1), len(k1), C.byref(F2), p(prev), int(window), C.c_float(th_low), C.c_float(nnratio), p(m))
    return m, prev, n
