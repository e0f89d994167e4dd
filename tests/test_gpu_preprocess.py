"""ImageProcessing::PreProcessImg on the device (hs_preprocess_device, hs_orb_extract_camera_batch; kernels_preprocess.hip) against the oracle's restatement
(oracle.preprocess): the camera's scale (copy, the 2x2 area path of exactly 0.5, the fixed-point bilinear), 1 / 3 / 4 channels in both colour orders, ragged and odd
sizes, odd row strides (byte-load path), batches; then the whole call surface: a colour frame in, features out, equal to oracle.extract(oracle.preprocess(frame))."""
import numpy as np
import pytest

import hipmem
import oracle
import hyslam_amd as HS
from hyslam_amd.synth import synth_image

pytestmark = pytest.mark.gpu


def colour_frame(seed, w, h, cn):
    """a structured grey scene per channel (different seeds: the channels differ), so that the grey result has corners"""
    if cn == 1:
        return synth_image(seed, w, h)
    chans = [synth_image(seed + 7 * k, w, h) for k in range(3)]
    if cn == 4:
        chans.append(np.full((h, w), 200, np.uint8))
    return np.ascontiguousarray(np.stack(chans, axis=2))


@pytest.mark.parametrize("cn", [1, 3, 4])
@pytest.mark.parametrize("scale", [1.0, 0.5, 0.75, 0.4, 1.25])
def test_preprocess_device_matches_the_oracle(gpu, cn, scale):
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=500))
    rng = np.random.default_rng(1000 * cn + int(100 * scale))
    for (w, h, pad, batch) in ((640, 480, 0, 1), (641, 479, 0, 2), (333, 201, 5, 3), (67, 35, 1, 1), (1280, 720, 0, 1)):
        frames = [rng.integers(0, 256, (h, w) if cn == 1 else (h, w, cn), dtype=np.uint8) for _ in range(batch)]
        ow, oh = oracle.preprocess_size(w, h, scale)
        row = w * cn + pad                                                # pad != 0: odd strides -> the byte-load path; grey rows padded too
        src = np.zeros((batch, h, row), np.uint8)
        for i, f in enumerate(frames):
            src[i, :, :w * cn] = f.reshape(h, w * cn)
        gpitch = ow + 3
        d_src, d_grey = hipmem.DevBuf.from_numpy(src), hipmem.DevBuf.from_numpy(np.full((batch, oh, gpitch), 0xAB, np.uint8))
        for rgb in (True, False):
            ex.preprocess_device(d_src.ptr, w, h, row, h * row, batch, cn, rgb, scale, d_grey.ptr, gpitch, oh * gpitch)
            ex.synchronize()
            got = d_grey.to_numpy(np.uint8, batch * oh * gpitch).reshape(batch, oh, gpitch)
            for i, f in enumerate(frames):
                assert np.array_equal(got[i, :, :ow], oracle.preprocess(f, rgb, scale)), (w, h, cn, scale, rgb, i)
            assert (got[:, :, ow:] == 0xAB).all(), "the kernel wrote past the scaled width"


@pytest.mark.parametrize("w,h,cn,rgb,scale,nfeat,fscale", [(2704, 2028, 3, True, 0.5, 3000, 1.4),      # the reference's "Imaging" camera (config/sample_primary_config_file.yaml:53-70)
                                                           (1280, 720, 3, True, 1.0, 1000, 1.2),       # its stereo camera: a copy + grey
                                                           (1280, 720, 1, True, 1.0, 1000, 1.2),       # grey frames: PreProcessImg changes nothing
                                                           (1920, 1080, 4, False, 0.5, 1000, 1.2),     # BGRA, area path
                                                           (1001, 777, 3, False, 0.6, 800, 1.2)])      # odd size, bilinear, BGR
def test_camera_frame_in_features_out(gpu, w, h, cn, rgb, scale, nfeat, fscale):
    st = HS.FeatureExtractorSettings(nFeatures=nfeat, fScaleFactor=fscale)
    ex = HS.ORBExtractor(st)
    frames = [colour_frame(31 + i, w, h, cn) for i in range(2)]
    (k, d, grey) = ex.extract_camera_batch(frames, rgb, scale, want_grey=True)
    p = oracle.default_params(nfeat, fscale)
    for i, f in enumerate(frames):
        og = oracle.preprocess(f, rgb, scale)
        assert np.array_equal(grey[i], og), i
        ok, od = oracle.extract(p, og)
        assert len(ok) > nfeat // 2
        assert k[i].tobytes() == ok.tobytes() and np.array_equal(d[i], od), i
    # the grey-frame call on the oracle's grey frame gives the same features (the two entry points share everything behind the level-0 buffer)
    k2, d2 = ex.extract_batch([oracle.preprocess(f, rgb, scale) for f in frames])
    assert all(a.tobytes() == b.tobytes() for a, b in zip(k, k2)) and all(np.array_equal(a, b) for a, b in zip(d, d2))


@pytest.mark.parametrize("cn,rgb,scale", [(3, True, 0.5), (4, False, 1.0), (3, False, 0.75), (1, True, 0.5)])
def test_stereo_front_end_from_camera_frames(gpu, cn, rgb, scale):
    """ProcessStereoImage with PreProcessImg inside (ImageProcessing.cpp:76-103) as ONE ticket of the pipelined ingest (hs_orb_submit_camera_batch): colour stereo
    pairs at the camera's own size in, keypoints / descriptors / uRight / depth out — against oracle.stereo_frontend on the oracle's grey frames; two tickets in
    flight, then a grey ticket on the same handle (the slots' raw buffers and the geometry switch)."""
    from hyslam_amd.synth import synth_stereo_pair
    from hyslam_amd import _native as N
    W, H, P = 640, 480, 3
    sw, sh = int(round(W / scale)), int(round(H / scale))
    ow, oh = oracle.preprocess_size(sw, sh, scale)
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=1000))
    p = oracle.default_params(1000)

    def colour(g, k):
        if cn == 1:
            return g
        ch = [g, np.roll(g, 2 + k, axis=1), np.roll(g, 1 + k, axis=0)] + ([np.full_like(g, 9)] if cn == 4 else [])
        return np.ascontiguousarray(np.stack(ch, axis=2))
    pairs = [synth_stereo_pair(700 + i, sw, sh) for i in range(P)]
    lefts, rights = [colour(a, 0) for a, _ in pairs], [colour(b, 0) for _, b in pairs]
    gsp = N.StereoParams(500.0, 60.0, oh, 100.0, 50.0, 31.0)
    osp = oracle.stereo_params(fx=500.0, mbf=60.0, n_rows=oh)
    t1 = ex.submit_camera_batch(lefts + rights, rgb, scale, gsp)
    t2 = ex.submit_camera_batch(lefts[:1] + rights[:1], rgb, scale, gsp)
    for t, idx in ((t1, range(P)), (t2, range(1))):
        n, k, d, u, z = ex.wait(t)
        npairs = len(idx)
        for j in idx:
            gl, gr = oracle.preprocess(lefts[j], rgb, scale), oracle.preprocess(rights[j], rgb, scale)
            assert gl.shape == (oh, ow)
            okL, odL, okR, odR, ou, oz = oracle.stereo_frontend(p, osp, gl, gr)
            a, b = int(n[j]), int(n[npairs + j])
            assert a == len(okL) and b == len(okR), (j, a, len(okL))
            assert k[j, :a].tobytes() == okL.tobytes() and np.array_equal(d[j, :a], odL) and k[npairs + j, :b].tobytes() == okR.tobytes() and np.array_equal(d[npairs + j, :b], odR), j
            assert np.array_equal(u[j, :a], ou) and np.array_equal(z[j, :a], oz), j
            assert (oz > 0).sum() > 30
    g = oracle.preprocess(lefts[0], rgb, scale)
    t3 = ex.submit_batch([g])
    n, k, d, _, _ = ex.wait(t3)
    ok, od = oracle.extract(p, g)
    assert k[0, :n[0]].tobytes() == ok.tobytes() and np.array_equal(d[0, :n[0]], od)


def test_bad_camera_parameters_are_refused(gpu):
    ex = HS.ORBExtractor(HS.FeatureExtractorSettings(nFeatures=500))
    with pytest.raises(Exception):
        ex.extract_camera_batch([np.zeros((4, 4), np.uint8)], True, 0.1)              # cvRound(0.4) = 0: empty
    with pytest.raises(Exception):
        ex.extract_camera_batch([np.zeros((64, 64, 2), np.uint8)], True, 1.0)         # two channels
    k, d = ex.extract_camera_batch([synth_image(3, 640, 480)], True, 1.0)              # and the handle still works
    assert len(k[0]) > 300
