"""Host-side mirror of the reference's feature interfaces for the ORB hot path, on top of the C ABI.

Names, argument meaning and error behaviour follow bmhopkinson/hyslam:
  FeatureExtractorSettings  src/core/FeatureExtractorSettings.h:19-32
  FeatureMatcherSettings    src/features/FeatureMatcher.h:98-103
  ORBExtractor              src/features/ORBExtractor.h:62-130 (FeatureExtractor ABC: FeatureExtractor.h:25-37)
  Stereomatcher             src/features/Stereomatcher.h:25-51
  ORBFactory                src/features/ORBFactory.h / FeatureFactory.h:21-33
All compute happens in libhyslam_amd.so (HIP, gfx950).  Nothing here falls back to a CPU path.
"""
import ctypes as C

import numpy as np

from . import _native as N
from ._native import KP_DTYPE, HsError  # noqa: F401


class FeatureExtractorSettings:
    def __init__(self, nFeatures=1000, fScaleFactor=1.2, nLevels=8, init_threshold=20, min_threshold=4, N_CELLS=30,
                 size_ref=31.0, sigma_ref=1.0):
        # defaults of ORBFactory::ORBFactory(), src/features/ORBFactory.cpp:13-25
        self.nFeatures, self.fScaleFactor, self.nLevels = nFeatures, fScaleFactor, nLevels
        self.init_threshold, self.min_threshold, self.N_CELLS = init_threshold, min_threshold, N_CELLS
        self.size_ref, self.sigma_ref = size_ref, sigma_ref


class FeatureMatcherSettings:
    def __init__(self, nnratio=0.6, TH_HIGH=100.0, TH_LOW=50.0, checkOri=True):
        self.nnratio, self.TH_HIGH, self.TH_LOW, self.checkOri = nnratio, TH_HIGH, TH_LOW, checkOri


class Camera:
    """The fields of HYSLAM::Camera the stereo matcher reads (src/features/Stereomatcher.cpp:7-24,44)."""

    def __init__(self, fx=1050.0, mbf=1050.0 * 0.12, mnMaxY=1080.0):
        self._fx, self.mbf, self.mnMaxY = fx, mbf, mnMaxY

    def fx(self):
        return self._fx


def _params(settings, blur_taps=None, fast_threshold=20):
    p = N.OrbParams()
    N.lib().hs_orb_default_params(C.byref(p))
    p.nfeatures, p.scale_factor, p.nlevels = settings.nFeatures, settings.fScaleFactor, settings.nLevels
    p.cell_px, p.ini_th_fast, p.min_th_fast = settings.N_CELLS, settings.init_threshold, settings.min_threshold
    p.fast_threshold = fast_threshold
    if blur_taps is not None:
        for i in range(7):
            p.blur_taps[i] = int(blur_taps[i])
    return p


class ORBExtractor:
    """HYSLAM::ORBExtractor.  `extractor(image)` returns (keypoints[KP_DTYPE], descriptors[n,32] uint8)."""

    def __init__(self, settings=None, device=0, blur_taps=None, fast_threshold=20):
        """fast_threshold: hs_orb_params::fast_threshold (the reference always runs 20: ORBFinder.cpp:58-60; other values are for tests and other callers)"""
        self.settings = settings or FeatureExtractorSettings()
        self._lib = N.lib()
        self._h = C.c_void_p()
        self._p = _params(self.settings, blur_taps, fast_threshold)
        st = self._lib.hs_orb_create(C.byref(self._p), device, C.byref(self._h))
        if st != N.HS_OK:
            self._h = C.c_void_p()
            raise HsError(st, self._lib.hs_status_string(st).decode())
        self.device = device
        self.last_frame_tokens = []                        # of the last extract_batch(..., publish=True)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.hs_orb_destroy(self._h)
            self._h = C.c_void_p()
            for ptr in getattr(self, "_pinned", []):
                self._lib.hs_host_free(ptr)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- FeatureExtractor interface (FeatureExtractor.h:25-37)
    def GetLevels(self):
        return self._lib.hs_orb_get_levels(self._h)

    def GetScaleFactor(self):
        return self._lib.hs_orb_get_scale_factor(self._h)

    def _tables(self):
        n = self.GetLevels()
        arrs = [np.zeros(n, np.float32) for _ in range(4)] + [np.zeros(n, np.int32)]
        N.check(self._h, self._lib.hs_orb_get_scale_tables(self._h, *(a.ctypes.data_as(C.c_void_p) for a in arrs)))
        return arrs

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def GetFeaturesPerLevel(self):
        return self._tables()[4]

    def max_keypoints(self):
        return self._lib.hs_orb_max_keypoints(self._h)

    def __call__(self, image, mask=None):
        """operator()(image, mask, keypoints, descriptors), ORBExtractor.cpp:496-562 (mask ignored, as in the reference)."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)     # silent return, ORBExtractor.cpp:499-500
        k, d = self.extract_batch([image])
        return k[0], d[0]

    def extract_batch(self, images, publish=False):
        """publish=True: every extracted frame also stays on the device (include/hyslam_amd.h "device-resident frames"); the tokens of the call are in
        `last_frame_tokens` — what the C++ extractor adaptor does for hySLAM's FeatureViews (host/HipORBExtractor.h)."""
        # row-strided views (a cv::Mat ROI: unit pixel stride, row stride >= width) go to the C ABI as they are
        strided = lambda im: isinstance(im, np.ndarray) and im.ndim == 2 and im.dtype == np.uint8 and im.strides[1] == 1 and im.strides[0] >= im.shape[1]
        imgs = [im if strided(im) else np.ascontiguousarray(im) for im in images]
        if len({im.strides[0] for im in imgs if im.ndim == 2}) > 1:           # one row stride per call
            imgs = [np.ascontiguousarray(im) for im in imgs]
        for im in imgs:
            if im.dtype != np.uint8 or im.ndim != 2:
                raise TypeError("image must be CV_8UC1 (2-D uint8)")          # assert(image.type() == CV_8UC1), :503
            if im.shape != imgs[0].shape:
                raise ValueError("batched frames must have equal size")
        b = len(imgs)
        h, w = imgs[0].shape
        self.reserve(w, h, b)                                   # max_keypoints() depends on the frame's aspect ratio (root nodes per level)
        cap = self.max_keypoints()
        kps = np.zeros((b, cap), KP_DTYPE)
        desc = np.zeros((b, cap, 32), np.uint8)
        n = np.zeros(b, np.int32)
        ptrs = (C.c_void_p * b)(*[im.ctypes.data for im in imgs])
        N.check(self._h, self._lib.hs_orb_extract_batch(self._h, ptrs, b, w, h, imgs[0].strides[0],
                                                        kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap,
                                                        n.ctypes.data_as(C.c_void_p)))
        self.last_frame_tokens = []
        if publish:
            for i in range(b):
                tok = C.c_uint64(0)
                if n[i] > 0:
                    N.check(self._h, self._lib.hs_frame_publish(self._h, i, kps[i].ctypes.data_as(C.c_void_p), int(n[i]), C.byref(tok)))
                self.last_frame_tokens.append(tok.value)
        return [kps[i, :n[i]].copy() for i in range(b)], [desc[i, :n[i]].copy() for i in range(b)]

    def extract_camera_batch(self, frames, rgb, scale, want_grey=False):
        """ImageProcessing::PreProcessImg + the extractor call in one (src/main/ImageProcessing.cpp:44,55 / :76-77,82-83; hs_orb_extract_camera_batch): `frames`
        as the camera delivers them — (h, w) or (h, w, 3 | 4) uint8, equal sizes — cross PCIe as they are, are scaled by the camera's `scale` and turned to grey on
        the device (`rgb`: the camera's RGB key) and extracted.  Returns (keypoints per frame, descriptors per frame[, grey frames])."""
        imgs = [np.ascontiguousarray(f, np.uint8) for f in frames]
        for im in imgs:
            if im.ndim not in (2, 3) or (im.ndim == 3 and im.shape[2] not in (3, 4)) or im.shape != imgs[0].shape:
                raise TypeError("camera frames must be equal-sized (h, w) or (h, w, 3 | 4) uint8 arrays")
        b = len(imgs)
        h, w = imgs[0].shape[:2]
        cn = 1 if imgs[0].ndim == 2 else imgs[0].shape[2]
        pp = N.PreprocessParams(cn, int(bool(rgb)), float(scale), 0)
        ow, oh = C.c_int32(), C.c_int32()
        self._lib.hs_preprocess_size(w, h, C.c_float(scale), C.byref(ow), C.byref(oh))
        if ow.value < 1 or oh.value < 1:
            raise N.HsError(N.HS_ERR_INVALID, "the camera scale reduces the frame to nothing")
        self.reserve(ow.value, oh.value, b)
        cap = self.max_keypoints()
        kps = np.zeros((b, cap), KP_DTYPE)
        desc = np.zeros((b, cap, 32), np.uint8)
        n = np.zeros(b, np.int32)
        grey = np.zeros((b, oh.value, ow.value), np.uint8) if want_grey else None
        ptrs = (C.c_void_p * b)(*[im.ctypes.data for im in imgs])
        N.check(self._h, self._lib.hs_orb_extract_camera_batch(self._h, ptrs, b, w, h, imgs[0].strides[0], C.byref(pp),
                                                               kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap, n.ctypes.data_as(C.c_void_p),
                                                               grey.ctypes.data_as(C.c_void_p) if want_grey else None))
        out = ([kps[i, :n[i]].copy() for i in range(b)], [desc[i, :n[i]].copy() for i in range(b)])
        return out + ([grey[i] for i in range(b)],) if want_grey else out

    def preprocess_device(self, d_src, w, h, row_stride, image_stride, batch, channels, rgb, scale, d_grey, grey_row_stride, grey_image_stride, stream=0):
        """hs_preprocess_device: device frames -> device grey frames of the scaled size (asynchronous)"""
        pp = N.PreprocessParams(int(channels), int(bool(rgb)), float(scale), 0)
        N.check(self._h, self._lib.hs_preprocess_device(self._h, d_src, w, h, row_stride, image_stride, batch, C.byref(pp), d_grey, grey_row_stride, grey_image_stride, stream or None))

    def find_frame(self, keypoints):
        """token of the cached frame whose keypoint array equals `keypoints` bit for bit (0: none) — hs_frame_find, what the matcher adaptors do with a FeatureViews"""
        k = np.ascontiguousarray(keypoints, KP_DTYPE)
        tok = C.c_uint64(0)
        if len(k) == 0 or self._lib.hs_frame_find(self.device, k.ctypes.data_as(C.c_void_p), len(k), C.byref(tok)) != N.HS_OK:
            return 0
        return tok.value

    def release_frame(self, token):
        return self._lib.hs_frame_release(self.device, C.c_uint64(token)) == N.HS_OK

    # ---- pipelined host ingest (hs_orb_submit_batch / hs_orb_wait): at most two tickets in flight, like the reference's frame queue
    # (System.cc:194-196).  The H2D copy of a submitted batch overlaps the kernels of the batch before it.
    def submit_batch(self, images, sp=None):
        """images: same-sized 2-D uint8 arrays with one common row stride (pinned_frames() gives page-locked ones); with `sp` (a StereoParams)
        the first half are the left frames and the second half the right ones and the stereo matcher runs too.  Returns a ticket.
        The arrays must stay alive and unchanged until wait(ticket) returns."""
        b = len(images)
        h, w = images[0].shape
        for im in images:
            if im.dtype != np.uint8 or im.ndim != 2 or im.shape != (h, w) or im.strides != images[0].strides or im.strides[1] != 1:
                raise TypeError("frames must be 2-D uint8 of one size and one row stride")
        ptrs = (C.c_void_p * b)(*[im.ctypes.data for im in images])
        t = C.c_int32()
        N.check(self._h, self._lib.hs_orb_submit_batch(self._h, ptrs, b, w, h, images[0].strides[0], C.byref(sp) if sp is not None else None, C.byref(t)))
        self._tickets = getattr(self, "_tickets", {})
        # the ticket remembers ITS output capacity (the geometry may change with a later submit) and keeps the frames alive: the C side reads
        # them asynchronously until wait() / cancel() returns (include/hyslam_amd.h: lifetime of the frames)
        self._tickets[t.value] = (b, sp is not None, images, self.max_keypoints())
        return t.value

    def submit_camera_batch(self, frames, rgb, scale, sp=None):
        """the same with the camera's own frames (hs_orb_submit_camera_batch): equal-sized (h, w) or (h, w, 3 | 4) uint8 arrays, C-contiguous; PreProcessImg (camera
        scale + grey) runs on the device in front of the pyramid; with `sp` the first half are the left frames, the second half the right ones."""
        b = len(frames)
        for im in frames:
            if im.dtype != np.uint8 or im.ndim not in (2, 3) or im.shape != frames[0].shape or not im.flags["C_CONTIGUOUS"] or (im.ndim == 3 and im.shape[2] not in (3, 4)):
                raise TypeError("camera frames must be equal-sized C-contiguous (h, w) or (h, w, 3 | 4) uint8 arrays")
        h, w = frames[0].shape[:2]
        cn = 1 if frames[0].ndim == 2 else frames[0].shape[2]
        pp = N.PreprocessParams(cn, int(bool(rgb)), float(scale), 0)
        ptrs = (C.c_void_p * b)(*[im.ctypes.data for im in frames])
        t = C.c_int32()
        N.check(self._h, self._lib.hs_orb_submit_camera_batch(self._h, ptrs, b, w, h, frames[0].strides[0], C.byref(pp), C.byref(sp) if sp is not None else None, C.byref(t)))
        self._tickets = getattr(self, "_tickets", {})
        self._tickets[t.value] = (b, sp is not None, frames, self.max_keypoints())
        return t.value

    def wait(self, ticket, out=None):
        """-> (n[b], kps[b, cap], desc[b, cap, 32], uRight[b/2, cap] | None, depth[b/2, cap] | None); entries beyond n[i] are undefined.
        `out` = a tuple of arrays from a previous call to reuse."""
        b, stereo, _, cap = self._tickets[ticket]          # looked up, NOT removed: a wait that fails (capacity, arguments) can be repeated
        if out is None:
            out = (np.zeros(b, np.int32), np.zeros((b, cap), KP_DTYPE), np.zeros((b, cap, 32), np.uint8),
                   np.zeros((b // 2, cap), np.float32) if stereo else None, np.zeros((b // 2, cap), np.float32) if stereo else None)
        n, kps, desc, uR, depth = out
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        # hs_orb_wait writes all five arrays in ONE [batch][c] layout: every reused array must have exactly that shape and dtype, or the C side
        # would write out of bounds
        c = kps.shape[1] if (isinstance(kps, np.ndarray) and kps.ndim == 2) else -1
        want = [(n, (b,), np.int32), (kps, (b, c), KP_DTYPE), (desc, (b, c, 32), np.uint8)]
        if stereo:
            want += [(uR, (b // 2, c), np.float32), (depth, (b // 2, c), np.float32)]
        for a, shape, dt in want:
            if not isinstance(a, np.ndarray) or a.shape != shape or a.dtype != dt or not a.flags.c_contiguous or not a.flags.writeable:
                raise ValueError("wait(): a reused output array does not have the ticket's layout (need %s %s, C-contiguous)" % (shape, np.dtype(dt).name))
        if c < cap:
            raise ValueError("wait(): the reused output arrays hold %d keypoints per frame, the ticket needs %d" % (c, cap))
        try:
            N.check(self._h, self._lib.hs_orb_wait(self._h, ticket, p(kps), p(desc), p(n), kps.shape[1], p(uR), p(depth)))
        except HsError as e:
            if e.status == N.HS_ERR_HIP:                    # the C side has released the slot: the ticket is gone
                self._tickets.pop(ticket, None)
            raise
        del self._tickets[ticket]                           # only now: the results are out and the frames may go
        return out

    def cancel(self, ticket):
        """give up a ticket: waits until its batch has drained, drops the results"""
        N.check(self._h, self._lib.hs_orb_cancel(self._h, ticket))
        self._tickets.pop(ticket, None)

    def frames_copied(self, ticket):
        """True once the copy-in of `ticket` is complete (its frame buffers may be recycled)"""
        return self._lib.hs_ticket_frames_copied(self._h, ticket) == 1

    def pinned_frames(self, count, h, w):
        """`count` h x w uint8 frames in page-locked host memory (hs_host_alloc): H2D copies from them are plain DMA, no staging.
        The memory is released when the extractor is closed."""
        ptr = C.c_void_p()
        st = self._lib.hs_host_alloc(count * h * w, C.byref(ptr))
        if st != N.HS_OK:
            raise HsError(st, "hs_host_alloc")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(ptr)
        buf = (C.c_uint8 * (count * h * w)).from_address(ptr.value)
        return np.frombuffer(buf, np.uint8).reshape(count, h, w)

    # ---- device-resident entry points (pointers are plain integers, e.g. torch.Tensor.data_ptr())
    def reserve(self, w, h, batch):
        N.check(self._h, self._lib.hs_orb_reserve(self._h, w, h, batch))

    def extract_batch_device(self, d_imgs, batch, w, h, row_stride, image_stride, d_kps, d_desc, d_n, cap, stream=0):
        N.check(self._h, self._lib.hs_orb_extract_batch_device(self._h, d_imgs, batch, w, h, row_stride, image_stride,
                                                               d_kps, d_desc, d_n, cap, stream or None))

    def stereo_match_batch_device(self, d_kpsL, d_descL, d_nL, d_kpsR, d_descR, d_nR, pairs, cap, sp, d_uRight, d_depth, stream=0):
        N.check(self._h, self._lib.hs_stereo_match_batch_device(self._h, d_kpsL, d_descL, d_nL, d_kpsR, d_descR, d_nR, pairs, cap,
                                                                C.byref(sp), d_uRight, d_depth, stream or None))

    def stereo_frontend_batch_device(self, d_left, d_right, pairs, w, h, row_stride, image_stride,
                                     d_kpsL, d_descL, d_nL, d_kpsR, d_descR, d_nR, cap, sp, d_uRight, d_depth, stream=0):
        N.check(self._h, self._lib.hs_stereo_frontend_batch_device(self._h, d_left, d_right, pairs, w, h, row_stride, image_stride,
                                                                   d_kpsL, d_descL, d_nL, d_kpsR, d_descR, d_nR, cap,
                                                                   C.byref(sp), d_uRight, d_depth, stream or None))

    def debug_stream_copy(self, d_dst, d_src, nbytes, width=16, stream=0):
        """measurement utility (hs_debug_stream_copy): a grid-stride copy kernel of `width` (4 or 16) bytes per lane — known HBM traffic"""
        N.check(self._h, self._lib.hs_debug_stream_copy(self._h, d_dst, d_src, nbytes, width, stream or None))

    def set_lanes(self, lanes):
        """1 or 2 internal launch sequences for the batched device entry points (hs_orb_set_lanes)."""
        N.check(self._h, self._lib.hs_orb_set_lanes(self._h, int(lanes)))

    def set_split(self, mode):
        """-1 auto, 0 never, 1 always: level 0's FAST + quadtree on a second stream beside the pyramid (hs_orb_set_split)"""
        N.check(self._h, self._lib.hs_orb_set_split(self._h, int(mode)))

    def synchronize(self, stream=0):
        N.check(self._h, self._lib.hs_orb_synchronize(self._h, stream or None))

    STAGES = ("pyramid", "fast_cells", "quadtree", "describe", "stereo_match", "stereo_median")

    def pyramid_launches(self):
        """kernel launches of the pyramid stage per call (hs_orb_stage_launches)"""
        return max(1, self._lib.hs_orb_stage_launches(self._h, 0))

    def profile_begin(self):
        N.check(self._h, self._lib.hs_orb_profile_begin(self._h))

    def profile_pause(self):
        """stop recording stage events; what was recorded stays for profile_end()"""
        N.check(self._h, self._lib.hs_orb_profile_pause(self._h))

    def profile_end(self):
        """-> {stage: (total_ms, launches)} measured with HIP events on the launch stream."""
        ms = np.zeros(6, np.float64)
        cnt = np.zeros(6, np.int32)
        N.check(self._h, self._lib.hs_orb_profile_end(self._h, ms.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)))
        return {s: (float(ms[i]), int(cnt[i])) for i, s in enumerate(self.STAGES)}

    # ---- stage taps (parity tests)
    def debug_level(self, image, level):
        buf = np.zeros(1 << 26, np.uint8)
        lw, lh = C.c_int32(), C.c_int32()
        N.check(self._h, self._lib.hs_orb_debug_level(self._h, image, level, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(lw), C.byref(lh)))
        return buf[:lw.value * lh.value].reshape(lh.value, lw.value).copy()

    def set_debug(self, on=True):
        """debug mode: the quadtree stage also gathers the FAST candidates into dense per-level lists (debug_candidates reads them)"""
        N.check(self._h, self._lib.hs_orb_set_debug(self._h, 1 if on else 0))

    def debug_candidates(self, image, level, cap=1 << 20):
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int32()
        N.check(self._h, self._lib.hs_orb_debug_candidates(self._h, image, level, out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value].copy()

    def debug_selected(self, image, level, cap=1 << 16):
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int32()
        N.check(self._h, self._lib.hs_orb_debug_selected(self._h, image, level, out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value].copy()


def stereo_params(camera, settings=None, size_ref=31.0):
    settings = settings or FeatureMatcherSettings()
    return N.StereoParams(camera.fx(), camera.mbf, int(camera.mnMaxY), settings.TH_HIGH, settings.TH_LOW, size_ref)


class Stereomatcher:
    """HYSLAM::Stereomatcher (src/features/Stereomatcher.h:25-51): construct from the left/right views, the camera and the
    matcher settings, call computeStereoMatches(), read uRight/depth with getData()."""

    def __init__(self, keys, keysR, descriptors, descriptorsR, camera, settings=None, extractor=None, size_ref=31.0):
        self.mvKeys = np.ascontiguousarray(keys, KP_DTYPE)
        self.mvKeysRight = np.ascontiguousarray(keysR, KP_DTYPE)
        self.mDescriptors = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        self.mDescriptorsRight = np.ascontiguousarray(descriptorsR, np.uint8).reshape(-1, 32)
        self.sp = stereo_params(camera, settings, size_ref)
        self._ex = extractor or ORBExtractor()
        self.mvuRight = np.full(len(self.mvKeys), -1.0, np.float32)
        self.mvDepth = np.full(len(self.mvKeys), -1.0, np.float32)

    def computeStereoMatches(self):
        ex = self._ex
        nL, nR = len(self.mvKeys), len(self.mvKeysRight)
        self.mvuRight = np.full(nL, -1.0, np.float32)
        self.mvDepth = np.full(nL, -1.0, np.float32)
        # both views still on the device (published by the extractor)?  Then only the results cross the bus (hs_stereo_match_frames).
        self.frames_on_device = False
        tl, tr = (ex.find_frame(self.mvKeys), ex.find_frame(self.mvKeysRight)) if nL > 0 and nR > 0 else (0, 0)
        if tl and tr:
            st = ex._lib.hs_stereo_match_frames(ex._h, C.c_uint64(tl), C.c_uint64(tr), C.byref(self.sp), self.mvuRight.ctypes.data_as(C.c_void_p), self.mvDepth.ctypes.data_as(C.c_void_p))
            if st == N.HS_OK:
                self.frames_on_device = True
                return
            if st != N.HS_ERR_INVALID:                       # (INVALID: a slot was reused between find and use — fall back to the host arrays)
                N.check(ex._h, st)
        N.check(ex._h, ex._lib.hs_stereo_match(ex._h, self.mvKeys.ctypes.data_as(C.c_void_p), self.mDescriptors.ctypes.data_as(C.c_void_p), nL,
                                               self.mvKeysRight.ctypes.data_as(C.c_void_p), self.mDescriptorsRight.ctypes.data_as(C.c_void_p), nR,
                                               C.byref(self.sp), self.mvuRight.ctypes.data_as(C.c_void_p),
                                               self.mvDepth.ctypes.data_as(C.c_void_p)))

    def getData(self):
        return self.mvuRight, self.mvDepth


class FeatureMatcher:
    """HYSLAM::FeatureMatcher (src/features/FeatureMatcher.h:105-176) on flat arrays.  `frame` is a _native.FrameView,
    `landmarks` a numpy array of _native.LM_DTYPE (one record per MapPoint, in the order the reference would iterate them)."""

    def __init__(self, settings=None, extractor=None):
        s = settings or FeatureMatcherSettings()
        self.mfNNratio, self.mbCheckOrientation, self.TH_LOW, self.TH_HIGH = s.nnratio, s.checkOri, s.TH_LOW, s.TH_HIGH
        self._ex = extractor or ORBExtractor()

    def _project(self, frame, landmarks, pp):
        ex = self._ex
        lms = np.ascontiguousarray(landmarks, N.LM_DTYPE)
        L = len(lms)
        midx = np.full(L, -1, np.int32)
        mdist = np.full(L, -1, np.float32)
        n = C.c_int32()
        self.frame_on_device = False
        if frame.n > 0 and frame.kps:                          # the frame's keypoints / descriptors still on the device?  (hs_frame_find + hs_search_by_projection_frame)
            tok = C.c_uint64(0)
            if ex._lib.hs_frame_find(ex.device, frame.kps, frame.n, C.byref(tok)) == N.HS_OK and tok.value:
                st = ex._lib.hs_search_by_projection_frame(ex._h, tok, C.byref(frame), lms.ctypes.data_as(C.c_void_p), L, C.byref(pp),
                                                           midx.ctypes.data_as(C.c_void_p), mdist.ctypes.data_as(C.c_void_p), C.byref(n))
                if st == N.HS_OK:
                    self.frame_on_device = True
                    return midx, mdist, n.value
                if st != N.HS_ERR_INVALID:
                    N.check(ex._h, st)
        N.check(ex._h, ex._lib.hs_search_by_projection(ex._h, C.byref(frame), lms.ctypes.data_as(C.c_void_p), L, C.byref(pp),
                                                       midx.ctypes.data_as(C.c_void_p), mdist.ctypes.data_as(C.c_void_p), C.byref(n)))
        return midx, mdist, n.value

    def SearchByProjection(self, frame, landmarks, th=3.0):
        """SearchByProjection(Frame&, vector<MapPoint*>&, th) — track the local map (FeatureMatcher.cc:123-143)."""
        return self._project(frame, landmarks, N.ProjParams(th, self.TH_HIGH, self.mfNNratio, 0.5, 1.5, 1, 1, 0))

    def SearchByProjectionLastFrame(self, frame, landmarks, th):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono) (FeatureMatcher.cc:145-176): landmarks = LastFrame's map points,
        prev_angle = angle of each one's keypoint in LastFrame."""
        return self._project(frame, landmarks, N.ProjParams(th, self.TH_HIGH, self.mfNNratio, 0.5, 1.5, 0, 1, 1))

    def SearchByProjectionKeyFrame(self, frame, landmarks, th, ORBdist):
        """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) — relocalisation (FeatureMatcher.cc:180-212);
        landmarks = pKF's map points minus sAlreadyFound.  Its rotation criterion is a no-op in the reference (no previous frame)."""
        return self._project(frame, landmarks, N.ProjParams(th, float(ORBdist), 1.0, 0.5, 1.5, 1, 0, 0))

    def SearchForInitialization(self, kps1, desc1, frame2, vbPrevMatched, windowSize=10):
        """SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (FeatureMatcher.cc:404-462).  frame2 is a FrameView (grid
        bounds + keypoints + descriptors).  Returns (vnMatches12, updated vbPrevMatched, number of matches)."""
        ex = self._ex
        k1 = np.ascontiguousarray(kps1, KP_DTYPE); d1 = np.ascontiguousarray(desc1, np.uint8)
        prev = np.ascontiguousarray(vbPrevMatched, np.float32).reshape(-1, 2).copy()
        m = np.full(len(k1), -1, np.int32)
        n = C.c_int32()
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        N.check(ex._h, ex._lib.hs_search_for_initialization(ex._h, p(k1), p(d1), len(k1), C.byref(frame2), p(prev), int(windowSize),
                                                            self.TH_LOW, self.mfNNratio, p(m), C.byref(n)))
        return m, prev, n.value

    def Fuse(self, keyframe, landmarks, th=3.0, reprojection_err=5.99):
        """Fuse(pKF, vpMapPoints, fuse_matches, th, reprojection_err) (FeatureMatcher.cc:464-521).  The caller sets skip = 1 on landmarks
        that are bad, already observed in pKF or protected (:480-485).  Returns per-landmark keypoint indices (first landmark per keypoint)."""
        return self._project(keyframe, landmarks, N.ProjParams(th, self.TH_LOW, 1.0, 0.5, 1.5, use_distance=1, use_stereo=0, check_rotation=0,
                                                               use_prev_matched=0, use_viewing_angle=1, max_view_angle=1.047,
                                                               use_reprojection=1, reproj_threshold=reprojection_err, sigma_ref=1.0, first_wins=1))

    def SearchByProjectionSim3(self, keyframe, Scw, landmarks, vpMatched, th):
        """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) — loop detection, legacy (FeatureMatcher.cc:628-737).  landmarks: min_dist / max_dist =
        the invariance range, skip = bad or already found; vpMatched: uint8[n] (keypoint already has a loop match).  Returns (match per landmark,
        updated vpMatched flags, nmatches)."""
        ex = self._ex
        lms = np.ascontiguousarray(landmarks, N.LM_DTYPE)
        S = np.ascontiguousarray(Scw, np.float32).reshape(16)
        taken = np.ascontiguousarray(vpMatched, np.uint8).copy()
        midx = np.full(len(lms), -1, np.int32)
        n = C.c_int32()
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        N.check(ex._h, ex._lib.hs_search_by_projection_sim3(ex._h, C.byref(keyframe), p(S), p(lms), len(lms), int(th), self.TH_LOW, p(taken), p(midx), C.byref(n)))
        return midx, taken, n.value

    def SearchBySim3(self, kf1, landmarks1, kf2, landmarks2, s12, R12, t12, th):
        """SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) — loop closing, legacy (FeatureMatcher.cc:739-934).  landmarks1[n1] / landmarks2[n2]:
        the landmark of every keypoint (skip = none / bad / already matched).  Returns (match12[n1], nFound)."""
        ex = self._ex
        l1 = np.ascontiguousarray(landmarks1, N.LM_DTYPE); l2 = np.ascontiguousarray(landmarks2, N.LM_DTYPE)
        R = np.ascontiguousarray(R12, np.float32).reshape(9); t = np.ascontiguousarray(t12, np.float32).reshape(3)
        m = np.full(kf1.n, -1, np.int32)
        n = C.c_int32()
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        N.check(ex._h, ex._lib.hs_search_by_sim3(ex._h, C.byref(kf1), p(l1), C.byref(kf2), p(l2), float(s12), p(R), p(t), float(th), self.TH_HIGH, p(m), C.byref(n)))
        return m, n.value

    def SearchForTriangulation(self, kps1, desc1, featvec1, kps2, desc2, featvec2, F12, keep1=None, keep2=None, size_ref=31.0, sigma_ref=1.0):
        """The matching core of SearchForTriangulation (FeatureMatcher.cc:373-402): keep1/keep2 = keypoints WITHOUT a landmark (and with a
        stereo observation when bOnlyStereo), epipolar gate with F12, best match under TH_LOW with ratio 1.0, rotation check."""
        return self.SearchByBoW(kps1, desc1, featvec1, kps2, desc2, featvec2, keep1, True, keep2=keep2, F12=F12, ratio=1.0,
                                size_ref=size_ref, sigma_ref=sigma_ref)

    def SearchByBoWLegacy(self, kps1, desc1, featvec1, kps2, desc2, featvec2, keep1=None, keep2=None):
        """the legacy SearchByBoW(pKF1, pKF2, vpMatches12) (FeatureMatcher.cc:938-1077): a key-frame-2 feature is matched at most once, the
        orientation histogram takes angle1 - angle2.  keep1 / keep2 = views with a good landmark.  Returns (match12, nmatches)."""
        ex = self._ex
        k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        a = [np.ascontiguousarray(x, np.int32) for x in featvec1]; b = [np.ascontiguousarray(x, np.int32) for x in featvec2]
        kp1 = None if keep1 is None else np.ascontiguousarray(keep1, np.uint8)
        kp2 = None if keep2 is None else np.ascontiguousarray(keep2, np.uint8)
        m = np.full(len(k1), -1, np.int32)
        n = C.c_int32()
        p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
        N.check(ex._h, ex._lib.hs_search_by_bow_legacy(ex._h, p(k1), p(d1), len(k1), p(a[0]), p(a[1]), p(a[2]), len(a[0]),
                                                       p(k2), p(d2), len(k2), p(b[0]), p(b[1]), p(b[2]), len(b[0]),
                                                       p(kp1), p(kp2), self.TH_LOW, self.mfNNratio, int(self.mbCheckOrientation), p(m), C.byref(n)))
        return m, n.value

    def SearchByBoW(self, kps1, desc1, featvec1, kps2, desc2, featvec2, keep1=None, check_rotation=True, keep2=None, F12=None, ratio=None,
                    size_ref=31.0, sigma_ref=1.0):
        """The matching core of SearchByBoW / SearchByBoW2 (FeatureMatcher.cc:216-371).  featvec = (node_id, node_ptr, idx) CSR arrays."""
        ex = self._ex
        k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        a = [np.ascontiguousarray(x, np.int32) for x in featvec1]
        b = [np.ascontiguousarray(x, np.int32) for x in featvec2]
        keep = None if keep1 is None else np.ascontiguousarray(keep1, np.uint8)
        kp2 = None if keep2 is None else np.ascontiguousarray(keep2, np.uint8)
        Fm = None if F12 is None else np.ascontiguousarray(F12, np.float32).reshape(9)
        m = np.full(len(k1), -1, np.int32)
        n = C.c_int32()
        p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
        N.check(ex._h, ex._lib.hs_search_by_bow_ex(ex._h, p(k1), p(d1), len(k1), p(a[0]), p(a[1]), p(a[2]), len(a[0]),
                                                   p(k2), p(d2), len(k2), p(b[0]), p(b[1]), p(b[2]), len(b[0]),
                                                   p(keep), p(kp2), p(Fm), size_ref, sigma_ref, self.TH_LOW,
                                                   self.mfNNratio if ratio is None else ratio, int(check_rotation), p(m), C.byref(n)))
        return m, n.value

    def HammingKnn2(self, query, train):
        ex = self._ex
        q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        bi, bd, sd = (np.zeros(len(q), np.int32) for _ in range(3))
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        N.check(ex._h, ex._lib.hs_hamming_knn2(ex._h, p(q), len(q), p(t), len(t), p(bi), p(bd), p(sd)))
        return bi, bd, sd


class ORBVocabulary:
    """HYSLAM::ORBVocabulary::transform (src/features/low_level/ORBVocabulary.cpp:31-42) over a flat vocabulary tree
    (_native.VocabTree; DBoW2 and ORBvoc are external to the reference).  `transform` returns the two containers Frame::ComputeBoW fills:
    the BoW vector {word id: L1-normalised tf-idf weight} and the feature vector as CSR (node ids ascending, node_ptr, indices ascending)."""

    def __init__(self, tree, extractor=None):
        """tree: a _native.VocabTree, or the path of a DBoW2 vocabulary file (".txt" = text format, else binary; ORBVocabulary.cpp:14-29)"""
        self._vocab = None
        if isinstance(tree, (str, bytes)):
            L = N.lib()
            v = C.c_void_p()
            st = L.hs_vocab_load(tree.encode() if isinstance(tree, str) else tree, C.byref(v))
            if st != N.HS_OK:
                raise HsError(st, "Wrong path to vocabulary. Failed to open at: %s" % tree)       # the reference prints this and exits (ORBVocabulary.cpp:22-27)
            self._vocab = v
            tree = N.VocabTree()
            L.hs_vocab_get_tree(v, C.byref(tree))
        self.tree = tree
        self._ex = extractor or ORBExtractor()

    def size(self):
        """ORBVocabulary::size(): number of words"""
        if self._vocab is not None:
            n = C.c_int32()
            N.lib().hs_vocab_info(self._vocab, None, None, None, C.byref(n), None, None)
            return n.value
        cc = np.ctypeslib.as_array(C.cast(self.tree.child_count, C.POINTER(C.c_int32)), shape=(self.tree.n_nodes,))
        return int((cc[1:] == 0).sum())

    def __del__(self):
        try:
            if self._vocab is not None:
                N.lib().hs_vocab_destroy(self._vocab)
                self._vocab = None
        except Exception:
            pass

    def transform(self, descriptors, levelsup=4):
        ex = self._ex
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        w = np.zeros(n, np.int32); wt = np.zeros(n, np.float32); nd = np.zeros(n, np.int32)
        N.check(ex._h, ex._lib.hs_bow_transform(ex._h, C.byref(self.tree), d.ctypes.data_as(C.c_void_p), n, levelsup,
                                                w.ctypes.data_as(C.c_void_p), wt.ctypes.data_as(C.c_void_p), nd.ctypes.data_as(C.c_void_p)))
        return self.containers(w, wt, nd)

    @staticmethod
    def containers(word, weight, node):
        """DBoW2: `if (w > 0) { bow.addWeight(id, w); fv.addFeature(nid, i); }`, then L1 normalisation of the BoW vector."""
        use = weight > 0
        bow = {}
        for wid, wv in zip(word[use].tolist(), weight[use].tolist()):
            bow[wid] = bow.get(wid, 0.0) + wv
        tot = sum(abs(v) for _, v in sorted(bow.items()))
        if tot > 0:
            bow = {k: v / tot for k, v in bow.items()}
        idx = np.nonzero(use)[0]
        order = np.lexsort((idx, node[idx]))
        ids, counts = np.unique(node[idx], return_counts=True)
        ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        return bow, (ids.astype(np.int32), ptr, idx[order].astype(np.int32)), (word, weight, node)


class ORBFactory:
    """HYSLAM::ORBFactory: hands out extractors and matcher settings (FeatureFactory.h:21-33, ORBFactory.cpp:13-45)."""

    def __init__(self, extractor_settings=None, matcher_settings=None, device=0):
        self.extractor_settings = extractor_settings or FeatureExtractorSettings()
        self.matcher_settings = matcher_settings or FeatureMatcherSettings()
        self.device = device

    def getExtractor(self, settings=None):
        return ORBExtractor(settings or self.extractor_settings, self.device)

    def getFeatureMatcher(self, extractor=None):
        return FeatureMatcher(self.matcher_settings, extractor)

    def getFeatureExtractorSettings(self):
        return self.extractor_settings

    def getFeatureMatcherSettings(self):
        return self.matcher_settings

    def setFeatureMatcherSettings(self, s):
        self.matcher_settings = s
