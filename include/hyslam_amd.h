/* hyslam_amd.h — C ABI of the MI355X-native ORB extract + Hamming match path for hySLAM.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch/OpenCV
 * types, int status codes, caller-allocated outputs.  A hySLAM maintainer binds it from a
 * `HipORBExtractor : FeatureExtractor` / `HipORBFactory : FeatureFactory` adaptor (see
 * INTEGRATION.md and hyslam_amd/host/).  File:line citations are relative to the reference
 * repository (bmhopkinson/hyslam).
 *
 * Threading: a handle is thread-compatible (one thread at a time); distinct handles are fully
 * concurrent — this matches the reference, which runs two separate extractor instances for the
 * left and right image (src/main/ImageProcessing.cpp:31-32,82-84).  No global mutable state.
 *
 * Device pointers: every `*_device` entry point takes pointers into HBM of the handle's device
 * and enqueues work on `stream` (a hipStream_t passed as void*; NULL = the handle's own stream)
 * without synchronising.  The plain entry points take host pointers, stage through pinned
 * buffers and synchronise before returning.
 */
#ifndef HYSLAM_AMD_H
#define HYSLAM_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HS_DESC_BYTES 32          /* ORBFinder::descriptor_cols(), src/features/low_level/ORBFinder.h:88 */

enum hs_status {
    HS_OK = 0,
    HS_ERR_INVALID = 1,           /* bad argument / unsupported parameter combination            */
    HS_ERR_HIP = 2,               /* a HIP runtime call failed; see hs_*_last_error               */
    HS_ERR_CAPACITY = 3,          /* caller's output capacity too small (nothing partial written) */
    HS_ERR_NO_DEVICE = 4          /* no usable gfx950 device                                       */
};

/* The cv::KeyPoint fields the reference sets (ORBExtractor.cpp:478-487,546-552; class_id stays -1). */
typedef struct hs_keypoint {
    float x, y;                   /* level-0 pixel coordinates (level coords * scale[octave])     */
    float size;                   /* (int)(31 * scale[octave])                                     */
    float angle;                  /* degrees [0,360), intensity centroid on the blurred level      */
    float response;               /* FAST corner score                                             */
    int32_t octave;
} hs_keypoint;

/* HYSLAM::FeatureExtractorSettings (src/core/FeatureExtractorSettings.h:19-32) + the two knobs the
 * reference hard-wires. */
typedef struct hs_orb_params {
    int32_t nfeatures;            /* nFeatures                                                     */
    float   scale_factor;         /* fScaleFactor                                                  */
    int32_t nlevels;              /* nLevels (1..16)                                               */
    int32_t cell_px;              /* N_CELLS: FAST cell edge in pixels (ORBExtractor.cpp:409)      */
    int32_t ini_th_fast;          /* init_threshold: accepted, unused — reference quirk, ORBFinder.cpp:58-60 */
    int32_t min_th_fast;          /* min_threshold:  accepted, unused — idem                       */
    int32_t fast_threshold;       /* effective FAST threshold; the reference always runs 20        */
    uint16_t blur_taps[7];        /* 7-tap Gaussian, unsigned 8.8 fixed point; all 0 => 18,34,49,55,49,34,18 */
    uint16_t _pad;
} hs_orb_params;

/* What Stereomatcher reads from Camera and FeatureMatcherSettings (src/features/Stereomatcher.cpp:7-24). */
typedef struct hs_stereo_params {
    float   fx;                   /* Camera::fx()                                                   */
    float   mbf;                  /* Camera::mbf                                                    */
    int32_t n_rows;               /* (int)Camera::mnMaxY                                            */
    float   th_high, th_low;      /* FeatureMatcherSettings::TH_HIGH / TH_LOW (FeatureMatcher.h:98-103) */
    float   size_ref;             /* FeatureExtractorSettings::size_ref (31)                        */
} hs_stereo_params;

typedef struct hs_orb hs_orb;     /* one FeatureExtractor instance (+ its device workspace and stream) */

/* ---- library ---- */
const char* hs_version(void);
const char* hs_status_string(int status);
int hs_device_count(int* count);

/* ---- extractor: replaces HYSLAM::ORBExtractor behind FeatureExtractor (src/features/FeatureExtractor.h:25-37) ---- */
/* defaults of ORBFactory::ORBFactory(), src/features/ORBFactory.cpp:13-25 */
void hs_orb_default_params(hs_orb_params* p);
/* ORBFactory::getExtractor(settings) -> ORBExtractor ctor, src/features/ORBFactory.cpp:37-40, ORBExtractor.cpp:76-119 */
int  hs_orb_create(const hs_orb_params* p, int device, hs_orb** out);
void hs_orb_destroy(hs_orb* h);
const char* hs_orb_last_error(const hs_orb* h);
/* GetLevels / GetScaleFactor(s) / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares,
 * FeatureExtractor.h:31-36.  Any output pointer may be NULL; arrays hold nlevels entries. */
int  hs_orb_get_levels(const hs_orb* h);
int  hs_orb_get_device(const hs_orb* h);          /* the device index given to hs_orb_create (-1 for a NULL handle) */
float hs_orb_get_scale_factor(const hs_orb* h);
int  hs_orb_get_scale_tables(const hs_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                             int32_t* features_per_level);
/* smallest `cap` that can never overflow: nfeatures + per-level overshoot of DistributeOctTree.  Frames wider than 8:1 start the
 * quadtree with more than 8 root nodes per level and need more: call hs_orb_reserve() for the frame size first, the value then covers it. */
int  hs_orb_max_keypoints(const hs_orb* h);
/* optional: size the device workspace up front for `batch` images of w x h (grows lazily otherwise) */
int  hs_orb_reserve(hs_orb* h, int w, int h_px, int batch);

/* ORBExtractor::operator()(image, mask, keypoints, descriptors), ORBExtractor.cpp:496-562.
 * img: CV_8UC1 rows of `stride` bytes (host).  Writes *n keypoints (<= cap) and n*32 descriptor bytes.
 * An empty image (w==0||h==0||!img) returns HS_OK with *n = 0, like the reference's silent return (:499-500). */
int  hs_orb_extract(hs_orb* h, const uint8_t* img, int w, int h_px, int stride,
                    hs_keypoint* kps, uint8_t* desc, int cap, int32_t* n);
/* `batch` same-sized host images; outputs are [batch][cap] / [batch][cap][32] / [batch]. */
int  hs_orb_extract_batch(hs_orb* h, const uint8_t* const* imgs, int batch, int w, int h_px, int stride,
                          hs_keypoint* kps, uint8_t* desc, int cap, int32_t* n);
/* ---- the camera frame as it arrives: ImageProcessing::PreProcessImg on the device (src/main/ImageProcessing.cpp:118-138; it sits inside the
 * reference's timing bracket, :70 / :112, in front of both extractor calls, :76-77).  cv::resize(img, img, Size(), scale, scale) on the 1-, 3- or
 * 4-channel 8-bit frame — a copy at scale 1, the rounded 2x2 mean INTER_LINEAR silently becomes at exactly 0.5 (the reference's "Imaging" camera:
 * 2704 x 2028 x 3 -> 1352 x 1014), OpenCV's 11-bit fixed-point bilinear otherwise — then cvtColor to grey with its 14-bit weights (R 4899, G 9617,
 * B 1868); `rgb` = the camera's `RGB:` key (1: channel 0 is red: CV_RGB(A)2GRAY, 0: CV_BGR(A)2GRAY).  The scaled size is cvRound(w * scale). */
typedef struct hs_preprocess_params { int32_t channels; int32_t rgb; float scale; int32_t _pad; } hs_preprocess_params;
void hs_preprocess_size(int w, int h_px, float scale, int32_t* ow, int32_t* oh);
/* device frames (image i at d_src + i * image_stride, rows `row_stride` bytes apart, channels interleaved) -> device grey frames of the scaled size;
 * asynchronous on `stream` (the handle's own when 0).  Only the ow x oh pixels of every grey frame are written. */
int  hs_preprocess_device(hs_orb* h, const uint8_t* d_src, int w, int h_px, size_t row_stride, size_t image_stride, int batch, const hs_preprocess_params* pp,
                          uint8_t* d_grey, size_t grey_row_stride, size_t grey_image_stride, void* stream);
/* ProcessMonoImage / ProcessStereoImage's `PreProcessImg(...)` + `(*extractor)(mImGray, ...)` in one call (ImageProcessing.cpp:44,55 / :76-77,82-83): `batch`
 * host frames of w x h_px x pp->channels cross PCIe as they are, are reduced to grey level-0 frames on the device and extracted; outputs as
 * hs_orb_extract_batch ([batch][cap] ...; cap >= hs_orb_max_keypoints after hs_orb_reserve(h, ow, oh, batch) with (ow, oh) = hs_preprocess_size).
 * grey_out (may be NULL): the grey frames, batch x oh x ow tight — what the reference keeps as track_data.image (:60,108). */
int  hs_orb_extract_camera_batch(hs_orb* h, const uint8_t* const* imgs, int batch, int w, int h_px, size_t row_stride, const hs_preprocess_params* pp,
                                 hs_keypoint* kps, uint8_t* desc, int cap, int32_t* n, uint8_t* grey_out);
/* the same through the pipelined ingest (hs_orb_submit_batch's ticket machinery, two tickets in flight): the camera's frames are copied in on the copy-in
 * stream, PreProcessImg runs on the compute stream in front of the pyramid; with `sp` the batch is [left frames | right frames] of batch / 2 stereo pairs and the
 * ticket also carries uRight / depth — ProcessStereoImage's PreProcessImg x 2 + extractors + Stereomatcher (ImageProcessing.cpp:76-103) as ONE ticket.
 * Results by hs_orb_wait as for any ticket (cap >= hs_orb_max_keypoints for the SCALED size). */
int  hs_orb_submit_camera_batch(hs_orb* h, const uint8_t* const* imgs, int batch, int w, int h_px, size_t row_stride, const hs_preprocess_params* pp,
                                const hs_stereo_params* sp, int32_t* ticket);

/* Device-resident batch: image i starts at d_imgs + i*image_stride, rows `row_stride` bytes apart.
 * d_kps [batch][cap], d_desc [batch][cap][32] (16-byte aligned), d_n [batch]; all device memory.  Asynchronous.
 * ONE STREAM AT A TIME PER HANDLE: the *_device entry points take a caller stream, but a handle's workspace, the FAST kernel's work-queue counter
 * rotation and its spill halves assume that every call on the handle is ordered after the previous one — drive a handle from one stream (or order the
 * streams yourself); use separate handles for concurrent streams. */
int  hs_orb_extract_batch_device(hs_orb* h, const uint8_t* d_imgs, int batch, int w, int h_px,
                                 size_t row_stride, size_t image_stride,
                                 hs_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n, int cap, void* stream);

/* ---- pipelined host ingest: the reference's bounded frame queue (System::TrackStereo throttles the producer at more than two waiting frames,
 * src/main/System.cc:194-196; ImageProcessing pops, extracts, matches: src/main/ImageProcessing.cpp:69-116) ----
 * hs_orb_submit_batch enqueues `batch` same-sized host frames and returns a ticket at once: the frames are copied in on a copy stream while the
 * kernels of the previously submitted batch still run, the results leave on a third stream into page-locked memory of the handle.  With `sp`
 * != NULL the batch is batch/2 stereo pairs — images [0, batch/2) left, [batch/2, batch) right — and the stereo matcher runs too.
 * hs_orb_wait blocks until that ticket's results are on the host and copies them out: kps [batch][cap], desc [batch][cap][32], n [batch],
 * uRight / depth [batch/2][cap] (stereo tickets only; NULL otherwise); cap >= hs_orb_max_keypoints().
 * At most TWO tickets may be in flight (two staging slots): a third submit returns HS_ERR_INVALID until the oldest was waited for.
 * Frames in page-locked memory (hs_host_alloc, or the caller's own hipHostMalloc / hipHostRegister) are DMA'd at link speed; pageable frames
 * work too and go through the runtime's staging path.
 * LIFETIME OF THE FRAMES: hs_orb_submit_batch returns BEFORE the frames have been read (page-locked frames are DMA'd asynchronously) — they must stay
 * valid and UNCHANGED until hs_orb_wait (or hs_orb_cancel) for that ticket returns, or until hs_ticket_frames_copied(h, ticket) returns 1 (the
 * copy-in of that ticket is complete: a capture buffer may be recycled from then on; 0 = not yet, -1 = unknown ticket).  A capture loop that reuses
 * its buffer right after submit gets silently corrupted features.
 * hs_orb_cancel(h, ticket): give up a ticket — waits until its batch has drained, drops the results, frees the slot.
 * A failed hs_orb_submit_batch leaves no ticket and no work behind (the streams are drained before it returns); a hs_orb_wait that fails with
 * HS_ERR_CAPACITY / HS_ERR_INVALID keeps the ticket (call again with valid arguments), one that fails with HS_ERR_HIP releases it. */
int  hs_host_alloc(size_t bytes, void** out);
void hs_host_free(void* p);
int  hs_orb_submit_batch(hs_orb* h, const uint8_t* const* imgs, int batch, int w, int h_px, int stride, const hs_stereo_params* sp, int32_t* ticket);
int  hs_orb_wait(hs_orb* h, int32_t ticket, hs_keypoint* kps, uint8_t* desc, int32_t* n, int cap, float* uRight, float* depth);
int  hs_orb_cancel(hs_orb* h, int32_t ticket);
int  hs_ticket_frames_copied(hs_orb* h, int32_t ticket);

/* ---- stereo: replaces Stereomatcher::computeStereoMatches + getData, src/features/Stereomatcher.cpp:26-156 ---- */
/* uRight[nL], depth[nL]: -1 where there is no stereo match (Stereomatcher.h:44-47). */
int  hs_stereo_match(hs_orb* h, const hs_keypoint* kpsL, const uint8_t* descL, int nL,
                     const hs_keypoint* kpsR, const uint8_t* descR, int nR,
                     const hs_stereo_params* sp, float* uRight, float* depth);
/* Device batch of `pairs` stereo pairs laid out like the extractor's outputs (stride `cap` per frame).
 * d_uRight / d_depth: [pairs][cap].  Asynchronous. */
int  hs_stereo_match_batch_device(hs_orb* h, const hs_keypoint* d_kpsL, const uint8_t* d_descL, const int32_t* d_nL,
                                  const hs_keypoint* d_kpsR, const uint8_t* d_descR, const int32_t* d_nR,
                                  int pairs, int cap, const hs_stereo_params* sp,
                                  float* d_uRight, float* d_depth, void* stream);

/* ImageProcessing::ProcessStereoImage's compute (src/main/ImageProcessing.cpp:82-84,100-103) for a
 * device-resident batch: extract left and right frames, then stereo-match each pair.  Asynchronous. */
int  hs_stereo_frontend_batch_device(hs_orb* h, const uint8_t* d_left, const uint8_t* d_right, int pairs,
                                     int w, int h_px, size_t row_stride, size_t image_stride,
                                     hs_keypoint* d_kpsL, uint8_t* d_descL, int32_t* d_nL,
                                     hs_keypoint* d_kpsR, uint8_t* d_descR, int32_t* d_nR, int cap,
                                     const hs_stereo_params* sp, float* d_uRight, float* d_depth, void* stream);

/* Concurrency inside one handle: with lanes = 2 the batched device entry points (hs_orb_extract_batch_device,
 * hs_stereo_frontend_batch_device) split their batch in two halves that run on two streams with separate workspaces (the second lane
 * is an internal child handle).  The kernels of this path are latency-bound rather than bandwidth-bound, so two interleaved launch
 * sequences fill each other's stalls (+15 % pairs/s measured).  Ordering towards the caller is unchanged: all work is complete when the
 * caller's stream reaches the point after the call.  Default 1. */
int  hs_orb_set_lanes(hs_orb* h, int lanes);

/* The launch sequence of an extraction can be SPLIT: level 0 needs no pyramid, so its FAST + quadtree can run on a second stream of the handle
 * beside the pyramid and the other levels' FAST + quadtree (joined before the describe stage; same kernels, same results).  mode -1 (default): split
 * one or two large frames (>= 6 Mpx per call); 0: never; 1: always.  Measured (profiles/README.md, row "C4"; profiles/r03_bench_lines.json -> c4): the
 * 4000 x 3000 "Imaging" extraction 0.413 ms unsplit (round 2) -> 0.207 ms split; for an isolated 1080p pair the fork / join between the streams costs
 * more than the overlap saves (0.132 -> 0.160 ms in round 3), but when several handles share the GPU (hySLAM's SLAM stereo camera + Imaging camera,
 * BASELINE config 4) splitting BOTH lets their kernels interleave: 2 592 -> 4 068 steps/s.  Ignored while stage events are on.
 * The two concurrent FAST launches of a split call rely on every earlier launch of the handle having completed — one more reason for the
 * one-stream-at-a-time rule of the *_device entry points (see hs_orb_extract_batch_device). */
int  hs_orb_set_split(hs_orb* h, int mode);

/* block until everything enqueued on the handle's own stream (or `stream`) has finished */
int  hs_orb_synchronize(hs_orb* h, void* stream);

/* ================= matchers on flat arrays: the cores of HYSLAM::FeatureMatcher (src/features/FeatureMatcher.h:105-176) =================
 * The C++ adaptor gathers Frame / KeyFrame / MapPoint fields into these arrays and replays associations
 * (Frame::associateLandMark) from the returned per-landmark results, in the reference's order.  Containers the
 * reference orders by MapPoint* address are replaced by landmark ARRAY ORDER (pass landmarks sorted by address to
 * reproduce its iteration order).  Host pointers; synchronous. */

/* what _SearchByProjection_ reads from Frame / Camera / FeatureViews / LandMarkMatches
 * (src/core/Frame.cc:45-72,137-180,416-469; src/core/Camera.cpp:116-153) */
typedef struct hs_frame_view {
    float Rcw[9], tcw[3], Ow[3];       /* mRcw (row-major), mtcw, mOw                                  */
    float fx, fy, cx, cy, mbf;         /* K, Camera::mbf                                                */
    int32_t sensor;                    /* Camera::sensor: 0 mono, 1 stereo, 2 RGBD                      */
    float min_x, max_x, min_y, max_y;  /* mnMinX .. mnMaxY                                              */
    float size_ref;                    /* views.orbParams().size_ref (31)                               */
    int32_t n;                         /* number of keypoints                                           */
    const hs_keypoint* kps;            /* [n]                                                           */
    const uint8_t* desc;               /* [n][32]                                                       */
    const float* uR;                   /* [n], < 0 = no stereo correspondence                           */
    const int32_t* kp_lm_obs;          /* [n]: -1 = keypoint has no landmark, else Observations() of it */
} hs_frame_view;

/* the MapPoint fields the matchers read (src/core/MapPoint.h:54-169) */
typedef struct hs_landmark {
    float pos[3];                      /* GetWorldPos()                                                 */
    float size;                        /* getSize(), world units                                        */
    float min_dist, max_dist;          /* mfMinDistance, mfMaxDistance (before the 0.8 / 1.2 factors, MapPoint.cc:139-149) */
    float normal[3];                   /* GetNormal()                                                   */
    int32_t assoc_kp;                  /* Frame::hasAssociation(lm) in THIS frame, -1 if none (Frame.cc:296-300) */
    float prev_angle;                  /* angle of its keypoint in the previous frame (rotation check)  */
    int32_t skip;                      /* 1 = nullptr entry                                             */
    uint8_t desc[32];                  /* GetDescriptor()                                               */
} hs_landmark;

typedef struct hs_proj_params {
    float th;                          /* search radius factor                                          */
    float score_threshold;             /* BestScoreCriterion threshold: TH_HIGH or ORBdist              */
    float second_best_ratio;           /* mfNNratio or 1.0                                              */
    float frac_smaller, frac_larger;   /* FeatureSizeCriterion(0.5, 1.5)                                */
    int32_t use_distance;              /* DistanceCriterion among the landmark criteria                 */
    int32_t use_stereo;                /* StereoConsistencyCriterion(th)                                */
    int32_t check_rotation;            /* RotationConsistencyCriterion (uses prev_angle)                */
    int32_t use_prev_matched;          /* PreviouslyMatchedCriterion (all Frame variants; not Fuse)     */
    int32_t use_viewing_angle;         /* ViewingAngleCriterion(max_view_angle): Fuse, FeatureMatcher.cc:469 */
    float   max_view_angle;            /* radians (1.047)                                               */
    int32_t use_reprojection;          /* ProjectionViewCriterion(reproj_threshold): Fuse, :473         */
    float   reproj_threshold;          /* 5.99                                                          */
    float   sigma_ref;                 /* FeatureExtractorSettings::sigma_ref (1.0), determineSigma2    */
    int32_t first_wins;                /* Fuse: the first landmark that matched a keypoint keeps it (:515) */
    int32_t dist_is_invariance_range;  /* 1: hs_landmark::min_dist / max_dist already hold GetMinDistanceInvariance() / GetMaxDistanceInvariance()
                                          (= 0.8f*mfMinDistance, 1.2f*mfMaxDistance, MapPoint.cc:139-149) — what a hySLAM adaptor can read */
} hs_proj_params;

/* FeatureMatcher::_SearchByProjection_ (FeatureMatcher.cc:57-121) with the criteria of SearchByProjection(Frame, MapPoints, th)
 * (:123-143: use_distance=1,use_stereo=1,check_rotation=0), (CurrentFrame, LastFrame, th, bMono) (:145-176: 0,1,1) and
 * (CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (:180-212: 1,0,0, ratio 1.0), and of Fuse(pKF, MapPoints, matches, th, err)
 * (:464-521: distance + viewing angle, size + reprojection + best score TH_LOW/1.0, first landmark per keypoint wins; the caller
 * marks bad / already-observed / protected landmarks with skip = 1).  match_idx[L] = keypoint index or -1,
 * match_dist[L] = Hamming distance of the match, *n_matches = matches.size(). */
int  hs_search_by_projection(hs_orb* h, const hs_frame_view* F, const hs_landmark* lms, int L, const hs_proj_params* pp,
                             int32_t* match_idx, float* match_dist, int32_t* n_matches);

/* Frame::AssignFeaturesToGrid / PosInGrid (src/core/Frame.cc:137-153,459-469): the 64 x 48 grid cell of every keypoint, cell = round((x - mnMinX) *
 * 64 / (mnMaxX - mnMinX)) (nearest, not floor), -1 when outside.  cell_xy [n][2] int8 (column, row).  The matchers build this grid on the device
 * themselves; the entry point exposes it for Frame construction on the host side and for tests.  Host pointers; synchronous. */
int  hs_frame_grid(hs_orb* h, const hs_frame_view* F, int8_t* cell_xy);

/* the same on device-resident data (FeatureViews stay in HBM between extraction and tracking): every pointer inside *F and
 * d_lms / d_match_idx / d_match_dist / d_n_matches are device pointers (F->kps / F->desc can be the extractor's own outputs).  Asynchronous. */
int  hs_search_by_projection_device(hs_orb* h, const hs_frame_view* F, const hs_landmark* d_lms, int L, const hs_proj_params* pp,
                                    int32_t* d_match_idx, float* d_match_dist, int32_t* d_n_matches, void* stream);

/* ---- device-resident frames: a frame's keypoints and descriptors stay in HBM between ImageProcessing and Tracking ----
 * The reference copies them into FeatureViews (src/core/FeatureViews.h:20-81, built in ImageProcessing.cpp:85,100 and stored by the Frame
 * constructor, src/core/Frame.cc:45-72), and every matcher call reads them back out of those host objects.  Here the extractor can keep what it
 * just produced on the device, and the matchers take it from there:
 *   hs_frame_publish   right after a host-pointer extraction (hs_orb_extract / hs_orb_extract_batch / hs_orb_wait) on `h`: keeps image `image` of
 *                      that call — device-to-device, on the handle's stream, no host round trip — in a per-device cache of 16 slots (oldest reused
 *                      first).  kps[n] = the keypoints the call returned for that image (kept beside the slot to recognise the frame later).
 *   hs_frame_find      hySLAM has no field that could carry a token through FeatureViews / Frame: a frame is recognised by its keypoint array
 *                      (exact comparison of all n records).  HS_ERR_INVALID when no live slot of `device` holds it.
 *   hs_frame_release   optional: give a slot back early.      hs_frame_info: its keypoint count.      hs_frame_cache_clear: free a device's cache.
 *   hs_search_by_projection_frame   hs_search_by_projection with F->kps / F->desc taken from the cache (both may be NULL in *F; F->n must equal the
 *                      published count; F->uR / F->kp_lm_obs are host arrays as before: they change between calls).
 *   hs_stereo_match_frames          hs_stereo_match on two published frames (left, right).
 * A token whose slot has been reused is unknown again: the call returns HS_ERR_INVALID and the caller uses the host-pointer entry point (the C++
 * adaptors do).  Tokens are per device; the calls are thread-safe against each other; a slot that a call is reading is never refilled. */
typedef uint64_t hs_frame_token;    /* 0 = none */
int  hs_frame_publish(hs_orb* h, int image, const hs_keypoint* kps, int n, hs_frame_token* token);
int  hs_frame_find(int device, const hs_keypoint* kps, int n, hs_frame_token* token);
int  hs_frame_release(int device, hs_frame_token token);
int  hs_frame_info(int device, hs_frame_token token, int32_t* n);
int  hs_frame_cache_clear(int device);  /* frees a device's cache (every token of it becomes unknown); HS_ERR_INVALID while a call is reading a slot.  The cache
                                            otherwise lives as long as the process: call this before unloading the library or resetting the device */
int  hs_search_by_projection_frame(hs_orb* h, hs_frame_token frame, const hs_frame_view* F, const hs_landmark* lms, int L, const hs_proj_params* pp,
                                   int32_t* match_idx, float* match_dist, int32_t* n_matches);
int  hs_stereo_match_frames(hs_orb* h, hs_frame_token left, hs_frame_token right, const hs_stereo_params* sp, float* uRight, float* depth);

/* ---- legacy loop-closing matchers (the reference keeps them for LoopClosing, which is a stub: src/main/System.cc:149) ----
 * FeatureMatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (FeatureMatcher.cc:628-737).  KF = the keyframe (its OWN pose is used by
 * landMarkSizePixels, KeyFrame.cc:258-279 — reference quirk), Scw = the caller's Sim3 as a row-major 4x4.  lms[L] in vpPoints order with
 * min_dist / max_dist = GetMin/MaxDistanceInvariance() and skip = pMP->isBad() || already in vpMatched.  kp_matched[KF->n] (in/out) =
 * vpMatched[idx] != NULL.  The search is sequential by definition: a landmark cannot take a keypoint that an earlier landmark took (:713,731);
 * everything that does not depend on vpMatched runs in parallel first, the assignment walks the landmarks in order on one wavefront.
 * match_idx[L] = keypoint taken by landmark i or -1; *n_matches = nmatches.  Host pointers; synchronous. */
int  hs_search_by_projection_sim3(hs_orb* h, const hs_frame_view* KF, const float* Scw, const hs_landmark* lms, int L, int th, float th_low,
                                  uint8_t* kp_matched, int32_t* match_idx, int32_t* n_matches);
/* FeatureMatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (FeatureMatcher.cc:739-934): lms1[KF1->n] / lms2[KF2->n] = the landmark
 * of each keypoint (skip = none, bad or already matched; assoc_kp = its view in the OTHER keyframe; min_dist / max_dist = the invariance range).
 * Both directions run in parallel, then the agreement check.  match12[KF1->n] = KF2 keypoint index or -1; *n_found = nFound. */
int  hs_search_by_sim3(hs_orb* h, const hs_frame_view* KF1, const hs_landmark* lms1, const hs_frame_view* KF2, const hs_landmark* lms2,
                       float s12, const float* R12, const float* t12, float th, float th_high, int32_t* match12, int32_t* n_found);

/* the inner loops of SearchByBoW / SearchByBoW2 / _SearchByBoW_ (FeatureMatcher.cc:216-371): for every vocabulary node present in
 * both feature vectors, best / second-best Hamming of each side-1 index over the node's side-2 indices (BestMatchBoWCriterion,
 * MatchCriteria.cpp:601-635: d < threshold and d < ratio*d2, both strict), then RotationConsistencyBoW (:679-726).
 * Feature vectors (DBoW2::FeatureVector) as CSR: node ids ascending, node_ptr[n_nodes+1], idx[].  keep1[n1] (may be NULL) = 1 for
 * side-1 indices that pass the index criteria (PreviouslyMatchedIndexCriterion).  match12[n1] = side-2 index or -1. */
int  hs_search_by_bow(hs_orb* h, const hs_keypoint* kps1, const uint8_t* desc1, int n1,
                      const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int n_nodes1,
                      const hs_keypoint* kps2, const uint8_t* desc2, int n2,
                      const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int n_nodes2,
                      const uint8_t* keep1, float score_threshold, float second_best_ratio, int check_rotation,
                      int32_t* match12, int32_t* n_matches);
/* the same with the index criteria on BOTH sides (keep2, may be NULL: _SearchByBoW_, FeatureMatcher.cc:306-309) and, when F12 (row-major
 * 3x3, may be NULL) is given, EpipolarConsistencyBoWCriterion (MatchCriteria.cpp:641-676: dsqr < 3.84*sigma2(kp2.size)) ahead of the
 * best-match criterion — the core of SearchForTriangulation (FeatureMatcher.cc:373-402; threshold TH_LOW, ratio 1.0). */
int  hs_search_by_bow_ex(hs_orb* h, const hs_keypoint* kps1, const uint8_t* desc1, int n1,
                         const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int n_nodes1,
                         const hs_keypoint* kps2, const uint8_t* desc2, int n2,
                         const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int n_nodes2,
                         const uint8_t* keep1, const uint8_t* keep2, const float* F12, float size_ref, float sigma_ref,
                         float score_threshold, float second_best_ratio, int check_rotation,
                         int32_t* match12, int32_t* n_matches);

/* The legacy FeatureMatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (FeatureMatcher.cc:938-1077; hySLAM never calls it — "aim to replace this with
 * SearchByBoW2" — bound for completeness of the FeatureMatcher surface): as hs_search_by_bow_ex without the epipolar gate, but a side-2 feature
 * can be matched only once (vbMatched2: the side-1 features of a node are processed in list order, one wavefront per shared node) and the
 * orientation histogram takes angle1 - angle2.  keep1 / keep2 = the view has a landmark that is not bad. */
int  hs_search_by_bow_legacy(hs_orb* h, const hs_keypoint* kps1, const uint8_t* desc1, int n1,
                             const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int n_nodes1,
                             const hs_keypoint* kps2, const uint8_t* desc2, int n2,
                             const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int n_nodes2,
                             const uint8_t* keep1, const uint8_t* keep2, float th_low, float nnratio, int check_orientation,
                             int32_t* match12, int32_t* n_matches);

/* FeatureMatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (FeatureMatcher.cc:404-462; monocular map
 * initialisation, MonoInitializer.cpp:83).  Inherently sequential over F1's keypoints — a later keypoint takes over an F2 keypoint only
 * with a strictly smaller distance (MatchCriteria.cpp:525-549) — so one workgroup walks F1 in order and parallelises the window search
 * and the best / second-best reduction of each step.  F2 supplies the grid bounds, keypoints and descriptors (pose fields unused).
 * prev_matched_xy [n1][2] is vbPrevMatched (in/out), matches12 [n1] is vnMatches12.  Host pointers; synchronous. */
int  hs_search_for_initialization(hs_orb* h, const hs_keypoint* kps1, const uint8_t* desc1, int n1, const hs_frame_view* F2,
                                  float* prev_matched_xy, int window, float th_low, float nnratio,
                                  int32_t* matches12, int32_t* n_matches);

/* Vocabulary transform: what Frame::ComputeBoW / KeyFrame::ComputeBoW obtain from ORBVocabulary::transform -> DBoW2::TemplatedVocabulary<FORB>
 * ::transform(features, bow, fv, levelsup=4) (src/core/Frame.cc:472-479, src/features/low_level/ORBVocabulary.cpp:31-42).  DBoW2 and the
 * ORBvoc data are not part of the reference tree; the tree descent follows DBoW2's published algorithm (first minimum wins at every level).
 * Flat tree: node 0 = root, the children of a node are contiguous, child_count == 0 marks a leaf (word).  Outputs per descriptor: word id,
 * word weight and the id of the node passed at level (levels - levelsup), which keys DBoW2::FeatureVector.  Host pointers; synchronous. */
typedef struct hs_vocab_tree {
    int32_t n_nodes, levels;
    const int32_t* child_begin;        /* [n_nodes] */
    const int32_t* child_count;        /* [n_nodes] */
    const uint8_t* desc;               /* [n_nodes][32] */
    const int32_t* word_id;            /* [n_nodes], valid at leaves */
    const float* weight;               /* [n_nodes], valid at leaves */
    const int32_t* orig_id;            /* [n_nodes] or NULL: the DBoW2 NodeId of every flat node (reported as the feature-vector key) when a loader
                                          had to renumber a vocabulary; NULL = the flat index IS the DBoW2 id */
} hs_vocab_tree;
/* Vocabulary files: what ORBVocabulary::ORBVocabulary(vocab_file) loads through DBoW2 (src/features/low_level/ORBVocabulary.cpp:14-29): a path ending
 * in ".txt" is DBoW2's text format (loadFromTextFile, ORBvoc.txt), anything else the binary format (loadFromBinaryFile, the output of
 * tools/bin_vocabulary.cc).  DBoW2 and the vocabulary file are external to the reference; the formats are restated from the published ORB-SLAM2
 * DBoW2 sources (hs_vocab.hip).  Host only, no device needed.  hs_vocab_get_tree's arrays stay valid until hs_vocab_destroy. */
typedef struct hs_vocab hs_vocab;
/* text of the last failed hs_vocab_load on the calling thread ("" after a success): the loaders print nothing */
const char* hs_vocab_last_error(void);
int  hs_vocab_load(const char* path, hs_vocab** out);
int  hs_vocab_from_tree(const hs_vocab_tree* tree, int k, hs_vocab** out);      /* deep copy of a caller-built flat tree (synthetic vocabularies) */
int  hs_vocab_save(const hs_vocab* v, const char* path);                        /* ".txt" -> text, else binary: tools/bin_vocabulary.cc's conversion */
void hs_vocab_destroy(hs_vocab* v);
int  hs_vocab_get_tree(const hs_vocab* v, hs_vocab_tree* out);
int  hs_vocab_info(const hs_vocab* v, int32_t* k, int32_t* L, int32_t* n_nodes, int32_t* n_words, int32_t* scoring, int32_t* weighting);

/* A vocabulary resident in HBM of h's device, prepared for feature vectors `levelsup` levels above the leaves (Frame::ComputeBoW uses 4,
 * src/core/Frame.cc:477).  Uploaded once; hs_vocab_dev_groups = number of distinct feature-vector nodes. */
typedef struct hs_vocab_dev hs_vocab_dev;
int  hs_vocab_upload(hs_orb* h, const hs_vocab_tree* tree, int levelsup, hs_vocab_dev** out);
void hs_vocab_dev_destroy(hs_vocab_dev* v);
int  hs_vocab_dev_groups(const hs_vocab_dev* v);
/* hs_bow_transform on descriptors that already live in HBM (the extractor's outputs): d_n (may be NULL) = device count clamped to n_max.
 * Asynchronous. */
int  hs_bow_transform_device(hs_orb* h, const hs_vocab_dev* v, const uint8_t* d_desc, const int32_t* d_n, int n_max,
                             int32_t* d_word, float* d_weight, int32_t* d_node, void* stream);
/* Cross-camera BoW matching over `world` gathered frame records (BASELINE config 5): for every peer p != rank the matching core of SearchByBoW /
 * _SearchByBoW_ (FeatureMatcher.cc:216-345: per shared vocabulary node, best / second-best Hamming of every side-1 feature over the node's side-2
 * features, `d < score_threshold && d < ratio * d2`, then RotationConsistencyBoW) between record `rank` (side 1) and record p (side 2), with the
 * vocabulary transform of all records done on the device.  d_match12 [world][cap] = side-2 index or -1 (row `rank`: all -1), d_n_matches [world].
 * No host synchronisation.  Asynchronous.  The matcher's scratch (feature groups, bucket lists) lives INSIDE `v`: one hs_vocab_dev serves one
 * caller stream at a time and is not thread-safe for this entry point (hs_bow_transform_device only reads `v` and may run concurrently); give
 * every handle / stream that matches concurrently its own hs_vocab_upload. */
int  hs_records_bow_match_device(hs_orb* h, hs_vocab_dev* v, const uint8_t* d_records, size_t record_stride, int world, int rank, int cap,
                                 float score_threshold, float second_best_ratio, int check_rotation,
                                 int32_t* d_match12, int32_t* d_n_matches, void* stream);

int  hs_bow_transform(hs_orb* h, const hs_vocab_tree* tree, const uint8_t* desc, int n, int levelsup,
                      int32_t* word_id, float* weight, int32_t* node_id);

/* brute-force Hamming 2-NN (cross-camera matching without a vocabulary): for each of nq query descriptors the first-minimum
 * train index, its distance and the second-smallest distance (-1 when absent). */
int  hs_hamming_knn2(hs_orb* h, const uint8_t* q, int nq, const uint8_t* t, int nt,
                     int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
/* same on device pointers, asynchronous */
int  hs_hamming_knn2_device(hs_orb* h, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                            int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist, void* stream);

/* ---- frame records: the fixed-size unit of the cross-camera exchange (BASELINE config 5; new — the reference has no
 * multi-camera exchange).  record = { int32 count; 12 bytes pad; hs_keypoint kps[cap]; pad to a 16-byte boundary; uint8 desc[cap][32] }: the
 * extractor's three outputs laid out in one buffer, so hs_orb_extract_batch_device writes a frame straight into the all-gather message (the
 * device entry points want the descriptor block 16-byte aligned — it is written with 16-byte vector stores — hence the padding for odd cap;
 * put the records themselves at 16-byte aligned addresses with a stride that is a multiple of 16: hs_record_bytes is). */
#define HS_RECORD_HEADER 16
size_t hs_record_bytes(int cap);
void   hs_record_offsets(int cap, size_t* off_count, size_t* off_kps, size_t* off_desc);
/* Cross-camera brute-force 2-NN over `world` gathered records (device memory, `record_stride` bytes apart): for every peer p != rank the
 * descriptors of record `rank` are matched against record p's.  The counts are read from the record headers ON THE DEVICE (clamped to
 * [0, cap]) — no host synchronisation between the all-gather and the matcher.  Outputs are [world][cap]; row `rank` and the entries beyond
 * the query count are left untouched.  One launch.  Asynchronous. */
int  hs_records_knn2_device(hs_orb* h, const uint8_t* d_records, size_t record_stride, int world, int rank, int cap,
                            int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist, void* stream);

/* ---- the exchange itself in C: an RCCL all-gather of the frame records (north_star: "RCCL all-gather over xGMI of per-frame keypoints /
 * descriptors"; hs_comm.hip).  hs_comm_get_unique_id on ONE rank; the caller carries the 128 bytes to the other ranks over its own channel
 * (file, socket, MPI, ...); every rank then calls hs_comm_create(handle of its GPU, id, world, rank), which blocks until all ranks arrived
 * (ncclCommInitRank).  One process per GPU.  hs_comm_allgather_records enqueues ncclAllGather of `record_bytes` bytes per rank on `stream`
 * (NULL = the handle's stream): d_gathered [world][record_bytes]; in place when d_record == d_gathered + rank * record_bytes.  With the
 * extraction before it and the matcher after it on the same stream a config-5 step needs no event and no host synchronisation.
 * librccl is loaded on first use (dlopen): HS_ERR_NO_DEVICE when it or a GPU is missing.  Asynchronous.
 * hs_comm_available(): non-collective probe (HS_OK / HS_ERR_NO_DEVICE, reason in hs_comm_unavailable_reason()) — ask it on every rank before
 * the first hs_comm_create, which blocks until all ranks arrive, and fall back together when any rank cannot.
 * Lifetime: a communicator BORROWS its handle (device, stream).  The handle is reference-counted (the owner + one per communicator):
 * hs_orb_destroy on a handle that still has communicators only drops the owner's reference and the last hs_comm_destroy frees it, so the two
 * destroy calls are safe in either order and from two threads; the handle must not be used for anything else after its hs_orb_destroy.
 * hs_orb_borrowers() = communicators alive on the handle.  librccl is taken from the directory of the HIP runtime the process runs on (a
 * process may hold two ROCm stacks: PyTorch ships its own libamdhip64 + librccl, and RCCL must match the runtime whose streams it is handed),
 * then by its usual names, and loaded RTLD_LOCAL so that a second copy in the process is left alone. */
#define HS_COMM_ID_BYTES 128
typedef struct hs_comm hs_comm;
int  hs_comm_available(void);
const char* hs_comm_unavailable_reason(void);
int  hs_orb_borrowers(const hs_orb* h);
int  hs_comm_get_unique_id(uint8_t* id /* [HS_COMM_ID_BYTES] */);
int  hs_comm_create(hs_orb* h, const uint8_t* id, int world, int rank, hs_comm** out);
void hs_comm_destroy(hs_comm* c);
int  hs_comm_rccl_ranks(const hs_comm* c);     /* ncclCommCount of the live communicator: what RCCL itself says, -1 = unknown */
int  hs_comm_rccl_rank(const hs_comm* c);      /* ncclCommUserRank, -1 = unknown */
int  hs_comm_rccl_version(void);               /* ncclGetVersion of the librccl in use (e.g. 22203), -1 = not loaded */
int  hs_comm_world(const hs_comm* c);
int  hs_comm_rank(const hs_comm* c);
const char* hs_comm_last_error(const hs_comm* c);
int  hs_comm_allgather_records(hs_comm* c, const void* d_record, void* d_gathered, size_t record_bytes, void* stream);

/* ---- per-stage device timing (HIP events recorded on the stream the kernels run on) ----
 * Stages: 0 pyramid, 1 FAST+NMS cells, 2 quadtree distribution, 3 blur+orient+rBRIEF, 4 stereo match, 5 stereo median.
 * begin: start collecting (events are recorded around every stage of every later call on this handle);
 * pause: stop recording without collecting (later calls run un-instrumented; what was recorded stays for `end`).  A recorded event
 *        drains the stream between two stages (≈5 µs each on MI355X, 28 µs per call): measure on some calls, not on all;
 * end:   synchronise, write the summed milliseconds per stage into ms[6] and the number of launches of each
 *        stage into launches[6] (pyramid counts one launch per call although it is nlevels-1 kernels), stop collecting. */
#define HS_NUM_STAGES 6
/* kernel launches that stage `stage` issues per call (the pyramid is several launches, the stereo match two) */
int  hs_orb_stage_launches(const hs_orb* h, int stage);
int  hs_orb_profile_begin(hs_orb* h);
int  hs_orb_profile_pause(hs_orb* h);
int  hs_orb_profile_end(hs_orb* h, double* ms, int32_t* launches);

/* Measurement utility: streams `bytes` from d_src to d_dst with `width` (4 or 16; 64 = four 16-byte vectors in flight per lane, non-temporal) bytes per lane — a kernel of KNOWN HBM traffic
 * in this library's own access widths, used to calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (tools/pmc_traffic.py). */
int  hs_debug_stream_copy(hs_orb* h, void* d_dst, const void* d_src, size_t bytes, int width, void* stream);

/* ---- stage taps for parity tests (host outputs; synchronous; valid after an extract call) ---- */
/* pyramid level `level` of image `image` of the last batch: tight w*h bytes; ORBExtractor::ComputePyramid :564-589 */
int  hs_orb_debug_level(hs_orb* h, int image, int level, uint8_t* out, size_t cap_bytes, int32_t* lw, int32_t* lh);
/* debug mode: the quadtree stage also gathers the FAST candidates into the dense per-level lists hs_orb_debug_candidates reads (the product path
 * works from the key histogram the FAST kernel leaves and never builds them).  Set before the extraction. */
int  hs_orb_set_debug(hs_orb* h, int on);
/* FAST candidates of that level before DistributeOctTree (unordered): (x,y,score) int32 triplets relative to
 * (16,16); ORBExtractor::ComputeKeyPointsOctTree :430-470 */
int  hs_orb_debug_candidates(hs_orb* h, int image, int level, int32_t* xys, int cap, int32_t* n);
/* keypoints kept by DistributeOctTree for that level, in list order: (x,y,score) level coords; :475-487 */
int  hs_orb_debug_selected(hs_orb* h, int image, int level, int32_t* xys, int cap, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif
