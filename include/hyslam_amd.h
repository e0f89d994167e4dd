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
float hs_orb_get_scale_factor(const hs_orb* h);
int  hs_orb_get_scale_tables(const hs_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                             int32_t* features_per_level);
/* smallest `cap` that can never overflow: nfeatures + per-level overshoot of DistributeOctTree */
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
/* Device-resident batch: image i starts at d_imgs + i*image_stride, rows `row_stride` bytes apart.
 * d_kps [batch][cap], d_desc [batch][cap][32], d_n [batch]; all device memory.  Asynchronous. */
int  hs_orb_extract_batch_device(hs_orb* h, const uint8_t* d_imgs, int batch, int w, int h_px,
                                 size_t row_stride, size_t image_stride,
                                 hs_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n, int cap, void* stream);

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

/* block until everything enqueued on the handle's own stream (or `stream`) has finished */
int  hs_orb_synchronize(hs_orb* h, void* stream);

/* ---- per-stage device timing (HIP events recorded on the stream the kernels run on) ----
 * Stages: 0 pyramid, 1 FAST+NMS cells, 2 quadtree distribution, 3 blur+orient+rBRIEF, 4 stereo match, 5 stereo median.
 * begin: start collecting (events are recorded around every stage of every later call on this handle);
 * end:   synchronise, write the summed milliseconds per stage into ms[6] and the number of launches of each
 *        stage into launches[6] (pyramid counts one launch per call although it is nlevels-1 kernels), stop collecting. */
#define HS_NUM_STAGES 6
int  hs_orb_profile_begin(hs_orb* h);
int  hs_orb_profile_end(hs_orb* h, double* ms, int32_t* launches);

/* ---- stage taps for parity tests (host outputs; synchronous; valid after an extract call) ---- */
/* pyramid level `level` of image `image` of the last batch: tight w*h bytes; ORBExtractor::ComputePyramid :564-589 */
int  hs_orb_debug_level(hs_orb* h, int image, int level, uint8_t* out, size_t cap_bytes, int32_t* lw, int32_t* lh);
/* FAST candidates of that level before DistributeOctTree (unordered): (x,y,score) int32 triplets relative to
 * (16,16); ORBExtractor::ComputeKeyPointsOctTree :430-470 */
int  hs_orb_debug_candidates(hs_orb* h, int image, int level, int32_t* xys, int cap, int32_t* n);
/* keypoints kept by DistributeOctTree for that level, in list order: (x,y,score) level coords; :475-487 */
int  hs_orb_debug_selected(hs_orb* h, int image, int level, int32_t* xys, int cap, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif
