/* hs_oracle.h — CPU ORACLE for the hySLAM ORB extract + Hamming match hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library.  The shipped path (hyslam_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * PARITY STATUS: **parity unpinned**.  The reference (bmhopkinson/hyslam) has no test, golden
 * vector or fixture for this path and cannot be compiled here: every
 * hot-path translation unit needs OpenCV 3.4 (cv::FAST, cv::resize, cv::GaussianBlur,
 * cv::fastAtan2, cvRound), which is neither vendored in /root/reference nor installed.
 * This file is therefore a dependency-free *restatement*:
 *   - hySLAM's own logic follows the reference line by line (citations below, all relative
 *     to /root/reference/);
 *   - the OpenCV 3.4 primitives follow OpenCV 3.4's published algorithms (modules/features2d/
 *     src/fast.cpp, fast_score.cpp; modules/imgproc/src/resize.cpp, smooth.cpp; modules/core/
 *     src/mathfuncs_core.simd.hpp), x86-64 baseline build (SSE2, no FMA, no IPP), see
 *     SURVEY.md Appendix A.
 * What pins it instead: the source-derived known-answer tables T1–T5 of SURVEY.md §8d
 * (tests/test_oracle_kat.py), an independent numpy restatement of every primitive
 * (tests/pyref.py), committed golden vectors produced by this oracle (tests/golden/), and
 * scikit-image — an implementation that shares nothing with this file or with OpenCV — for the
 * FAST corner set, score and non-max suppression (exact), the orientation (within fastAtan2's
 * error), the disc table and the rBRIEF pattern (exact), the steered-BRIEF bits (identical on
 * the test frame; double vs float rotation tolerated); PyTorch for the geometry of resize and
 * blur (tests/test_skimage_crosscheck.py).
 *
 * Documented deviations from the (non-deterministic / UB) reference behaviour:
 *   D1  DistributeOctTree sorts pair<int,ExtractorNode*> (src/features/ORBExtractor.cpp:321-324),
 *       i.e. ties between equally populated nodes are broken by heap address.  The oracle
 *       replaces the pointer with the node's creation sequence number (what a monotone
 *       allocator yields).
 *   D2  Stereomatcher indexes vRowIndices[yi] without bounds checks (Stereomatcher.cpp:53-63)
 *       and reads vDistIdx[size/2] of an empty vector (:143).  The oracle drops rows outside
 *       [0,nRows) and skips the median filter when there is no match.
 *   D3  GaussianBlur fixed-point taps are OpenCV-version dependent; default {18,34,49,55,49,34,18}
 *       (OpenCV 3.4.1..3.4.8 ufixedpoint16 rounding), overridable per call.
 */
#ifndef HS_ORACLE_H
#define HS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hso_keypoint {      /* the cv::KeyPoint fields the reference sets */
    float x, y, size, angle, response;
    int32_t octave;
} hso_keypoint;

typedef struct hso_orb_params {    /* HYSLAM::FeatureExtractorSettings (src/core/FeatureExtractorSettings.h:19-32) */
    int32_t nfeatures;             /* nFeatures      */
    float   scale_factor;          /* fScaleFactor   */
    int32_t nlevels;               /* nLevels        */
    int32_t cell_px;               /* N_CELLS — used as cell edge in pixels (ORBExtractor.cpp:409,425) */
    int32_t ini_th_fast;           /* init_threshold — ignored by the reference (ORBFinder.cpp:58-60) */
    int32_t min_th_fast;           /* min_threshold  — ignored likewise                             */
    int32_t fast_threshold;        /* effective FAST threshold; reference: always 20 (ORBFinder.h:92) */
    uint16_t blur_taps[7];         /* 8.8 fixed-point 7-tap Gaussian; all-zero => default           */
    uint16_t _pad;
} hso_orb_params;

typedef struct hso_stereo_params { /* what Stereomatcher reads from Camera / settings (Stereomatcher.cpp:7-24) */
    float   fx;                    /* camera.fx()                 */
    float   mbf;                   /* camera.mbf                  */
    int32_t n_rows;                /* (int)camera.mnMaxY          */
    float   th_high, th_low;       /* FeatureMatcherSettings      */
    float   size_ref;              /* orb_params.size_ref (31)    */
} hso_stereo_params;

/* ---- scalar helpers / tables (E0) ---- */
int   hso_cv_round_f(float v);
int   hso_cv_round_d(double v);
float hso_fast_atan2(float y, float x);
void  hso_default_params(hso_orb_params* p);
int   hso_scale_tables(const hso_orb_params* p, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* quotas);
void  hso_pyramid_size(const hso_orb_params* p, int w, int h, int level, int32_t* lw, int32_t* lh);
void  hso_cell_grid(const hso_orb_params* p, int lw, int lh, int32_t* ncols, int32_t* nrows, int32_t* wcell, int32_t* hcell);
void  hso_umax(int32_t* umax16);
const int32_t* hso_pattern(void);

/* ---- OpenCV primitives (Appendix A) ---- */
void hso_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
/* ImageProcessing::PreProcessImg (src/main/ImageProcessing.cpp:118-138): cv::resize by the camera's scale on the 1 / 3 / 4-channel frame, then
 * cvtColor to grey (rgb: 1 = RGB(A) order, 0 = BGR(A)).  Output size = hso_preprocess_size.  Returns 0, or -1 for an empty result / bad channels. */
void hso_preprocess_size(int w, int h, float scale, int32_t* ow, int32_t* oh);
int  hso_preprocess(const uint8_t* src, int w, int h, int sstride, int channels, int rgb, float scale, uint8_t* dst, int dstride);
/* out: triplets (x, y, score) int32; returns number found (may exceed cap; only cap written) */
int  hso_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax, int32_t* out_xys, int cap);
void hso_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, const uint16_t* taps7);

/* ---- ORBFinder pieces ---- */
float hso_ic_angle(const uint8_t* img, int stride, float x, float y);
void  hso_orb_descriptor(const uint8_t* img, int stride, float x, float y, float angle, uint8_t* desc32);
int   hso_hamming256(const uint8_t* a, const uint8_t* b);

/* ---- ORBExtractor pieces ---- */
/* candidates: (x,y,response) float triplets in vToDistributeKeys order; out_idx: index of the kept
 * candidate per surviving node in final list order.  Returns count. */
int hso_distribute_octtree(const float* cand_xyr, int n, int minX, int maxX, int minY, int maxY, int N,
                           int32_t* out_idx, int cap);
/* per-level candidate list as produced by ComputeKeyPointsOctTree before distribution:
 * (x,y,response) relative to (minBorderX,minBorderY)=(16,16).  Returns count (may exceed cap). */
int hso_level_candidates(const hso_orb_params* p, const uint8_t* level_img, int lw, int lh, int stride,
                         float* out_xyr, int cap);

typedef struct hso_extract_debug {  /* optional taps into the stages, caller-allocated, may be NULL */
    uint8_t** pyramid;              /* [nlevels] buffers of lw*lh (tight stride) or NULL */
    uint8_t** blurred;              /* [nlevels] idem */
    int32_t*  n_candidates;         /* [nlevels] */
    int32_t*  n_selected;           /* [nlevels] */
    float**   candidates;           /* [nlevels] (x,y,resp) triplets, cap cand_cap each, or NULL */
    int32_t   cand_cap;
} hso_extract_debug;

/* ORBExtractor::operator() (ORBExtractor.cpp:496-562).  Returns number of keypoints (<= cap), or <0 on error. */
int hso_orb_extract(const hso_orb_params* p, const uint8_t* img, int w, int h, int stride,
                    hso_keypoint* kps, uint8_t* desc, int cap, hso_extract_debug* dbg);

/* ---- Stereomatcher::computeStereoMatches (Stereomatcher.cpp:36-156) ---- */
int hso_stereo_match(const hso_keypoint* kpsL, const uint8_t* descL, int nL,
                     const hso_keypoint* kpsR, const uint8_t* descR, int nR,
                     const hso_stereo_params* sp, float* uRight, float* depth, int32_t* best_idx, int32_t* best_dist);

/* ---- CPU baseline harness: ImageProcessing::ProcessStereoImage structure (ImageProcessing.cpp:69-116):
 * left image on a spawned std::thread, right on the caller, then the stereo matcher.  Returns nL. */
int hso_stereo_frontend(const hso_orb_params* p, const hso_stereo_params* sp,
                        const uint8_t* imgL, const uint8_t* imgR, int w, int h, int stride,
                        hso_keypoint* kpsL, uint8_t* descL, int32_t* nL,
                        hso_keypoint* kpsR, uint8_t* descR, int32_t* nR, int cap,
                        float* uRight, float* depth);

#ifdef __cplusplus
}
#endif
#endif

/* ======================================================================================================
 * Matchers on flat arrays (restated in oracle/hs_oracle_match.cpp).  Same status: test infrastructure,
 * parity unpinned.  The structs are the gather of what the reference reads from Frame / KeyFrame / Camera /
 * FeatureViews / MapPoint / LandMarkMatches; pointer-ordered containers (std::map<MapPoint*,...>) are replaced
 * by landmark ARRAY ORDER (documented deviation D6: the adaptor passes landmarks sorted by address to
 * reproduce the reference's iteration order).
 * ====================================================================================================== */
#ifndef HS_ORACLE_MATCH_DECLS
#define HS_ORACLE_MATCH_DECLS
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hso_frame_view {        /* Frame.cc:45-72,137-180,416-469; Camera.cpp:116-153 */
    float Rcw[9], tcw[3], Ow[3];       /* mRcw (row-major), mtcw, mOw */
    float fx, fy, cx, cy, mbf;
    int32_t sensor;                    /* Camera::sensor: 0 mono, 1 stereo, 2 RGBD */
    float min_x, max_x, min_y, max_y;  /* mnMinX.. */
    float size_ref;                    /* FeatureExtractorSettings::size_ref of the views (31) */
    int32_t n;                         /* number of keypoints */
    const hso_keypoint* kps;
    const uint8_t* desc;               /* n x 32 */
    const float* uR;                   /* n, <0 = no stereo correspondence */
    const int32_t* kp_lm_obs;          /* n: -1 = keypoint has no landmark, else Observations() of the associated landmark */
} hso_frame_view;

typedef struct hso_landmark {          /* MapPoint.h:54-169 fields the matchers read */
    float pos[3];                      /* GetWorldPos() */
    float size;                        /* getSize(), world units */
    float min_dist, max_dist;          /* mfMinDistance, mfMaxDistance (before the 0.8 / 1.2 factors) */
    float normal[3];                   /* GetNormal() */
    int32_t assoc_kp;                  /* Frame::hasAssociation(lm) in THIS frame, -1 if none */
    float prev_angle;                  /* angle of the keypoint it is associated with in the previous frame (rotation check) */
    int32_t skip;                      /* 1 = nullptr entry in the reference's vector */
    uint8_t desc[32];                  /* GetDescriptor() */
} hso_landmark;

typedef struct hso_proj_params {
    float th;                          /* search radius factor */
    float score_threshold;             /* BestScoreCriterion: TH_HIGH or ORBdist */
    float second_best_ratio;           /* mfNNratio or 1.0 */
    float frac_smaller, frac_larger;   /* FeatureSizeCriterion(0.5, 1.5) */
    int32_t use_distance;              /* DistanceCriterion in the landmark criteria */
    int32_t use_stereo;                /* StereoConsistencyCriterion(th) */
    int32_t check_rotation;            /* RotationConsistencyCriterion (needs prev_angle) */
    int32_t use_prev_matched;          /* PreviouslyMatchedCriterion (all Frame variants; not Fuse) */
    int32_t use_viewing_angle;         /* ViewingAngleCriterion(max_view_angle) — Fuse, FeatureMatcher.cc:469 */
    float   max_view_angle;            /* radians (1.047) */
    int32_t use_reprojection;          /* ProjectionViewCriterion(reproj_threshold) — Fuse, :473 */
    float   reproj_threshold;          /* 5.99 */
    float   sigma_ref;                 /* FeatureExtractorSettings::sigma_ref (1.0) for determineSigma2 */
    int32_t first_wins;                /* Fuse: fuse_matches.insert(idx, lm) keeps the FIRST landmark that matched a keypoint (:515) */
    int32_t dist_is_invariance_range;  /* 1: min_dist / max_dist already carry the 0.8 / 1.2 factors (GetMin/MaxDistanceInvariance()) */
} hso_proj_params;

/* FeatureMatcher::_SearchByProjection_ (FeatureMatcher.cc:57-121) with the criteria lists of the three Frame variants
 * (:123-143, :145-176, :180-212).  match_idx[L] = keypoint index or -1, match_dist[L].  Returns matches.size(). */
int hso_search_by_projection(const hso_frame_view* F, const hso_landmark* lms, int L, const hso_proj_params* pp,
                             int32_t* match_idx, float* match_dist);
/* FeatureMatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) — loop detection, legacy (FeatureMatcher.cc:628-737).  KF = the keyframe
 * (its own pose: landMarkSizePixels projects with it), Scw = 4x4 row-major Sim3, lms in vpPoints order with min_dist / max_dist = the
 * invariance range and skip = isBad || already found; kp_matched[n] = vpMatched[idx] != NULL, in/out (SEQUENTIAL: a keypoint taken by an
 * earlier landmark is invisible to the later ones).  match_idx[L] = keypoint taken or -1.  Returns nmatches. */
int hso_search_by_projection_sim3(const hso_frame_view* KF, const float* Scw, const hso_landmark* lms, int L, int th, float th_low,
                                  uint8_t* kp_matched, int32_t* match_idx);
/* FeatureMatcher::SearchBySim3 (FeatureMatcher.cc:739-934): lms1[n1] / lms2[n2] = the landmark of each keypoint of KF1 / KF2 (skip = none, bad
 * or already matched; assoc_kp = its view in the OTHER keyframe, for landMarkSizePixels); match12[n1] = agreed KF2 index or -1.  Returns nFound. */
int hso_search_by_sim3(const hso_frame_view* KF1, const hso_landmark* lms1, const hso_frame_view* KF2, const hso_landmark* lms2,
                       float s12, const float* R12, const float* t12, float th, float th_high, int32_t* match12);
/* Frame::AssignFeaturesToGrid / PosInGrid (Frame.cc:137-153,459-469): cell_xy[2*i] = column or -1 (outside), cell_xy[2*i+1] = row */
void hso_frame_grid(const hso_frame_view* F, int32_t* cell_xy);
/* the inner loops of SearchByBoW / _SearchByBoW_ (FeatureMatcher.cc:216-345) + BestMatchBoWCriterion (MatchCriteria.cpp:601-635)
 * + RotationConsistencyBoW (:679-726).  Feature vectors are CSR: node ids ascending, node_ptr[n_nodes+1], idx[].
 * keep1[n1] = 1 for indices of side 1 that pass the index criteria.  match12[n1] = index in side 2 or -1.  Returns #matches. */
int hso_search_by_bow(const hso_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                      const hso_keypoint* kps2, const uint8_t* desc2, int n2, const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                      const uint8_t* keep1, float score_threshold, float second_best_ratio, int check_rotation, int32_t* match12);
/* the same with the index criteria applied to BOTH sides (keep1, keep2: _SearchByBoW_, FeatureMatcher.cc:306-309) and, when F12 != NULL,
 * EpipolarConsistencyBoWCriterion (MatchCriteria.cpp:641-676) ahead of the best-match criterion: SearchForTriangulation (:373-402) */
int hso_search_by_bow_ex(const hso_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                         const hso_keypoint* kps2, const uint8_t* desc2, int n2, const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                         const uint8_t* keep1, const uint8_t* keep2, const float* F12, float size_ref, float sigma_ref,
                         float score_threshold, float second_best_ratio, int check_rotation, int32_t* match12);
/* the legacy FeatureMatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (FeatureMatcher.cc:938-1077): like the above, but a side-2 feature can be matched
 * only once (sequential inside a node) and the rotation check takes angle1 - angle2 */
int hso_search_by_bow_legacy(const hso_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                             const hso_keypoint* kps2, const uint8_t* desc2, int n2, const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                             const uint8_t* keep1, const uint8_t* keep2, float th_low, float nnratio, int check_orientation, int32_t* match12);
/* DBoW2::TemplatedVocabulary<FORB>::transform(features, bow, fv, levelsup) as called by Frame::ComputeBoW (Frame.cc:472-479) through
 * ORBVocabulary::transform (ORBVocabulary.cpp:31-42).  DBoW2 is NOT in the reference tree; restated from its published
 * algorithm: descend from the root, at each level take the child with the smallest Hamming distance (first minimum wins);
 * the leaf gives word id + weight, the node passed at level (L - levelsup) keys the feature vector.  Flat tree: node 0 = root, children of a
 * node are contiguous [child_begin, child_begin+child_count), child_count == 0 marks a leaf. */
typedef struct hso_vocab_tree {
    int32_t n_nodes, levels;
    const int32_t* child_begin; const int32_t* child_count;
    const uint8_t* desc;               /* n_nodes x 32 (root's is unused) */
    const int32_t* word_id;            /* leaves */
    const float* weight;               /* leaves */
    const int32_t* orig_id;            /* NULL or the DBoW2 NodeId of every flat node (renumbered vocabularies) */
} hso_vocab_tree;
void hso_bow_transform(const hso_vocab_tree* T, const uint8_t* desc, int n, int levelsup, int32_t* word_id, float* weight, int32_t* node_id);

/* FeatureMatcher::SearchForInitialization (FeatureMatcher.cc:404-462) with MonoInitScoreExceedsPrevious / MonoInitBestScore
 * (MatchCriteria.cpp:486-549): sequential over frame-1 keypoints (a later keypoint can steal a frame-2 keypoint only with a strictly
 * smaller distance).  F2 = frame 2 (grid bounds, keypoints, descriptors); prev_matched_xy[n1][2] in/out; matches12[n1] = frame-2 index or -1. */
int hso_search_for_initialization(const hso_keypoint* kps1, const uint8_t* desc1, int n1, const hso_frame_view* F2,
                                  float* prev_matched_xy, int window, float th_low, float nnratio, int32_t* matches12);

/* brute-force Hamming 2-NN of every query against every train descriptor: best index, best and second-best distance
 * (first minimum wins, like every best/second-best loop of the reference, e.g. MatchCriteria.cpp:248-280) */
void hso_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
/* RotationConsistency (MatchCriteria.cpp:684-726) + ComputeThreeMaxima (:727-767): keep[i] = 1 if pair i survives */
void hso_rotation_consistency(const float* angle_a, const float* angle_b, int n, uint8_t* keep);

#ifdef __cplusplus
}
#endif
#endif
