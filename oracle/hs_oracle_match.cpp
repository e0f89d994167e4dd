/* hs_oracle_match.cpp — CPU ORACLE, matcher half (test infrastructure; parity unpinned, see hs_oracle.h).
 *
 * Restates on flat arrays:
 *   FeatureMatcher::_SearchByProjection_ + wrappers    /root/reference/src/features/FeatureMatcher.cc:57-212
 *   the criteria classes                               /root/reference/src/features/MatchCriteria.cpp
 *   Frame grid / projection / landMarkSizePixels       /root/reference/src/core/Frame.cc:137-180,296-317,416-469
 *   Camera::Project                                    /root/reference/src/core/Camera.cpp:116-153
 *   MapPoint distance invariance                       /root/reference/src/core/MapPoint.cc:139-149
 *   SearchByBoW / _SearchByBoW_ inner loops            /root/reference/src/features/FeatureMatcher.cc:216-345
 * cv::Mat products (mRcw*P+mtcw, K*Pch) are OpenCV gemm calls: float inputs, double accumulation, one rounding to float
 * (GEMMSingleMul<float,double>); restated as such.
 */
#include "hs_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

namespace {

const int FRAME_GRID_COLS = 64, FRAME_GRID_ROWS = 48;   // Frame.h

float ORBDistance(const uint8_t* D1, const uint8_t* D2)   // DescriptorDistance.cpp:9-25
{
    int32_t pa[8], pb[8];
    std::memcpy(pa, D1, 32); std::memcpy(pb, D2, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return static_cast<float>(dist);
}

struct Frame {
    const hso_frame_view& V;
    float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];

    explicit Frame(const hso_frame_view& v) : V(v) {
        mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (V.max_x - V.min_x);     // Frame.cc:66-67
        mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (V.max_y - V.min_y);
        for (int i = 0; i < V.n; i++) {                                                          // AssignFeaturesToGrid :137-153
            int x, y;
            if (PosInGrid(V.kps[i], x, y)) mGrid[x][y].push_back(i);
        }
    }
    bool PosInGrid(const hso_keypoint& kp, int& posX, int& posY) const {                         // :459-469
        posX = (int)std::round((kp.x - V.min_x) * mfGridElementWidthInv);
        posY = (int)std::round((kp.y - V.min_y) * mfGridElementHeightInv);
        if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) return false;
        return true;
    }
    // Frame::ProjectLandMark(cv::Mat P, uv_ur) :170-180 + Camera::Project (Camera.cpp:116-153)
    bool ProjectLandMark(const float P[3], float uv[3]) const {
        float Pc[3];
        for (int i = 0; i < 3; i++) {   // Pc = mRcw*P + mtcw : one gemm, double accumulation
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)V.Rcw[3 * i + k] * (double)P[k];
            Pc[i] = (float)(s + (double)V.tcw[i]);
        }
        return CameraProject(Pc, uv);
    }
    // Camera::Project(Pc, uv) (Camera.cpp:116-153) on a point already in camera coordinates
    bool CameraProject(const float Pc[3], float uv[3]) const {
        float PcZ = Pc[2];
        float invz = 1.0f / PcZ;
        float Pch[3] = { Pc[0] / PcZ, Pc[1] / PcZ, Pc[2] / PcZ };
        // uv = K*Pch, K = [fx 0 cx; 0 fy cy; 0 0 1]
        float u = (float)((double)V.fx * (double)Pch[0] + 0.0 * (double)Pch[1] + (double)V.cx * (double)Pch[2]);
        float v = (float)(0.0 * (double)Pch[0] + (double)V.fy * (double)Pch[1] + (double)V.cy * (double)Pch[2]);
        uv[0] = u; uv[1] = v;
        if (V.sensor == 1) uv[2] = u - V.mbf * invz; else uv[2] = -1.0f;
        bool valid = false;
        if (PcZ > 0.0f) if (u >= V.min_x && u <= V.max_x) if (v >= V.min_y && v <= V.max_y) valid = true;
        return valid;
    }
    // Frame::landMarkSizePixels :296-317
    float landMarkSizePixels(const hso_landmark& lm) const {
        if (lm.assoc_kp >= 0) return V.kps[lm.assoc_kp].size;
        float left[3] = { lm.pos[0] - lm.size / 2, lm.pos[1], lm.pos[2] };
        float right[3] = { lm.pos[0] + lm.size / 2, lm.pos[1], lm.pos[2] };
        float uvl[3], uvr[3];
        ProjectLandMark(left, uvl);
        ProjectLandMark(right, uvr);
        return uvr[0] - uvl[0];
    }
    // Frame::GetFeaturesInAreaNEW :416-457
    std::vector<size_t> GetFeaturesInAreaNEW(float x, float y, float r) const {
        std::vector<size_t> vIndices;
        const int nMinCellX = std::max(0, (int)std::floor((x - V.min_x - r) * mfGridElementWidthInv));
        if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
        const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)std::ceil((x - V.min_x + r) * mfGridElementWidthInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = std::max(0, (int)std::floor((y - V.min_y - r) * mfGridElementHeightInv));
        if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
        const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)std::ceil((y - V.min_y + r) * mfGridElementHeightInv));
        if (nMaxCellY < 0) return vIndices;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t>& vCell = mGrid[ix][iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const hso_keypoint& kpUn = V.kps[vCell[j]];
                    const float distx = kpUn.x - x, disty = kpUn.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    }
};

struct SingleMatchData { int idx; float distance; float distance_2ndbest; };

// ComputeThreeMaxima, MatchCriteria.cpp:727-767
void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// RotationConsistency, MatchCriteria.cpp:684-726; matches: idx_curr -> idx_prev; angle lookups through callbacks
typedef std::map<size_t, size_t> MatchesIdx;
template <class AC, class AP>
MatchesIdx RotationConsistency(const MatchesIdx& current_matches, AC angle_curr, AP angle_prev)
{
    const int HISTO_LENGTH = 30;
    MatchesIdx matches_passed = current_matches;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (auto it = current_matches.begin(); it != current_matches.end(); ++it) {
        size_t idx_curr = it->first, idx_prev = it->second;
        float rot = angle_prev(idx_prev) - angle_curr(idx_curr);
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back((int)idx_curr);
    }
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
        if (i != ind1 && i != ind2 && i != ind3)
            for (size_t j = 0; j < rotHist[i].size(); j++) matches_passed.erase(rotHist[i][j]);
    return matches_passed;
}

} // namespace

extern "C" {

void hso_frame_grid(const hso_frame_view* F, int32_t* cell_xy)
{
    Frame fr(*F);
    for (int i = 0; i < F->n; i++) {
        int x, y;
        if (fr.PosInGrid(F->kps[i], x, y)) { cell_xy[2 * i] = x; cell_xy[2 * i + 1] = y; }
        else { cell_xy[2 * i] = -1; cell_xy[2 * i + 1] = -1; }
    }
}

int hso_search_by_projection(const hso_frame_view* F, const hso_landmark* lms, int L, const hso_proj_params* pp,
                             int32_t* match_idx, float* match_dist)
{
    Frame frame(*F);
    for (int i = 0; i < L; i++) { match_idx[i] = -1; match_dist[i] = -1.f; }
    // ---- landmark criteria (FeatureMatcher.cc:71-74): ProjectionCriterion, then DistanceCriterion
    std::vector<int> cand_lms;
    for (int i = 0; i < L; i++) {
        if (lms[i].skip) continue;
        float uv[3];
        if (frame.ProjectLandMark(lms[i].pos, uv)) cand_lms.push_back(i);            // MatchCriteria.cpp:13-28
    }
    if (pp->use_distance) {                                                          // DistanceCriterionCore :58-78
        std::vector<int> passed;
        for (int i : cand_lms) {
            const hso_landmark& lm = lms[i];
            float PO[3] = { lm.pos[0] - F->Ow[0], lm.pos[1] - F->Ow[1], lm.pos[2] - F->Ow[2] };
            const float lm_dist = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);   // cv::norm: double accumulation
            const float maxDistance = pp->dist_is_invariance_range ? lm.max_dist : 1.2f * lm.max_dist;                       // MapPoint.cc:139-149
            const float minDistance = pp->dist_is_invariance_range ? lm.min_dist : 0.8f * lm.min_dist;
            if (lm_dist < minDistance || lm_dist > maxDistance) continue;
            passed.push_back(i);
        }
        cand_lms.swap(passed);
    }
    if (pp->use_viewing_angle) {                                                     // ViewingAngleCriterionCore :94-110
        std::vector<int> passed;
        const float thr = std::cos(pp->max_view_angle);                              // <math.h> in C++: the float overload
        for (int i : cand_lms) {
            const hso_landmark& lm = lms[i];
            float PO[3] = { lm.pos[0] - F->Ow[0], lm.pos[1] - F->Ow[1], lm.pos[2] - F->Ow[2] };
            const float distance = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
            const double alpha = 1.0 / (double)distance;                             // PO = PO/distance: cv::Mat scaled by 1/s
            double dot = 0;
            for (int k = 0; k < 3; k++) dot += (double)(float)((double)PO[k] * alpha) * (double)lm.normal[k];   // Mat::dot accumulates in double
            if (dot > thr) passed.push_back(i);
        }
        cand_lms.swap(passed);
    }
    // ---- per landmark: candidate views and view criteria (:77-103)
    std::map<int, SingleMatchData> matches;   // key: landmark array index (stands in for MapPoint*; D6)
    for (int li : cand_lms) {
        const hso_landmark& lm = lms[li];
        float uv[3];
        frame.ProjectLandMark(lm.pos, uv);
        const float u = uv[0], v = uv[1], ur = uv[2];
        const float sizePx = frame.landMarkSizePixels(lm);
        const float radius = pp->th * sizePx / F->size_ref;
        std::vector<size_t> cand = frame.GetFeaturesInAreaNEW(u, v, radius);
        if (pp->use_prev_matched) {   // PreviouslyMatchedCriterionCore :124-144
            std::vector<size_t> passed;
            for (size_t idx : cand) { bool save = true; if (F->kp_lm_obs && F->kp_lm_obs[idx] >= 0 && F->kp_lm_obs[idx] > 0) save = false; if (save) passed.push_back(idx); }
            cand.swap(passed);
        }
        {   // FeatureSizeCriterionCore :350-360
            std::vector<size_t> passed;
            for (size_t idx : cand) { const hso_keypoint& kp = F->kps[idx]; if (kp.size > pp->frac_smaller * sizePx && kp.size < pp->frac_larger * sizePx) passed.push_back(idx); }
            cand.swap(passed);
        }
        if (pp->use_stereo && F->sensor != 0) {   // StereoConsistencyCriterion :149-177
            std::vector<size_t> passed;
            const float radius2 = pp->th * sizePx / F->size_ref;
            for (size_t idx : cand) { float ur_view = F->uR[idx]; const float er = std::fabs(ur - ur_view); if (er < radius2 && ur_view > 0) passed.push_back(idx); }
            cand.swap(passed);
        }
        if (pp->use_reprojection) {   // ProjectionViewCriterion :285-307 with KeyFrame::ReprojectionError (KeyFrame.cc:548-573)
            std::vector<size_t> passed;
            for (size_t idx : cand) {
                float reproj_err;
                {
                    const hso_keypoint& kpt = F->kps[idx];
                    float errX = u - kpt.x, errY = v - kpt.y, errXr = 0.000f;
                    float ur_view = F->uR ? F->uR[idx] : -1.f;
                    if (ur_view >= 0.0) errXr = ur - ur_view;
                    reproj_err = errX * errX + errY * errY + errXr * errXr;        // the landmark projects (it passed ProjectionCriterion)
                }
                float scale_factor = F->kps[idx].size / F->size_ref;                // determineSigma2, FeatureExtractorSettings.cpp:5-8
                float sigma_size_corrected = pp->sigma_ref * (scale_factor * scale_factor);
                float stereo_factor = 1.00;
                if (F->uR && F->uR[idx] > 0) stereo_factor = 1.30;
                if ((reproj_err / sigma_size_corrected) < stereo_factor * pp->reproj_threshold) passed.push_back(idx);
            }
            cand.swap(passed);
        }
        // BestScoreCriterionCore :248-280 + accept rule :214-230
        float bestDist = std::numeric_limits<float>::max(), bestDist2 = std::numeric_limits<float>::max();
        int bestIdx = -1;
        for (size_t idx : cand) {
            const float d = ORBDistance(lm.desc, F->desc + idx * 32);
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx = (int)idx; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist <= pp->score_threshold) {
            if (bestDist > pp->second_best_ratio * bestDist2) { }
            else matches[li] = SingleMatchData{ bestIdx, bestDist, bestDist2 };
        }
    }
    // ---- global criteria: RotationConsistencyCriterion :363-401
    if (pp->check_rotation) {
        MatchesIdx current_matches_idx; std::map<size_t, int> inverse;
        for (auto& m : matches) { current_matches_idx[m.second.idx] = (size_t)m.first; inverse[m.second.idx] = m.first; }   // later landmark overwrites
        MatchesIdx passed = RotationConsistency(current_matches_idx,
                                                [&](size_t idx_curr) { return F->kps[idx_curr].angle; },
                                                [&](size_t li) { return lms[li].prev_angle; });
        std::map<int, SingleMatchData> alt;
        for (auto& p : passed) { int li = inverse[p.first]; alt[li] = matches[li]; }
        matches.swap(alt);
    }
    if (pp->first_wins) {   // Fuse: std::map<idx, MapPoint*>::insert keeps the first landmark (in candidate order) per keypoint
        std::map<int, int> owner;
        for (auto& m : matches) owner.insert(std::make_pair(m.second.idx, m.first));
        std::map<int, SingleMatchData> alt;
        for (auto& o : owner) alt[o.second] = matches[o.second];
        matches.swap(alt);
    }
    for (auto& m : matches) { match_idx[m.first] = m.second.idx; match_dist[m.first] = m.second.distance; }
    return (int)matches.size();
}

// EpipolarConsistencyBoWCriterion::CheckDistEpipolarLine, MatchCriteria.cpp:658-676
static bool CheckDistEpipolarLine(const hso_keypoint& kp1, const hso_keypoint& kp2, const float* F12, float sigma2)
{
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2;
}

int hso_search_by_bow(const hso_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                      const hso_keypoint* kps2, const uint8_t* desc2, int n2, const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                      const uint8_t* keep1, float score_threshold, float second_best_ratio, int check_rotation, int32_t* match12)
{
    return hso_search_by_bow_ex(kps1, desc1, n1, node_id1, node_ptr1, idx1, nn1, kps2, desc2, n2, node_id2, node_ptr2, idx2, nn2,
                                keep1, nullptr, nullptr, 31.f, 1.f, score_threshold, second_best_ratio, check_rotation, match12);
}

int hso_search_by_bow_ex(const hso_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                         const hso_keypoint* kps2, const uint8_t* desc2, int n2, const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                         const uint8_t* keep1, const uint8_t* keep2, const float* F12, float size_ref, float sigma_ref,
                         float score_threshold, float second_best_ratio, int check_rotation, int32_t* match12)
{
    (void)n2;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    MatchesIdx matches_internal;
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {                                   // merge-walk of the two feature vectors, FeatureMatcher.cc:230-265
        if (node_id1[a] == node_id2[b]) {
            for (int p = node_ptr1[a]; p < node_ptr1[a + 1]; p++) {
                const int i1 = idx1[p];
                if (keep1 && !keep1[i1]) continue;                  // PreviouslyMatchedIndexCriterion, MatchCriteria.cpp:555-574
                float bestDist1 = std::numeric_limits<float>::max(), bestDist2 = std::numeric_limits<float>::max();
                int bestIdx2 = -1;
                for (int q = node_ptr2[b]; q < node_ptr2[b + 1]; q++) {   // BestMatchBoWCriterion :601-635
                    const int i2 = idx2[q];
                    if (keep2 && !keep2[i2]) continue;             // index criteria on side 2 (:308)
                    if (F12) {                                      // EpipolarConsistencyBoWCriterion::apply :645-657
                        float scale_factor = kps2[i2].size / size_ref;
                        float sigma_size_corrected = sigma_ref * (scale_factor * scale_factor);
                        if (!CheckDistEpipolarLine(kps1[i1], kps2[i2], F12, sigma_size_corrected)) continue;
                    }
                    const float dist = ORBDistance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = i2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < score_threshold)
                    if (static_cast<float>(bestDist1) < second_best_ratio * static_cast<float>(bestDist2))
                        matches_internal.insert(std::make_pair((size_t)i1, (size_t)bestIdx2));
            }
            a++; b++;
        } else if (node_id1[a] < node_id2[b]) { while (a < nn1 && node_id1[a] < node_id2[b]) a++; }   // lower_bound
        else { while (b < nn2 && node_id2[b] < node_id1[a]) b++; }
    }
    if (check_rotation)   // RotationConsistencyBoW::apply(matches, views1, views2): "curr" = side 1, "prev" = side 2
        matches_internal = RotationConsistency(matches_internal, [&](size_t i) { return kps1[i].angle; }, [&](size_t i) { return kps2[i].angle; });
    for (auto& m : matches_internal) match12[m.first] = (int32_t)m.second;
    return (int)matches_internal.size();
}

/* FeatureMatcher::SearchByBoW(pKF1, pKF2, vpMatches12) — the legacy key-frame / key-frame matcher (FeatureMatcher.cc:938-1077; no call site in
 * hySLAM, "aim to replace this with SearchByBoW2").  Differences from _SearchByBoW_: a side-2 feature that an earlier side-1 feature matched is
 * out of the game (vbMatched2, :959,999,1027: sequential inside a node; the nodes' index sets are disjoint), and the rotation histogram takes
 * angle1 - angle2 (:1031), the opposite sign of RotationConsistencyBoW.  keep1 / keep2 = the view has a landmark that is not bad (:985-990,1001-1005). */
int hso_search_by_bow_legacy(const hso_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                             const hso_keypoint* kps2, const uint8_t* desc2, int n2, const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                             const uint8_t* keep1, const uint8_t* keep2, float th_low, float nnratio, int check_orientation, int32_t* match12)
{
    const int HISTO_LENGTH = 30;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<bool> vbMatched2(std::max(n2, 1), false);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {
        if (node_id1[a] == node_id2[b]) {
            for (int p = node_ptr1[a]; p < node_ptr1[a + 1]; p++) {
                const int i1 = idx1[p];
                if (keep1 && !keep1[i1]) continue;                                  // !pMP1 || pMP1->isBad()
                float bestDist1 = std::numeric_limits<float>::max(), bestDist2 = std::numeric_limits<float>::max();
                int bestIdx2 = -1;
                for (int q = node_ptr2[b]; q < node_ptr2[b + 1]; q++) {
                    const int i2 = idx2[q];
                    if (vbMatched2[i2] || (keep2 && !keep2[i2])) continue;          // vbMatched2[idx2] || !pMP2 || pMP2->isBad()
                    const float dist = ORBDistance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = i2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < th_low && static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    match12[i1] = bestIdx2;
                    vbMatched2[bestIdx2] = true;
                    if (check_orientation) {
                        float rot = kps1[i1].angle - kps2[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(i1);
                    }
                    nmatches++;
                }
            }
            a++; b++;
        } else if (node_id1[a] < node_id2[b]) { while (a < nn1 && node_id1[a] < node_id2[b]) a++; }
        else { while (b < nn2 && node_id2[b] < node_id1[a]) b++; }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int hso_search_for_initialization(const hso_keypoint* kps1, const uint8_t* desc1, int n1, const hso_frame_view* F2v,
                                  float* prev_matched_xy, int window, float th_low, float nnratio, int32_t* matches12)
{
    Frame F2(*F2v);
    MatchesIdx matches;                        // idx in frame 2 -> idx in frame 1
    std::map<size_t, float> distances;         // MonoCriteriaData::distances
    for (int i1 = 0; i1 < n1; i1++) {
        std::vector<size_t> cand = F2.GetFeaturesInAreaNEW(prev_matched_xy[2 * i1], prev_matched_xy[2 * i1 + 1], (float)window);   // no level check (:415)
        if (cand.empty()) continue;
        const uint8_t* d1 = desc1 + (size_t)i1 * 32;
        {   // MonoInitScoreExceedsPrevious :525-549
            std::vector<size_t> passed;
            for (size_t i2 : cand) {
                auto it = distances.find(i2);
                float dist_prev = it != distances.end() ? it->second : -1;
                if (dist_prev < 0) passed.push_back(i2);
                else { int dist = (int)ORBDistance(d1, F2v->desc + i2 * 32); if (dist < dist_prev) passed.push_back(i2); }
            }
            cand.swap(passed);
        }
        // MonoInitBestScore :486-523
        float bestDist = std::numeric_limits<float>::max(), bestDist2 = std::numeric_limits<float>::max();
        int bestIdx = -1;
        for (size_t i2 : cand) {
            float dist = ORBDistance(d1, F2v->desc + i2 * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = (int)i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= th_low && bestDist < (float)bestDist2 * nnratio) {
            matches[bestIdx] = i1;                                            // overwrites any old match (:433)
            distances[bestIdx] = ORBDistance(d1, F2v->desc + (size_t)bestIdx * 32);
        }
    }
    matches = RotationConsistency(matches, [&](size_t i2) { return F2v->kps[i2].angle; }, [&](size_t i1) { return kps1[i1].angle; });   // (matches, views2, views1) :443
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    for (auto& m : matches) {
        matches12[m.second] = (int32_t)m.first;
        prev_matched_xy[2 * m.second] = F2v->kps[m.first].x; prev_matched_xy[2 * m.second + 1] = F2v->kps[m.first].y;          // :456
    }
    return (int)matches.size();
}

void hso_bow_transform(const hso_vocab_tree* T, const uint8_t* desc, int n, int levelsup, int32_t* word_id, float* weight, int32_t* node_id)
{
    const int nid_level = T->levels - levelsup;
    for (int i = 0; i < n; i++) {
        const uint8_t* f = desc + (size_t)i * 32;
        int final_id = 0, current_level = 0, nid = 0;      // nid stays 0 (root) when nid_level <= 0
        do {
            ++current_level;
            const int cb = T->child_begin[final_id], cc = T->child_count[final_id];
            final_id = cb;
            double best_d = ORBDistance(f, T->desc + (size_t)final_id * 32);
            for (int c = cb + 1; c < cb + cc; c++) {
                double d = ORBDistance(f, T->desc + (size_t)c * 32);
                if (d < best_d) { best_d = d; final_id = c; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (T->child_count[final_id] != 0);
        word_id[i] = T->word_id[final_id]; weight[i] = T->weight[final_id]; node_id[i] = T->orig_id ? T->orig_id[nid] : nid;
    }
}

void hso_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist)
{
    for (int i = 0; i < nq; i++) {
        float b1 = std::numeric_limits<float>::max(), b2 = std::numeric_limits<float>::max();
        int bi = -1;
        for (int j = 0; j < nt; j++) {
            const float d = ORBDistance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < b1) { b2 = b1; b1 = d; bi = j; }
            else if (d < b2) b2 = d;
        }
        best_idx[i] = bi;
        best_dist[i] = bi >= 0 ? (int)b1 : -1;
        second_dist[i] = b2 == std::numeric_limits<float>::max() ? -1 : (int)b2;
    }
}

void hso_rotation_consistency(const float* angle_a, const float* angle_b, int n, uint8_t* keep)
{
    MatchesIdx m;
    for (int i = 0; i < n; i++) m[i] = i;
    MatchesIdx p = RotationConsistency(m, [&](size_t i) { return angle_a[i]; }, [&](size_t i) { return angle_b[i]; });
    for (int i = 0; i < n; i++) keep[i] = p.count(i) ? 1 : 0;
}


/* ---- legacy loop-closing matchers (FeatureMatcher.cc:628-934).  cv::Mat expressions restated operation by operation:
 *   A*B+C / -A*B      one gemm: float inputs, double accumulation (alpha, beta applied in double), one rounding to float
 *   M/s, s*M          Mat scaled through convertTo: every element multiplied by (float)alpha in FLOAT arithmetic (cvtScale_<float,float,float>)
 *   row.dot(row), cv::norm   double accumulation */
namespace {
inline float gemm3(const float* A /*row of 3*/, const float* B, float c, double alpha = 1.0)
{
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)A[k] * (double)B[k];
    return (float)(alpha * s + (double)c);
}
}

int hso_search_by_projection_sim3(const hso_frame_view* KF, const float* Scw, const hso_landmark* lms, int L, int th, float th_low,
                                  uint8_t* kp_matched, int32_t* match_idx)
{
    Frame kf(*KF);
    for (int i = 0; i < L; i++) match_idx[i] = -1;
    // Decompose Scw (:641-646)
    float sRcw[9], sT[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) sRcw[3 * r + c] = Scw[4 * r + c]; sT[r] = Scw[4 * r + 3]; }
    const float scw = (float)std::sqrt((double)sRcw[0] * sRcw[0] + (double)sRcw[1] * sRcw[1] + (double)sRcw[2] * sRcw[2]);
    const float inv = (float)(1.0 / (double)scw);
    float Rcw[9], tcw[3], Ow[3];
    for (int i = 0; i < 9; i++) Rcw[i] = sRcw[i] * inv + 0.0f;
    for (int i = 0; i < 3; i++) tcw[i] = sT[i] * inv + 0.0f;
    for (int i = 0; i < 3; i++) { const float col[3] = { Rcw[i], Rcw[3 + i], Rcw[6 + i] }; Ow[i] = gemm3(col, tcw, 0.f, -1.0); }      // Ow = -Rcw.t()*tcw
    int nmatches = 0;
    for (int iMP = 0; iMP < L; iMP++) {
        const hso_landmark& lm = lms[iMP];
        if (lm.skip) continue;                                              // pMP->isBad() || spAlreadyFound.count(pMP)
        float p3Dc[3];
        for (int i = 0; i < 3; i++) p3Dc[i] = gemm3(&Rcw[3 * i], lm.pos, tcw[i]);
        if (p3Dc[2] < 0.0) continue;
        const float invz = 1 / p3Dc[2];
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = KF->fx * x + KF->cx, v = KF->fy * y + KF->cy;
        if (!(u >= KF->min_x && u < KF->max_x && v >= KF->min_y && v < KF->max_y)) continue;        // KeyFrame::IsInImage (KeyFrame.cc:371-374)
        const float maxDistance = lm.max_dist, minDistance = lm.min_dist;   // the invariance range (already 0.8 / 1.2 scaled: dist_is_invariance_range semantics)
        const float PO[3] = { lm.pos[0] - Ow[0], lm.pos[1] - Ow[1], lm.pos[2] - Ow[2] };
        const float dist = (float)std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        if (dist < minDistance || dist > maxDistance) continue;
        const double dot = (double)PO[0] * lm.normal[0] + (double)PO[1] * lm.normal[1] + (double)PO[2] * lm.normal[2];
        if (dot < 0.5 * dist) continue;
        const float radius = th * kf.landMarkSizePixels(lm) / KF->size_ref;   // KeyFrame::landMarkSizePixels projects with the KEYFRAME's pose (KeyFrame.cc:258-279)
        const std::vector<size_t> vIndices = kf.GetFeaturesInAreaNEW(u, v, radius);                 // KeyFrame::GetFeaturesInArea (KeyFrame.cc:329-368): same walk
        if (vIndices.empty()) continue;
        float bestDist = std::numeric_limits<float>::max(); int bestIdx = -1;
        for (size_t idx : vIndices) {
            if (kp_matched[idx]) continue;
            const float d = ORBDistance(lm.desc, KF->desc + idx * 32);
            if (d < bestDist) { bestDist = d; bestIdx = (int)idx; }
        }
        if (bestDist <= th_low) { kp_matched[bestIdx] = 1; match_idx[iMP] = bestIdx; nmatches++; }
    }
    return nmatches;
}

int hso_search_by_sim3(const hso_frame_view* KF1, const hso_landmark* lms1, const hso_frame_view* KF2, const hso_landmark* lms2,
                       float s12, const float* R12, const float* t12, float th, float th_high, int32_t* match12)
{
    Frame kf1(*KF1), kf2(*KF2);
    const int N1 = KF1->n, N2 = KF2->n;
    // Transformation between cameras (:757-760)
    float sR12[9], sR21[9], t21[3];
    const float a12 = (float)(double)s12, a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { sR12[3 * r + c] = R12[3 * r + c] * a12 + 0.0f; sR21[3 * r + c] = R12[3 * c + r] * a21 + 0.0f; }
    for (int i = 0; i < 3; i++) t21[i] = gemm3(&sR21[3 * i], t12, 0.f, -1.0);
    std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
    auto direction = [&](const hso_frame_view* Fsrc, const hso_landmark* lms, int n, const hso_frame_view* Fdst, Frame& dst, const float* sR, const float* tt, std::vector<int>& out) {
        for (int i = 0; i < n; i++) {
            const hso_landmark& lm = lms[i];
            if (lm.skip) continue;                                          // !pMP || vbAlreadyMatched || isBad
            float pc_src[3], pc_dst[3];
            for (int k = 0; k < 3; k++) pc_src[k] = gemm3(&Fsrc->Rcw[3 * k], lm.pos, Fsrc->tcw[k]);
            for (int k = 0; k < 3; k++) pc_dst[k] = gemm3(&sR[3 * k], pc_src, tt[k]);
            float uv[3];
            if (!dst.CameraProject(pc_dst, uv)) continue;
            const float dist3D = (float)std::sqrt((double)pc_dst[0] * pc_dst[0] + (double)pc_dst[1] * pc_dst[1] + (double)pc_dst[2] * pc_dst[2]);
            if (dist3D < lm.min_dist || dist3D > lm.max_dist) continue;
            const float radius = th * dst.landMarkSizePixels(lm) / Fdst->size_ref;
            const std::vector<size_t> vIndices = dst.GetFeaturesInAreaNEW(uv[0], uv[1], radius);
            if (vIndices.empty()) continue;
            float bestDist = std::numeric_limits<float>::max(); int bestIdx = -1;
            for (size_t idx : vIndices) {
                const float d = ORBDistance(lm.desc, Fdst->desc + idx * 32);
                if (d < bestDist) { bestDist = d; bestIdx = (int)idx; }
            }
            if (bestDist <= th_high) out[i] = bestIdx;
        }
    };
    direction(KF1, lms1, N1, KF2, kf2, sR21, t21, vnMatch1);
    direction(KF2, lms2, N2, KF1, kf1, sR12, t12, vnMatch2);
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    return nFound;
}

} // extern "C"
