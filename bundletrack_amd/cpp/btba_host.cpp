// btba_host.cpp -- see btba_host.hpp.  Plain C++17 on top of include/btba.h.
#include "btba_host.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <cmath>
#include <limits>

namespace btba {

OptimizerGpu::OptimizerGpu(std::shared_ptr<Config> yml1, void *hip_stream) : yml(std::move(yml1))
{
    if (!yml) yml = std::make_shared<Config>();
    // the structs of include/btba.h (btba_params, btba_stats, btba_zn_aux) carry no size field: a library built from another header version must
    // not be called with them
    if (btba_version() != BTBA_VERSION) throw Error(BTBA_EINVAL, "libbtba.so was built from another include/btba.h (btba_version() != BTBA_VERSION)");
    const int rc = btba_workspace_create_on_stream(&ws_, hip_stream);       // nullptr = the legacy NULL stream, as the reference
    if (rc != BTBA_OK) throw Error(rc, "btba_workspace_create_on_stream");
}

OptimizerGpu::~OptimizerGpu() { btba_workspace_destroy(ws_); }

void OptimizerGpu::optimizeFrames(const std::vector<EntryJ> &global_corres, const std::vector<int> &n_match_per_pair, int n_frames, int H, int W,
                                  const std::vector<float *> &depths_gpu, const std::vector<uchar4 *> & /*colors_gpu*/, const std::vector<float4 *> &normals_gpu,
                                  std::vector<Matrix4f> &poses, const Matrix3f &K)
{
    if ((int)depths_gpu.size() != n_frames || (int)normals_gpu.size() != n_frames || (int)poses.size() != n_frames) throw Error(BTBA_EINVAL, "optimizeFrames");
    btba_params prm;
    btba_params_default(&prm);
    prm.n_gn_iters = yml->num_iter_outter;                      // CUDASolverBundling.cpp:193
    prm.n_pcg_iters = yml->num_iter_inner;
    prm.robust_delta = yml->robust_delta;
    prm.image_downscale = yml->image_downscale;                 // LossGPU.cu:55
    prm.dense_dist_thresh = yml->p2p_max_dist;                  // CUDASolverBundling.cpp:93
    prm.dense_normal_thresh = std::cos(yml->p2p_max_normal_angle / 180.0 * M_PI);      // :94
    float Krm[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Krm[3 * r + c] = K(r, c);
    std::vector<float> P(16 * (size_t)n_frames);                // column-major Eigen -> row-major float4x4, LossGPU.cu:88-97
    for (int i = 0; i < n_frames; i++) for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) P[16 * (size_t)i + 4 * r + c] = poses[i](r, c);
    std::vector<const float *> depth(n_frames), nrm(n_frames);
    for (int i = 0; i < n_frames; i++) { depth[i] = depths_gpu[i]; nrm[i] = reinterpret_cast<const float *>(normals_gpu[i]); }
    // stored and never read by the reference (SBA.cpp:85): any length is legal there; only a vector of P = n(n-1)/2 segment
    // lengths is handed on (it lets the library skip its host pass), anything else is ignored
    const int *nm = ((long)n_match_per_pair.size() == (long)n_frames * (n_frames - 1) / 2) ? n_match_per_pair.data() : nullptr;
    int rc;
    if (persistent_frame_cache) {
        if ((int)frame_ids.size() != n_frames) throw Error(BTBA_EINVAL, "optimizeFrames: frame_ids");
        if (keyed_correspondences) prm.flags |= BTBA_FLAG_KEYED_CORR;
        rc = btba_optimize_frames_keyed(ws_, &prm, n_frames, H, W, Krm, global_corres.data(), (uint32_t)global_corres.size(), nm,
                                        depth.data(), nrm.data(), frame_ids.data(), nullptr, 0, P.data(), &last_stats);
    } else {
        rc = btba_optimize_frames(ws_, &prm, n_frames, H, W, Krm, global_corres.data(), (uint32_t)global_corres.size(), nm,
                                  depth.data(), nrm.data(), nullptr, 0, P.data(), &last_stats);
    }
    if (rc != BTBA_OK) throw Error(rc, "btba_optimize_frames");
    for (int i = 0; i < n_frames; i++) for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) poses[i](r, c) = P[16 * (size_t)i + 4 * r + c];      // :121-130
}

// Thin SVD of a 3x3 matrix S = U diag(sig) V^T by one-sided (Hestenes) Jacobi rotations on the columns, singular values
// sorted descending like Eigen::JacobiSVD; a vanishing third singular value (coplanar points) gets the cross product of
// the first two left vectors, so U stays orthonormal.
static void svd3(const float S[9], float U[9], float sig[3], float V[9])
{
    double A[9], W[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; k++) A[k] = S[k];
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0.0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double al = 0, be = 0, ga = 0;
                for (int k = 0; k < 3; k++) { al += A[3 * k + p] * A[3 * k + p]; be += A[3 * k + q] * A[3 * k + q]; ga += A[3 * k + p] * A[3 * k + q]; }
                if (ga == 0.0 || std::fabs(ga) <= 1e-15 * std::sqrt(al * be)) continue;
                off = std::max(off, std::fabs(ga) / std::sqrt(al * be));
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; k++) {
                    const double ap = A[3 * k + p], aq = A[3 * k + q];
                    A[3 * k + p] = c * ap - sn * aq; A[3 * k + q] = sn * ap + c * aq;
                    const double wp = W[3 * k + p], wq = W[3 * k + q];
                    W[3 * k + p] = c * wp - sn * wq; W[3 * k + q] = sn * wp + c * wq;
                }
            }
        if (off < 1e-14) break;
    }
    double norm[3];
    int order[3] = { 0, 1, 2 };
    for (int j = 0; j < 3; j++) norm[j] = std::sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    std::sort(order, order + 3, [&](int a, int b) { return norm[a] > norm[b]; });
    double Ud[9];
    for (int j = 0; j < 3; j++) {
        const int o = order[j];
        sig[j] = (float)norm[o];
        for (int k = 0; k < 3; k++) { V[3 * k + j] = (float)W[3 * k + o]; Ud[3 * k + j] = norm[o] > 0 ? A[3 * k + o] / norm[o] : 0.0; }
    }
    if (norm[order[2]] <= 1e-12 * norm[order[0]]) {                   // rank 2: complete U with u0 x u1
        Ud[2] = Ud[3] * Ud[7] - Ud[6] * Ud[4]; Ud[5] = Ud[6] * Ud[1] - Ud[0] * Ud[7]; Ud[8] = Ud[0] * Ud[4] - Ud[3] * Ud[1];
    }
    for (int k = 0; k < 9; k++) U[k] = (float)Ud[k];
}

void solveRigidTransformBetweenPoints(const std::vector<float> &points1, const std::vector<float> &points2, Matrix4f &pose)
{
    pose = Matrix4f::Identity();
    const size_t n = points1.size() / 3;
    if (n < 3 || points1.size() != points2.size() || points1.size() % 3) return;      // the reference asserts
    float m1[3] = { 0, 0, 0 }, m2[3] = { 0, 0, 0 };
    for (size_t i = 0; i < n; i++) for (int c = 0; c < 3; c++) { m1[c] += points1[3 * i + c]; m2[c] += points2[3 * i + c]; }
    for (int c = 0; c < 3; c++) { m1[c] /= (float)n; m2[c] /= (float)n; }
    float S[9] = {};                                                  // P^T Q, row-major
    for (size_t i = 0; i < n; i++)
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) S[3 * r + c] += (points1[3 * i + r] - m1[r]) * (points2[3 * i + c] - m2[c]);
    for (float v : S) if (!std::isfinite(v)) return;
    float U[9], sig[3], V[9];
    svd3(S, U, sig, V);
    auto vut = [&](float R[9]) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { R[3 * r + c] = 0; for (int k = 0; k < 3; k++) R[3 * r + c] += V[3 * r + k] * U[3 * c + k]; } };
    float R[9];
    vut(R);
    for (int r = 0; r < 3; r++)                                        // (R^T R).isApprox(I), float precision
        for (int c = 0; c < 3; c++) {
            float d = 0;
            for (int k = 0; k < 3; k++) d += R[3 * k + r] * R[3 * k + c];
            if (std::fabs(d - (r == c ? 1.0f : 0.0f)) > 1e-5f) return;
        }
    const float det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0) { for (int k = 0; k < 3; k++) V[3 * k + 2] = -V[3 * k + 2]; vut(R); }
    Matrix4f out = Matrix4f::Identity();
    for (int r = 0; r < 3; r++) {
        float t = m2[r];
        for (int c = 0; c < 3; c++) { out(r, c) = R[3 * r + c]; t -= R[3 * r + c] * m1[c]; }
        out(r, 3) = t;
    }
    for (float v : out.d) if (!std::isfinite(v)) return;
    pose = out;
}

std::string formatPoseTxt(const Matrix4f &M)
{
    char cell[16][32];
    size_t width = 0;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            std::snprintf(cell[4 * r + c], sizeof cell[0], "%.10g", (double)M(r, c));
            width = std::max(width, std::strlen(cell[4 * r + c]));
        }
    std::string out;
    for (int r = 0; r < 4; r++) {
        for (int c = 0; c < 4; c++) {
            if (c) out += ' ';
            out.append(width - std::strlen(cell[4 * r + c]), ' ');
            out += cell[4 * r + c];
        }
        out += '\n';
    }
    return out;
}

float rotationGeodesicDistance(const Matrix4f &A, const Matrix4f &B)
{
    // trace(R1 R2^T): diagonal entry i is the dot product of row i of R1 with row i of R2, each summed left to right and then the
    // three of them (the shape of Eigen's coefficient-based 3 x 3 product; oracle/btba_oracle_keyframes.c sums the same way)
    float tr = 0.0f;
    for (int i = 0; i < 3; i++) {
        float dot = A(i, 0) * B(i, 0);
        dot += A(i, 1) * B(i, 1);
        dot += A(i, 2) * B(i, 2);
        tr = i == 0 ? dot : tr + dot;
    }
    float tmp = (tr - 1.0f) / 2.0f;
    tmp = std::max(std::min(1.0f, tmp), -1.0f);
    return std::acos(tmp);
}

Window marshalWindow(std::vector<std::shared_ptr<Frame>> local_frames, const std::map<std::pair<int, int>, Correspondences> &matches,
                     const std::shared_ptr<Frame> &newframe, int min_fm_edges_newframe)
{
    Window w;
    std::sort(local_frames.begin(), local_frames.end(), [](const std::shared_ptr<Frame> &a, const std::shared_ptr<Frame> &b) { return a->_id < b->_id; });   // :286
    for (size_t i = 0; i < local_frames.size(); i++)
        for (size_t j = i + 1; j < local_frames.size(); j++) {
            const auto &frameA = local_frames[j], &frameB = local_frames[i];
            const auto it = matches.find({ frameA->_id, frameB->_id });
            int m = 0;
            if (it != matches.end()) {
                m = (int)(it->second.ptA_cam.size() / 3);
                for (int k = 0; k < m; k++) {
                    EntryJ e;
                    e.imgIdx_i = (uint32_t)i; e.imgIdx_j = (uint32_t)j;                      // :311-316: pos_j = ptA, pos_i = ptB
                    for (int c = 0; c < 3; c++) { e.pos_j[c] = it->second.ptA_cam[3 * k + c]; e.pos_i[c] = it->second.ptB_cam[3 * k + c]; }
                    w.global_corres.push_back(e);
                    if (frameA == newframe || frameB == newframe) w.n_edges_newframe++;
                }
            }
            w.n_match_per_pair.push_back(m);
        }
    w.frames = std::move(local_frames);
    w.run_ba = w.n_edges_newframe > min_fm_edges_newframe;
    if (!w.run_ba) newframe->_status = Frame::NO_BA;
    return w;
}

bool KeyframeMemory::checkAndAddKeyframe(const std::shared_ptr<Frame> &frame)
{
    if (frame->_id == 0) { _keyframes.push_back(frame); return true; }
    if (frame->_status != Frame::OTHER) return false;
    if (frame->_n_keypts < yml->keyframe_min_feat_num) return false;
    for (const auto &kf : _keyframes) {
        float rot_diff = rotationGeodesicDistance(frame->_pose_in_model, kf->_pose_in_model);
        rot_diff = (float)((double)(rot_diff * 180.0f) / M_PI);          // rot_diff*180/M_PI: the product in float, the division in double, stored as float (:208)
        if (rot_diff < yml->keyframe_min_rot) return false;
    }
    _keyframes.push_back(frame);
    return true;
}

std::vector<std::shared_ptr<Frame>> KeyframeMemory::selectKeyFramesForBA(const std::shared_ptr<Frame> &newframe)
{
    // The reference keeps the chosen set in a std::set<std::shared_ptr<Frame>>: ordered by the frames' ADDRESSES, which decides the order cum_dist is
    // summed in and nothing else (Bundler.cpp:252-256; the caller sorts by id, :286).  Here the set is ordered by frame id -- what an allocator that
    // hands out ascending addresses gives the reference -- so a selection does not depend on where the heap put a frame.
    auto by_id = [](const std::shared_ptr<Frame> &a, const std::shared_ptr<Frame> &b) { return a->_id < b->_id; };
    std::vector<std::shared_ptr<Frame>> frames = { newframe };
    auto has = [&](const std::shared_ptr<Frame> &f) { return std::find(frames.begin(), frames.end(), f) != frames.end(); };
    auto insert = [&](const std::shared_ptr<Frame> &f) { if (!has(f)) frames.insert(std::upper_bound(frames.begin(), frames.end(), f, by_id), f); };
    if ((int)(_keyframes.size() + frames.size()) <= yml->max_BA_frames) {
        for (const auto &kf : _keyframes) insert(kf);
        return frames;
    }
    insert(_keyframes[0]);
    while ((int)frames.size() < yml->max_BA_frames) {                // "greedy_rot"
        float best_dist = std::numeric_limits<float>::max();
        std::shared_ptr<Frame> best_kf;
        for (const auto &kf : _keyframes) {
            if (has(kf)) continue;
            float cum_dist = 0.0f;
            for (const auto &f : frames) cum_dist += rotationGeodesicDistance(kf->_pose_in_model, f->_pose_in_model);
            if (cum_dist < best_dist) { best_dist = cum_dist; best_kf = kf; }
        }
        if (!best_kf) break;
        insert(best_kf);
    }
    return frames;
}

// ---- the slice of SiftManager that Bundler calls ----------------------------------------------------------------
void FeatureManager::forgetFrame(const std::shared_ptr<Frame> &frame)
{
    for (auto it = _matches.begin(); it != _matches.end();)
        it = (it->first.first == frame->_id || it->first.second == frame->_id) ? _matches.erase(it) : std::next(it);
}

int FeatureManager::countInlierCorres(const std::shared_ptr<Frame> &frameA, const std::shared_ptr<Frame> &frameB) const
{
    const auto it = _matches.find({ frameA->_id, frameB->_id });
    return it == _matches.end() ? 0 : (int)(it->second.ptA_cam.size() / 3);      // every stored match is an inlier (pruned ones are removed, :717-738)
}

static void transform_points(const Matrix4f &T, const std::vector<float> &in, std::vector<float> &out)       // pcl::transformPointWithNormal, fp32
{
    out.resize(in.size());
    for (size_t i = 0; i + 2 < in.size(); i += 3)
        for (int r = 0; r < 3; r++) out[i + r] = T(r, 0) * in[i] + T(r, 1) * in[i + 1] + T(r, 2) * in[i + 2] + T(r, 3);
}

Matrix4f FeatureManager::procrustesByCorrespondence(const std::shared_ptr<Frame> &frameA, const std::shared_ptr<Frame> &frameB)
{
    Matrix4f pose = Matrix4f::Identity();
    if (countInlierCorres(frameA, frameB) < 5) return pose;                      // :527
    const Correspondences &m = _matches.at({ frameA->_id, frameB->_id });
    std::vector<float> src, dst;
    transform_points(frameA->_pose_in_model, m.ptA_cam, src);                    // :537-538
    transform_points(frameB->_pose_in_model, m.ptB_cam, dst);
    solveRigidTransformBetweenPoints(src, dst, pose);                            // :544 (the reference then aborts on a mean error > 1e-3 between neighbours)
    return pose;
}

void FeatureManager::runRansacMultiPairGPU(btba_workspace *ws, const std::vector<std::pair<std::shared_ptr<Frame>, std::shared_ptr<Frame>>> &pairs,
                                           int max_iter, float inlier_dist)
{
    if (pairs.empty()) return;
    std::vector<float> A, B;                                                     // float4 (x, y, z, 1) of all pairs back to back (:680-705)
    std::vector<int32_t> n_pts;
    std::vector<float> pa, pb;
    for (const auto &pr : pairs) {
        const auto it = _matches.find({ pr.first->_id, pr.second->_id });
        const size_t n = it == _matches.end() ? 0 : it->second.ptA_cam.size() / 3;
        if (n) {
            transform_points(pr.first->_pose_in_model, it->second.ptA_cam, pa);
            transform_points(pr.second->_pose_in_model, it->second.ptB_cam, pb);
            for (size_t i = 0; i < n; i++) {
                A.insert(A.end(), { pa[3 * i], pa[3 * i + 1], pa[3 * i + 2], 1.0f });
                B.insert(B.end(), { pb[3 * i], pb[3 * i + 1], pb[3 * i + 2], 1.0f });
            }
        }
        n_pts.push_back((int32_t)n);
    }
    std::vector<int32_t> ids(std::max<size_t>(A.size() / 4, 1)), n_in(pairs.size()), best(pairs.size());
    const int rc = btba_ransac_pairs(ws, (int)pairs.size(), A.data(), B.data(), n_pts.data(), max_iter, inlier_dist, /*samples=*/nullptr, /*seed=*/0,
                                     ids.data(), n_in.data(), best.data(), nullptr, nullptr, nullptr);
    if (rc != BTBA_OK) throw Error(rc, "btba_ransac_pairs");
    size_t o = 0;
    for (size_t p = 0; p < pairs.size(); p++) {                                  // :715-739
        const auto it = _matches.find({ pairs[p].first->_id, pairs[p].second->_id });
        if (it != _matches.end()) {
            Correspondences kept;
            if (n_in[p] >= 5)
                for (int k = 0; k < n_in[p]; k++) {
                    const int i = ids[o + k];
                    kept.ptA_cam.insert(kept.ptA_cam.end(), it->second.ptA_cam.begin() + 3 * i, it->second.ptA_cam.begin() + 3 * i + 3);
                    kept.ptB_cam.insert(kept.ptB_cam.end(), it->second.ptB_cam.begin() + 3 * i, it->second.ptB_cam.begin() + 3 * i + 3);
                }
            it->second = std::move(kept);                                        // fewer than 5 survivors: every match goes (:733-737)
        }
        o += (size_t)n_pts[p];
    }
}

// ---- Bundler ---------------------------------------------------------------------------------------------------
Bundler::Bundler(std::shared_ptr<Config> yml1, std::shared_ptr<FeatureManager> fm, const Matrix3f &K1, int H1, int W1, OptimizeFn optimize)
    : yml(yml1 ? std::move(yml1) : std::make_shared<Config>()), _fm(std::move(fm)), memory(yml), K(K1), H(H1), W(W1), optimize_(std::move(optimize))
{
    if (!_fm) throw Error(BTBA_EINVAL, "Bundler: no feature manager");
}

void Bundler::processNewFrame(std::shared_ptr<Frame> frame)
{
    _newframe = frame;
    std::shared_ptr<Frame> last_frame;
    if (!_frames.empty()) {                                                      // :75-80
        last_frame = _frames.back();
        frame->_id = last_frame->_id + 1;
        frame->_pose_in_model = last_frame->_pose_in_model;
    }
    if (frame->_roi[1] - frame->_roi[0] < 10 || frame->_roi[3] - frame->_roi[2] < 10) {      // :88-93: "cloud is empty, marked FAIL" -- a plain return:
        frame->_status = Frame::FAIL;                                            // no forgetFrame, no re-initialisation request
        return;
    }
    if (frame->_status == Frame::FAIL) {                                         // :96-101 (marked FAIL before it got here)
        _fm->forgetFrame(frame);
        _need_reinit = true;
        return;
    }
    try {
        _fm->detectFeature(frame);                                               // :103-117
    } catch (const std::exception &) {
        frame->_status = Frame::FAIL;
        _need_reinit = true;
        _fm->forgetFrame(frame);
        return;
    }
    if (last_frame) {
        _fm->findCorres(frame, last_frame);                                      // :121
        if (frame->_status == Frame::FAIL) {
            _need_reinit = true;
            _fm->forgetFrame(frame);
            return;
        }
        const Matrix4f offset = _fm->procrustesByCorrespondence(frame, last_frame);      // :134-136: pose = offset * pose
        Matrix4f moved;
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                float acc = 0.0f;
                for (int k = 0; k < 4; k++) acc += offset(r, k) * frame->_pose_in_model(k, c);
                moved(r, c) = acc;
            }
        frame->_pose_in_model = moved;
    }
    if ((int)_frames.size() >= yml->window_size + 3) {                           // :150-158
        if (std::find(memory._keyframes.begin(), memory._keyframes.end(), _frames.front()) == memory._keyframes.end()) _fm->forgetFrame(_frames.front());
        _frames.pop_front();
    }
    _frames.push_back(frame);
    if (frame->_id == 0) {                                                       // :162-166
        memory.checkAndAddKeyframe(frame);
        return;
    }
    _local_frames = memory.selectKeyFramesForBA(frame);                          // :168-172
    optimizeGPU();
    if (frame->_status == Frame::FAIL) {                                         // :174-180
        _fm->forgetFrame(frame);
        _frames.pop_back();
        if (own_opt_ && own_opt_->persistent_frame_cache) btba_frame_cache_evict(own_opt_->workspace(), (uint64_t)frame->_id);      // its id is handed out again
        _need_reinit = true;
        return;
    }
    memory.checkAndAddKeyframe(frame);                                           // :182
    if (!yml->pose_dir.empty()) saveNewframeResult();
}

void Bundler::optimizeGPU()
{
    std::sort(_local_frames.begin(), _local_frames.end(), [](const std::shared_ptr<Frame> &a, const std::shared_ptr<Frame> &b) { return a->_id < b->_id; });   // :286
    for (size_t i = 0; i < _local_frames.size(); i++)
        for (size_t j = i + 1; j < _local_frames.size(); j++) _fm->findCorres(_local_frames[j], _local_frames[i]);                                       // :303
    last_window = marshalWindow(_local_frames, _fm->_matches, _newframe, yml->min_fm_edges_newframe);          // :296-347 (sets NO_BA)
    if (!last_window.run_ba) return;
    const auto &fr = last_window.frames;
    std::vector<float *> depths_gpu;
    std::vector<uchar4 *> colors_gpu;
    std::vector<float4 *> normals_gpu;
    std::vector<Matrix4f> poses;
    for (const auto &f : fr) { depths_gpu.push_back(f->_depth_gpu); colors_gpu.push_back(f->_color_gpu); normals_gpu.push_back(f->_normal_gpu); poses.push_back(f->_pose_in_model); }
    if (optimize_) {
        optimize_(last_window.global_corres, last_window.n_match_per_pair, (int)fr.size(), H, W, depths_gpu, colors_gpu, normals_gpu, poses, K);
    } else {
        if (!own_opt_) own_opt_ = std::make_unique<OptimizerGpu>(yml);
        if (own_opt_->persistent_frame_cache) {
            own_opt_->frame_ids.clear();
            for (const auto &f : fr) own_opt_->frame_ids.push_back((uint64_t)f->_id);
        }
        own_opt_->optimizeFrames(last_window.global_corres, last_window.n_match_per_pair, (int)fr.size(), H, W, depths_gpu, colors_gpu, normals_gpu, poses, K);
    }
    n_ba_calls++;
    for (size_t i = 0; i < fr.size(); i++) fr[i]->_pose_in_model = poses[i];     // :353-357
}

Matrix4f inverse(const Matrix4f &M)
{
    double a[4][8];                                                              // [M | I] -> [I | M^-1], Gauss-Jordan with partial pivoting
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { a[r][c] = M(r, c); a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int k = 0; k < 4; k++) {
        int piv = k;
        for (int r = k + 1; r < 4; r++) if (std::fabs(a[r][k]) > std::fabs(a[piv][k])) piv = r;
        if (piv != k) for (int c = 0; c < 8; c++) std::swap(a[k][c], a[piv][c]);
        const double d = a[k][k];                                                // 0 for a singular matrix: the result is inf / nan, as Eigen's
        for (int c = 0; c < 8; c++) a[k][c] /= d;
        for (int r = 0; r < 4; r++) {
            if (r == k) continue;
            const double f = a[r][k];
            for (int c = 0; c < 8; c++) a[r][c] -= f * a[k][c];
        }
    }
    Matrix4f R;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) R(r, c) = (float)a[r][4 + c];
    return R;
}

void Bundler::saveNewframeResult()
{
    if (yml->pose_dir.empty() || !_newframe) return;
    const std::string name = _newframe->_id_str.empty() ? std::to_string(_newframe->_id) : _newframe->_id_str;
    std::ofstream ff(yml->pose_dir + "/" + name + ".txt");                       // the caller creates the directory (the reference: system("mkdir -p"))
    ff << formatPoseTxt(inverse(_newframe->_pose_in_model));                     // ob_in_cam = cur_in_model.inverse(), :371-375
}

// ---- problem dumps (bundletrack_amd/problem_io.py documents the layout) ---------------------------------------
namespace {
const char kProblemMagic[8] = { 'B', 'T', 'B', 'A', 'P', 'R', 'B', '1' };
struct ProblemHeader { char magic[8]; int32_t n_frames, H, W; uint32_t n_corr; int32_t flags; float image_downscale; };
static_assert(sizeof(ProblemHeader) == 32, "problem dump header");
template <class T> void read_array(std::ifstream &f, std::vector<T> &v, size_t n, const std::string &path)
{
    v.resize(n);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(sizeof(T) * n));
    if (!f) throw Error(BTBA_EINVAL, path + ": truncated problem dump");
}
template <class T> void write_array(std::ofstream &f, const std::vector<T> &v) { f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(sizeof(T) * v.size())); }
}  // namespace

ProblemDump loadProblem(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    ProblemHeader h{};
    f.read(reinterpret_cast<char *>(&h), sizeof h);
    if (!f || std::memcmp(h.magic, kProblemMagic, 8) != 0) throw Error(BTBA_EINVAL, path + ": not a BTBAPRB1 problem dump");
    if (h.n_frames < 1 || h.H < 1 || h.W < 1) throw Error(BTBA_EINVAL, path + ": corrupt problem dump header");
    ProblemDump pb;
    pb.n_frames = h.n_frames; pb.H = h.H; pb.W = h.W; pb.image_downscale = h.image_downscale;
    f.read(reinterpret_cast<char *>(pb.K), sizeof pb.K);
    const size_t N = (size_t)h.n_frames, P = N * (N - 1) / 2, npix = (size_t)h.H * h.W;
    read_array(f, pb.corr, h.n_corr, path);
    std::vector<int32_t> nm;
    read_array(f, nm, P, path);
    pb.n_match_per_pair.assign(nm.begin(), nm.end());
    read_array(f, pb.poses_init, 16 * N, path);
    if (h.flags & 1) read_array(f, pb.poses_gt, 16 * N, path);
    read_array(f, pb.depth, N * npix, path);
    read_array(f, pb.normals, 4 * N * npix, path);
    if (f.peek() != std::ifstream::traits_type::eof()) throw Error(BTBA_EINVAL, path + ": trailing bytes after the problem dump");
    return pb;
}

void saveProblem(const std::string &path, const ProblemDump &pb)
{
    const size_t N = (size_t)pb.n_frames, P = N * (N - 1) / 2, npix = (size_t)pb.H * pb.W;
    if (pb.n_frames < 1 || pb.n_match_per_pair.size() != P || pb.poses_init.size() != 16 * N || (!pb.poses_gt.empty() && pb.poses_gt.size() != 16 * N) ||
        pb.depth.size() != N * npix || pb.normals.size() != 4 * N * npix)
        throw Error(BTBA_EINVAL, "saveProblem: array sizes do not match n_frames / H / W");
    ProblemHeader h{};
    std::memcpy(h.magic, kProblemMagic, 8);
    h.n_frames = pb.n_frames; h.H = pb.H; h.W = pb.W; h.n_corr = (uint32_t)pb.corr.size(); h.flags = pb.poses_gt.empty() ? 0 : 1;
    h.image_downscale = pb.image_downscale;
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(&h), sizeof h);
    f.write(reinterpret_cast<const char *>(pb.K), sizeof pb.K);
    write_array(f, pb.corr);
    write_array(f, std::vector<int32_t>(pb.n_match_per_pair.begin(), pb.n_match_per_pair.end()));
    write_array(f, pb.poses_init);
    write_array(f, pb.poses_gt);
    write_array(f, pb.depth);
    write_array(f, pb.normals);
    if (!f) throw Error(BTBA_EINVAL, path + ": write failed");
}

}  // namespace btba
