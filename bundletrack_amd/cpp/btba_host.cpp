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
    float tr = 0.0f;                                            // trace(R1 R2^T) = sum_ij R1_ij R2_ij
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tr += A(i, j) * B(i, j);
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
        rot_diff = rot_diff * 180.0f / (float)M_PI;
        if (rot_diff < yml->keyframe_min_rot) return false;
    }
    _keyframes.push_back(frame);
    return true;
}

std::vector<std::shared_ptr<Frame>> KeyframeMemory::selectKeyFramesForBA(const std::shared_ptr<Frame> &newframe)
{
    std::vector<std::shared_ptr<Frame>> frames = { newframe };       // insertion order (the reference: a std::set in pointer order)
    auto has = [&](const std::shared_ptr<Frame> &f) { return std::find(frames.begin(), frames.end(), f) != frames.end(); };
    auto by_id = [](const std::shared_ptr<Frame> &a, const std::shared_ptr<Frame> &b) { return a->_id < b->_id; };
    if ((int)(_keyframes.size() + frames.size()) <= yml->max_BA_frames) {
        for (const auto &kf : _keyframes) if (!has(kf)) frames.push_back(kf);
        std::sort(frames.begin(), frames.end(), by_id);
        return frames;
    }
    if (!has(_keyframes[0])) frames.push_back(_keyframes[0]);
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
        frames.push_back(best_kf);
    }
    std::sort(frames.begin(), frames.end(), by_id);
    return frames;
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
