// host_driver.cpp -- test driver for the C++ host layer (bundletrack_amd/cpp/btba_host.*): what Bundler.cpp does with
// OptimizerGpu, on a problem dumped by the Python tests.  Built by __graft_entry__.build().
//   host_driver ba <problem.bin> <poses_out.bin>        window -> marshalWindow -> OptimizerGpu::optimizeFrames (GPU)
//   host_driver keyframes <poses.bin> <ids_out.bin>     checkAndAddKeyframe over a pose sequence + selectKeyFramesForBA (CPU); input: int32 M, max_BA_frames; float min_rot; M poses
//   host_driver problem <dump.btba> <copy_out.btba>     loadProblem -> saveProblem round trip + a one-line summary (CPU)
//   host_driver kabsch <pairs.bin> <poses_out.bin>      solveRigidTransformBetweenPoints over a list of point-set pairs (CPU)
//   host_driver posetxt <poses.bin> <out.txt>           formatPoseTxt of every 4x4 (row-major floats) in the file (CPU)
//   host_driver bundler <scenario.bin> <log.txt>        a frame sequence through btba::Bundler::processNewFrame: CPU with a stand-in
//                                                       optimiser (control flow), or with images on the GPU (OptimizerGpu, optional RANSAC)
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>

#include "../../bundletrack_amd/cpp/btba_host.hpp"

using namespace btba;

template <class T> static void rd(std::ifstream &f, T *p, size_t n) { f.read(reinterpret_cast<char *>(p), sizeof(T) * n); if (!f) throw std::runtime_error("short read"); }
#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 3; } } while (0)

static Matrix4f from_rowmajor(const float *m) { Matrix4f M; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M(r, c) = m[4 * r + c]; return M; }

static int run_ba(const char *in, const char *out)
{
    std::ifstream f(in, std::ios::binary);
    int32_t hdr[4];
    rd(f, hdr, 4);
    const int N = hdr[0], H = hdr[1], W = hdr[2], C = hdr[3];
    float Krm[9];
    rd(f, Krm, 9);
    std::vector<EntryJ> corr(C);
    rd(f, corr.data(), C);
    std::vector<float> P(16 * (size_t)N);
    rd(f, P.data(), P.size());
    Matrix3f K;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) K(r, c) = Krm[3 * r + c];
    // frames with non-contiguous ids, handed over in shuffled order: marshalWindow sorts by id (Bundler.cpp:286)
    std::vector<std::shared_ptr<Frame>> frames(N);
    std::vector<float> buf((size_t)H * W * 4);
    for (int k = 0; k < N; k++) {
        auto fr = std::make_shared<Frame>();
        fr->_id = 10 * k + 3;
        fr->_pose_in_model = from_rowmajor(&P[16 * (size_t)k]);
        rd(f, buf.data(), (size_t)H * W);
        HIP_OK(hipMalloc(reinterpret_cast<void **>(&fr->_depth_gpu), sizeof(float) * H * W));
        HIP_OK(hipMemcpy(fr->_depth_gpu, buf.data(), sizeof(float) * H * W, hipMemcpyHostToDevice));
        rd(f, buf.data(), (size_t)H * W * 4);
        HIP_OK(hipMalloc(reinterpret_cast<void **>(&fr->_normal_gpu), sizeof(float) * 4 * H * W));
        HIP_OK(hipMemcpy(fr->_normal_gpu, buf.data(), sizeof(float) * 4 * H * W, hipMemcpyHostToDevice));
        frames[k] = fr;
    }
    std::map<std::pair<int, int>, Correspondences> matches;     // _fm->_matches[{frameA (newer), frameB}]
    for (const EntryJ &e : corr) {
        auto &m = matches[{ frames[e.imgIdx_j]->_id, frames[e.imgIdx_i]->_id }];
        for (int c = 0; c < 3; c++) { m.ptA_cam.push_back(e.pos_j[c]); m.ptB_cam.push_back(e.pos_i[c]); }
    }
    auto yml = std::make_shared<Config>();
    std::vector<std::shared_ptr<Frame>> local = frames;
    std::mt19937 rng(5);
    std::shuffle(local.begin(), local.end(), rng);
    Window w = marshalWindow(local, matches, frames[N - 1], yml->min_fm_edges_newframe);
    if (!w.run_ba) { std::fprintf(stderr, "window gated: NO_BA\n"); return 4; }
    std::vector<float> result;
    for (int pass = 0; pass < 3; pass++) {                      // stateless call, then twice with the persistent frame cache
        OptimizerGpu *opt_ptr = nullptr;
        static std::unique_ptr<OptimizerGpu> keyed;
        std::unique_ptr<OptimizerGpu> plain;
        if (pass == 0) { plain.reset(new OptimizerGpu(yml)); opt_ptr = plain.get(); }
        else { if (!keyed) { keyed.reset(new OptimizerGpu(yml)); keyed->persistent_frame_cache = true; } opt_ptr = keyed.get(); }
        std::vector<float *> depths_gpu; std::vector<uchar4 *> colors_gpu; std::vector<float4 *> normals_gpu; std::vector<Matrix4f> poses;
        opt_ptr->frame_ids.clear();
        for (const auto &fr : w.frames) {
            depths_gpu.push_back(fr->_depth_gpu); colors_gpu.push_back(fr->_color_gpu); normals_gpu.push_back(fr->_normal_gpu);
            poses.push_back(from_rowmajor(&P[16 * (size_t)((fr->_id - 3) / 10)]));
            opt_ptr->frame_ids.push_back((uint64_t)fr->_id);
        }
        opt_ptr->optimizeFrames(w.global_corres, w.n_match_per_pair, (int)w.frames.size(), H, W, depths_gpu, colors_gpu, normals_gpu, poses, K);
        std::printf("pass %d: frames cached in this call %d, solve %.3f ms\n", pass, opt_ptr->last_stats.cache_frames_built, opt_ptr->last_stats.ms_solve);
        for (const Matrix4f &M : poses) for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) result.push_back(M(r, c));
        if (pass == 2) keyed.reset();
    }
    // error path: a status becomes an exception, nothing exits
    try {
        OptimizerGpu bad(yml);
        std::vector<Matrix4f> one(1, Matrix4f::Identity());
        bad.optimizeFrames({}, {}, 1, H, W, { frames[0]->_depth_gpu }, { nullptr }, { frames[0]->_normal_gpu }, one, K);
        return 5;
    } catch (const Error &e) { if (e.status != BTBA_EINVAL) return 6; }
    std::ofstream o(out, std::ios::binary);
    o.write(reinterpret_cast<const char *>(result.data()), sizeof(float) * result.size());
    for (auto &fr : frames) { (void)hipFree(fr->_depth_gpu); (void)hipFree(fr->_normal_gpu); }
    return 0;
}

static int run_keyframes(const char *in, const char *out)
{
    std::ifstream f(in, std::ios::binary);
    int32_t hdr[2];
    rd(f, hdr, 2);
    const int M = hdr[0];
    auto yml = std::make_shared<Config>();
    yml->max_BA_frames = hdr[1];
    float min_rot;
    rd(f, &min_rot, 1);
    yml->keyframe_min_rot = min_rot;
    std::vector<float> P(16 * (size_t)M);
    rd(f, P.data(), P.size());
    KeyframeMemory mem(yml);
    std::vector<int32_t> res;
    std::shared_ptr<Frame> last;
    for (int k = 0; k < M; k++) {
        auto fr = std::make_shared<Frame>();
        fr->_id = k; fr->_n_keypts = 100;
        fr->_pose_in_model = from_rowmajor(&P[16 * (size_t)k]);
        if (k < M - 1) res.push_back(mem.checkAndAddKeyframe(fr) ? 1 : 0);
        last = fr;
    }
    res.push_back(-1);
    for (const auto &fr : mem.selectKeyFramesForBA(last)) res.push_back(fr->_id);
    std::ofstream o(out, std::ios::binary);
    o.write(reinterpret_cast<const char *>(res.data()), sizeof(int32_t) * res.size());
    return 0;
}

static int run_kabsch(const char *in, const char *out)
{
    std::ifstream f(in, std::ios::binary);
    int32_t n_sets;
    rd(f, &n_sets, 1);
    std::vector<float> res;
    for (int s = 0; s < n_sets; s++) {
        int32_t n;
        rd(f, &n, 1);
        std::vector<float> a(3 * (size_t)n), b(3 * (size_t)n);
        rd(f, a.data(), a.size()); rd(f, b.data(), b.size());
        Matrix4f pose;
        solveRigidTransformBetweenPoints(a, b, pose);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) res.push_back(pose(r, c));
    }
    std::ofstream o(out, std::ios::binary);
    o.write(reinterpret_cast<const char *>(res.data()), sizeof(float) * res.size());
    return 0;
}

static int run_posetxt(const char *in, const char *out)
{
    std::ifstream f(in, std::ios::binary);
    int32_t n;
    rd(f, &n, 1);
    std::vector<float> P(16 * (size_t)n);
    rd(f, P.data(), P.size());
    std::ofstream o(out);
    for (int k = 0; k < n; k++) o << formatPoseTxt(from_rowmajor(&P[16 * (size_t)k]));
    return 0;
}

static int run_problem(const char *in, const char *out)
{
    const ProblemDump pb = loadProblem(in);
    long valid = 0;
    for (float d : pb.depth) valid += d >= 0.1f;
    int longest = 0;
    for (int m : pb.n_match_per_pair) longest = std::max(longest, m);
    std::printf("problem: %d frames %dx%d, %zu correspondences (longest pair segment %d), %ld valid depth pixels, ground truth %s\n",
                pb.n_frames, pb.W, pb.H, pb.corr.size(), longest, valid, pb.poses_gt.empty() ? "no" : "yes");
    saveProblem(out, pb);
    return 0;
}

// ---- a whole tracked sequence through btba::Bundler ---------------------------------------------------------------
// scenario file (written by tests/test_cpp_bundler.py): int32 n_frames, H, W, with_images, window_size, max_BA_frames,
// min_fm_edges_newframe, use_ransac; float K[9]; float keyframe_min_rot; per frame: int32 fail, int32 n_keypts, float pose[16]
// (row-major, used for the first frame only), and with_images: depth[H W], normals[H W 4]; then int32 n_tables and per table
// int32 seqA, seqB (A newer), n, float ptA_cam[3 n], ptB_cam[3 n].
struct TableFeatureManager : FeatureManager {
    std::map<std::pair<int, int>, Correspondences> table;           // keyed by SEQUENCE index (Frame::_id_str): ids are re-assigned by Bundler
    btba_workspace *ws = nullptr;                                   // non-null: RANSAC after matching, like SiftManager::findCorres (:191)
    int max_iter = 2000;
    float inlier_dist = 0.01f;
    void findCorres(const std::shared_ptr<Frame> &frameA, const std::shared_ptr<Frame> &frameB) override
    {
        const std::pair<int, int> key{ frameA->_id, frameB->_id };
        if (_matches.count(key)) return;
        const auto it = table.find({ std::stoi(frameA->_id_str), std::stoi(frameB->_id_str) });
        _matches[key] = it == table.end() ? Correspondences{} : it->second;
        if (ws) runRansacMultiPairGPU(ws, { { frameA, frameB } }, max_iter, inlier_dist);
    }
};

static int run_bundler(const char *in, const char *out)
{
    std::ifstream f(in, std::ios::binary);
    int32_t hdr[8];
    rd(f, hdr, 8);
    const int N = hdr[0], H = hdr[1], W = hdr[2], with_images = hdr[3], use_ransac = hdr[7];
    float Krm[9], min_rot;
    rd(f, Krm, 9);
    rd(f, &min_rot, 1);
    auto yml = std::make_shared<Config>();
    yml->window_size = hdr[4]; yml->max_BA_frames = hdr[5]; yml->min_fm_edges_newframe = hdr[6]; yml->keyframe_min_rot = min_rot;
    Matrix3f K;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) K(r, c) = Krm[3 * r + c];
    std::vector<std::shared_ptr<Frame>> seq(N);
    std::vector<float> buf((size_t)H * W * 4);
    for (int k = 0; k < N; k++) {
        auto fr = std::make_shared<Frame>();
        int32_t fl[2];
        float P[16];
        rd(f, fl, 2);
        rd(f, P, 16);
        fr->_id_str = std::to_string(k);
        fr->_status = fl[0] == 1 ? Frame::FAIL : Frame::OTHER;
        if (fl[0] == 2) { fr->_roi[0] = 10.0f; fr->_roi[1] = 14.0f; fr->_roi[2] = 10.0f; fr->_roi[3] = 200.0f; }      // an empty segmentation roi
        fr->_n_keypts = fl[1];
        fr->_pose_in_model = from_rowmajor(P);
        if (with_images) {
            rd(f, buf.data(), (size_t)H * W);
            HIP_OK(hipMalloc(reinterpret_cast<void **>(&fr->_depth_gpu), sizeof(float) * H * W));
            HIP_OK(hipMemcpy(fr->_depth_gpu, buf.data(), sizeof(float) * H * W, hipMemcpyHostToDevice));
            rd(f, buf.data(), (size_t)H * W * 4);
            HIP_OK(hipMalloc(reinterpret_cast<void **>(&fr->_normal_gpu), sizeof(float) * 4 * H * W));
            HIP_OK(hipMemcpy(fr->_normal_gpu, buf.data(), sizeof(float) * 4 * H * W, hipMemcpyHostToDevice));
        }
        seq[k] = fr;
    }
    auto fm = std::make_shared<TableFeatureManager>();
    int32_t n_tables;
    rd(f, &n_tables, 1);
    for (int t = 0; t < n_tables; t++) {
        int32_t h3[3];
        rd(f, h3, 3);
        Correspondences c;
        c.ptA_cam.resize(3 * (size_t)h3[2]); c.ptB_cam.resize(3 * (size_t)h3[2]);
        rd(f, c.ptA_cam.data(), c.ptA_cam.size());
        rd(f, c.ptB_cam.data(), c.ptB_cam.size());
        fm->table[{ h3[0], h3[1] }] = std::move(c);
    }
    btba_workspace *ransac_ws = nullptr;
    if (use_ransac) {
        if (btba_workspace_create(&ransac_ws, nullptr) != BTBA_OK) return 3;
        fm->ws = ransac_ws;
    }
    // without images the optimiser is a CPU stand-in that moves frame i of the window by i * 2^-10 m along x (exact in fp32, and
    // the same in tests/test_cpp_bundler.py): the test is about the control flow around the call
    Bundler::OptimizeFn mock = [](const std::vector<EntryJ> &, const std::vector<int> &, int n, int, int, const std::vector<float *> &,
                                  const std::vector<uchar4 *> &, const std::vector<float4 *> &, std::vector<Matrix4f> &poses, const Matrix3f &) {
        for (int i = 1; i < n; i++) poses[i](0, 3) += 0.0009765625f * (float)i;
    };
    Bundler bundler(yml, fm, K, H, W, with_images ? Bundler::OptimizeFn{} : mock);
    std::FILE *o = std::fopen(out, "w");
    if (!o) return 2;
    for (int k = 0; k < N; k++) {
        const auto &fr = seq[k];
        bundler.processNewFrame(fr);
        // one line per frame: what was decided, then the frame's pose
        std::fprintf(o, "%d %d %d %d %d", k, fr->_id, (int)fr->_status, bundler._need_reinit ? 1 : 0, bundler.n_ba_calls);
        const bool ran = fr->_status != Frame::FAIL && fr->_id >= 1;
        std::fprintf(o, " %zu", ran ? bundler.last_window.frames.size() : (size_t)0);
        if (ran) for (const auto &lf : bundler.last_window.frames) std::fprintf(o, " %d", lf->_id);
        std::fprintf(o, " %zu %d %d", ran ? bundler.last_window.global_corres.size() : (size_t)0, ran ? bundler.last_window.n_edges_newframe : 0, ran && bundler.last_window.run_ba ? 1 : 0);
        std::fprintf(o, " %zu", bundler.keyframes().size());
        for (const auto &kf : bundler.keyframes()) std::fprintf(o, " %d", kf->_id);
        std::fprintf(o, " %zu", bundler._frames.size());
        std::fprintf(o, " %zu", fm->_matches.size());
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) std::fprintf(o, " %.9g", (double)fr->_pose_in_model(r, c));
        std::fprintf(o, "\n");
    }
    std::fclose(o);
    if (ransac_ws) btba_workspace_destroy(ransac_ws);
    return 0;
}

int main(int argc, char **argv)
{
    try {
        if (argc == 4 && !std::strcmp(argv[1], "ba")) return run_ba(argv[2], argv[3]);
        if (argc == 4 && !std::strcmp(argv[1], "keyframes")) return run_keyframes(argv[2], argv[3]);
        if (argc == 4 && !std::strcmp(argv[1], "problem")) return run_problem(argv[2], argv[3]);
        if (argc == 4 && !std::strcmp(argv[1], "kabsch")) return run_kabsch(argv[2], argv[3]);
        if (argc == 4 && !std::strcmp(argv[1], "posetxt")) return run_posetxt(argv[2], argv[3]);
        if (argc == 4 && !std::strcmp(argv[1], "bundler")) return run_bundler(argv[2], argv[3]);
    } catch (const std::exception &e) { std::fprintf(stderr, "host_driver: %s\n", e.what()); return 2; }
    std::fprintf(stderr, "usage: host_driver ba|keyframes|problem|kabsch|posetxt|bundler <in> <out>\n");
    return 1;
}
