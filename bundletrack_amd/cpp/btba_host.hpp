// btba_host.hpp -- the host side above the C ABI, in C++ like the reference's own host code.
//
// Mirrors, name for name and argument for argument, what sits on either side of the optimiser boundary in
// wenbowen123/BundleTrack:
//   OptimizerGpu::optimizeFrames                        src/cuda/LossGPU.h:40-52, LossGPU.cu:53-139
//   Bundler::optimizeGPU's marshalling                  src/Bundler.cpp:286-347   (marshalWindow)
//   Bundler::checkAndAddKeyframe / selectKeyFramesForBA src/Bundler.cpp:185-274   (KeyframeMemory)
//   Bundler::processNewFrame / optimizeGPU / saveNewframeResult   src/Bundler.cpp:56-183, 279-359, 362-377   (Bundler)
//   SiftManager::forgetFrame / procrustesByCorrespondence / runRansacMultiPairGPU   src/FeatureManager.cpp:142-170, 523-556, 659-741
//                                                        (FeatureManager: the slice of SiftManager that Bundler calls)
//   Utils::rotationGeodesicDistance                     src/Utils.cpp:42-47
//   Utils::solveRigidTransformBetweenPoints             src/Utils.cpp:180-214     (Kabsch; a 3x3 one-sided Jacobi SVD stands in
//                                                        for Eigen::JacobiSVD)
// The reference builds these on Eigen, yaml-cpp and PCL, none of which exist in this image, so the two value types
// the interface needs are defined here with Eigen's conventions (column-major storage, (row, col) access): a
// maintainer swaps `btba::Matrix4f` for `Eigen::Matrix4f` and `btba::Config` for the YAML node and nothing else
// changes.  Only libbtba.so's C ABI (include/btba.h) is called: plain host C++ (g++), no
// device code, no torch; HIP contributes the float4 / uchar4 pixel types only.
#pragma once
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <hip/hip_vector_types.h>     // float4, uchar4: the device pixel types of the reference's signatures (plain structs on the host)

#include "../../include/btba.h"

namespace btba {

struct Matrix4f {                      // Eigen::Matrix4f: column-major
    float d[16];
    float &operator()(int r, int c) { return d[c * 4 + r]; }
    float operator()(int r, int c) const { return d[c * 4 + r]; }
    static Matrix4f Identity() { Matrix4f M{}; for (int k = 0; k < 4; k++) M(k, k) = 1.0f; return M; }
};
struct Matrix3f {                      // Eigen::Matrix3f: column-major
    float d[9];
    float &operator()(int r, int c) { return d[c * 3 + r]; }
    float operator()(int r, int c) const { return d[c * 3 + r]; }
};

using EntryJ = btba_entryj;            // src/cuda/SIFTImageManager.h:44-59, 32 bytes

// The keys of config_ycbineoat.yml the path reads (at their shipping values): bundle.* :23-36, p2p.* :63-65
struct Config {
    int num_iter_outter = 7, num_iter_inner = 5;
    float robust_delta = 0.005f, image_downscale = 4.0f;
    float p2p_max_dist = 0.02f, p2p_max_normal_angle = 45.0f;
    int max_BA_frames = 15, min_fm_edges_newframe = 5;
    float keyframe_min_rot = 10.0f;
    int keyframe_min_feat_num = 0;
    int window_size = 2;                          // bundle.window_size :26 ("exclude keyframes, include new frame")
    int ransac_max_iter = 2000;                   // ransac.* :53-56
    float ransac_inlier_dist = 0.01f;
    std::string pose_dir;                         // debug_dir + "/poses/" (:5; Bundler.cpp:366); empty = do not write pose files
};

class Error : public std::runtime_error {       // the reference exits / spins instead (cutil_inline_runtime.h:261-269)
public:
    int status;
    Error(int status_, const std::string &where) : std::runtime_error(where + ": " + btba_strerror(status_)), status(status_) {}
};

class OptimizerGpu {
public:
    std::shared_ptr<Config> yml;
    btba_stats last_stats{};
    bool persistent_frame_cache = false;         // hand frame ids to btba_optimize_frames_keyed (needs frame_ids below)
    std::vector<uint64_t> frame_ids;             // Frame::_id of every window frame, when persistent_frame_cache is set
    bool keyed_correspondences = false;          // with persistent_frame_cache: pair segments stay on the device too (BTBA_FLAG_KEYED_CORR)

    // The reference runs the whole path on the legacy NULL stream (no explicit streams anywhere in src/cuda), so the drop-in
    // does too: depth / normal maps produced by earlier default-stream work are ordered before the cache build.  A caller with
    // its own non-blocking producer stream passes it (the workspace then runs there) or orders with workspace().
    explicit OptimizerGpu(std::shared_ptr<Config> yml1, void *hip_stream = nullptr);
    ~OptimizerGpu();
    btba_workspace *workspace() const { return ws_; }      // btba_workspace_wait_stream / _signal_stream / _frame_cache_evict
    OptimizerGpu(const OptimizerGpu &) = delete;
    OptimizerGpu &operator=(const OptimizerGpu &) = delete;

    // LossGPU.h:50.  poses: camera -> model, updated in place; colors_gpu ignored (weight 0, SBA.cpp:32).
    void optimizeFrames(const std::vector<EntryJ> &global_corres, const std::vector<int> &n_match_per_pair, int n_frames, int H, int W,
                        const std::vector<float *> &depths_gpu, const std::vector<uchar4 *> &colors_gpu, const std::vector<float4 *> &normals_gpu,
                        std::vector<Matrix4f> &poses, const Matrix3f &K);

private:
    btba_workspace *ws_ = nullptr;               // grow-only scratch kept across calls (the reference reallocates everything)
};

// ---- the caller's side -------------------------------------------------------------------------------------
struct Frame {                                   // the fields of Frame (src/Frame.h:45-96) the BA caller touches
    enum Status { FAIL, NO_BA, OTHER };
    int _id = 0;
    std::string _id_str;                         // names the pose file (Bundler.cpp:374)
    Status _status = OTHER;
    Matrix4f _pose_in_model = Matrix4f::Identity();
    int _n_keypts = 0;
    float _roi[4] = { 0.0f, 1e9f, 0.0f, 1e9f };  // (umin, umax, vmin, vmax) of the segmentation mask (Frame.h:81): a frame whose roi is under 10 px wide or high is FAIL (Bundler.cpp:88-93)
    float *_depth_gpu = nullptr;
    float4 *_normal_gpu = nullptr;
    uchar4 *_color_gpu = nullptr;
};

// Utils::solveRigidTransformBetweenPoints (Utils.cpp:180-214): the rigid transform points1 -> points2 (n x 3 each, xyz
// triples), identity when fewer than 3 points, a non-orthonormal V U^T or a non-finite result.
void solveRigidTransformBetweenPoints(const std::vector<float> &points1, const std::vector<float> &points2, Matrix4f &pose);

// `ff << std::setprecision(10) << ob_in_cam << std::endl` (Bundler.cpp:372-377) with Eigen's default IOFormat: %.10g
// coefficients, right-aligned to the widest one, one space between columns, one row per line.
std::string formatPoseTxt(const Matrix4f &ob_in_cam);

float rotationGeodesicDistance(const Matrix4f &A, const Matrix4f &B);           // rotation blocks only, radians (Utils.cpp:42-47)

struct Correspondences { std::vector<float> ptA_cam, ptB_cam; };                // xyz triples; A = the newer frame

struct Window {                                                                  // what optimizeGPU hands to optimizeFrames
    std::vector<std::shared_ptr<Frame>> frames;                                  // sorted by id: index 0 is never moved
    std::vector<EntryJ> global_corres;
    std::vector<int> n_match_per_pair;
    int n_edges_newframe = 0;
    bool run_ba = false;                                                         // false <=> newframe->_status = NO_BA (:343-347)
};
// Bundler.cpp:286-347.  matches: keyed by (newer frame id, older frame id) like _fm->_matches[{frameA, frameB}].
Window marshalWindow(std::vector<std::shared_ptr<Frame>> local_frames, const std::map<std::pair<int, int>, Correspondences> &matches,
                     const std::shared_ptr<Frame> &newframe, int min_fm_edges_newframe);

class KeyframeMemory {
public:
    std::vector<std::shared_ptr<Frame>> _keyframes;
    explicit KeyframeMemory(std::shared_ptr<Config> yml1) : yml(std::move(yml1)) {}
    bool checkAndAddKeyframe(const std::shared_ptr<Frame> &frame);                                     // Bundler.cpp:185-219
    std::vector<std::shared_ptr<Frame>> selectKeyFramesForBA(const std::shared_ptr<Frame> &newframe);  // :222-274, sorted by id
private:
    std::shared_ptr<Config> yml;
};

// The slice of SiftManager (src/FeatureManager.h:86-130) that Bundler calls.  Feature detection and matching themselves are out of
// scope (LF-Net over zmq, OpenCV): findCorres is the hook a tracker fills in; what the reference does with
// the matches afterwards is implemented here.
class FeatureManager {
public:
    std::map<std::pair<int, int>, Correspondences> _matches;        // _matches[{frameA, frameB}], keyed (newer id, older id)
    virtual ~FeatureManager() = default;
    virtual void detectFeature(const std::shared_ptr<Frame> & /*frame*/) {}
    // fills _matches[{frameA->_id, frameB->_id}] unless present (:176); may mark frameA FAIL
    virtual void findCorres(const std::shared_ptr<Frame> &frameA, const std::shared_ptr<Frame> &frameB) = 0;
    virtual void forgetFrame(const std::shared_ptr<Frame> &frame);                                                   // :142-170
    int countInlierCorres(const std::shared_ptr<Frame> &frameA, const std::shared_ptr<Frame> &frameB) const;          // :746-758
    // :523-556: Kabsch of the matches moved into the model frame with the frames' current poses; identity below 5 matches
    virtual Matrix4f procrustesByCorrespondence(const std::shared_ptr<Frame> &frameA, const std::shared_ptr<Frame> &frameB);
    // :659-741 on btba_ransac_pairs (one call for all pairs; the reference's cuRAND triples and procrustesKernel hypotheses):
    // every pair's matches are replaced by their RANSAC inliers, or emptied when fewer than 5 survive.  Needs the GPU.
    void runRansacMultiPairGPU(btba_workspace *ws, const std::vector<std::pair<std::shared_ptr<Frame>, std::shared_ptr<Frame>>> &pairs,
                               int max_iter, float inlier_dist);
};

// Bundler (src/Bundler.h, Bundler.cpp:56-377) from the point where a frame has its mask, depth and normals on the device:
// pose initialisation from the previous frame, the sliding window, the keyframe subset, bundle adjustment, keyframe insertion,
// the pose file.  `optimize` defaults to one persistent OptimizerGpu (the reference constructs a new one per call, :349);
// tests inject a CPU stand-in with the same signature.
class Bundler {
public:
    using OptimizeFn = std::function<void(const std::vector<EntryJ> &, const std::vector<int> &, int, int, int, const std::vector<float *> &,
                                          const std::vector<uchar4 *> &, const std::vector<float4 *> &, std::vector<Matrix4f> &, const Matrix3f &)>;
    std::shared_ptr<Config> yml;
    std::shared_ptr<FeatureManager> _fm;
    std::deque<std::shared_ptr<Frame>> _frames;
    std::vector<std::shared_ptr<Frame>> _local_frames;
    std::shared_ptr<Frame> _newframe;
    bool _need_reinit = false;
    KeyframeMemory memory;                                          // _keyframes + checkAndAddKeyframe + selectKeyFramesForBA
    Matrix3f K{};
    int H = 0, W = 0;
    int n_ba_calls = 0;
    Window last_window;                                             // what the last optimizeGPU marshalled

    Bundler(std::shared_ptr<Config> yml1, std::shared_ptr<FeatureManager> fm, const Matrix3f &K1, int H1, int W1, OptimizeFn optimize = {});
    void processNewFrame(std::shared_ptr<Frame> frame);             // :56-183
    void optimizeGPU();                                             // :279-359
    void saveNewframeResult();                                      // :362-377: <pose_dir>/<_id_str>.txt = inverse(pose_in_model), 10 digits
    const std::vector<std::shared_ptr<Frame>> &keyframes() const { return memory._keyframes; }

private:
    OptimizeFn optimize_;
    std::unique_ptr<OptimizerGpu> own_opt_;                         // created at the first BA call when no OptimizeFn was given
};

// Eigen's inverse() for a general 4x4, computed in double (used for the pose files)
Matrix4f inverse(const Matrix4f &M);

// One bundle-adjustment call as a file (the layout is documented in bundletrack_amd/problem_io.py, which writes and reads
// the same bytes): what Bundler::optimizeGPU hands to OptimizerGpu::optimizeFrames, with the frames on the HOST -- the
// caller uploads them (hipMalloc + hipMemcpy, as Frame's constructor does, src/Frame.cpp:68-70,107-149).
struct ProblemDump {
    int n_frames = 0, H = 0, W = 0;
    float image_downscale = 4.0f;
    float K[9] = {};                                  // row-major
    std::vector<EntryJ> corr;                         // pair-major
    std::vector<int> n_match_per_pair;                // P = n (n - 1) / 2 segment lengths
    std::vector<float> poses_init;                    // n x 16, row-major, camera -> model
    std::vector<double> poses_gt;                     // n x 16 or empty
    std::vector<float> depth;                         // n x H x W
    std::vector<float> normals;                       // n x H x W x 4
};
ProblemDump loadProblem(const std::string &path);                    // throws btba::Error(BTBA_EINVAL) on a malformed file
void saveProblem(const std::string &path, const ProblemDump &pb);

}  // namespace btba
