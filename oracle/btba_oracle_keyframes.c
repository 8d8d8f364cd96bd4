/* btba_oracle_keyframes.c -- CPU restatement of BundleTrack's keyframe memory.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ as the checker of bundletrack_amd/bundler.py::KeyframeMemory and
 * btba::KeyframeMemory (bundletrack_amd/cpp/btba_host.cpp).  Never linked, imported or executed by the product path.
 *
 * PARITY UNPINNED for this file: the functions it follows (Bundler::checkAndAddKeyframe, Bundler::selectKeyFramesForBA,
 * Utils::rotationGeodesicDistance) sit in translation units that need Eigen and yaml-cpp, which this image lacks, so the
 * reference's own code for them cannot be compiled here (a build on stand-in headers is not a reference build), and the
 * reference holds no golden vectors for them.  What follows is written from the reference text, statement by statement:
 *
 *   orc_rotation_geodesic_distance   /root/reference/src/Utils.cpp:42-47
 *   orc_check_and_add_keyframe       /root/reference/src/Bundler.cpp:185-219
 *   orc_select_keyframes_for_ba      /root/reference/src/Bundler.cpp:222-274
 *
 * Poses are row-major 4 x 4 floats (`_pose_in_model`); a frame is its index into the pose array.  The reference keeps the
 * chosen set in a std::set<std::shared_ptr<Frame>>, i.e. ordered by the ADDRESS of the Frame objects: that order decides the
 * order in which cum_dist is summed (a float sum: Bundler.cpp:247-251) and nothing else -- optimizeGPU sorts the result by
 * frame id (Bundler.cpp:286).  `addr_rank` restates it: the rank of every frame's address (NULL: allocation order = index order).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define ORC_API __attribute__((visibility("default")))

/* Utils.cpp:42-47.  Eigen evaluates (R1 * R2.transpose()).trace() as the sum over i of the dot product of row i of R1 with row i
 * of R2, each dot product and the trace summed left to right in float; `/ 2.0` is a double division of a float, rounded back
 * into the float `tmp` (halving is exact); std::acos(float) is acosf. */
ORC_API float orc_rotation_geodesic_distance(const float *pose1, const float *pose2)
{
    float trace = 0.0f;
    for (int i = 0; i < 3; i++) {
        float dot = pose1[4 * i + 0] * pose2[4 * i + 0];
        dot += pose1[4 * i + 1] * pose2[4 * i + 1];
        dot += pose1[4 * i + 2] * pose2[4 * i + 2];
        trace = (i == 0) ? dot : trace + dot;
    }
    float tmp = (float)((double)(trace - 1.0f) / 2.0);
    tmp = fmaxf(fminf(1.0f, tmp), -1.0f);          /* std::max(std::min(1.0f, tmp), -1.0f): NaN never reaches here from finite poses */
    return acosf(tmp);
}

/* Bundler.cpp:185-219.  keyframes[0 .. *n_keyframes): indices into `poses`, appended to on success.  status_other: frame->_status ==
 * Frame::OTHER.  Returns 1 when the frame was appended.  (trans_diff, :207, is computed by the reference and never used.) */
ORC_API int orc_check_and_add_keyframe(const float *poses, int frame, int frame_id, int status_other, int n_keypts, int min_feat_num, float min_rot_deg,
                                       int32_t *keyframes, int32_t *n_keyframes)
{
    if (frame_id == 0) { keyframes[(*n_keyframes)++] = frame; return 1; }          /* :187-191 */
    if (!status_other) return 0;                                                    /* :192 */
    if (n_keypts < min_feat_num) return 0;                                          /* :199-202 */
    for (int i = 0; i < *n_keyframes; i++) {                                        /* :204-215 */
        float rot_diff = orc_rotation_geodesic_distance(poses + 16 * (size_t)frame, poses + 16 * (size_t)keyframes[i]);
        rot_diff = (float)((double)(rot_diff * 180) / M_PI);                        /* rot_diff*180/M_PI: float * int -> float, float / double -> double, stored as float */
        if (rot_diff < min_rot_deg) return 0;
    }
    keyframes[(*n_keyframes)++] = frame;                                            /* :218 */
    return 1;
}

typedef struct { int32_t frame; int32_t rank; } orc_set_entry;

/* insert into the address-ordered set (std::set::insert: no duplicates) */
static int set_insert(orc_set_entry *set, int n, int32_t frame, int32_t rank)
{
    int pos = 0;
    for (; pos < n; pos++) {
        if (set[pos].frame == frame) return n;
        if (set[pos].rank > rank) break;
    }
    for (int k = n; k > pos; k--) set[k] = set[k - 1];
    set[pos].frame = frame; set[pos].rank = rank;
    return n + 1;
}
static int set_has(const orc_set_entry *set, int n, int32_t frame)
{
    for (int k = 0; k < n; k++) if (set[k].frame == frame) return 1;
    return 0;
}

/* Bundler.cpp:222-274, method "greedy_rot".  newframe / keyframes: indices into `poses`; addr_rank[frame]: the frame's place in address
 * order (NULL: index order).  chosen_out receives the chosen frames in SET order (what `_local_frames` holds at :273; the caller sorts by id,
 * :286); returns their number.  Only when the pool does not fit: keyframe 0 joins first (:237), then the keyframe with the SMALLEST summed
 * distance to the chosen set, strict `<` against FLT_MAX, first keyframe in pool order on ties (:241-262). */
ORC_API int orc_select_keyframes_for_ba(const float *poses, int newframe, const int32_t *keyframes, int n_keyframes, int max_BA_frames,
                                        const int32_t *addr_rank, int32_t *chosen_out)
{
    orc_set_entry *set = (orc_set_entry *)malloc(sizeof(orc_set_entry) * (size_t)(n_keyframes + 2));
    int n = 0;
#define RANK(f) (addr_rank ? addr_rank[f] : (int32_t)(f))
    n = set_insert(set, n, newframe, RANK(newframe));                              /* :224 */
    if (n_keyframes + n <= max_BA_frames) {                                         /* :227-235 */
        for (int k = 0; k < n_keyframes; k++) n = set_insert(set, n, keyframes[k], RANK(keyframes[k]));
    } else {
        n = set_insert(set, n, keyframes[0], RANK(keyframes[0]));                   /* :237 */
        while (n < max_BA_frames) {                                                 /* :243 */
            float best_dist = FLT_MAX;
            int32_t best_kf = -1;
            for (int i = 0; i < n_keyframes; i++) {
                const int32_t kf = keyframes[i];
                if (set_has(set, n, kf)) continue;                                  /* :250 */
                float cum_dist = 0;
                for (int s = 0; s < n; s++)                                         /* :252-256: set order */
                    cum_dist += orc_rotation_geodesic_distance(poses + 16 * (size_t)kf, poses + 16 * (size_t)set[s].frame);
                if (cum_dist < best_dist) { best_dist = cum_dist; best_kf = kf; }   /* :257-261 */
            }
            if (best_kf < 0) break;          /* (the reference would insert a null pointer and stop growing: the set is full of every keyframe) */
            n = set_insert(set, n, best_kf, RANK(best_kf));                         /* :263 */
        }
    }
#undef RANK
    for (int k = 0; k < n; k++) chosen_out[k] = set[k].frame;
    free(set);
    return n;
}
