/* vio_posegraph.h -- C ABI of the loop-closure slice of pose_graph.
 *
 * The reference's pose_graph nodelet (pose_graph/src) builds a KeyFrame per estimator keyframe, asks DBoW2 for a loop candidate
 * (PoseGraph::detectLoop, pose_graph.cpp:308 = vio_pg_detect_loop below; the vocabulary blob is missing from the reference tree, the caller
 * supplies the file), verifies it (KeyFrame::findConnection) and, on success, hands the estimator the match list (Estimator::setReloFrame =
 * vio_set_relo_frame in vio_abi.h) and runs the 4-DoF pose-graph optimisation.  Each entry point names the reference interface it replaces.
 * Descriptor extraction and matching run as HIP kernels on the current device (no CPU fallback: VIO_EDEVICE without a GPU); the geometric
 * verification and the pose-graph optimisation are per-keyframe host code, like the reference's.  Plain C types, caller-owned buffers.
 */
#ifndef VIO_POSEGRAPH_H
#define VIO_POSEGRAPH_H
#include <stdint.h>

#include "vio_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* KeyFrame::computeWindowBRIEFPoint + KeyFrame::computeBRIEFPoint (pose_graph/src/keyframe/keyframe.cpp:80-124) with DVision::BRIEF::compute
 * (ThirdParty/DVision/BRIEF.cpp): GaussianBlur(9x9, sigma 2) of `gray` (cfg->height x cfg->width, u8, host), 256-bit BRIEF descriptors of the
 * n_win window points win_uv[n_win][2] (pixel coordinates of the tracked features) into win_desc[n_win][4]; cv::FAST(image, fast_threshold,
 * NMS) keypoints of the raw image in row-major order into kp_xy[cap][2], their descriptors into kp_desc[cap][4] and their normalised
 * coordinates (m_camera->liftProjective) into kp_norm[cap][2].  pattern1024 = x1[256] y1[256] x2[256] y2[256] of BRIEF_PATTERN_FILE
 * (support_files/brief_pattern.yml).  Returns the number of keypoints found (may exceed cap: only cap are written) or a negative status. */
int vio_pg_describe(const vio_config *cfg, const uint8_t *gray, int n_win, const float *win_uv, const int32_t *pattern1024, int fast_threshold,
                    uint64_t *win_desc, int cap, float *kp_xy, uint64_t *kp_desc, float *kp_norm);

/* KeyFrame::searchByBRIEFDes / searchInAera / HammingDis (keyframe.cpp:126-169, 530): for each of the n window descriptors the FIRST old
 * descriptor with the smallest Hamming distance below 128; best_index[i] = its index if the distance is below 80, else -1. */
int vio_pg_match(const uint64_t *win_desc, int n, const uint64_t *old_desc, int m, int32_t *best_index, int32_t *best_dist);

/* KeyFrame::findConnection (keyframe.cpp:252-528) after the descriptor search, incl. KeyFrame::PnPRANSAC (:195-250).  Current keyframe: n
 * window points -- world point pt3d[n][3], feature id pt_id[n], match[n] from vio_pg_match against the old keyframe -- and its origin_vio
 * pose (vio_T[3], vio_R[9] row-major); old_norm[m][2] = the old keyframe's normalised keypoints; qic[9], tic[3] = the camera extrinsic.
 * Returns 1 if more than min_loop_num (MIN_LOOP_NUM = 25) matches survive the PnP RANSAC and |relative yaw| < 30 deg, |relative t| < 20 m:
 * loop_info[8] = relative_t(3), relative_q(w, x, y, z), relative_yaw (deg); match_points[n_match][3] = (x, y of the old normalised keypoint,
 * feature id) in ascending window order -- the list pose_graph publishes for Estimator::setReloFrame (:491-520; pass it with the OLD keyframe's
 * pose to vio_set_relo_frame).  0 = no loop; negative = status.  pnp_T[3], pnp_R[9]: PnP_T_old / PnP_R_old (diagnostics, may be NULL). */
int vio_pg_find_connection(int n, const float *pt3d, const double *pt_id, const int32_t *match, const float *old_norm, const double *vio_T,
                           const double *vio_R, const double *qic, const double *tic, int min_loop_num, double *loop_info, double *match_points,
                           int32_t *n_match, double *pnp_T, double *pnp_R);

/* PoseGraph::optimize4DoF (pose_graph/src/pose_graph/pose_graph.cpp:410-581; residuals pose_graph.h:102-256) over the n keyframes from the
 * earliest looped one to the current one, in list order: t[n][3], R[n][9] = their VIO poses, sequence[n], loop_to[n] = local index of the loop
 * partner or -1, loop_info[n][8].  Node 0 and the nodes of sequence 0 are held constant; sequential edges to the 1..4 previous keyframes of the
 * same sequence, loop edges under HuberLoss(0.1), yaw + translation free, five Levenberg-Marquardt iterations.  Writes the optimised poses
 * t_out[n][3], R_out[n][9] and drift[4] = (yaw_drift deg, t_drift(3)) of the newest keyframe (:547-553). */
int vio_pg_optimize4dof(int n, const double *t, const double *R, const int32_t *sequence, const int32_t *loop_to, const double *loop_info,
                        double *t_out, double *R_out, double *drift);

/* PoseGraph::optimize6DoF (pose_graph.cpp:583-740; RelativeRTError pose_graph.h:256-320), the `imu: 0` variant: full poses (quaternion with
 * ceres::QuaternionParameterization + translation), same node / edge selection as optimize4DoF, edges = RelativeRTError(relative t, relative q,
 * t_var 0.1, q_var 0.01) with the loop edges (loop_info[0..2] = relative t, [3..6] = relative q (w, x, y, z)) under HuberLoss(0.1), five
 * Levenberg-Marquardt iterations.  drift12 = r_drift (9, row-major) = R_cur R_vio^T, t_drift (3) = t_cur - r_drift t_vio (:717-721). */
int vio_pg_optimize6dof(int n, const double *t, const double *R, const int32_t *sequence, const int32_t *loop_to, const double *loop_info,
                        double *t_out, double *R_out, double *drift12);

/* ---- place recognition: BriefVocabulary + BriefDatabase of PoseGraph (pose_graph.h:83-84; vendored DBoW2 under pose_graph/src/ThirdParty) ----
 * The tree walk of every descriptor runs as a HIP kernel (no CPU fallback); bag-of-words vector, inverted file and L1 query are per-keyframe
 * host code.  Only L1_NORM scoring (what brief_k10L6.bin uses) is supported; all four weightings are.  The reference tree does not contain the
 * vocabulary blob (support_files/brief_k10L6.bin, .MISSING_LARGE_BLOBS): the caller supplies the file. */
typedef struct vio_pg_voc vio_pg_voc;
/* PoseGraph::loadVocabulary (pose_graph.cpp:44-47) = BriefVocabulary(path) -> TemplatedVocabulary::loadBin (TemplatedVocabulary.h:1509-1561) on the
 * file format of VINSLoop::Vocabulary::deserialize (ThirdParty/VocabularyBinary.{hpp,cpp}): 6 x int32 (k, L, scoringType, weightingType, nNodes,
 * nWords), nNodes x {int32 nodeId, int32 parentId, double weight, uint64 descriptor[4]}, nWords x {int32 nodeId, int32 wordId};
 * db.setVocabulary(*voc, false, 0) (no direct index).  NULL on failure (vio_last_error). */
vio_pg_voc *vio_pg_voc_load(const char *path);
/* the same from arrays (tests, vocabularies kept in another container): node i = (node_id[i], parent_id[i], weight[i], desc[i][4]) in file order */
vio_pg_voc *vio_pg_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *node_id, const int32_t *parent_id,
                              const double *weight, const uint64_t *desc, int n_words, const int32_t *word_node, const int32_t *word_id);
void vio_pg_voc_destroy(vio_pg_voc *v);
/* out7 = k, L, scoring, weighting, nodes, words, database entries */
int vio_pg_voc_info(const vio_pg_voc *v, int32_t *out7);
/* TemplatedVocabulary::transform(feature, word_id, weight) (TemplatedVocabulary.h:1217-1260) for n descriptors desc[n][4] */
int vio_pg_voc_transform(vio_pg_voc *v, const uint64_t *desc, int n, int32_t *word_id, double *word_weight);
/* TemplatedVocabulary::transform(features, BowVector) (:1065-1122): ascending word ids and L1-normalised values; returns the vector's size */
int vio_pg_voc_bow(vio_pg_voc *v, const uint64_t *desc, int n, int cap, int32_t *word_id, double *value);
/* BriefDatabase::add (TemplatedDatabase.h:408-475; PoseGraph::addKeyFrameIntoVoc, pose_graph.cpp:395-408): returns the entry id (>= 0) */
int vio_pg_db_add(vio_pg_voc *v, const uint64_t *desc, int n);
/* BriefDatabase::query -> queryL1 (TemplatedDatabase.h): entries with id < max_id (or max_id == -1) and, as upstream, the newest entry; best
 * first, at most max_results (> 0), scores in [0, 1].  Ties are ordered by entry id (upstream: std::sort, unspecified).  Returns the count. */
int vio_pg_db_query(vio_pg_voc *v, const uint64_t *desc, int n, int max_results, int max_id, int32_t *ids, double *scores);
/* PoseGraph::detectLoop (pose_graph.cpp:308-393) for the keyframe's FAST-keypoint descriptors (kp_desc of vio_pg_describe): query(4,
 * frame_index - 50), add, then the score gates (best > 0.05 and another > 0.015, frame_index > 50) and the smallest candidate id.
 * *loop_index = that id or -1; ids4 / scores4 / n_ret (each may be NULL) = the query's results.  Returns a status. */
int vio_pg_detect_loop(vio_pg_voc *v, const uint64_t *desc, int n, int frame_index, int32_t *loop_index, int32_t *ids4, double *scores4,
                       int32_t *n_ret);

/* test entry: the blurred image alone */
int vio_pg_stage_blur(const uint8_t *gray, int width, int height, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
