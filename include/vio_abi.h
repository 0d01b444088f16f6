/* MI355X-native VIO hot path — C ABI (the drop-in boundary).
 *
 * The reference has no FFI layer: its boundary is the public member-function surface of FeatureTracker / Estimator /
 * FeatureManager plus mutable globals (vins_estimator/src/utility/parameters.h:17-79).  Each entry point below names
 * the reference interface it replaces.  A handle owns a *batch* of S independent sequences (S = 1 reproduces the
 * reference's one-Estimator-per-process use); all state lives in HBM behind the handle, configuration is per handle
 * instead of process-global.  Plain C types only; the caller owns input buffers for the duration of a call.
 *
 * Threading: a handle is not re-entrant.  vio_push_imu may be called from another thread than vio_track/vio_process
 * (internal lock), mirroring Estimator::inputIMU being called from ROS callback threads (estimator.cpp:1749-1766).
 */
#ifndef VIO_ABI_H
#define VIO_ABI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* error conventions (reference: void everywhere, ROS_ASSERT aborts, busy-wait for IMU at estimator.cpp:178-183) */
enum {
    VIO_OK = 0,
    VIO_NEED_IMU = 1,   /* IMU not yet pushed through stamp + td; nothing was consumed, call again later */
    VIO_REBOOTED = 2,   /* failureDetection() fired (estimator.cpp:345-353); the sequence was reset */
    VIO_EINVAL = -1,
    VIO_EDEVICE = -2,   /* HIP error / no GPU: the product path never falls back to the CPU */
    VIO_ECAPACITY = -3
};

/* Replaces the globals read by the hot path (parameters.h:11-79) and config/realsense/vio.yaml keys. */
typedef struct vio_config {
    int32_t width, height;          /* COL, ROW                      image_width / image_height */
    int32_t max_cnt, min_dist;      /* MAX_CNT, MIN_DIST             max_cnt / min_dist */
    int32_t grid_rows, grid_cols;   /* NUM_GRID_ROWS / NUM_GRID_COLS */
    int32_t window_size;            /* WINDOW_SIZE (compile-time 10 upstream, parameters.h:12); 2..20 here */
    int32_t max_landmarks;          /* NUM_OF_F (parameters.h:14): capacity of the landmark table */
    int32_t fix_depth;              /* FIX_DEPTH */
    int32_t estimate_extrinsic;     /* ESTIMATE_EXTRINSIC: 0 or 1 (2 = calibrate from scratch is out of scope) */
    int32_t estimate_td;            /* ESTIMATE_TD */
    int32_t max_iterations;         /* NUM_ITERATIONS  max_num_iterations */
    int32_t ransac_max_iters;       /* cv::findFundamentalMat RANSAC iteration cap (1000) */
    int32_t lk_max_level;           /* maxLevel of calcOpticalFlowPyrLK: 1 for the IMU-aided call (feature_tracker.cpp:303) */
    int32_t reserved0;
    double fx, fy, cx, cy, k1, k2, p1, p2; /* pinhole projection_parameters / distortion_parameters */
    double focal_length;            /* FOCAL_LENGTH = 460 (parameters.h:11) */
    double f_threshold;             /* F_THRESHOLD */
    double depth_min, depth_max;    /* DEPTH_MIN_DIST / DEPTH_MAX_DIST */
    double acc_n, acc_w, gyr_n, gyr_w, g_norm;
    double ric[9];                  /* extrinsicRotation, row-major, imu<-cam */
    double tic[3];                  /* extrinsicTranslation */
    double td, tr;                  /* TD, TR */
    double min_parallax_px;         /* keyframe_parallax */
    double init_depth;              /* INIT_DEPTH = 5 (parameters.cpp:215) */
} vio_config;

typedef struct vio_batch vio_batch; /* opaque */

/* parameters.cpp:81-243 readParameters() defaults for config/realsense/vio.yaml at 150 features */
void vio_config_default(vio_config *cfg);

/* Estimator::Estimator + setParameter() (estimator.cpp:9-41) for S sequences on the current HIP device.
 * imu_capacity = ring size per sequence (samples). Returns NULL on failure (see vio_last_error). */
vio_batch *vio_create(const vio_config *cfg, int n_seq, int imu_capacity);
void vio_destroy(vio_batch *h);
const char *vio_last_error(void);
/* Estimator::clearState() + setParameter() for every sequence (estimator_nodelet.cpp:255-258) */
int vio_reset(vio_batch *h);

/* Estimator::inputIMU(t, acc, gyr) (estimator.cpp:1749-1766) for sequence seq; n samples, t strictly increasing. */
int vio_push_imu(vio_batch *h, int seq, int n, const double *t, const double *acc_xyz, const double *gyr_xyz);

/* One camera frame for every sequence: the body of EstimatorNodelet::process_tracker for one synchronised
 * colour+depth pair (estimator_nodelet.cpp:234-393) followed by EstimatorNodelet::process (:462-549):
 *   predictMotion (estimator.cpp:1790-1860) -> FeatureTracker::readImage (feature_tracker.cpp:263-439) -> updateID loop
 *   -> feature-map packaging -> FeatureManager::inputDepth -> Estimator::processImage (estimator.cpp:156-374).
 * gray: S images [S][height][width] u8, depth_mm: [S][height][width] u16 (CV_16UC1 millimetres).
 * on_device != 0: pointers are HBM addresses (no PCIe in the call); otherwise host buffers that are uploaded first.
 * stamps: S timestamps (seconds). The call is asynchronous on the batch's stream; vio_sync or any getter waits.
 * Per-sequence status codes are read back with vio_get_status. */
int vio_feed(vio_batch *h, const uint8_t *gray, const uint16_t *depth_mm, const double *stamps, int on_device);

/* The two halves of vio_feed, exposed separately the way the reference exposes them to its two threads.
 * vio_track   = predictMotion + readImage(img, t, relative_R) + updateID + packaging; publish = PUB_THIS_FRAME.
 * vio_process = inputDepth + processImage on the features packaged by the last vio_track. */
int vio_track(vio_batch *h, const uint8_t *gray, const double *stamps, int publish, int on_device);
int vio_process(vio_batch *h, const uint16_t *depth_mm, int on_device);
int vio_sync(vio_batch *h);

/* Results read out of the path (SURVEY.md §8b "Results read out").  All getters synchronise first. */
typedef struct vio_status {
    int32_t code;                 /* VIO_OK / VIO_NEED_IMU / VIO_REBOOTED of the last vio_process */
    int32_t solver_flag;          /* 0 INITIAL, 1 NON_LINEAR (estimator.h SolverFlag) */
    int32_t frame_count;
    int32_t marginalization_flag; /* 0 MARGIN_OLD, 1 MARGIN_SECOND_NEW */
    int32_t n_landmarks;          /* f_manager.feature.size() */
    int32_t last_track_num;
    int32_t n_tracks;             /* FeatureTracker::ids.size() */
    int32_t processed;            /* 1 if processImage ran for the last frame */
    int32_t iterations, successful_steps;
    int32_t n_in_problem, n_residuals, n_var_landmarks, has_prior;
    int32_t reboot_count, frames_processed;
    double initial_cost, final_cost, td;
} vio_status;
int vio_get_status(vio_batch *h, int seq, vio_status *out);
/* window state Ps/Rs/Vs/Bas/Bgs/Headers (estimator.h:121-135): (W+1) rows of 17 doubles
 * [P(3) Q(w,x,y,z) V(3) Ba(3) Bg(3) stamp] */
int vio_get_window(vio_batch *h, int seq, double *out);
/* the CSV row of visualization.cpp:214-225 for every sequence: [S][11] = stamp, P(3), Q(w,x,y,z), V(3) of frame W */
int vio_get_odometry(vio_batch *h, double *out);
/* every CSV row written so far for sequence seq (HBM ring of 2048 rows): returns the number of rows produced */
int vio_get_odometry_history(vio_batch *h, int seq, int cap, double *out);
/* tic(3), ric(9 row-major), td */
int vio_get_extrinsic(vio_batch *h, int seq, double *out13);
/* FeatureTracker public vectors after readImage (estimator_nodelet.cpp:337-343): returns count */
int vio_get_tracks(vio_batch *h, int seq, int cap, int32_t *ids, int32_t *track_cnt, float *cur_pts_xy, float *cur_un_pts_xy,
                   float *pts_velocity_xy);
/* f_manager.feature in list order: 7 doubles per landmark
 * [feature_id, start_frame, n_obs, estimated_depth, estimate_flag, solve_flag, is_dynamic]; returns total count */
int vio_get_landmarks(vio_batch *h, int seq, int cap, double *out);
/* last_marginalization_info in the canonical layout: returns n (0 = none) */
int vio_get_prior(vio_batch *h, int seq, double *J_nxn, double *r_n, double *x0, uint8_t *present);

/* Per-stage device time of the last vio_feed in milliseconds (hipEvent on the batch stream):
 * out[0] front-end, out[1] back-end, out[2] total; plus kernel-level entries, see DESIGN.md. Returns count. */
int vio_get_timings(vio_batch *h, int cap, double *out_ms);
/* Per-kernel HIP-event timing of the next max_steps vio_feed calls, recorded on the batch stream.
 * vio_profile_end: out_ms[11] = average ms of fe_begin, fe_pyrdown, fe_predict, fe_lk, fe_select, fe_fast, fe_add,
 * be_ingest, be_solve, be_marg, be_finish; returns the number of recorded steps. */
int vio_profile_begin(vio_batch *h, int max_steps);
int vio_profile_end(vio_batch *h, int cap, double *out_ms);
/* stream the batch launches on (hipStream_t) so callers can bracket it with their own events */
void *vio_get_stream(vio_batch *h);

/* ---- single stages, exposed for the parity tests (same kernels the pipeline launches) ---- */
/* cv::pyrDown inside calcOpticalFlowPyrLK (feature_tracker.cpp:302-305) */
int vio_stage_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst);
/* FastFeatureDetector::detect on a ROI, before the mask filter (feature_tracker.cpp:109-110): returns count, out = x,y,score */
int vio_stage_fast_roi(const uint8_t *img, int W, int H, int rx, int ry, int rw, int rh, int cap, float *out_xys);
/* cv::calcOpticalFlowPyrLK(21x21, maxLevel, {COUNT+EPS,30,0.01}, OPTFLOW_USE_INITIAL_FLOW) */
int vio_stage_lk(const uint8_t *prev, const uint8_t *next, int w, int h, int max_level, int n, const float *prev_pts,
                 float *next_pts_inout, uint8_t *status);
/* cv::findFundamentalMat(FM_RANSAC, F_THRESHOLD, 0.99) on virtual-pinhole points (feature_tracker.cpp:462) */
int vio_stage_ransac(const vio_config *cfg, int n, const float *p1, const float *p2, uint8_t *status);
/* IntegrationBase::push_back x n + IMUFactor::Evaluate (integration_base.h:32-162, imu_factor.h:20-205)
 * out: delta_p(3) delta_q(wxyz) delta_v(3) sum_dt jacobian(225) covariance(225); r(15); J = 15x7,15x9,15x7,15x9 */
int vio_stage_imu_factor(const vio_config *cfg, int n, const double *dt, const double *acc, const double *gyr,
                         const double *acc0, const double *gyr0, const double *ba, const double *bg, const double *pose_i,
                         const double *sb_i, const double *pose_j, const double *sb_j, double *preint_out, double *r15,
                         double *J480);
/* ProjectionFactor / ProjectionTdFactor::Evaluate (projection_factor.cpp:22-130, projection_td_factor.cpp:34-150)
 * obs = 9 doubles (x,y,z,u,v,vx,vy,cur_td,depth); J = Ji(2x7) Jj(2x7) Jex(2x7) Jl(2) Jtd(2) */
int vio_stage_projection(const vio_config *cfg, const double *pose_i, const double *pose_j, const double *ex, double inv_dep,
                         double td, const double *obs_i, const double *obs_j, int use_td, double *r2, double *J46);

#ifdef __cplusplus
}
#endif
#endif
