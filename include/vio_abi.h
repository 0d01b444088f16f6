/* MI355X-native VIO hot path — C ABI (the drop-in boundary).
 *
 * The reference has no FFI layer: its boundary is the public member-function surface of FeatureTracker / Estimator /
 * FeatureManager plus mutable globals (vins_estimator/src/utility/parameters.h:17-79).  Each entry point below names
 * the reference interface it replaces.  A handle owns a *batch* of S independent sequences (S = 1 reproduces the
 * reference's one-Estimator-per-process use); all state lives in HBM behind the handle, configuration is per handle
 * instead of process-global.  Plain C types only; the caller owns input buffers for the duration of a call.
 *
 * Host buffers: every entry point that takes host memory (images with on_device == 0, stamps, modes, R_rel, feature maps) enqueues
 * asynchronous uploads.  From pageable memory the runtime stages the copy before the call returns; from page-locked memory
 * (vio_host_alloc) the copy is truly asynchronous: such buffers must stay untouched until the NEXT call on the handle that takes host
 * buffers, any getter, or vio_sync has returned (those calls wait for the pending uploads first).  Exception (round 5): the IMAGES handed to
 * vio_feed / vio_feed_modes keep TWO uploads in flight -- they are free when vio_host_buffers_done(h, calls_ago) returns 1, and at the latest
 * when the second next vio_feed THAT TAKES HOST IMAGES, any getter or vio_sync has returned (rotate three page-locked image sets, or poll the
 * query; feeds of device-resident frames in between do not count).  This is a BREAKING change against the round-4 contract ("free when the
 * next vio_feed has returned"): a caller that ping-pongs TWO page-locked image sets must move to three or poll.  vio_abi_version() tells the
 * contracts apart (>= 5: this one); VIO_UPLOADS_IN_FLIGHT=1 in the environment restores the old wait (one upload in flight, two sets suffice).
 *
 * Threading: a handle is not re-entrant.  vio_push_imu / vio_push_imu_batch may be called from another thread than
 * vio_track / vio_process / vio_feed (internal lock), mirroring Estimator::inputIMU being called from ROS callback threads
 * (estimator.cpp:1749-1766).  vio_last_error is thread-local: it reports the last failure of the CALLING thread.
 */
#ifndef VIO_ABI_H
#define VIO_ABI_H
#include <stddef.h>
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
    int32_t window_size;            /* WINDOW_SIZE (compile-time 10 upstream, parameters.h:12); 4..20 here */
    int32_t max_landmarks;          /* NUM_OF_F (parameters.h:14): capacity of the landmark table */
    int32_t fix_depth;              /* FIX_DEPTH */
    int32_t estimate_extrinsic;     /* ESTIMATE_EXTRINSIC: 0 or 1 (2 = calibrate from scratch is out of scope) */
    int32_t estimate_td;            /* ESTIMATE_TD */
    int32_t max_iterations;         /* NUM_ITERATIONS  max_num_iterations */
    int32_t ransac_max_iters;       /* cv::findFundamentalMat RANSAC iteration cap (1000) */
    int32_t lk_max_level;           /* maxLevel of calcOpticalFlowPyrLK: 1 for the IMU-aided call (feature_tracker.cpp:303) */
    int32_t dynamic_init;           /* !STATIC_INIT (parameters.cpp:167): 0 = static initialisation (gyro-bias + optimisation on the
                                       IMU-propagated window, estimator.cpp:266-283), 1 = SfM + visual-inertial alignment
                                       (initialStructure, estimator.cpp:384-579) */
    int32_t use_imu;                /* USE_IMU (yaml key `imu`, parameters.cpp:148): 0 = visual odometry on RGB-D: no IMU factors, pose 0
                                       constant, initFramePoseByPnP per frame, LK maxLevel = lk_max_level (3 upstream) without prediction */
    int32_t reference_quirks;       /* bit 0 (VIO_QUIRK_LATEST_FRONT): vio_get_latest_odometry replays the buffered IMU the way
                                       Estimator::updateLatestStates does (estimator.cpp:1779-1786): every step uses the values of the queue's
                                       FRONT sample (the reference's behaviour).  0 = every sample with its own values. */
    int32_t marg_exact;             /* 1 = the marginalisation follows MarginalizationInfo::marginalize literally (marginalization_factor.cpp:
                                       281-315): eigen-decomposition of the FULL m x m marginalised block (pose 0, speed-bias 0 and every
                                       landmark that starts in frame 0) with the 1e-8 cut, then eigen-decomposition of the reduced system and the
                                       prior rebuilt from the truncated factors J = S^1/2 V^T, r = S^-1/2 V^T b.  0 (default) = the fast form
                                       (analytic landmark elimination, prior kept as a quadratic form; DESIGN.md deviations 10 / 13).
                                       2 (round 5) = the literal algorithm with a CERTIFIED first inverse: A_mm^+ = A_mm^-1 is formed by blocks
                                       (exact for the diagonal landmark block) and every frame proves lambda_min(A_mm) > 1e-7 from a bound on
                                       |A_mm^-1|_F, i.e. that the reference's 1e-8 cut could not have dropped anything; the second half (the
                                       new prior's eigen-decomposition with its cut, which does drop directions) is the literal one. */
    int32_t equalize;               /* EQUALIZE (yaml key `equalize`, parameters.cpp:110): 1 = cv::createCLAHE(3.0, Size(8, 8)) on every image
                                       before tracking (feature_tracker.cpp:269-275).  Occupies what used to be alignment padding: the
                                       offsets of all other fields are unchanged. */
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

enum { VIO_QUIRK_LATEST_FRONT = 1,
       /* bit 2 (value 4): ignored by the library; the test oracle mirrors deviation 15 (extrinsic held in a relocalisation solve) when set */
       VIO_ORACLE_RELO_HOLDS_EXTRINSIC = 4,
       /* bit 3 (value 8), TEST SWITCH, not a reference behaviour (both sides read it): the inverse-depth bound of depth-less landmarks
        * (estimator.cpp:1282-1297) is honoured by clamping candidates only, as rounds 1 - 5 did (former deviation 5).  0 = Ceres'
        * treatment of a bounds-constrained program: x0 projected before the first evaluation, projected Armijo line search along every
        * trust-region step (be_phased.h ps_eval; oracle/backend.cpp solve()). */
       VIO_QUIRK_BOUND_CLAMP_ONLY = 8,
       /* TEST HOOK, not a reference behaviour: bits 8..11 = n: the first n Cholesky factorisations of EVERY solve are reported as failed
        * (on both sides: the oracle reads the same bits), which walks the mu *= 10 retry ladder of the trust-region loop
        * (oracle/backend.cpp solve(); be_phased.h ps_serial).  Every retry uses one iteration slot: see VIO_EXTRA_SLOTS. */
       VIO_TEST_CHOL_FAIL_SHIFT = 8, VIO_TEST_CHOL_FAIL_MASK = 15 };

typedef struct vio_batch vio_batch; /* opaque */

/* parameters.cpp:81-243 readParameters() defaults for config/realsense/vio.yaml at 150 features */
void vio_config_default(vio_config *cfg);

/* Estimator::Estimator + setParameter() (estimator.cpp:9-41) for S sequences on the current HIP device.
 * imu_capacity = ring size per sequence (samples). Returns NULL on failure (see vio_last_error): no HIP device (there is no CPU
 * fallback), an allocation that failed, or a configuration whose kernels would need more LDS than a workgroup may own (window size /
 * landmark capacity / feature count beyond what a CU's 160 KB hold) -- checked here so that it cannot surface as a failed launch
 * in the middle of a frame. */
vio_batch *vio_create(const vio_config *cfg, int n_seq, int imu_capacity);
/* The same on a named HIP device (device < 0: the calling thread's current device, i.e. vio_create).  The handle OWNS its device: all of its
 * memory, streams and events live there, and every entry point that takes the handle makes that device current for the duration of the call
 * and restores the caller's afterwards -- so one process may drive one handle per GPU from one thread or from a thread per GPU
 * without ever calling hipSetDevice itself.  Device pointers passed with on_device != 0 must belong to the handle's device.  No upstream
 * counterpart (the reference is one Estimator per process on the CPU). */
vio_batch *vio_create_on_device(const vio_config *cfg, int n_seq, int imu_capacity, int device);
int vio_get_device(vio_batch *h);   /* the device the handle lives on; VIO_EINVAL for NULL */
void vio_destroy(vio_batch *h);
const char *vio_last_error(void);
/* Every sequence back to the state after vio_create: fresh FeatureTracker AND Estimator::clearState() + setParameter(). */
int vio_reset(vio_batch *h);

/* Estimator::inputIMU(t, acc, gyr) (estimator.cpp:1749-1766) for sequence seq; n samples, t strictly increasing. */
int vio_push_imu(vio_batch *h, int seq, int n, const double *t, const double *acc_xyz, const double *gyr_xyz);
/* The same for every sequence in one call (SURVEY.md 8b "batched variants taking SoA pointers"): n[s] samples for sequence s,
 * stored at t[s * stride + i], acc_xyz / gyr_xyz[(s * stride + i) * 3 + k]; n == NULL means `stride` samples for every sequence. */
int vio_push_imu_batch(vio_batch *h, const int32_t *n, int stride, const double *t, const double *acc_xyz, const double *gyr_xyz);
/* Estimator::clearState() + setParameter() for ONE sequence (estimator.cpp:15-116): what the stream-discontinuity branch of
 * process_tracker does (estimator_nodelet.cpp:243-262) -- window, landmarks, prior, pre-integrations and the buffered IMU of that
 * sequence are dropped, the pending feature map is discarded (feature_buf popped, :250-253), the nodelet's first_image_flag is set
 * again and last_image_time zeroed (:246-247; they only matter to vio_feed's device-side gating).  The FeatureTracker is NOT touched:
 * trackerData keeps its points, ids, track counts and previous image, init_pub / init_feature keep their values, exactly as upstream. */
int vio_reset_seq(vio_batch *h, int seq);
/* A fresh FeatureTracker for one sequence (no upstream counterpart: the reference only ever constructs its tracker once). */
int vio_reset_tracker_seq(vio_batch *h, int seq);

/* One camera frame for every sequence: the body of EstimatorNodelet::process_tracker for one synchronised
 * colour+depth pair (estimator_nodelet.cpp:234-393) followed by EstimatorNodelet::process (:462-549):
 *   predictMotion (estimator.cpp:1790-1860) -> FeatureTracker::readImage (feature_tracker.cpp:263-439) -> updateID loop
 *   -> feature-map packaging -> FeatureManager::inputDepth -> Estimator::processImage (estimator.cpp:156-374).
 * gray: S images [S][height][width] u8, depth_mm: [S][height][width] u16 (CV_16UC1 millimetres).
 * on_device != 0: pointers are HBM addresses (no PCIe in the call); otherwise host buffers that are uploaded first.
 * stamps: S timestamps (seconds). The call is asynchronous on the batch's stream; vio_sync or any getter waits.
 * Per-sequence status codes are read back with vio_get_status. */
/* Host IMAGE buffers (on_device == 0) are uploaded by a copy stream beside the previous frames' kernels, two uploads in flight: pageable
 * buffers are free when the call returns (the runtime stages them), page-locked ones (vio_host_alloc) when vio_host_buffers_done says so and at
 * the latest once the SECOND next vio_feed / vio_feed_modes call on the handle (or any getter / vio_sync) has returned.  `stamps` (and `modes`)
 * are copied into a library-owned page-locked ring before the call returns: they are free at once, page-locked or not. */
int vio_feed(vio_batch *h, const uint8_t *gray, const uint16_t *depth_mm, const double *stamps, int on_device);
/* Non-blocking: 1 if the image uploads of the vio_feed call issued `calls_ago` calls ago (0 = the latest) have completed and its host buffers
 * may be rewritten, 0 if they are still in flight; calls_ago >= 2 is always 1.  No upstream counterpart (the reference's images arrive as ROS
 * messages it owns). */
int vio_host_buffers_done(vio_batch *h, int calls_ago);

/* The two halves of vio_feed, exposed separately the way the reference exposes them to its two threads.
 * vio_track   = predictMotion + readImage(img, t, relative_R) + updateID + packaging; publish = PUB_THIS_FRAME.
 * vio_process = inputDepth + processImage on the features packaged by the last vio_track. */
int vio_track(vio_batch *h, const uint8_t *gray, const double *stamps, int publish, int on_device);
int vio_process(vio_batch *h, const uint16_t *depth_mm, int on_device);
int vio_sync(vio_batch *h);

/* Per-sequence frame modes = the outcome of the nodelet's frequency control for one frame (estimator_nodelet.cpp:264-286):
 * SKIP    the frame is dropped before readImage ("Skip this frame", :266-271): no state changes at all;
 * TRACK   readImage with PUB_THIS_FRAME == false: LK + culling + track_cnt++ only, no RANSAC / mask / detection
 *         (feature_tracker.cpp:351), nothing handed to processImage;
 * PUBLISH readImage with PUB_THIS_FRAME == true + processImage. */
enum { VIO_FRAME_SKIP = 0, VIO_FRAME_TRACK = 1, VIO_FRAME_PUBLISH = 2 };
/* vio_feed with a mode per sequence (modes[S], host memory; NULL = PUBLISH for all). */
int vio_feed_modes(vio_batch *h, const uint8_t *gray, const uint16_t *depth_mm, const double *stamps, const uint8_t *modes, int on_device);
/* vio_track with a mode per sequence and an optional caller-supplied relative rotation per sequence:
 * FeatureTracker::readImage(img, cur_time, relative_R) (feature_tracker.h:36-37).  R_rel: [S][9] row-major host memory or NULL;
 * a NaN in the first element of a sequence's matrix = predict it on the device (Estimator::predictMotion). */
int vio_track_ex(vio_batch *h, const uint8_t *gray, const double *stamps, const uint8_t *modes, const double *R_rel, int on_device);
/* Estimator::predictMotion(t0, t1) (estimator.cpp:1790-1860) for sequence seq on the IMU pushed so far: R[9] row-major. */
int vio_predict_motion(vio_batch *h, int seq, double t0, double t1, double *R9);
/* Estimator::processImage(image, header) (estimator.h:46, estimator.cpp:156-374) + FeatureManager::inputDepth with a CALLER-SUPPLIED
 * feature map for sequence seq: n entries in ascending feature id (std::map order), xyz_uv_vel[n][7] = (x, y, z = 1, u, v, vx, vy)
 * exactly as estimator_nodelet.cpp:336-363 packs them, depth_mm = the CV_16UC1 depth image of THAT frame (host memory), stamp =
 * header stamp.  This is what the process thread pops from feature_buf (estimator_nodelet.cpp:380-384, 539), so the tracker may
 * run ahead of the estimator by any number of frames.  n <= the tracker capacity (max_cnt + grid slack, see vio_get_capacity). */
int vio_process_obs(vio_batch *h, int seq, int n, const int32_t *ids, const double *xyz_uv_vel, const uint16_t *depth_mm, double stamp);
/* The same for the whole batch: n_obs[S] (a negative count skips the sequence), ids[S][cap], xyz_uv_vel[S][cap][7], depth_mm
 * [S][height][width], stamps[S]; the feature arrays are host memory, depth_mm is HBM when on_device != 0. */
int vio_process_obs_batch(vio_batch *h, const int32_t *n_obs, const int32_t *ids, const double *xyz_uv_vel, int cap,
                          const uint16_t *depth_mm, const double *stamps, int on_device);
/* The feature map packaged by the last vio_track / vio_feed for sequence seq (what the nodelet would push to feature_buf):
 * returns the count (0 when the frame was not published), ids ascending. */
int vio_get_packaged(vio_batch *h, int seq, int cap, int32_t *ids, double *xyz_uv_vel);
/* HBM buffers for callers that do not link HIP themselves (the on_device != 0 arguments are plain device addresses): synchronous
 * hipMalloc / hipFree / hipMemcpy on the current device.  vio_device_alloc returns NULL on failure. */
void *vio_device_alloc(size_t bytes);
void vio_device_free(void *p);
int vio_device_upload(void *dst_device, const void *src_host, size_t bytes);
int vio_device_download(void *dst_host, const void *src_device, size_t bytes);
/* page-locked host memory for the on_device == 0 image arguments: uploads from it run at PCIe rate and asynchronously (from pageable
 * memory the runtime stages every copy).  NULL on failure. */
void *vio_host_alloc(size_t bytes);
void vio_host_free(void *p);
/* sizeof(vio_config) (what = 0) / sizeof(vio_status) (what = 1) as compiled into the library: lets a binding check its struct mirrors */
int vio_abi_sizeof(int what);
/* Contract version of this header: 4 = a vio_feed host image set is free when the next vio_feed has returned; 5 = two uploads in flight (see
 * "Host buffers" above, vio_host_buffers_done); 6 = + vio_get_bound_stats, the inverse-depth bound handled as Ceres does (projected line search). */
int vio_abi_version(void);
/* capacities derived from the configuration: out[0] = tracker points per sequence, out[1] = landmark slots, out[2] = IMU ring */
int vio_get_capacity(vio_batch *h, int32_t *out3);
/* which solver the handle's configuration selected: 0 = the persistent one-workgroup-per-sequence kernel (fallback), 1 = the phased solver
 * with the Schur complement resident in LDS (windows up to W = 10), 2 = the phased solver with it in HBM / L2 (larger windows) */
int vio_get_solver_kind(vio_batch *h);
/* Bounds-constrained solves (estimator.cpp:1282-1297: SetParameterUpperBound(para_Feature, 0, 2 / DEPTH_MAX_DIST) on landmarks triangulated
 * without a depth measurement, which makes Ceres project x0 onto the box and run its Armijo line search along every step).  Counters of
 * sequence seq since vio_create / vio_reset: out4 = {inverse depths cut by the bound while a point was formed, bounded landmarks that
 * entered solves, trial evaluations of the line search, shortened steps}.  Synchronises the device. */
int vio_get_bound_stats(vio_batch *h, int seq, int64_t *out4);
/* vio_config.marg_exact = 2 only: out2 = {marginalisations of sequence seq whose certificate FAILED since vio_create / vio_reset (such a frame
 * keeps the block inverse without the proof that the reference's 1e-8 eigenvalue cut of marginalization_factor.cpp:281-291 drops nothing),
 * 1 if the last marginalisation was certified}.  Synchronises the device. */
int vio_get_marg_certificate(vio_batch *h, int seq, int32_t *out2);

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
    int32_t overflow_flags;       /* capacity flags of the last frame: 1 landmark table, 2 IMU slot (> 64 samples per frame interval),
                                     4 FAST candidates of a cell, 8 residual list, 16 IMU ring overwritten, 32 solver iteration slots
                                     exhausted before the trust-region loop finished, 64 relocalisation request dropped (vio_set_relo_frame)
                                     (code = VIO_ECAPACITY) */
    int32_t overflow_frames;      /* frames that raised any capacity flag since the last reset / reboot */
    int32_t iterations_total, solves_total; /* solver iterations / solves since vio_create */
} vio_status;
int vio_get_status(vio_batch *h, int seq, vio_status *out);
/* the same for every sequence of the handle in two device reads: out[n_seq] */
int vio_get_status_all(vio_batch *h, vio_status *out);
/* window state Ps/Rs/Vs/Bas/Bgs/Headers (estimator.h:121-135): (W+1) rows of 17 doubles
 * [P(3) Q(w,x,y,z) V(3) Ba(3) Bg(3) stamp] */
int vio_get_window(vio_batch *h, int seq, double *out);
/* the CSV row of visualization.cpp:214-225 for every sequence: [S][11] = stamp, P(3), Q(w,x,y,z), V(3) of frame W */
int vio_get_odometry(vio_batch *h, double *out);
/* CSV rows written for sequence seq.  The device keeps a ring of the last 2048 rows: out receives the most recent min(rows, 2048,
 * cap) of them in time order; returns the number of rows produced since vio_create / the last reset (may exceed what fits). */
int vio_get_odometry_history(vio_batch *h, int seq, int cap, double *out);
/* FISHEYE (parameters.cpp:111-114, estimator.cpp:29-36): FeatureTracker::setMask starts from fisheye_mask instead of an all-255 image
 * (feature_tracker.cpp:175-176), so tracked and new features are only kept where the mask is 255 and FAST corners only where it is not 0.
 * mask = ROW x COL u8 (the decoded config/fisheye_mask.jpg; decoding is the caller's), shared by all sequences of the handle; NULL turns
 * the option off.  Synchronises the handle. */
int vio_set_fisheye_mask(vio_batch *h, const uint8_t *mask, int on_device);
/* How far the estimator runs behind the tracker inside vio_feed / vio_feed_modes.  The reference's process_tracker and process()
 * threads run concurrently (estimator_nodelet.cpp:192-459 / :462-549): Estimator::predictMotion of frame f+1 reads latest_Bg / td as the
 * estimator thread left them, i.e. after frame f when the estimator keeps up and after frame f-1 when it is still optimising frame f.
 *   lag 0 (default): the tracker of frame f+1 starts after the optimisation of frame f (estimator keeps up);
 *   lag 1: it reads the state as of frame f-1 (a snapshot be_ingest takes) and overlaps the optimisation of frame f, which takes
 *          the front-end off the critical path.  Results are deterministic in both modes; the oracle implements the same two
 *          orderings (ovio_set_tracker_lag).  Not available on dynamic_init handles.  vio_track / vio_process_obs callers choose the
 *          ordering themselves.
 * Changing the lag synchronises the handle and re-creates its streams (with lag 1 the front-end streams are confined to their own
 * compute units): a stream obtained from vio_get_stream before the call is no longer valid. */
int vio_set_tracker_lag(vio_batch *h, int lag);
/* the IMU-rate pose of pubLatestOdometry (Estimator::predict, estimator.cpp:1862-1880, fed by inputIMU :1749-1766 after
 * updateLatestStates :1768-1788): the newest window state propagated through every IMU sample pushed after it.
 * out11 = t, P(3), Q(w, x, y, z), V(3).  INITIAL sequences and VO mode return the window state unchanged. */
int vio_get_latest_odometry(vio_batch *h, int seq, double *out11);
/* Estimator::setReloFrame(frame_stamp, frame_index, match_points, relo_t, relo_r) (estimator.h:48-49, estimator.cpp:1728-1747) for
 * sequence seq: the window frame stamped frame_stamp has been matched with an old keyframe (pose relo_t / relo_r in the loop-closed
 * world, relo_r row-major); match_points[n][3] = (x, y) normalised point of the old keyframe and the feature id it matched (z), in
 * ascending id like the pose graph sends them (pose_graph/src/keyframe.cpp).  If the stamp is one of Headers[0 .. W-1] the NEXT
 * optimisation of that sequence carries the relocalisation factors (estimator.cpp:1307-1346: ProjectionFactor(first observation,
 * matched point) on (para_Pose[start], relo_Pose, extrinsic, inverse depth) for every in-problem landmark with start_frame <=
 * relo_frame_local_index among the matches) and computes the drift outputs (:1034-1056); otherwise nothing happens, like upstream.
 * relo_Pose borrows the six tangent columns of the extrinsic in the reduced system: when ESTIMATE_EXTRINSIC has opened the extrinsic it is
 * held constant in the ONE solve that carries the factors and refined again from the next solve on (DESIGN.md deviation 15: below a
 * millimetre against the joint optimisation).  Limits of this build: IMU mode and the phased solver; on the persistent solver (windows
 * beyond its range) the request is dropped and the frame carries overflow flag 64. */
int vio_set_relo_frame(vio_batch *h, int seq, double frame_stamp, int frame_index, int n, const double *match_points, const double *relo_t3,
                       const double *relo_r9);
/* what pubRelocalization and the pose graph read after that solve (visualization.cpp:454-538): out30 = relo_relative_t(3),
 * relo_relative_q(w x y z), relo_relative_yaw (degrees), drift_correct_t(3), drift_correct_r(9 row-major), relo_Pose(7: p, q x y z w),
 * relocalization_info (1 = still pending), relo_frame_local_index, number of relocalisation factors of the last solve */
int vio_get_relo(vio_batch *h, int seq, double *out30);
/* tic(3), ric(9 row-major), td */
int vio_get_extrinsic(vio_batch *h, int seq, double *out13);
/* FeatureTracker public vectors after readImage (estimator_nodelet.cpp:337-343): returns count */
int vio_get_tracks(vio_batch *h, int seq, int cap, int32_t *ids, int32_t *track_cnt, float *cur_pts_xy, float *cur_un_pts_xy,
                   float *pts_velocity_xy);
/* f_manager.feature in list order: 7 doubles per landmark
 * [feature_id, start_frame, n_obs, estimated_depth, estimate_flag, solve_flag, is_dynamic]; returns total count */
int vio_get_landmarks(vio_batch *h, int seq, int cap, double *out);
/* the same plus what pubPointCloud reads (visualization.cpp:333-395): 12 doubles per landmark
 * [feature_id, start_frame, n_obs, estimated_depth, estimate_flag, solve_flag, is_dynamic,
 *  feature_per_frame[0].point (x, y, z), feature_per_frame[0].depth (metres, 0 = none), feature_per_frame.back().depth] */
int vio_get_landmarks_ex(vio_batch *h, int seq, int cap, double *out12);
/* last_marginalization_info in the canonical layout: returns n (0 = none) */
int vio_get_prior(vio_batch *h, int seq, double *J_nxn, double *r_n, double *x0, uint8_t *present);

/* Per-stage device time of the last vio_feed in milliseconds (hipEvent on the batch stream):
 * out[0] front-end, out[1] back-end, out[2] total; plus kernel-level entries, see DESIGN.md. Returns count. */
int vio_get_timings(vio_batch *h, int cap, double *out_ms);
/* Per-kernel HIP-event timing of the next max_steps vio_feed calls, recorded on the batch's streams.
 * vio_profile_end: out_ms[10] = average ms of fe_begin, fe_pyrdown, fe_predict, fe_lk, fe_select, fe_fast, fe_add,
 * be_ingest, be_solve, be_marg (be_finish is fused into be_marg); returns the number of recorded steps (steps driven by
 * vio_track / vio_process instead of vio_feed are not counted). */
int vio_profile_begin(vio_batch *h, int max_steps);
int vio_profile_end(vio_batch *h, int cap, double *out_ms);
/* stream the batch launches on (hipStream_t) so callers can bracket it with their own events */
void *vio_get_stream(vio_batch *h);

/* ---- single stages, exposed for the parity tests (same kernels the pipeline launches) ---- */
/* cv::pyrDown inside calcOpticalFlowPyrLK (feature_tracker.cpp:302-305) */
int vio_stage_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst);
/* cv::createCLAHE(3.0, Size(8, 8))->apply (feature_tracker.cpp:269-275, EQUALIZE) */
int vio_stage_clahe(const uint8_t *src, int w, int h, uint8_t *dst);
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
/* the same factor through the per-residual device routine the marginalisation and the outlier rejection use (the solver's hot loop
 * uses the frame-pair form above) */
int vio_stage_projection_residual(const vio_config *cfg, const double *pose_i, const double *pose_j, const double *ex, double inv_dep,
                                  double td, const double *obs_i, const double *obs_j, int use_td, double *r2, double *J46);
/* cv::solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess) with K = I as FeatureManager::solvePoseByPnP calls it in VO mode
 * (feature_manager.cpp:545-588): obj[n][3], img[n][2] normalised points, rvec3 / tvec3 (Rodrigues vector, translation) in and out */
int vio_stage_pnp(int n, const double *obj, const double *img, double *rvec3, double *tvec3);
/* Host-only building blocks of the dynamic (static_init: 0) initialisation (csrc/dyninit_host.cpp), callable without a GPU:
 * cv::solvePnP(ITERATIVE, useExtrinsicGuess) (R9 / t3 in: guess, out: result; returns 1 = converged), cv::solvePnPRansac(EPNP) as
 * solveRelativeRT_PNP uses it (solve_5pts.cpp:248-294), and relativePose + GlobalSFM::construct over a window (estimator.cpp:884-920,
 * initial_sfm.cpp:184-412: tracks as start[nf], nobs[nf], obs = (x, y, depth) per observation; out: reference frame l, q[(W+1)*4]
 * (w x y z), T[(W+1)*3], pts[nf*4] = (solved, X, Y, Z), stats[2] = (BA iterations, points in the BA); returns 0 ok, 1 no frame pair
 * with enough parallax, 2 SfM failed). */
int vio_stage_host_pnp(int n, const double *obj, const double *img, double *R9, double *t3);
int vio_stage_host_pnp_ransac(int n, const double *obj, const double *img, int max_iters, double thresh, double confidence, double *R9, double *t3);
/* LinearAlignmentWithDepth + RefineGravityWithDepth (initial_aligment.cpp:170-244, 337-405): frames19 = n x {R[9] row-major, T[3], sum_dt,
 * delta_p[3], delta_v[3]}; g_out3 = gravity in the SfM frame, x_out[3 n + 3] = body velocities per frame + the last correction;
 * returns 1 when |g| came out within 1 m/s^2 of g_norm before the refinement. */
int vio_stage_host_alignment(int n, const double *frames19, const double *tic3, double g_norm, double *g_out3, double *x_out);
int vio_stage_host_sfm_window(int window_size, int nf, const int32_t *start, const int32_t *nobs, const double *obs, int32_t *l_out, double *q_out,
                              double *T_out, double *pts_out, double *stats_out);
/* IMUFactor::Evaluate as the solver consumes it: G961 = [J r]^T [J r] (31 x 31 row-major; columns pose_i(6) speedbias_i(9) pose_j(6)
 * speedbias_j(9) | residual) through the wavefront code of the solve kernel (split raw Jacobians, on-the-fly whitening, FP64 MFMA) */
int vio_stage_imu_block(const vio_config *cfg, int n, const double *dt, const double *acc, const double *gyr, const double *acc0,
                        const double *gyr0, const double *ba, const double *bg, const double *pose_i, const double *sb_i,
                        const double *pose_j, const double *sb_j, double *G961);
/* The dense solve at the bottom of every trust-region step (Ceres DENSE_SCHUR on the reduced camera system, estimator.cpp:1251-1263;
 * Eigen LLT underneath) through the LDS-tile Cholesky of the solve kernels: S [16 nb][16 nb] row-major symmetric positive definite
 * (nb <= 11; 12 <= nb <= 24 or blocks = -7: the streaming factorisation of windows beyond 10 keyframes, tiles in HBM -- usec5 is then
 * {factorisation, backward substitution, update of the block columns incl. the wait for their diagonal tiles, panels + tile stores, 0}),
 * rhs [16 nb].  L_out: the lower triangle of the factor (row-major; entries above the diagonal are left as passed in),
 * x_out = S^-1 rhs (NaN when a pivot was not positive), usec5 = {factorisation with the forward substitution riding along, backward
 * substitution, then the factorisation as thread 0 sees it: panels, diagonal block + trailing update, barrier wait}: microseconds inside the kernel, mean over `reps` repetitions, with `blocks` identical workgroups side by side.  blocks = -1 .. -6
 * select the timing micro-modes of tools/chol_bench.py (one diagonal tile, panel tiles, dependent FP64 chains, v_mfma_f64_16x16x4 issue
 * rates; usec5 then carries clock64 ticks / 100 in slots 1 .. 4, L_out / x_out are not written): measurement only, see stage_linalg.hip. */
int vio_stage_chol(int nb, int reps, int blocks, const double *S, const double *rhs, double *L_out, double *x_out, double *usec5);
/* Harness of the HBM-resident symmetric eigen-solver of the literal marginalisation (marg_exact = 1, blocks beyond the LDS-resident size: Householder
 * tridiagonalisation + implicit QL by one workgroup, be_linalg.h sym_eig_hbm; Eigen::SelfAdjointEigenSolver at marginalization_factor.cpp:281, :298 is
 * what it stands in for).  A: symmetric n x n, row-major, 2 <= n <= 512.  evals[n] unsorted, evecs[i * n + k] = component i of the eigenvector of
 * evals[k], usec (may be NULL; FOUR doubles) = device time of the decomposition and of its tridiagonalisation / accumulation / QL phases. */
int vio_stage_sym_eig(int n, const double *A, double *evals, double *evecs, double *usec);

#ifdef __cplusplus
}
#endif
#endif
