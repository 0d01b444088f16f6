// ORACLE (test infrastructure only — never linked into or called by the product path).
// CPU restatement of the VINS-RGBD-FAST hot path:
//   front-end  FeatureTracker::readImage        vins_estimator/src/feature_tracker/feature_tracker.cpp:263-439
//   back-end   Estimator::processImage          vins_estimator/src/estimator/estimator.cpp:156-374
//              Estimator::optimization          vins_estimator/src/estimator/estimator.cpp:1161-1578
// Parity status: "parity unpinned" (no golden vectors upstream; OpenCV/Ceres/Eigen are absent).
#pragma once
#include <array>
#include <cstdint>
#include <list>
#include <map>
#include <utility>
#include <vector>
#include "om.h"

namespace ovio {

// Mirrors the globals of vins_estimator/src/utility/parameters.h:11-79 that the hot path reads,
// per instance instead of process-global.
struct Config {
    int width = 640, height = 480;     // COL, ROW
    int max_cnt = 150, min_dist = 15;  // MAX_CNT, MIN_DIST
    int grid_rows = 5, grid_cols = 6;  // NUM_GRID_ROWS/COLS
    int window_size = 10;              // WINDOW_SIZE (compile-time 10 upstream, parameters.h:12)
    int max_landmarks = 1000;          // NUM_OF_F (parameters.h:14)
    int fix_depth = 1;                 // FIX_DEPTH
    int estimate_extrinsic = 0;        // ESTIMATE_EXTRINSIC (0/1 supported)
    int estimate_td = 0;               // ESTIMATE_TD
    int max_iterations = 8;            // NUM_ITERATIONS
    int ransac_max_iters = 1000;
    int lk_max_level = 1;              // IMU-aided call uses maxLevel=1 (feature_tracker.cpp:303)
    int dynamic_init = 0;              // = !STATIC_INIT (parameters.cpp:167): 0 = gyro-bias + optimisation on the IMU-propagated window (the built hot
                                       // path), 1 = SfM + visual-inertial alignment
    int use_imu = 1;                   // USE_IMU (parameters.cpp:148, yaml key `imu`): 0 = visual odometry on RGB-D (no IMU factors, pose of the
                                       // oldest frame constant, per-frame solvePnP initial guess, LK with maxLevel 3 and no prediction)
    int reference_quirks = 0;          // bit 0: latestOdometry replays the buffered IMU with the FRONT sample's values (estimator.cpp:1779-1786)
    int marg_exact = 0;                // product-side switch (the oracle always follows marginalization_factor.cpp:281-315); layout only
    int equalize = 0;                  // EQUALIZE (parameters.cpp:110): cv::createCLAHE(3.0, Size(8, 8)) on the image before tracking
    double fx = 604.5821781259577, fy = 604.2544712985845, cx = 321.2638233484251, cy = 239.70969315130674;
    double k1 = 0.13387871564774004, k2 = -0.2731913133377051, p1 = 0.0020296263577681264, p2 = -0.00044384544608203714;
    double focal_length = 460.0;       // FOCAL_LENGTH
    double f_threshold = 1.0;          // F_THRESHOLD
    double depth_min = 0.3, depth_max = 6.0;
    double acc_n = 0.1, acc_w = 0.001, gyr_n = 0.01, gyr_w = 0.0001, g_norm = 9.805;
    double ric[9] = {0.02629567, -0.00713751, 0.99962873, -0.99934346, 0.02474397, 0.02646484, -0.02492368, -0.99966834, -0.00648216};
    double tic[3] = {0.17336835, 0.049596, -0.10574841};
    double td = 0.0, tr = 0.0;         // TD, TR (rolling shutter readout)
    double min_parallax_px = 10.0;     // keyframe_parallax
    double init_depth = 5.0;           // INIT_DEPTH (parameters.cpp:215)
};

struct P2f { float x, y; };

struct KeyPt { float x, y, response; };

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
};

// ------------------------------------------------------------------------------------ camera
// camera_model/src/camera_models/PinholeCamera.cc:449-510 (liftProjective), :519-542 (spaceToPlane),
// :645-662 (distortion)
void cam_distortion(const Config &c, double x, double y, double &dx, double &dy);
void cam_lift(const Config &c, double u, double v, double &x, double &y);  // z = 1
void cam_project(const Config &c, double X, double Y, double Z, double &u, double &v);

// ------------------------------------------------------------------------------------ vision primitives
void pyr_down(const Image &src, Image &dst);
void clahe_apply(const uint8_t *src, int W, int H, uint8_t *dst, double clip = 3.0, int tiles = 8);
// FAST-9/16 threshold 10 + NMS on a ROI, row-major order, ROI-relative coordinates
void fast_detect_roi(const uint8_t *img, int W, int H, int rx, int ry, int rw, int rh, std::vector<KeyPt> &out, int thr = 10);
int fast_corner_score(const uint8_t *p, int stride, int thr);  // returns 0 if not a corner
void circle_halfwidths(int radius, std::vector<int> &hw);      // cv::circle(filled) raster shape
void lk_track(const std::vector<Image> &prev, const std::vector<Image> &next, const std::vector<P2f> &prevPts,
              std::vector<P2f> &nextPts, std::vector<uint8_t> &status, int maxLevel, bool useInitialFlow);
int seven_point_models(const double *x1, const double *y1, const double *x2, const double *y2, double F[3][9]);
void ransac_fundamental(const Config &c, const std::vector<P2f> &p1, const std::vector<P2f> &p2,
                        std::vector<uint8_t> &status);

// ------------------------------------------------------------------------------------ pose_graph (posegraph.cpp; no DBoW2 query)
// place recognition (bow.cpp): BriefVocabulary + BriefDatabase of PoseGraph (pose_graph.h:83-84), L1 scoring only
struct BowVoc {
    struct Node { int parent = 0, word_id = 0; double weight = 0; uint64_t desc[4] = {0, 0, 0, 0}; std::vector<int> children; };
    int k = 0, L = 0, scoring = 0, weighting = 0, nentries = 0;
    std::vector<Node> nodes;                                  // [0] = root
    std::vector<int> words;                                   // word id -> node id
    std::vector<std::vector<std::pair<int, double>>> ifile;   // inverted file: word id -> (entry id, weight)
    bool load_bin(const char *path);
    bool build(int k, int L, int scoring, int weighting, int nn, const int32_t *nid, const int32_t *pid, const double *w, const uint64_t *d, int nw,
               const int32_t *wn, const int32_t *wi);
    void transform_one(const uint64_t *f, int &word_id, double &weight) const;
    void transform(const uint64_t *desc, int n, std::map<int, double> &v) const;
    int add(const uint64_t *desc, int n);
    int query(const uint64_t *desc, int n, int max_results, int max_id, std::vector<std::pair<int, double>> &ret) const;
    int detect_loop(const uint64_t *desc, int n, int frame_index);
};
void gaussian_blur_9x9(const uint8_t *src, int W, int H, uint8_t *dst);
void brief_compute(const uint8_t *blur, int W, int H, const float *xy, int n, const int *pat, uint64_t *desc);
void brief_match(const uint64_t *wd, int n, const uint64_t *od, int m, int *best_index, int *best_dist);
int find_connection(int n, const float *pt3d, const float *pt_norm, const double *pt_id, const int *match, const float *old_norm,
                    const double *vio_T, const double *vio_R, const double *qic9, const double *tic3, int min_loop_num,
                    double *loop_info8, double *match_points, int *n_match_out, double *pnp_T3, double *pnp_R9);
void optimize_6dof(int n, const double *t_in, const double *R_in, const int *sequence, const int *loop_to, const double *loop_info,
                   double *t_out, double *R_out, double *drift12);
void optimize_4dof(int n, const double *t_in, const double *R_in, const int *sequence, const int *loop_to, const double *loop_info,
                   double *t_out, double *R_out, double *drift4);

// ------------------------------------------------------------------------------------ FeatureTracker
struct Tracker {
    Config cfg;
    std::vector<Image> cur_pyr, forw_pyr;
    bool has_img = false;
    std::vector<uint8_t> mask;
    std::vector<P2f> cur_pts, forw_pts, predict_pts, unstable_pts, cur_un_pts, pts_velocity;
    std::vector<int> ids, track_cnt;
    std::map<int, P2f> cur_un_pts_map, prev_un_pts_map;
    double cur_time = 0, prev_time = 0;
    int n_id = 0;
    // grid detector (feature_tracker.cpp:33-94)
    struct Rect { int x, y, w, h; };
    std::vector<Rect> grids_rect;
    std::vector<int> grids_track_num;
    std::vector<uint8_t> grids_texture_status;
    int grid_height = 0, grid_width = 0, grid_res_height = 0, grid_res_width = 0, grids_threshold = 0;
    std::vector<int> circle_hw;

    explicit Tracker(const Config &c);
    void readImage(const uint8_t *img, double t, const double R[9], bool publish);
    std::vector<uint8_t> fisheye_mask;   // FISHEYE (estimator.cpp:29-36, feature_tracker.h:68): ROW x COL, empty = off
    void updateIDs();
    bool inBorder(const P2f &pt) const;
    void rejectWithF();
    void setMask();
    std::vector<KeyPt> gridDetect(int grid_id);
    void addPoints(const std::vector<KeyPt> &kps);
    void undistortedPoints();
    void predictPtsInNextFrame(const double R[9]);
    void drawCircle(const P2f &pt);
    uint8_t maskAt(const P2f &pt) const;
};

// ------------------------------------------------------------------------------------ back-end types
// factor/integration_base.h:9-217
struct Integration {
    double acc_n, acc_w, gyr_n, gyr_w;
    om::V3 acc_0, gyr_0, linearized_acc, linearized_gyr, linearized_ba, linearized_bg;
    double jacobian[15][15], covariance[15][15];
    double sum_dt = 0;
    om::V3 delta_p, delta_v;
    om::Q delta_q;
    std::vector<double> dt_buf;
    std::vector<om::V3> acc_buf, gyr_buf;
    Integration(const Config &c, const om::V3 &a0, const om::V3 &g0, const om::V3 &ba, const om::V3 &bg);
    void push_back(double dt, const om::V3 &acc, const om::V3 &gyr);
    void propagate(double dt, const om::V3 &acc1, const om::V3 &gyr1);
    void repropagate(const om::V3 &ba, const om::V3 &bg);
    void evaluate(const om::V3 &G, const om::V3 &Pi, const om::Q &Qi, const om::V3 &Vi, const om::V3 &Bai, const om::V3 &Bgi,
                  const om::V3 &Pj, const om::Q &Qj, const om::V3 &Vj, const om::V3 &Baj, const om::V3 &Bgj,
                  double r[15]) const;
};

struct Obs {  // feature_manager.h:40-65 FeaturePerFrame
    double x, y, z;  // normalised point (z = 1)
    double u, v;
    double vx, vy;
    double cur_td;
    double depth;
};
struct Landmark {  // feature_manager.h:67-87 FeaturePerId
    int feature_id;
    int start_frame;
    std::vector<Obs> obs;
    int used_num = 0;
    bool is_dynamic = false;
    double estimated_depth = -1.0;
    int estimate_flag = 0;
    int solve_flag = 0;
    int endFrame() const { return start_frame + (int)obs.size() - 1; }
};

struct ImuSample { double t; om::V3 acc, gyr; };

enum { MAXW = 20 };

// Result of one solve (for tests / diagnostics)
struct SolveStats {
    int iterations = 0, successful = 0;
    double initial_cost = 0, final_cost = 0;
    int n_landmarks = 0, n_residuals = 0, n_var_landmarks = 0;
};

struct Estimator {
    Config cfg;
    int W;
    // window state (estimator.h:121-171)
    om::V3 Ps[MAXW + 1], Vs[MAXW + 1], Bas[MAXW + 1], Bgs[MAXW + 1];
    om::M3 Rs[MAXW + 1];
    double Headers[MAXW + 1];
    Integration *pre_integrations[MAXW + 1];
    om::V3 acc_0, gyr_0, g;
    om::M3 ric;
    om::V3 tic;
    double td;
    bool first_imu = false, initFirstPoseFlag = false, openExEstimation = false, failure_occur = false;
    int frame_count = 0;
    int solver_flag = 0;           // 0 INITIAL, 1 NON_LINEAR
    int marginalization_flag = 0;  // 0 MARGIN_OLD, 1 MARGIN_SECOND_NEW
    double prevTime = -1;
    std::vector<ImuSample> imu_buf;  // queue (front = index imu_head)
    size_t imu_head = 0;
    size_t imu_at_update = 0;   // imu_buf.size() when updateLatestStates last ran: later samples reach predict() through inputIMU (estimator.cpp:1758-1764)
    om::V3 latest_Bg;
    om::M3 last_R, last_R0, back_R0;
    om::V3 last_P, last_P0, back_P0;
    // feature manager
    std::list<Landmark> feature;
    int last_track_num = 0;
    const uint16_t *depth_img = nullptr;
    // flat parameter arrays (estimator.h para_*)
    double para_Pose[MAXW + 1][7], para_SpeedBias[MAXW + 1][9], para_Ex_Pose[7], para_Td;
    std::vector<double> para_Feature;
    // prior, canonical layout (DESIGN.md "prior layout"): [pose slots 0..W-1 (6 each) | speedbias slot 0 (9) | ex (6) | td (1)]
    bool has_prior = false;
    int prior_n = 0;
    om::Mat prior_J;              // n×n  linearized_jacobians
    std::vector<double> prior_r;  // n    linearized_residuals
    // attribution experiment only (deviations & ODEV_QUADRATIC_PRIOR, see ORACLE_DEVIATIONS below): the prior kept as the quadratic form
    // (A, b, c0) = (J^T J, J^T r, |r|^2) without the second eigen-decomposition; prior_J / prior_r are then unused
    om::Mat prior_A;
    std::vector<double> prior_b;
    double prior_c0 = 0;
    int deviations = 0;
    std::vector<double> prior_x0; // keep_block_data in canonical global layout: W*7 + 9 + 7 + 1
    std::vector<uint8_t> prior_present;  // per block: W poses, sb0, ex, td
    SolveStats last_stats;
    int reboot_count = 0;
    // how often a candidate step was cut by the inverse-depth upper bound (estimator.cpp:1282-1297) since construction, and how many
    // bounded landmarks (estimate_flag == 2) entered solves: Ceres would run its projected line search in exactly those steps
    // (DESIGN.md deviation 5); tests/test_oracle_kat.py measures that the canonical workload never gets there
    long bound_clamps = 0, bounded_landmark_solves = 0;
    long line_search_evals = 0, line_search_contractions = 0;   // Armijo line search of bounds-constrained solves (trial evaluations, shortened steps)
    // relocalisation inside optimization() (estimator.h:173-186, estimator.cpp:1307-1346, 1034-1056 / 1071-1090, 1728-1747): the pose of
    // the window frame matched with an old keyframe gets a copy relo_Pose that is optimised against the old keyframe's observations
    bool relocalization_info = false;
    double relo_frame_stamp = 0;
    int relo_frame_index = 0, relo_frame_local_index = 0;
    std::vector<std::array<double, 3>> match_points;   // (x, y) normalised point in the OLD keyframe, z = feature id; ascending id
    om::V3 prev_relo_t;
    om::M3 prev_relo_r;
    double relo_Pose[7] = {0, 0, 0, 0, 0, 0, 1};
    om::M3 drift_correct_r;
    om::V3 drift_correct_t, relo_relative_t;
    om::Q relo_relative_q;
    double relo_relative_yaw = 0;
    int relo_residuals = 0;        // diagnostics: relocalisation factors of the last solve
    void setReloFrame(double frame_stamp, int frame_index, const std::vector<std::array<double, 3>> &match_points_, const om::V3 &relo_t, const om::M3 &relo_r);
    // dynamic initialisation (static_init == 0): every image frame since start-up / the oldest window frame (estimator.h all_image_frame)
    struct ImageFrameO {
        std::map<int, std::array<double, 2>> points;  // feature id -> normalised point
        om::M3 R;                                     // body rotation in the SfM frame
        om::V3 T;                                     // camera position in the SfM frame
        Integration *pre_integration = nullptr;       // from the previous image frame
        bool is_key_frame = false;
    };
    std::map<double, ImageFrameO> all_image_frame;
    Integration *tmp_pre_integration = nullptr;
    double initial_timestamp = 0;
    int init_attempts = 0, init_failures = 0;         // diagnostics
    bool initialStructure();
    bool visualInitialAlignWithDepth();

    explicit Estimator(const Config &c);
    ~Estimator();
    void clearState();
    void inputIMU(double t, const om::V3 &acc, const om::V3 &gyr);
    void predictMotion(double t0, double t1, double R[9], const om::V3 *bg_override = nullptr);
    bool IMUAvailable(double t) const;
    // image: ascending-id map of 7-vectors (x,y,1,u,v,vx,vy); returns 0 ok, 1 need imu, 2 rebooted
    int processImage(std::map<int, std::array<double, 7>> &image, const uint16_t *depth, double stamp);

    // internals
    void processIMU(double dt, const om::V3 &acc, const om::V3 &gyr);
    void initFirstIMUPose(const std::vector<ImuSample> &v);
    bool addFeatureCheckParallax(int frame_count, std::map<int, std::array<double, 7>> &image, double td);
    double compensatedParallax2(const Landmark &l, int frame_count) const;
    int getFeatureCount();
    void triangulateWithDepth();
    void solveGyroscopeBias();
    void initFramePoseByPnP(int frameCnt);
    void latestOdometry(double out[11]) const;
    void vector2double();
    void double2vector();
    void optimization();
    void solve();
    void marginalize_old();
    void marginalize_second_new();
    void slideWindow();
    void slideWindowOld();
    void slideWindowNew();
    void removeBackShiftDepth(const om::M3 &mR, const om::V3 &mP, const om::M3 &nR, const om::V3 &nP);
    void removeBack();
    void removeFront(int frame_count);
    void removeFailures();
    void movingConsistencyCheck();
    bool failureDetection();
    void setDepth(const std::vector<double> &x);
    std::vector<double> getDepthVector();
};

// Attribution experiment (tests/oracle_control.py, DESIGN.md 3 "Round 5"; never used as a checker): environment variable OVIO_DEVIATIONS,
// read when an Estimator is constructed, switches this restatement to the equivalent formulations the HIP path uses, one bit each, so that the
// difference between the two can be assigned to a deviation of DESIGN.md's table.  0 (unset) = the reference's formulation everywhere.
enum {
    ODEV_IMU_WHITEN = 1,        // deviation 8: sqrt_info = chol(cov)^-1 (lower triangular) instead of LLT(cov^-1).L^T (imu_factor.h:80)
    ODEV_QUADRATIC_PRIOR = 2,   // deviation 13: the new prior kept as (A, b, c0), no factorisation J = S^1/2 V^T (marginalization_factor.cpp:293-315)
    ODEV_LANDMARK_ELIM = 4,     // deviation 10: landmarks of the marginalised block eliminated analytically (1/d), eigen pseudo-inverse of the
                                //               remaining 15x15 (6x6) block only (marginalization_factor.cpp:281-291)
    ODEV_PAIR_PROJECTION = 8,   // deviation 11: projection residual / Jacobians through the frame-pair matrices A1 = ric^T Rj^T, A2 = A1 Ri, M = A2 ric
    ODEV_CHOL_PINV = 16,        // (part of deviation 10) the pseudo-inverse of the 15x15 / 6x6 block that remains after the analytic landmark elimination
                                //               is the Cholesky inverse whenever that proves every eigenvalue above 1e-6 (lambda_min >= 1 / |A^-1|_F), so
                                //               nothing would be truncated; only otherwise the eigen-decomposition decides (be_linalg.h spd_inverse_wave16)
};
extern int oracle_deviations;   // the mask of the most recently constructed Estimator (the free factor functions read it)

// factor evaluation (exposed for known-answer tests)
// ProjectionFactor / ProjectionTdFactor: factor/projection_factor.cpp:22-130, projection_td_factor.cpp:34-150
// Jacobians row-major 2×7,2×7,2×7,2×1,2×1 (td ignored when use_td == 0)
void eval_projection(const Config &c, const double *pose_i, const double *pose_j, const double *ex, double inv_dep, double td,
                     const Obs &oi, const Obs &oj, bool use_td, double r[2], double *J_i, double *J_j, double *J_ex,
                     double *J_l, double *J_td);
// IMUFactor::Evaluate factor/imu_factor.h:20-205; Jacobians row-major 15×7, 15×9, 15×7, 15×9
void eval_imu(const Integration &pre, const om::V3 &G, const double *pose_i, const double *sb_i, const double *pose_j,
              const double *sb_j, double r[15], double *J_pi, double *J_sbi, double *J_pj, double *J_sbj);

// ------------------------------------------------------------------------------------ dynamic initialisation (part)
// Visual-inertial alignment of the static_init: 0 branch, oracle/initial.cpp.  One AlignFrame per image
// frame: R = rotation of the BODY in the SfM reference frame, T = position of the CAMERA in it (ImageFrame::R / ::T as filled
// at estimator.cpp:560-574), and the pre-integration from the previous frame (unused for the first one).
struct AlignFrame {
    om::M3 R;
    om::V3 T;
    double sum_dt = 0;
    om::V3 delta_p, delta_v;
};
void tangent_basis(const om::V3 &g0, om::V3 &b, om::V3 &c);
void refine_gravity_with_depth(const std::vector<AlignFrame> &f, const om::V3 &tic, double g_norm, om::V3 &g, std::vector<double> &x);
bool linear_alignment_with_depth(const std::vector<AlignFrame> &f, const om::V3 &tic, double g_norm, om::V3 &g, std::vector<double> &x);
void align_window_to_gravity(int n, om::V3 *Ps, om::M3 *Rs, om::V3 *Vs, const std::vector<double> &x, const om::V3 &tic, om::V3 &g);
struct SfmFeature {  // initial_sfm.h:12-22
    bool state = false;
    int id = 0;
    std::vector<std::pair<int, std::array<double, 2>>> observation;  // (frame, normalised point)
    std::vector<std::pair<int, double>> observation_depth;           // (frame, depth in metres)
    double position[3] = {0, 0, 0};
};
struct SfmStats { int iterations = 0, points = 0; double initial_cost = 0, final_cost = 0; bool converged = false; };
// GlobalSFM::construct: q[i] / T[i] = rotation / position of camera i in the frame of camera l
bool sfm_construct(int frame_num, om::Q *q, om::V3 *T, int l, const om::M3 &relative_R, const om::V3 &relative_T,
                   std::vector<SfmFeature> &sfm_f, std::map<int, om::V3> &sfm_tracked_points, SfmStats *stats = nullptr);
bool sfm_relative_pose(int window_size, const std::vector<SfmFeature> &sfm_f, om::M3 &relative_R, om::V3 &relative_T, int &l);
// SfM front (restated OpenCV routines, see oracle/initial.cpp): camera_point = R X + t
bool solve_pnp_iterative(const std::vector<om::V3> &obj, const std::vector<std::array<double, 2>> &img, om::M3 &R, om::V3 &t);
bool solve_pnp_ransac_epnp(const std::vector<om::V3> &obj, const std::vector<std::array<double, 2>> &img, int max_iters, double thresh,
                           double confidence, om::M3 &R, om::V3 &t, std::vector<uint8_t> &inliers);

// ------------------------------------------------------------------------------------ nodelet-side driver
// Restates the parts of estimator_nodelet.cpp:192-459 (process_tracker) and :462-568 (process) that sit between
// the two entry points: first-image skip, updateID loop, feature-map packaging (track_cnt>1, ascending id),
// init_pub / init_feature skipping.  Frequency control is an input (frame mode, see FrameGate).
struct Pipeline {
    Config cfg;
    Tracker tracker;
    Estimator est;
    bool first_image_flag = true, init_pub = false, init_feature = false;
    double last_image_time = 0;
    int frames_processed = 0;
    int tracker_lag = 0;       // 0: the estimator keeps up with the tracker; 1: it runs one frame behind (see feed)
    om::V3 snap_bg;
    double snap_td;
    explicit Pipeline(const Config &c) : cfg(c), tracker(c), est(c), snap_td(c.td) {}
    // the stream-discontinuity branch of process_tracker (estimator_nodelet.cpp:243-262): first_image_flag = true, last_image_time = 0,
    // feature_buf emptied, estimator.clearState() + setParameter().  trackerData, init_pub and init_feature are left alone.
    void restart();
    // returns 1 if processImage ran for this frame; mode: 0 skip, 1 track only (PUB_THIS_FRAME false), 2 publish
    int feed(const uint8_t *gray, const uint16_t *depth, double t, int mode = 2);
    // the two halves the nodelet runs on separate threads with feature_buf between them (estimator_nodelet.cpp:380-384, 539)
    int track(const uint8_t *gray, double t, int mode, const double *R_in, std::map<int, std::array<double, 7>> &image, const om::V3 *bg = nullptr,
              const double *td = nullptr);
    int process(std::map<int, std::array<double, 7>> &image, const uint16_t *depth, double t);
};

// Colour / depth pairing of EstimatorNodelet::process_tracker (estimator_nodelet.cpp:200-232) applied to two stamp lists in arrival
// order: returns the (colour index, depth index) pairs the two-queue +-3 ms rule forms; thrown[0] / thrown[1] = dropped colour / depth.
std::vector<std::pair<int, int>> pair_color_depth(const std::vector<double> &color, const std::vector<double> &depth, int thrown[2]);

// Frequency control / stream checks of EstimatorNodelet::process_tracker (estimator_nodelet.cpp:94-95, 234-286)
enum { GATE_SKIP = 0, GATE_TRACK = 1, GATE_PUBLISH = 2, GATE_FIRST = 3, GATE_RESET = 4 };
struct FrameGate {
    int freq, frontend_freq;
    bool first_image_flag = true;
    double first_image_time = 0, last_image_time = 0;
    int pub_count = 1, input_count = 0;
    FrameGate(int freq_, int frontend_freq_);
    int step(double t);        // decision for the frame stamped t
    void empty_map(double t);  // call when a PUBLISH frame produced an empty feature map
};

}  // namespace ovio
