// Dynamic (static_init: 0) initialisation, product side: Estimator::initialStructure (vins_estimator/src/estimator/estimator.cpp:384-579)
// with relativePose (:884-920), MotionEstimator::solveRelativeRT_PNP (initial/solve_5pts.cpp:248-294), GlobalSFM::construct
// (initial/initial_sfm.cpp:184-412) and visualInitialAlignWithDepth (estimator.cpp:799-869; initial/initial_aligment.cpp:3-36,
// 170-244, 337-405).  It runs ONCE per sequence, on the host, behind the C ABI: vio_abi.hip collects the image
// frames of a sequence while it is INITIAL, calls run() when the window is full and uploads the result; everything per frame stays
// on the GPU.  Nothing here includes, links or calls oracle/.
#pragma once
#include <array>
#include <map>
#include <vector>

#include "vio_state.h"

namespace vinit {

// one image frame since start-up / since the oldest window frame: Estimator::all_image_frame entry (estimator.h ImageFrame)
struct ImageFrame {
    double stamp = 0;
    std::vector<int> ids;              // feature ids of the frame, ascending
    std::vector<double> xy;            // normalised image points, 2 per feature
    // tmp_pre_integration handed over at estimator.cpp:203-206: linearisation point and the raw samples of the interval
    double lin_acc[3] = {0, 0, 0}, lin_gyr[3] = {0, 0, 0};
    std::vector<double> dt, acc, gyr;  // n, 3 n, 3 n
    double bg_lin[3] = {0, 0, 0};      // gyroscope bias the pre-integration is currently linearised at (Bgs[frame_count] at creation,
                                       // Bgs[0] after a repropagate of an earlier attempt)
    // filled by run()
    dm::m3 R;                          // body rotation in the SfM frame
    dm::v3 T;                          // camera position in the SfM frame
    bool is_key_frame = false;
    // pre-integration results at the current gyroscope bias (repropagate)
    double sum_dt = 0;
    dm::v3 delta_p, delta_v;
    dm::quat delta_q;
    dm::m3 dq_dbg;                     // jacobian block (O_R, O_BG)
};

struct Landmark {                      // FeaturePerId restricted to what the SfM reads (estimator.cpp:425-447)
    int id = 0, start = 0;
    std::vector<std::array<double, 3>> obs;   // (x, y, depth [m]) in consecutive window frames from `start`
};

struct Result {
    bool ok = false;
    bool force_margin_old = false;     // GlobalSFM::construct failed: marginalization_flag = MARGIN_OLD (estimator.cpp:459-463)
    double Ps[VIO_MAXW + 1][3], Rs[VIO_MAXW + 1][9], Vs[VIO_MAXW + 1][3];
    double delta_bg[3] = {0, 0, 0};    // added to every Bgs[i] (solveGyroscopeBias, initial_aligment.cpp:30-33)
    bool set_ba = false;               // accelerometer bias from the mean specific force (estimator.cpp:552-570)
    double Ba[3] = {0, 0, 0};
    double g[3] = {0, 0, 0};
    int stage = 0;                     // diagnostics: where a failed attempt stopped (1 relativePose, 2 SfM, 3 PnP, 4 alignment)
    int ba_iterations = 0, sfm_points = 0;
};

// frames: every image frame (time order); headers: stamps of the W + 1 window frames; bgs0 = Bgs[0] before the attempt.
void run(const vio_config &cfg, int W, const double *headers, const double *bgs0, const double *ric9, const double *tic3,
         std::vector<ImageFrame> &frames, const std::vector<Landmark> &landmarks, Result &out);

// cv::solvePnP(ITERATIVE, useExtrinsicGuess) on normalised points: camera_point = R X + t.  Exposed for Estimator's VO mode
// (FeatureManager::initFramePoseByPnP, feature_manager.cpp:590-642) and for tests.
bool solve_pnp_iterative(const std::vector<dm::v3> &obj, const std::vector<std::array<double, 2>> &img, dm::m3 &R, dm::v3 &t);

// cv::solvePnPRansac (EPnP minimal solver, refit on the inliers) on normalised points with the inlier mask: camera_point = R X + t
bool pnp_ransac_with_inliers(const std::vector<dm::v3> &obj, const std::vector<std::array<double, 2>> &img, int max_iters, double thresh, double confidence,
                             dm::m3 &R, dm::v3 &t, std::vector<uint8_t> &inliers);

bool stage_alignment(int n, const double *frames19, const double *tic3, double g_norm, double *g_out, double *x_out);
bool stage_pnp_ransac_epnp(int n, const double *obj, const double *img, int max_iters, double thresh, double confidence, double *R9, double *t3);
int stage_sfm_window(int window_size, int nf, const int *start, const int *nobs, const double *obs, int *l_out, double *q_out, double *T_out,
                     double *pts_out, double *stats_out);

}  // namespace vinit
