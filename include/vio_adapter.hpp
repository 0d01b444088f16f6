// C++ host-side mirror of the reference's hot-path classes over the C ABI (vio_abi.h) -- SURVEY.md 8b / 8f rank 3.
//
//   vio_hip::Estimator       ~  Estimator       (vins_estimator/src/estimator/estimator.h:29-135)
//   vio_hip::FeatureTracker  ~  FeatureTracker  (vins_estimator/src/feature_tracker/feature_tracker.h:28-90)
//
// Same member names and argument meaning as the reference, plain C++11 types instead of Eigen / OpenCV / ROS so that this
// header compiles anywhere (the reference's nodelet would wrap cv::Mat::data / Eigen::Vector3d::data()).  Differences forced by
// the device-resident design are marked "device:".  One object pair drives one sequence (n_seq = 1), exactly like the nodelet.
// Errors: the reference returns void and aborts through ROS_ASSERT; here failures throw std::runtime_error with vio_last_error().
#ifndef VIO_ADAPTER_HPP
#define VIO_ADAPTER_HPP
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "vio_abi.h"

namespace vio_hip {

struct Point2f { float x, y; };

class Estimator {
  public:
    enum SolverFlag { INITIAL = 0, NON_LINEAR = 1 };                      // estimator.h:62-66
    enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };   // estimator.h:68-72
    static const int MAX_WINDOW = 20;

    // Estimator() + setParameter() (estimator.cpp:9-41); configuration per object instead of the globals of parameters.h
    explicit Estimator(const vio_config &cfg, int imu_capacity = 1 << 14) : cfg_(cfg) {
        h_ = vio_create(&cfg_, 1, imu_capacity);
        if (!h_) throw std::runtime_error(std::string("vio_create: ") + vio_last_error());
        WINDOW_SIZE = cfg_.window_size;
        clearMirror();
    }
    ~Estimator() { vio_destroy(h_); }
    Estimator(const Estimator &) = delete;
    Estimator &operator=(const Estimator &) = delete;

    void setParameter() {}                                    // estimator.cpp:15-41: parameters are bound at construction
    void clearState() { check(vio_reset(h_), "vio_reset"); clearMirror(); }   // estimator.cpp:43-116

    // estimator.cpp:1749-1766
    void inputIMU(double t, const double linearAcceleration[3], const double angularVelocity[3]) {
        check(vio_push_imu(h_, 0, 1, &t, linearAcceleration, angularVelocity), "vio_push_imu");
    }

    // FeatureManager::inputDepth + Estimator::processImage (estimator_nodelet.cpp:534-539, estimator.cpp:156-374).
    // device: the feature map produced by FeatureTracker::readImage stays in HBM, so only the depth image is passed.
    // Returns VIO_OK, VIO_NEED_IMU (IMU has not reached header_stamp + td: the reference busy-waits at :178-183; call again
    // after more inputIMU) or VIO_REBOOTED (failureDetection fired, :345-353).
    int processImage(const uint16_t *depth_mm, double /*header_stamp: taken from the preceding readImage*/) {
        check(vio_process(h_, depth_mm, 0), "vio_process");
        refresh();
        return status_.code;
    }

    // public state read by the publishers (estimator.h:117-135, visualization.cpp:97-538); valid after processImage
    int WINDOW_SIZE;
    SolverFlag solver_flag;
    MarginalizationFlag marginalization_flag;
    int frame_count;
    double Ps[MAX_WINDOW + 1][3], Rs[MAX_WINDOW + 1][9] /* row-major */, Vs[MAX_WINDOW + 1][3], Bas[MAX_WINDOW + 1][3], Bgs[MAX_WINDOW + 1][3];
    double Headers[MAX_WINDOW + 1];
    double tic[3], ric[9], td;
    vio_status last_status() const { return status_; }

    // f_manager.feature (feature_manager.h:63-99) as rows {feature_id, start_frame, n_obs, estimated_depth, estimate_flag, solve_flag, is_dynamic}
    std::vector<double> landmarks() {
        std::vector<double> out(7 * 4096);
        int n = vio_get_landmarks(h_, 0, 4096, out.data());
        check(n, "vio_get_landmarks");
        out.resize(7 * (size_t)(n < 4096 ? n : 4096));
        return out;
    }
    vio_batch *handle() { return h_; }

  private:
    friend class FeatureTracker;
    static void check(int rc, const char *what) {
        if (rc < 0) throw std::runtime_error(std::string(what) + ": " + vio_last_error());
    }
    void clearMirror() {
        solver_flag = INITIAL; marginalization_flag = MARGIN_OLD; frame_count = 0; td = cfg_.td;
        std::memset(Ps, 0, sizeof(Ps)); std::memset(Rs, 0, sizeof(Rs)); std::memset(Vs, 0, sizeof(Vs));
        std::memset(Bas, 0, sizeof(Bas)); std::memset(Bgs, 0, sizeof(Bgs)); std::memset(Headers, 0, sizeof(Headers));
        std::memset(&status_, 0, sizeof(status_));
        for (int i = 0; i <= MAX_WINDOW; i++) Rs[i][0] = Rs[i][4] = Rs[i][8] = 1.0;
        std::memcpy(tic, cfg_.tic, sizeof(tic)); std::memcpy(ric, cfg_.ric, sizeof(ric));
    }
    void refresh() {
        check(vio_get_status(h_, 0, &status_), "vio_get_status");
        solver_flag = status_.solver_flag ? NON_LINEAR : INITIAL;
        marginalization_flag = status_.marginalization_flag ? MARGIN_SECOND_NEW : MARGIN_OLD;
        frame_count = status_.frame_count;
        std::vector<double> w(17 * (size_t)(WINDOW_SIZE + 1));
        check(vio_get_window(h_, 0, w.data()), "vio_get_window");
        for (int i = 0; i <= WINDOW_SIZE; i++) {
            const double *r = &w[17 * (size_t)i];
            for (int k = 0; k < 3; k++) { Ps[i][k] = r[k]; Vs[i][k] = r[7 + k]; Bas[i][k] = r[10 + k]; Bgs[i][k] = r[13 + k]; }
            const double qw = r[3], qx = r[4], qy = r[5], qz = r[6];  // Eigen::Quaterniond::toRotationMatrix
            double *R = Rs[i];
            R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qz * qw);     R[2] = 2 * (qx * qz + qy * qw);
            R[3] = 2 * (qx * qy + qz * qw);     R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qx * qw);
            R[6] = 2 * (qx * qz - qy * qw);     R[7] = 2 * (qy * qz + qx * qw);     R[8] = 1 - 2 * (qx * qx + qy * qy);
            Headers[i] = r[16];
        }
        double ex[13];
        check(vio_get_extrinsic(h_, 0, ex), "vio_get_extrinsic");
        std::memcpy(tic, ex, sizeof(tic)); std::memcpy(ric, ex + 3, sizeof(ric)); td = ex[12];
    }
    vio_config cfg_;
    vio_batch *h_;
    vio_status status_;
};

class FeatureTracker {
  public:
    // device: the tracker state lives in the same handle as the estimator (predictMotion reads its IMU buffer and biases)
    explicit FeatureTracker(Estimator &estimator) : e_(estimator) {}

    // feature_tracker.h:36-37 readImage(const cv::Mat &_img, double _cur_time, const Matrix3d &_relative_R).
    // device: relative_R is computed on the GPU by predictMotion (estimator.cpp:1790-1860) from the handle's IMU buffer, the
    // argument is accepted for signature compatibility and ignored.  PUB_THIS_FRAME (global in the reference) is an argument.
    void readImage(const uint8_t *img /* ROW x COL mono8, contiguous */, double cur_time, const double * /*relative_R*/ = nullptr,
                   bool PUB_THIS_FRAME = true) {
        Estimator::check(vio_track(e_.h_, img, &cur_time, PUB_THIS_FRAME ? 1 : 0, 0), "vio_track");
        const int cap = 4096;
        ids.assign(cap, 0); track_cnt.assign(cap, 0);
        cur_pts.assign(cap, Point2f()); cur_un_pts.assign(cap, Point2f()); pts_velocity.assign(cap, Point2f());
        int n = vio_get_tracks(e_.h_, 0, cap, ids.data(), track_cnt.data(), &cur_pts[0].x, &cur_un_pts[0].x, &pts_velocity[0].x);
        Estimator::check(n, "vio_get_tracks");
        n = n < cap ? n : cap;
        ids.resize(n); track_cnt.resize(n); cur_pts.resize(n); cur_un_pts.resize(n); pts_velocity.resize(n);
        cur_time_ = cur_time;
    }
    // feature_tracker.h:41 -- the nodelet loops `for (i = 0;; i++) if (!updateID(i)) break;` (estimator_nodelet.cpp:324-330).
    // device: ids are already assigned inside readImage; the loop contract (true while i < ids.size()) is kept.
    bool updateID(unsigned int i) const { return i < ids.size(); }

    std::vector<Point2f> cur_pts, cur_un_pts, pts_velocity;   // feature_tracker.h:66-72
    std::vector<int> ids, track_cnt;
    double cur_time() const { return cur_time_; }

  private:
    Estimator &e_;
    double cur_time_ = 0;
};

}  // namespace vio_hip
#endif
