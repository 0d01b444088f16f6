// C++ host-side mirror of the reference's hot-path classes over the C ABI (vio_abi.h) -- SURVEY.md 8b / 8f rank 3.
//
//   vio_hip::Estimator       ~  Estimator       (vins_estimator/src/estimator/estimator.h:29-135)
//   vio_hip::FeatureTracker  ~  FeatureTracker  (vins_estimator/src/feature_tracker/feature_tracker.h:28-90)
//   vio_hip::FrameGate       ~  the stream checks + frequency control at the top of EstimatorNodelet::process_tracker
//                               (vins_estimator/src/estimator_nodelet.cpp:94-95, 234-286)
//
// Same member names, signatures and argument meaning as the reference, plain C++11 types instead of Eigen / OpenCV / ROS so that
// this header compiles anywhere (the reference's nodelet would wrap cv::Mat::data / Eigen::Vector3d::data()).  Differences forced
// by the device-resident design are marked "device:".  One object pair drives one sequence (n_seq = 1), exactly like the nodelet.
// Errors: the reference returns void and aborts through ROS_ASSERT; here failures throw std::runtime_error with vio_last_error().
#ifndef VIO_ADAPTER_HPP
#define VIO_ADAPTER_HPP
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vio_abi.h"

namespace vio_hip {

struct Point2f { float x, y; };
typedef std::array<double, 7> Vector7d;                      // Eigen::Matrix<double, 7, 1>: x y z u v vx vy
typedef std::map<int, Vector7d> FeatureMap;                  // map<int, Eigen::Matrix<double, 7, 1>> (estimator.h:46)

// The colour / depth pairing at the top of process_tracker (estimator_nodelet.cpp:200-232): img_buf and depth_buf are FIFO queues
// filled by the two image callbacks; while both hold a message the front stamps are compared -- colour more than 3 ms older than depth:
// "throw color"; more than 3 ms newer: "throw depth"; otherwise both are popped as a pair.  T = whatever the caller queues per message.
template <class T> class ColorDepthSync {
  public:
    void pushColor(double stamp, const T &msg) { img_buf.push_back(std::make_pair(stamp, msg)); }      // img_callback (:128-140)
    void pushDepth(double stamp, const T &msg) { depth_buf.push_back(std::make_pair(stamp, msg)); }    // depth_callback (:142-154)
    // true: color / depth (and time_color) hold the next synchronised pair; false: one queue ran dry (the nodelet waits there)
    bool pop(T &color, T &depth, double &time_color) {
        while (!img_buf.empty() && !depth_buf.empty()) {
            const double tc = img_buf.front().first, td = depth_buf.front().first;
            if (tc < td - 0.003) { img_buf.pop_front(); ++thrown_color; }
            else if (tc > td + 0.003) { depth_buf.pop_front(); ++thrown_depth; }
            else {
                color = img_buf.front().second; depth = depth_buf.front().second; time_color = tc;
                img_buf.pop_front(); depth_buf.pop_front();
                return true;
            }
        }
        return false;
    }
    std::deque<std::pair<double, T> > img_buf, depth_buf;
    int thrown_color = 0, thrown_depth = 0;
};

// Stream checks + frequency control of process_tracker (estimator_nodelet.cpp:234-286).  step(t) returns what the nodelet does
// with the frame stamped t: FIRST (only sets the time base), RESET (stream discontinuity: caller restarts the estimator,
// :243-262), SKIP ("Skip this frame", before readImage), TRACK (readImage with PUB_THIS_FRAME false) or PUBLISH.
class FrameGate {
  public:
    enum Decision { SKIP = VIO_FRAME_SKIP, TRACK = VIO_FRAME_TRACK, PUBLISH = VIO_FRAME_PUBLISH, FIRST = 3, RESET = 4 };
    FrameGate(int FREQ, int FRONTEND_FREQ) : freq_(FREQ == 0 ? 100 : FREQ), frontend_freq_(FRONTEND_FREQ) {}   // parameters.cpp:133-134
    Decision step(double time_color) {
        if (first_image_flag) {
            first_image_flag = false;
            first_image_time = time_color;
            last_image_time = time_color;
            return FIRST;
        }
        if (time_color - last_image_time > 1.0 || time_color < last_image_time) {
            first_image_flag = true;
            last_image_time = 0;
            pub_count = 1;
            return RESET;
        }
        if (std::round(1.0 * input_count / (time_color - first_image_time)) > frontend_freq_) return SKIP;
        ++input_count;
        bool pub = false;
        if (std::round(1.0 * pub_count / (time_color - first_image_time)) <= freq_) {
            pub = true;
            if (std::fabs(1.0 * pub_count / (time_color - first_image_time) - freq_) < 0.01 * freq_) {
                first_image_time = time_color;
                pub_count = 0;
                input_count = 0;
            }
        }
        last_image_time = time_color;
        if (pub) pub_count++;
        return pub ? PUBLISH : TRACK;
    }
    // a PUBLISH frame whose feature map came out empty restarts the rate window (estimator_nodelet.cpp:386-392)
    void emptyMap(double time_color) { first_image_time = time_color; pub_count = 0; input_count = 0; }
    bool first_image_flag = true;
    double first_image_time = 0, last_image_time = 0;
    int pub_count = 1, input_count = 0;

  private:
    int freq_, frontend_freq_;
};

class Estimator {
  public:
    enum SolverFlag { INITIAL = 0, NON_LINEAR = 1 };                      // estimator.h:62-66
    enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };   // estimator.h:68-72
    static const int MAX_WINDOW = 20;

    // Estimator() + setParameter() (estimator.cpp:9-41); configuration per object instead of the globals of parameters.h
    explicit Estimator(const vio_config &cfg, int imu_capacity = 1 << 14) : f_manager(*this), cfg_(cfg) {
        h_ = vio_create(&cfg_, 1, imu_capacity);
        if (!h_) throw std::runtime_error(std::string("vio_create: ") + vio_last_error());
        WINDOW_SIZE = cfg_.window_size;
        clearMirror();
    }
    ~Estimator() { vio_destroy(h_); }
    Estimator(const Estimator &) = delete;
    Estimator &operator=(const Estimator &) = delete;

    void setParameter() {}                                    // estimator.cpp:15-41: parameters are bound at construction
    void clearState() { check(vio_reset_seq(h_, 0), "vio_reset_seq"); clearMirror(); }   // estimator.cpp:43-116: the estimator side only -- featureTracker keeps its points / ids / previous image

    // estimator.cpp:1749-1766
    void inputIMU(double t, const double linearAcceleration[3], const double angularVelocity[3]) {
        check(vio_push_imu(h_, 0, 1, &t, linearAcceleration, angularVelocity), "vio_push_imu");
    }

    // Matrix3d predictMotion(double t0, double t1) (estimator.h:56, estimator.cpp:1790-1860): relative_R row-major
    void predictMotion(double t0, double t1, double relative_R[9]) { check(vio_predict_motion(h_, 0, t0, t1, relative_R), "vio_predict_motion"); }

    // the IMU-rate pose pubLatestOdometry publishes from inputIMU (estimator.cpp:1760-1765): latest_time, latest_P, latest_Q (w x y z), latest_V
    void latestOdometry(double out11[11]) { check(vio_get_latest_odometry(h_, 0, out11), "vio_get_latest_odometry"); }

    // void setReloFrame(double _frame_stamp, int _frame_index, vector<Vector3d> &_match_points, Vector3d _relo_t, Matrix3d _relo_r)
    // (estimator.h:48-49, estimator.cpp:1728-1747): match_points = (x, y, feature id), ascending id; relo_r row-major.  The next
    // processImage optimises relo_Pose with the relocalisation factors (estimator.cpp:1307-1346); relocalization() then holds what
    // pubRelocalization and the pose graph read (visualization.cpp:454-538).
    void setReloFrame(double frame_stamp, int frame_index, const std::vector<std::array<double, 3> > &match_points, const double relo_t[3],
                      const double relo_r[9]) {
        check(vio_set_relo_frame(h_, 0, frame_stamp, frame_index, (int)match_points.size(), match_points.empty() ? nullptr : match_points[0].data(),
                                 relo_t, relo_r), "vio_set_relo_frame");
    }
    struct Relocalization {
        double relo_relative_t[3], relo_relative_q[4] /* w x y z */, relo_relative_yaw, drift_correct_t[3], drift_correct_r[9], relo_Pose[7];
        bool relocalization_info;
        int relo_frame_local_index, factors;
    };
    Relocalization relocalization() {
        double o[30];
        check(vio_get_relo(h_, 0, o), "vio_get_relo");
        Relocalization r;
        std::memcpy(r.relo_relative_t, o, 24); std::memcpy(r.relo_relative_q, o + 3, 32); r.relo_relative_yaw = o[7];
        std::memcpy(r.drift_correct_t, o + 8, 24); std::memcpy(r.drift_correct_r, o + 11, 72); std::memcpy(r.relo_Pose, o + 20, 56);
        r.relocalization_info = o[27] != 0; r.relo_frame_local_index = (int)o[28]; r.factors = (int)o[29];
        return r;
    }

    // FeatureManager::inputDepth (feature_manager.cpp:43-46): the nodelet calls estimator.f_manager.inputDepth(depth) right before
    // processImage (estimator_nodelet.cpp:537-539); the pointer must stay valid until processImage returns
    struct FeatureManagerMirror {
        explicit FeatureManagerMirror(Estimator &e) : e_(e) {}
        void inputDepth(const uint16_t *depth_mm /* ROW x COL CV_16UC1, contiguous */) { e_.depth_ = depth_mm; }
      private:
        Estimator &e_;
    } f_manager;

    // void processImage(const map<int, Eigen::Matrix<double, 7, 1>> &image, const std_msgs::Header &header) (estimator.h:46,
    // estimator.cpp:156-374).  header = stamp in seconds.  Returns VIO_OK, VIO_NEED_IMU (IMU has not reached header + td: the
    // reference busy-waits at :178-183; nothing was consumed, call again after more inputIMU) or VIO_REBOOTED (failureDetection
    // fired, :345-353).  The map is whatever the caller popped from its feature_buf: the tracker may be any number of frames ahead.
    int processImage(const FeatureMap &image, double header) {
        if (!depth_) throw std::runtime_error("processImage: f_manager.inputDepth was not called");
        ids_.clear(); obs_.clear();
        for (FeatureMap::const_iterator it = image.begin(); it != image.end(); ++it) {   // std::map order = ascending feature id
            ids_.push_back(it->first);
            obs_.insert(obs_.end(), it->second.begin(), it->second.end());
        }
        check(vio_process_obs(h_, 0, (int)ids_.size(), ids_.data(), obs_.data(), depth_, header), "vio_process_obs");
        refresh();
        return status_.code;
    }
    // device: short cut without the host round trip of the feature map -- inputDepth + processImage on the map the last
    // FeatureTracker::readImage packaged in HBM (only valid when the estimator does not lag the tracker)
    int processLastTracked(const uint16_t *depth_mm) {
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

    // f_manager.feature (feature_manager.h:63-99) as rows of 12: {feature_id, start_frame, n_obs, estimated_depth, estimate_flag,
    // solve_flag, is_dynamic, feature_per_frame[0].point (3), feature_per_frame[0].depth, feature_per_frame.back().depth} --
    // everything pubPointCloud reads (visualization.cpp:333-395)
    std::vector<double> landmarks() {
        std::vector<double> out(12 * 4096);
        int n = vio_get_landmarks_ex(h_, 0, 4096, out.data());
        check(n, "vio_get_landmarks_ex");
        out.resize(12 * (size_t)(n < 4096 ? n : 4096));
        return out;
    }
    // the world points of pubPointCloud (visualization.cpp:333-366): xyz per landmark that passes its filters
    std::vector<double> pointCloud() {
        std::vector<double> lm = landmarks(), out;
        for (size_t k = 0; k + 12 <= lm.size(); k += 12) {
            const double *q = &lm[k];
            const int start = (int)q[1], used = (int)q[2];
            if (q[6] != 0) continue;
            if (!(used >= 2 && start < WINDOW_SIZE - 2)) continue;
            if (start > WINDOW_SIZE * 3.0 / 4.0 || (int)q[5] != 1) continue;
            const double d = q[10] == 0 ? q[3] : q[10];
            const double pc[3] = {q[7] * d, q[8] * d, q[9] * d};
            double pi[3], pw[3];
            for (int r = 0; r < 3; r++) pi[r] = ric[3 * r] * pc[0] + ric[3 * r + 1] * pc[1] + ric[3 * r + 2] * pc[2] + tic[r];
            for (int r = 0; r < 3; r++) pw[r] = Rs[start][3 * r] * pi[0] + Rs[start][3 * r + 1] * pi[1] + Rs[start][3 * r + 2] * pi[2] + Ps[start][r];
            out.insert(out.end(), pw, pw + 3);
        }
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
        depth_ = nullptr;
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
    const uint16_t *depth_ = nullptr;
    std::vector<int32_t> ids_;
    std::vector<double> obs_;
};

class FeatureTracker {
  public:
    // device: the tracker state lives in the same handle as the estimator (one HBM-resident sequence)
    explicit FeatureTracker(Estimator &estimator) : e_(estimator) {}

    // void readImage(const cv::Mat &_img, double _cur_time, const Matrix3d &_relative_R = Matrix3d::Identity())
    // (feature_tracker.h:36-37).  relative_R (row-major) is honoured exactly as given (predictPtsInNextFrame, feature_tracker.cpp:
    // 595-608); relative_R == nullptr lets the device compute Estimator::predictMotion(last image time, cur_time + td) itself,
    // which saves the round trip.  PUB_THIS_FRAME is a global in the reference (parameters.h:60) and an argument here.
    void readImage(const uint8_t *img /* ROW x COL mono8, contiguous */, double cur_time, const double *relative_R = nullptr,
                   bool PUB_THIS_FRAME = true) {
        const uint8_t mode = PUB_THIS_FRAME ? VIO_FRAME_PUBLISH : VIO_FRAME_TRACK;
        Estimator::check(vio_track_ex(e_.h_, img, &cur_time, &mode, relative_R, 0), "vio_track_ex");
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
