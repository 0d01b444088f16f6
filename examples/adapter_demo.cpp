// Drives the C++ mirror (include/vio_adapter.hpp) the way EstimatorNodelet does for one camera, on the synthetic workload:
// process_tracker (stream checks, frequency control, predictMotion + readImage, updateID loop, feature-map packaging into
// feature_buf, estimator_nodelet.cpp:192-459) and process (pop feature_buf, inputDepth, processImage, :462-549), with the
// estimator allowed to lag the tracker by `lag` frames like the reference's second thread does.
// Build:  g++ -std=c++11 -Iinclude examples/adapter_demo.cpp -Lvins-rgbd-fast_amd -lvio_hip -Wl,-rpath,$PWD/vins-rgbd-fast_amd
// Usage:  adapter_demo [seq] [n_frames] [cam_rate] [freq] [frontend_freq] [lag] [use_imu] [depth_jitter]
//         use_imu 0 = the reference's `imu: 0`; depth_jitter 1 = the depth stamps are offset from the colour stamps by a fixed pattern
//         (some beyond +-3 ms) so that the colour / depth pairing of process_tracker (:206-232) drops frames through both branches
// Prints one line per NON_LINEAR frame: stamp px py pz n_tracks  (tests/test_gpu_adapter.py compares it with the oracle).
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <vector>

#include "vio_adapter.hpp"
#include "vio_synth.h"

struct QueuedFrame {  // feature_buf entry: ((header, depth_msg), image)
    double header;
    std::vector<uint16_t> depth;
    vio_hip::FeatureMap image;
};

int main(int argc, char **argv) {
    const int seq = argc > 1 ? std::atoi(argv[1]) : 2, n_frames = argc > 2 ? std::atoi(argv[2]) : 22;
    const double cam_rate = argc > 3 ? std::atof(argv[3]) : 0.0;
    const int FREQ = argc > 4 ? std::atoi(argv[4]) : 10, FRONTEND_FREQ = argc > 5 ? std::atoi(argv[5]) : 30;
    const size_t lag = argc > 6 ? (size_t)std::atoi(argv[6]) : 0;
    const bool USE_IMU = argc > 7 ? std::atoi(argv[7]) != 0 : true;
    const bool depth_jitter = argc > 8 ? std::atoi(argv[8]) != 0 : false;
    static const double kDepthOffset[8] = {0.0, 0.001, -0.002, 0.0045, 0.0, 0.0029, -0.0035, 0.002};
    vio_config cfg;
    vio_config_default(&cfg);
    cfg.fix_depth = 0; cfg.depth_max = 10.0;  // the 150-feature setting used by bench.py (canonical_config)
    if (!USE_IMU) { cfg.use_imu = 0; cfg.lk_max_level = 3; cfg.fix_depth = 1; }   // imu: 0 (parameters.cpp:98-107), 4-level LK (feature_tracker.cpp:307-311)
    vio_synth_config sc;
    vio_synth_config_default(&sc);
    if (cam_rate > 0) sc.cam_rate = cam_rate;
    if (!USE_IMU) sc.t_static = 0.0;   // no stationary prefix: VO starts from the first frame (estimator.cpp:581-626)
    const int nimu = (int)(n_frames / sc.cam_rate * sc.imu_rate) + 64;
    std::vector<double> t(nimu), acc(3 * nimu), gyr(3 * nimu);
    vio_synth_imu(&sc, seq, nimu, t.data(), acc.data(), gyr.data());
    std::vector<uint8_t> gray((size_t)cfg.width * cfg.height);
    std::vector<uint16_t> depth((size_t)cfg.width * cfg.height);
    try {
        vio_hip::Estimator estimator(cfg);
        vio_hip::FeatureTracker tracker(estimator);
        vio_hip::FrameGate gate(FREQ, FRONTEND_FREQ);
        estimator.setParameter();
        std::deque<QueuedFrame> feature_buf;
        bool init_pub = false, init_feature = false;   // estimator_nodelet.cpp:365-377
        int k = 0;
        vio_hip::ColorDepthSync<int> sync;             // img_buf / depth_buf of the nodelet; a message = its frame index
        auto process = [&](bool drain) {   // EstimatorNodelet::process: one queued frame per call unless draining
            while (feature_buf.size() > (drain ? 0 : lag)) {
                QueuedFrame &f = feature_buf.front();
                estimator.f_manager.inputDepth(f.depth.data());                 // :537
                int rc = estimator.processImage(f.image, f.header);             // :539
                if (rc == VIO_NEED_IMU) { std::fprintf(stderr, "stamp %.3f: IMU not available\n", f.header); return; }
                if (estimator.solver_flag == vio_hip::Estimator::NON_LINEAR && estimator.last_status().processed) {
                    const int W = estimator.WINDOW_SIZE;
                    std::printf("%.4f %.9f %.9f %.9f %zu\n", f.header, estimator.Ps[W][0], estimator.Ps[W][1], estimator.Ps[W][2], f.image.size());
                }
                feature_buf.pop_front();
            }
        };
        for (int fm = 0; fm < n_frames; fm++) {
            // img_callback / depth_callback (:128-154): both messages of frame fm arrive, then process_tracker pairs what it can (:200-232)
            sync.pushColor(fm / sc.cam_rate, fm);
            sync.pushDepth(fm / sc.cam_rate + (depth_jitter ? kDepthOffset[fm % 8] : 0.0), fm);
            int fi = 0, fd = 0;
            double time_color = 0;
            if (!sync.pop(fi, fd, time_color)) continue;
            while (USE_IMU && k < nimu && t[k] < time_color + 1.5 / sc.imu_rate) { estimator.inputIMU(t[k], &acc[3 * k], &gyr[3 * k]); k++; }  // imu_callback
            vio_synth_render_host(&sc, seq, time_color, gray.data(), depth.data());   // (the depth image of message fd; same pixels, its stamp only decides the pairing)
            // ---- process_tracker
            const double last_image_time = gate.last_image_time;
            const vio_hip::FrameGate::Decision d = gate.step(time_color);
            if (d == vio_hip::FrameGate::FIRST) continue;
            if (d == vio_hip::FrameGate::RESET) {          // :243-262
                feature_buf.clear();
                estimator.clearState();
                estimator.setParameter();
                continue;                                  // (init_pub / init_feature and the tracker keep their state, like upstream)
            }
            if (d == vio_hip::FrameGate::SKIP) continue;
            const bool PUB_THIS_FRAME = d == vio_hip::FrameGate::PUBLISH;
            if (USE_IMU) {
                double relative_R[9];
                estimator.predictMotion(last_image_time, time_color + estimator.td, relative_R);   // :309-313
                tracker.readImage(gray.data(), time_color, relative_R, PUB_THIS_FRAME);
            } else
                tracker.readImage(gray.data(), time_color, nullptr, PUB_THIS_FRAME);              // :315-316
            for (unsigned i = 0;; i++) if (!tracker.updateID(i)) break;                      // :324-330
            if (PUB_THIS_FRAME) {
                vio_hip::FeatureMap image;                                                   // :336-363
                for (size_t j = 0; j < tracker.ids.size(); j++)
                    if (tracker.track_cnt[j] > 1) {
                        vio_hip::Vector7d v = {{tracker.cur_un_pts[j].x, tracker.cur_un_pts[j].y, 1.0, tracker.cur_pts[j].x, tracker.cur_pts[j].y,
                                                tracker.pts_velocity[j].x, tracker.pts_velocity[j].y}};
                        image[tracker.ids[j]] = v;
                    }
                if (!init_pub) init_pub = true;                                               // first published frame is dropped
                else if (!init_feature) init_feature = true;                                  // "skip the first detected feature ..."
                else if (!image.empty()) {
                    QueuedFrame q;
                    q.header = time_color; q.depth = depth; q.image.swap(image);
                    feature_buf.push_back(std::move(q));
                } else
                    gate.emptyMap(time_color);
            }
            process(false);
        }
        process(true);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
