// Drives the C++ mirror (include/vio_adapter.hpp) the way EstimatorNodelet::process_tracker / process do for one camera, on the
// synthetic workload.  Build:  g++ -std=c++11 -Iinclude examples/adapter_demo.cpp -Lvins-rgbd-fast_amd -lvio_hip -Wl,-rpath,$PWD/vins-rgbd-fast_amd
// Prints one line per NON_LINEAR frame: stamp px py pz  (tests/test_gpu_adapter.py compares it with the ctypes path).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vio_adapter.hpp"
#include "vio_synth.h"

int main(int argc, char **argv) {
    const int seq = argc > 1 ? std::atoi(argv[1]) : 2, n_frames = argc > 2 ? std::atoi(argv[2]) : 22;
    vio_config cfg;
    vio_config_default(&cfg);
    cfg.fix_depth = 0; cfg.depth_max = 10.0;  // the 150-feature setting used by bench.py (canonical_config)
    vio_synth_config sc;
    vio_synth_config_default(&sc);
    const int nimu = (int)(n_frames / sc.cam_rate * sc.imu_rate) + 64;
    std::vector<double> t(nimu), acc(3 * nimu), gyr(3 * nimu);
    vio_synth_imu(&sc, seq, nimu, t.data(), acc.data(), gyr.data());
    std::vector<uint8_t> gray((size_t)cfg.width * cfg.height);
    std::vector<uint16_t> depth((size_t)cfg.width * cfg.height);
    try {
        vio_hip::Estimator estimator(cfg);
        vio_hip::FeatureTracker tracker(estimator);
        estimator.setParameter();
        int k = 0;
        bool first_image_flag = true, init_pub = false, init_feature = false;   // estimator_nodelet.cpp:234-240, :365-377
        for (int f = 0; f < n_frames; f++) {
            const double stamp = f / sc.cam_rate;
            while (k < nimu && t[k] < stamp + 1.5 / sc.imu_rate) { estimator.inputIMU(t[k], &acc[3 * k], &gyr[3 * k]); k++; }  // imu_callback
            vio_synth_render_host(&sc, seq, stamp, gray.data(), depth.data());
            if (first_image_flag) { first_image_flag = false; continue; }   // the first image only sets the time base
            tracker.readImage(gray.data(), stamp);                      // process_tracker (PUB_THIS_FRAME: every frame, freq 0)
            for (unsigned i = 0;; i++) if (!tracker.updateID(i)) break;
            if (!init_pub) { init_pub = true; continue; }               // first published frame is dropped
            if (!init_feature) { init_feature = true; continue; }       // "skip the first detected feature, which doesn't contain optical flow speed"
            int rc = estimator.processImage(depth.data(), stamp);       // process
            if (rc == VIO_NEED_IMU) { std::fprintf(stderr, "frame %d: IMU not available\n", f); continue; }
            if (estimator.solver_flag == vio_hip::Estimator::NON_LINEAR && estimator.last_status().processed) {
                const int W = estimator.WINDOW_SIZE;
                std::printf("%.3f %.9f %.9f %.9f %zu\n", stamp, estimator.Ps[W][0], estimator.Ps[W][1], estimator.Ps[W][2], tracker.ids.size());
            }
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
