/* Synthetic RGB-D + 200 Hz IMU workload generator — C ABI.
 *
 * The reference has no counterpart: upstream the frames come from a RealSense driver / rosbag through
 * vins_estimator/src/estimator_nodelet.cpp:125-139 (image_callback / depth_callback) and :99-123 (imu_callback).
 * bench.py and the tests use this generator to produce identical inputs for the HIP path and for the oracle.
 * Host functions need no GPU; vio_synth_render_device renders straight into HBM (outside any timed region). */
#ifndef VIO_SYNTH_H
#define VIO_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vio_synth_config {
    int32_t width, height;
    double fx, fy, cx, cy, k1, k2, p1, p2;
    double ric[9]; /* row-major imu<-cam rotation */
    double tic[3];
    double g_norm;
    double imu_rate;      /* Hz (200) */
    double cam_rate;      /* Hz (10)  */
    double t_static;      /* s of stand-still at the start (1.5) */
    double acc_noise;     /* per-sample white noise sigma, m/s^2 */
    double gyr_noise;     /* per-sample white noise sigma, rad/s */
    double acc_bias_walk; /* bias random-walk density */
    double gyr_bias_walk;
    uint64_t seed;        /* base seed; sequence s uses seed + s */
} vio_synth_config;

void vio_synth_config_default(vio_synth_config *c);
/* ground-truth IMU(body) pose in the world frame at time t: p[3], R[9] row-major (world<-body), v[3] world velocity */
void vio_synth_pose(const vio_synth_config *c, uint64_t seq, double t, double *p, double *R, double *v);
/* IMU samples k = 0..n-1 at t_k = k / imu_rate: t[n], acc[3n], gyr[3n] (noise + bias walk included) */
void vio_synth_imu(const vio_synth_config *c, uint64_t seq, int n, double *t, double *acc, double *gyr);
/* CPU renderer: gray[H*W] u8, depth[H*W] u16 millimetres (0 = no return / beyond 10 m) */
void vio_synth_render_host(const vio_synth_config *c, uint64_t seq, double t, uint8_t *gray, uint16_t *depth_mm);
/* GPU renderer: S sequences (seq0 .. seq0+S-1) at time t into device buffers gray[S][H][W], depth[S][H][W].
 * stream is a hipStream_t (may be NULL). Returns 0 on success, a negative vio error code otherwise. */
int vio_synth_render_device(const vio_synth_config *c, int S, uint64_t seq0, double t, uint8_t *d_gray, uint16_t *d_depth_mm,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif
