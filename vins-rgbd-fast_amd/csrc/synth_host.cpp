// Synthetic workload generator — host side (trajectory, IMU, CPU renderer). See include/vio_synth.h.
#include "synth_scene.h"
#include <string.h>
#include <vector>
#include <mutex>

using namespace vsyn;

namespace {

struct TrajParams { double phi[6], psi0; };

TrajParams traj_params(const vio_synth_config *c, uint64_t seq) {
    TrajParams tp;
    uint64_t s = c->seed + seq;
    for (int i = 0; i < 6; i++) {
        s = mix64(s);
        tp.phi[i] = (double)(s >> 11) * (1.0 / 9007199254740992.0) * 2.0 * M_PI;
    }
    s = mix64(s);
    tp.psi0 = (double)(s >> 11) * (1.0 / 9007199254740992.0) * 2.0 * M_PI;
    return tp;
}

// time warp: tau = 0 while static, then tau' ramps 0 -> 1 (continuous velocity)
double warp(const vio_synth_config *c, double t) {
    double s = t - c->t_static;
    if (s <= 0) return 0.0;
    const double kappa = 2.0;
    return s - (1.0 - exp(-kappa * s)) / kappa;
}

void pose_at(const vio_synth_config *c, const TrajParams &tp, double t, double p[3], double R[9]) {
    double tau = warp(c, t);
    p[0] = 0.0 + 1.5 * sin(0.4 * tau + tp.phi[0]);
    p[1] = 0.0 + 1.0 * sin(0.6 * tau + tp.phi[1]);
    p[2] = 1.5 + 0.3 * sin(0.9 * tau + tp.phi[2]);
    double yaw = tp.psi0 + 0.5 * sin(0.3 * tau + tp.phi[3]);
    double pitch = 0.1 * sin(0.7 * tau + tp.phi[4]);
    double roll = 0.1 * sin(0.7 * tau + tp.phi[5]);
    double cy = cos(yaw), sy = sin(yaw), cp = cos(pitch), sp = sin(pitch), cr = cos(roll), sr = sin(roll);
    // Rz(yaw) Ry(pitch) Rx(roll)
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

double gauss(uint64_t &s) {
    s = mix64(s);
    double u1 = ((double)(s >> 11) + 1.0) * (1.0 / 9007199254740993.0);
    s = mix64(s);
    double u2 = (double)(s >> 11) * (1.0 / 9007199254740992.0);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

}  // namespace

extern "C" {

void vio_synth_config_default(vio_synth_config *c) {
    memset(c, 0, sizeof(*c));
    c->width = 640; c->height = 480;
    c->fx = 604.5821781259577; c->fy = 604.2544712985845; c->cx = 321.2638233484251; c->cy = 239.70969315130674;
    c->k1 = 0.13387871564774004; c->k2 = -0.2731913133377051; c->p1 = 0.0020296263577681264; c->p2 = -0.00044384544608203714;
    const double ric[9] = {0.02629567, -0.00713751, 0.99962873, -0.99934346, 0.02474397, 0.02646484, -0.02492368, -0.99966834, -0.00648216};
    const double tic[3] = {0.17336835, 0.049596, -0.10574841};
    memcpy(c->ric, ric, sizeof(ric));
    memcpy(c->tic, tic, sizeof(tic));
    c->g_norm = 9.805;
    c->imu_rate = 200.0;
    c->cam_rate = 10.0;
    c->t_static = 1.5;
    c->acc_noise = 0.05;
    c->gyr_noise = 0.005;
    c->acc_bias_walk = 1e-3;
    c->gyr_bias_walk = 1e-4;
    c->seed = 0x56494F00ULL;
}

void vio_synth_pose(const vio_synth_config *c, uint64_t seq, double t, double *p, double *R, double *v) {
    TrajParams tp = traj_params(c, seq);
    pose_at(c, tp, t, p, R);
    if (v) {
        const double h = 1e-4;
        double p0[3], p1[3], Rt[9];
        pose_at(c, tp, t - h, p0, Rt);
        pose_at(c, tp, t + h, p1, Rt);
        for (int i = 0; i < 3; i++) v[i] = (p1[i] - p0[i]) / (2 * h);
    }
}

void vio_synth_imu(const vio_synth_config *c, uint64_t seq, int n, double *t, double *acc, double *gyr) {
    TrajParams tp = traj_params(c, seq);
    const double h = 1e-4;
    double dt = 1.0 / c->imu_rate;
    double ba[3] = {0, 0, 0}, bg[3] = {0, 0, 0};
    uint64_t rs = mix64((c->seed + seq) ^ 0xA5A5A5A55A5A5A5AULL);
    for (int k = 0; k < n; k++) {
        double tk = k * dt;
        t[k] = tk;
        double pm[3], p0[3], pp[3], Rm[9], R0[9], Rp[9];
        pose_at(c, tp, tk - h, pm, Rm);
        pose_at(c, tp, tk, p0, R0);
        pose_at(c, tp, tk + h, pp, Rp);
        double a_w[3];
        for (int i = 0; i < 3; i++) a_w[i] = (pp[i] - 2 * p0[i] + pm[i]) / (h * h);
        a_w[2] += c->g_norm;
        // omega_body = vee( R0^T (Rp - Rm) / 2h )
        double dR[9], M[9];
        for (int i = 0; i < 9; i++) dR[i] = (Rp[i] - Rm[i]) / (2 * h);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int q = 0; q < 3; q++) s += R0[q * 3 + i] * dR[q * 3 + j];
                M[i * 3 + j] = s;
            }
        double w[3] = {0.5 * (M[7] - M[5]), 0.5 * (M[2] - M[6]), 0.5 * (M[3] - M[1])};
        for (int i = 0; i < 3; i++) {
            double s = 0;
            for (int q = 0; q < 3; q++) s += R0[q * 3 + i] * a_w[q];
            ba[i] += c->acc_bias_walk * sqrt(dt) * gauss(rs);
            bg[i] += c->gyr_bias_walk * sqrt(dt) * gauss(rs);
            acc[3 * k + i] = s + ba[i] + c->acc_noise * gauss(rs);
            gyr[3 * k + i] = w[i] + bg[i] + c->gyr_noise * gauss(rs);
        }
    }
}

void vio_synth_render_host(const vio_synth_config *c, uint64_t seq, double t, uint8_t *gray, uint16_t *depth_mm) {
    TrajParams tp = traj_params(c, seq);
    double p[3], R[9];
    pose_at(c, tp, t, p, R);
    CamPose cp;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int q = 0; q < 3; q++) s += R[i * 3 + q] * c->ric[q * 3 + j];
            cp.R[i * 3 + j] = (float)s;
        }
        cp.p[i] = (float)(p[i] + R[i * 3 + 0] * c->tic[0] + R[i * 3 + 1] * c->tic[1] + R[i * 3 + 2] * c->tic[2]);
    }
    uint32_t seed = (uint32_t)((c->seed + seq) & 0xFFFFFFFFu);
    // the undistorted ray of every pixel depends on the camera only: computed once per camera (the device renderer keeps the same table)
    static std::mutex mu;
    static std::vector<float> rays;
    static vio_synth_config key;
    static bool valid = false;
    const float *rp;
    {
        std::lock_guard<std::mutex> lk(mu);
        const bool same = valid && key.width == c->width && key.height == c->height && key.fx == c->fx && key.fy == c->fy && key.cx == c->cx &&
                          key.cy == c->cy && key.k1 == c->k1 && key.k2 == c->k2 && key.p1 == c->p1 && key.p2 == c->p2;
        if (!same) {
            // (a new table, never resized in place: another thread may still be rendering from the old one -- it keeps its copy alive below)
            std::vector<float> nr((size_t)c->width * c->height * 2);
            for (int y = 0; y < c->height; y++)
                for (int x = 0; x < c->width; x++) {
                    double rx, ry;
                    syn_lift(c, (double)x, (double)y, &rx, &ry);
                    nr[2 * ((size_t)y * c->width + x)] = (float)rx;
                    nr[2 * ((size_t)y * c->width + x) + 1] = (float)ry;
                }
            rays.swap(nr);
            key = *c;
            valid = true;
        }
    }
    std::vector<float> mine;
    {
        std::lock_guard<std::mutex> lk(mu);
        mine = rays;   // 2.4 MB copy per frame (a frame costs ~100 ms): no lifetime questions when cameras alternate between threads
    }
    rp = mine.data();
    for (int y = 0; y < c->height; y++)
        for (int x = 0; x < c->width; x++) {
            const size_t i = (size_t)y * c->width + x;
            render_pixel(seed, cp, rp[2 * i], rp[2 * i + 1], &gray[i], &depth_mm[i]);
        }
}

}  // extern "C"
