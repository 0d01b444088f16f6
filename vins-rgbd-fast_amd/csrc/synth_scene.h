// Synthetic RGB-D + IMU workload generator: scene, camera and trajectory definitions shared by the
// host generator (synth_host.cpp) and the device renderer (synth_render.hip). This is workload data generation for
// tests and bench.py — it is not part of the reference's hot path and has no upstream counterpart.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VIO_HD __host__ __device__ __forceinline__
#else
#define VIO_HD inline
#endif

#include "../../include/vio_synth.h"

namespace vsyn {

VIO_HD uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
VIO_HD uint32_t hash4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    uint64_t h = mix64(((uint64_t)a << 32) | b);
    h = mix64(h ^ (((uint64_t)c << 32) | d));
    return (uint32_t)(h >> 32);
}
VIO_HD float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// Deterministic transcendental functions of the texture (round 6: ONE renderer).  sinf / expf of two math libraries (glibc on the host,
// ocml on the device) round differently in the last place, which flipped 4 of 29.5 M rendered pixels between vio_synth_render_host and
// vio_synth_render_device.  These forms use only IEEE-754 double + - * (no contraction: -ffp-contract=off), a round-to-nearest by the
// 1.5 * 2^52 constant and an exponent built from bits, so every conforming implementation returns the same float.
VIO_HD float sin_det(float xf) {   // |x| < 2^20 * pi / 2; the scheme of dmath.h's sincos_det (Cody-Waite by pi / 2, fdlibm kernel polynomials)
    const double x = (double)xf;
    const double kd = (x * 0.63661977236758134308 + 6755399441055744.0) - 6755399441055744.0;
    const double r = ((x - kd * 1.57079632673412561417e+00) - kd * 6.07710050630396597660e-11) - kd * 2.02226624879595063154e-21;
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 +
                      z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double s0 = r + (r * z) * ps;
    const double c0 = (1.0 - 0.5 * z) + (z * z) * pc;
    const int q = (int)((long long)kd & 3);
    return (float)(q == 0 ? s0 : (q == 1 ? c0 : (q == 2 ? -s0 : -c0)));
}
VIO_HD float exp_det(float xf) {   // -80 < x <= 0 (the dots evaluate it on (-12, 0])
    const double x = (double)xf;
    const double kd = (x * 1.44269504088896338700e+00 + 6755399441055744.0) - 6755399441055744.0;
    const double r = (x - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;   // |r| <= ln 2 / 2
    // Taylor polynomial of degree 13 (truncation 0.347^14 / 14! = 4e-18 relative), Horner
    double p = 1.0 / 6227020800.0;
    p = 1.0 / 479001600.0 + r * p;
    p = 1.0 / 39916800.0 + r * p;
    p = 1.0 / 3628800.0 + r * p;
    p = 1.0 / 362880.0 + r * p;
    p = 1.0 / 40320.0 + r * p;
    p = 1.0 / 5040.0 + r * p;
    p = 1.0 / 720.0 + r * p;
    p = 1.0 / 120.0 + r * p;
    p = 1.0 / 24.0 + r * p;
    p = 1.0 / 6.0 + r * p;
    p = 0.5 + r * p;
    p = 1.0 + r * p;
    p = 1.0 + r * p;
    union { uint64_t u; double d; } sc;
    sc.u = (uint64_t)(1023 + (int)kd) << 52;   // 2^k, k in [-116, 0]: a normal double, the product is exact up to the one rounding
    return (float)(p * sc.d);
}

// room: x in [-4,4], y in [-3,3], z in [0,3]
VIO_HD float smooth01(float x) {
    x = x < 0.f ? 0.f : (x > 1.f ? 1.f : x);
    return x * x * (3.f - 2.f * x);
}
VIO_HD float scene_texture(uint32_t seed, int wall, float u, float v) {
    float val = 128.f;
    const float edge[3] = {0.4f, 0.2f, 0.1f};
    const float amp[3] = {50.f, 30.f, 15.f};
    for (int o = 0; o < 3; o++) {
        float e = edge[o];
        float fu = u / e - 0.5f, fv = v / e - 0.5f;
        float flu = floorf(fu), flv = floorf(fv);
        int iu = (int)flu, iv = (int)flv;
        float b = 0.012f / e;
        float wu = smooth01((fu - flu - 0.5f) / b + 0.5f), wv = smooth01((fv - flv - 0.5f) / b + 0.5f);
        uint32_t key = (uint32_t)(wall * 4 + o) ^ (seed * 2654435761u);
        float v00 = (u01(hash4(key, (uint32_t)iu, (uint32_t)iv, 1u)) * 2.f - 1.f) * amp[o];
        float v10 = (u01(hash4(key, (uint32_t)(iu + 1), (uint32_t)iv, 1u)) * 2.f - 1.f) * amp[o];
        float v01 = (u01(hash4(key, (uint32_t)iu, (uint32_t)(iv + 1), 1u)) * 2.f - 1.f) * amp[o];
        float v11 = (u01(hash4(key, (uint32_t)(iu + 1), (uint32_t)(iv + 1), 1u)) * 2.f - 1.f) * amp[o];
        float top = v00 + (v10 - v00) * wu, bot = v01 + (v11 - v01) * wu;
        val += top + (bot - top) * wv;
    }
    // Gaussian dots, one candidate per 0.1 m cell
    {
        const float cell = 0.1f;
        float cu = floorf(u / cell), cv = floorf(v / cell);
        uint32_t key = (uint32_t)(wall + 64) ^ (seed * 2246822519u);
        for (int dv = -1; dv <= 1; dv++)
            for (int du = -1; du <= 1; du++) {
                int iu = (int)cu + du, iv = (int)cv + dv;
                uint32_t h0 = hash4(key, (uint32_t)iu, (uint32_t)iv, 2u);
                if ((h0 & 0xFF) >= 154) continue;  // ~60 % of the cells carry a dot
                uint32_t h1 = hash4(key, (uint32_t)iu, (uint32_t)iv, 3u), h2 = hash4(key, (uint32_t)iu, (uint32_t)iv, 4u);
                float ox = ((float)iu + u01(h1)) * cell, oy = ((float)iv + u01(h2)) * cell;
                float sg = 0.005f + 0.01f * u01(h0 ^ 0x5bd1e995u);
                float a = (h0 & 0x100) ? 70.f : -70.f;
                float dx = u - ox, dy = v - oy;
                float q = (dx * dx + dy * dy) / (2.f * sg * sg);
                if (q < 12.f) val += a * exp_det(-q);
            }
    }
    val += 10.f * sin_det(2.1f * u + (float)wall) * sin_det(1.7f * v + 0.5f * (float)wall);
    return val;
}

// pinhole radial-tangential model (same model as camera_model/src/camera_models/PinholeCamera.cc, used here to build rays)
VIO_HD void syn_distort(const vio_synth_config *c, double x, double y, double *dx, double *dy) {
    double mx2 = x * x, my2 = y * y, mxy = x * y, rho2 = mx2 + my2;
    double rad = c->k1 * rho2 + c->k2 * rho2 * rho2;
    *dx = x * rad + 2.0 * c->p1 * mxy + c->p2 * (rho2 + 2.0 * mx2);
    *dy = y * rad + 2.0 * c->p2 * mxy + c->p1 * (rho2 + 2.0 * my2);
}
VIO_HD void syn_lift(const vio_synth_config *c, double u, double v, double *x, double *y) {
    double mxd = (u - c->cx) / c->fx, myd = (v - c->cy) / c->fy, dx, dy;
    syn_distort(c, mxd, myd, &dx, &dy);
    double mx = mxd - dx, my = myd - dy;
    for (int i = 1; i < 8; i++) {
        syn_distort(c, mx, my, &dx, &dy);
        mx = mxd - dx;
        my = myd - dy;
    }
    *x = mx;
    *y = my;
}

struct CamPose { float R[9]; float p[3]; };  // world <- camera

// one pixel: ray (x,y,1) in the camera frame -> nearest wall -> texture + z-depth
VIO_HD void render_pixel(uint32_t seed, const CamPose &cp, float rx, float ry, uint8_t *gray, uint16_t *depth) {
    float d[3] = {cp.R[0] * rx + cp.R[1] * ry + cp.R[2], cp.R[3] * rx + cp.R[4] * ry + cp.R[5], cp.R[6] * rx + cp.R[7] * ry + cp.R[8]};
    const float lo[3] = {-4.f, -3.f, 0.f}, hi[3] = {4.f, 3.f, 3.f};
    float tbest = 1e30f;
    int wall = 0;
    for (int a = 0; a < 3; a++) {
        if (d[a] > 1e-9f) {
            float t = (hi[a] - cp.p[a]) / d[a];
            if (t < tbest) { tbest = t; wall = 2 * a; }
        } else if (d[a] < -1e-9f) {
            float t = (lo[a] - cp.p[a]) / d[a];
            if (t < tbest) { tbest = t; wall = 2 * a + 1; }
        }
    }
    float hx = cp.p[0] + tbest * d[0], hy = cp.p[1] + tbest * d[1], hz = cp.p[2] + tbest * d[2];
    float u, v;
    if (wall < 2) { u = hy; v = hz; }
    else if (wall < 4) { u = hx; v = hz; }
    else { u = hx; v = hy; }
    float val = scene_texture(seed, wall, u, v);
    val = val < 0.f ? 0.f : (val > 255.f ? 255.f : val);
    *gray = (uint8_t)(val + 0.5f);
    float mm = tbest * 1000.f;  // ray has z = 1 in the camera frame, so t is the z-depth
    *depth = (mm > 10000.f || mm < 0.f) ? (uint16_t)0 : (uint16_t)(mm + 0.5f);
}

}  // namespace vsyn
