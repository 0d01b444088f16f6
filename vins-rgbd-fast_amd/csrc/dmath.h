// Small fixed-size FP64 algebra for the device kernels (gfx950). Everything is __host__ __device__ so the
// per-thread math can also be exercised from host-side self tests; the kernels are the only product callers.
// Conventions follow the reference's Eigen usage: Hamilton quaternions stored (w,x,y,z), row-major 3x3.
// Reference helpers restated here: vins_estimator/src/utility/utility.h:11-108, utility.cpp:5-15.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DM_HD __host__ __device__ __forceinline__
#else
#define DM_HD inline
#endif

namespace dm {

struct v3 { double x, y, z; };
struct m3 { double a[9]; };
struct quat { double w, x, y, z; };

DM_HD v3 mk(double x, double y, double z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
DM_HD v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DM_HD v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DM_HD v3 scl(double s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
DM_HD v3 neg(v3 a) { return mk(-a.x, -a.y, -a.z); }
DM_HD double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DM_HD v3 cross(v3 a, v3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DM_HD double nrm(v3 a) { return sqrt(dot(a, a)); }
DM_HD double get(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
DM_HD v3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
DM_HD void st3(double *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

DM_HD m3 eye() { m3 r; for (int i = 0; i < 9; i++) r.a[i] = 0; r.a[0] = r.a[4] = r.a[8] = 1; return r; }
DM_HD m3 zero3() { m3 r; for (int i = 0; i < 9; i++) r.a[i] = 0; return r; }
DM_HD m3 ldm(const double *p) { m3 r; for (int i = 0; i < 9; i++) r.a[i] = p[i]; return r; }
DM_HD void stm(double *p, const m3 &m) { for (int i = 0; i < 9; i++) p[i] = m.a[i]; }
DM_HD m3 mul(const m3 &A, const m3 &B) {
    m3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A.a[i * 3 + k] * B.a[k * 3 + j];
            r.a[i * 3 + j] = s;
        }
    return r;
}
DM_HD v3 mul(const m3 &A, v3 v) {
    return mk(A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z, A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z,
              A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z);
}
DM_HD m3 tr(const m3 &A) { m3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[i * 3 + j] = A.a[j * 3 + i]; return r; }
DM_HD m3 scl(double s, const m3 &A) { m3 r; for (int i = 0; i < 9; i++) r.a[i] = s * A.a[i]; return r; }
DM_HD m3 add(const m3 &A, const m3 &B) { m3 r; for (int i = 0; i < 9; i++) r.a[i] = A.a[i] + B.a[i]; return r; }
DM_HD m3 sub(const m3 &A, const m3 &B) { m3 r; for (int i = 0; i < 9; i++) r.a[i] = A.a[i] - B.a[i]; return r; }
DM_HD m3 neg(const m3 &A) { m3 r; for (int i = 0; i < 9; i++) r.a[i] = -A.a[i]; return r; }
DM_HD m3 skew(v3 q) {  // utility.h:26-34
    m3 r = zero3();
    r.a[1] = -q.z; r.a[2] = q.y; r.a[3] = q.z; r.a[5] = -q.x; r.a[6] = -q.y; r.a[7] = q.x;
    return r;
}

DM_HD quat mkq(double w, double x, double y, double z) { quat q; q.w = w; q.x = x; q.y = y; q.z = z; return q; }
DM_HD v3 qvec(quat q) { return mk(q.x, q.y, q.z); }
DM_HD quat qmul(quat a, quat b) {
    return mkq(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
               a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
DM_HD quat qnormalized(quat q) {
    double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    return mkq(q.w / n, q.x / n, q.y / n, q.z / n);
}
DM_HD quat qinv(quat q) {  // Eigen inverse(): conjugate / squaredNorm
    double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return mkq(q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2);
}
DM_HD m3 q2R(quat q) {  // Eigen toRotationMatrix (no normalisation)
    m3 r;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    r.a[0] = 1 - (tyy + tzz); r.a[1] = txy - twz;       r.a[2] = txz + twy;
    r.a[3] = txy + twz;       r.a[4] = 1 - (txx + tzz); r.a[5] = tyz - twx;
    r.a[6] = txz - twy;       r.a[7] = tyz + twx;       r.a[8] = 1 - (txx + tyy);
    return r;
}
DM_HD v3 qrot(quat q, v3 v) {  // Eigen q * v
    v3 u = qvec(q);
    v3 uv = cross(u, v);
    uv = add(uv, uv);
    return add(add(v, scl(q.w, uv)), cross(u, uv));
}
DM_HD quat R2q(const m3 &m) {  // Eigen Quaternion(Matrix3)
    quat q;
    double t = m.a[0] + m.a[4] + m.a[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m.a[7] - m.a[5]) * t;
        q.y = (m.a[2] - m.a[6]) * t;
        q.z = (m.a[3] - m.a[1]) * t;
    } else {
        int i = 0;
        if (m.a[4] > m.a[0]) i = 1;
        if (m.a[8] > m.a[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m.a[i * 4] - m.a[j * 4] - m.a[k * 4] + 1.0);
        double qv[3];
        qv[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m.a[k * 3 + j] - m.a[j * 3 + k]) * t;
        qv[j] = (m.a[j * 3 + i] + m.a[i * 3 + j]) * t;
        qv[k] = (m.a[k * 3 + i] + m.a[i * 3 + k]) * t;
        q.x = qv[0]; q.y = qv[1]; q.z = qv[2];
    }
    return q;
}
#define SINCOS_DET_QUAL DM_HD
// Deterministic sin / cos for Estimator::predictMotion's AngleAxisd (estimator.cpp:1853-1856): the tracker's prediction is the one place where
// a transcendental function feeds a bit-exact comparison, and two math libraries round sin / cos differently in the last place.  Only + - *
// and a round-to-nearest by the 1.5 * 2^52 constant: every IEEE-754 double implementation without contraction returns the same bits.
// Cody-Waite reduction by pi/2 in three parts (exact products for |x| < 2^20 * pi/2), then the classical minimax polynomials for
// |r| <= pi/4 (coefficients of the Sun fdlibm kernels); < 2 ulp.  THE SAME TEXT lives in oracle/om.h and csrc/dmath.h.
SINCOS_DET_QUAL void sincos_det(double x, double *sn, double *cs) {
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
    *sn = q == 0 ? s0 : (q == 1 ? c0 : (q == 2 ? -s0 : -c0));
    *cs = q == 0 ? c0 : (q == 1 ? -s0 : (q == 2 ? -c0 : s0));
}
#undef SINCOS_DET_QUAL
#define LS1D_QUAL DM_HD
// ---- LS1D: the one-dimensional part of Ceres' projected Armijo line search (bounds-constrained programs) ----------------------------------
// TrustRegionMinimizer::DoLineSearch runs ArmijoLineSearch (line_search.cc) with CUBIC interpolation along the trust-region step whenever the
// program has a bounded parameter (estimator.cpp:1282-1297 puts an upper bound on the inverse depth of depth-less landmarks).  This block is
// the scalar machinery of that search -- LineSearch::InterpolatingPolynomialMinimizingStepSize, polynomial.cc FindInterpolatingPolynomial /
// MinimizePolynomial / FindPolynomialRoots -- restated from Ceres 2.x's published algorithm (Ceres is not in the image).  Samples
// carry value AND gradient (CUBIC): two samples give a cubic, three a quintic.  Ceres finds the critical points as the eigenvalues of the
// balanced companion matrix; here degree 1 and 2 use Ceres' closed forms and degree 3 / 4 a Durand-Kerner iteration on the monic polynomial
// (same roots to ~1e-15; like Ceres, the REAL PARTS of all roots are candidates).  Only + - * / sqrt: the same bits on the host and on gfx950.
// Everything that is indexed at run time lives in the caller's workspace ws (>= 96 doubles; an LDS region on the GPU, so that the kernel
// that carries the search keeps no private-memory arrays).  THE SAME TEXT lives in oracle/om.h and csrc/dmath.h.
struct LsSample { double x, value, gradient; int valid; };
LS1D_QUAL double ls_poly_eval(const double *c, int n, double x) {   // n coefficients, highest degree first (Horner, EvaluatePolynomial)
    double v = 0.0;
    for (int i = 0; i < n; i++) v = v * x + c[i];
    return v;
}
// real parts of all roots of the polynomial cin[0 .. nin-1] (highest degree first, nin <= 5); returns how many.  ws: >= 13 doubles
LS1D_QUAL int ls_poly_roots_real(const double *cin, int nin, double *re, double *ws) {
    int lead = 0;
    while (lead < nin - 1 && cin[lead] == 0.0) lead++;   // RemoveLeadingZeros
    const double *c = cin + lead;
    const int deg = nin - lead - 1;
    if (deg <= 0) return 0;
    if (deg == 1) { re[0] = -c[1] / c[0]; return 1; }
    if (deg == 2) {   // FindQuadraticPolynomialRoots (BKP Horn's stable form)
        const double a = c[0], b = c[1], cc = c[2];
        const double D = b * b - 4 * a * cc;
        const double sD = sqrt(fabs(D));
        if (D >= 0) {
            if (b >= 0) { re[0] = (-b - sD) / (2.0 * a); re[1] = (2.0 * cc) / (-b - sD); }
            else { re[0] = (2.0 * cc) / (-b + sD); re[1] = (-b + sD) / (2.0 * a); }
        } else { re[0] = -b / (2.0 * a); re[1] = -b / (2.0 * a); }
        return 2;
    }
    // degree 3 / 4: Durand-Kerner on the monic polynomial, start points on a circle of the Cauchy root bound
    double *m = ws, *zr = ws + 5, *zi = ws + 9;
    double bound = 0.0;
    for (int i = 0; i <= deg; i++) { m[i] = c[i] / c[0]; if (i > 0 && fabs(m[i]) > bound) bound = fabs(m[i]); }
    bound = 1.0 + bound;
    {
        double pr = 1.0, pi = 0.0;   // powers of 0.4 + 0.9 i (not a root of unity, not real)
        for (int k = 0; k < deg; k++) { zr[k] = bound * pr; zi[k] = bound * pi; const double nr = pr * 0.4 - pi * 0.9, ni = pr * 0.9 + pi * 0.4; pr = nr; pi = ni; }
    }
    for (int it = 0; it < 500; it++) {
        double change = 0.0, size = 0.0;
        for (int k = 0; k < deg; k++) {
            double vr = 1.0, vi = 0.0;   // p(z_k), Horner in complex arithmetic
            for (int i = 1; i <= deg; i++) { const double nr = vr * zr[k] - vi * zi[k] + m[i], ni = vr * zi[k] + vi * zr[k]; vr = nr; vi = ni; }
            double dr = 1.0, di = 0.0;   // prod_{j != k} (z_k - z_j)
            for (int j = 0; j < deg; j++) {
                if (j == k) continue;
                const double er = zr[k] - zr[j], ei = zi[k] - zi[j];
                const double nr = dr * er - di * ei, ni = dr * ei + di * er;
                dr = nr; di = ni;
            }
            const double dn = dr * dr + di * di;
            if (dn == 0.0) continue;
            const double qr = (vr * dr + vi * di) / dn, qi = (vi * dr - vr * di) / dn;
            zr[k] -= qr; zi[k] -= qi;
            change += fabs(qr) + fabs(qi);
            size += fabs(zr[k]) + fabs(zi[k]);
        }
        if (change <= 1e-16 * size) break;
    }
    for (int k = 0; k < deg; k++) re[k] = zr[k];
    return deg;
}
// FindInterpolatingPolynomial: value and gradient of ns samples (ns = 2, 3; smp = rows x, value, gradient) -> 2 ns coefficients, highest
// degree first; Gaussian elimination with full pivoting (Eigen::FullPivLU with threshold 0).  ws: >= 54 doubles
LS1D_QUAL void ls_fit_poly(const double *smp, int ns, double *coef, double *ws) {
    const int n = 2 * ns, degree = n - 1, ld = 7;
    double *A = ws, *colp = ws + 42, *y = ws + 48;
    for (int i = 0; i < ns; i++) {
        const double x = smp[3 * i];
        for (int j = 0; j <= degree; j++) {
            double pw = 1.0;
            for (int e = 0; e < degree - j; e++) pw *= x;
            A[(2 * i) * ld + j] = pw;
            double pd = 0.0;
            if (j < degree) { pd = (double)(degree - j); for (int e = 0; e < degree - j - 1; e++) pd *= x; }
            A[(2 * i + 1) * ld + j] = pd;
        }
        A[(2 * i) * ld + n] = smp[3 * i + 1];
        A[(2 * i + 1) * ld + n] = smp[3 * i + 2];
    }
    for (int j = 0; j < n; j++) colp[j] = (double)j;
    for (int k = 0; k < n; k++) {
        int pr = k, pc = k;
        double best = -1.0;
        for (int i = k; i < n; i++) for (int j = k; j < n; j++) if (fabs(A[i * ld + j]) > best) { best = fabs(A[i * ld + j]); pr = i; pc = j; }
        if (best <= 0.0) break;
        if (pr != k) for (int j = 0; j <= n; j++) { const double t = A[k * ld + j]; A[k * ld + j] = A[pr * ld + j]; A[pr * ld + j] = t; }
        if (pc != k) { for (int i = 0; i < n; i++) { const double t = A[i * ld + k]; A[i * ld + k] = A[i * ld + pc]; A[i * ld + pc] = t; } const double t = colp[k]; colp[k] = colp[pc]; colp[pc] = t; }
        for (int i = k + 1; i < n; i++) {
            const double f = A[i * ld + k] / A[k * ld + k];
            for (int j = k; j <= n; j++) A[i * ld + j] -= f * A[k * ld + j];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        double acc = A[k * ld + n];
        for (int j = k + 1; j < n; j++) acc -= A[k * ld + j] * y[j];
        y[k] = A[k * ld + k] != 0.0 ? acc / A[k * ld + k] : 0.0;
    }
    for (int k = 0; k < n; k++) coef[(int)colp[k]] = y[k];
}
// LineSearch::InterpolatingPolynomialMinimizingStepSize (CUBIC) + MinimizeInterpolatingPolynomial: the next trial step in [lo, hi]
LS1D_QUAL double ls_next_step(const LsSample &lower, const LsSample &previous, const LsSample &current, double lo, double hi, double *ws) {
    if (!current.valid) { const double h = current.x * 0.5; return h < lo ? (lo < hi ? lo : hi) : (h < hi ? h : hi); }   // min(max(x / 2, lo), hi)
    double *smp = ws, *coef = ws + 9, *der = ws + 15, *roots = ws + 20, *sub = ws + 24;
    int ns = 2;
    smp[0] = lower.x; smp[1] = lower.value; smp[2] = lower.gradient;
    smp[3] = current.x; smp[4] = current.value; smp[5] = current.gradient;
    if (previous.valid) { smp[6] = previous.x; smp[7] = previous.value; smp[8] = previous.gradient; ns = 3; }
    ls_fit_poly(smp, ns, coef, sub);
    const int n = 2 * ns;
    // MinimizePolynomial: the middle of the interval first, then the ends, then the critical points inside
    double best_x = (lo + hi) / 2.0, best_v = ls_poly_eval(coef, n, best_x);
    const double vlo = ls_poly_eval(coef, n, lo);
    if (vlo < best_v) { best_v = vlo; best_x = lo; }
    const double vhi = ls_poly_eval(coef, n, hi);
    if (vhi < best_v) { best_v = vhi; best_x = hi; }
    for (int i = 0; i < n - 1; i++) der[i] = (double)(n - 1 - i) * coef[i];   // DifferentiatePolynomial
    const int nr = ls_poly_roots_real(der, n - 1, roots, sub);
    for (int i = 0; i < nr; i++) {
        const double r = roots[i];
        if (r < lo || r > hi) continue;
        const double v = ls_poly_eval(coef, n, r);
        if (v < best_v) { best_v = v; best_x = r; }
    }
    for (int i = 0; i < ns; i++) {   // MinimizeInterpolatingPolynomial: the samples themselves
        const double x = smp[3 * i];
        if (x < lo || x > hi) continue;
        const double v = ls_poly_eval(coef, n, x);
        if (v < best_v) { best_v = v; best_x = x; }
    }
    return best_x;
}
// ---- end LS1D ----------------------------------------------------------------------------------------------------------------------------
#undef LS1D_QUAL
DM_HD quat deltaQ(v3 th) { return mkq(1.0, th.x / 2, th.y / 2, th.z / 2); }  // utility.h:11-24

DM_HD v3 R2ypr(const m3 &R) {  // utility.h:66-81, degrees
    const double PI = 3.14159265358979323846;
    v3 n = mk(R.a[0], R.a[3], R.a[6]), o = mk(R.a[1], R.a[4], R.a[7]), a = mk(R.a[2], R.a[5], R.a[8]);
    double y = atan2(n.y, n.x);
    double p = atan2(-n.z, n.x * cos(y) + n.y * sin(y));
    double r = atan2(a.x * sin(y) - a.y * cos(y), -o.x * sin(y) + o.y * cos(y));
    return mk(y / PI * 180.0, p / PI * 180.0, r / PI * 180.0);
}
DM_HD m3 ypr2R(v3 ypr) {  // utility.h:83-108
    const double PI = 3.14159265358979323846;
    double y = ypr.x / 180.0 * PI, p = ypr.y / 180.0 * PI, r = ypr.z / 180.0 * PI;
    m3 Rz = zero3(), Ry = zero3(), Rx = zero3();
    Rz.a[0] = cos(y); Rz.a[1] = -sin(y); Rz.a[3] = sin(y); Rz.a[4] = cos(y); Rz.a[8] = 1;
    Ry.a[0] = cos(p); Ry.a[2] = sin(p); Ry.a[4] = 1; Ry.a[6] = -sin(p); Ry.a[8] = cos(p);
    Rx.a[0] = 1; Rx.a[4] = cos(r); Rx.a[5] = -sin(r); Rx.a[7] = sin(r); Rx.a[8] = cos(r);
    return mul(mul(Rz, Ry), Rx);
}
DM_HD quat fromTwoVectors(v3 a, v3 b) {  // Eigen FromTwoVectors, non-antiparallel branch
    v3 v0 = scl(1.0 / nrm(a), a), v1 = scl(1.0 / nrm(b), b);
    double c = dot(v1, v0);
    if (c < -1.0 + 1e-12) {
        v3 ax = fabs(v0.x) < 0.9 ? cross(v0, mk(1, 0, 0)) : cross(v0, mk(0, 1, 0));
        ax = scl(1.0 / nrm(ax), ax);
        return mkq(0, ax.x, ax.y, ax.z);
    }
    v3 axis = cross(v0, v1);
    double s = sqrt((1.0 + c) * 2.0);
    double invs = 1.0 / s;
    return mkq(s * 0.5, axis.x * invs, axis.y * invs, axis.z * invs);
}
DM_HD m3 g2R(v3 g) {  // utility.cpp:5-15
    m3 R0 = q2R(fromTwoVectors(g, mk(0, 0, 1)));
    double yaw = R2ypr(R0).x;
    return mul(ypr2R(mk(-yaw, 0, 0)), R0);
}

}  // namespace dm
