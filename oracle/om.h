// ORACLE (test infrastructure only — never linked into or called by the product path).
// Small dense-algebra helpers for the CPU restatement of the VINS-RGBD-FAST hot path.
// Parity status: "parity unpinned" — the reference ships no tests/golden vectors and its
// third-party arithmetic (OpenCV / Ceres / Eigen) cannot be built in this image.
//
// Mirrors the handful of Eigen operations the reference uses:
//   Utility::deltaQ/Qleft/Qright/R2ypr/ypr2R/g2R   vins_estimator/src/utility/utility.h:11-108, utility.cpp:5-15
//   Eigen::Quaterniond(Matrix3d), toRotationMatrix, FromTwoVectors (Eigen/Geometry, un-vendored)
//   LLT / LDLT / SelfAdjointEigenSolver / JacobiSVD call sites listed in SURVEY.md §8c
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cstdio>

namespace om {

struct V3 {
    double x = 0, y = 0, z = 0;
    V3() {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double &operator[](int i) { return (&x)[i]; }
    double operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(const V3 &a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(double s, const V3 &a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(const V3 &a, double s) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(const V3 &a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline double norm(const V3 &a) { return std::sqrt(dot(a, a)); }

struct M3 {
    double m[3][3];
    M3() { std::memset(m, 0, sizeof(m)); }
    static M3 I() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = 1; return r; }
    double &operator()(int i, int j) { return m[i][j]; }
    double operator()(int i, int j) const { return m[i][j]; }
    V3 col(int j) const { return V3(m[0][j], m[1][j], m[2][j]); }
};
inline M3 operator*(const M3 &a, const M3 &b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
inline V3 operator*(const M3 &a, const V3 &v) {
    return V3(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
              a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
              a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
}
inline M3 operator*(double s, const M3 &a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j]; return r; }
inline M3 operator+(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j]; return r; }
inline M3 operator-(const M3 &a, const M3 &b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] - b.m[i][j]; return r; }
inline M3 operator-(const M3 &a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = -a.m[i][j]; return r; }
inline M3 T(const M3 &a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i]; return r; }
inline M3 skew(const V3 &q) {  // utility.h:26-34
    M3 r;
    r.m[0][1] = -q.z; r.m[0][2] = q.y;
    r.m[1][0] = q.z;  r.m[1][2] = -q.x;
    r.m[2][0] = -q.y; r.m[2][1] = q.x;
    return r;
}

// Hamilton quaternion, Eigen conventions (w,x,y,z constructor order, q*v rotates).
struct Q {
    double w = 1, x = 0, y = 0, z = 0;
    Q() {}
    Q(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
    V3 vec() const { return V3(x, y, z); }
};
inline Q operator*(const Q &a, const Q &b) {
    return Q(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
             a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
             a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}
inline Q normalized(const Q &q) {
    double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    return Q(q.w / n, q.x / n, q.y / n, q.z / n);
}
// Eigen's inverse(): conjugate / squaredNorm
inline Q inverse(const Q &q) {
    double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return Q(q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2);
}
// Eigen's toRotationMatrix (no normalisation inside, as Eigen)
inline M3 toR(const Q &q) {
    M3 r;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    r.m[0][0] = 1 - (tyy + tzz); r.m[0][1] = txy - twz;       r.m[0][2] = txz + twy;
    r.m[1][0] = txy + twz;       r.m[1][1] = 1 - (txx + tzz); r.m[1][2] = tyz - twx;
    r.m[2][0] = txz - twy;       r.m[2][1] = tyz + twx;       r.m[2][2] = 1 - (txx + tyy);
    return r;
}
// Eigen's q * v  (Quaternion::_transformVector)
inline V3 rot(const Q &q, const V3 &v) {
    V3 u = q.vec();
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + q.w * uv + cross(u, uv);
}
// Eigen's Quaternion(Matrix3)
inline Q fromR(const M3 &m) {
    Q q;
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m(2, 1) - m(1, 2)) * t;
        q.y = (m(0, 2) - m(2, 0)) * t;
        q.z = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        double qv[3];
        qv[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m(k, j) - m(j, k)) * t;
        qv[j] = (m(j, i) + m(i, j)) * t;
        qv[k] = (m(k, i) + m(i, k)) * t;
        q.x = qv[0]; q.y = qv[1]; q.z = qv[2];
    }
    return q;
}
#define SINCOS_DET_QUAL inline
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
#define LS1D_QUAL inline
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
inline Q deltaQ(const V3 &theta) { return Q(1.0, theta.x / 2, theta.y / 2, theta.z / 2); }  // utility.h:11-24 (unnormalised)

inline V3 R2ypr(const M3 &R) {  // utility.h:66-81 (degrees)
    V3 n = R.col(0), o = R.col(1), a = R.col(2);
    double y = std::atan2(n.y, n.x);
    double p = std::atan2(-n.z, n.x * std::cos(y) + n.y * std::sin(y));
    double r = std::atan2(a.x * std::sin(y) - a.y * std::cos(y), -o.x * std::sin(y) + o.y * std::cos(y));
    return V3(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
inline M3 ypr2R(const V3 &ypr) {  // utility.h:83-108
    double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
    M3 Rz, Ry, Rx;
    Rz(0, 0) = std::cos(y); Rz(0, 1) = -std::sin(y); Rz(1, 0) = std::sin(y); Rz(1, 1) = std::cos(y); Rz(2, 2) = 1;
    Ry(0, 0) = std::cos(p); Ry(0, 2) = std::sin(p); Ry(1, 1) = 1; Ry(2, 0) = -std::sin(p); Ry(2, 2) = std::cos(p);
    Rx(0, 0) = 1; Rx(1, 1) = std::cos(r); Rx(1, 2) = -std::sin(r); Rx(2, 1) = std::sin(r); Rx(2, 2) = std::cos(r);
    return Rz * Ry * Rx;
}
// Eigen's Quaternion::FromTwoVectors for unit inputs (non-antiparallel branch; the
// antiparallel branch needs an SVD in Eigen and is unreachable for accelerometer gravity).
inline Q fromTwoVectors(const V3 &a, const V3 &b) {
    V3 v0 = a / norm(a), v1 = b / norm(b);
    double c = dot(v1, v0);
    if (c < -1.0 + 1e-12) {  // pick any orthogonal axis
        V3 ax = std::fabs(v0.x) < 0.9 ? cross(v0, V3(1, 0, 0)) : cross(v0, V3(0, 1, 0));
        ax = ax / norm(ax);
        return Q(0, ax.x, ax.y, ax.z);
    }
    V3 axis = cross(v0, v1);
    double s = std::sqrt((1.0 + c) * 2.0);
    double invs = 1.0 / s;
    return Q(s * 0.5, axis.x * invs, axis.y * invs, axis.z * invs);
}
inline M3 g2R(const V3 &g) {  // utility.cpp:5-15
    M3 R0 = toR(fromTwoVectors(g, V3(0, 0, 1)));
    double yaw = R2ypr(R0).x;
    return ypr2R(V3(-yaw, 0, 0)) * R0;
}

// ---------------------------------------------------------------- dynamic row-major matrix
struct Mat {
    int r = 0, c = 0;
    std::vector<double> d;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return d[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
    void zero() { std::fill(d.begin(), d.end(), 0.0); }
};

// Cyclic Jacobi eigen-decomposition of a symmetric n×n matrix (stand-in for
// Eigen::SelfAdjointEigenSolver; agreement is to round-off, order = ascending).
// A is destroyed; V columns are eigenvectors.
inline void sym_eig(Mat &A, std::vector<double> &w, Mat &V) {
    int n = A.r;
    V = Mat(n, n);
    for (int i = 0; i < n; i++) V(i, i) = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        // threshold Jacobi: rotate only if |a_pq| > 1e-15 sqrt(|a_pp a_qq|) and > 1e-18 max|a_ii|; stop after a sweep without rotations
        double dmax = 0;
        for (int i = 0; i < n; i++) dmax = std::max(dmax, std::fabs(A(i, i)));
        const double absfloor = 1e-18 * dmax;
        int nrot = 0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A(p, q);
                double app = A(p, p), aqq = A(q, q);
                if (!(std::fabs(apq) > absfloor && std::fabs(apq) > 1e-15 * std::sqrt(std::fabs(app * aqq)))) continue;
                nrot++;
                double tau = (aqq - app) / (2.0 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                double cs = 1.0 / std::sqrt(1.0 + t * t), sn = t * cs;
                for (int k = 0; k < n; k++) {
                    double akp = A(k, p), akq = A(k, q);
                    A(k, p) = cs * akp - sn * akq;
                    A(k, q) = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A(p, k), aqk = A(q, k);
                    A(p, k) = cs * apk - sn * aqk;
                    A(q, k) = sn * apk + cs * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V(k, p), vkq = V(k, q);
                    V(k, p) = cs * vkp - sn * vkq;
                    V(k, q) = sn * vkp + cs * vkq;
                }
            }
        if (nrot == 0) break;
    }
    w.resize(n);
    std::vector<int> idx(n);
    for (int i = 0; i < n; i++) { w[i] = A(i, i); idx[i] = i; }
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return w[a] < w[b]; });
    Mat V2(n, n);
    std::vector<double> w2(n);
    for (int j = 0; j < n; j++) {
        w2[j] = w[idx[j]];
        for (int i = 0; i < n; i++) V2(i, j) = V(i, idx[j]);
    }
    w = w2;
    V = V2;
}

// In-place Cholesky (lower) of symmetric positive definite A (n×n). Returns false on failure.
inline bool chol(Mat &A) {
    int n = A.r;
    for (int j = 0; j < n; j++) {
        double s = A(j, j);
        for (int k = 0; k < j; k++) s -= A(j, k) * A(j, k);
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        double l = std::sqrt(s);
        A(j, j) = l;
        for (int i = j + 1; i < n; i++) {
            double t = A(i, j);
            for (int k = 0; k < j; k++) t -= A(i, k) * A(j, k);
            A(i, j) = t / l;
        }
    }
    return true;
}
inline void chol_solve(const Mat &L, std::vector<double> &b) {
    int n = L.r;
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L(i, k) * b[k];
        b[i] = s / L(i, i);
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L(k, i) * b[k];
        b[i] = s / L(i, i);
    }
}

}  // namespace om
