// Dynamic initialisation on the host (see dyninit_host.h).  Third-party routines the reference calls here are un-vendored (OpenCV
// 3.x solvePnP / solvePnPRansac, Ceres 1.x/2.x Solve, Eigen LDLT); they are restated from their published algorithms:
//   cv::solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess)   CvLevMarq on (rvec, tvec): lambda = 10^k, diagonal x (1 + lambda), 20
//                                                          iterations / FLT_EPSILON (calib3d/calibration.cpp, compat_ptsetreg.cpp)
//   cv::solvePnPRansac(SOLVEPNP_EPNP, 100, 1/460, 0.99)    RANSACPointSetRegistrator with cv::RNG((uint64)-1), 5-point EPnP models
//                                                          (Lepetit, Moreno-Noguer, Fua 2009), final EPnP over the inliers
//   ceres::Solve(DENSE_SCHUR) in GlobalSFM::construct      Levenberg-Marquardt trust region, Jacobi column scaling, quaternion
//                                                          local parameterisation, point block eliminated first
// Points pass through float where the reference stores them in cv::Point2f / cv::Point3f.
#include "dyninit_host.h"

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>

#include "be_factors.h"

namespace vinit {
using namespace dm;

namespace {

// ------------------------------------------------------------------------------------------------ small dense algebra
struct Dense {
    int r = 0, c = 0;
    std::vector<double> a;
    Dense() {}
    Dense(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};

// cyclic Jacobi for a symmetric matrix: A = V diag(w) V^T, eigenvalues ascending, eigenvectors in the columns of V
void sym_eigen(Dense A, std::vector<double> &w, Dense &V) {
    const int n = A.r;
    V = Dense(n, n);
    for (int i = 0; i < n; i++) V(i, i) = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, dia = 0;
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) (i == j ? dia : off) += A(i, j) * A(i, j);
        if (!(off > 1e-30 * dia) || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A(p, q);
                if (apq == 0.0) continue;
                const double th = (A(q, q) - A(p, p)) / (2 * apq);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
                const double cs = 1 / sqrt(t * t + 1), sn = t * cs;
                for (int k = 0; k < n; k++) { const double x = A(k, p), y = A(k, q); A(k, p) = cs * x - sn * y; A(k, q) = sn * x + cs * y; }
                for (int k = 0; k < n; k++) { const double x = A(p, k), y = A(q, k); A(p, k) = cs * x - sn * y; A(q, k) = sn * x + cs * y; }
                for (int k = 0; k < n; k++) { const double x = V(k, p), y = V(k, q); V(k, p) = cs * x - sn * y; V(k, q) = sn * x + cs * y; }
            }
    }
    std::vector<int> ord(n);
    for (int i = 0; i < n; i++) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return A(x, x) < A(y, y); });
    w.resize(n);
    Dense Vs(n, n);
    for (int j = 0; j < n; j++) { w[j] = A(ord[j], ord[j]); for (int i = 0; i < n; i++) Vs(i, j) = V(i, ord[j]); }
    V = Vs;
}
// x = pinv(A) b, A symmetric (cv::solve(DECOMP_SVD) on a normal matrix)
std::vector<double> solve_sym_pinv(const Dense &A, const std::vector<double> &b) {
    const int n = A.r;
    std::vector<double> w, x(n, 0.0);
    Dense V;
    sym_eigen(A, w, V);
    double wmax = 0;
    for (double v : w) wmax = std::max(wmax, fabs(v));
    for (int k = 0; k < n; k++) {
        if (!(fabs(w[k]) > wmax * 2 * 2.220446049250313e-16 * n)) continue;
        double s = 0;
        for (int i = 0; i < n; i++) s += V(i, k) * b[i];
        s /= w[k];
        for (int i = 0; i < n; i++) x[i] += V(i, k) * s;
    }
    return x;
}
// min |A x - b|, minimum-norm solution (cvSolve(CV_SVD) / qr_solve in the published EPnP code): thin SVD of A by one-sided Jacobi
// (Hestenes) -- the columns of A V are orthogonalised by plane rotations, their norms are the singular values -- then
// x = V S^+ (A V)^T b / S with singular values below n eps s_max dropped.  A has at most a handful of columns here.
std::vector<double> least_squares(const Dense &A, const std::vector<double> &b) {
    const int m = A.r, n = A.c;
    Dense U = A, V(n, n);
    for (int j = 0; j < n; j++) V(j, j) = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p + 1 < n; p++)
            for (int q = p + 1; q < n; q++) {
                double app = 0, aqq = 0, apq = 0;
                for (int k = 0; k < m; k++) { app += U(k, p) * U(k, p); aqq += U(k, q) * U(k, q); apq += U(k, p) * U(k, q); }
                if (!(fabs(apq) > 1e-15 * sqrt(app * aqq)) || apq == 0.0) continue;
                rotated = true;
                const double zeta = (aqq - app) / (2 * apq);
                const double tn = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                const double cs = 1 / sqrt(1 + tn * tn), sn = cs * tn;
                for (int k = 0; k < m; k++) { const double x = U(k, p), y = U(k, q); U(k, p) = cs * x - sn * y; U(k, q) = sn * x + cs * y; }
                for (int k = 0; k < n; k++) { const double x = V(k, p), y = V(k, q); V(k, p) = cs * x - sn * y; V(k, q) = sn * x + cs * y; }
            }
        if (!rotated) break;
    }
    std::vector<double> sv(n), x(n, 0.0);
    double smax = 0;
    for (int j = 0; j < n; j++) { double s2 = 0; for (int k = 0; k < m; k++) s2 += U(k, j) * U(k, j); sv[j] = sqrt(s2); smax = std::max(smax, sv[j]); }
    for (int j = 0; j < n; j++) {
        if (!(sv[j] > smax * 2.220446049250313e-16 * std::max(m, n))) continue;
        double ub = 0;
        for (int k = 0; k < m; k++) ub += U(k, j) * b[k];
        const double coef = ub / (sv[j] * sv[j]);
        for (int i = 0; i < n; i++) x[i] += V(i, j) * coef;
    }
    return x;
}
// in-place Cholesky (lower) of a symmetric positive definite matrix; false if a pivot is not positive
bool cholesky(Dense &A) {
    const int n = A.r;
    for (int j = 0; j < n; j++) {
        double d = A(j, j);
        for (int k = 0; k < j; k++) d -= A(j, k) * A(j, k);
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        const double l = sqrt(d);
        A(j, j) = l;
        for (int i = j + 1; i < n; i++) {
            double s = A(i, j);
            for (int k = 0; k < j; k++) s -= A(i, k) * A(j, k);
            A(i, j) = s / l;
        }
    }
    return true;
}
void cholesky_solve(const Dense &L, std::vector<double> &b) {
    const int n = L.r;
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L(i, k) * b[k]; b[i] = s / L(i, i); }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L(k, i) * b[k]; b[i] = s / L(i, i); }
}
// Eigen::LDLT::solve for a symmetric positive semi-definite matrix: pivot = largest remaining diagonal entry, an exactly zero
// pivot ends the factorisation (the remaining block is treated as zero)
std::vector<double> ldlt_solve(Dense A, const std::vector<double> &b) {
    const int n = A.r;
    std::vector<int> perm(n);
    for (int i = 0; i < n; i++) perm[i] = i;
    std::vector<double> d(n, 0.0);
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++) if (fabs(A(i, i)) > fabs(A(p, p))) p = i;
        if (p != k) {
            for (int j = 0; j < n; j++) std::swap(A(k, j), A(p, j));
            for (int i = 0; i < n; i++) std::swap(A(i, k), A(i, p));
            std::swap(perm[k], perm[p]);
        }
        d[k] = A(k, k);
        if (d[k] == 0.0) {
            for (int i = k; i < n; i++) d[i] = 0;
            for (int i = k; i < n; i++) for (int j = k + 1; j < n; j++) if (j > i) A(j, i) = 0;
            break;
        }
        for (int i = k + 1; i < n; i++) A(i, k) /= d[k];
        for (int i = k + 1; i < n; i++)
            for (int j = k + 1; j <= i; j++) { A(i, j) -= A(i, k) * d[k] * A(j, k); A(j, i) = A(i, j); }
    }
    std::vector<double> y(n), x(n);
    for (int i = 0; i < n; i++) y[i] = b[perm[i]];
    for (int i = 0; i < n; i++) for (int k = 0; k < i; k++) y[i] -= A(i, k) * y[k];
    for (int i = 0; i < n; i++) y[i] = d[i] != 0 ? y[i] / d[i] : 0.0;
    for (int i = n - 1; i >= 0; i--) for (int k = i + 1; k < n; k++) y[i] -= A(k, i) * y[k];
    for (int i = 0; i < n; i++) x[perm[i]] = y[i];
    return x;
}

inline double as_float(double v) { return (double)(float)v; }
inline v3 col(const m3 &A, int j) { return mk(A.a[j], A.a[3 + j], A.a[6 + j]); }
inline double det(const m3 &R) {
    return R.a[0] * (R.a[4] * R.a[8] - R.a[5] * R.a[7]) - R.a[1] * (R.a[3] * R.a[8] - R.a[5] * R.a[6]) + R.a[2] * (R.a[3] * R.a[7] - R.a[4] * R.a[6]);
}
m3 inverse3(const m3 &A) {
    const double d = det(A);
    m3 r;
    r.a[0] = (A.a[4] * A.a[8] - A.a[5] * A.a[7]) / d; r.a[1] = (A.a[2] * A.a[7] - A.a[1] * A.a[8]) / d; r.a[2] = (A.a[1] * A.a[5] - A.a[2] * A.a[4]) / d;
    r.a[3] = (A.a[5] * A.a[6] - A.a[3] * A.a[8]) / d; r.a[4] = (A.a[0] * A.a[8] - A.a[2] * A.a[6]) / d; r.a[5] = (A.a[2] * A.a[3] - A.a[0] * A.a[5]) / d;
    r.a[6] = (A.a[3] * A.a[7] - A.a[4] * A.a[6]) / d; r.a[7] = (A.a[1] * A.a[6] - A.a[0] * A.a[7]) / d; r.a[8] = (A.a[0] * A.a[4] - A.a[1] * A.a[3]) / d;
    return r;
}
// A = U diag(s) V^T (3 x 3), singular values descending, through the eigen-decomposition of A^T A
void svd_3x3(const m3 &A, m3 &U, double s[3], m3 &V) {
    Dense N(3, 3), Ve;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double t = 0; for (int k = 0; k < 3; k++) t += A.a[k * 3 + i] * A.a[k * 3 + j]; N(i, j) = t; }
    std::vector<double> w;
    sym_eigen(N, w, Ve);
    for (int j = 0; j < 3; j++) {
        s[j] = sqrt(std::max(w[2 - j], 0.0));
        for (int i = 0; i < 3; i++) V.a[i * 3 + j] = Ve(i, 2 - j);
    }
    v3 u[3];
    for (int j = 0; j < 3; j++) {
        v3 cj = mul(A, col(V, j));
        u[j] = s[j] > 1e-12 * std::max(s[0], 1e-300) ? scl(1.0 / s[j], cj) : mk(0, 0, 0);
    }
    if (nrm(u[1]) < 0.5) { v3 h = fabs(u[0].x) < 0.9 ? mk(1, 0, 0) : mk(0, 1, 0); u[1] = cross(u[0], h); u[1] = scl(1.0 / nrm(u[1]), u[1]); }
    if (nrm(u[2]) < 0.5) u[2] = cross(u[0], u[1]);
    for (int j = 0; j < 3; j++) { U.a[j] = u[j].x; U.a[3 + j] = u[j].y; U.a[6 + j] = u[j].z; }
}

// ------------------------------------------------------------------------------------------------ cv::Rodrigues
m3 rodrigues(v3 r) {
    const double th = nrm(r);
    if (th < 2.220446049250313e-16) return eye();
    const v3 k = scl(1.0 / th, r);
    const double c = cos(th), s = sin(th), c1 = 1 - c;
    const m3 K = skew(k);
    const double kk[3] = {k.x, k.y, k.z};
    m3 R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.a[i * 3 + j] = (i == j ? c : 0.0) + c1 * kk[i] * kk[j] + s * K.a[i * 3 + j];
    return R;
}
v3 rodrigues_inv(const m3 &R) {
    v3 r = mk(R.a[7] - R.a[5], R.a[2] - R.a[6], R.a[3] - R.a[1]);
    const double s = sqrt(dot(r, r) * 0.25);
    double c = (R.a[0] + R.a[4] + R.a[8] - 1) * 0.5;
    c = c > 1 ? 1 : (c < -1 ? -1 : c);
    const double th = acos(c);
    if (s < 1e-5) {
        if (c > 0) return mk(0, 0, 0);
        v3 v;
        v.x = sqrt(std::max((R.a[0] + 1) * 0.5, 0.0));
        v.y = sqrt(std::max((R.a[4] + 1) * 0.5, 0.0)) * (R.a[1] < 0 ? -1.0 : 1.0);
        v.z = sqrt(std::max((R.a[8] + 1) * 0.5, 0.0)) * (R.a[2] < 0 ? -1.0 : 1.0);
        if (fabs(v.x) < fabs(v.y) && fabs(v.x) < fabs(v.z) && (R.a[5] > 0) != (v.y * v.z > 0)) v.z = -v.z;
        return scl(th / nrm(v), v);
    }
    return scl(th / (2 * s), r);
}
void rodrigues_derivative(v3 r, const m3 &R, m3 dR[3]) {  // dR / dr_i in closed form (Gallego & Yezzi 2015)
    const double th2 = dot(r, r);
    for (int i = 0; i < 3; i++) {
        const v3 e = mk(i == 0, i == 1, i == 2);
        if (th2 < 1e-24) { dR[i] = skew(e); continue; }
        const v3 w = cross(r, mul(sub(eye(), R), e));
        dR[i] = scl(1.0 / th2, mul(add(scl(get(r, i), skew(r)), skew(w)), R));
    }
}

// ------------------------------------------------------------------------------------------------ EPnP (5-point models and inlier refit)
struct EpnpSolver {
    int n = 0;
    std::vector<v3> pw, pc;
    std::vector<std::array<double, 2>> uv;
    std::vector<std::array<double, 4>> alpha;
    v3 cw[4], cc[4];

    void control_points() {
        cw[0] = mk(0, 0, 0);
        for (int i = 0; i < n; i++) cw[0] = add(cw[0], pw[i]);
        cw[0] = scl(1.0 / (double)n, cw[0]);
        m3 C = zero3();
        for (int i = 0; i < n; i++) {
            const v3 d = sub(pw[i], cw[0]);
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) C.a[a * 3 + b] += get(d, a) * get(d, b);
        }
        m3 U, V;
        double dc[3];
        svd_3x3(C, U, dc, V);
        for (int i = 1; i < 4; i++) cw[i] = add(cw[0], scl(sqrt(dc[i - 1] / n), col(U, i - 1)));
    }
    void barycentric() {
        m3 CC;
        for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) CC.a[i * 3 + j - 1] = get(cw[j], i) - get(cw[0], i);
        const m3 Ci = inverse3(CC);
        alpha.resize(n);
        for (int i = 0; i < n; i++) {
            const v3 a = mul(Ci, sub(pw[i], cw[0]));
            alpha[i] = {1.0 - a.x - a.y - a.z, a.x, a.y, a.z};
        }
    }
    double pose_from_betas(const Dense &ut, const double *betas, m3 &R, v3 &t) {
        for (int j = 0; j < 4; j++) cc[j] = mk(0, 0, 0);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                cc[j] = add(cc[j], scl(betas[i], mk(ut(11 - i, 3 * j), ut(11 - i, 3 * j + 1), ut(11 - i, 3 * j + 2))));
        pc.resize(n);
        for (int i = 0; i < n; i++) {
            v3 p = mk(0, 0, 0);
            for (int j = 0; j < 4; j++) p = add(p, scl(alpha[i][j], cc[j]));
            pc[i] = p;
        }
        if (pc[0].z < 0.0) {
            for (int j = 0; j < 4; j++) cc[j] = neg(cc[j]);
            for (int i = 0; i < n; i++) pc[i] = neg(pc[i]);
        }
        // absolute orientation (Horn / Arun): R = U V^T of sum (pc - pc0)(pw - pw0)^T
        v3 pc0 = mk(0, 0, 0), pw0 = mk(0, 0, 0);
        for (int i = 0; i < n; i++) { pc0 = add(pc0, pc[i]); pw0 = add(pw0, pw[i]); }
        pc0 = scl(1.0 / (double)n, pc0); pw0 = scl(1.0 / (double)n, pw0);
        m3 ABt = zero3();
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) ABt.a[j * 3 + k] += (get(pc[i], j) - get(pc0, j)) * (get(pw[i], k) - get(pw0, k));
        m3 U, V;
        double d[3];
        svd_3x3(ABt, U, d, V);
        R = mul(U, tr(V));
        if (det(R) < 0) for (int j = 0; j < 3; j++) R.a[6 + j] = -R.a[6 + j];
        t = sub(pc0, mul(R, pw0));
        double s = 0;
        for (int i = 0; i < n; i++) {
            const v3 Y = add(mul(R, pw[i]), t);
            const double ue = Y.x / Y.z, ve = Y.y / Y.z;
            s += sqrt((uv[i][0] - ue) * (uv[i][0] - ue) + (uv[i][1] - ve) * (uv[i][1] - ve));
        }
        return s / n;
    }
    // The six control-point distance constraints |sum_p beta_p (v_p^a - v_p^b)|^2 = |c_a - c_b|^2 are quadratic forms in beta:
    // beta^T Q_i beta = rho_i with Q_i(p, q) = d_p^i . d_q^i (d_p^i = difference of control points a, b of null vector p).
    struct QuadForm { double q[4][4]; };
    // column order of the linearised unknowns the paper uses: B11 B12 B22 B13 B23 B33 B14 B24 B34 B44
    static void beta_pair(int col, int &p, int &q) {
        static const int P[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3}, Q[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};
        p = P[col]; q = Q[col];
    }
    // five Gauss-Newton steps on e_i(beta) = rho_i - beta^T Q_i beta (de_i / dbeta = -2 Q_i beta)
    static void refine_betas(const QuadForm *Qf, const double *rho, double *beta) {
        for (int it = 0; it < 5; it++) {
            Dense Jm(6, 4);
            std::vector<double> e(6);
            for (int i = 0; i < 6; i++) {
                double quad = 0;
                for (int p = 0; p < 4; p++) {
                    double Qb = 0;
                    for (int q = 0; q < 4; q++) Qb += Qf[i].q[p][q] * beta[q];
                    Jm(i, p) = 2 * Qb;
                    quad += beta[p] * Qb;
                }
                e[i] = rho[i] - quad;
            }
            const std::vector<double> step = least_squares(Jm, e);
            for (int k = 0; k < 4; k++) beta[k] += step[k];
        }
    }
    bool solve(m3 &Rout, v3 &tout) {
        control_points();
        barycentric();
        Dense M(2 * n, 12);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 4; j++) {
                M(2 * i, 3 * j) = alpha[i][j]; M(2 * i, 3 * j + 2) = alpha[i][j] * (0.0 - uv[i][0]);
                M(2 * i + 1, 3 * j + 1) = alpha[i][j]; M(2 * i + 1, 3 * j + 2) = alpha[i][j] * (0.0 - uv[i][1]);
            }
        Dense MtM(12, 12), V;
        for (int a = 0; a < 12; a++) for (int b = 0; b < 12; b++) { double s = 0; for (int k = 0; k < 2 * n; k++) s += M(k, a) * M(k, b); MtM(a, b) = s; }
        std::vector<double> w;
        sym_eigen(MtM, w, V);
        Dense ut(12, 12);                      // rows = singular vectors, descending: row 11 belongs to the smallest
        for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) ut(r, c) = V(c, 11 - r);
        // quadratic forms of the six constraints (pairs (a, b) of control points in the order 01 02 03 12 13 23) and, from them, the
        // 6 x 10 matrix of the linearised problem (off-diagonal products appear twice in beta^T Q beta)
        QuadForm Qf[6];
        Dense L(6, 10);
        double rho[6];
        {
            static const int CA[6] = {0, 0, 0, 1, 1, 2}, CB[6] = {1, 2, 3, 2, 3, 3};
            for (int i = 0; i < 6; i++) {
                double d[4][3];
                for (int p = 0; p < 4; p++)
                    for (int k = 0; k < 3; k++) d[p][k] = ut(11 - p, 3 * CA[i] + k) - ut(11 - p, 3 * CB[i] + k);
                for (int p = 0; p < 4; p++)
                    for (int q = 0; q < 4; q++) Qf[i].q[p][q] = d[p][0] * d[q][0] + d[p][1] * d[q][1] + d[p][2] * d[q][2];
                for (int col = 0; col < 10; col++) {
                    int p, q;
                    beta_pair(col, p, q);
                    L(i, col) = (p == q ? 1.0 : 2.0) * Qf[i].q[p][q];
                }
                const v3 dc = sub(cw[CA[i]], cw[CB[i]]);
                rho[i] = dot(dc, dc);
            }
        }
        const std::vector<double> rho_v(rho, rho + 6);
        double betas[3][4], errs[3];
        m3 Rk[3];
        v3 tk[3];
        {   // N = 4 approximation: columns B11 B12 B13 B14
            Dense L4(6, 4);
            for (int i = 0; i < 6; i++) { L4(i, 0) = L(i, 0); L4(i, 1) = L(i, 1); L4(i, 2) = L(i, 3); L4(i, 3) = L(i, 6); }
            const std::vector<double> b4 = least_squares(L4, rho_v);
            double *b = betas[0];
            if (b4[0] < 0) { b[0] = sqrt(-b4[0]); b[1] = -b4[1] / b[0]; b[2] = -b4[2] / b[0]; b[3] = -b4[3] / b[0]; }
            else { b[0] = sqrt(b4[0]); b[1] = b4[1] / b[0]; b[2] = b4[2] / b[0]; b[3] = b4[3] / b[0]; }
        }
        {   // N = 2: B11 B12 B22
            Dense L3(6, 3);
            for (int i = 0; i < 6; i++) { L3(i, 0) = L(i, 0); L3(i, 1) = L(i, 1); L3(i, 2) = L(i, 2); }
            const std::vector<double> b3 = least_squares(L3, rho_v);
            double *b = betas[1];
            if (b3[0] < 0) { b[0] = sqrt(-b3[0]); b[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
            else { b[0] = sqrt(b3[0]); b[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
            if (b3[1] < 0) b[0] = -b[0];
            b[2] = 0; b[3] = 0;
        }
        {   // N = 3: B11 B12 B22 B13 B23
            Dense L5(6, 5);
            for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) L5(i, k) = L(i, k);
            const std::vector<double> b5 = least_squares(L5, rho_v);
            double *b = betas[2];
            if (b5[0] < 0) { b[0] = sqrt(-b5[0]); b[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
            else { b[0] = sqrt(b5[0]); b[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
            if (b5[1] < 0) b[0] = -b[0];
            b[2] = b5[3] / b[0]; b[3] = 0;
        }
        for (int k = 0; k < 3; k++) {
            refine_betas(Qf, rho, betas[k]);
            errs[k] = pose_from_betas(ut, betas[k], Rk[k], tk[k]);
        }
        int N = 0;
        if (errs[1] < errs[0]) N = 1;
        if (errs[2] < errs[N]) N = 2;
        Rout = Rk[N]; tout = tk[N];
        return std::isfinite(tout.x) && std::isfinite(tout.y) && std::isfinite(tout.z) && std::isfinite(errs[N]);
    }
};
bool epnp_subset(const std::vector<v3> &obj, const std::vector<std::array<double, 2>> &img, const std::vector<int> &idx, m3 &R, v3 &t) {
    EpnpSolver e;
    e.n = (int)idx.size();
    e.pw.resize(e.n); e.uv.resize(e.n);
    for (int i = 0; i < e.n; i++) { e.pw[i] = obj[idx[i]]; e.uv[i] = img[idx[i]]; }
    return e.solve(R, t);
}

struct OpenCvRng {  // cv::RNG: multiply-with-carry
    uint64_t state;
    explicit OpenCvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
int ransac_iterations(double p, double ep, int model_points, int max_iters) {  // cv::RANSACUpdateNumIters
    p = std::min(std::max(p, 0.0), 1.0);
    ep = std::min(std::max(ep, 0.0), 1.0);
    double num = std::max(1.0 - p, 2.2250738585072014e-308);
    double denom = 1.0 - pow(1.0 - ep, model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}
// cv::solvePnPRansac(EPNP) with K = I: camera_point = R X + t
bool pnp_ransac_epnp(const std::vector<v3> &obj_in, const std::vector<std::array<double, 2>> &img_in, int max_iters, double thresh, double confidence,
                     m3 &R, v3 &t, std::vector<uint8_t> *inliers_out = nullptr) {
    if (inliers_out) inliers_out->assign(obj_in.size(), 0);
    const int count = (int)obj_in.size(), model_points = 5;
    if (count < model_points) return false;
    std::vector<v3> obj(count);
    std::vector<std::array<double, 2>> img(count);
    for (int i = 0; i < count; i++) {
        obj[i] = mk(as_float(obj_in[i].x), as_float(obj_in[i].y), as_float(obj_in[i].z));
        img[i] = {as_float(img_in[i][0]), as_float(img_in[i][1])};
    }
    OpenCvRng rng((uint64_t)-1);
    const double th2 = thresh * thresh;
    int niters = max_iters, best = 0;
    m3 bestR = eye();
    v3 bestt = mk(0, 0, 0);
    std::vector<uint8_t> mask(count), best_mask(count, 0);
    for (int iter = 0; iter < niters; iter++) {
        std::vector<int> idx(model_points);
        {
            int i = 0, tries = 0;
            for (; tries < 1000; tries++) {
                for (i = 0; i < model_points && tries < 1000;) {
                    const int cand = rng.uniform(0, count);
                    int j = 0;
                    for (; j < i; j++) if (cand == idx[j]) break;
                    if (j < i) continue;
                    idx[i++] = cand;
                }
                if (i == model_points) break;
            }
            if (i < model_points) { if (iter == 0) return false; break; }
        }
        m3 Rm;
        v3 tm;
        if (!epnp_subset(obj, img, idx, Rm, tm)) continue;
        int good = 0;
        for (int i = 0; i < count; i++) {
            const v3 Y = add(mul(Rm, obj[i]), tm);
            const float px = (float)(Y.x / Y.z), py = (float)(Y.y / Y.z);
            const float dx = (float)img[i][0] - px, dy = (float)img[i][1] - py;
            mask[i] = (dx * dx + dy * dy) <= (float)th2;
            good += mask[i];
        }
        if (good > std::max(best, model_points - 1)) {
            bestR = Rm; bestt = tm; best_mask = mask; best = good;
            niters = ransac_iterations(confidence, (double)(count - good) / count, model_points, niters);
        }
    }
    if (best <= 0) return false;
    std::vector<int> in_idx;
    for (int i = 0; i < count; i++) if (best_mask[i]) in_idx.push_back(i);
    if (inliers_out) *inliers_out = best_mask;
    if (!epnp_subset(obj, img, in_idx, R, t)) { R = bestR; t = bestt; }
    return true;
}

// ------------------------------------------------------------------------------------------------ SfM containers
struct Track {                      // SFMFeature (initial_sfm.h:12-22)
    bool solved = false;
    int id = 0, start = 0;
    std::vector<std::array<double, 3>> obs;   // x, y, depth per consecutive frame
    v3 X = mk(0, 0, 0);
    int n() const { return (int)obs.size(); }
    bool sees(int frame) const { return frame >= start && frame < start + n(); }
    const std::array<double, 3> &at(int frame) const { return obs[frame - start]; }
};

// Estimator::relativePose (estimator.cpp:884-920) with getCorrespondingWithDepth (feature_manager.cpp:168-195) and
// solveRelativeRT_PNP (solve_5pts.cpp:248-294): first window frame with > 20 depth-valid matches and > 30 px mean parallax against
// the newest frame; its 3-D points against the newest frame's normalised points give the relative pose
bool relative_pose(int W, const std::vector<Track> &tracks, m3 &rel_R, v3 &rel_T, int &l) {
    for (int i = 0; i < W; i++) {
        std::vector<v3> a3, b3;
        for (const Track &f : tracks) {
            if (f.obs.empty() || !(f.start <= i && f.start + f.n() - 1 >= W)) continue;
            const std::array<double, 3> &oa = f.at(i), &ob = f.at(W);
            if (oa[2] < 0.1 || oa[2] > 10) continue;
            if (ob[2] < 0.1 || ob[2] > 10) continue;
            a3.push_back(mk(oa[0] * oa[2], oa[1] * oa[2], oa[2]));
            b3.push_back(mk(ob[0] * ob[2], ob[1] * ob[2], ob[2]));
        }
        if (a3.size() <= 20) continue;
        double sum = 0;
        for (size_t k = 0; k < a3.size(); k++) {
            const double dx = a3[k].x / a3[k].z - b3[k].x / b3[k].z, dy = a3[k].y / a3[k].z - b3[k].y / b3[k].z;
            sum += sqrt(dx * dx + dy * dy);
        }
        if (!(sum / (int)a3.size() * 460 > 30)) continue;
        std::vector<v3> X;
        std::vector<std::array<double, 2>> u;
        for (size_t k = 0; k < a3.size(); k++)
            if (a3[k].z > 0 && b3[k].z > 0) { X.push_back(a3[k]); u.push_back({b3[k].x / b3[k].z, b3[k].y / b3[k].z}); }
        m3 R = eye();
        v3 t = mk(0, 0, 0);
        pnp_ransac_epnp(X, u, 100, 1.0 / 460, 0.99, R, t);   // the reference ignores the return value
        rel_R = tr(R);
        rel_T = neg(mul(tr(R), t));
        l = i;
        return true;
    }
    return false;
}

// GlobalSFM::triangulateTwoFramesWithDepth (initial_sfm.cpp:113-171)
void triangulate_with_depth(int f0, const m3 &R0, v3 t0, int f1, const m3 &R1, v3 t1, std::vector<Track> &tracks) {
    for (Track &f : tracks) {
        if (f.solved) continue;
        bool has0 = false, has1 = false;
        v3 p0 = mk(0, 0, 0);
        double u1 = 0, v1 = 0;
        for (int k = 0; k < f.n(); k++) {
            const double d = f.obs[k][2];
            if (d < 0.1 || d > 10) continue;
            if (f.start + k == f0) { p0 = mk(f.obs[k][0] * d, f.obs[k][1] * d, d); has0 = true; }
            if (f.start + k == f1) { u1 = f.obs[k][0]; v1 = f.obs[k][1]; has1 = true; }
        }
        if (!(has0 && has1)) continue;
        const v3 X = sub(mul(tr(R0), p0), mul(tr(R0), t0));
        const v3 rp = add(mul(R1, X), t1);
        const double rx = u1 - rp.x / rp.z, ry = v1 - rp.y / rp.z;
        if (sqrt(rx * rx + ry * ry) < 1.0 / 460) { f.solved = true; f.X = X; }
    }
}
// GlobalSFM::solveFrameByPnP (initial_sfm.cpp:22-71)
bool frame_by_pnp(m3 &R, v3 &t, int frame, const std::vector<Track> &tracks) {
    std::vector<v3> X;
    std::vector<std::array<double, 2>> u;
    for (const Track &f : tracks)
        if (f.solved && f.sees(frame)) { u.push_back({f.at(frame)[0], f.at(frame)[1]}); X.push_back(f.X); }
    if ((int)u.size() < 10) return false;    // "unstable features tracking" below 15 is only a message
    return solve_pnp_iterative(X, u, R, t);
}

quat quaternion_plus(quat x, v3 d) {  // ceres::QuaternionParameterization::Plus
    const double nd = nrm(d);
    if (!(nd > 0.0)) return x;
    const double s = sin(nd) / nd;
    return qmul(mkq(cos(nd), s * d.x, s * d.y, s * d.z), x);
}

// The full bundle adjustment of GlobalSFM::construct (initial_sfm.cpp:330-396): reprojection error on the normalised plane for
// every observation of every solved track; rotation of frame l and the translations of frames l and frame_num - 1 constant
struct BundleAdjust {
    int nf, l, nc = 0, np = 0;
    std::vector<quat> q;
    std::vector<v3> t, X;
    std::vector<int> roff, toff;
    struct Ob { int frame, p; double u, v; };
    std::vector<Ob> ob;
    std::vector<std::array<double, 18>> J;    // 2 x [rot(3) trans(3) point(3)]
    std::vector<std::vector<int>> obs_of;      // residuals per point (built on first use)
    std::vector<double> r;

    int column(const Ob &o, int j) const {
        if (j < 3) return roff[o.frame] < 0 ? -1 : roff[o.frame] + j;
        if (j < 6) return toff[o.frame] < 0 ? -1 : toff[o.frame] + (j - 3);
        return nc + 3 * o.p + (j - 6);
    }
    double evaluate(const std::vector<quat> &qq, const std::vector<v3> &tt, const std::vector<v3> &pp, std::vector<double> &res, bool jac) {
        double cost = 0;
        res.resize(2 * ob.size());
        if (jac) J.resize(ob.size());
        for (size_t k = 0; k < ob.size(); k++) {
            const Ob &o = ob[k];
            const quat qc = qq[o.frame];
            const v3 P = pp[o.p];
            const double nq = sqrt(qc.w * qc.w + qc.x * qc.x + qc.y * qc.y + qc.z * qc.z);
            const quat u = mkq(qc.w / nq, qc.x / nq, qc.y / nq, qc.z / nq);
            const v3 pc = add(qrot(u, P), tt[o.frame]);
            const double iz = 1.0 / pc.z, xp = pc.x * iz, yp = pc.y * iz;
            res[2 * k] = xp - o.u; res[2 * k + 1] = yp - o.v;
            cost += 0.5 * (res[2 * k] * res[2 * k] + res[2 * k + 1] * res[2 * k + 1]);
            if (!jac) continue;
            const double drdp[2][3] = {{iz, 0, -xp * iz}, {0, iz, -yp * iz}};
            const v3 v = qvec(u);
            const v3 vxP = cross(v, P);
            m3 dpdv = scl(-2.0 * u.w, skew(P));
            const double vP = dot(v, P);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) dpdv.a[i * 3 + j] += 2.0 * ((i == j ? vP : 0.0) + get(v, i) * get(P, j) - 2.0 * get(P, i) * get(v, j));
            double dpdq[3][4];
            for (int i = 0; i < 3; i++) { dpdq[i][0] = 2.0 * get(vxP, i); for (int j = 0; j < 3; j++) dpdq[i][1 + j] = dpdv.a[i * 3 + j]; }
            const double Jl[4][3] = {{-u.x, -u.y, -u.z}, {u.w, u.z, -u.y}, {-u.z, u.w, u.x}, {u.y, -u.x, u.w}};   // local parameterisation
            double dpdth[3][3];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int m = 0; m < 4; m++) s += dpdq[i][m] * Jl[m][j]; dpdth[i][j] = s; }
            const m3 Ru = q2R(u);
            std::array<double, 18> &out = J[k];
            for (int a = 0; a < 2; a++)
                for (int j = 0; j < 3; j++) {
                    double sr = 0, sp = 0;
                    for (int i = 0; i < 3; i++) { sr += drdp[a][i] * dpdth[i][j]; sp += drdp[a][i] * Ru.a[i * 3 + j]; }
                    out[a * 9 + j] = sr; out[a * 9 + 3 + j] = drdp[a][j]; out[a * 9 + 6 + j] = sp;
                }
        }
        return cost;
    }
    double gradient_max(int ntot) const {
        std::vector<double> g(ntot, 0.0);
        for (size_t k = 0; k < ob.size(); k++)
            for (int a = 0; a < 2; a++) for (int j = 0; j < 9; j++) { const int c = column(ob[k], j); if (c >= 0) g[c] += J[k][a * 9 + j] * r[2 * k + a]; }
        double m = 0;
        for (double v : g) m = std::max(m, fabs(v));
        return m;
    }
    // returns converged; cost_out = final cost
    bool solve(int &iterations, double &cost_out) {
        const int ntot = nc + 3 * np;
        double cost = evaluate(q, t, X, r, true);
        std::vector<double> scale(ntot, 1.0), rc;
        {
            std::vector<double> cn(ntot, 0.0);
            for (size_t k = 0; k < ob.size(); k++)
                for (int a = 0; a < 2; a++) for (int j = 0; j < 9; j++) { const int c = column(ob[k], j); if (c >= 0) cn[c] += J[k][a * 9 + j] * J[k][a * 9 + j]; }
            for (int c = 0; c < ntot; c++) scale[c] = 1.0 / (1.0 + sqrt(cn[c]));
        }
        bool converged = false;
        iterations = 0;
        int invalid = 0;
        double radius = 1e4, decrease = 2.0;
        if (ntot == 0 || ob.empty()) converged = true;
        else if (gradient_max(ntot) <= 1e-10) converged = true;
        while (!converged && iterations < 50) {
            iterations++;
            // Normal equations of the column-scaled problem, eliminated point by point (the sparse bundle-adjustment form of the Schur
            // complement; the dense camera-point block is never built).  For point p observed by the residuals k in obs_of[p]:
            //   V_p = sum_k Jp_k^T Jp_k + damping,  g_p = sum_k Jp_k^T r_k,  W_k = Jc_k^T Jp_k (6 x 3, the observing camera's columns)
            //   S = sum_k Jc_k^T Jc_k + damping - sum_p sum_{k, k'} W_k V_p^-1 W_k'^T,   rhs = -(g_c - sum_p sum_k W_k V_p^-1 g_p)
            if (obs_of.empty() && np > 0) {
                obs_of.assign(np, std::vector<int>());
                for (size_t k = 0; k < ob.size(); k++) obs_of[ob[k].p].push_back((int)k);
            }
            auto damp = [&](double d) { return std::min(std::max(d, 1e-6), 1e32) / radius; };
            Dense S(nc, nc);
            std::vector<double> rhs(nc, 0.0), dx(ntot, 0.0);
            struct CamRow { int col[6]; double Jc[2][6]; };
            std::vector<CamRow> cam(ob.size());
            for (size_t k = 0; k < ob.size(); k++) {
                CamRow &cr = cam[k];
                for (int j = 0; j < 6; j++) {
                    cr.col[j] = column(ob[k], j);
                    for (int a = 0; a < 2; a++) cr.Jc[a][j] = cr.col[j] >= 0 ? J[k][a * 9 + j] * scale[cr.col[j]] : 0.0;
                }
                for (int i = 0; i < 6; i++) {
                    if (cr.col[i] < 0) continue;
                    rhs[cr.col[i]] -= cr.Jc[0][i] * r[2 * k] + cr.Jc[1][i] * r[2 * k + 1];
                    for (int j = 0; j < 6; j++)
                        if (cr.col[j] >= 0) S(cr.col[i], cr.col[j]) += cr.Jc[0][i] * cr.Jc[0][j] + cr.Jc[1][i] * cr.Jc[1][j];
                }
            }
            for (int c = 0; c < nc; c++) S(c, c) += damp(S(c, c));
            std::vector<m3> Vinv(np);
            std::vector<v3> gpt(np);
            std::vector<std::array<double, 18>> Wk(ob.size());   // W_k, 6 x 3 row-major
            bool lin_ok = true;
            for (int p = 0; p < np && lin_ok; p++) {
                m3 Vp = zero3();
                double gp[3] = {0, 0, 0};
                for (int k : obs_of[p]) {
                    double Jp[2][3];
                    for (int a = 0; a < 2; a++) for (int j = 0; j < 3; j++) Jp[a][j] = J[k][a * 9 + 6 + j] * scale[nc + 3 * p + j];
                    for (int i = 0; i < 3; i++) {
                        gp[i] += Jp[0][i] * r[2 * k] + Jp[1][i] * r[2 * k + 1];
                        for (int j = 0; j < 3; j++) Vp.a[i * 3 + j] += Jp[0][i] * Jp[0][j] + Jp[1][i] * Jp[1][j];
                    }
                    for (int i = 0; i < 6; i++) for (int j = 0; j < 3; j++) Wk[k][i * 3 + j] = cam[k].Jc[0][i] * Jp[0][j] + cam[k].Jc[1][i] * Jp[1][j];
                }
                for (int i = 0; i < 3; i++) Vp.a[i * 4] += damp(Vp.a[i * 4]);
                if (!(fabs(det(Vp)) > 0)) { lin_ok = false; break; }
                Vinv[p] = inverse3(Vp);
                gpt[p] = mk(gp[0], gp[1], gp[2]);
                for (int k : obs_of[p]) {
                    double Y[6][3];   // W_k V_p^-1
                    for (int i = 0; i < 6; i++) for (int j = 0; j < 3; j++)
                        Y[i][j] = Wk[k][i * 3] * Vinv[p].a[j] + Wk[k][i * 3 + 1] * Vinv[p].a[3 + j] + Wk[k][i * 3 + 2] * Vinv[p].a[6 + j];
                    for (int i = 0; i < 6; i++) {
                        const int ci = cam[k].col[i];
                        if (ci < 0) continue;
                        rhs[ci] += Y[i][0] * gp[0] + Y[i][1] * gp[1] + Y[i][2] * gp[2];
                        for (int k2 : obs_of[p])
                            for (int j = 0; j < 6; j++) {
                                const int cj = cam[k2].col[j];
                                if (cj >= 0) S(ci, cj) -= Y[i][0] * Wk[k2][j * 3] + Y[i][1] * Wk[k2][j * 3 + 1] + Y[i][2] * Wk[k2][j * 3 + 2];
                            }
                    }
                }
            }
            if (lin_ok) {
                if (nc > 0) { if (cholesky(S)) cholesky_solve(S, rhs); else lin_ok = false; }
                if (lin_ok) {
                    for (int c = 0; c < nc; c++) dx[c] = rhs[c];
                    for (int p = 0; p < np; p++) {   // back-substitution: dx_p = V_p^-1 (-g_p - sum_k W_k^T dx_cam(k))
                        v3 b3 = neg(gpt[p]);
                        for (int k : obs_of[p])
                            for (int i = 0; i < 6; i++) {
                                const int ci = cam[k].col[i];
                                if (ci >= 0) b3 = sub(b3, scl(dx[ci], mk(Wk[k][i * 3], Wk[k][i * 3 + 1], Wk[k][i * 3 + 2])));
                            }
                        const v3 d = mul(Vinv[p], b3);
                        dx[nc + 3 * p] = d.x; dx[nc + 3 * p + 1] = d.y; dx[nc + 3 * p + 2] = d.z;
                    }
                }
            }
            double model_change = 0;
            if (lin_ok)
                for (size_t k = 0; k < ob.size(); k++)
                    for (int a = 0; a < 2; a++) {
                        double mr = 0;
                        for (int j = 0; j < 9; j++) { const int c = column(ob[k], j); if (c >= 0) mr += J[k][a * 9 + j] * scale[c] * dx[c]; }
                        model_change -= mr * (mr / 2 + r[2 * k + a]);
                    }
            if (!lin_ok || !(model_change > 0)) {
                if (++invalid >= 5) break;
                radius /= decrease; decrease *= 2;
                continue;
            }
            invalid = 0;
            std::vector<quat> qn = q;
            std::vector<v3> tn = t, Xn = X;
            double step2 = 0, x2 = 0;
            for (int i = 0; i < nf; i++) {
                if (roff[i] >= 0) {
                    const v3 d = mk(dx[roff[i]] * scale[roff[i]], dx[roff[i] + 1] * scale[roff[i] + 1], dx[roff[i] + 2] * scale[roff[i] + 2]);
                    qn[i] = quaternion_plus(q[i], d);
                    step2 += dot(d, d);
                    x2 += q[i].w * q[i].w + q[i].x * q[i].x + q[i].y * q[i].y + q[i].z * q[i].z;
                }
                if (toff[i] >= 0) {
                    const v3 d = mk(dx[toff[i]] * scale[toff[i]], dx[toff[i] + 1] * scale[toff[i] + 1], dx[toff[i] + 2] * scale[toff[i] + 2]);
                    tn[i] = add(t[i], d);
                    step2 += dot(d, d);
                    x2 += dot(t[i], t[i]);
                }
            }
            for (int p = 0; p < np; p++) {
                const v3 d = mk(dx[nc + 3 * p] * scale[nc + 3 * p], dx[nc + 3 * p + 1] * scale[nc + 3 * p + 1], dx[nc + 3 * p + 2] * scale[nc + 3 * p + 2]);
                Xn[p] = add(X[p], d);
                step2 += dot(d, d);
                x2 += dot(X[p], X[p]);
            }
            std::vector<std::array<double, 18>> Jkeep;
            Jkeep.swap(J);
            const double ccost = evaluate(qn, tn, Xn, rc, false);
            J.swap(Jkeep);
            if (sqrt(step2) <= 1e-8 * (sqrt(x2) + 1e-8)) { converged = true; break; }
            if (fabs(cost - ccost) <= 1e-6 * cost) { converged = true; break; }
            const double rel = (cost - ccost) / model_change;
            if (rel > 1e-3) {
                q = qn; t = tn; X = Xn;
                cost = evaluate(q, t, X, r, true);
                if (gradient_max(ntot) <= 1e-10) { converged = true; break; }
                radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
                decrease = 2.0;
            } else {
                radius /= decrease; decrease *= 2;
                if (radius < 1e-32) { converged = true; break; }
            }
        }
        cost_out = cost;
        return converged;
    }
};

// GlobalSFM::construct (initial_sfm.cpp:184-412): rotations (q: frame -> frame l) and positions of every window frame in the
// frame of camera l, 3-D points of the solved tracks
bool global_sfm(int nf, std::vector<quat> &qw, std::vector<v3> &Tw, int l, const m3 &rel_R, v3 rel_T, std::vector<Track> &tracks, Result &res) {
    qw.assign(nf, mkq(1, 0, 0, 0));
    Tw.assign(nf, mk(0, 0, 0));
    qw[nf - 1] = qmul(qw[l], R2q(rel_R));
    Tw[nf - 1] = rel_T;
    std::vector<m3> cR(nf, eye());
    std::vector<v3> ct(nf, mk(0, 0, 0));
    std::vector<quat> cq(nf, mkq(1, 0, 0, 0));
    auto from_world = [&](int i) { cq[i] = qinv(qw[i]); cR[i] = q2R(cq[i]); ct[i] = neg(mul(cR[i], Tw[i])); };
    from_world(l);
    from_world(nf - 1);
    for (int i = l; i < nf - 1; i++) {
        if (i > l) {
            m3 R0 = cR[i - 1];
            v3 P0 = ct[i - 1];
            if (!frame_by_pnp(R0, P0, i, tracks)) return false;
            cR[i] = R0; ct[i] = P0; cq[i] = R2q(R0);
        }
        triangulate_with_depth(i, cR[i], ct[i], nf - 1, cR[nf - 1], ct[nf - 1], tracks);
    }
    for (int i = l + 1; i < nf - 1; i++) triangulate_with_depth(l, cR[l], ct[l], i, cR[i], ct[i], tracks);
    for (int i = l - 1; i >= 0; i--) {
        m3 R0 = cR[i + 1];
        v3 P0 = ct[i + 1];
        if (!frame_by_pnp(R0, P0, i, tracks)) return false;
        cR[i] = R0; ct[i] = P0; cq[i] = R2q(R0);
        triangulate_with_depth(i, cR[i], ct[i], l, cR[l], ct[l], tracks);
    }
    for (Track &f : tracks) {   // everything else: first observation's depth, checked in the last observing frame (initial_sfm.cpp:283-327)
        if (f.solved || f.n() < 2) continue;
        const double d = f.obs[0][2];
        if (d < 0.1 || d > 10) continue;
        const int f0 = f.start, f1 = f.start + f.n() - 1;
        const v3 p0 = mk(f.obs[0][0] * d, f.obs[0][1] * d, d);
        const v3 X = sub(mul(tr(cR[f0]), p0), mul(tr(cR[f0]), ct[f0]));
        const v3 rp = add(mul(cR[f1], X), ct[f1]);
        const double rx = f.obs.back()[0] - rp.x / rp.z, ry = f.obs.back()[1] - rp.y / rp.z;
        if (sqrt(rx * rx + ry * ry) < 1.0 / 460) { f.solved = true; f.X = X; }
    }
    BundleAdjust ba;
    ba.nf = nf; ba.l = l; ba.q = cq; ba.t = ct;
    ba.roff.assign(nf, -1); ba.toff.assign(nf, -1);
    for (int i = 0; i < nf; i++) {
        if (i != l) { ba.roff[i] = ba.nc; ba.nc += 3; }
        if (i != l && i != nf - 1) { ba.toff[i] = ba.nc; ba.nc += 3; }
    }
    std::vector<int> owner;
    for (int i = 0; i < (int)tracks.size(); i++) {
        if (!tracks[i].solved) continue;
        const int p = (int)owner.size();
        owner.push_back(i);
        ba.X.push_back(tracks[i].X);
        for (int k = 0; k < tracks[i].n(); k++) ba.ob.push_back({tracks[i].start + k, p, tracks[i].obs[k][0], tracks[i].obs[k][1]});
    }
    ba.np = (int)owner.size();
    double final_cost = 0;
    const bool converged = ba.solve(res.ba_iterations, final_cost);
    res.sfm_points = ba.np;
    if (!(converged || final_cost < 5e-3)) return false;
    for (int p = 0; p < ba.np; p++) tracks[owner[p]].X = ba.X[p];
    for (int i = 0; i < nf; i++) {
        qw[i] = qinv(ba.q[i]);
        Tw[i] = neg(qrot(qw[i], ba.t[i]));
    }
    return true;
}

// IntegrationBase::repropagate(0, bg) on the raw samples of one image frame (integration_base.h:44-54): only what the alignment
// reads is kept (delta_p / delta_q / delta_v / sum_dt and the d(delta_q)/d(bg) block of the Jacobian)
void propagate_frame(ImageFrame &f, const double *bg) {
    PreInt p;
    memset(&p, 0, sizeof(p));
    bf::preint_init(p, ld3(f.lin_acc), ld3(f.lin_gyr), mk(0, 0, 0), ld3(bg));
    static thread_local double F[225], V[270], FJ[225];
    for (size_t k = 0; k < f.dt.size(); k++) {
        const v3 a1 = ld3(&f.acc[3 * k]), g1 = ld3(&f.gyr[3 * k]);
        bf::PreintStep o = bf::preint_midpoint(p, f.dt[k], a1, g1, F, V);
        for (int i = 0; i < 15; i++)
            for (int j = 0; j < 15; j++) { double s = 0; for (int q = 0; q < 15; q++) s += F[i * 15 + q] * p.jac[q * 15 + j]; FJ[i * 15 + j] = s; }
        memcpy(p.jac, FJ, sizeof(FJ));
        st3(p.dp, o.dp); st3(p.dv, o.dv);
        const quat qn = qnormalized(o.dq);
        p.dq[0] = qn.w; p.dq[1] = qn.x; p.dq[2] = qn.y; p.dq[3] = qn.z;
        p.sum_dt += f.dt[k];
        st3(p.acc0, a1); st3(p.gyr0, g1);
    }
    f.sum_dt = p.sum_dt;
    f.delta_p = ld3(p.dp); f.delta_v = ld3(p.dv);
    f.delta_q = mkq(p.dq[0], p.dq[1], p.dq[2], p.dq[3]);
    f.dq_dbg = bf::get33(p.jac, 15, bf::O_R, bf::O_BG);
}

// tmp_A^T tmp_A / tmp_A^T tmp_b of one consecutive frame pair (6 x m, m = 9 or 8), scattered into the normal equations like
// initial_aligment.cpp:213-226 / :376-386: the first 6 columns belong to the velocities of frames i, i + 1, the rest to gravity
struct PairRows {
    int m;
    double A[6][9], b[6];
    explicit PairRows(int m_) : m(m_) { memset(A, 0, sizeof(A)); memset(b, 0, sizeof(b)); }
    void block(int r, int c, const m3 &M) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[r + i][c + j] = M.a[i * 3 + j]; }
    void rhs(int r, v3 v) { b[r] = v.x; b[r + 1] = v.y; b[r + 2] = v.z; }
    void add_to(Dense &H, std::vector<double> &g, int n_state, int i, int ng) const {
        auto idx = [&](int a) { return a < 6 ? i * 3 + a : n_state - ng + (a - 6); };
        for (int a = 0; a < m; a++) {
            for (int c = 0; c < m; c++) { double s = 0; for (int k = 0; k < 6; k++) s += A[k][a] * A[k][c]; H(idx(a), idx(c)) += s; }
            double s = 0;
            for (int k = 0; k < 6; k++) s += A[k][a] * b[k];
            g[idx(a)] += s;
        }
    }
};
void tangent_basis(v3 g0, v3 &b, v3 &c) {  // initial_aligment.cpp:78-91
    const v3 a = scl(1.0 / nrm(g0), g0);
    v3 tmp = mk(0, 0, 1);
    if (a.x == tmp.x && a.y == tmp.y && a.z == tmp.z) tmp = mk(1, 0, 0);
    b = sub(tmp, scl(dot(a, tmp), a));
    b = scl(1.0 / nrm(b), b);
    c = cross(a, b);
}
// RefineGravityWithDepth (initial_aligment.cpp:170-244).  As upstream, A and b are zeroed once before the four iterations and
// multiplied by 1000 inside the loop, so each iteration solves the accumulated, repeatedly rescaled system.
void refine_gravity(const std::vector<ImageFrame> &f, v3 tic, double g_norm, v3 &g, std::vector<double> &x) {
    v3 g0 = scl(g_norm / nrm(g), g);
    const int n = (int)f.size(), n_state = n * 3 + 2;
    Dense A(n_state, n_state);
    std::vector<double> b(n_state, 0.0);
    for (int k = 0; k < 4; k++) {
        v3 lx, ly;
        tangent_basis(g0, lx, ly);
        for (int i = 0; i + 1 < n; i++) {
            const ImageFrame &fi = f[i], &fj = f[i + 1];
            const double dt = fj.sum_dt;
            const m3 RiT = tr(fi.R);
            PairRows pr(8);
            pr.block(0, 0, scl(-dt, eye()));
            {
                const v3 c0 = scl(dt * dt / 2, mul(RiT, lx)), c1 = scl(dt * dt / 2, mul(RiT, ly));
                for (int r = 0; r < 3; r++) { pr.A[r][6] = get(c0, r); pr.A[r][7] = get(c1, r); }
            }
            pr.rhs(0, sub(sub(sub(add(fj.delta_p, mul(RiT, mul(fj.R, tic))), tic), scl(dt * dt / 2, mul(RiT, g0))), mul(RiT, sub(fj.T, fi.T))));
            pr.block(3, 0, neg(eye()));
            pr.block(3, 3, mul(RiT, fj.R));
            {
                const v3 c0 = scl(dt, mul(RiT, lx)), c1 = scl(dt, mul(RiT, ly));
                for (int r = 0; r < 3; r++) { pr.A[3 + r][6] = get(c0, r); pr.A[3 + r][7] = get(c1, r); }
            }
            pr.rhs(3, sub(fj.delta_v, scl(dt, mul(RiT, g0))));
            pr.add_to(A, b, n_state, i, 2);
        }
        for (double &v : A.a) v *= 1000.0;
        for (double &v : b) v *= 1000.0;
        x = ldlt_solve(A, b);
        const v3 ng = add(add(g0, scl(x[n_state - 2], lx)), scl(x[n_state - 1], ly));
        g0 = scl(g_norm / nrm(ng), ng);
    }
    g = g0;
}
// LinearAlignmentWithDepth (initial_aligment.cpp:337-405): body velocities of every image frame and gravity in the SfM frame
bool linear_alignment(const std::vector<ImageFrame> &f, v3 tic, double g_norm, v3 &g, std::vector<double> &x) {
    const int n = (int)f.size(), n_state = n * 3 + 3;
    Dense A(n_state, n_state);
    std::vector<double> b(n_state, 0.0);
    for (int i = 0; i + 1 < n; i++) {
        const ImageFrame &fi = f[i], &fj = f[i + 1];
        const double dt = fj.sum_dt;
        const m3 RiT = tr(fi.R);
        PairRows pr(9);
        pr.block(0, 0, scl(-dt, eye()));
        pr.block(0, 6, scl(dt * dt / 2, RiT));
        pr.rhs(0, sub(sub(add(fj.delta_p, mul(RiT, mul(fj.R, tic))), tic), mul(RiT, sub(fj.T, fi.T))));
        pr.block(3, 0, neg(eye()));
        pr.block(3, 3, mul(RiT, fj.R));
        pr.block(3, 6, scl(dt, RiT));
        pr.rhs(3, fj.delta_v);
        pr.add_to(A, b, n_state, i, 3);
    }
    for (double &v : A.a) v *= 1000.0;
    for (double &v : b) v *= 1000.0;
    x = ldlt_solve(A, b);
    g = mk(x[n_state - 3], x[n_state - 2], x[n_state - 1]);
    if (fabs(nrm(g) - g_norm) > 1.0) return false;
    refine_gravity(f, tic, g_norm, g, x);
    return true;
}

}  // namespace

// cv::solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess = true), K = I (call sites: initial_sfm.cpp:59, estimator.cpp:537,
// feature_manager.cpp:571)
bool solve_pnp_iterative(const std::vector<v3> &obj_in, const std::vector<std::array<double, 2>> &img_in, m3 &R, v3 &t) {
    const int n = (int)obj_in.size();
    if (n < 4) return false;
    std::vector<v3> obj(n);
    std::vector<std::array<double, 2>> img(n);
    for (int i = 0; i < n; i++) {
        obj[i] = mk(as_float(obj_in[i].x), as_float(obj_in[i].y), as_float(obj_in[i].z));
        img[i] = {as_float(img_in[i][0]), as_float(img_in[i][1])};
    }
    double param[6], prev[6];
    {
        const v3 r = rodrigues_inv(R);
        param[0] = r.x; param[1] = r.y; param[2] = r.z; param[3] = t.x; param[4] = t.y; param[5] = t.z;
    }
    std::vector<double> err(2 * n);
    Dense J(2 * n, 6), JtJ(6, 6);
    std::vector<double> JtErr(6);
    auto project = [&](const double *p, bool jac) {
        const v3 r = mk(p[0], p[1], p[2]), tt = mk(p[3], p[4], p[5]);
        const m3 Rm = rodrigues(r);
        m3 dR[3];
        if (jac) rodrigues_derivative(r, Rm, dR);
        for (int i = 0; i < n; i++) {
            const v3 Y = add(mul(Rm, obj[i]), tt);
            const double iz = 1.0 / Y.z, x = Y.x * iz, y = Y.y * iz;
            err[2 * i] = x - img[i][0];
            err[2 * i + 1] = y - img[i][1];
            if (!jac) continue;
            for (int k = 0; k < 3; k++) {
                const v3 d = mul(dR[k], obj[i]);
                J(2 * i, k) = iz * d.x - x * iz * d.z;
                J(2 * i + 1, k) = iz * d.y - y * iz * d.z;
            }
            J(2 * i, 3) = iz; J(2 * i, 4) = 0; J(2 * i, 5) = -x * iz;
            J(2 * i + 1, 3) = 0; J(2 * i + 1, 4) = iz; J(2 * i + 1, 5) = -y * iz;
        }
    };
    auto norm2 = [&]() { double s = 0; for (double x : err) s += x * x; return sqrt(s); };
    int lambda_lg10 = -3, iters = 0;
    double prev_err = 0;
    auto take_step = [&]() {
        const double lambda = exp(lambda_lg10 * 2.302585092994046);
        Dense N = JtJ;
        for (int i = 0; i < 6; i++) N(i, i) *= 1.0 + lambda;
        const std::vector<double> d = solve_sym_pinv(N, JtErr);
        for (int i = 0; i < 6; i++) param[i] = prev[i] - d[i];
    };
    for (;;) {
        project(param, true);
        for (int a = 0; a < 6; a++) {
            for (int b = 0; b < 6; b++) { double s = 0; for (int k = 0; k < 2 * n; k++) s += J(k, a) * J(k, b); JtJ(a, b) = s; }
            double s = 0;
            for (int k = 0; k < 2 * n; k++) s += J(k, a) * err[k];
            JtErr[a] = s;
        }
        for (int i = 0; i < 6; i++) prev[i] = param[i];
        take_step();
        if (iters == 0) prev_err = norm2();
        bool done = false;
        for (;;) {
            project(param, false);
            const double e = norm2();
            if (e > prev_err && ++lambda_lg10 <= 16) { take_step(); continue; }
            lambda_lg10 = std::max(lambda_lg10 - 1, -16);
            double dn = 0, pn = 0;
            for (int i = 0; i < 6; i++) { dn += (param[i] - prev[i]) * (param[i] - prev[i]); pn += prev[i] * prev[i]; }
            if (++iters >= 20 || sqrt(dn) / sqrt(pn) < 1.1920928955078125e-07) done = true;
            prev_err = e;
            break;
        }
        if (done) break;
    }
    for (int i = 0; i < 6; i++) if (!std::isfinite(param[i])) return false;
    R = rodrigues(mk(param[0], param[1], param[2]));
    t = mk(param[3], param[4], param[5]);
    return true;
}

void run(const vio_config &cfg, int W, const double *headers, const double *bgs0, const double *ric9, const double *tic3,
         std::vector<ImageFrame> &frames, const std::vector<Landmark> &landmarks, Result &out) {
    out = Result();
    const m3 ric = ldm(ric9);
    const v3 tic = ld3(tic3);
    const int nfr = (int)frames.size();
    double bg[3] = {bgs0[0], bgs0[1], bgs0[2]};
    for (ImageFrame &f : frames) propagate_frame(f, f.bg_lin);
    // IMU excitation (estimator.cpp:386-420): only decides whether the accelerometer bias is estimated at the end
    bool excited = false;
    if (nfr > 1) {
        v3 sum = mk(0, 0, 0);
        for (int k = 1; k < nfr; k++) sum = add(sum, scl(1.0 / frames[k].sum_dt, frames[k].delta_v));
        const v3 mean = scl(1.0 / (nfr - 1), sum);
        double var = 0;
        for (int k = 1; k < nfr; k++) { const v3 d = sub(scl(1.0 / frames[k].sum_dt, frames[k].delta_v), mean); var += dot(d, d); }
        if (!(sqrt(var / (nfr - 1)) < 0.25)) excited = true;
    }
    // ---- global SfM over the window (estimator.cpp:422-463)
    std::vector<Track> tracks(landmarks.size());
    for (size_t i = 0; i < landmarks.size(); i++) { tracks[i].id = landmarks[i].id; tracks[i].start = landmarks[i].start; tracks[i].obs = landmarks[i].obs; }
    m3 rel_R;
    v3 rel_T;
    int l = 0;
    if (!relative_pose(W, tracks, rel_R, rel_T, l)) { out.stage = 1; return; }
    std::vector<quat> Q;
    std::vector<v3> Ts;
    if (!global_sfm(W + 1, Q, Ts, l, rel_R, rel_T, tracks, out)) { out.stage = 2; out.force_margin_old = true; return; }
    std::map<int, v3> solved;
    for (const Track &f : tracks) if (f.solved) solved[f.id] = f.X;
    // ---- pose of every image frame: window frames from the SfM, the others by PnP from the neighbouring window frame (:466-548)
    {
        int i = 0;
        for (ImageFrame &f : frames) {
            if (f.stamp == headers[i]) {
                f.is_key_frame = true;
                f.R = mul(q2R(Q[i]), tr(ric));
                f.T = Ts[i];
                i++;
                continue;
            }
            if (f.stamp > headers[i]) i++;
            m3 R0 = q2R(qinv(Q[i]));
            v3 P0 = neg(mul(R0, Ts[i]));
            f.is_key_frame = false;
            std::vector<v3> X;
            std::vector<std::array<double, 2>> u;
            for (size_t k = 0; k < f.ids.size(); k++) {
                auto it = solved.find(f.ids[k]);
                if (it == solved.end()) continue;
                X.push_back(it->second);
                u.push_back({f.xy[2 * k], f.xy[2 * k + 1]});
            }
            if (X.size() < 6 || !solve_pnp_iterative(X, u, R0, P0)) { out.stage = 3; return; }
            const m3 Rp = tr(R0);
            f.R = mul(Rp, tr(ric));
            f.T = mul(Rp, neg(P0));
        }
    }
    // ---- visualInitialAlignWithDepth (estimator.cpp:799-869)
    {   // solveGyroscopeBias over all image frames (initial_aligment.cpp:3-36)
        Dense A(3, 3);
        std::vector<double> b(3, 0.0);
        for (int k = 0; k + 1 < nfr; k++) {
            const ImageFrame &fi = frames[k], &fj = frames[k + 1];
            const quat q_ij = R2q(mul(tr(fi.R), fj.R));
            const m3 tA = fj.dq_dbg;
            const v3 tb = scl(2.0, qvec(qmul(qinv(fj.delta_q), q_ij)));
            const m3 AtA = mul(tr(tA), tA);
            const v3 Atb = mul(tr(tA), tb);
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) A(r, c) += AtA.a[r * 3 + c]; b[r] += get(Atb, r); }
        }
        const std::vector<double> x = ldlt_solve(A, b);
        for (int k = 0; k < 3; k++) { out.delta_bg[k] = x[k]; bg[k] += x[k]; }
        for (int k = 1; k < nfr; k++) { for (int q = 0; q < 3; q++) frames[k].bg_lin[q] = bg[q]; propagate_frame(frames[k], bg); }
    }
    v3 g;
    std::vector<double> x;
    if (!linear_alignment(frames, tic, cfg.g_norm, g, x)) { out.stage = 4; return; }
    // window states (:824-869).  As upstream, the velocity of window frame k is read from x at 3 k with k counting KEY frames,
    // although x is indexed by image frames; the two agree when no image frame was dropped from the window yet.
    std::vector<v3> P(W + 1), V(W + 1);
    std::vector<m3> Rw(W + 1);
    for (int i = 0; i <= W; i++)
        for (const ImageFrame &f : frames)
            if (f.stamp == headers[i]) { P[i] = f.T; Rw[i] = f.R; }
    const v3 p0 = sub(P[0], mul(Rw[0], tic));
    for (int i = W; i >= 0; i--) P[i] = sub(sub(P[i], mul(Rw[i], tic)), p0);
    for (int k = 0; k <= W; k++) V[k] = mul(Rw[k], mk(x[3 * k], x[3 * k + 1], x[3 * k + 2]));
    m3 R0 = g2R(g);
    const double yaw = R2ypr(mul(R0, Rw[0])).x;
    R0 = mul(ypr2R(mk(-yaw, 0, 0)), R0);
    g = mul(R0, g);
    for (int i = 0; i <= W; i++) {
        st3(out.Ps[i], mul(R0, P[i]));
        stm(out.Rs[i], mul(R0, Rw[i]));
        st3(out.Vs[i], mul(R0, V[i]));
    }
    st3(out.g, g);
    if (!excited) {   // estimator.cpp:552-570
        v3 sum = mk(0, 0, 0);
        for (int k = 1; k < nfr; k++) sum = add(sum, scl(1.0 / frames[k].sum_dt, frames[k].delta_v));
        const v3 avg = scl(1.0 / (nfr - 1), sum);
        st3(out.Ba, sub(avg, mul(tr(g2R(avg)), mk(0, 0, cfg.g_norm))));
        out.set_ba = true;
    }
    out.ok = true;
}

// ---- host-only stage entry points (vio_stage_host_*): the building blocks above on caller-supplied arrays, for parity tests that
// need no GPU
// the same with the inlier mask (KeyFrame::PnPRANSAC reads it, pose_graph/src/keyframe/keyframe.cpp:231-238)
bool pnp_ransac_with_inliers(const std::vector<dm::v3> &obj, const std::vector<std::array<double, 2>> &img, int max_iters, double thresh, double confidence,
                             dm::m3 &R, dm::v3 &t, std::vector<uint8_t> &inliers) {
    return pnp_ransac_epnp(obj, img, max_iters, thresh, confidence, R, t, &inliers);
}
bool stage_pnp_ransac_epnp(int n, const double *obj, const double *img, int max_iters, double thresh, double confidence, double *R9, double *t3) {
    std::vector<v3> o(n);
    std::vector<std::array<double, 2>> im(n);
    for (int i = 0; i < n; i++) { o[i] = mk(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2]); im[i] = {img[2 * i], img[2 * i + 1]}; }
    m3 R = eye();
    v3 t = mk(0, 0, 0);
    const bool ok = pnp_ransac_epnp(o, im, max_iters, thresh, confidence, R, t);
    stm(R9, R); st3(t3, t);
    return ok;
}
// visual-inertial alignment: frames = n x {R[9] row-major, T[3], sum_dt, delta_p[3], delta_v[3]} (19 doubles)
bool stage_alignment(int n, const double *frames19, const double *tic3, double g_norm, double *g_out, double *x_out) {
    std::vector<ImageFrame> f(n);
    for (int i = 0; i < n; i++) {
        const double *p = frames19 + 19 * i;
        f[i].R = ldm(p);
        f[i].T = ld3(p + 9);
        f[i].sum_dt = p[12];
        f[i].delta_p = ld3(p + 13);
        f[i].delta_v = ld3(p + 16);
    }
    v3 g = mk(0, 0, 0);
    std::vector<double> x;
    const bool ok = linear_alignment(f, ld3(tic3), g_norm, g, x);
    st3(g_out, g);
    for (size_t i = 0; i < x.size() && i < (size_t)(3 * n + 3); i++) x_out[i] = x[i];
    return ok;
}
int stage_sfm_window(int window_size, int nf, const int *start, const int *nobs, const double *obs, int *l_out, double *q_out, double *T_out,
                     double *pts_out, double *stats_out) {
    std::vector<Track> tracks(nf);
    size_t off = 0;
    for (int i = 0; i < nf; i++) {
        tracks[i].id = i; tracks[i].start = start[i];
        for (int k = 0; k < nobs[i]; k++, off++) tracks[i].obs.push_back({obs[3 * off], obs[3 * off + 1], obs[3 * off + 2]});
    }
    m3 rel_R;
    v3 rel_T;
    int l = -1;
    if (!relative_pose(window_size, tracks, rel_R, rel_T, l)) return 1;
    *l_out = l;
    std::vector<quat> Q;
    std::vector<v3> Ts;
    Result res;
    if (!global_sfm(window_size + 1, Q, Ts, l, rel_R, rel_T, tracks, res)) return 2;
    for (int i = 0; i <= window_size; i++) {
        q_out[4 * i] = Q[i].w; q_out[4 * i + 1] = Q[i].x; q_out[4 * i + 2] = Q[i].y; q_out[4 * i + 3] = Q[i].z;
        st3(T_out + 3 * i, Ts[i]);
    }
    for (int i = 0; i < nf; i++) { pts_out[4 * i] = tracks[i].solved ? 1.0 : 0.0; st3(pts_out + 4 * i + 1, tracks[i].X); }
    stats_out[0] = res.ba_iterations; stats_out[1] = res.sfm_points;
    return 0;
}

}  // namespace vinit

extern "C" {
int vio_stage_host_pnp(int n, const double *obj, const double *img, double *R9, double *t3) {
    if (n < 1 || !obj || !img || !R9 || !t3) return -1;
    std::vector<dm::v3> o(n);
    std::vector<std::array<double, 2>> im(n);
    for (int i = 0; i < n; i++) { o[i] = dm::mk(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2]); im[i] = {img[2 * i], img[2 * i + 1]}; }
    dm::m3 R = dm::ldm(R9);
    dm::v3 t = dm::ld3(t3);
    const bool ok = vinit::solve_pnp_iterative(o, im, R, t);
    dm::stm(R9, R); dm::st3(t3, t);
    return ok ? 1 : 0;
}
int vio_stage_host_pnp_ransac(int n, const double *obj, const double *img, int max_iters, double thresh, double confidence, double *R9, double *t3) {
    if (n < 1 || !obj || !img || !R9 || !t3) return -1;
    return vinit::stage_pnp_ransac_epnp(n, obj, img, max_iters, thresh, confidence, R9, t3) ? 1 : 0;
}
int vio_stage_host_alignment(int n, const double *frames19, const double *tic3, double g_norm, double *g_out3, double *x_out) {
    if (n < 2 || !frames19 || !tic3 || !g_out3 || !x_out) return -1;
    return vinit::stage_alignment(n, frames19, tic3, g_norm, g_out3, x_out) ? 1 : 0;
}
int vio_stage_host_sfm_window(int window_size, int nf, const int32_t *start, const int32_t *nobs, const double *obs, int32_t *l_out, double *q_out,
                              double *T_out, double *pts_out, double *stats_out) {
    if (window_size < 2 || window_size > VIO_MAXW || nf < 1 || !start || !nobs || !obs || !l_out || !q_out || !T_out || !pts_out || !stats_out) return -1;
    return vinit::stage_sfm_window(window_size, nf, start, nobs, obs, l_out, q_out, T_out, pts_out, stats_out);
}
}
