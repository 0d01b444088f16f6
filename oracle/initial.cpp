// ORACLE (test infrastructure only — never linked into or called by the product path).
// CPU restatement of the visual-inertial alignment step of the dynamic (static_init: 0) initialisation, SURVEY.md §8f rank 1:
//   TangentBasis                   vins_estimator/src/initial/initial_aligment.cpp:78-91
//   RefineGravityWithDepth         initial_aligment.cpp:170-244
//   LinearAlignmentWithDepth       initial_aligment.cpp:337-405
//   the state hand-over at the end of Estimator::visualInitialAlignWithDepth   estimator/estimator.cpp:839-869
// The structure-from-motion front of that initialisation (relativePose / solveRelativeRT_PNP / GlobalSFM::construct,
// estimator.cpp:384-463, 884-920) follows further down with restatements of cv::solvePnP, cv::solvePnPRansac (EPnP) and the
// Ceres bundle adjustment, and Estimator::initialStructure / visualInitialAlignWithDepth at the end of the file; the
// static_init: 0 branch of processImage and the all_image_frame bookkeeping live in backend.cpp (Config::dynamic_init).
// Parity status: "parity unpinned" (no reference tests / golden vectors; Eigen's LDLT is un-vendored: restated as a pivoted LDL^T).
#include "oracle.h"

namespace ovio {
using namespace om;

namespace {

// x = A^-1 b for a symmetric positive semi-definite A (n x n, row-major, destroyed).  Eigen::LDLT (the reference's
// `A.ldlt().solve(b)`) pivots on the largest remaining diagonal entry and treats pivots below a tolerance as zero.
std::vector<double> ldlt_solve(std::vector<double> A, std::vector<double> b, int n) {
    std::vector<int> perm(n);
    for (int i = 0; i < n; i++) perm[i] = i;
    std::vector<double> d(n, 0.0);
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++) if (std::fabs(A[i * n + i]) > std::fabs(A[p * n + p])) p = i;
        if (p != k) {
            for (int j = 0; j < n; j++) std::swap(A[k * n + j], A[p * n + j]);
            for (int i = 0; i < n; i++) std::swap(A[i * n + k], A[i * n + p]);
            std::swap(perm[k], perm[p]);
        }
        d[k] = A[k * n + k];
        if (d[k] == 0.0) {  // Eigen stops at an exactly zero pivot (the remaining block is zero); D^-1 then skips those entries
            for (int i = k; i < n; i++) d[i] = 0;
            for (int i = k; i < n; i++) for (int j = k + 1; j < n; j++) if (j > i) A[j * n + i] = 0;
            break;
        }
        for (int i = k + 1; i < n; i++) A[i * n + k] /= d[k];
        for (int i = k + 1; i < n; i++)
            for (int j = k + 1; j <= i; j++) {
                A[i * n + j] -= A[i * n + k] * d[k] * A[j * n + k];
                A[j * n + i] = A[i * n + j];
            }
    }
    std::vector<double> y(n);
    for (int i = 0; i < n; i++) y[i] = b[perm[i]];
    for (int i = 0; i < n; i++) for (int k = 0; k < i; k++) y[i] -= A[i * n + k] * y[k];   // L y = P b
    for (int i = 0; i < n; i++) y[i] = d[i] != 0 ? y[i] / d[i] : 0.0;                      // D z = y
    for (int i = n - 1; i >= 0; i--) for (int k = i + 1; k < n; k++) y[i] -= A[k * n + i] * y[k];  // L^T w = z
    std::vector<double> x(n);
    for (int i = 0; i < n; i++) x[perm[i]] = y[i];
    return x;
}

// tmp_A^T tmp_A and tmp_A^T tmp_b of one frame pair (6 x m block, m = 9 or 8), scattered as the reference does
struct PairBlock {
    int m;
    double A[6][9], b[6];
    PairBlock(int m_) : m(m_) { std::memset(A, 0, sizeof(A)); std::memset(b, 0, sizeof(b)); }
    void set33(int r, int c, const M3 &M) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[r + i][c + j] = M(i, j); }
    void setb(int r, const V3 &v) { b[r] = v.x; b[r + 1] = v.y; b[r + 2] = v.z; }
    void scatter(std::vector<double> &H, std::vector<double> &g, int n_state, int i, int ng) const {
        double rA[9][9], rb[9];
        for (int a = 0; a < m; a++) {
            for (int c = 0; c < m; c++) { double s = 0; for (int k = 0; k < 6; k++) s += A[k][a] * A[k][c]; rA[a][c] = s; }
            double s = 0; for (int k = 0; k < 6; k++) s += A[k][a] * b[k]; rb[a] = s;
        }
        auto idx = [&](int a) { return a < 6 ? i * 3 + a : n_state - ng + (a - 6); };
        for (int a = 0; a < m; a++) {
            for (int c = 0; c < m; c++) H[idx(a) * n_state + idx(c)] += rA[a][c];
            g[idx(a)] += rb[a];
        }
    }
};

}  // namespace

void tangent_basis(const V3 &g0, V3 &b, V3 &c) {  // initial_aligment.cpp:78-91
    V3 a = g0 / norm(g0);
    V3 tmp(0, 0, 1);
    if (a.x == tmp.x && a.y == tmp.y && a.z == tmp.z) tmp = V3(1, 0, 0);
    b = tmp - dot(a, tmp) * a;
    b = b / norm(b);
    c = cross(a, b);
}

// initial_aligment.cpp:170-244.  Quirk kept: A and b are zeroed once, before the four iterations, and scaled by 1000 inside
// the loop, so every iteration solves the accumulated (and repeatedly rescaled) system, exactly as the reference does.
void refine_gravity_with_depth(const std::vector<AlignFrame> &f, const V3 &tic, double g_norm, V3 &g, std::vector<double> &x) {
    V3 g0 = (g_norm / norm(g)) * g;
    const int n = (int)f.size(), n_state = n * 3 + 2;
    std::vector<double> A((size_t)n_state * n_state, 0.0), b(n_state, 0.0);
    for (int k = 0; k < 4; k++) {
        V3 lx, ly;
        tangent_basis(g0, lx, ly);
        for (int i = 0; i + 1 < n; i++) {
            const AlignFrame &fi = f[i], &fj = f[i + 1];
            const double dt = fj.sum_dt;
            const M3 RiT = T(fi.R);
            PairBlock pb(8);
            pb.set33(0, 0, (-dt) * M3::I());
            {
                V3 c0 = (dt * dt / 2) * (RiT * lx), c1 = (dt * dt / 2) * (RiT * ly);
                for (int r = 0; r < 3; r++) { pb.A[r][6] = c0[r]; pb.A[r][7] = c1[r]; }
            }
            pb.setb(0, fj.delta_p + RiT * (fj.R * tic) - tic - (dt * dt / 2) * (RiT * g0) - RiT * (fj.T - fi.T));
            pb.set33(3, 0, -M3::I());
            pb.set33(3, 3, RiT * fj.R);
            {
                V3 c0 = dt * (RiT * lx), c1 = dt * (RiT * ly);
                for (int r = 0; r < 3; r++) { pb.A[3 + r][6] = c0[r]; pb.A[3 + r][7] = c1[r]; }
            }
            pb.setb(3, fj.delta_v - dt * (RiT * g0));
            pb.scatter(A, b, n_state, i, 2);
        }
        for (auto &v : A) v *= 1000.0;
        for (auto &v : b) v *= 1000.0;
        x = ldlt_solve(A, b, n_state);
        V3 ng = g0 + x[n_state - 2] * lx + x[n_state - 1] * ly;
        g0 = (g_norm / norm(ng)) * ng;
    }
    g = g0;
}

// initial_aligment.cpp:337-405: per-frame body velocities and gravity in the SfM reference frame (metric depth: no scale)
bool linear_alignment_with_depth(const std::vector<AlignFrame> &f, const V3 &tic, double g_norm, V3 &g, std::vector<double> &x) {
    const int n = (int)f.size(), n_state = n * 3 + 3;
    std::vector<double> A((size_t)n_state * n_state, 0.0), b(n_state, 0.0);
    for (int i = 0; i + 1 < n; i++) {
        const AlignFrame &fi = f[i], &fj = f[i + 1];
        const double dt = fj.sum_dt;
        const M3 RiT = T(fi.R);
        PairBlock pb(9);
        pb.set33(0, 0, (-dt) * M3::I());
        pb.set33(0, 6, (dt * dt / 2) * RiT);
        pb.setb(0, fj.delta_p + RiT * (fj.R * tic) - tic - RiT * (fj.T - fi.T));
        pb.set33(3, 0, -M3::I());
        pb.set33(3, 3, RiT * fj.R);
        pb.set33(3, 6, dt * RiT);
        pb.setb(3, fj.delta_v);
        pb.scatter(A, b, n_state, i, 3);
    }
    for (auto &v : A) v *= 1000.0;
    for (auto &v : b) v *= 1000.0;
    x = ldlt_solve(A, b, n_state);
    g = V3(x[n_state - 3], x[n_state - 2], x[n_state - 1]);
    if (std::fabs(norm(g) - g_norm) > 1.0) return false;
    refine_gravity_with_depth(f, tic, g_norm, g, x);
    return true;
}

// estimator.cpp:839-869: window positions become body positions relative to frame 0, velocities go to the reference frame,
// then everything is rotated so that gravity points along +z with the yaw of frame 0 removed.
// Ps / Rs enter as the SfM camera positions / body rotations (frame .T / .R of the window frames), x = alignment solution.
// Quirk kept: the reference reads the velocity of window frame k from x.segment<3>(3 k) with k counting KEY frames, although x is
// indexed by all image frames (estimator.cpp:845-852); the two agree when every image frame since start-up is a window frame.
void align_window_to_gravity(int n, V3 *Ps, M3 *Rs, V3 *Vs, const std::vector<double> &x, const V3 &tic, V3 &g) {
    const V3 p0 = Ps[0] - Rs[0] * tic;
    for (int i = n - 1; i >= 0; i--) Ps[i] = Ps[i] - Rs[i] * tic - p0;
    for (int k = 0; k < n; k++) Vs[k] = Rs[k] * V3(x[3 * k], x[3 * k + 1], x[3 * k + 2]);
    M3 R0 = g2R(g);
    double yaw = R2ypr(R0 * Rs[0]).x;
    R0 = ypr2R(V3(-yaw, 0, 0)) * R0;
    g = R0 * g;
    for (int i = 0; i < n; i++) { Ps[i] = R0 * Ps[i]; Rs[i] = R0 * Rs[i]; Vs[i] = R0 * Vs[i]; }
}

}  // namespace ovio

// =====================================================================================================================
// Structure-from-motion front of the dynamic initialisation (estimator.cpp:384-579, initial_sfm.cpp, solve_5pts.cpp:248-294).
// The OpenCV / Ceres routines it calls are un-vendored; they are restated here from their published algorithms:
//   cv::Rodrigues                                   rotation vector <-> matrix, derivative in closed form (Gallego & Yezzi 2015)
//   cv::solvePnP(ITERATIVE, useExtrinsicGuess=1)    CvLevMarq on (rvec, tvec), 20 iterations / FLT_EPSILON (calibration.cpp,
//                                                   compat_ptsetreg.cpp of OpenCV 3.4)
//   cv::solvePnPRansac(EPNP, 100, 1/460, 0.99)      RANSACPointSetRegistrator with OpenCV's RNG(-1), 5-point EPnP models, final
//                                                   EPnP over the inliers (solvepnp.cpp, ptsetreg.cpp, epnp.cpp; Lepetit et al. 2009)
//   ceres::Solve (LM, DENSE_SCHUR) of GlobalSFM     Levenberg-Marquardt trust region with Jacobi scaling, quaternion plus
// Inputs go through float like the reference's cv::Point3f / cv::Point2f containers.  max_solver_time_in_seconds (0.2 s) is not
// modelled (it makes the reference itself non-deterministic).
namespace ovio {
namespace {

inline double f32(double v) { return (double)(float)v; }

M3 rodrigues_to_R(const V3 &r) {
    const double th = norm(r);
    if (th < 2.220446049250313e-16) return M3::I();
    const V3 k = r / th;
    const double c = std::cos(th), s = std::sin(th), c1 = 1 - c;
    M3 R;
    const double kk[3] = {k.x, k.y, k.z};
    const M3 K = skew(k);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R(i, j) = (i == j ? c : 0.0) + c1 * kk[i] * kk[j] + s * K(i, j);
    return R;
}
V3 R_to_rodrigues(const M3 &R) {  // cv::Rodrigues matrix -> vector for a proper rotation (its SVD re-orthonormalisation is skipped)
    V3 r(R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1));
    const double s = std::sqrt((r.x * r.x + r.y * r.y + r.z * r.z) * 0.25);
    double c = (R(0, 0) + R(1, 1) + R(2, 2) - 1) * 0.5;
    c = c > 1 ? 1 : (c < -1 ? -1 : c);
    const double th = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) return V3(0, 0, 0);
        double t = (R(0, 0) + 1) * 0.5;
        V3 v;
        v.x = std::sqrt(std::max(t, 0.0));
        t = (R(1, 1) + 1) * 0.5;
        v.y = std::sqrt(std::max(t, 0.0)) * (R(0, 1) < 0 ? -1.0 : 1.0);
        t = (R(2, 2) + 1) * 0.5;
        v.z = std::sqrt(std::max(t, 0.0)) * (R(0, 2) < 0 ? -1.0 : 1.0);
        if (std::fabs(v.x) < std::fabs(v.y) && std::fabs(v.x) < std::fabs(v.z) && (R(1, 2) > 0) != (v.y * v.z > 0)) v.z = -v.z;
        return (th / norm(v)) * v;
    }
    return (th / (2 * s)) * r;
}
// dR/dr_i for i = 0..2
void rodrigues_jac(const V3 &r, const M3 &R, M3 dR[3]) {
    const double th2 = dot(r, r);
    for (int i = 0; i < 3; i++) {
        V3 e(i == 0, i == 1, i == 2);
        if (th2 < 1e-24) { dR[i] = skew(e); continue; }
        M3 ImR = M3::I() - R;
        V3 w = cross(r, ImR * e);
        dR[i] = (1.0 / th2) * ((r[i] * skew(r) + skew(w)) * R);
    }
}

// x = pinv(A) b for symmetric A through its eigen-decomposition (stands in for cv::solve(..., DECOMP_SVD) on the normal matrix)
std::vector<double> sym_solve_svd(const Mat &A, const std::vector<double> &b) {
    const int n = A.r;
    Mat Ac = A, V;
    std::vector<double> w;
    sym_eig(Ac, w, V);
    double wmax = 0;
    for (double v : w) wmax = std::max(wmax, std::fabs(v));
    std::vector<double> x(n, 0.0);
    for (int k = 0; k < n; k++) {
        if (!(std::fabs(w[k]) > wmax * 2 * 2.220446049250313e-16 * n)) continue;
        double s = 0;
        for (int i = 0; i < n; i++) s += V(i, k) * b[i];
        s /= w[k];
        for (int i = 0; i < n; i++) x[i] += V(i, k) * s;
    }
    return x;
}
// least squares min |A x - b| (m x n, m >= n) through the normal equations' eigen-decomposition (cvSolve(..., CV_SVD) stand-in)
std::vector<double> lstsq(const Mat &A, const std::vector<double> &b) {
    const int m = A.r, n = A.c;
    Mat N(n, n);
    std::vector<double> g(n, 0.0);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < m; k++) s += A(k, i) * A(k, j); N(i, j) = s; }
        double s = 0; for (int k = 0; k < m; k++) s += A(k, i) * b[k]; g[i] = s;
    }
    return sym_solve_svd(N, g);
}
// singular value decomposition of a 3 x 3 matrix: A = U diag(s) V^T, s descending
void svd3(const M3 &A, M3 &U, double s[3], M3 &V) {
    Mat N(3, 3), Ve;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double t = 0; for (int k = 0; k < 3; k++) t += A(k, i) * A(k, j); N(i, j) = t; }
    std::vector<double> w;
    sym_eig(N, w, Ve);  // ascending
    for (int j = 0; j < 3; j++) {
        const int src = 2 - j;
        s[j] = std::sqrt(std::max(w[src], 0.0));
        for (int i = 0; i < 3; i++) V(i, j) = Ve(i, src);
    }
    V3 u[3];
    for (int j = 0; j < 3; j++) {
        V3 c = A * V.col(j);
        u[j] = s[j] > 1e-12 * std::max(s[0], 1e-300) ? c / s[j] : V3(0, 0, 0);
    }
    // complete a rank-deficient U to an orthonormal basis
    if (norm(u[1]) < 0.5) { V3 h = std::fabs(u[0].x) < 0.9 ? V3(1, 0, 0) : V3(0, 1, 0); u[1] = cross(u[0], h); u[1] = u[1] / norm(u[1]); }
    if (norm(u[2]) < 0.5) u[2] = cross(u[0], u[1]);
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) U(i, j) = u[j][i];
}
double det3(const M3 &R) {
    return R(0, 0) * (R(1, 1) * R(2, 2) - R(1, 2) * R(2, 1)) - R(0, 1) * (R(1, 0) * R(2, 2) - R(1, 2) * R(2, 0)) +
           R(0, 2) * (R(1, 0) * R(2, 1) - R(1, 1) * R(2, 0));
}
M3 inv3(const M3 &A) {
    const double d = det3(A);
    M3 r;
    r(0, 0) = (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) / d; r(0, 1) = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) / d; r(0, 2) = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) / d;
    r(1, 0) = (A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2)) / d; r(1, 1) = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) / d; r(1, 2) = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) / d;
    r(2, 0) = (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0)) / d; r(2, 1) = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) / d; r(2, 2) = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) / d;
    return r;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- solvePnP
// cv::solvePnP(obj, img, K = I, no distortion, rvec, tvec, useExtrinsicGuess = true, SOLVEPNP_ITERATIVE): Levenberg-Marquardt
// refinement of the guess (R, t: camera_point = R X + t).  Call sites: initial_sfm.cpp:59, estimator.cpp:537.
bool solve_pnp_iterative(const std::vector<V3> &obj_in, const std::vector<std::array<double, 2>> &img_in, M3 &R, V3 &t) {
    const int n = (int)obj_in.size();
    if (n < 4) return false;  // cv::solvePnP asserts npoints >= 4
    std::vector<V3> obj(n);
    std::vector<std::array<double, 2>> img(n);
    for (int i = 0; i < n; i++) {
        obj[i] = V3(f32(obj_in[i].x), f32(obj_in[i].y), f32(obj_in[i].z));
        img[i] = {f32(img_in[i][0]), f32(img_in[i][1])};
    }
    double param[6], prev[6];
    {
        V3 r = R_to_rodrigues(R);
        param[0] = r.x; param[1] = r.y; param[2] = r.z; param[3] = t.x; param[4] = t.y; param[5] = t.z;
    }
    auto project = [&](const double *p, std::vector<double> &err, Mat *J) {
        V3 r(p[0], p[1], p[2]), tt(p[3], p[4], p[5]);
        M3 Rm = rodrigues_to_R(r), dR[3];
        if (J) rodrigues_jac(r, Rm, dR);
        for (int i = 0; i < n; i++) {
            V3 Y = Rm * obj[i] + tt;
            const double iz = 1.0 / Y.z, x = Y.x * iz, y = Y.y * iz;
            err[2 * i] = x - img[i][0];
            err[2 * i + 1] = y - img[i][1];
            if (J) {
                for (int k = 0; k < 3; k++) {
                    V3 d = dR[k] * obj[i];
                    (*J)(2 * i, k) = iz * d.x - x * iz * d.z;
                    (*J)(2 * i + 1, k) = iz * d.y - y * iz * d.z;
                }
                (*J)(2 * i, 3) = iz; (*J)(2 * i, 4) = 0; (*J)(2 * i, 5) = -x * iz;
                (*J)(2 * i + 1, 3) = 0; (*J)(2 * i + 1, 4) = iz; (*J)(2 * i + 1, 5) = -y * iz;
            }
        }
    };
    auto nrm = [](const std::vector<double> &v) { double s = 0; for (double x : v) s += x * x; return std::sqrt(s); };
    // CvLevMarq state machine (CALC_J -> CHECK_ERR ...), lambda = 10^lambdaLg10, diagonal scaled by (1 + lambda)
    const int max_iter = 20;
    const double eps = 1.1920928955078125e-07;  // FLT_EPSILON
    int lambdaLg10 = -3, iters = 0;
    std::vector<double> err(2 * n);
    Mat J(2 * n, 6), JtJ(6, 6);
    std::vector<double> JtErr(6);
    double prevErrNorm = 0;
    auto step = [&]() {
        const double lambda = std::exp(lambdaLg10 * 2.302585092994046);
        Mat N = JtJ;
        for (int i = 0; i < 6; i++) N(i, i) *= 1.0 + lambda;
        std::vector<double> d = sym_solve_svd(N, JtErr);
        for (int i = 0; i < 6; i++) param[i] = prev[i] - d[i];
    };
    for (;;) {
        // CALC_J
        project(param, err, &J);
        for (int a = 0; a < 6; a++) {
            for (int b = 0; b < 6; b++) { double s = 0; for (int k = 0; k < 2 * n; k++) s += J(k, a) * J(k, b); JtJ(a, b) = s; }
            double s = 0; for (int k = 0; k < 2 * n; k++) s += J(k, a) * err[k]; JtErr[a] = s;
        }
        for (int i = 0; i < 6; i++) prev[i] = param[i];
        step();
        if (iters == 0) prevErrNorm = nrm(err);
        // CHECK_ERR (repeats with a larger lambda while the error grows)
        bool done = false;
        for (;;) {
            project(param, err, nullptr);
            const double errNorm = nrm(err);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) { step(); continue; }
            }
            lambdaLg10 = std::max(lambdaLg10 - 1, -16);
            double dn = 0, pn = 0;
            for (int i = 0; i < 6; i++) { dn += (param[i] - prev[i]) * (param[i] - prev[i]); pn += prev[i] * prev[i]; }
            // cvNorm(param, prevParam, CV_RELATIVE_L2) = |param - prev| / |prev|
            if (++iters >= max_iter || std::sqrt(dn) / std::sqrt(pn) < eps) done = true;
            prevErrNorm = errNorm;
            break;
        }
        if (done) break;
    }
    for (int i = 0; i < 6; i++) if (!std::isfinite(param[i])) return false;
    R = rodrigues_to_R(V3(param[0], param[1], param[2]));
    t = V3(param[3], param[4], param[5]);
    return true;
}

// -------------------------------------------------------------------------------------------------------------------- EPnP
namespace {

struct Epnp {
    int n;
    std::vector<V3> pws, pcs;
    std::vector<std::array<double, 2>> us;
    std::vector<std::array<double, 4>> alphas;
    V3 cws[4], ccs[4];

    void choose_control_points() {
        cws[0] = V3(0, 0, 0);
        for (int i = 0; i < n; i++) cws[0] = cws[0] + pws[i];
        cws[0] = cws[0] / (double)n;
        M3 C;
        for (int i = 0; i < n; i++) {
            V3 d = pws[i] - cws[0];
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) C(a, b) += d[a] * d[b];
        }
        M3 U, V;
        double dc[3];
        svd3(C, U, dc, V);  // symmetric: U == V up to sign; rows of UCt = eigenvectors, descending
        for (int i = 1; i < 4; i++) {
            const double k = std::sqrt(dc[i - 1] / n);
            cws[i] = cws[0] + k * U.col(i - 1);
        }
    }
    void compute_barycentric() {
        M3 CC;
        for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) CC(i, j - 1) = cws[j][i] - cws[0][i];
        M3 Ci = inv3(CC);
        alphas.resize(n);
        for (int i = 0; i < n; i++) {
            V3 d = pws[i] - cws[0];
            V3 a = Ci * d;
            alphas[i] = {1.0 - a.x - a.y - a.z, a.x, a.y, a.z};
        }
    }
    void compute_ccs(const double *betas, const Mat &ut) {
        for (int i = 0; i < 4; i++) ccs[i] = V3(0, 0, 0);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * ut(11 - i, 3 * j + k);
    }
    void compute_pcs() {
        pcs.resize(n);
        for (int i = 0; i < n; i++) {
            V3 p(0, 0, 0);
            for (int j = 0; j < 4; j++) p = p + alphas[i][j] * ccs[j];
            pcs[i] = p;
        }
    }
    double reprojection_error(const M3 &R, const V3 &t) const {
        double s = 0;
        for (int i = 0; i < n; i++) {
            V3 Y = R * pws[i] + t;
            const double ue = Y.x / Y.z, ve = Y.y / Y.z;
            s += std::sqrt((us[i][0] - ue) * (us[i][0] - ue) + (us[i][1] - ve) * (us[i][1] - ve));
        }
        return s / n;
    }
    void estimate_R_and_t(M3 &R, V3 &t) const {
        V3 pc0(0, 0, 0), pw0(0, 0, 0);
        for (int i = 0; i < n; i++) { pc0 = pc0 + pcs[i]; pw0 = pw0 + pws[i]; }
        pc0 = pc0 / (double)n; pw0 = pw0 / (double)n;
        M3 ABt;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) ABt(j, k) += (pcs[i][j] - pc0[j]) * (pws[i][k] - pw0[k]);
        M3 U, V;
        double d[3];
        svd3(ABt, U, d, V);
        R = U * T(V);
        if (det3(R) < 0) for (int j = 0; j < 3; j++) R(2, j) = -R(2, j);
        t = pc0 - R * pw0;
    }
    double compute_R_and_t(const Mat &ut, const double *betas, M3 &R, V3 &t) {
        compute_ccs(betas, ut);
        compute_pcs();
        if (pcs[0].z < 0.0) {  // solve_for_sign
            for (int i = 0; i < 4; i++) ccs[i] = -ccs[i];
            for (int i = 0; i < n; i++) pcs[i] = -pcs[i];
        }
        estimate_R_and_t(R, t);
        return reprojection_error(R, t);
    }
    static void gauss_newton(const Mat &L, const double *rho, double *b) {
        for (int it = 0; it < 5; it++) {
            Mat A(6, 4);
            std::vector<double> rhs(6);
            for (int i = 0; i < 6; i++) {
                const double *l = &L.d[(size_t)i * 10];
                A(i, 0) = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
                A(i, 1) = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
                A(i, 2) = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
                A(i, 3) = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
                rhs[i] = rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2] +
                                   l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
            }
            std::vector<double> x = lstsq(A, rhs);  // qr_solve in the published code
            for (int k = 0; k < 4; k++) b[k] += x[k];
        }
    }
    bool compute_pose(M3 &Rout, V3 &tout) {
        choose_control_points();
        compute_barycentric();
        Mat M(2 * n, 12);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 4; j++) {
                M(2 * i, 3 * j) = alphas[i][j]; M(2 * i, 3 * j + 2) = alphas[i][j] * (0.0 - us[i][0]);
                M(2 * i + 1, 3 * j + 1) = alphas[i][j]; M(2 * i + 1, 3 * j + 2) = alphas[i][j] * (0.0 - us[i][1]);
            }
        Mat MtM(12, 12), V;
        for (int a = 0; a < 12; a++) for (int b = 0; b < 12; b++) { double s = 0; for (int k = 0; k < 2 * n; k++) s += M(k, a) * M(k, b); MtM(a, b) = s; }
        std::vector<double> w;
        sym_eig(MtM, w, V);  // ascending; cvSVD gives descending with Ut rows = vectors: Ut row 11 = smallest
        Mat ut(12, 12);
        for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) ut(r, c) = V(c, 11 - r);
        // L_6x10 and rho
        Mat L(6, 10);
        double rho[6];
        {
            double dv[4][6][3];
            for (int i = 0; i < 4; i++) {
                int a = 0, b = 1;
                for (int j = 0; j < 6; j++) {
                    for (int k = 0; k < 3; k++) dv[i][j][k] = ut(11 - i, 3 * a + k) - ut(11 - i, 3 * b + k);
                    b++;
                    if (b > 3) { a++; b = a + 1; }
                }
            }
            auto d3 = [](const double *x, const double *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
            for (int i = 0; i < 6; i++) {
                L(i, 0) = d3(dv[0][i], dv[0][i]); L(i, 1) = 2 * d3(dv[0][i], dv[1][i]); L(i, 2) = d3(dv[1][i], dv[1][i]);
                L(i, 3) = 2 * d3(dv[0][i], dv[2][i]); L(i, 4) = 2 * d3(dv[1][i], dv[2][i]); L(i, 5) = d3(dv[2][i], dv[2][i]);
                L(i, 6) = 2 * d3(dv[0][i], dv[3][i]); L(i, 7) = 2 * d3(dv[1][i], dv[3][i]); L(i, 8) = 2 * d3(dv[2][i], dv[3][i]);
                L(i, 9) = d3(dv[3][i], dv[3][i]);
            }
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                V3 d = cws[a] - cws[b];
                rho[j] = dot(d, d);
                b++;
                if (b > 3) { a++; b = a + 1; }
            }
        }
        std::vector<double> rho_v(rho, rho + 6);
        double betas[3][4], errs[3];
        M3 Rs_[3];
        V3 ts_[3];
        {   // approximation 1: betas10 columns B11 B12 B13 B14
            Mat L4(6, 4);
            for (int i = 0; i < 6; i++) { L4(i, 0) = L(i, 0); L4(i, 1) = L(i, 1); L4(i, 2) = L(i, 3); L4(i, 3) = L(i, 6); }
            std::vector<double> b4 = lstsq(L4, rho_v);
            double *b = betas[0];
            if (b4[0] < 0) { b[0] = std::sqrt(-b4[0]); b[1] = -b4[1] / b[0]; b[2] = -b4[2] / b[0]; b[3] = -b4[3] / b[0]; }
            else { b[0] = std::sqrt(b4[0]); b[1] = b4[1] / b[0]; b[2] = b4[2] / b[0]; b[3] = b4[3] / b[0]; }
        }
        {   // approximation 2: B11 B12 B22
            Mat L3(6, 3);
            for (int i = 0; i < 6; i++) { L3(i, 0) = L(i, 0); L3(i, 1) = L(i, 1); L3(i, 2) = L(i, 2); }
            std::vector<double> b3 = lstsq(L3, rho_v);
            double *b = betas[1];
            if (b3[0] < 0) { b[0] = std::sqrt(-b3[0]); b[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0; }
            else { b[0] = std::sqrt(b3[0]); b[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0; }
            if (b3[1] < 0) b[0] = -b[0];
            b[2] = 0; b[3] = 0;
        }
        {   // approximation 3: B11 B12 B22 B13 B23
            Mat L5(6, 5);
            for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) L5(i, k) = L(i, k);
            std::vector<double> b5 = lstsq(L5, rho_v);
            double *b = betas[2];
            if (b5[0] < 0) { b[0] = std::sqrt(-b5[0]); b[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0; }
            else { b[0] = std::sqrt(b5[0]); b[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0; }
            if (b5[1] < 0) b[0] = -b[0];
            b[2] = b5[3] / b[0]; b[3] = 0;
        }
        for (int k = 0; k < 3; k++) {
            gauss_newton(L, rho, betas[k]);
            errs[k] = compute_R_and_t(ut, betas[k], Rs_[k], ts_[k]);
        }
        int N = 0;
        if (errs[1] < errs[0]) N = 1;
        if (errs[2] < errs[N]) N = 2;
        Rout = Rs_[N]; tout = ts_[N];
        for (int i = 0; i < 3; i++) if (!std::isfinite(tout[i])) return false;
        return std::isfinite(errs[N]);
    }
};

bool epnp(const std::vector<V3> &obj, const std::vector<std::array<double, 2>> &img, const std::vector<int> &idx, M3 &R, V3 &t) {
    Epnp e;
    e.n = (int)idx.size();
    e.pws.resize(e.n); e.us.resize(e.n);
    for (int i = 0; i < e.n; i++) { e.pws[i] = obj[idx[i]]; e.us[i] = img[idx[i]]; }
    return e.compute_pose(R, t);
}

// cv::RNG (multiply-with-carry) as used by RANSACPointSetRegistrator: RNG rng((uint64)-1)
struct CvRng {
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::max(p, 0.0); p = std::min(p, 1.0);
    ep = std::max(ep, 0.0); ep = std::min(ep, 1.0);
    double num = std::max(1.0 - p, 2.2250738585072014e-308);
    double denom = 1.0 - std::pow(1.0 - ep, modelPoints);
    if (denom < 2.2250738585072014e-308) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)std::lrint(num / denom);
}

}  // namespace

// cv::solvePnPRansac(obj, img, K = I, no distortion, rvec, tvec, false, 100, 1/460, 0.99, inliers, SOLVEPNP_EPNP) as called by
// MotionEstimator::solveRelativeRT_PNP (solve_5pts.cpp:248-294).  Returns camera_point = R X + t; inliers = mask over the input.
bool solve_pnp_ransac_epnp(const std::vector<V3> &obj_in, const std::vector<std::array<double, 2>> &img_in, int max_iters, double thresh,
                           double confidence, M3 &R, V3 &t, std::vector<uint8_t> &inliers) {
    const int count = (int)obj_in.size(), model_points = 5;
    inliers.assign(count, 0);
    if (count < model_points) return false;  // (count == 4 switches the reference to P3P; not reachable: relativePose needs > 20)
    std::vector<V3> obj(count);
    std::vector<std::array<double, 2>> img(count);
    for (int i = 0; i < count; i++) {
        obj[i] = V3(f32(obj_in[i].x), f32(obj_in[i].y), f32(obj_in[i].z));
        img[i] = {f32(img_in[i][0]), f32(img_in[i][1])};
    }
    CvRng rng((uint64_t)-1);
    const double th2 = thresh * thresh;
    int niters = max_iters, maxGood = 0;
    M3 bestR;
    V3 bestt;
    std::vector<uint8_t> mask(count), bestMask(count, 0);
    for (int iter = 0; iter < niters; iter++) {
        std::vector<int> idx(model_points);
        {   // getSubset: distinct random indices (checkSubset is the default "true" for the PnP callback)
            int i = 0, iters2 = 0;
            for (; iters2 < 1000; iters2++) {
                for (i = 0; i < model_points && iters2 < 1000;) {
                    int idx_i = rng.uniform(0, count), j;
                    for (j = 0; j < i; j++) if (idx_i == idx[j]) break;
                    if (j < i) continue;
                    idx[i] = idx_i;
                    i++;
                }
                if (i == model_points) break;
            }
            if (i < model_points) { if (iter == 0) return false; break; }
        }
        M3 Rm;
        V3 tm;
        if (!epnp(obj, img, idx, Rm, tm)) continue;
        int good = 0;
        for (int i = 0; i < count; i++) {
            V3 Y = Rm * obj[i] + tm;
            const float px = (float)(Y.x / Y.z), py = (float)(Y.y / Y.z);  // projectPoints writes Point2f
            const float dx = (float)img[i][0] - px, dy = (float)img[i][1] - py;
            const float e = dx * dx + dy * dy;
            mask[i] = e <= (float)th2;
            good += mask[i];
        }
        if (good > std::max(maxGood, model_points - 1)) {
            bestR = Rm; bestt = tm; bestMask = mask; maxGood = good;
            niters = ransac_update_num_iters(confidence, (double)(count - good) / count, model_points, niters);
        }
    }
    if (maxGood <= 0) return false;
    std::vector<int> in_idx;
    for (int i = 0; i < count; i++) if (bestMask[i]) in_idx.push_back(i);
    inliers = bestMask;
    // final model: the same solver over all inliers
    if (!epnp(obj, img, in_idx, R, t)) { R = bestR; t = bestt; }
    return true;
}

}  // namespace ovio

// =====================================================================================================================
// GlobalSFM::construct (initial_sfm.cpp:184-412): PnP chain + depth-checked triangulation + full bundle adjustment
namespace ovio {
namespace {

bool sfm_solve_frame_by_pnp(M3 &R_initial, V3 &P_initial, int i, const std::vector<SfmFeature> &sfm_f) {  // initial_sfm.cpp:22-71
    std::vector<V3> pts3;
    std::vector<std::array<double, 2>> pts2;
    for (const SfmFeature &f : sfm_f) {
        if (!f.state) continue;
        for (size_t k = 0; k < f.observation.size(); k++)
            if (f.observation[k].first == i) {
                pts2.push_back(f.observation[k].second);
                pts3.push_back(V3(f.position[0], f.position[1], f.position[2]));
                break;
            }
    }
    if ((int)pts2.size() < 15 && (int)pts2.size() < 10) return false;  // "unstable features tracking" below 15, failure below 10
    return solve_pnp_iterative(pts3, pts2, R_initial, P_initial);
}

// initial_sfm.cpp:113-171: the depth of the observation in frame0 places the point, accepted when it reprojects into frame1
// within 1/460 (normalised plane)
void sfm_triangulate_two_frames_with_depth(int frame0, const M3 &R0, const V3 &t0, int frame1, const M3 &R1, const V3 &t1,
                                           std::vector<SfmFeature> &sfm_f) {
    for (SfmFeature &f : sfm_f) {
        if (f.state) continue;
        bool has_0 = false, has_1 = false;
        V3 point0;
        std::array<double, 2> point1{0, 0};
        for (size_t k = 0; k < f.observation.size(); k++) {
            const double d = f.observation_depth[k].second;
            if (d < 0.1 || d > 10) continue;
            if (f.observation[k].first == frame0) { point0 = V3(f.observation[k].second[0] * d, f.observation[k].second[1] * d, d); has_0 = true; }
            if (f.observation[k].first == frame1) { point1 = f.observation[k].second; has_1 = true; }
        }
        if (!(has_0 && has_1)) continue;
        V3 p3 = T(R0) * point0 - T(R0) * t0;
        V3 rp = R1 * p3 + t1;
        const double rx = point1[0] - rp.x / rp.z, ry = point1[1] - rp.y / rp.z;
        if (std::sqrt(rx * rx + ry * ry) < 1.0 / 460) { f.state = true; f.position[0] = p3.x; f.position[1] = p3.y; f.position[2] = p3.z; }
    }
}

// ceres::QuaternionParameterization: x_plus = [cos|d|, sin|d| d/|d|] (x) x, q stored (w, x, y, z)
Q quat_plus(const Q &x, const V3 &d) {
    const double nd = norm(d);
    if (!(nd > 0.0)) return x;
    const double s = std::sin(nd) / nd;
    return Q(std::cos(nd), s * d.x, s * d.y, s * d.z) * x;
}

struct SfmBa {
    int frame_num, l;
    std::vector<Q> q;       // camera rotations (world -> camera), one per frame
    std::vector<V3> t;
    std::vector<SfmFeature> *feat;
    std::vector<int> pidx;  // indices of features with state == true
    std::vector<int> rot_off, trans_off;  // tangent offsets of the free camera blocks (-1 = constant)
    int nc = 0;

    struct Obs { int frame, p; double u, v; };
    std::vector<Obs> obs;

    void setup() {
        rot_off.assign(frame_num, -1); trans_off.assign(frame_num, -1);
        nc = 0;
        for (int i = 0; i < frame_num; i++) {
            if (i != l) { rot_off[i] = nc; nc += 3; }
            if (i != l && i != frame_num - 1) { trans_off[i] = nc; nc += 3; }
        }
        for (int i = 0; i < (int)feat->size(); i++) {
            if (!(*feat)[i].state) continue;
            const int p = (int)pidx.size();
            pidx.push_back(i);
            for (auto &o : (*feat)[i].observation) obs.push_back({o.first, p, o.second[0], o.second[1]});
        }
    }
    // residuals (2 per observation) and, optionally, Jacobian blocks in tangent coordinates
    double evaluate(const std::vector<Q> &qq, const std::vector<V3> &tt, const std::vector<V3> &pp, std::vector<double> &r,
                    std::vector<std::array<double, 24>> *J) const {
        double cost = 0;
        r.resize(2 * obs.size());
        if (J) J->resize(obs.size());
        for (size_t k = 0; k < obs.size(); k++) {
            const Obs &o = obs[k];
            const Q &qc = qq[o.frame];
            const V3 &X = pp[o.p];
            const double nq = std::sqrt(qc.w * qc.w + qc.x * qc.x + qc.y * qc.y + qc.z * qc.z);  // QuaternionRotatePoint normalises
            const Q u(qc.w / nq, qc.x / nq, qc.y / nq, qc.z / nq);
            const V3 pc = rot(u, X) + tt[o.frame];
            const double iz = 1.0 / pc.z, xp = pc.x * iz, yp = pc.y * iz;
            r[2 * k] = xp - o.u; r[2 * k + 1] = yp - o.v;
            cost += 0.5 * (r[2 * k] * r[2 * k] + r[2 * k + 1] * r[2 * k + 1]);
            if (!J) continue;
            const double drdp[2][3] = {{iz, 0, -xp * iz}, {0, iz, -yp * iz}};
            // d(R(u) X)/d(a, v): a = u.w, v = u.vec
            const V3 v = u.vec();
            const V3 vxX = cross(v, X);
            M3 dpdv = (-2.0 * u.w) * skew(X);
            const double vX = dot(v, X);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) dpdv(i, j) += 2.0 * ((i == j ? vX : 0.0) + v[i] * X[j] - 2.0 * X[i] * v[j]);
            double dpdq[3][4];
            for (int i = 0; i < 3; i++) { dpdq[i][0] = 2.0 * vxX[i]; for (int j = 0; j < 3; j++) dpdq[i][1 + j] = dpdv(i, j); }
            // QuaternionParameterization::ComputeJacobian (4 x 3)
            const double Jl[4][3] = {{-u.x, -u.y, -u.z}, {u.w, u.z, -u.y}, {-u.z, u.w, u.x}, {u.y, -u.x, u.w}};
            double dpdth[3][3];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int m = 0; m < 4; m++) s += dpdq[i][m] * Jl[m][j]; dpdth[i][j] = s; }
            const M3 Ru = toR(u);
            std::array<double, 24> &out = (*J)[k];  // [2][3 rot | 3 trans | 3 point], row-major 2 x 9 + padding
            for (int a = 0; a < 2; a++)
                for (int j = 0; j < 3; j++) {
                    double sr = 0, sp = 0;
                    for (int i = 0; i < 3; i++) { sr += drdp[a][i] * dpdth[i][j]; sp += drdp[a][i] * Ru(i, j); }
                    out[a * 9 + j] = sr; out[a * 9 + 3 + j] = drdp[a][j]; out[a * 9 + 6 + j] = sp;
                }
        }
        return cost;
    }
};

}  // namespace

// returns false when a PnP fails or the bundle adjustment neither converges nor reaches final_cost < 5e-3
bool sfm_construct(int frame_num, Q *q, V3 *Tw, int l, const M3 &relative_R, const V3 &relative_T, std::vector<SfmFeature> &sfm_f,
                   std::map<int, V3> &sfm_tracked_points, SfmStats *stats) {
    q[l] = Q(1, 0, 0, 0);
    Tw[l] = V3(0, 0, 0);
    q[frame_num - 1] = q[l] * fromR(relative_R);
    Tw[frame_num - 1] = relative_T;
    std::vector<M3> cR(frame_num);
    std::vector<V3> ct(frame_num);
    std::vector<Q> cQ(frame_num);
    auto set_from_world = [&](int i) {
        cQ[i] = inverse(q[i]);
        cR[i] = toR(cQ[i]);
        ct[i] = -1.0 * (cR[i] * Tw[i]);
    };
    set_from_world(l);
    set_from_world(frame_num - 1);
    for (int i = l; i < frame_num - 1; i++) {
        if (i > l) {
            M3 R0 = cR[i - 1];
            V3 P0 = ct[i - 1];
            if (!sfm_solve_frame_by_pnp(R0, P0, i, sfm_f)) return false;
            cR[i] = R0; ct[i] = P0; cQ[i] = fromR(R0);
        }
        sfm_triangulate_two_frames_with_depth(i, cR[i], ct[i], frame_num - 1, cR[frame_num - 1], ct[frame_num - 1], sfm_f);
    }
    for (int i = l + 1; i < frame_num - 1; i++) sfm_triangulate_two_frames_with_depth(l, cR[l], ct[l], i, cR[i], ct[i], sfm_f);
    for (int i = l - 1; i >= 0; i--) {
        M3 R0 = cR[i + 1];
        V3 P0 = ct[i + 1];
        if (!sfm_solve_frame_by_pnp(R0, P0, i, sfm_f)) return false;
        cR[i] = R0; ct[i] = P0; cQ[i] = fromR(R0);
        sfm_triangulate_two_frames_with_depth(i, cR[i], ct[i], l, cR[l], ct[l], sfm_f);
    }
    for (SfmFeature &f : sfm_f) {  // all other points: first observation's depth, checked against the last observation
        if (f.state || f.observation.size() < 2) continue;
        const double d = f.observation_depth[0].second;
        if (d < 0.1 || d > 10) continue;
        const int f0 = f.observation[0].first, f1 = f.observation.back().first;
        V3 point0(f.observation[0].second[0] * d, f.observation[0].second[1] * d, d);
        V3 p3 = T(cR[f0]) * point0 - T(cR[f0]) * ct[f0];
        V3 rp = cR[f1] * p3 + ct[f1];
        const double rx = f.observation.back().second[0] - rp.x / rp.z, ry = f.observation.back().second[1] - rp.y / rp.z;
        if (std::sqrt(rx * rx + ry * ry) < 1.0 / 460) { f.state = true; f.position[0] = p3.x; f.position[1] = p3.y; f.position[2] = p3.z; }
    }

    // ---- full BA: ceres defaults (trust region, Levenberg-Marquardt, Jacobi scaling, 50 iterations), DENSE_SCHUR
    SfmBa ba;
    ba.frame_num = frame_num; ba.l = l; ba.q = cQ; ba.t = ct; ba.feat = &sfm_f;
    ba.setup();
    const int np = (int)ba.pidx.size(), nc = ba.nc, ntot = nc + 3 * np;
    std::vector<V3> pts(np);
    for (int p = 0; p < np; p++) { const double *x = sfm_f[ba.pidx[p]].position; pts[p] = V3(x[0], x[1], x[2]); }
    std::vector<double> r, rc;
    std::vector<std::array<double, 24>> J;
    double cost = ba.evaluate(ba.q, ba.t, pts, r, &J);
    const double initial_cost = cost;
    std::vector<double> scale(ntot, 1.0);
    auto col_of = [&](const SfmBa::Obs &o, int j) -> int {  // tangent column of local column j (0-2 rot, 3-5 trans, 6-8 point)
        if (j < 3) return ba.rot_off[o.frame] < 0 ? -1 : ba.rot_off[o.frame] + j;
        if (j < 6) return ba.trans_off[o.frame] < 0 ? -1 : ba.trans_off[o.frame] + (j - 3);
        return nc + 3 * o.p + (j - 6);
    };
    {
        std::vector<double> cn(ntot, 0.0);
        for (size_t k = 0; k < ba.obs.size(); k++)
            for (int a = 0; a < 2; a++) for (int j = 0; j < 9; j++) { int c = col_of(ba.obs[k], j); if (c >= 0) cn[c] += J[k][a * 9 + j] * J[k][a * 9 + j]; }
        for (int c = 0; c < ntot; c++) scale[c] = 1.0 / (1.0 + std::sqrt(cn[c]));
    }
    auto gradient_max = [&]() {
        std::vector<double> g(ntot, 0.0);
        for (size_t k = 0; k < ba.obs.size(); k++)
            for (int a = 0; a < 2; a++) for (int j = 0; j < 9; j++) { int c = col_of(ba.obs[k], j); if (c >= 0) g[c] += J[k][a * 9 + j] * r[2 * k + a]; }
        double m = 0;
        for (double v : g) m = std::max(m, std::fabs(v));
        return m;
    };
    bool converged = false;
    int iterations = 0, invalid = 0;
    double radius = 1e4, decrease_factor = 2.0;
    if (ntot == 0 || ba.obs.empty()) converged = true;
    else if (gradient_max() <= 1e-10) converged = true;
    while (!converged && iterations < 50) {
        iterations++;
        // normal equations in the scaled space, LM diagonal sqrt(clamp(diag) / radius)
        Mat Hcc(nc, nc);
        std::vector<double> gc(nc, 0.0), gp(3 * np, 0.0), Hpp(9 * (size_t)np, 0.0);
        Mat Hcp(nc, 3 * np);
        for (size_t k = 0; k < ba.obs.size(); k++) {
            const SfmBa::Obs &o = ba.obs[k];
            int cols[9];
            double Js[2][9];
            for (int j = 0; j < 9; j++) { cols[j] = col_of(o, j); for (int a = 0; a < 2; a++) Js[a][j] = cols[j] >= 0 ? J[k][a * 9 + j] * scale[cols[j]] : 0.0; }
            for (int i = 0; i < 9; i++) {
                if (cols[i] < 0) continue;
                const double gi = Js[0][i] * r[2 * k] + Js[1][i] * r[2 * k + 1];
                if (i < 6) gc[cols[i]] += gi; else gp[cols[i] - nc] += gi;
                for (int j = 0; j < 9; j++) {
                    if (cols[j] < 0) continue;
                    const double h = Js[0][i] * Js[0][j] + Js[1][i] * Js[1][j];
                    if (i < 6 && j < 6) Hcc(cols[i], cols[j]) += h;
                    else if (i < 6 && j >= 6) Hcp(cols[i], cols[j] - nc) += h;
                    else if (i >= 6 && j >= 6) Hpp[9 * (size_t)o.p + 3 * (i - 6) + (j - 6)] += h;
                }
            }
        }
        auto lm = [&](double d) { return std::min(std::max(d, 1e-6), 1e32) / radius; };
        for (int c = 0; c < nc; c++) Hcc(c, c) += lm(Hcc(c, c));
        for (int p = 0; p < np; p++) for (int i = 0; i < 3; i++) Hpp[9 * (size_t)p + 4 * i] += lm(Hpp[9 * (size_t)p + 4 * i]);
        // Schur complement on the points
        std::vector<M3> Hpp_inv(np);
        bool lin_ok = true;
        for (int p = 0; p < np; p++) {
            M3 A;
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A(i, j) = Hpp[9 * (size_t)p + 3 * i + j];
            if (!(std::fabs(det3(A)) > 0)) { lin_ok = false; break; }
            Hpp_inv[p] = inv3(A);
        }
        std::vector<double> dxs(ntot, 0.0);
        if (lin_ok) {
            Mat S = Hcc;
            std::vector<double> rhs(nc);
            for (int c = 0; c < nc; c++) rhs[c] = -gc[c];
            for (int p = 0; p < np; p++) {
                // W = Hcp[:, 3p..3p+2] (nc x 3), S -= W Hpp^-1 W^T, rhs += W Hpp^-1 gp
                for (int a = 0; a < nc; a++) {
                    double wa[3] = {Hcp(a, 3 * p), Hcp(a, 3 * p + 1), Hcp(a, 3 * p + 2)};
                    if (wa[0] == 0 && wa[1] == 0 && wa[2] == 0) continue;
                    double wi[3];
                    for (int j = 0; j < 3; j++) wi[j] = wa[0] * Hpp_inv[p](0, j) + wa[1] * Hpp_inv[p](1, j) + wa[2] * Hpp_inv[p](2, j);
                    rhs[a] += wi[0] * gp[3 * p] + wi[1] * gp[3 * p + 1] + wi[2] * gp[3 * p + 2];
                    for (int b = 0; b < nc; b++) S(a, b) -= wi[0] * Hcp(b, 3 * p) + wi[1] * Hcp(b, 3 * p + 1) + wi[2] * Hcp(b, 3 * p + 2);
                }
            }
            if (nc > 0) {
                if (chol(S)) chol_solve(S, rhs); else lin_ok = false;
            }
            if (lin_ok) {
                for (int c = 0; c < nc; c++) dxs[c] = rhs[c];
                for (int p = 0; p < np; p++) {
                    double b3[3] = {-gp[3 * p], -gp[3 * p + 1], -gp[3 * p + 2]};
                    for (int a = 0; a < nc; a++) for (int j = 0; j < 3; j++) b3[j] -= Hcp(a, 3 * p + j) * dxs[a];
                    for (int i = 0; i < 3; i++) dxs[nc + 3 * p + i] = Hpp_inv[p](i, 0) * b3[0] + Hpp_inv[p](i, 1) * b3[1] + Hpp_inv[p](i, 2) * b3[2];
                }
            }
        }
        double model_cost_change = 0;
        if (lin_ok) {
            for (size_t k = 0; k < ba.obs.size(); k++)
                for (int a = 0; a < 2; a++) {
                    double mr = 0;
                    for (int j = 0; j < 9; j++) { int c = col_of(ba.obs[k], j); if (c >= 0) mr += J[k][a * 9 + j] * scale[c] * dxs[c]; }
                    model_cost_change -= mr * (mr / 2 + r[2 * k + a]);
                }
        }
        if (!lin_ok || !(model_cost_change > 0)) {
            if (++invalid >= 5) break;
            radius /= decrease_factor; decrease_factor *= 2;  // StepRejected
            continue;
        }
        invalid = 0;
        // candidate
        std::vector<Q> qn = ba.q;
        std::vector<V3> tn = ba.t, pn = pts;
        double step2 = 0, x2 = 0;
        for (int i = 0; i < frame_num; i++) {
            if (ba.rot_off[i] >= 0) {
                V3 d(dxs[ba.rot_off[i]] * scale[ba.rot_off[i]], dxs[ba.rot_off[i] + 1] * scale[ba.rot_off[i] + 1], dxs[ba.rot_off[i] + 2] * scale[ba.rot_off[i] + 2]);
                qn[i] = quat_plus(ba.q[i], d);
                step2 += dot(d, d);
                x2 += ba.q[i].w * ba.q[i].w + ba.q[i].x * ba.q[i].x + ba.q[i].y * ba.q[i].y + ba.q[i].z * ba.q[i].z;
            }
            if (ba.trans_off[i] >= 0) {
                V3 d(dxs[ba.trans_off[i]] * scale[ba.trans_off[i]], dxs[ba.trans_off[i] + 1] * scale[ba.trans_off[i] + 1], dxs[ba.trans_off[i] + 2] * scale[ba.trans_off[i] + 2]);
                tn[i] = ba.t[i] + d;
                step2 += dot(d, d);
                x2 += dot(ba.t[i], ba.t[i]);
            }
        }
        for (int p = 0; p < np; p++) {
            V3 d(dxs[nc + 3 * p] * scale[nc + 3 * p], dxs[nc + 3 * p + 1] * scale[nc + 3 * p + 1], dxs[nc + 3 * p + 2] * scale[nc + 3 * p + 2]);
            pn[p] = pts[p] + d;
            step2 += dot(d, d);
            x2 += dot(pts[p], pts[p]);
        }
        const double ccost = ba.evaluate(qn, tn, pn, rc, nullptr);
        if (std::sqrt(step2) <= 1e-8 * (std::sqrt(x2) + 1e-8)) { converged = true; break; }      // parameter tolerance
        if (std::fabs(cost - ccost) <= 1e-6 * cost) { converged = true; break; }                    // function tolerance
        const double rel = (cost - ccost) / model_cost_change;
        if (rel > 1e-3) {
            ba.q = qn; ba.t = tn; pts = pn;
            cost = ba.evaluate(ba.q, ba.t, pts, r, &J);
            if (gradient_max() <= 1e-10) { converged = true; break; }
            radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
            decrease_factor = 2.0;
        } else {
            radius /= decrease_factor; decrease_factor *= 2;
            if (radius < 1e-32) { converged = true; break; }
        }
    }
    if (stats) { stats->iterations = iterations; stats->initial_cost = initial_cost; stats->final_cost = cost; stats->converged = converged; stats->points = np; }
    if (!(converged || cost < 5e-3)) return false;
    for (int p = 0; p < np; p++) { double *x = sfm_f[ba.pidx[p]].position; x[0] = pts[p].x; x[1] = pts[p].y; x[2] = pts[p].z; }
    for (int i = 0; i < frame_num; i++) {
        q[i] = inverse(ba.q[i]);
        Tw[i] = -1.0 * rot(q[i], ba.t[i]);
    }
    for (const SfmFeature &f : sfm_f) if (f.state) sfm_tracked_points[f.id] = V3(f.position[0], f.position[1], f.position[2]);
    return true;
}

}  // namespace ovio

// =====================================================================================================================
// Estimator::relativePose (estimator.cpp:884-920) + MotionEstimator::solveRelativeRT_PNP (solve_5pts.cpp:248-294) on the SfM
// feature list (observations of a feature are in consecutive frames starting at observation[0].first)
namespace ovio {

bool sfm_relative_pose(int window_size, const std::vector<SfmFeature> &sfm_f, M3 &relative_R, V3 &relative_T, int &l) {
    for (int i = 0; i < window_size; i++) {
        // FeatureManager::getCorrespondingWithDepth(i, WINDOW_SIZE)  feature_manager.cpp:168-195
        std::vector<V3> a3, b3;
        for (const SfmFeature &f : sfm_f) {
            if (f.observation.empty()) continue;
            const int start = f.observation[0].first, end = start + (int)f.observation.size() - 1;
            if (!(start <= i && end >= window_size)) continue;
            const int il = i - start, ir = window_size - start;
            const double da = f.observation_depth[il].second, db = f.observation_depth[ir].second;
            if (da < 0.1 || da > 10) continue;
            if (db < 0.1 || db > 10) continue;
            a3.push_back(V3(f.observation[il].second[0] * da, f.observation[il].second[1] * da, da));
            b3.push_back(V3(f.observation[ir].second[0] * db, f.observation[ir].second[1] * db, db));
        }
        if (a3.size() <= 20) continue;
        double sum_parallax = 0;
        for (size_t k = 0; k < a3.size(); k++) {
            const double dx = a3[k].x / a3[k].z - b3[k].x / b3[k].z, dy = a3[k].y / a3[k].z - b3[k].y / b3[k].z;
            sum_parallax += std::sqrt(dx * dx + dy * dy);
        }
        const double average_parallax = sum_parallax / (int)a3.size();
        if (!(average_parallax * 460 > 30)) continue;
        // solveRelativeRT_PNP: 3-D points of frame i against the normalised points of the newest frame
        std::vector<V3> lll;
        std::vector<std::array<double, 2>> rr;
        for (size_t k = 0; k < a3.size(); k++)
            if (a3[k].z > 0 && b3[k].z > 0) { lll.push_back(a3[k]); rr.push_back({b3[k].x / b3[k].z, b3[k].y / b3[k].z}); }
        M3 R;
        V3 t;
        std::vector<uint8_t> inl;
        solve_pnp_ransac_epnp(lll, rr, 100, 1.0 / 460, 0.99, R, t, inl);  // the reference ignores the return value and returns true
        // Sophus::SO3(rvec).matrix() of the returned rotation vector = the rotation itself
        relative_R = T(R);
        relative_T = -1.0 * (T(R) * t);
        l = i;
        return true;
    }
    return false;
}

}  // namespace ovio

// =====================================================================================================================
// Estimator::initialStructure (estimator.cpp:384-579) and Estimator::visualInitialAlignWithDepth (:799-869)
namespace ovio {

bool Estimator::initialStructure() {
    // IMU excitation check (:386-420); sum_g is not initialised upstream: zero here.  Only gates the Bas estimate.
    bool is_imu_excited = false;
    {
        V3 sum_g;
        const int nf = (int)all_image_frame.size() - 1;
        for (auto it = std::next(all_image_frame.begin()); it != all_image_frame.end(); ++it)
            sum_g = sum_g + it->second.pre_integration->delta_v / it->second.pre_integration->sum_dt;
        V3 aver_g = sum_g * (1.0 / nf);
        double var = 0;
        for (auto it = std::next(all_image_frame.begin()); it != all_image_frame.end(); ++it) {
            V3 d = it->second.pre_integration->delta_v / it->second.pre_integration->sum_dt - aver_g;
            var += dot(d, d);
        }
        var = std::sqrt(var / nf);
        if (!(var < 0.25)) is_imu_excited = true;
    }
    // global SfM (:422-463)
    std::vector<SfmFeature> sfm_f;
    for (const Landmark &lm : feature) {
        SfmFeature f;
        f.id = lm.feature_id;
        int imu_j = lm.start_frame - 1;
        for (const Obs &o : lm.obs) {
            imu_j++;
            f.observation.push_back({imu_j, {o.x, o.y}});
            f.observation_depth.push_back({imu_j, o.depth});
        }
        sfm_f.push_back(f);
    }
    M3 relative_R;
    V3 relative_T;
    int l = 0;
    if (!sfm_relative_pose(W, sfm_f, relative_R, relative_T, l)) return false;
    std::vector<Q> Qs(frame_count + 1);
    std::vector<V3> Ts(frame_count + 1);
    std::map<int, V3> sfm_tracked_points;
    if (!sfm_construct(frame_count + 1, Qs.data(), Ts.data(), l, relative_R, relative_T, sfm_f, sfm_tracked_points)) {
        marginalization_flag = 0;  // MARGIN_OLD
        return false;
    }
    // PnP for every image frame (:466-548)
    {
        int i = 0;
        for (auto it = all_image_frame.begin(); it != all_image_frame.end(); ++it) {
            if (it->first == Headers[i]) {
                it->second.is_key_frame = true;
                it->second.R = toR(Qs[i]) * T(ric);
                it->second.T = Ts[i];
                i++;
                continue;
            }
            if (it->first > Headers[i]) i++;
            M3 R_initial = toR(inverse(Qs[i]));
            V3 P_initial = -1.0 * (R_initial * Ts[i]);
            it->second.is_key_frame = false;
            std::vector<V3> pts3;
            std::vector<std::array<double, 2>> pts2;
            for (auto &kv : it->second.points) {
                auto f = sfm_tracked_points.find(kv.first);
                if (f == sfm_tracked_points.end()) continue;
                pts3.push_back(f->second);
                pts2.push_back(kv.second);
            }
            if (pts3.size() < 6) return false;
            if (!solve_pnp_iterative(pts3, pts2, R_initial, P_initial)) return false;
            M3 R_pnp = T(R_initial);
            V3 T_pnp = R_pnp * (-1.0 * P_initial);
            it->second.R = R_pnp * T(ric);
            it->second.T = T_pnp;
        }
    }
    if (!visualInitialAlignWithDepth()) return false;
    if (!is_imu_excited) {  // :552-570 accelerometer bias from the mean specific force
        V3 sum_a;
        for (auto it = std::next(all_image_frame.begin()); it != all_image_frame.end(); ++it)
            sum_a = sum_a + it->second.pre_integration->delta_v / it->second.pre_integration->sum_dt;
        V3 avg_a = sum_a * (1.0 / ((int)all_image_frame.size() - 1));
        V3 tmp_Bas = avg_a - T(g2R(avg_a)) * V3(0, 0, cfg.g_norm);  // g2R(avg_a).inverse() * G
        for (int i = 0; i <= W; i++) Bas[i] = tmp_Bas;
    }
    return true;
}

bool Estimator::visualInitialAlignWithDepth() {
    // solveGyroscopeBias over all image frames (initial_aligment.cpp:3-36)
    {
        double A[3][3] = {{0}}, b[3] = {0};
        for (auto fi = all_image_frame.begin(); std::next(fi) != all_image_frame.end(); ++fi) {
            auto fj = std::next(fi);
            Q q_ij = fromR(T(fi->second.R) * fj->second.R);
            M3 tA;
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) tA(r, c) = fj->second.pre_integration->jacobian[3 + r][12 + c];  // O_R, O_BG
            V3 tb = 2.0 * (inverse(fj->second.pre_integration->delta_q) * q_ij).vec();
            M3 AtA = T(tA) * tA;
            V3 Atb = T(tA) * tb;
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) A[r][c] += AtA(r, c); b[r] += Atb[r]; }
        }
        std::vector<double> Av(9), bv(3);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Av[3 * r + c] = A[r][c]; bv[r] = b[r]; }
        std::vector<double> x = ldlt_solve(Av, bv, 3);
        V3 dbg(x[0], x[1], x[2]);
        for (int i = 0; i <= W; i++) Bgs[i] = Bgs[i] + dbg;
        for (auto fi = all_image_frame.begin(); std::next(fi) != all_image_frame.end(); ++fi)
            std::next(fi)->second.pre_integration->repropagate(V3(), Bgs[0]);
    }
    std::vector<AlignFrame> fr;
    for (auto &it : all_image_frame) {
        AlignFrame a;
        a.R = it.second.R; a.T = it.second.T;
        if (it.second.pre_integration) { a.sum_dt = it.second.pre_integration->sum_dt; a.delta_p = it.second.pre_integration->delta_p; a.delta_v = it.second.pre_integration->delta_v; }
        fr.push_back(a);
    }
    std::vector<double> x;
    V3 gg;
    if (!linear_alignment_with_depth(fr, tic, cfg.g_norm, gg, x)) return false;
    g = gg;
    for (int i = 0; i <= W; i++) pre_integrations[i]->repropagate(V3(), Bgs[i]);
    for (int i = 0; i <= frame_count; i++) {
        ImageFrameO &f = all_image_frame[Headers[i]];
        Ps[i] = f.T; Rs[i] = f.R;
        f.is_key_frame = true;
    }
    for (int i = 0; i <= W; i++) pre_integrations[i]->repropagate(V3(), Bgs[i]);  // repeated upstream (:833-836)
    // positions / velocities / gravity alignment (:839-869); the key-frame counter indexes x (quirk, see align_window_to_gravity)
    align_window_to_gravity(frame_count + 1, Ps, Rs, Vs, x, tic, g);
    return true;
}

}  // namespace ovio
