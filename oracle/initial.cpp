// ORACLE (test infrastructure only — never linked into or called by the product path).
// CPU restatement of the visual-inertial alignment step of the dynamic (static_init: 0) initialisation, SURVEY.md §8f rank 1:
//   TangentBasis                   vins_estimator/src/initial/initial_aligment.cpp:78-91
//   RefineGravityWithDepth         initial_aligment.cpp:170-244
//   LinearAlignmentWithDepth       initial_aligment.cpp:337-405
//   the state hand-over at the end of Estimator::visualInitialAlignWithDepth   estimator/estimator.cpp:839-869
// The structure-from-motion front of that initialisation (relativePose / solveRelativeRT_PNP / GlobalSFM::construct / per-frame
// solvePnP, estimator.cpp:384-579) needs restatements of cv::solvePnP, cv::solvePnPRansac (EPnP) and a Ceres bundle adjustment and
// is NOT built yet; these routines take the frame poses it would deliver as input.
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
