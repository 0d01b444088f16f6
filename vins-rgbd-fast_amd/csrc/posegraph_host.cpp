// Host half of the loop-closure slice (include/vio_posegraph.h): the geometric verification of a loop candidate and the 4-DoF pose-graph
// optimisation -- per-keyframe work in the reference too (KeyFrame::findConnection / PnPRANSAC, pose_graph/src/keyframe/keyframe.cpp:195-528;
// PoseGraph::optimize4DoF, pose_graph/src/pose_graph/pose_graph.cpp:410-581 with the residuals of pose_graph.h:102-256).  Nothing of oracle/
// is included or linked.  The PnP RANSAC is the one the dynamic initialisation already uses (dyninit_host.cpp, cv::solvePnPRansac restated).
#include <math.h>
#include <string.h>

#include <array>
#include <string>
#include <vector>

#include "../../include/vio_posegraph.h"
#include "dyninit_host.h"

extern thread_local std::string g_err;   // vio_abi.hip

namespace {
using namespace dm;

const double kDeg = 3.14159265358979323846 / 180.0;

double wrap180(double a) { return a > 180.0 ? a - 360.0 : (a < -180.0 ? a + 360.0 : a); }   // Utility::normalizeAngle / NormalizeAngle

v3 euler_deg(const m3 &R) {   // Utility::R2ypr
    const double nx = R.a[0], ny = R.a[3], nz = R.a[6], ox = R.a[1], oy = R.a[4], ax = R.a[2], ay = R.a[5];
    const double y = atan2(ny, nx);
    const double p = atan2(-nz, nx * cos(y) + ny * sin(y));
    const double r = atan2(ax * sin(y) - ay * cos(y), -ox * sin(y) + oy * cos(y));
    return mk(y / kDeg, p / kDeg, r / kDeg);
}
m3 from_euler_deg(double yaw, double pitch, double roll) {   // Utility::ypr2R = Rz Ry Rx
    const double cy = cos(yaw * kDeg), sy = sin(yaw * kDeg), cp = cos(pitch * kDeg), sp = sin(pitch * kDeg), cr = cos(roll * kDeg), sr = sin(roll * kDeg);
    m3 R;
    R.a[0] = cy * cp; R.a[1] = cy * sp * sr - sy * cr; R.a[2] = cy * sp * cr + sy * sr;
    R.a[3] = sy * cp; R.a[4] = sy * sp * sr + cy * cr; R.a[5] = sy * sp * cr - cy * sr;
    R.a[6] = -sp;     R.a[7] = cp * sr;                R.a[8] = cp * cr;
    return R;
}

// ---- pose graph: nodes carry (yaw, t); an edge measures the pose of `b` in the yaw-only-variable frame of `a`
struct Link {
    int a, b;
    double t[3], yaw, pitch_a, roll_a;
    bool loop;   // FourDOFWeightError (yaw residual / 10) under HuberLoss(0.1) instead of FourDOFError
};
struct LinkEval { double r[4]; double Ja[4][4], Jb[4][4]; };   // d r / d (yaw, tx, ty, tz) of either end

void evaluate(const Link &e, const std::vector<double> &yaw, const std::vector<double> &pos, LinkEval &o) {
    // R(yaw) = Rz(yaw) M with M = Ry(pitch) Rx(roll) fixed: R^T d = M^T Rz^T d, and d/dyaw of Rz^T d is a 90 degree turn of it
    const double c = cos(yaw[e.a] * kDeg), s = sin(yaw[e.a] * kDeg);
    const m3 M = from_euler_deg(0.0, e.pitch_a, e.roll_a);
    const double d[3] = {pos[3 * e.b] - pos[3 * e.a], pos[3 * e.b + 1] - pos[3 * e.a + 1], pos[3 * e.b + 2] - pos[3 * e.a + 2]};
    const double u[3] = {c * d[0] + s * d[1], -s * d[0] + c * d[1], d[2]};            // Rz^T d
    const double du[3] = {(-s * d[0] + c * d[1]) * kDeg, (-c * d[0] - s * d[1]) * kDeg, 0.0};   // its derivative with respect to yaw (degrees)
    const double RzT[3][3] = {{c, s, 0}, {-s, c, 0}, {0, 0, 1}};
    memset(&o, 0, sizeof(o));
    for (int k = 0; k < 3; k++) {
        double v = 0, dv = 0;
        for (int q = 0; q < 3; q++) { v += M.a[3 * q + k] * u[q]; dv += M.a[3 * q + k] * du[q]; }
        o.r[k] = v - e.t[k];
        o.Ja[k][0] = dv;
        for (int q = 0; q < 3; q++) {
            double g = 0;
            for (int w = 0; w < 3; w++) g += M.a[3 * w + k] * RzT[w][q];   // (M^T Rz^T)[k][q]
            o.Ja[k][1 + q] = -g;
            o.Jb[k][1 + q] = g;
        }
    }
    const double wy = e.loop ? 0.1 : 1.0;
    o.r[3] = wrap180(yaw[e.b] - yaw[e.a] - e.yaw) * wy;
    o.Ja[3][0] = -wy;
    o.Jb[3][0] = wy;
}
double huber_half(double s) { return 0.5 * (s <= 0.01 ? s : 0.2 * sqrt(s) - 0.01); }   // 0.5 rho(s), HuberLoss(0.1)

double total_cost(const std::vector<Link> &links, const std::vector<double> &yaw, const std::vector<double> &pos) {
    double c = 0;
    LinkEval ev;
    for (const Link &e : links) {
        evaluate(e, yaw, pos, ev);
        const double s = ev.r[0] * ev.r[0] + ev.r[1] * ev.r[1] + ev.r[2] * ev.r[2] + ev.r[3] * ev.r[3];
        c += e.loop ? huber_half(s) : 0.5 * s;
    }
    return c;
}

// The normal equations of a pose graph are banded -- a node meets its four predecessors -- plus one long row per loop edge (the reference hands the
// problem to SPARSE_NORMAL_CHOLESKY, pose_graph.cpp:421).  They are kept in ENVELOPE form: row i of the lower triangle from its first structural
// non-zero first[i] to the diagonal.  A Cholesky factor has the envelope of its matrix, so the factorisation costs O(sum of len_i * band) and the
// storage O(n * band + loop spans) instead of the dense n^3 / n^2 (3000 keyframes were 3 GB and minutes per solve).  Every sum runs over
// the same terms in the same order as the dense row-oriented Cholesky minus its exact zeros: the factor has the same bits.
struct SkyMat {
    int n = 0;
    std::vector<int> first;        // first stored column of row i
    std::vector<size_t> off;       // v[off[i] + (j - first[i])] = A(i, j), first[i] <= j <= i
    std::vector<double> v;
    void init(const std::vector<int> &first_col) {
        n = (int)first_col.size();
        first = first_col;
        off.resize(n + 1);
        off[0] = 0;
        for (int i = 0; i < n; i++) off[i + 1] = off[i] + (size_t)(i - first[i] + 1);
        v.assign(off[n], 0.0);
    }
    double &at(int i, int j) { return v[off[i] + (size_t)(j - first[i])]; }          // i >= j >= first[i]
    double get(int i, int j) const { if (i < j) { int q = i; i = j; j = q; } return j < first[i] ? 0.0 : v[off[i] + (size_t)(j - first[i])]; }
    void add_sym(int r, int c, double x) { if (r >= c) at(r, c) += x; }               // callers visit (r, c) and (c, r): the lower one is kept
    void matvec(const std::vector<double> &x, std::vector<double> &y) const {
        y.assign(n, 0.0);
        for (int i = 0; i < n; i++) {
            const double *row = &v[off[i]];
            double acc = 0;
            for (int j = first[i]; j < i; j++) { acc += row[j - first[i]] * x[j]; y[j] += row[j - first[i]] * x[i]; }
            y[i] += acc + row[i - first[i]] * x[i];
        }
    }
    // in-place L L^T and the solve of L L^T x = b; false if a pivot is not positive
    bool chol_solve(std::vector<double> &x) {
        for (int i = 0; i < n; i++) {
            double *ri = &v[off[i]];
            const int fi = first[i];
            for (int j = fi; j < i; j++) {
                const double *rj = &v[off[j]];
                const int fj = first[j], k0 = fi > fj ? fi : fj;
                double acc = ri[j - fi];
                for (int k = k0; k < j; k++) acc -= ri[k - fi] * rj[k - fj];
                ri[j - fi] = acc / rj[j - fj];
            }
            double d = ri[i - fi];
            for (int k = fi; k < i; k++) d -= ri[k - fi] * ri[k - fi];
            if (!(d > 0.0) || !isfinite(d)) return false;
            ri[i - fi] = sqrt(d);
        }
        for (int i = 0; i < n; i++) {
            const double *ri = &v[off[i]];
            double acc = x[i];
            for (int k = first[i]; k < i; k++) acc -= ri[k - first[i]] * x[k];
            x[i] = acc / ri[i - first[i]];
        }
        for (int i = n - 1; i >= 0; i--) {
            const double *ri = &v[off[i]];
            x[i] /= ri[i - first[i]];
            for (int k = first[i]; k < i; k++) x[k] -= ri[k - first[i]] * x[i];
        }
        return true;
    }
};
// envelope of the normal equations: a row of node b's block starts at the first column of the lowest-slot node b shares an edge with
template <class L> std::vector<int> envelope_of(const std::vector<L> &links, const std::vector<int> &slot, int nvar, int dof) {
    std::vector<int> first(nvar);
    for (int a = 0; a < nvar; a++) first[a] = a - a % dof;
    for (const L &e : links) {
        const int sa = slot[e.a], sb = slot[e.b];
        if (sa < 0 || sb < 0) continue;
        const int lo = sa < sb ? sa : sb, hi = sa < sb ? sb : sa;
        for (int c = 0; c < dof; c++) if (first[hi + c] > lo) first[hi + c] = lo;
    }
    return first;
}

}  // namespace

extern "C" int vio_pg_find_connection(int n, const float *pt3d, const double *pt_id, const int32_t *match, const float *old_norm, const double *vio_T,
                                      const double *vio_R, const double *qic, const double *tic, int min_loop_num, double *loop_info,
                                      double *match_points, int32_t *n_match, double *pnp_T, double *pnp_R) {
    if (n < 0 || !vio_T || !vio_R || !qic || !tic || !loop_info || !match_points || !n_match || (n > 0 && (!pt3d || !pt_id || !match || !old_norm))) {
        g_err = "vio_pg_find_connection: null argument";
        return VIO_EINVAL;
    }
    *n_match = 0;
    // the matched subset, window order (reduceVector keeps the order, keyframe.cpp:4-11)
    std::vector<v3> world;
    std::vector<std::array<double, 2>> seen_old;
    std::vector<double> ids;
    for (int i = 0; i < n; i++) {
        if (match[i] < 0) continue;
        world.push_back(mk(pt3d[3 * i], pt3d[3 * i + 1], pt3d[3 * i + 2]));
        seen_old.push_back({(double)old_norm[2 * match[i]], (double)old_norm[2 * match[i] + 1]});
        ids.push_back(pt_id[i]);
    }
    if ((int)world.size() <= min_loop_num) return 0;                       // :404
    const m3 Rwi = ldm(vio_R), Ric = ldm(qic);
    const v3 Twi = ld3(vio_T), tci = ld3(tic);
    // KeyFrame::PnPRANSAC: the guess is the keyframe's own camera pose, the answer the world -> old-camera transform
    m3 Rcw = tr(mul(Rwi, Ric));
    v3 tcw = neg(mul(Rcw, add(Twi, mul(Rwi, tci))));
    std::vector<uint8_t> keep;
    {
        m3 R;
        v3 t;
        if (vinit::pnp_ransac_with_inliers(world, seen_old, 100, 10.0 / 460.0, 0.99, R, t, keep)) { Rcw = R; tcw = t; }
        else keep.assign(world.size(), 0);
    }
    const m3 Rw_cold = tr(Rcw);
    const v3 Tw_cold = mul(Rw_cold, neg(tcw));
    const m3 R_old = mul(Rw_cold, tr(Ric));                                // PnP_R_old
    const v3 T_old = sub(Tw_cold, mul(R_old, tci));                        // PnP_T_old
    if (pnp_T) st3(pnp_T, T_old);
    if (pnp_R) stm(pnp_R, R_old);
    int k = 0;
    for (size_t i = 0; i < world.size(); i++)
        if (keep[i]) { match_points[3 * k] = seen_old[i][0]; match_points[3 * k + 1] = seen_old[i][1]; match_points[3 * k + 2] = ids[i]; k++; }
    if (k <= min_loop_num) return 0;                                       // :482
    const v3 rel_t = mul(tr(R_old), sub(Twi, T_old));
    const quat rel_q = R2q(mul(tr(R_old), Rwi));
    const double rel_yaw = wrap180(euler_deg(Rwi).x - euler_deg(R_old).x);
    if (!(fabs(rel_yaw) < 30.0 && nrm(rel_t) < 20.0)) return 0;            // :487
    loop_info[0] = rel_t.x; loop_info[1] = rel_t.y; loop_info[2] = rel_t.z;
    loop_info[3] = rel_q.w; loop_info[4] = rel_q.x; loop_info[5] = rel_q.y; loop_info[6] = rel_q.z; loop_info[7] = rel_yaw;
    *n_match = k;
    return 1;
}

extern "C" int vio_pg_optimize4dof(int n, const double *t, const double *R, const int32_t *sequence, const int32_t *loop_to, const double *loop_info,
                                   double *t_out, double *R_out, double *drift) {
    if (n < 1 || !t || !R || !sequence || !loop_to || !loop_info || !t_out || !R_out || !drift) { g_err = "vio_pg_optimize4dof: bad argument"; return VIO_EINVAL; }
    std::vector<double> yaw(n), pitch(n), roll(n), pos(t, t + 3 * (size_t)n);
    std::vector<int> slot(n, -1);
    int nvar = 0;
    for (int i = 0; i < n; i++) {
        const v3 e = euler_deg(q2R(R2q(ldm(R + 9 * i))));   // through the quaternion like the reference (q_array)
        yaw[i] = e.x; pitch[i] = e.y; roll[i] = e.z;
        if (!(i == 0 || sequence[i] == 0)) { slot[i] = nvar; nvar += 4; }
    }
    std::vector<Link> links;
    for (int i = 0; i < n; i++) {
        for (int back = 1; back <= 4; back++) {
            const int a = i - back;
            if (a < 0 || sequence[a] != sequence[i]) continue;
            const v3 d = mul(tr(q2R(R2q(ldm(R + 9 * a)))), mk(t[3 * i] - t[3 * a], t[3 * i + 1] - t[3 * a + 1], t[3 * i + 2] - t[3 * a + 2]));
            links.push_back(Link{a, i, {d.x, d.y, d.z}, yaw[i] - yaw[a], pitch[a], roll[a], false});
        }
        if (loop_to[i] >= 0) {
            if (loop_to[i] >= n) { g_err = "vio_pg_optimize4dof: loop partner out of range"; return VIO_EINVAL; }
            const int a = loop_to[i];
            links.push_back(Link{a, i, {loop_info[8 * i], loop_info[8 * i + 1], loop_info[8 * i + 2]}, loop_info[8 * i + 7], pitch[a], roll[a], true});
        }
    }
    if (nvar > 0) {
        // Levenberg-Marquardt as Ceres runs it by default: (J^T J + diag(J^T J) / radius) step = -J^T r in Jacobi-scaled variables, step quality
        // rho against the quadratic model, radius / max(1/3, 1 - (2 rho - 1)^3) on success, radius / 2, / 4, ... on failure; 5 iterations
        double radius = 1e4, shrink = 2.0, cost = total_cost(links, yaw, pos);
        const std::vector<int> env = envelope_of(links, slot, nvar, 4);
        std::vector<double> colscale;
        int iterations = 0;
        while (iterations < 5) {
            SkyMat H;
            H.init(env);
            std::vector<double> g(nvar, 0.0);
            LinkEval ev;
            for (const Link &e : links) {
                evaluate(e, yaw, pos, ev);
                double w = 1.0;
                if (e.loop) {   // robustified Gauss-Newton: rho'' <= 0 for Huber, so residual and Jacobian are scaled by sqrt(rho')
                    const double s = ev.r[0] * ev.r[0] + ev.r[1] * ev.r[1] + ev.r[2] * ev.r[2] + ev.r[3] * ev.r[3];
                    if (s > 0.01) w = sqrt(0.1 / sqrt(s));
                }
                const int sa = slot[e.a], sb = slot[e.b];
                auto J = [&](int end, int row, int c) { return w * (end == 0 ? ev.Ja[row][c] : ev.Jb[row][c]); };
                const int base[2] = {sa, sb};
                for (int e0 = 0; e0 < 2; e0++) {
                    if (base[e0] < 0) continue;
                    for (int c0 = 0; c0 < 4; c0++) {
                        double gs = 0;
                        for (int row = 0; row < 4; row++) gs += J(e0, row, c0) * w * ev.r[row];
                        g[base[e0] + c0] += gs;
                        for (int e1 = 0; e1 < 2; e1++) {
                            if (base[e1] < 0) continue;
                            for (int c1 = 0; c1 < 4; c1++) {
                                double hs = 0;
                                for (int row = 0; row < 4; row++) hs += J(e0, row, c0) * J(e1, row, c1);
                                H.add_sym(base[e0] + c0, base[e1] + c1, hs);
                            }
                        }
                    }
                }
            }
            if (colscale.empty()) { colscale.resize(nvar); for (int a = 0; a < nvar; a++) colscale[a] = 1.0 / (1.0 + sqrt(H.at(a, a))); }
            double gmax = 0;
            for (int a = 0; a < nvar; a++) gmax = fmax(gmax, fabs(g[a]));
            if (gmax <= 1e-10) break;
            SkyMat Hs = H;
            std::vector<double> gsv(nvar), hv;
            for (int a = 0; a < nvar; a++) { gsv[a] = colscale[a] * g[a]; for (int b = Hs.first[a]; b <= a; b++) Hs.at(a, b) = colscale[a] * colscale[b] * H.at(a, b); }
            bool moved = false;
            while (!moved && iterations < 5) {
                SkyMat A = Hs;
                std::vector<double> step = gsv;
                for (int a = 0; a < nvar; a++) A.at(a, a) += fmin(fmax(Hs.at(a, a), 1e-6), 1e32) / radius;
                if (!A.chol_solve(step)) { radius /= shrink; shrink *= 2; iterations++; continue; }
                double lin = 0, quad = 0;
                Hs.matvec(step, hv);
                for (int a = 0; a < nvar; a++) { lin += gsv[a] * step[a]; quad += step[a] * hv[a]; }
                const double model = lin - 0.5 * quad;   // decrease predicted for x - step
                std::vector<double> yc = yaw, pc = pos;
                for (int i = 0; i < n; i++) {
                    if (slot[i] < 0) continue;
                    yc[i] = wrap180(yaw[i] - step[slot[i]] * colscale[slot[i]]);   // AngleLocalParameterization
                    for (int a = 0; a < 3; a++) pc[3 * i + a] = pos[3 * i + a] - step[slot[i] + 1 + a] * colscale[slot[i] + 1 + a];
                }
                const double cnew = total_cost(links, yc, pc);
                const double rho = model > 0 ? (cost - cnew) / model : -1.0;
                iterations++;
                if (rho > 1e-3) {
                    const double rel = fabs(cost - cnew) / cost;
                    yaw.swap(yc); pos.swap(pc); cost = cnew;
                    radius = fmin(radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3)), 1e16);
                    shrink = 2.0;
                    moved = true;
                    if (rel < 1e-6) iterations = 5;   // function tolerance
                } else { radius /= shrink; shrink *= 2; }
            }
            if (!moved) break;
        }
    }
    for (int i = 0; i < n; i++) {
        const m3 Ro = from_euler_deg(yaw[i], pitch[i], roll[i]);
        for (int a = 0; a < 3; a++) t_out[3 * i + a] = pos[3 * i + a];
        stm(R_out + 9 * i, Ro);
    }
    {   // drift of the newest keyframe against its VIO pose (:547-553)
        const int l = n - 1;
        const double yd = euler_deg(from_euler_deg(yaw[l], pitch[l], roll[l])).x - euler_deg(ldm(R + 9 * l)).x;
        const m3 Rd = from_euler_deg(yd, 0.0, 0.0);
        const v3 td = sub(mk(pos[3 * l], pos[3 * l + 1], pos[3 * l + 2]), mul(Rd, mk(t[3 * l], t[3 * l + 1], t[3 * l + 2])));
        drift[0] = yd; drift[1] = td.x; drift[2] = td.y; drift[3] = td.z;
    }
    return VIO_OK;
}

// ---------------------------------------------------------------------------------------------------------------- optimize6DoF
namespace {
using namespace dm;

// RelativeRTError (pose_graph.h:256-320): the pose of node b in the frame of node a against a measured (t, q); t_var 0.1, q_var 0.01
struct Link6 { int a, b; v3 t; quat q; bool loop; };
struct Link6Eval { double r[6]; double Ja[6][6], Jb[6][6]; };   // columns: left half-angle perturbation of the quaternion (3), translation (3)

v3 qvec_of_sandwich(quat A, v3 d, quat B) {   // vector part of A * [0, d] * B
    const v3 av = mk(A.x, A.y, A.z), bv = mk(B.x, B.y, B.z);
    const double s = -dot(d, bv);
    const v3 v = add(scl(B.w, d), cross(d, bv));
    return add(add(scl(A.w, v), scl(s, av)), cross(av, v));
}

void evaluate6(const Link6 &e, const std::vector<quat> &q, const std::vector<double> &pos, Link6Eval &o, bool withJ) {
    const double tv = 0.1, qv = 0.01;
    const m3 RaT = tr(q2R(qnormalized(q[e.a])));     // ceres::QuaternionRotatePoint normalises
    const v3 d = mk(pos[3 * e.b] - pos[3 * e.a], pos[3 * e.b + 1] - pos[3 * e.a + 1], pos[3 * e.b + 2] - pos[3 * e.a + 2]);
    const v3 u = mul(RaT, d);
    o.r[0] = (u.x - e.t.x) / tv; o.r[1] = (u.y - e.t.y) / tv; o.r[2] = (u.z - e.t.z) / tv;
    const quat qac = mkq(q[e.a].w, -q[e.a].x, -q[e.a].y, -q[e.a].z), qmc = mkq(e.q.w, -e.q.x, -e.q.y, -e.q.z);
    const quat A = qmul(qmc, qac), err = qmul(A, q[e.b]);
    o.r[3] = 2.0 * err.x / qv; o.r[4] = 2.0 * err.y / qv; o.r[5] = 2.0 * err.z / qv;
    if (!withJ) return;
    memset(o.Ja, 0, sizeof(o.Ja)); memset(o.Jb, 0, sizeof(o.Jb));
    // R_a -> (I + 2 [dtheta]x) R_a:  R_a^T d -> R_a^T d + 2 R_a^T [d]x dtheta
    const m3 Rd = mul(RaT, skew(d));
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            o.Ja[r][c] = 2.0 * Rd.a[3 * r + c] / tv;
            o.Ja[r][3 + c] = -RaT.a[3 * r + c] / tv;
            o.Jb[r][3 + c] = RaT.a[3 * r + c] / tv;
        }
    for (int c = 0; c < 3; c++) {
        const v3 col = qvec_of_sandwich(A, mk(c == 0, c == 1, c == 2), q[e.b]);
        const double v[3] = {2.0 * col.x / qv, 2.0 * col.y / qv, 2.0 * col.z / qv};
        for (int r = 0; r < 3; r++) { o.Jb[3 + r][c] = v[r]; o.Ja[3 + r][c] = -v[r]; }
    }
}

double total_cost6(const std::vector<Link6> &links, const std::vector<quat> &q, const std::vector<double> &pos) {
    double c = 0;
    Link6Eval ev;
    for (const Link6 &e : links) {
        evaluate6(e, q, pos, ev, false);
        double s = 0;
        for (int k = 0; k < 6; k++) s += ev.r[k] * ev.r[k];
        c += e.loop ? huber_half(s) : 0.5 * s;
    }
    return c;
}

quat quat_plus(quat q, const double *d) {   // ceres::QuaternionParameterization::Plus
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (!(n > 0.0)) return q;
    const double s = sin(n) / n;
    return qmul(mkq(cos(n), s * d[0], s * d[1], s * d[2]), q);
}

}  // namespace

extern "C" int vio_pg_optimize6dof(int n, const double *t, const double *R, const int32_t *sequence, const int32_t *loop_to, const double *loop_info,
                                   double *t_out, double *R_out, double *drift12) {
    if (n < 1 || !t || !R || !sequence || !loop_to || !loop_info || !t_out || !R_out || !drift12) { g_err = "vio_pg_optimize6dof: bad argument"; return VIO_EINVAL; }
    std::vector<quat> q(n);
    std::vector<double> pos(t, t + 3 * (size_t)n);
    std::vector<int> slot(n, -1);
    int nvar = 0;
    for (int i = 0; i < n; i++) {
        q[i] = R2q(ldm(R + 9 * i));
        if (!(i == 0 || sequence[i] == 0)) { slot[i] = nvar; nvar += 6; }
    }
    std::vector<Link6> links;
    for (int i = 0; i < n; i++) {
        for (int back = 1; back <= 4; back++) {
            const int a = i - back;
            if (a < 0 || sequence[a] != sequence[i]) continue;
            const quat qai = qinv(q[a]);
            links.push_back(Link6{a, i, mul(q2R(qai), mk(t[3 * i] - t[3 * a], t[3 * i + 1] - t[3 * a + 1], t[3 * i + 2] - t[3 * a + 2])), qmul(qai, q[i]), false});
        }
        if (loop_to[i] >= 0) {
            if (loop_to[i] >= n) { g_err = "vio_pg_optimize6dof: loop partner out of range"; return VIO_EINVAL; }
            links.push_back(Link6{loop_to[i], i, mk(loop_info[8 * i], loop_info[8 * i + 1], loop_info[8 * i + 2]),
                                  mkq(loop_info[8 * i + 3], loop_info[8 * i + 4], loop_info[8 * i + 5], loop_info[8 * i + 6]), true});
        }
    }
    if (nvar > 0) {   // the trust-region loop of vio_pg_optimize4dof with six columns per node
        double radius = 1e4, shrink = 2.0, cost = total_cost6(links, q, pos);
        const std::vector<int> env = envelope_of(links, slot, nvar, 6);
        std::vector<double> colscale;
        int iterations = 0;
        while (iterations < 5) {
            SkyMat H;
            H.init(env);
            std::vector<double> g(nvar, 0.0);
            Link6Eval ev;
            for (const Link6 &e : links) {
                evaluate6(e, q, pos, ev, true);
                double w = 1.0;
                if (e.loop) {
                    double s = 0;
                    for (int k = 0; k < 6; k++) s += ev.r[k] * ev.r[k];
                    if (s > 0.01) w = sqrt(0.1 / sqrt(s));
                }
                const int base[2] = {slot[e.a], slot[e.b]};
                auto J = [&](int end, int row, int c) { return w * (end == 0 ? ev.Ja[row][c] : ev.Jb[row][c]); };
                for (int e0 = 0; e0 < 2; e0++) {
                    if (base[e0] < 0) continue;
                    for (int c0 = 0; c0 < 6; c0++) {
                        double gs = 0;
                        for (int row = 0; row < 6; row++) gs += J(e0, row, c0) * w * ev.r[row];
                        g[base[e0] + c0] += gs;
                        for (int e1 = 0; e1 < 2; e1++) {
                            if (base[e1] < 0) continue;
                            for (int c1 = 0; c1 < 6; c1++) {
                                double hs = 0;
                                for (int row = 0; row < 6; row++) hs += J(e0, row, c0) * J(e1, row, c1);
                                H.add_sym(base[e0] + c0, base[e1] + c1, hs);
                            }
                        }
                    }
                }
            }
            if (colscale.empty()) { colscale.resize(nvar); for (int a = 0; a < nvar; a++) colscale[a] = 1.0 / (1.0 + sqrt(H.at(a, a))); }
            double gmax = 0;
            for (int a = 0; a < nvar; a++) gmax = fmax(gmax, fabs(g[a]));
            if (gmax <= 1e-10) break;
            SkyMat Hs = H;
            std::vector<double> gsv(nvar), hv;
            for (int a = 0; a < nvar; a++) { gsv[a] = colscale[a] * g[a]; for (int b = Hs.first[a]; b <= a; b++) Hs.at(a, b) = colscale[a] * colscale[b] * H.at(a, b); }
            bool moved = false;
            int tries = 0;
            while (!moved && tries++ < 20) {
                SkyMat A = Hs;
                std::vector<double> step = gsv;
                for (int a = 0; a < nvar; a++) A.at(a, a) += fmin(fmax(Hs.at(a, a), 1e-6), 1e32) / radius;
                if (!A.chol_solve(step)) { radius /= shrink; shrink *= 2; continue; }
                double lin = 0, quad = 0;
                Hs.matvec(step, hv);
                for (int a = 0; a < nvar; a++) { lin += gsv[a] * step[a]; quad += step[a] * hv[a]; }
                const double model = lin - 0.5 * quad;
                std::vector<quat> qc = q;
                std::vector<double> pc = pos;
                for (int i = 0; i < n; i++) {
                    if (slot[i] < 0) continue;
                    double d[3];
                    for (int a = 0; a < 3; a++) d[a] = -step[slot[i] + a] * colscale[slot[i] + a];
                    qc[i] = quat_plus(q[i], d);
                    for (int a = 0; a < 3; a++) pc[3 * i + a] = pos[3 * i + a] - step[slot[i] + 3 + a] * colscale[slot[i] + 3 + a];
                }
                const double cnew = total_cost6(links, qc, pc);
                const double rho = model > 0 ? (cost - cnew) / model : -1.0;
                if (rho > 1e-3) {
                    const double rel = fabs(cost - cnew) / cost;
                    q.swap(qc); pos.swap(pc); cost = cnew;
                    radius = fmin(radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rho - 1.0, 3)), 1e16);
                    shrink = 2.0;
                    moved = true;
                    iterations++;
                    if (rel < 1e-6) iterations = 5;
                } else {
                    radius /= shrink; shrink *= 2;
                    iterations++;          // an unsuccessful step counts as an iteration
                    if (iterations >= 5) break;
                }
            }
            if (!moved) break;
        }
    }
    for (int i = 0; i < n; i++) {
        for (int a = 0; a < 3; a++) t_out[3 * i + a] = pos[3 * i + a];
        stm(R_out + 9 * i, q2R(q[i]));
    }
    {   // r_drift = R_cur R_vio^T, t_drift = t_cur - r_drift t_vio (pose_graph.cpp:717-721)
        const int l = n - 1;
        const m3 rd = mul(q2R(q[l]), tr(ldm(R + 9 * l)));
        const v3 td = sub(mk(pos[3 * l], pos[3 * l + 1], pos[3 * l + 2]), mul(rd, mk(t[3 * l], t[3 * l + 1], t[3 * l + 2])));
        stm(drift12, rd);
        drift12[9] = td.x; drift12[10] = td.y; drift12[11] = td.z;
    }
    return VIO_OK;
}
