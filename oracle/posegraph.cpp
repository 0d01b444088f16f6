// ORACLE (test infrastructure) -- CPU restatement of the loop-closure path of pose_graph (place recognition: bow.cpp):
//   KeyFrame::computeWindowBRIEFPoint / computeBRIEFPoint   pose_graph/src/keyframe/keyframe.cpp:80-124
//   DVision::BRIEF::compute                                 pose_graph/src/ThirdParty/DVision/BRIEF.cpp (GaussianBlur 9x9 sigma 2, then 256 pair tests)
//   KeyFrame::searchInAera / searchByBRIEFDes / HammingDis  keyframe.cpp:126-169, 530
//   KeyFrame::PnPRANSAC                                     keyframe.cpp:195-250  (cv::solvePnPRansac, restated in initial.cpp)
//   KeyFrame::findConnection                                keyframe.cpp:252-528  (gating :404, :482-490; match list :491-520)
//   PoseGraph::optimize4DoF                                 pose_graph/src/pose_graph/pose_graph.cpp:410-581, residuals pose_graph.h:102-256
// PoseGraph::detectLoop (:308, the DBoW2 query) is restated in bow.cpp.  Parity unpinned like the rest of oracle/: OpenCV's GaussianBlur / FAST / solvePnPRansac and Ceres' LM are restated
// from their published algorithms.  Only tests/ may use this file.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "oracle.h"

namespace ovio {
using namespace om;

// cv::GaussianBlur(src, dst, Size(9, 9), 2, 2), CV_8U, BORDER_REFLECT_101: separable, the float kernel exp(-x^2 / 8) / sum quantised to 8
// fractional bits per pass (52 46 32 17 7, sum 256) and the result rounded once: (sum + 2^15) >> 16 (the 8-bit fixed-point filter engine)
void gaussian_blur_9x9(const uint8_t *src, int W, int H, uint8_t *dst) {
    static const int k[9] = {7, 17, 32, 46, 52, 46, 32, 17, 7};
    std::vector<int> tmp((size_t)W * H);
    auto refl = [](int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); };
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int a = 0;
            for (int i = 0; i < 9; i++) a += k[i] * src[(size_t)y * W + refl(x + i - 4, W)];
            tmp[(size_t)y * W + x] = a;
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int a = 0;
            for (int i = 0; i < 9; i++) a += k[i] * tmp[(size_t)refl(y + i - 4, H) * W + x];
            dst[(size_t)y * W + x] = (uint8_t)((a + (1 << 15)) >> 16);
        }
}

// DVision::BRIEF::compute on an already blurred image: bit i = I(p + (x1, y1)_i) < I(p + (x2, y2)_i) when both samples are inside the image
// ((int)(pt + offset): float addition, truncation toward zero).  pat = x1[256] y1[256] x2[256] y2[256]; desc = 4 x 64 bits per point.
void brief_compute(const uint8_t *blur, int W, int H, const float *xy, int n, const int *pat, uint64_t *desc) {
    for (int p = 0; p < n; p++) {
        uint64_t d[4] = {0, 0, 0, 0};
        const float px = xy[2 * p], py = xy[2 * p + 1];
        for (int i = 0; i < 256; i++) {
            const int x1 = (int)(px + (float)pat[i]), y1 = (int)(py + (float)pat[256 + i]);
            const int x2 = (int)(px + (float)pat[512 + i]), y2 = (int)(py + (float)pat[768 + i]);
            if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H && x2 >= 0 && x2 < W && y2 >= 0 && y2 < H)
                if (blur[(size_t)y1 * W + x1] < blur[(size_t)y2 * W + x2]) d[i >> 6] |= 1ULL << (i & 63);
        }
        for (int q = 0; q < 4; q++) desc[4 * (size_t)p + q] = d[q];
    }
}

// searchByBRIEFDes: for every window descriptor the first old descriptor with the smallest Hamming distance below 128; a match iff < 80
void brief_match(const uint64_t *wd, int n, const uint64_t *od, int m, int *best_index, int *best_dist) {
    for (int i = 0; i < n; i++) {
        int bd = 128, bi = -1;
        for (int j = 0; j < m; j++) {
            int dis = 0;
            for (int q = 0; q < 4; q++) dis += __builtin_popcountll(wd[4 * (size_t)i + q] ^ od[4 * (size_t)j + q]);
            if (dis < bd) { bd = dis; bi = j; }
        }
        best_dist[i] = bd;
        best_index[i] = (bi != -1 && bd < 80) ? bi : -1;
    }
}

static double normalize_angle_deg(double a) {  // Utility::normalizeAngle (pose_graph/src/utility/utility.h)
    if (a > 180.0) return a - 360.0;
    if (a < -180.0) return a + 360.0;
    return a;
}

// KeyFrame::findConnection after the descriptor search.  cur: n window points (3-D world point, normalised observation, feature id) with
// their match (index into the old keyframe's keypoints or -1), the keyframe's origin_vio pose; old_norm: the old keyframe's normalised
// keypoints.  Returns 1 and fills loop_info (relative_t, relative_q wxyz, relative_yaw) and the match list (x_old_norm, y_old_norm, id) --
// exactly what pose_graph publishes to the estimator (:491-520 -> Estimator::setReloFrame) -- iff more than min_loop_num matches survive the
// PnP RANSAC and the relative pose passes the |yaw| < 30 deg, |t| < 20 m gate.
int find_connection(int n, const float *pt3d, const float *pt_norm, const double *pt_id, const int *match, const float *old_norm,
                    const double *vio_T, const double *vio_R, const double *qic9, const double *tic3, int min_loop_num,
                    double *loop_info8, double *match_points, int *n_match_out, double *pnp_T3, double *pnp_R9) {
    *n_match_out = 0;
    std::vector<V3> m3;
    std::vector<std::array<double, 2>> m2o;
    std::vector<double> mid;
    for (int i = 0; i < n; i++) {
        if (match[i] < 0) continue;
        m3.push_back(V3(pt3d[3 * i], pt3d[3 * i + 1], pt3d[3 * i + 2]));
        m2o.push_back({(double)old_norm[2 * match[i]], (double)old_norm[2 * match[i] + 1]});
        mid.push_back(pt_id[i]);
    }
    (void)pt_norm;
    if ((int)m3.size() <= min_loop_num) return 0;
    M3 oR, qic;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { oR(i, j) = vio_R[3 * i + j]; qic(i, j) = qic9[3 * i + j]; }
    const V3 oT(vio_T[0], vio_T[1], vio_T[2]), tic(tic3[0], tic3[1], tic3[2]);
    // PnPRANSAC (:195-250): initial guess = the keyframe's own camera pose (only used when the RANSAC finds nothing)
    M3 R_w_c = oR * qic;
    V3 T_w_c = oT + oR * tic;
    M3 R_pnp = T(R_w_c);
    V3 T_pnp = V3() - R_pnp * T_w_c;
    std::vector<uint8_t> inl;
    M3 Rr;
    V3 tr;
    if (solve_pnp_ransac_epnp(m3, m2o, 100, 10.0 / 460.0, 0.99, Rr, tr, inl)) { R_pnp = Rr; T_pnp = tr; }
    else inl.assign(m3.size(), 0);
    M3 R_w_c_old = T(R_pnp);
    V3 T_w_c_old = R_w_c_old * (V3() - T_pnp);
    M3 PnP_R_old = R_w_c_old * T(qic);
    V3 PnP_T_old = T_w_c_old - PnP_R_old * tic;
    int k = 0;
    for (size_t i = 0; i < m3.size(); i++)
        if (inl[i]) { match_points[3 * k] = m2o[i][0]; match_points[3 * k + 1] = m2o[i][1]; match_points[3 * k + 2] = mid[i]; k++; }
    for (int i = 0; i < 3; i++) { pnp_T3[i] = i == 0 ? PnP_T_old.x : (i == 1 ? PnP_T_old.y : PnP_T_old.z); for (int j = 0; j < 3; j++) pnp_R9[3 * i + j] = PnP_R_old(i, j); }
    if (k <= min_loop_num) return 0;
    V3 rel_t = T(PnP_R_old) * (oT - PnP_T_old);
    Q rel_q = fromR(T(PnP_R_old) * oR);
    const double rel_yaw = normalize_angle_deg(R2ypr(oR).x - R2ypr(PnP_R_old).x);
    if (!(std::fabs(rel_yaw) < 30.0 && norm(rel_t) < 20.0)) return 0;
    loop_info8[0] = rel_t.x; loop_info8[1] = rel_t.y; loop_info8[2] = rel_t.z;
    loop_info8[3] = rel_q.w; loop_info8[4] = rel_q.x; loop_info8[5] = rel_q.y; loop_info8[6] = rel_q.z; loop_info8[7] = rel_yaw;
    *n_match_out = k;
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------- optimize4DoF
namespace {
void ypr_to_R(double yaw, double pitch, double roll, double R[9]) {  // pose_graph.h:128-145 (degrees)
    const double y = yaw / 180.0 * M_PI, p = pitch / 180.0 * M_PI, r = roll / 180.0 * M_PI;
    R[0] = std::cos(y) * std::cos(p); R[1] = -std::sin(y) * std::cos(r) + std::cos(y) * std::sin(p) * std::sin(r); R[2] = std::sin(y) * std::sin(r) + std::cos(y) * std::sin(p) * std::cos(r);
    R[3] = std::sin(y) * std::cos(p); R[4] = std::cos(y) * std::cos(r) + std::sin(y) * std::sin(p) * std::sin(r); R[5] = -std::cos(y) * std::sin(r) + std::sin(y) * std::sin(p) * std::cos(r);
    R[6] = -std::sin(p); R[7] = std::cos(p) * std::sin(r); R[8] = std::cos(p) * std::cos(r);
}
struct Edge4 { int i, j; double t[3], yaw, pitch_i, roll_i; bool loop; };
// residual (4) and Jacobian wrt (yaw_i, t_i(3), yaw_j, t_j(3)) = 4 x 8 of FourDOFError / FourDOFWeightError (weight 1, yaw / 10)
void edge_eval(const Edge4 &e, const double *yaw, const double *t, double r[4], double J[32]) {
    const double yi = yaw[e.i], yj = yaw[e.j];
    double R[9], dR[9];
    ypr_to_R(yi, e.pitch_i, e.roll_i, R);
    {   // d R / d yaw (yaw in degrees)
        const double y = yi / 180.0 * M_PI, p = e.pitch_i / 180.0 * M_PI, rr = e.roll_i / 180.0 * M_PI, s = M_PI / 180.0;
        dR[0] = -std::sin(y) * std::cos(p) * s; dR[1] = (-std::cos(y) * std::cos(rr) - std::sin(y) * std::sin(p) * std::sin(rr)) * s; dR[2] = (std::cos(y) * std::sin(rr) - std::sin(y) * std::sin(p) * std::cos(rr)) * s;
        dR[3] = std::cos(y) * std::cos(p) * s; dR[4] = (-std::sin(y) * std::cos(rr) + std::cos(y) * std::sin(p) * std::sin(rr)) * s; dR[5] = (std::sin(y) * std::sin(rr) + std::cos(y) * std::sin(p) * std::cos(rr)) * s;
        dR[6] = 0; dR[7] = 0; dR[8] = 0;
    }
    const double d[3] = {t[3 * e.j] - t[3 * e.i], t[3 * e.j + 1] - t[3 * e.i + 1], t[3 * e.j + 2] - t[3 * e.i + 2]};
    const double wy = e.loop ? 0.1 : 1.0;
    std::memset(J, 0, 32 * sizeof(double));
    for (int a = 0; a < 3; a++) {
        // t_i_ij = R^T d
        r[a] = R[0 + a] * d[0] + R[3 + a] * d[1] + R[6 + a] * d[2] - e.t[a];
        J[a * 8 + 0] = dR[0 + a] * d[0] + dR[3 + a] * d[1] + dR[6 + a] * d[2];
        for (int b = 0; b < 3; b++) { J[a * 8 + 1 + b] = -R[3 * b + a]; J[a * 8 + 5 + b] = R[3 * b + a]; }
    }
    r[3] = normalize_angle_deg(yj - yi - e.yaw) * wy;
    J[3 * 8 + 0] = -wy; J[3 * 8 + 4] = wy;
}
}  // namespace

// PoseGraph::optimize4DoF for the keyframes first_looped_index .. cur_index, given as arrays of n nodes in list order:
// t[n][3], R[n][9] = the VIO poses (getVioPose), sequence[n], loop_to[n] = local index of the loop partner or -1, loop_info[n][8].
// Node 0 (the earliest looped keyframe) and nodes of sequence 0 are constant.  Sequential edges to the 1..4 previous nodes of the same
// sequence, loop edges with HuberLoss(0.1); Ceres defaults otherwise: Levenberg-Marquardt, Jacobi scaling, 5 iterations.
// Output: optimised t / yaw per node (pitch, roll kept) as t_out[n][3], R_out[n][9]; drift = (yaw_drift, t_drift(3)) of the last node (:547-553).
void optimize_4dof(int n, const double *t_in, const double *R_in, const int *sequence, const int *loop_to, const double *loop_info,
                   double *t_out, double *R_out, double *drift4) {
    std::vector<double> yaw(n), pitch(n), roll(n), t(t_in, t_in + 3 * n);
    std::vector<uint8_t> fixed(n, 0);
    std::vector<Edge4> edges;
    for (int i = 0; i < n; i++) {
        M3 R;
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) R(a, b) = R_in[9 * i + 3 * a + b];
        // (the reference converts R -> quaternion -> R before R2ypr; the round trip is the identity up to round-off)
        const V3 e = R2ypr(toR(fromR(R)));
        yaw[i] = e.x; pitch[i] = e.y; roll[i] = e.z;
        fixed[i] = (i == 0 || sequence[i] == 0) ? 1 : 0;
    }
    for (int i = 0; i < n; i++) {
        for (int j = 1; j < 5; j++)
            if (i - j >= 0 && sequence[i] == sequence[i - j]) {
                M3 Rp;
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rp(a, b) = R_in[9 * (i - j) + 3 * a + b];
                Rp = toR(fromR(Rp));
                const V3 rel = T(Rp) * V3(t_in[3 * i] - t_in[3 * (i - j)], t_in[3 * i + 1] - t_in[3 * (i - j) + 1], t_in[3 * i + 2] - t_in[3 * (i - j) + 2]);
                Edge4 e{i - j, i, {rel.x, rel.y, rel.z}, yaw[i] - yaw[i - j], pitch[i - j], roll[i - j], false};
                edges.push_back(e);
            }
        if (loop_to[i] >= 0) {
            const int c = loop_to[i];
            Edge4 e{c, i, {loop_info[8 * i], loop_info[8 * i + 1], loop_info[8 * i + 2]}, loop_info[8 * i + 7], pitch[c], roll[c], true};
            edges.push_back(e);
        }
    }
    // variable map
    std::vector<int> col(n, -1);
    int nv = 0;
    for (int i = 0; i < n; i++) if (!fixed[i]) { col[i] = nv; nv += 4; }
    auto cost_of = [&](const std::vector<double> &yw, const std::vector<double> &tt) {
        double c = 0;
        for (const Edge4 &e : edges) {
            double r[4], J[32];
            edge_eval(e, yw.data(), tt.data(), r, J);
            const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
            c += 0.5 * (e.loop ? (s <= 0.01 ? s : 2 * 0.1 * std::sqrt(s) - 0.01) : s);
        }
        return c;
    };
    if (nv > 0) {
        double radius = 1e4, decrease = 2.0;
        double cost = cost_of(yaw, t);
        std::vector<double> scale;
        for (int it = 0; it < 5; it++) {
            Mat A(nv, nv);
            std::vector<double> g(nv, 0.0);
            for (const Edge4 &e : edges) {
                double r[4], J[32];
                edge_eval(e, yaw.data(), t.data(), r, J);
                if (e.loop) {   // Huber(0.1) through the corrector: rho'' <= 0 -> plain sqrt(rho') scaling
                    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
                    const double w = s <= 0.01 ? 1.0 : std::sqrt(0.1 / std::sqrt(s));
                    for (int a = 0; a < 4; a++) { r[a] *= w; for (int b = 0; b < 8; b++) J[a * 8 + b] *= w; }
                }
                const int base[2] = {col[e.i], col[e.j]};
                for (int p = 0; p < 8; p++) {
                    const int cp = base[p >> 2];
                    if (cp < 0) continue;
                    double gs = 0;
                    for (int a = 0; a < 4; a++) gs += J[a * 8 + p] * r[a];
                    g[cp + (p & 3)] += gs;
                    for (int q = 0; q < 8; q++) {
                        const int cq = base[q >> 2];
                        if (cq < 0) continue;
                        double s2 = 0;
                        for (int a = 0; a < 4; a++) s2 += J[a * 8 + p] * J[a * 8 + q];
                        A(cp + (p & 3), cq + (q & 3)) += s2;
                    }
                }
            }
            if (scale.empty()) { scale.resize(nv); for (int a = 0; a < nv; a++) scale[a] = 1.0 / (1.0 + std::sqrt(A(a, a))); }   // Jacobi scaling, fixed at the first point
            Mat As(nv, nv);
            std::vector<double> gs(nv);
            for (int a = 0; a < nv; a++) { gs[a] = scale[a] * g[a]; for (int b = 0; b < nv; b++) As(a, b) = scale[a] * scale[b] * A(a, b); }
            double gmax = 0;
            for (int a = 0; a < nv; a++) gmax = std::max(gmax, std::fabs(g[a]));
            if (gmax <= 1e-10) break;
            bool accepted = false;
            for (int tries = 0; tries < 20 && !accepted; tries++) {
                Mat M = As;
                for (int a = 0; a < nv; a++) M(a, a) += std::min(std::max(As(a, a), 1e-6), 1e32) / radius;
                std::vector<double> rhs = gs;
                if (!chol(M)) { radius /= decrease; decrease *= 2; continue; }
                chol_solve(M, rhs);
                std::vector<double> yc = yaw, tc = t;
                double model = 0;
                {   // model decrease = -(g^T d + 0.5 d^T A d), d = -rhs (scaled)
                    double lin = 0, quad = 0;
                    for (int a = 0; a < nv; a++) { lin += gs[a] * (-rhs[a]); double s2 = 0; for (int b = 0; b < nv; b++) s2 += As(a, b) * (-rhs[b]); quad += (-rhs[a]) * s2; }
                    model = -(lin + 0.5 * quad);
                }
                for (int i = 0; i < n; i++) {
                    if (col[i] < 0) continue;
                    yc[i] = normalize_angle_deg(yaw[i] - rhs[col[i]] * scale[col[i]]);   // AngleLocalParameterization
                    for (int a = 0; a < 3; a++) tc[3 * i + a] = t[3 * i + a] - rhs[col[i] + 1 + a] * scale[col[i] + 1 + a];
                }
                const double cc = cost_of(yc, tc);
                const double rho = model > 0 ? (cost - cc) / model : -1;
                if (rho > 1e-3) {
                    yaw = yc; t = tc;
                    const double rel = std::fabs(cost - cc) / cost;
                    cost = cc;
                    radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2 * rho - 1, 3));
                    radius = std::min(radius, 1e16);
                    decrease = 2.0;
                    accepted = true;
                    if (rel < 1e-6) it = 5;   // function tolerance
                } else {
                    radius /= decrease; decrease *= 2;
                    it++;                      // an unsuccessful step counts as an iteration
                    if (it >= 5) break;
                }
            }
            if (!accepted) break;
        }
    }
    for (int i = 0; i < n; i++) {
        const M3 R = ypr2R(V3(yaw[i], pitch[i], roll[i]));
        for (int a = 0; a < 3; a++) { t_out[3 * i + a] = t[3 * i + a]; for (int b = 0; b < 3; b++) R_out[9 * i + 3 * a + b] = R(a, b); }
    }
    {   // drift of the newest keyframe (:547-553)
        M3 Rv;
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rv(a, b) = R_in[9 * (n - 1) + 3 * a + b];
        const M3 Rc = ypr2R(V3(yaw[n - 1], pitch[n - 1], roll[n - 1]));
        const double yd = R2ypr(Rc).x - R2ypr(Rv).x;
        const M3 rd = ypr2R(V3(yd, 0, 0));
        const V3 td = V3(t[3 * (n - 1)], t[3 * (n - 1) + 1], t[3 * (n - 1) + 2]) - rd * V3(t_in[3 * (n - 1)], t_in[3 * (n - 1) + 1], t_in[3 * (n - 1) + 2]);
        drift4[0] = yd; drift4[1] = td.x; drift4[2] = td.y; drift4[3] = td.z;
    }
}

// ---------------------------------------------------------------- PoseGraph::optimize6DoF (pose_graph.cpp:583-740; RelativeRTError pose_graph.h:256-320)
// The `imu: 0` variant: every node carries a full pose (quaternion with ceres::QuaternionParameterization + translation); sequential edges to
// the 1..4 previous nodes of the same sequence and loop edges (HuberLoss(0.1)) are RelativeRTError(relative t, relative q, t_var 0.1, q_var
// 0.01): r = ((R_i^T (t_j - t_i) - t_m) / t_var, 2 vec(q_m^-1 (q_i^-1 q_j)) / q_var).  Node 0 and the nodes of sequence 0 are constant.
// QuaternionParameterization::Plus(q, d) = [cos|d|, sin|d| d / |d|] * q, so the tangent is a left HALF-angle perturbation; the Jacobians
// below are the directional derivatives along it (what autodiff times the Plus-Jacobian gives).  Same trust-region loop as optimize_4dof.
// Output: t_out[n][3], R_out[n][9]; drift12 = r_drift (9, row-major) = R_cur R_vio^T and t_drift = t_cur - r_drift t_vio (:717-721).
namespace {
struct Edge6 { int i, j; V3 tm; Q qm; bool loop; };
// residual (6) and Jacobians wrt (dtheta_i, t_i, dtheta_j, t_j): J[6][12]
void edge6_eval(const Edge6 &e, const std::vector<Q> &q, const std::vector<double> &t, double *r, double *J) {
    const Q qi = q[e.i], qj = q[e.j];
    const M3 RiT = T(toR(normalized(qi)));            // ceres::QuaternionRotatePoint scales the quaternion to unit length
    const V3 d(t[3 * e.j] - t[3 * e.i], t[3 * e.j + 1] - t[3 * e.i + 1], t[3 * e.j + 2] - t[3 * e.i + 2]);
    const V3 tij = RiT * d;
    const double tv = 0.1, qv = 0.01;
    r[0] = (tij.x - e.tm.x) / tv; r[1] = (tij.y - e.tm.y) / tv; r[2] = (tij.z - e.tm.z) / tv;
    const Q qic(qi.w, -qi.x, -qi.y, -qi.z), qmc(e.qm.w, -e.qm.x, -e.qm.y, -e.qm.z);   // QuaternionInverse of the functor = conjugate
    const Q A = qmc * qic;
    const Q err = A * qj;
    r[3] = 2.0 * err.x / qv; r[4] = 2.0 * err.y / qv; r[5] = 2.0 * err.z / qv;
    if (!J) return;
    for (int k = 0; k < 72; k++) J[k] = 0.0;
    // d r_t / d dtheta_i = R_i^T 2 [d]x ; / d t_i = -R_i^T ; / d t_j = R_i^T
    const M3 dx = skew(d);
    const M3 Jr = RiT * dx;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            J[a * 12 + b] = 2.0 * Jr(a, b) / tv;
            J[a * 12 + 3 + b] = -RiT(a, b) / tv;
            J[a * 12 + 9 + b] = RiT(a, b) / tv;
        }
    // d err = A * [0, dtheta_j - dtheta_i] * q_j
    for (int b = 0; b < 3; b++) {
        const Q unit(0.0, b == 0 ? 1.0 : 0.0, b == 1 ? 1.0 : 0.0, b == 2 ? 1.0 : 0.0);
        const Q de = (A * unit) * qj;
        const double v[3] = {2.0 * de.x / qv, 2.0 * de.y / qv, 2.0 * de.z / qv};
        for (int a = 0; a < 3; a++) { J[(3 + a) * 12 + 6 + b] = v[a]; J[(3 + a) * 12 + b] = -v[a]; }
    }
}
Q quat_plus(const Q &q, const double *d) {   // ceres::QuaternionParameterization::Plus
    const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0.0) {
        const double s = std::sin(n) / n;
        return Q(std::cos(n), s * d[0], s * d[1], s * d[2]) * q;
    }
    return q;
}
}  // namespace

void optimize_6dof(int n, const double *t_in, const double *R_in, const int *sequence, const int *loop_to, const double *loop_info,
                   double *t_out, double *R_out, double *drift12) {
    std::vector<Q> q(n);
    std::vector<double> t(t_in, t_in + 3 * n);
    std::vector<uint8_t> fixed(n, 0);
    for (int i = 0; i < n; i++) {
        M3 R;
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) R(a, b) = R_in[9 * i + 3 * a + b];
        q[i] = fromR(R);
        fixed[i] = (i == 0 || sequence[i] == 0) ? 1 : 0;
    }
    std::vector<Edge6> edges;
    for (int i = 0; i < n; i++) {
        for (int j = 1; j < 5; j++)
            if (i - j >= 0 && sequence[i] == sequence[i - j]) {
                const Q qa = q[i - j];
                const V3 rel = toR(inverse(qa)) * V3(t_in[3 * i] - t_in[3 * (i - j)], t_in[3 * i + 1] - t_in[3 * (i - j) + 1], t_in[3 * i + 2] - t_in[3 * (i - j) + 2]);
                edges.push_back(Edge6{i - j, i, rel, inverse(qa) * q[i], false});
            }
        if (loop_to[i] >= 0)
            edges.push_back(Edge6{loop_to[i], i, V3(loop_info[8 * i], loop_info[8 * i + 1], loop_info[8 * i + 2]),
                                  Q(loop_info[8 * i + 3], loop_info[8 * i + 4], loop_info[8 * i + 5], loop_info[8 * i + 6]), true});
    }
    std::vector<int> col(n, -1);
    int nv = 0;
    for (int i = 0; i < n; i++) if (!fixed[i]) { col[i] = nv; nv += 6; }
    auto cost_of = [&](const std::vector<Q> &qq, const std::vector<double> &tt) {
        double c = 0;
        for (const Edge6 &e : edges) {
            double r[6];
            edge6_eval(e, qq, tt, r, nullptr);
            double s = 0;
            for (int a = 0; a < 6; a++) s += r[a] * r[a];
            c += 0.5 * (e.loop ? (s <= 0.01 ? s : 2 * 0.1 * std::sqrt(s) - 0.01) : s);
        }
        return c;
    };
    if (nv > 0) {
        double radius = 1e4, decrease = 2.0;
        double cost = cost_of(q, t);
        std::vector<double> scale;
        for (int it = 0; it < 5; it++) {
            Mat A(nv, nv);
            std::vector<double> g(nv, 0.0);
            for (const Edge6 &e : edges) {
                double r[6], J[72];
                edge6_eval(e, q, t, r, J);
                if (e.loop) {
                    double s = 0;
                    for (int a = 0; a < 6; a++) s += r[a] * r[a];
                    const double w = s <= 0.01 ? 1.0 : std::sqrt(0.1 / std::sqrt(s));
                    for (int a = 0; a < 6; a++) { r[a] *= w; for (int b = 0; b < 12; b++) J[a * 12 + b] *= w; }
                }
                const int base[2] = {col[e.i], col[e.j]};
                for (int p = 0; p < 12; p++) {
                    const int cp = base[p / 6];
                    if (cp < 0) continue;
                    double gs = 0;
                    for (int a = 0; a < 6; a++) gs += J[a * 12 + p] * r[a];
                    g[cp + p % 6] += gs;
                    for (int qq = 0; qq < 12; qq++) {
                        const int cq = base[qq / 6];
                        if (cq < 0) continue;
                        double s2 = 0;
                        for (int a = 0; a < 6; a++) s2 += J[a * 12 + p] * J[a * 12 + qq];
                        A(cp + p % 6, cq + qq % 6) += s2;
                    }
                }
            }
            if (scale.empty()) { scale.resize(nv); for (int a = 0; a < nv; a++) scale[a] = 1.0 / (1.0 + std::sqrt(A(a, a))); }
            Mat As(nv, nv);
            std::vector<double> gs(nv);
            for (int a = 0; a < nv; a++) { gs[a] = scale[a] * g[a]; for (int b = 0; b < nv; b++) As(a, b) = scale[a] * scale[b] * A(a, b); }
            double gmax = 0;
            for (int a = 0; a < nv; a++) gmax = std::max(gmax, std::fabs(g[a]));
            if (gmax <= 1e-10) break;
            bool accepted = false;
            for (int tries = 0; tries < 20 && !accepted; tries++) {
                Mat M = As;
                for (int a = 0; a < nv; a++) M(a, a) += std::min(std::max(As(a, a), 1e-6), 1e32) / radius;
                std::vector<double> rhs = gs;
                if (!chol(M)) { radius /= decrease; decrease *= 2; continue; }
                chol_solve(M, rhs);
                double lin = 0, quad = 0;
                for (int a = 0; a < nv; a++) { lin += gs[a] * (-rhs[a]); double s2 = 0; for (int b = 0; b < nv; b++) s2 += As(a, b) * (-rhs[b]); quad += (-rhs[a]) * s2; }
                const double model = -(lin + 0.5 * quad);
                std::vector<Q> qc = q;
                std::vector<double> tc = t;
                for (int i = 0; i < n; i++) {
                    if (col[i] < 0) continue;
                    double d[3];
                    for (int a = 0; a < 3; a++) d[a] = -rhs[col[i] + a] * scale[col[i] + a];
                    qc[i] = quat_plus(q[i], d);
                    for (int a = 0; a < 3; a++) tc[3 * i + a] = t[3 * i + a] - rhs[col[i] + 3 + a] * scale[col[i] + 3 + a];
                }
                const double cc = cost_of(qc, tc);
                const double rho = model > 0 ? (cost - cc) / model : -1;
                if (rho > 1e-3) {
                    q = qc; t = tc;
                    const double rel = std::fabs(cost - cc) / cost;
                    cost = cc;
                    radius = std::min(radius / std::max(1.0 / 3.0, 1.0 - std::pow(2 * rho - 1, 3)), 1e16);
                    decrease = 2.0;
                    accepted = true;
                    if (rel < 1e-6) it = 5;
                } else {
                    radius /= decrease; decrease *= 2;
                    it++;
                    if (it >= 5) break;
                }
            }
            if (!accepted) break;
        }
    }
    for (int i = 0; i < n; i++) {
        const M3 R = toR(q[i]);
        for (int a = 0; a < 3; a++) { t_out[3 * i + a] = t[3 * i + a]; for (int b = 0; b < 3; b++) R_out[9 * i + 3 * a + b] = R(a, b); }
    }
    {
        M3 Rv;
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rv(a, b) = R_in[9 * (n - 1) + 3 * a + b];
        const M3 rd = toR(q[n - 1]) * T(Rv);
        const V3 td = V3(t[3 * (n - 1)], t[3 * (n - 1) + 1], t[3 * (n - 1) + 2]) - rd * V3(t_in[3 * (n - 1)], t_in[3 * (n - 1) + 1], t_in[3 * (n - 1) + 2]);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) drift12[3 * a + b] = rd(a, b);
        drift12[9] = td.x; drift12[10] = td.y; drift12[11] = td.z;
    }
}

}  // namespace ovio
