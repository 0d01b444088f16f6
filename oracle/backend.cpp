// ORACLE (test infrastructure only — never linked into or called by the product path).
// CPU restatement of the sliding-window back-end:
//   vins_estimator/src/estimator/estimator.cpp:15-374,922-1716,1749-2009
//   vins_estimator/src/feature_manager/feature_manager.cpp:31-123,197-233,302-324,386-543,660-768
//   vins_estimator/src/factor/{integration_base.h,imu_factor.h,projection_factor.cpp,projection_td_factor.cpp,
//                              pose_local_parameterization.cpp,marginalization_factor.cpp}
//   vins_estimator/src/initial/initial_aligment.cpp:3-36
// The Ceres trust-region solve (estimator.cpp:1348-1363: DENSE_SCHUR + traditional DOGLEG, 8 iterations) is
// restated from SURVEY.md Appendix B.5. "parity unpinned": Ceres/Eigen are absent from this image.
#include <array>
#include "oracle.h"
#include <cfloat>
#include <numeric>
#include <cstdio>
#include <cstdlib>

namespace ovio {
using namespace om;

enum { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };
int oracle_deviations = 0;   // attribution experiment (oracle.h ODEV_*): 0 everywhere except tests/oracle_control.py's variants

// ------------------------------------------------------------------ IntegrationBase (integration_base.h)
Integration::Integration(const Config &c, const V3 &a0, const V3 &g0, const V3 &ba, const V3 &bg)
    : acc_n(c.acc_n), acc_w(c.acc_w), gyr_n(c.gyr_n), gyr_w(c.gyr_w), acc_0(a0), gyr_0(g0), linearized_acc(a0),
      linearized_gyr(g0), linearized_ba(ba), linearized_bg(bg) {
    std::memset(jacobian, 0, sizeof(jacobian));
    std::memset(covariance, 0, sizeof(covariance));
    for (int i = 0; i < 15; i++) jacobian[i][i] = 1;
}
void Integration::push_back(double dt, const V3 &acc, const V3 &gyr) {  // :32-38
    dt_buf.push_back(dt);
    acc_buf.push_back(acc);
    gyr_buf.push_back(gyr);
    propagate(dt, acc, gyr);
}
void Integration::repropagate(const V3 &ba, const V3 &bg) {  // :40-54
    sum_dt = 0;
    acc_0 = linearized_acc;
    gyr_0 = linearized_gyr;
    delta_p = V3();
    delta_q = Q();
    delta_v = V3();
    linearized_ba = ba;
    linearized_bg = bg;
    std::memset(jacobian, 0, sizeof(jacobian));
    std::memset(covariance, 0, sizeof(covariance));
    for (int i = 0; i < 15; i++) jacobian[i][i] = 1;
    for (size_t i = 0; i < dt_buf.size(); i++) propagate(dt_buf[i], acc_buf[i], gyr_buf[i]);
}
static void setblk(double M[15][18], int r, int c, const M3 &B) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[r + i][c + j] = B(i, j);
}
static void setblk15(double M[15][15], int r, int c, const M3 &B) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[r + i][c + j] = B(i, j);
}
void Integration::propagate(double dt, const V3 &acc_1, const V3 &gyr_1) {  // :56-162 (midPointIntegration)
    V3 un_acc_0 = rot(delta_q, acc_0 - linearized_ba);
    V3 un_gyr = 0.5 * (gyr_0 + gyr_1) - linearized_bg;
    Q rq = delta_q * Q(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2);
    V3 un_acc_1 = rot(rq, acc_1 - linearized_ba);
    V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
    V3 rp = delta_p + delta_v * dt + 0.5 * un_acc * dt * dt;
    V3 rv = delta_v + un_acc * dt;

    V3 w_x = un_gyr, a_0_x = acc_0 - linearized_ba, a_1_x = acc_1 - linearized_ba;
    M3 R_w_x = skew(w_x), R_a_0_x = skew(a_0_x), R_a_1_x = skew(a_1_x);
    M3 Rq = toR(delta_q), Rr = toR(rq), I = M3::I();
    M3 ImW = I - dt * R_w_x;
    double F[15][15], V[15][18];
    std::memset(F, 0, sizeof(F));
    std::memset(V, 0, sizeof(V));
    setblk15(F, 0, 0, I);
    setblk15(F, 0, 3, (-0.25 * dt * dt) * (Rq * R_a_0_x) + (-0.25 * dt * dt) * (Rr * R_a_1_x * ImW));
    setblk15(F, 0, 6, dt * I);
    setblk15(F, 0, 9, (-0.25 * dt * dt) * (Rq + Rr));
    setblk15(F, 0, 12, (-0.25 * dt * dt * -dt) * (Rr * R_a_1_x));
    setblk15(F, 3, 3, ImW);
    setblk15(F, 3, 12, (-dt) * I);
    setblk15(F, 6, 3, (-0.5 * dt) * (Rq * R_a_0_x) + (-0.5 * dt) * (Rr * R_a_1_x * ImW));
    setblk15(F, 6, 6, I);
    setblk15(F, 6, 9, (-0.5 * dt) * (Rq + Rr));
    setblk15(F, 6, 12, (-0.5 * dt * -dt) * (Rr * R_a_1_x));
    setblk15(F, 9, 9, I);
    setblk15(F, 12, 12, I);
    M3 V03 = (0.25 * dt * dt * 0.5 * dt) * (-(Rr * R_a_1_x));
    M3 V63 = (0.5 * dt * 0.5 * dt) * (-(Rr * R_a_1_x));
    setblk(V, 0, 0, (0.25 * dt * dt) * Rq);
    setblk(V, 0, 3, V03);
    setblk(V, 0, 6, (0.25 * dt * dt) * Rr);
    setblk(V, 0, 9, V03);
    setblk(V, 3, 3, (0.5 * dt) * I);
    setblk(V, 3, 9, (0.5 * dt) * I);
    setblk(V, 6, 0, (0.5 * dt) * Rq);
    setblk(V, 6, 3, V63);
    setblk(V, 6, 6, (0.5 * dt) * Rr);
    setblk(V, 6, 9, V63);
    setblk(V, 9, 12, dt * I);
    setblk(V, 12, 15, dt * I);
    double noise[18];
    for (int i = 0; i < 3; i++) {
        noise[i] = acc_n * acc_n; noise[3 + i] = gyr_n * gyr_n; noise[6 + i] = acc_n * acc_n;
        noise[9 + i] = gyr_n * gyr_n; noise[12 + i] = acc_w * acc_w; noise[15 + i] = gyr_w * gyr_w;
    }
    double FJ[15][15], FP[15][15], NP[15][15];
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            double s = 0, s2 = 0;
            for (int k = 0; k < 15; k++) { s += F[i][k] * jacobian[k][j]; s2 += F[i][k] * covariance[k][j]; }
            FJ[i][j] = s; FP[i][j] = s2;
        }
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            double s = 0;
            for (int k = 0; k < 15; k++) s += FP[i][k] * F[j][k];
            double t = 0;
            for (int k = 0; k < 18; k++) t += V[i][k] * noise[k] * V[j][k];
            NP[i][j] = s + t;
        }
    std::memcpy(jacobian, FJ, sizeof(FJ));
    std::memcpy(covariance, NP, sizeof(NP));
    delta_p = rp;
    delta_q = normalized(rq);
    delta_v = rv;
    sum_dt += dt;
    acc_0 = acc_1;
    gyr_0 = gyr_1;
}
static M3 blk(const double J[15][15], int r, int c) {
    M3 B;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B(i, j) = J[r + i][c + j];
    return B;
}
void Integration::evaluate(const V3 &G, const V3 &Pi, const Q &Qi, const V3 &Vi, const V3 &Bai, const V3 &Bgi,
                           const V3 &Pj, const Q &Qj, const V3 &Vj, const V3 &Baj, const V3 &Bgj, double r[15]) const {
    // :164-195
    M3 dp_dba = blk(jacobian, O_P, O_BA), dp_dbg = blk(jacobian, O_P, O_BG), dq_dbg = blk(jacobian, O_R, O_BG),
       dv_dba = blk(jacobian, O_V, O_BA), dv_dbg = blk(jacobian, O_V, O_BG);
    V3 dba = Bai - linearized_ba, dbg = Bgi - linearized_bg;
    Q cq = delta_q * deltaQ(dq_dbg * dbg);
    V3 cv = delta_v + dv_dba * dba + dv_dbg * dbg;
    V3 cp = delta_p + dp_dba * dba + dp_dbg * dbg;
    Q Qi_inv = inverse(Qi);
    V3 rp = rot(Qi_inv, 0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt) - cp;
    V3 rq = 2.0 * (inverse(cq) * (Qi_inv * Qj)).vec();
    V3 rv = rot(Qi_inv, G * sum_dt + Vj - Vi) - cv;
    V3 rba = Baj - Bai, rbg = Bgj - Bgi;
    for (int i = 0; i < 3; i++) { r[O_P + i] = rp[i]; r[O_R + i] = rq[i]; r[O_V + i] = rv[i]; r[O_BA + i] = rba[i]; r[O_BG + i] = rbg[i]; }
}

static void Qleft(const Q &q, double M[4][4]) {  // utility.h:46-54
    M[0][0] = q.w; M[0][1] = -q.x; M[0][2] = -q.y; M[0][3] = -q.z;
    M3 S = skew(q.vec());
    double v[3] = {q.x, q.y, q.z};
    for (int i = 0; i < 3; i++) {
        M[1 + i][0] = v[i];
        for (int j = 0; j < 3; j++) M[1 + i][1 + j] = (i == j ? q.w : 0.0) + S(i, j);
    }
}
static void Qright(const Q &q, double M[4][4]) {  // utility.h:56-64
    M[0][0] = q.w; M[0][1] = -q.x; M[0][2] = -q.y; M[0][3] = -q.z;
    M3 S = skew(q.vec());
    double v[3] = {q.x, q.y, q.z};
    for (int i = 0; i < 3; i++) {
        M[1 + i][0] = v[i];
        for (int j = 0; j < 3; j++) M[1 + i][1 + j] = (i == j ? q.w : 0.0) - S(i, j);
    }
}
static M3 br33(const double M[4][4]) { M3 B; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B(i, j) = M[1 + i][1 + j]; return B; }

// IMUFactor::Evaluate imu_factor.h:20-205
void eval_imu(const Integration &pre, const V3 &G, const double *pi, const double *sbi, const double *pj, const double *sbj,
              double r[15], double *J_pi, double *J_sbi, double *J_pj, double *J_sbj) {
    V3 Pi(pi[0], pi[1], pi[2]); Q Qi(pi[6], pi[3], pi[4], pi[5]);
    V3 Vi(sbi[0], sbi[1], sbi[2]), Bai(sbi[3], sbi[4], sbi[5]), Bgi(sbi[6], sbi[7], sbi[8]);
    V3 Pj(pj[0], pj[1], pj[2]); Q Qj(pj[6], pj[3], pj[4], pj[5]);
    V3 Vj(sbj[0], sbj[1], sbj[2]), Baj(sbj[3], sbj[4], sbj[5]), Bgj(sbj[6], sbj[7], sbj[8]);
    double raw[15];
    pre.evaluate(G, Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, raw);
    // sqrt_info = LLT(cov^-1).matrixL().transpose()
    Mat C(15, 15);
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) C(i, j) = 0.5 * (pre.covariance[i][j] + pre.covariance[j][i]);
    Mat Lc = C;
    double sqrt_info[15][15];
    std::memset(sqrt_info, 0, sizeof(sqrt_info));
    if (chol(Lc)) {
        // cov^-1 = Lc^-T Lc^-1
        Mat Li(15, 15);
        for (int c = 0; c < 15; c++) {
            std::vector<double> e(15, 0.0);
            e[c] = 1;
            for (int i = 0; i < 15; i++) {
                double s = e[i];
                for (int k = 0; k < i; k++) s -= Lc(i, k) * e[k];
                e[i] = s / Lc(i, i);
            }
            for (int i = 0; i < 15; i++) Li(i, c) = e[i];
        }
        Mat Ci(15, 15);
        for (int i = 0; i < 15; i++)
            for (int j = 0; j < 15; j++) {
                double s = 0;
                for (int k = 0; k < 15; k++) s += Li(k, i) * Li(k, j);
                Ci(i, j) = s;
            }
        if (oracle_deviations & ODEV_IMU_WHITEN) {
            // deviation 8 (attribution experiment): M = chol(cov)^-1, lower triangular, M^T M = cov^-1 as well
            for (int i = 0; i < 15; i++) for (int j = 0; j <= i; j++) sqrt_info[i][j] = Li(i, j);
        } else if (chol(Ci))
            for (int i = 0; i < 15; i++) for (int j = i; j < 15; j++) sqrt_info[i][j] = Ci(j, i);  // L^T
    }
    for (int i = 0; i < 15; i++) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += sqrt_info[i][k] * raw[k];
        r[i] = s;
    }
    if (!J_pi) return;
    double sum_dt = pre.sum_dt;
    M3 dp_dba = blk(pre.jacobian, O_P, O_BA), dp_dbg = blk(pre.jacobian, O_P, O_BG), dq_dbg = blk(pre.jacobian, O_R, O_BG),
       dv_dba = blk(pre.jacobian, O_V, O_BA), dv_dbg = blk(pre.jacobian, O_V, O_BG);
    Q Qi_inv = inverse(Qi), Qj_inv = inverse(Qj);
    M3 RiT = toR(Qi_inv);
    Q cq = pre.delta_q * deltaQ(dq_dbg * (Bgi - pre.linearized_bg));
    double Ji[15][7], Jsi[15][9], Jj[15][7], Jsj[15][9];
    std::memset(Ji, 0, sizeof(Ji)); std::memset(Jsi, 0, sizeof(Jsi)); std::memset(Jj, 0, sizeof(Jj)); std::memset(Jsj, 0, sizeof(Jsj));
    auto put = [](double *base, int ld, int r0, int c0, const M3 &B) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) base[(r0 + i) * ld + c0 + j] = B(i, j);
    };
    double L4[4][4], R4[4][4];
    // pose_i
    put(&Ji[0][0], 7, O_P, O_P, -RiT);
    put(&Ji[0][0], 7, O_P, O_R, skew(rot(Qi_inv, 0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt)));
    Qleft(Qj_inv * Qi, L4); Qright(cq, R4);
    {
        double LR[4][4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += L4[i][k] * R4[k][j]; LR[i][j] = s; }
        put(&Ji[0][0], 7, O_R, O_R, -br33(LR));
    }
    put(&Ji[0][0], 7, O_V, O_R, skew(rot(Qi_inv, G * sum_dt + Vj - Vi)));
    // speedbias_i
    put(&Jsi[0][0], 9, O_P, O_V - O_V, (-sum_dt) * RiT);
    put(&Jsi[0][0], 9, O_P, O_BA - O_V, -dp_dba);
    put(&Jsi[0][0], 9, O_P, O_BG - O_V, -dp_dbg);
    Qleft(Qj_inv * Qi * pre.delta_q, L4);
    put(&Jsi[0][0], 9, O_R, O_BG - O_V, -(br33(L4) * dq_dbg));
    put(&Jsi[0][0], 9, O_V, O_V - O_V, -RiT);
    put(&Jsi[0][0], 9, O_V, O_BA - O_V, -dv_dba);
    put(&Jsi[0][0], 9, O_V, O_BG - O_V, -dv_dbg);
    put(&Jsi[0][0], 9, O_BA, O_BA - O_V, -M3::I());
    put(&Jsi[0][0], 9, O_BG, O_BG - O_V, -M3::I());
    // pose_j
    put(&Jj[0][0], 7, O_P, O_P, RiT);
    Qleft(inverse(cq) * Qi_inv * Qj, L4);
    put(&Jj[0][0], 7, O_R, O_R, br33(L4));
    // speedbias_j
    put(&Jsj[0][0], 9, O_V, O_V - O_V, RiT);
    put(&Jsj[0][0], 9, O_BA, O_BA - O_V, M3::I());
    put(&Jsj[0][0], 9, O_BG, O_BG - O_V, M3::I());
    auto premul = [&](const double *Jin, int nc, double *out) {
        for (int i = 0; i < 15; i++)
            for (int j = 0; j < nc; j++) {
                double s = 0;
                const int k0 = (oracle_deviations & ODEV_IMU_WHITEN) ? 0 : i, k1 = (oracle_deviations & ODEV_IMU_WHITEN) ? i + 1 : 15;
                for (int k = k0; k < k1; k++) s += sqrt_info[i][k] * Jin[k * nc + j];
                out[i * nc + j] = s;
            }
    };
    premul(&Ji[0][0], 7, J_pi);
    premul(&Jsi[0][0], 9, J_sbi);
    premul(&Jj[0][0], 7, J_pj);
    premul(&Jsj[0][0], 9, J_sbj);
}

// ProjectionFactor / ProjectionTdFactor  (projection_factor.cpp:22-130, projection_td_factor.cpp:34-150)
void eval_projection(const Config &c, const double *pi, const double *pj, const double *ex, double inv_dep, double td,
                     const Obs &oi, const Obs &oj, bool use_td, double r[2], double *J_i, double *J_j, double *J_ex,
                     double *J_l, double *J_td) {
    V3 Pi(pi[0], pi[1], pi[2]); Q Qi(pi[6], pi[3], pi[4], pi[5]);
    V3 Pj(pj[0], pj[1], pj[2]); Q Qj(pj[6], pj[3], pj[4], pj[5]);
    V3 tic(ex[0], ex[1], ex[2]); Q qic(ex[6], ex[3], ex[4], ex[5]);
    V3 pts_i(oi.x, oi.y, oi.z), pts_j(oj.x, oj.y, oj.z);
    V3 vel_i(oi.vx, oi.vy, 0), vel_j(oj.vx, oj.vy, 0);
    if (use_td) {
        double ROW = (double)c.height;
        double row_i = oi.v - ROW / 2, row_j = oj.v - ROW / 2;
        pts_i = pts_i - (td - oi.cur_td + c.tr / ROW * row_i) * vel_i;
        pts_j = pts_j - (td - oj.cur_td + c.tr / ROW * row_j) * vel_j;
    }
    double sq = c.focal_length / 1.5;  // sqrt_info = FOCAL_LENGTH/1.5 * I (estimator.cpp:23-24)
    if (oracle_deviations & ODEV_PAIR_PROJECTION) {
        // deviation 11 (attribution experiment): the same residual and Jacobians through the matrices of the frame pair,
        // A1 = ric^T Rj^T, A2 = A1 Ri, M = A2 ric, t = A1 (Ri tic + Pi - Pj) - ric^T tic, so that pts_camera_j = M pts_camera_i + t
        const M3 Ri = toR(Qi), Rj = toR(Qj), ric = toR(qic), ricT = T(ric);
        const M3 A1 = ricT * T(Rj), A2 = A1 * Ri, M = A2 * ric;
        const V3 t = A1 * (Ri * tic + (Pi - Pj)) - ricT * tic;
        const V3 pc_i = pts_i / inv_dep;
        const V3 pc_j = M * pc_i + t;
        const double dep = pc_j.z;
        r[0] = sq * (pc_j.x / dep - pts_j.x);
        r[1] = sq * (pc_j.y / dep - pts_j.y);
        if (!J_i) return;
        const V3 red[2] = {V3(sq / dep, 0, -sq * pc_j.x / (dep * dep)), V3(0, sq / dep, -sq * pc_j.y / (dep * dep))};
        const V3 pim_i = ric * pc_i + tic, pim_j = ric * pc_j + tic;
        auto rowmul = [](const V3 &rw, const M3 &A) { return V3(rw.x * A(0, 0) + rw.y * A(1, 0) + rw.z * A(2, 0), rw.x * A(0, 1) + rw.y * A(1, 1) + rw.z * A(2, 1),
                                                                rw.x * A(0, 2) + rw.y * A(1, 2) + rw.z * A(2, 2)); };
        auto rowskew = [](const V3 &rw, const V3 &w) { return V3(rw.y * w.z - rw.z * w.y, rw.z * w.x - rw.x * w.z, rw.x * w.y - rw.y * w.x); };   // rw * skew(w)
        auto put3 = [](double *o, const V3 &v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; };
        for (int a = 0; a < 2; a++) {
            const V3 ra1 = rowmul(red[a], A1), ra2 = rowmul(red[a], A2), rrt = rowmul(red[a], ricT), rm = rowmul(red[a], M);
            put3(J_i + a * 7, ra1); put3(J_i + a * 7 + 3, rowskew(ra2, -pim_i)); J_i[a * 7 + 6] = 0;
            put3(J_j + a * 7, -ra1); put3(J_j + a * 7 + 3, rowskew(rrt, pim_j)); J_j[a * 7 + 6] = 0;
            put3(J_ex + a * 7, ra2 - rrt); put3(J_ex + a * 7 + 3, rowskew(red[a], pc_j) - rowskew(rm, pc_i)); J_ex[a * 7 + 6] = 0;
            const V3 v = (M * pts_i) * (-1.0 / (inv_dep * inv_dep));
            J_l[a] = red[a].x * v.x + red[a].y * v.y + red[a].z * v.z;
            if (J_td) {
                if (use_td) {
                    const V3 w = (M * vel_i) / inv_dep * -1.0;
                    J_td[a] = (red[a].x * w.x + red[a].y * w.y + red[a].z * w.z) + sq * (a == 0 ? vel_j.x : vel_j.y);
                } else
                    J_td[a] = 0;
            }
        }
        return;
    }
    V3 pts_camera_i = pts_i / inv_dep;
    V3 pts_imu_i = rot(qic, pts_camera_i) + tic;
    V3 pts_w = rot(Qi, pts_imu_i) + Pi;
    V3 pts_imu_j = rot(inverse(Qj), pts_w - Pj);
    V3 pts_camera_j = rot(inverse(qic), pts_imu_j - tic);
    double dep_j = pts_camera_j.z;
    r[0] = sq * (pts_camera_j.x / dep_j - pts_j.x);
    r[1] = sq * (pts_camera_j.y / dep_j - pts_j.y);
    if (!J_i) return;
    M3 Ri = toR(Qi), Rj = toR(Qj), ric = toR(qic);
    double red[2][3] = {{sq / dep_j, 0, -sq * pts_camera_j.x / (dep_j * dep_j)},
                        {0, sq / dep_j, -sq * pts_camera_j.y / (dep_j * dep_j)}};
    auto red_mul = [&](const M3 &A, const M3 &B, double *out) {  // out(2×7) = red * [A | B], last col 0
        for (int i = 0; i < 2; i++) {
            for (int j = 0; j < 3; j++) {
                double s = 0, t = 0;
                for (int k = 0; k < 3; k++) { s += red[i][k] * A(k, j); t += red[i][k] * B(k, j); }
                out[i * 7 + j] = s;
                out[i * 7 + 3 + j] = t;
            }
            out[i * 7 + 6] = 0;
        }
    };
    M3 ricT = T(ric), RjT = T(Rj);
    red_mul(ricT * RjT, ricT * RjT * Ri * (-skew(pts_imu_i)), J_i);
    red_mul(ricT * (-RjT), ricT * skew(pts_imu_j), J_j);
    {
        M3 tmp_r = ricT * RjT * Ri * ric;
        M3 A = ricT * (RjT * Ri - M3::I());
        M3 B = -(tmp_r * skew(pts_camera_i)) + skew(tmp_r * pts_camera_i) +
               skew(ricT * (RjT * (Ri * tic + Pi - Pj) - tic));
        red_mul(A, B, J_ex);
    }
    {
        M3 tmp_r = ricT * RjT * Ri * ric;
        V3 v = tmp_r * pts_i * (-1.0 / (inv_dep * inv_dep));
        for (int i = 0; i < 2; i++) J_l[i] = red[i][0] * v.x + red[i][1] * v.y + red[i][2] * v.z;
        if (J_td) {
            if (use_td) {
                V3 w = tmp_r * vel_i / inv_dep * -1.0;
                for (int i = 0; i < 2; i++) J_td[i] = red[i][0] * w.x + red[i][1] * w.y + red[i][2] * w.z;
                J_td[0] += sq * vel_j.x;
                J_td[1] += sq * vel_j.y;
            } else
                J_td[0] = J_td[1] = 0;
        }
    }
}

// ------------------------------------------------------------------ Estimator
Estimator::Estimator(const Config &c) : cfg(c), W(c.window_size) {
    for (int i = 0; i <= MAXW; i++) pre_integrations[i] = nullptr;
    {
        const char *dv = std::getenv("OVIO_DEVIATIONS");
        deviations = dv ? std::atoi(dv) : 0;
        oracle_deviations = deviations;
    }
    clearState();
    // readParameters() re-orthonormalises the extrinsic rotation through a normalised quaternion (parameters.cpp:202-209)
    {
        M3 Rc;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rc(i, j) = c.ric[i * 3 + j];
        Rc = toR(normalized(fromR(Rc)));
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cfg.ric[i * 3 + j] = Rc(i, j);
    }
    // setParameter() estimator.cpp:15-41
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) ric(i, j) = cfg.ric[i * 3 + j];
    tic = V3(c.tic[0], c.tic[1], c.tic[2]);
    td = c.td;
    g = V3(0, 0, c.g_norm);
}
Estimator::~Estimator() {
    for (int i = 0; i <= MAXW; i++) delete pre_integrations[i];
    for (auto &it : all_image_frame) delete it.second.pre_integration;
    delete tmp_pre_integration;
}
void Estimator::clearState() {  // estimator.cpp:43-116
    imu_buf.clear();
    imu_head = 0;
    imu_at_update = 0;
    for (int i = 0; i <= MAXW; i++) {
        Rs[i] = M3::I();
        Ps[i] = Vs[i] = Bas[i] = Bgs[i] = V3();
        Headers[i] = 0;
        delete pre_integrations[i];
        pre_integrations[i] = nullptr;
    }
    tic = V3();
    ric = M3::I();
    first_imu = false;
    frame_count = 0;
    solver_flag = 0;
    td = cfg.td;
    openExEstimation = false;
    has_prior = false;
    feature.clear();
    failure_occur = false;
    relocalization_info = false;   // :96
    initFirstPoseFlag = false;
    prevTime = -1;
    latest_Bg = V3();
    for (auto &it : all_image_frame) delete it.second.pre_integration;
    all_image_frame.clear();
    delete tmp_pre_integration;
    tmp_pre_integration = nullptr;
    initial_timestamp = 0;
}
void Estimator::inputIMU(double t, const V3 &acc, const V3 &gyr) {  // :1749-1766 (predict() path is output-only)
    imu_buf.push_back(ImuSample{t, acc, gyr});
}
bool Estimator::IMUAvailable(double t) const {  // :1882-1888
    return imu_head < imu_buf.size() && t <= imu_buf.back().t;
}
void Estimator::predictMotion(double t0, double t1, double R[9], const V3 *bg_override) {  // :1790-1860
    const V3 bg_used = bg_override ? *bg_override : latest_Bg;
    M3 rel = M3::I();
    auto out = [&]() { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = rel(i, j); };
    if (imu_head >= imu_buf.size()) { out(); return; }
    if (t1 <= imu_buf.back().t) {
        size_t k = imu_head;
        while (k < imu_buf.size() && imu_buf[k].t <= t0) k++;
        bool first = true;
        double prev_t = 0;
        V3 prev_gyr;
        while (k < imu_buf.size() && imu_buf[k].t <= t1) {
            double t = imu_buf[k].t;
            V3 w = imu_buf[k].gyr;
            k++;
            if (first) { prev_t = t; first = false; prev_gyr = w; continue; }
            double dt = t - prev_t;
            prev_t = t;
            V3 un_gyr = 0.5 * (prev_gyr + w) - bg_used;
            prev_gyr = w;
            M3 RIC;  // the GLOBAL RIC.back() (estimator.cpp:1852): the configured extrinsic, not the refined Estimator::ric
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RIC(i, j) = cfg.ric[i * 3 + j];
            V3 aa = (T(RIC) * un_gyr) * dt;  // RIC.back().transpose() * un_gyr * dt
            double ang = norm(aa);
            // AngleAxisd(|aa|, aa.normalized()).toRotationMatrix().transpose(); zero vector normalises to zero -> identity
            M3 Rk = M3::I();
            if (ang > 0) {
                V3 ax = aa / ang;
                double s, c;
                sincos_det(ang, &s, &c);   // same bits as the device code (om.h)
                M3 K = skew(ax);
                Rk = M3::I() + s * K + (1 - c) * (K * K);
            }
            rel = rel * T(Rk);
        }
    }
    out();
}
// Estimator::predict (estimator.cpp:1862-1880) from the newest window state through every buffered sample newer than it (the pose
// pubLatestOdometry publishes at IMU rate).  Default: every sample with its own values.  cfg.reference_quirks bit 0: literally what
// updateLatestStates does (estimator.cpp:1779-1786) -- the loop pops a COPY of the queue for the stamps but passes predict() the values of
// the real queue's front sample every time, and predict() (:1862-1880) never advances acc_0 / gyr_0.
void Estimator::latestOdometry(double out[11]) const {
    double latest_time = Headers[frame_count] + td;
    V3 P = Ps[frame_count], V = Vs[frame_count], Ba = Bas[frame_count], Bg = Bgs[frame_count];
    M3 R = Rs[frame_count];
    V3 a0 = acc_0, g0 = gyr_0;
    const bool front = (cfg.reference_quirks & 1) != 0;
    if (solver_flag == 1 && cfg.use_imu)
        for (size_t k = imu_head; k < imu_buf.size(); k++) {
            const double t = imu_buf[k].t;
            if (!(t > latest_time)) continue;
            const double dt = t - latest_time;
            latest_time = t;
            // quirk mode: the samples that were buffered when updateLatestStates ran are replayed with the FRONT sample's values
            // (:1779-1786); the ones that arrived afterwards went through inputIMU -> predict(t, own acc, own gyr) (:1758-1764).
            // predict() never advances acc_0 / gyr_0 either way (:1862-1880)
            const ImuSample &v = (front && k < imu_at_update) ? imu_buf[imu_head] : imu_buf[k];
            V3 un_acc_0 = R * (a0 - Ba) - g;
            V3 un_gyr = 0.5 * (g0 + v.gyr) - Bg;
            R = R * toR(deltaQ(un_gyr * dt));
            V3 un_acc_1 = R * (v.acc - Ba) - g;
            V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
            P = P + dt * V + (0.5 * dt * dt) * un_acc;
            V = V + dt * un_acc;
            if (!front) { a0 = v.acc; g0 = v.gyr; }
        }
    Q q = fromR(R);
    out[0] = latest_time; out[1] = P.x; out[2] = P.y; out[3] = P.z; out[4] = q.w; out[5] = q.x; out[6] = q.y; out[7] = q.z;
    out[8] = V.x; out[9] = V.y; out[10] = V.z;
}
void Estimator::processIMU(double dt, const V3 &acc, const V3 &gyr) {  // :118-154
    if (!first_imu) { first_imu = true; acc_0 = acc; gyr_0 = gyr; }
    if (!pre_integrations[frame_count])
        pre_integrations[frame_count] = new Integration(cfg, acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]);
    if (cfg.dynamic_init && solver_flag == 0 && !tmp_pre_integration) tmp_pre_integration = new Integration(cfg, acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]);
    if (frame_count != 0) {
        pre_integrations[frame_count]->push_back(dt, acc, gyr);
        if (tmp_pre_integration) tmp_pre_integration->push_back(dt, acc, gyr);  // :137 (only the dynamic initialisation reads it)
        int j = frame_count;
        V3 un_acc_0 = Rs[j] * (acc_0 - Bas[j]) - g;
        V3 un_gyr = 0.5 * (gyr_0 + gyr) - Bgs[j];
        Rs[j] = Rs[j] * toR(deltaQ(un_gyr * dt));
        V3 un_acc_1 = Rs[j] * (acc - Bas[j]) - g;
        V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
        Ps[j] = Ps[j] + dt * Vs[j] + 0.5 * dt * dt * un_acc;
        Vs[j] = Vs[j] + dt * un_acc;
    }
    acc_0 = acc;
    gyr_0 = gyr;
}
void Estimator::initFirstIMUPose(const std::vector<ImuSample> &v) {  // :1890-1909
    initFirstPoseFlag = true;
    V3 aver;
    for (auto &s : v) aver = aver + s.acc;
    aver = aver / (double)v.size();
    M3 R0 = g2R(aver);
    double yaw = R2ypr(R0).x;
    R0 = ypr2R(V3(-yaw, 0, 0)) * R0;
    Rs[0] = R0;
}

// ------------------------------------------------------------------ feature manager
static inline bool in_problem(const Landmark &l, int W) { return l.obs.size() >= 2 && l.start_frame < W - 2; }
int Estimator::getFeatureCount() {  // feature_manager.cpp:31-49
    int cnt = 0;
    for (auto &it : feature) {
        if (it.is_dynamic) continue;
        it.used_num = (int)it.obs.size();
        if (in_problem(it, W)) cnt++;
    }
    return cnt;
}
bool Estimator::addFeatureCheckParallax(int fc, std::map<int, std::array<double, 7>> &image, double td_) {  // :56-123
    double parallax_sum = 0;
    int parallax_num = 0;
    last_track_num = 0;
    for (auto iter = image.begin(); iter != image.end();) {
        const auto &p = iter->second;
        unsigned short mm = depth_img[(size_t)(int)p[4] * cfg.width + (int)p[3]];
        double dm = mm / 1000.0;
        if (0 < dm && dm < cfg.depth_min) { iter = image.erase(iter); continue; }
        int fid = iter->first;
        auto it = std::find_if(feature.begin(), feature.end(), [fid](const Landmark &l) { return l.feature_id == fid; });
        Obs o{p[0], p[1], p[2], p[3], p[4], p[5], p[6], td_, dm};
        if (it == feature.end()) {
            Landmark l;
            l.feature_id = fid;
            l.start_frame = fc;
            l.obs.push_back(o);
            feature.push_back(l);
        } else {
            it->obs.push_back(o);
            last_track_num++;
        }
        ++iter;
    }
    if (fc < 2 || last_track_num < 20) return true;
    for (auto &l : feature)
        if (l.start_frame <= fc - 2 && l.start_frame + (int)l.obs.size() - 1 >= fc - 1) {
            parallax_sum += compensatedParallax2(l, fc);
            parallax_num++;
        }
    if (parallax_num == 0) return true;
    return parallax_sum / parallax_num >= cfg.min_parallax_px / cfg.focal_length;
}
double Estimator::compensatedParallax2(const Landmark &l, int fc) const {  // :732-768
    const Obs &fi = l.obs[fc - 2 - l.start_frame], &fj = l.obs[fc - 1 - l.start_frame];
    double u_j = fj.x, v_j = fj.y;
    double dep_i = fi.z, u_i = fi.x / dep_i, v_i = fi.y / dep_i;
    double du = u_i - u_j, dv = v_i - v_j;
    return std::max(0.0, std::sqrt(std::min(du * du + dv * dv, du * du + dv * dv)));
}
void Estimator::setDepth(const std::vector<double> &x) {  // :197-223
    int idx = -1;
    for (auto &l : feature) {
        if (l.is_dynamic) continue;
        l.used_num = (int)l.obs.size();
        if (!in_problem(l, W)) continue;
        l.estimated_depth = 1.0 / x[++idx];
        l.solve_flag = l.estimated_depth < 0 ? 2 : 1;
    }
}
std::vector<double> Estimator::getDepthVector() {  // :302-324
    std::vector<double> d;
    for (auto &l : feature) {
        if (l.is_dynamic) continue;
        l.used_num = (int)l.obs.size();
        if (!in_problem(l, W)) continue;
        d.push_back(1. / l.estimated_depth);
    }
    return d;
}
void Estimator::removeFailures() {  // :225-233
    for (auto it = feature.begin(); it != feature.end();) {
        if (it->solve_flag == 2) it = feature.erase(it); else ++it;
    }
}
void Estimator::triangulateWithDepth() {  // :386-543
    for (auto &l : feature) {
        if (l.estimated_depth > 0) continue;
        if (l.is_dynamic) continue;
        l.used_num = (int)l.obs.size();
        if (!in_problem(l, W)) continue;
        int imu_i = l.start_frame;
        V3 tr = Ps[imu_i] + Rs[imu_i] * tic;
        M3 Rr = Rs[imu_i] * ric;
        std::vector<double> verified, rough;
        int no_depth_num = 0;
        int K = (int)l.obs.size();
        for (int k = 0; k < K; k++) {
            if (l.obs[k].depth == 0) { no_depth_num++; continue; }
            V3 t0 = Ps[imu_i + k] + Rs[imu_i + k] * tic;
            M3 R0 = Rs[imu_i + k] * ric;
            V3 point0 = V3(l.obs[k].x, l.obs[k].y, l.obs[k].z) * l.obs[k].depth;
            V3 t2r = T(Rr) * (t0 - tr);
            M3 R2r = T(Rr) * R0;
            for (int j = 0; j < K; j++) {
                if (k == j) continue;
                V3 t1 = Ps[imu_i + j] + Rs[imu_i + j] * tic;
                M3 R1 = Rs[imu_i + j] * ric;
                V3 t20 = T(R0) * (t1 - t0);
                M3 R20 = T(R0) * R1;
                V3 pp = T(R20) * point0 - T(R20) * t20;
                double rx = l.obs[j].x - pp.x / pp.z, ry = l.obs[j].y - pp.y / pp.z;
                if (std::sqrt(rx * rx + ry * ry) < 10.0 / 460) {
                    V3 pr = R2r * point0 + t2r;
                    if (l.obs[k].depth > cfg.depth_max) rough.push_back(pr.z); else verified.push_back(pr.z);
                }
            }
        }
        if (verified.empty()) {
            if (rough.empty()) {
                if (no_depth_num == K) {
                    // DLT: smallest right singular vector of svd_A == smallest eigenvector of A^T A
                    Mat AtA(4, 4);
                    V3 t0 = Ps[imu_i] + Rs[imu_i] * tic;
                    M3 R0 = Rs[imu_i] * ric;
                    for (int k = 0; k < K; k++) {
                        int imu_j = imu_i + k;
                        V3 t1 = Ps[imu_j] + Rs[imu_j] * tic;
                        M3 R1 = Rs[imu_j] * ric;
                        V3 t = T(R0) * (t1 - t0);
                        M3 R = T(R0) * R1;
                        double Pm[3][4];
                        M3 Rt = T(R);
                        V3 mt = -(Rt * t);
                        for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) Pm[a][b] = Rt(a, b); Pm[a][3] = mt[a]; }
                        V3 f(l.obs[k].x, l.obs[k].y, l.obs[k].z);
                        f = f / norm(f);
                        double row[2][4];
                        for (int b = 0; b < 4; b++) { row[0][b] = f.x * Pm[2][b] - f.z * Pm[0][b]; row[1][b] = f.y * Pm[2][b] - f.z * Pm[1][b]; }
                        for (int rr = 0; rr < 2; rr++)
                            for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) AtA(a, b) += row[rr][a] * row[rr][b];
                    }
                    std::vector<double> w;
                    Mat V;
                    sym_eig(AtA, w, V);
                    double svd_method = V(2, 0) / V(3, 0);
                    l.estimated_depth = svd_method < cfg.depth_min ? cfg.depth_max : svd_method;
                    l.estimate_flag = 2;
                } else
                    continue;
            } else {
                l.estimated_depth = std::accumulate(rough.begin(), rough.end(), 0.0) / rough.size();
                l.estimate_flag = 0;
            }
        } else {
            l.estimated_depth = std::accumulate(verified.begin(), verified.end(), 0.0) / verified.size();
            l.estimate_flag = 1;
        }
        if (l.estimated_depth < 0.1) { l.estimated_depth = cfg.init_depth; l.estimate_flag = 0; }
    }
}
void Estimator::removeBackShiftDepth(const M3 &mR, const V3 &mP, const M3 &nR, const V3 &nP) {  // :660-691
    for (auto it = feature.begin(); it != feature.end();) {
        if (it->start_frame != 0) { it->start_frame--; ++it; continue; }
        V3 uv_i(it->obs[0].x, it->obs[0].y, it->obs[0].z);
        it->obs.erase(it->obs.begin());
        if (it->obs.size() < 2) { it = feature.erase(it); continue; }
        V3 pts_i = uv_i * it->estimated_depth;
        V3 w_pts_i = mR * pts_i + mP;
        V3 pts_j = T(nR) * (w_pts_i - nP);
        it->estimated_depth = pts_j.z > 0 ? pts_j.z : cfg.init_depth;
        ++it;
    }
}
void Estimator::removeBack() {  // :693-708
    for (auto it = feature.begin(); it != feature.end();) {
        if (it->start_frame != 0) { it->start_frame--; ++it; continue; }
        it->obs.erase(it->obs.begin());
        if (it->obs.empty()) it = feature.erase(it); else ++it;
    }
}
void Estimator::removeFront(int fc) {  // :710-730
    for (auto it = feature.begin(); it != feature.end();) {
        if (it->start_frame == fc) { it->start_frame--; ++it; continue; }
        int j = W - 1 - it->start_frame;
        if (it->endFrame() < fc - 1) { ++it; continue; }
        it->obs.erase(it->obs.begin() + j);
        if (it->obs.empty()) it = feature.erase(it); else ++it;
    }
}

// ------------------------------------------------------------------ init helper (initial_aligment.cpp:3-36)
void Estimator::solveGyroscopeBias() {
    double A[3][3] = {{0}}, b[3] = {0};
    for (int i = 0; i < W; i++) {
        int j = i + 1;
        Q q_ij = fromR(T(Rs[i]) * Rs[j]);
        M3 tA = blk(pre_integrations[j]->jacobian, O_R, O_BG);
        V3 tb = 2.0 * (inverse(pre_integrations[j]->delta_q) * q_ij).vec();
        M3 AtA = T(tA) * tA;
        V3 Atb = T(tA) * tb;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) A[r][c] += AtA(r, c); b[r] += Atb[r]; }
    }
    // A.ldlt().solve(b) — 3×3 SPD; Gaussian elimination with symmetric pivot order
    double M[3][4];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[r][c] = A[r][c]; M[r][3] = b[r]; }
    for (int i = 0; i < 3; i++) {
        int p = i;
        for (int r = i + 1; r < 3; r++) if (std::fabs(M[r][i]) > std::fabs(M[p][i])) p = r;
        for (int c = 0; c < 4; c++) std::swap(M[i][c], M[p][c]);
        if (M[i][i] == 0) continue;
        for (int r = i + 1; r < 3; r++) {
            double f = M[r][i] / M[i][i];
            for (int c = i; c < 4; c++) M[r][c] -= f * M[i][c];
        }
    }
    double x[3] = {0, 0, 0};
    for (int i = 2; i >= 0; i--) {
        double s = M[i][3];
        for (int c = i + 1; c < 3; c++) s -= M[i][c] * x[c];
        x[i] = M[i][i] != 0 ? s / M[i][i] : 0;
    }
    V3 dbg(x[0], x[1], x[2]);
    for (int i = 0; i <= W; i++) Bgs[i] = Bgs[i] + dbg;
}

// ------------------------------------------------------------------ state <-> flat arrays
// FeatureManager::initFramePoseByPnP + solvePoseByPnP (feature_manager.cpp:545-642): VO mode only.  Every landmark with a depth that is
// observed in frame frameCnt gives a 3-D (world) / 2-D (normalised) pair; cv::solvePnP(ITERATIVE) starts from the previous frame's pose.
void Estimator::initFramePoseByPnP(int frameCnt) {
    if (frameCnt <= 0) return;
    std::vector<V3> pts3;
    std::vector<std::array<double, 2>> pts2;
    for (auto &l : feature) {
        if (!(l.estimated_depth > 0)) continue;
        const int index = frameCnt - l.start_frame;
        if ((int)l.obs.size() >= index + 1 && index >= 0) {
            V3 ptsInCam = ric * (V3(l.obs[0].x, l.obs[0].y, l.obs[0].z) * l.estimated_depth) + tic;
            V3 ptsInWorld = Rs[l.start_frame] * ptsInCam + Ps[l.start_frame];
            pts3.push_back(ptsInWorld);
            pts2.push_back({l.obs[index].x, l.obs[index].y});
        }
    }
    M3 RCam = Rs[frameCnt - 1] * ric;
    V3 PCam = Rs[frameCnt - 1] * tic + Ps[frameCnt - 1];
    // solvePoseByPnP: w_T_cam -> cam_T_w, refine, back
    M3 R_initial = T(RCam);
    V3 P_initial = -1.0 * (R_initial * PCam);
    if ((int)pts2.size() < 4) return;
    if (!solve_pnp_iterative(pts3, pts2, R_initial, P_initial)) return;
    RCam = T(R_initial);
    PCam = RCam * (-1.0 * P_initial);
    Rs[frameCnt] = RCam * T(ric);
    Ps[frameCnt] = -1.0 * (RCam * (T(ric) * tic)) + PCam;
}

void Estimator::setReloFrame(double frame_stamp, int frame_index, const std::vector<std::array<double, 3>> &mp, const V3 &relo_t, const M3 &relo_r) {  // :1728-1747
    relo_frame_stamp = frame_stamp;
    relo_frame_index = frame_index;
    match_points = mp;
    prev_relo_t = relo_t;
    prev_relo_r = relo_r;
    for (int i = 0; i < W; i++)
        if (relo_frame_stamp == Headers[i]) {
            relo_frame_local_index = i;
            relocalization_info = true;
            for (int j = 0; j < 7; j++) relo_Pose[j] = para_Pose[i][j];   // (para_Pose as the last vector2double left it, like upstream)
        }
}

void Estimator::vector2double() {  // estimator.cpp:936-981
    for (int i = 0; i <= W; i++) {
        para_Pose[i][0] = Ps[i].x; para_Pose[i][1] = Ps[i].y; para_Pose[i][2] = Ps[i].z;
        Q q = fromR(Rs[i]);
        para_Pose[i][3] = q.x; para_Pose[i][4] = q.y; para_Pose[i][5] = q.z; para_Pose[i][6] = q.w;
        for (int k = 0; k < 3; k++) { para_SpeedBias[i][k] = Vs[i][k]; para_SpeedBias[i][3 + k] = Bas[i][k]; para_SpeedBias[i][6 + k] = Bgs[i][k]; }
    }
    para_Ex_Pose[0] = tic.x; para_Ex_Pose[1] = tic.y; para_Ex_Pose[2] = tic.z;
    Q q = fromR(ric);
    para_Ex_Pose[3] = q.x; para_Ex_Pose[4] = q.y; para_Ex_Pose[5] = q.z; para_Ex_Pose[6] = q.w;
    para_Feature = getDepthVector();
    if (cfg.estimate_td) para_Td = td;
}
void Estimator::double2vector() {  // estimator.cpp:985-1111
    V3 origin_R0 = R2ypr(Rs[0]);
    V3 origin_P0 = Ps[0];
    if (failure_occur) {
        origin_R0 = R2ypr(last_R0);
        origin_P0 = last_P0;
        failure_occur = false;
    }
    auto poseQ = [&](int i) { return Q(para_Pose[i][6], para_Pose[i][3], para_Pose[i][4], para_Pose[i][5]); };
    auto relo_outputs = [&](const M3 &relo_r, const V3 &relo_t) {   // "relative info between two loop frame" (:1046-1056 / 1080-1090)
        const double drift_correct_yaw = R2ypr(prev_relo_r).x - R2ypr(relo_r).x;
        drift_correct_r = ypr2R(V3(drift_correct_yaw, 0, 0));
        drift_correct_t = prev_relo_t - drift_correct_r * relo_t;
        relo_relative_t = T(relo_r) * (Ps[relo_frame_local_index] - relo_t);
        relo_relative_q = fromR(T(relo_r) * Rs[relo_frame_local_index]);
        double a = R2ypr(Rs[relo_frame_local_index]).x - R2ypr(relo_r).x;   // Utility::normalizeAngle (utility.h:131-139), degrees
        relo_relative_yaw = a > 0 ? a - 360.0 * std::floor((a + 180.0) / 360.0) : a + 360.0 * std::floor((-a + 180.0) / 360.0);
        relocalization_info = false;
    };
    auto reloQ = [&]() { return toR(normalized(Q(relo_Pose[6], relo_Pose[3], relo_Pose[4], relo_Pose[5]))); };
    if (!cfg.use_imu) {   // :1060-1067: no gauge fix, no speed / bias / extrinsic / td hand-back
        for (int i = 0; i <= W; i++) {
            Rs[i] = toR(normalized(poseQ(i)));
            Ps[i] = V3(para_Pose[i][0], para_Pose[i][1], para_Pose[i][2]);
        }
        if (relocalization_info) relo_outputs(reloQ(), V3(relo_Pose[0], relo_Pose[1], relo_Pose[2]));
        setDepth(para_Feature);
        return;
    }
    V3 origin_R00 = R2ypr(toR(poseQ(0)));
    double y_diff = origin_R0.x - origin_R00.x;
    M3 rot_diff = ypr2R(V3(y_diff, 0, 0));
    if (std::fabs(std::fabs(origin_R0.y) - 90) < 1.0 || std::fabs(std::fabs(origin_R00.y) - 90) < 1.0)
        rot_diff = Rs[0] * T(toR(poseQ(0)));
    for (int i = 0; i <= W; i++) {
        Rs[i] = rot_diff * toR(normalized(poseQ(i)));
        Ps[i] = rot_diff * V3(para_Pose[i][0] - para_Pose[0][0], para_Pose[i][1] - para_Pose[0][1], para_Pose[i][2] - para_Pose[0][2]) + origin_P0;
        Vs[i] = rot_diff * V3(para_SpeedBias[i][0], para_SpeedBias[i][1], para_SpeedBias[i][2]);
        Bas[i] = V3(para_SpeedBias[i][3], para_SpeedBias[i][4], para_SpeedBias[i][5]);
        Bgs[i] = V3(para_SpeedBias[i][6], para_SpeedBias[i][7], para_SpeedBias[i][8]);
    }
    if (relocalization_info)
        relo_outputs(rot_diff * reloQ(),
                     rot_diff * V3(relo_Pose[0] - para_Pose[0][0], relo_Pose[1] - para_Pose[0][1], relo_Pose[2] - para_Pose[0][2]) + origin_P0);
    tic = V3(para_Ex_Pose[0], para_Ex_Pose[1], para_Ex_Pose[2]);
    ric = toR(normalized(Q(para_Ex_Pose[6], para_Ex_Pose[3], para_Ex_Pose[4], para_Ex_Pose[5])));
    setDepth(para_Feature);
    if (cfg.estimate_td) td = para_Td;
}

// ------------------------------------------------------------------ the solve (Ceres restatement)
namespace {
struct LmRef { Landmark *l; int idx; bool is_const; double ub; };
struct NormalEq {
    int P, F;
    Mat H;                    // P×P
    std::vector<double> g;    // P
    std::vector<double> Hll, gl;  // F
    Mat Hpl;                  // F×P
    double cost;
};
}  // namespace

static void pose_dx(const double *x, const double *x0, double *dx) {  // marginalization_factor.cpp:374-393
    for (int k = 0; k < 3; k++) dx[k] = x[k] - x0[k];
    Q q0(x0[6], x0[3], x0[4], x0[5]), q(x[6], x[3], x[4], x[5]);
    Q d = inverse(q0) * q;
    V3 v = 2.0 * d.vec();
    if (!(d.w >= 0)) v = -v;
    dx[3] = v.x; dx[4] = v.y; dx[5] = v.z;
}

// MarginalizationFactor::Evaluate (marginalization_factor.cpp:353-404) linearised in the prior's own index space: cost 0.5 |r + J dx|^2,
// gradient J^T (r + J dx), Gauss-Newton block J^T J.  With ODEV_QUADRATIC_PRIOR (attribution experiment) the same three quantities from the
// quadratic form (A, b, c0): 0.5 c0 + dx^T b + 0.5 dx^T A dx, b + A dx, A.
static double prior_gradient(const Estimator &e, const std::vector<double> &dx, std::vector<double> &g, std::vector<double> &r) {
    const int n = e.prior_n;
    double cost = 0;
    g.assign(n, 0.0);
    if (e.deviations & ODEV_QUADRATIC_PRIOR) {
        for (int a = 0; a < n; a++) {
            double acc = 0;
            for (int j = 0; j < n; j++) acc += e.prior_A(a, j) * dx[j];
            const double q = e.prior_b[a] + acc;
            g[a] = q;
            cost += 0.5 * dx[a] * (e.prior_b[a] + q);
        }
        cost += 0.5 * e.prior_c0;
        return cost;
    }
    r.assign(n, 0.0);
    for (int i = 0; i < n; i++) {
        double s = e.prior_r[i];
        for (int j = 0; j < n; j++) s += e.prior_J(i, j) * dx[j];
        r[i] = s;
        cost += 0.5 * s * s;
    }
    for (int a = 0; a < n; a++) {
        double gs = 0;
        for (int i = 0; i < n; i++) gs += e.prior_J(i, a) * r[i];
        g[a] = gs;
    }
    return cost;
}
static inline double prior_hessian(const Estimator &e, int a, int b) {
    if (e.deviations & ODEV_QUADRATIC_PRIOR) return e.prior_A(a, b);
    double s = 0;
    for (int i = 0; i < e.prior_n; i++) s += e.prior_J(i, a) * e.prior_J(i, b);
    return s;
}

// Evaluate all factors at the flat parameters and accumulate the (robustified) normal equations.
// Tangent layout: pose k at 6k, speed-bias k at 6(W+1)+9k, ex at 15(W+1), td at 15(W+1)+6.
// relo: the copy of the matched frame's pose (relo_Pose) or NULL; its 6 tangent dimensions sit behind td at index 15(W+1)+7.
static void build_normal_eq(Estimator &e, const double pose[][7], const double sb[][9], const double *ex, double tdv,
                            const std::vector<double> &feat, std::vector<LmRef> &lms, NormalEq &ne, bool withJ, const double *relo = nullptr) {
    const int W = e.W;
    const int oR = 15 * (W + 1) + 7;
    const int P = oR + (relo ? 6 : 0);
    const int oP = 0, oS = 6 * (W + 1), oE = 15 * (W + 1), oT = 15 * (W + 1) + 6;
    ne.P = P; ne.F = (int)lms.size();
    if (withJ) {
        ne.H = Mat(P, P); ne.g.assign(P, 0.0); ne.Hll.assign(ne.F, 0.0); ne.gl.assign(ne.F, 0.0); ne.Hpl = Mat(ne.F, P);
    }
    double cost = 0;
    // prior (MarginalizationFactor::Evaluate)
    if (e.has_prior) {
        int n = e.prior_n;
        std::vector<double> dx(n, 0.0);
        std::vector<int> map(n, 0);
        for (int k = 0; k < W; k++) { pose_dx(pose[k], &e.prior_x0[k * 7], &dx[6 * k]); for (int d = 0; d < 6; d++) map[6 * k + d] = oP + 6 * k + d; }
        for (int d = 0; d < 9; d++) { dx[6 * W + d] = sb[0][d] - e.prior_x0[W * 7 + d]; map[6 * W + d] = oS + d; }
        pose_dx(ex, &e.prior_x0[W * 7 + 9], &dx[6 * W + 9]);
        for (int d = 0; d < 6; d++) map[6 * W + 9 + d] = oE + d;
        dx[6 * W + 15] = tdv - e.prior_x0[W * 7 + 16];
        map[6 * W + 15] = oT;
        // absent blocks have zero Jacobian columns; zero their dx so stale x0 never contributes
        for (int k = 0; k < W; k++) if (!e.prior_present[k]) for (int d = 0; d < 6; d++) dx[6 * k + d] = 0;
        if (!e.prior_present[W]) for (int d = 0; d < 9; d++) dx[6 * W + d] = 0;
        if (!e.prior_present[W + 1]) for (int d = 0; d < 6; d++) dx[6 * W + 9 + d] = 0;
        if (!e.prior_present[W + 2]) dx[6 * W + 15] = 0;
        std::vector<double> r, pg;
        cost += prior_gradient(e, dx, pg, r);
        if (withJ) {
            for (int a = 0; a < n; a++) {
                ne.g[map[a]] += pg[a];
                for (int b = 0; b < n; b++) ne.H(map[a], map[b]) += prior_hessian(e, a, b);
            }
        }
    }
    // IMU factors
    for (int i = 0; i < W && e.cfg.use_imu; i++) {
        int j = i + 1;
        if (e.pre_integrations[j]->sum_dt > 10.0) continue;
        double r[15], Ji[15 * 7], Jsi[15 * 9], Jj[15 * 7], Jsj[15 * 9];
        eval_imu(*e.pre_integrations[j], e.g, pose[i], sb[i], pose[j], sb[j], r, withJ ? Ji : nullptr, Jsi, Jj, Jsj);
        for (int k = 0; k < 15; k++) cost += 0.5 * r[k] * r[k];
        if (withJ) {
            // assemble 15×30 local Jacobian: [pose_i(6) sb_i(9) pose_j(6) sb_j(9)]
            double J[15][30];
            int idx[30];
            for (int k = 0; k < 15; k++) {
                for (int d = 0; d < 6; d++) { J[k][d] = Ji[k * 7 + d]; J[k][15 + d] = Jj[k * 7 + d]; }
                for (int d = 0; d < 9; d++) { J[k][6 + d] = Jsi[k * 9 + d]; J[k][21 + d] = Jsj[k * 9 + d]; }
            }
            for (int d = 0; d < 6; d++) { idx[d] = oP + 6 * i + d; idx[15 + d] = oP + 6 * j + d; }
            for (int d = 0; d < 9; d++) { idx[6 + d] = oS + 9 * i + d; idx[21 + d] = oS + 9 * j + d; }
            for (int a = 0; a < 30; a++) {
                double gs = 0;
                for (int k = 0; k < 15; k++) gs += J[k][a] * r[k];
                ne.g[idx[a]] += gs;
                for (int b = 0; b < 30; b++) {
                    double s = 0;
                    for (int k = 0; k < 15; k++) s += J[k][a] * J[k][b];
                    ne.H(idx[a], idx[b]) += s;
                }
            }
        }
    }
    // projection factors with CauchyLoss(1.0)
    int nres = 0;
    // ORACLE_PERTURB_ORDER (control experiment only, tests/oracle_control.py): the same sums accumulated in the opposite order --
    // landmarks last to first, observations newest to oldest. Mathematically identical normal equations, different round-off; used
    // to measure how far two arithmetically different builds of THIS restatement drift apart over a long run.
    for (size_t lmi = 0; lmi < lms.size(); lmi++) {
#ifdef ORACLE_PERTURB_ORDER
        LmRef &lr = lms[lms.size() - 1 - lmi];
#else
        LmRef &lr = lms[lmi];
#endif
        Landmark &l = *lr.l;
        int imu_i = l.start_frame;
        double inv_dep = feat[lr.idx];
        for (int kk = 1; kk < (int)l.obs.size(); kk++) {
#ifdef ORACLE_PERTURB_ORDER
            const int k = (int)l.obs.size() - kk;
#else
            const int k = kk;
#endif
            int imu_j = imu_i + k;
            double r[2], Ji[14], Jj[14], Je[14], Jl[2], Jt[2];
            eval_projection(e.cfg, pose[imu_i], pose[imu_j], ex, inv_dep, tdv, l.obs[0], l.obs[k], e.cfg.estimate_td != 0, r,
                            withJ ? Ji : nullptr, Jj, Je, Jl, Jt);
            nres++;
            double s = r[0] * r[0] + r[1] * r[1];
            cost += 0.5 * std::log(1.0 + s);  // rho(s) = log(1+s)
            if (!withJ) continue;
            double wgt = std::sqrt(1.0 / (1.0 + s));  // sqrt(rho'); rho'' < 0 => alpha = 0 (Triggs correction degenerate branch)
            double J[2][20];
            int idx[20];
            for (int a = 0; a < 2; a++) {
                for (int d = 0; d < 6; d++) { J[a][d] = wgt * Ji[a * 7 + d]; J[a][6 + d] = wgt * Jj[a * 7 + d]; J[a][12 + d] = wgt * Je[a * 7 + d]; }
                J[a][18] = wgt * Jt[a];
                J[a][19] = wgt * Jl[a];
                r[a] *= wgt;
            }
            for (int d = 0; d < 6; d++) { idx[d] = oP + 6 * imu_i + d; idx[6 + d] = oP + 6 * imu_j + d; idx[12 + d] = oE + d; }
            idx[18] = oT;
            for (int a = 0; a < 19; a++) {
                ne.g[idx[a]] += J[0][a] * r[0] + J[1][a] * r[1];
                for (int b = 0; b < 19; b++) ne.H(idx[a], idx[b]) += J[0][a] * J[0][b] + J[1][a] * J[1][b];
                ne.Hpl(lr.idx, idx[a]) += J[0][a] * J[0][19] + J[1][a] * J[1][19];
            }
            ne.Hll[lr.idx] += J[0][19] * J[0][19] + J[1][19] * J[1][19];
            ne.gl[lr.idx] += J[0][19] * r[0] + J[1][19] * r[1];
        }
    }
    // relocalisation factors (estimator.cpp:1307-1346): ProjectionFactor(first observation, matched point of the old keyframe) on
    // (para_Pose[start], relo_Pose, para_Ex_Pose, para_Feature) for every in-problem landmark with start <= relo_frame_local_index whose
    // id is among the matches; the match list is walked with one monotone cursor exactly like upstream (both lists ascend in id)
    int nrelo = 0;
    if (relo) {
        size_t cur = 0;
        for (auto &lr : lms) {
            Landmark &l = *lr.l;
            if (l.start_frame > e.relo_frame_local_index) continue;
            while (cur < e.match_points.size() && (int)e.match_points[cur][2] < l.feature_id) cur++;
            if (cur >= e.match_points.size()) break;   // (upstream reads past the end here; nothing can match any more)
            if ((int)e.match_points[cur][2] != l.feature_id) continue;
            Obs oj = l.obs[0];
            oj.x = e.match_points[cur][0]; oj.y = e.match_points[cur][1]; oj.z = 1.0;
            cur++;
            double r[2], Ji[14], Jj[14], Je[14], Jl[2], Jt[2];
            eval_projection(e.cfg, pose[l.start_frame], relo, ex, feat[lr.idx], tdv, l.obs[0], oj, false, r, withJ ? Ji : nullptr, Jj, Je, Jl, Jt);
            nrelo++;
            const double sq = r[0] * r[0] + r[1] * r[1];
            cost += 0.5 * std::log(1.0 + sq);
            if (!withJ) continue;
            const double wgt = std::sqrt(1.0 / (1.0 + sq));
            double J[2][19];
            int idx[18];
            for (int a = 0; a < 2; a++) {
                for (int d = 0; d < 6; d++) { J[a][d] = wgt * Ji[a * 7 + d]; J[a][6 + d] = wgt * Jj[a * 7 + d]; J[a][12 + d] = wgt * Je[a * 7 + d]; }
                J[a][18] = wgt * Jl[a];
                r[a] *= wgt;
            }
            for (int d = 0; d < 6; d++) { idx[d] = oP + 6 * l.start_frame + d; idx[6 + d] = oR + d; idx[12 + d] = oE + d; }
            for (int a = 0; a < 18; a++) {
                ne.g[idx[a]] += J[0][a] * r[0] + J[1][a] * r[1];
                for (int b = 0; b < 18; b++) ne.H(idx[a], idx[b]) += J[0][a] * J[0][b] + J[1][a] * J[1][b];
                ne.Hpl(lr.idx, idx[a]) += J[0][a] * J[0][18] + J[1][a] * J[1][18];
            }
            ne.Hll[lr.idx] += J[0][18] * J[0][18] + J[1][18] * J[1][18];
            ne.gl[lr.idx] += J[0][18] * r[0] + J[1][18] * r[1];
        }
    }
    e.relo_residuals = nrelo;
    e.last_stats.n_residuals = nres;
    ne.cost = cost;
}

void Estimator::solve() {
    const bool relo_on = relocalization_info;
    const int oR = 15 * (W + 1) + 7;
    const int P = oR + (relo_on ? 6 : 0);   // the relocalisation pose is one more 6-dof block of the reduced system (:1310-1311)
    const int oE = 15 * (W + 1), oT = 15 * (W + 1) + 6;
    // in-problem landmarks, list order (estimator.cpp:1243-1302)
    std::vector<LmRef> lms;
    {
        int idx = -1;
        for (auto &l : feature) {
            if (l.is_dynamic) continue;
            l.used_num = (int)l.obs.size();
            if (!in_problem(l, W)) continue;
            ++idx;
            LmRef r;
            r.l = &l; r.idx = idx;
            r.is_const = (l.estimate_flag == 1 && cfg.fix_depth);
            r.ub = (l.estimate_flag == 2) ? 2.0 / cfg.depth_max : DBL_MAX;
            lms.push_back(r);
        }
    }
    const int F = (int)lms.size();
    // Ceres: Program::IsBoundsConstrained() of the reduced program -- any NON-constant block with a finite bound (estimator.cpp:1282-1297)
    bool constrained = false;
    for (const auto &r : lms) if (r.ub < DBL_MAX && !r.is_const) { bounded_landmark_solves++; constrained = true; }
    if (cfg.reference_quirks & 8) constrained = false;   // test switch (NOT a reference behaviour): rounds 1 - 5's clamp-only treatment of the bound
    // constness (estimator.cpp:1187-1212)
    std::vector<uint8_t> active(P, 1);
    bool ex_active;
    if ((cfg.estimate_extrinsic && frame_count == W && norm(Vs[0]) > 0.2) || openExEstimation) { openExEstimation = true; ex_active = true; }
    else ex_active = false;
    // mirror of the product's deviation 15 for its parity tests (reference_quirks bit 2, NOT a reference behaviour): the extrinsic is held
    // constant in a solve that carries relocalisation factors
    if (relo_on && (cfg.reference_quirks & 4)) ex_active = false;
    bool td_active = cfg.estimate_td && !(norm(Vs[0]) < 0.2);
    for (int d = 0; d < 6; d++) active[oE + d] = ex_active;
    active[oT] = td_active;
    if (!cfg.use_imu) {   // :1178-1185, 1204: no speed-bias / td blocks in the problem, the oldest pose is constant
        for (int d = 0; d < 6; d++) active[d] = 0;
        for (int a = 6 * (W + 1); a < 15 * (W + 1); a++) active[a] = 0;
        active[oT] = 0;
    }
    std::vector<int> act;
    for (int i = 0; i < P; i++) if (active[i]) act.push_back(i);
    const int Pa = (int)act.size();
    std::vector<int> lact;
    for (int k = 0; k < F; k++) if (!lms[k].is_const) lact.push_back(k);
    const int Fa = (int)lact.size();

    // current point
    double pose[MAXW + 1][7], sb[MAXW + 1][9], ex[7], tdv = cfg.estimate_td ? para_Td : td;
    std::memcpy(pose, para_Pose, sizeof(pose)); std::memcpy(sb, para_SpeedBias, sizeof(sb)); std::memcpy(ex, para_Ex_Pose, sizeof(ex));
    std::vector<double> feat = para_Feature;
    double relo[7], crelo[7];
    std::memcpy(relo, relo_Pose, sizeof(relo));
    auto plus_pose = [](double *x, const double *d) {  // PoseLocalParameterization::Plus
        x[0] += d[0]; x[1] += d[1]; x[2] += d[2];
        Q q(x[6], x[3], x[4], x[5]);
        Q r = normalized(q * deltaQ(V3(d[3], d[4], d[5])));
        x[3] = r.x; x[4] = r.y; x[5] = r.z; x[6] = r.w;
    };
    if (constrained) {
        // TrustRegionMinimizer::IterationZero on a bounds-constrained program: x <- Plus(x, 0), i.e. every non-constant block goes through its
        // local parameterisation once (the quaternions are re-normalised) and is PROJECTED onto the box (ParameterBlock::Plus) before the
        // first evaluation -- a depth-less landmark triangulated nearer than DEPTH_MAX_DIST / 2 starts the solve on the bound
        const double zero6[6] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k <= W; k++) if (cfg.use_imu || k > 0) plus_pose(pose[k], zero6);
        if (ex_active) plus_pose(ex, zero6);
        if (relo_on) plus_pose(relo, zero6);
        for (int li = 0; li < F; li++)
            if (!lms[li].is_const && feat[li] > lms[li].ub) { feat[li] = lms[li].ub; bound_clamps++; }
    }

    NormalEq ne, ne2;
    build_normal_eq(*this, pose, sb, ex, tdv, feat, lms, ne, true, relo_on ? relo : nullptr);
    last_stats = SolveStats();
    last_stats.initial_cost = ne.cost;
    last_stats.n_landmarks = F; last_stats.n_var_landmarks = Fa;

    // Jacobi column scaling, computed once at the initial point: 1/(1+||J_j||)
    std::vector<double> sp(Pa), sl(Fa);
    for (int a = 0; a < Pa; a++) sp[a] = 1.0 / (1.0 + std::sqrt(ne.H(act[a], act[a])));
    for (int k = 0; k < Fa; k++) sl[k] = 1.0 / (1.0 + std::sqrt(ne.Hll[lact[k]]));

    double radius = 1e4, mu = 1e-8;
    int test_fail = (cfg.reference_quirks >> 8) & 15;   // test hook (vio_abi.h VIO_TEST_CHOL_FAIL_SHIFT): factorisations reported as failed
    bool reuse = false;
    int invalid = 0;
    // scaled reduced quantities
    Mat Hs(Pa, Pa), Hpls(Fa, Pa);
    std::vector<double> gs(Pa), gls(Fa), Hlls(Fa), dgp(Pa), dgl(Fa), gradp(Pa), gradl(Fa), gnp(Pa), gnl(Fa);
    double alpha = 0, dogleg_norm = 0;
    double cost = ne.cost;
    auto gmax = [&]() {
        double m = 0;
        for (int a = 0; a < Pa; a++) m = std::max(m, std::fabs(ne.g[act[a]]));
        for (int k = 0; k < Fa; k++) m = std::max(m, std::fabs(ne.gl[lact[k]]));
        return m;
    };
    int iter = 0;
    if (gmax() > 1e-10)
    for (iter = 1; iter <= cfg.max_iterations; iter++) {
        last_stats.iterations = iter;
        if (!reuse) {
            for (int a = 0; a < Pa; a++) {
                gs[a] = sp[a] * ne.g[act[a]];
                for (int b = 0; b < Pa; b++) Hs(a, b) = sp[a] * sp[b] * ne.H(act[a], act[b]);
            }
            for (int k = 0; k < Fa; k++) {
                gls[k] = sl[k] * ne.gl[lact[k]];
                Hlls[k] = sl[k] * sl[k] * ne.Hll[lact[k]];
                for (int a = 0; a < Pa; a++) Hpls(k, a) = sl[k] * sp[a] * ne.Hpl(lact[k], act[a]);
            }
            for (int a = 0; a < Pa; a++) dgp[a] = std::sqrt(std::min(std::max(Hs(a, a), 1e-6), 1e32));
            for (int k = 0; k < Fa; k++) dgl[k] = std::sqrt(std::min(std::max(Hlls[k], 1e-6), 1e32));
            double g2 = 0;
            for (int a = 0; a < Pa; a++) { gradp[a] = gs[a] / dgp[a]; g2 += gradp[a] * gradp[a]; }
            for (int k = 0; k < Fa; k++) { gradl[k] = gls[k] / dgl[k]; g2 += gradl[k] * gradl[k]; }
            // Cauchy point: alpha = |grad|^2 / |J D^-1 grad|^2
            std::vector<double> sgp(Pa), sgl(Fa);
            for (int a = 0; a < Pa; a++) sgp[a] = gradp[a] / dgp[a];
            for (int k = 0; k < Fa; k++) sgl[k] = gradl[k] / dgl[k];
            double jg2 = 0;
            for (int a = 0; a < Pa; a++) { double s = 0; for (int b = 0; b < Pa; b++) s += Hs(a, b) * sgp[b]; jg2 += sgp[a] * s; }
            for (int k = 0; k < Fa; k++) {
                double s = 0;
                for (int a = 0; a < Pa; a++) s += Hpls(k, a) * sgp[a];
                jg2 += 2.0 * sgl[k] * s + sgl[k] * sgl[k] * Hlls[k];
            }
            alpha = g2 / jg2;
            // Gauss-Newton step through the landmark Schur complement, regularised by mu * D^2
            bool ok = false;
            while (mu < 1.0) {
                Mat S(Pa, Pa);
                std::vector<double> rhs(Pa);
                for (int a = 0; a < Pa; a++) { rhs[a] = gs[a]; for (int b = 0; b < Pa; b++) S(a, b) = Hs(a, b); S(a, a) += mu * dgp[a] * dgp[a]; }
                std::vector<double> hll(Fa);
                for (int k = 0; k < Fa; k++) {
                    hll[k] = Hlls[k] + mu * dgl[k] * dgl[k];
                    double inv = 1.0 / hll[k];
                    for (int a = 0; a < Pa; a++) {
                        double f = Hpls(k, a) * inv;
                        if (f == 0.0) continue;
                        rhs[a] -= f * gls[k];
                        for (int b = 0; b < Pa; b++) S(a, b) -= f * Hpls(k, b);
                    }
                }
                const bool forced = test_fail > 0;
                if (forced) test_fail--;
                if (!forced && chol(S)) {
                    chol_solve(S, rhs);
                    bool fin = true;
                    for (int a = 0; a < Pa; a++) fin &= std::isfinite(rhs[a]);
                    if (fin) {
                        for (int a = 0; a < Pa; a++) gnp[a] = -rhs[a] * dgp[a];
                        for (int k = 0; k < Fa; k++) {
                            double s = gls[k];
                            for (int a = 0; a < Pa; a++) s -= Hpls(k, a) * rhs[a];
                            gnl[k] = -(s / hll[k]) * dgl[k];
                        }
                        ok = true;
                        break;
                    }
                }
                mu *= 10.0;
            }
            if (!ok) break;  // linear solver failure at max mu: give up (Ceres: LINEAR_SOLVER_FAILURE)
            if (const char *dump = std::getenv("OVIO_DUMP_SOLVE")) {
                // test hook (tests/test_oracle_numpy_cpu.py): the scaled linear system of this iteration and what the solver derived from
                // it: Pa, Fa, mu, alpha, Hs (Pa^2), Hpls (Fa x Pa), Hlls, gs, gls, dgp, dgl, gnp, gnl as raw doubles (appended)
                if (FILE *fp = std::fopen(dump, "ab")) {
                    double hdr[4] = {(double)Pa, (double)Fa, mu, alpha};
                    std::fwrite(hdr, 8, 4, fp);
                    std::fwrite(Hs.d.data(), 8, (size_t)Pa * Pa, fp);
                    std::fwrite(Hpls.d.data(), 8, (size_t)Fa * Pa, fp);
                    std::fwrite(Hlls.data(), 8, Fa, fp); std::fwrite(gs.data(), 8, Pa, fp); std::fwrite(gls.data(), 8, Fa, fp);
                    std::fwrite(dgp.data(), 8, Pa, fp); std::fwrite(dgl.data(), 8, Fa, fp);
                    std::fwrite(gnp.data(), 8, Pa, fp); std::fwrite(gnl.data(), 8, Fa, fp);
                    std::fclose(fp);
                }
            }
            reuse = true;
        }
        // traditional dogleg in the D-scaled space
        double gnorm = 0, gnn = 0, gdot = 0;
        for (int a = 0; a < Pa; a++) { gnorm += gradp[a] * gradp[a]; gnn += gnp[a] * gnp[a]; gdot += gradp[a] * gnp[a]; }
        for (int k = 0; k < Fa; k++) { gnorm += gradl[k] * gradl[k]; gnn += gnl[k] * gnl[k]; gdot += gradl[k] * gnl[k]; }
        gnorm = std::sqrt(gnorm); gnn = std::sqrt(gnn);
        std::vector<double> stp(Pa), stl(Fa);
        if (gnn <= radius) {
            for (int a = 0; a < Pa; a++) stp[a] = gnp[a];
            for (int k = 0; k < Fa; k++) stl[k] = gnl[k];
            dogleg_norm = gnn;
        } else if (gnorm * alpha >= radius) {
            double f = -(radius / gnorm);
            for (int a = 0; a < Pa; a++) stp[a] = f * gradp[a];
            for (int k = 0; k < Fa; k++) stl[k] = f * gradl[k];
            dogleg_norm = radius;
        } else {
            double b_dot_a = -alpha * gdot;
            double a_sq = (alpha * gnorm) * (alpha * gnorm);
            double bma = a_sq - 2 * b_dot_a + gnn * gnn;
            double c = b_dot_a - a_sq;
            double d = std::sqrt(c * c + bma * (radius * radius - a_sq));
            double beta = (c <= 0) ? (d - c) / bma : (radius * radius - a_sq) / (d + c);
            double n2 = 0;
            for (int a = 0; a < Pa; a++) { stp[a] = (-alpha * (1.0 - beta)) * gradp[a] + beta * gnp[a]; n2 += stp[a] * stp[a]; }
            for (int k = 0; k < Fa; k++) { stl[k] = (-alpha * (1.0 - beta)) * gradl[k] + beta * gnl[k]; n2 += stl[k] * stl[k]; }
            dogleg_norm = std::sqrt(n2);
        }
        for (int a = 0; a < Pa; a++) stp[a] /= dgp[a];
        for (int k = 0; k < Fa; k++) stl[k] /= dgl[k];
        // model cost change = -(step^T g' + 1/2 step^T H' step)
        double lin = 0, quad = 0;
        for (int a = 0; a < Pa; a++) { lin += stp[a] * gs[a]; double s = 0; for (int b = 0; b < Pa; b++) s += Hs(a, b) * stp[b]; quad += stp[a] * s; }
        for (int k = 0; k < Fa; k++) {
            lin += stl[k] * gls[k];
            double s = 0;
            for (int a = 0; a < Pa; a++) s += Hpls(k, a) * stp[a];
            quad += 2.0 * stl[k] * s + stl[k] * stl[k] * Hlls[k];
        }
        double model_change = -(lin + 0.5 * quad);
        if (!(model_change > 0)) {
            if (++invalid >= 5) break;
            mu *= 10.0;
            reuse = false;
            continue;
        }
        invalid = 0;
        // candidate = Plus(x, alpha * step .* scale), projected onto the box (ParameterBlock::Plus clamps to the bounds)
        double cpose[MAXW + 1][7], csb[MAXW + 1][9], cex[7], ctd = tdv;
        std::vector<double> cfeat;
        std::vector<double> delta(P, 0.0), dlm(Fa);
        for (int a = 0; a < Pa; a++) delta[act[a]] = stp[a] * sp[a];
        for (int k = 0; k < Fa; k++) dlm[k] = stl[k] * sl[k];
        auto make_candidate = [&](double alpha) {
            std::memcpy(cpose, pose, sizeof(pose)); std::memcpy(csb, sb, sizeof(sb)); std::memcpy(cex, ex, sizeof(ex));
            ctd = tdv;
            cfeat = feat;
            std::vector<double> sd(P);
            for (int a = 0; a < P; a++) sd[a] = alpha * delta[a];   // LineSearchFunction: scaled_direction = x * direction (alpha = 1: delta itself)
            for (int k = 0; k <= W; k++) {
                plus_pose(cpose[k], &sd[6 * k]);
                for (int d = 0; d < 9; d++) csb[k][d] += sd[6 * (W + 1) + 9 * k + d];
            }
            if (ex_active) plus_pose(cex, &sd[oE]);
            if (td_active) ctd += sd[oT];
            std::memcpy(crelo, relo, sizeof(relo));
            if (relo_on) plus_pose(crelo, &sd[oR]);
            for (int k = 0; k < Fa; k++) {
                int li = lact[k];
                cfeat[li] += alpha * dlm[k];
                if (cfeat[li] > lms[li].ub) { cfeat[li] = lms[li].ub; bound_clamps++; }  // projection onto the box
            }
        };
        bool have_ne2 = false;
        if (constrained) {
            // TrustRegionMinimizer::DoLineSearch -> ArmijoLineSearch::DoSearch (line_search.cc; Solver::Options defaults: CUBIC interpolation,
            // sufficient decrease 1e-4, contraction in [1e-3, 0.6], at most 20 iterations, min step 1e-9): the step is shortened until
            // f(Plus(x, a delta)) <= f(x) + 1e-4 a g^T delta; on success delta *= a, on failure delta is left as it was.  The model cost
            // change and the dogleg norm of the FULL step keep driving the trust-region logic below, as in Ceres.
            double g0 = 0, dmax = 0;   // initial_gradient = gradient . delta; LineSearchFunction::DirectionInfinityNorm
            for (int a = 0; a < Pa; a++) { g0 += delta[act[a]] * ne.g[act[a]]; dmax = std::max(dmax, std::fabs(delta[act[a]])); }
            for (int k = 0; k < Fa; k++) { g0 += dlm[k] * ne.gl[lact[k]]; dmax = std::max(dmax, std::fabs(dlm[k])); }
            LsSample lower = {0.0, cost, g0, 1}, previous = {0, 0, 0, 0}, current = {0, 0, 0, 0};
            auto ls_eval = [&](double a) {
                make_candidate(a);
                build_normal_eq(*this, cpose, csb, cex, ctd, cfeat, lms, ne2, true, relo_on ? crelo : nullptr);
                line_search_evals++;
                double gd = 0;
                for (int q = 0; q < Pa; q++) gd += delta[act[q]] * ne2.g[act[q]];
                for (int k = 0; k < Fa; k++) gd += dlm[k] * ne2.gl[lact[k]];
                current.x = a; current.value = ne2.cost; current.gradient = gd;
                current.valid = std::isfinite(ne2.cost) && std::isfinite(gd);
            };
            // test hook (tests/test_oracle_linesearch_cpu.py): every search as raw doubles -- header (-1, cost, g0, dmax, candidates clamped at
            // alpha = 1), one row (alpha, value, gradient, valid, 0) per trial, trailer (-2, chosen alpha, success, 0, 0)
            FILE *lsfp = nullptr;
            if (const char *dump = std::getenv("OVIO_DUMP_LS")) lsfp = std::fopen(dump, "ab");
            const long clamps_before = bound_clamps;
            ls_eval(1.0);
            if (lsfp) { double row[5] = {-1.0, cost, g0, dmax, (double)(bound_clamps - clamps_before)}; std::fwrite(row, 8, 5, lsfp); }
            auto dump_trial = [&]() { if (lsfp) { double row[5] = {current.x, current.value, current.gradient, (double)current.valid, 0.0}; std::fwrite(row, 8, 5, lsfp); } };
            dump_trial();
            int ls_it = 0;
            bool ls_ok = true;
            while (!current.valid || current.value > cost + 1e-4 * g0 * current.x) {
                if (++ls_it >= 20) { ls_ok = false; break; }
                double ls_ws[96];
                const double a = ls_next_step(lower, previous, current, 1e-3 * current.x, 0.6 * current.x, ls_ws);
                if (a * dmax < 1e-9) { ls_ok = false; break; }
                previous = current;
                ls_eval(a);
                dump_trial();
                line_search_contractions++;
            }
            if (lsfp) { double row[5] = {-2.0, ls_ok ? current.x : 1.0, ls_ok ? 1.0 : 0.0, 0.0, 0.0}; std::fwrite(row, 8, 5, lsfp); std::fclose(lsfp); }
            // success: the candidate IS the last sample (same vector alpha * delta, same Plus); failure: the full step, evaluated again
            have_ne2 = ls_ok || current.x == 1.0;
        }
        if (!have_ne2) {
            make_candidate(1.0);
            build_normal_eq(*this, cpose, csb, cex, ctd, cfeat, lms, ne2, true, relo_on ? crelo : nullptr);
        }
        // parameter tolerance
        double xn = 0, dn = 0;
        for (int k = 0; k <= W; k++) {
            for (int d = 0; d < 7; d++) { xn += pose[k][d] * pose[k][d]; double t = pose[k][d] - cpose[k][d]; dn += t * t; }
            for (int d = 0; d < 9; d++) { xn += sb[k][d] * sb[k][d]; double t = sb[k][d] - csb[k][d]; dn += t * t; }
        }
        if (ex_active) for (int d = 0; d < 7; d++) { xn += ex[d] * ex[d]; double t = ex[d] - cex[d]; dn += t * t; }
        if (td_active) { xn += tdv * tdv; dn += (tdv - ctd) * (tdv - ctd); }
        if (relo_on) for (int d = 0; d < 7; d++) { xn += relo[d] * relo[d]; double t = relo[d] - crelo[d]; dn += t * t; }
        for (int k = 0; k < Fa; k++) { int li = lact[k]; xn += feat[li] * feat[li]; double t = feat[li] - cfeat[li]; dn += t * t; }
        if (std::sqrt(dn) <= 1e-8 * (std::sqrt(xn) + 1e-8)) break;
        if (std::fabs(cost - ne2.cost) <= 1e-6 * cost) break;
        double rel = (cost - ne2.cost) / model_change;
        if (rel > 1e-3) {
            std::memcpy(pose, cpose, sizeof(pose)); std::memcpy(sb, csb, sizeof(sb)); std::memcpy(ex, cex, sizeof(ex));
            tdv = ctd;
            feat = cfeat;
            std::memcpy(relo, crelo, sizeof(relo));
            std::swap(ne, ne2);
            cost = ne.cost;
            last_stats.successful++;
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = std::max(radius, 3.0 * dogleg_norm);
            mu = std::max(1e-8, 2.0 * mu / 10.0);
            reuse = false;
            if (gmax() <= 1e-10) break;
        } else {
            radius *= 0.5;
            reuse = true;
        }
    }
    last_stats.final_cost = cost;
    std::memcpy(para_Pose, pose, sizeof(pose)); std::memcpy(para_SpeedBias, sb, sizeof(sb)); std::memcpy(para_Ex_Pose, ex, sizeof(ex));
    if (cfg.estimate_td) para_Td = tdv;
    para_Feature = feat;
    if (relo_on) std::memcpy(relo_Pose, relo, sizeof(relo));
}

// ------------------------------------------------------------------ marginalisation (marginalization_factor.cpp:181-315)
void marg_finish(Estimator &e, Mat &A, std::vector<double> &b, int m, int n) {
    const double eps = 1e-8;
    if ((e.deviations & ODEV_LANDMARK_ELIM) && m > 15) {
        // deviation 10 (attribution experiment): the landmark part of the marginalised block is diagonal (one inverse depth per landmark, no
        // landmark-landmark terms), so it is eliminated analytically -- 1/d where d > eps, dropped otherwise -- and only the 15x15 pose /
        // speed-bias block that remains goes through the truncated eigen-decomposition below.  Identical to the full decomposition when
        // nothing is truncated.
        const int F = m - 15, q = 15 + n;
        auto qi = [&](int a) { return a < 15 ? a : a + F; };   // index of reduced row a in the full system
        Mat A2(q, q);
        std::vector<double> b2(q);
        for (int a = 0; a < q; a++) {
            double sb = b[qi(a)];
            for (int l = 0; l < F; l++) { const double d = A(15 + l, 15 + l); if (d > eps) sb -= A(qi(a), 15 + l) * (b[15 + l] / d); }
            b2[a] = sb;
            for (int c = 0; c < q; c++) {
                double sa = A(qi(a), qi(c));
                for (int l = 0; l < F; l++) { const double d = A(15 + l, 15 + l); if (d > eps) sa -= A(qi(a), 15 + l) * A(15 + l, qi(c)) / d; }
                A2(a, c) = sa;
            }
        }
        A = A2;
        b = b2;
        m = 15;
    }
    // Amm^-1 via symmetric eigen-decomposition with truncation
    Mat Amm(m, m);
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Amm(i, j) = 0.5 * (A(i, j) + A(j, i));
    Mat Ainv(m, m);
    bool direct = false;
    if ((e.deviations & ODEV_CHOL_PINV) && m <= 16) {
        // attribution experiment: A^-1 = L^-T L^-1 from a Cholesky factorisation, accepted when 1 / |A^-1|_F > 1e-6 (then no eigenvalue is
        // anywhere near the 1e-8 cut and the pseudo-inverse is the inverse)
        Mat L = Amm;
        if (chol(L)) {
            Mat Li(m, m);
            for (int c0 = 0; c0 < m; c0++)
                for (int i = 0; i < m; i++) {
                    double sacc = (i == c0) ? 1.0 : 0.0;
                    for (int k = c0; k < i; k++) sacc -= L(i, k) * Li(k, c0);
                    Li(i, c0) = i < c0 ? 0.0 : sacc / L(i, i);
                }
            double fro = 0;
            for (int i = 0; i < m; i++)
                for (int j = 0; j < m; j++) {
                    double sacc = 0;
                    for (int k = (i > j ? i : j); k < m; k++) sacc += Li(k, i) * Li(k, j);
                    Ainv(i, j) = sacc;
                    fro += sacc * sacc;
                }
            direct = std::isfinite(fro) && fro > 0.0 && 1.0 / std::sqrt(fro) > 1e-6;
        }
    }
    if (!direct) {
        std::vector<double> w;
        Mat V;
        sym_eig(Amm, w, V);
        for (int i = 0; i < m; i++)
            for (int j = 0; j < m; j++) {
                double s = 0;
                for (int k = 0; k < m; k++) if (w[k] > eps) s += V(i, k) * V(j, k) / w[k];
                Ainv(i, j) = s;
            }
    }
    // Arm * Amm_inv
    Mat T1(n, m);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) {
            double s = 0;
            for (int k = 0; k < m; k++) s += A(m + i, k) * Ainv(k, j);
            T1(i, j) = s;
        }
    Mat Ar(n, n);
    std::vector<double> br(n);
    for (int i = 0; i < n; i++) {
        double s = b[m + i];
        for (int k = 0; k < m; k++) s -= T1(i, k) * b[k];
        br[i] = s;
        for (int j = 0; j < n; j++) {
            double t = A(m + i, m + j);
            for (int k = 0; k < m; k++) t -= T1(i, k) * A(k, m + j);
            Ar(i, j) = t;
        }
    }
    Mat As(n, n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) As(i, j) = 0.5 * (Ar(i, j) + Ar(j, i));
    if (const char *dump = std::getenv("OVIO_DUMP_PRIOR")) {
        // test hook (tests/test_oracle_numpy_cpu.py): append m, n, A ((m+n)^2), b (m+n), A_s (n x n), b_r (n) as raw doubles
        if (FILE *fp = std::fopen(dump, "ab")) {
            double hdr[2] = {(double)m, (double)n};
            std::fwrite(hdr, 8, 2, fp);
            std::fwrite(A.d.data(), 8, (size_t)(m + n) * (m + n), fp);
            std::fwrite(b.data(), 8, m + n, fp);
            std::fwrite(As.d.data(), 8, (size_t)n * n, fp);
            std::fwrite(br.data(), 8, n, fp);
            std::fclose(fp);
        }
    }
    if (e.deviations & ODEV_QUADRATIC_PRIOR) {
        // deviation 13 (attribution experiment): the prior kept as (A, b, c0) = (sym(A_r), b_r, b^T A^+ b); c0 = |L^-1 b|^2 with
        // L L^T = A + delta I, delta = 64 eps n max|diag| (lifts the gauge directions off the round-off floor)
        e.prior_n = n;
        e.prior_A = As;
        e.prior_b = br;
        double dmax = 0;
        for (int i = 0; i < n; i++) dmax = std::max(dmax, std::fabs(Ar(i, i)));
        const double delta = 64.0 * 2.220446049250313e-16 * (double)n * dmax;
        Mat L = As;
        for (int i = 0; i < n; i++) L(i, i) += delta;
        double c0 = 0;
        if (dmax > 0 && chol(L)) {
            std::vector<double> y(br);
            for (int i = 0; i < n; i++) {
                double sacc = y[i];
                for (int k = 0; k < i; k++) sacc -= L(i, k) * y[k];
                y[i] = sacc / L(i, i);
                c0 += y[i] * y[i];
            }
            if (!std::isfinite(c0)) c0 = 0;
        }
        e.prior_c0 = c0;
        return;
    }
    std::vector<double> w2;
    Mat V2;
    sym_eig(As, w2, V2);
    e.prior_n = n;
    e.prior_J = Mat(n, n);
    e.prior_r.assign(n, 0.0);
    for (int k = 0; k < n; k++) {
        double S = w2[k] > eps ? w2[k] : 0.0, Sinv = w2[k] > eps ? 1.0 / w2[k] : 0.0;
        double ss = std::sqrt(S), si = std::sqrt(Sinv);
        double vb = 0;
        for (int i = 0; i < n; i++) { e.prior_J(k, i) = ss * V2(i, k); vb += V2(i, k) * br[i]; }
        e.prior_r[k] = si * vb;
    }
    // control experiment only (tests/oracle_control.py): OVIO_PERTURB_EPS = eps scales the new prior's Jacobian by (1 + eps) --
    // a perturbation of KNOWN relative size at the place where implementations of this algorithm differ most (the eigen-decomposition of a
    // matrix of norm 1e10 with a 1e-8 cut-off, deviation 12).  Unset in every other use.
    const char *pv = std::getenv("OVIO_PERTURB_EPS");
    const double perturb = pv ? std::atof(pv) : 0.0;
    if (perturb != 0.0) for (double &v : e.prior_J.d) v *= 1.0 + perturb;
}

void Estimator::marginalize_old() {  // estimator.cpp:1376-1502
    const int n = 6 * W + 16;
    // m blocks: pose0(6) sb0(9) then landmarks with start_frame==0 (list order)
    std::vector<LmRef> lms;
    {
        int idx = -1;
        for (auto &l : feature) {
            if (l.is_dynamic) continue;
            l.used_num = (int)l.obs.size();
            if (!in_problem(l, W)) continue;
            ++idx;
            if (l.start_frame != 0) continue;
            LmRef r; r.l = &l; r.idx = idx; r.is_const = false; r.ub = DBL_MAX;
            lms.push_back(r);
        }
    }
    const int m = 15 + (int)lms.size();
    const int pos = m + n;
    Mat A(pos, pos);
    std::vector<double> b(pos, 0.0);
    std::vector<uint8_t> present(W + 3, 0);
    const int rP = m, rS = m + 6 * W, rE = m + 6 * W + 9, rT = m + 6 * W + 15;
    auto poseIdx = [&](int k) { return k == 0 ? 0 : rP + 6 * (k - 1); };  // window pose k -> A index
    double tdv = cfg.estimate_td ? para_Td : td;
    auto accumulate = [&](int nr, int nc, const double *J, const double *r, const int *idx) {
        for (int a = 0; a < nc; a++) {
            double gsum = 0;
            for (int k = 0; k < nr; k++) gsum += J[k * nc + a] * r[k];
            b[idx[a]] += gsum;
            for (int c = 0; c < nc; c++) {
                double s = 0;
                for (int k = 0; k < nr; k++) s += J[k * nc + a] * J[k * nc + c];
                A(idx[a], idx[c]) += s;
            }
        }
    };
    if (has_prior) {
        int pn = prior_n;
        std::vector<double> dx(pn, 0.0);
        std::vector<int> map(pn, 0);
        for (int k = 0; k < W; k++) { pose_dx(para_Pose[k], &prior_x0[k * 7], &dx[6 * k]); for (int d = 0; d < 6; d++) map[6 * k + d] = poseIdx(k) + d; }
        for (int d = 0; d < 9; d++) { dx[6 * W + d] = para_SpeedBias[0][d] - prior_x0[W * 7 + d]; map[6 * W + d] = 6 + d; }
        pose_dx(para_Ex_Pose, &prior_x0[W * 7 + 9], &dx[6 * W + 9]);
        for (int d = 0; d < 6; d++) map[6 * W + 9 + d] = rE + d;
        dx[6 * W + 15] = tdv - prior_x0[W * 7 + 16];
        map[6 * W + 15] = rT;
        for (int k = 0; k < W; k++) if (!prior_present[k]) for (int d = 0; d < 6; d++) dx[6 * k + d] = 0;
        if (!prior_present[W]) for (int d = 0; d < 9; d++) dx[6 * W + d] = 0;
        if (!prior_present[W + 1]) for (int d = 0; d < 6; d++) dx[6 * W + 9 + d] = 0;
        if (!prior_present[W + 2]) dx[6 * W + 15] = 0;
        std::vector<double> r, pg;
        prior_gradient(*this, dx, pg, r);
        for (int a = 0; a < pn; a++) {
            b[map[a]] += pg[a];
            for (int c = 0; c < pn; c++) A(map[a], map[c]) += prior_hessian(*this, a, c);
        }
        for (int k = 1; k < W; k++) if (prior_present[k]) present[k - 1] = 1;
        if (prior_present[W + 1]) present[W + 1] = 1;
        if (prior_present[W + 2]) present[W + 2] = 1;
    }
    if (cfg.use_imu && pre_integrations[1]->sum_dt < 10.0) {
        double r[15], Ji[15 * 7], Jsi[15 * 9], Jj[15 * 7], Jsj[15 * 9];
        eval_imu(*pre_integrations[1], g, para_Pose[0], para_SpeedBias[0], para_Pose[1], para_SpeedBias[1], r, Ji, Jsi, Jj, Jsj);
        double J[15 * 30];
        int idx[30];
        for (int k = 0; k < 15; k++) {
            for (int d = 0; d < 6; d++) { J[k * 30 + d] = Ji[k * 7 + d]; J[k * 30 + 15 + d] = Jj[k * 7 + d]; }
            for (int d = 0; d < 9; d++) { J[k * 30 + 6 + d] = Jsi[k * 9 + d]; J[k * 30 + 21 + d] = Jsj[k * 9 + d]; }
        }
        for (int d = 0; d < 6; d++) { idx[d] = d; idx[15 + d] = poseIdx(1) + d; }
        for (int d = 0; d < 9; d++) { idx[6 + d] = 6 + d; idx[21 + d] = rS + d; }
        accumulate(15, 30, J, r, idx);
        present[0] = 1;
        present[W] = 1;
    }
    for (size_t lq = 0; lq < lms.size(); lq++) {
#ifdef ORACLE_PERTURB_ORDER
        const size_t li = lms.size() - 1 - lq;   // control experiment: opposite accumulation order (see build_normal_eq)
#else
        const size_t li = lq;
#endif
        Landmark &l = *lms[li].l;
        double inv_dep = para_Feature[lms[li].idx];
        for (int k = 1; k < (int)l.obs.size(); k++) {
            int imu_j = k;
            double r[2], Ji[14], Jj[14], Je[14], Jl[2], Jt[2];
            eval_projection(cfg, para_Pose[0], para_Pose[imu_j], para_Ex_Pose, inv_dep, tdv, l.obs[0], l.obs[k], cfg.estimate_td != 0, r, Ji, Jj, Je, Jl, Jt);
            // ResidualBlockInfo::Evaluate loss re-weighting (marginalization_factor.cpp:39-72) with CauchyLoss(1.0)
            double sq_norm = r[0] * r[0] + r[1] * r[1];
            double rho1 = 1.0 / (1.0 + sq_norm), rho2 = -rho1 * rho1;
            double sqrt_rho1 = std::sqrt(rho1), residual_scaling, alpha_sq_norm;
            if (sq_norm == 0.0 || rho2 <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
            else { double D = 1.0 + 2.0 * sq_norm * rho2 / rho1; double al = 1.0 - std::sqrt(D); residual_scaling = sqrt_rho1 / (1 - al); alpha_sq_norm = al / sq_norm; }
            double J[2 * 20];
            int idx[20];
            for (int a = 0; a < 2; a++) {
                for (int d = 0; d < 6; d++) { J[a * 20 + d] = Ji[a * 7 + d]; J[a * 20 + 6 + d] = Jj[a * 7 + d]; J[a * 20 + 12 + d] = Je[a * 7 + d]; }
                J[a * 20 + 18] = Jt[a];
                J[a * 20 + 19] = Jl[a];
            }
            for (int c = 0; c < 20; c++) {
                double rtJ = r[0] * J[c] + r[1] * J[20 + c];
                J[c] = sqrt_rho1 * (J[c] - alpha_sq_norm * r[0] * rtJ);
                J[20 + c] = sqrt_rho1 * (J[20 + c] - alpha_sq_norm * r[1] * rtJ);
            }
            r[0] *= residual_scaling; r[1] *= residual_scaling;
            for (int d = 0; d < 6; d++) { idx[d] = d; idx[6 + d] = poseIdx(imu_j) + d; idx[12 + d] = rE + d; }
            idx[18] = rT;
            idx[19] = 15 + (int)li;
            if (cfg.estimate_td) accumulate(2, 20, J, r, idx);
            else {
                // ProjectionFactor has no td block: drop column 18
                double J2[2 * 19];
                int idx2[19];
                for (int a = 0; a < 2; a++) for (int c = 0, o = 0; c < 20; c++) if (c != 18) J2[a * 19 + o++] = J[a * 20 + c];
                for (int c = 0, o = 0; c < 20; c++) if (c != 18) idx2[o++] = idx[c];
                accumulate(2, 19, J2, r, idx2);
            }
            present[imu_j - 1] = 1;
            present[W + 1] = 1;
            if (cfg.estimate_td) present[W + 2] = 1;
        }
    }
    marg_finish(*this, A, b, m, n);
    // keep_block_data: values at marginalisation time, shifted i -> i-1 (addr_shift, estimator.cpp:1483-1497)
    prior_x0.assign(W * 7 + 17, 0.0);
    for (int k = 1; k <= W; k++) for (int d = 0; d < 7; d++) prior_x0[(k - 1) * 7 + d] = para_Pose[k][d];
    for (int d = 0; d < 9; d++) prior_x0[W * 7 + d] = para_SpeedBias[1][d];
    for (int d = 0; d < 7; d++) prior_x0[W * 7 + 9 + d] = para_Ex_Pose[d];
    prior_x0[W * 7 + 16] = tdv;
    prior_present = present;
    has_prior = true;
}

void Estimator::marginalize_second_new() {  // estimator.cpp:1503-1574
    if (!(has_prior && prior_present[W - 1])) return;
    const int n = 6 * W + 16;
    const int m = 6, pos = m + n;
    Mat A(pos, pos);
    std::vector<double> b(pos, 0.0);
    double tdv = cfg.estimate_td ? para_Td : td;
    int pn = prior_n;
    std::vector<double> dx(pn, 0.0);
    std::vector<int> map(pn, 0);
    // prior slot k -> new index: k == W-1 -> m block; others keep their canonical slot (new slot W-1 stays empty)
    for (int k = 0; k < W; k++) {
        pose_dx(para_Pose[k], &prior_x0[k * 7], &dx[6 * k]);
        for (int d = 0; d < 6; d++) map[6 * k + d] = (k == W - 1) ? d : m + 6 * k + d;
    }
    for (int d = 0; d < 9; d++) { dx[6 * W + d] = para_SpeedBias[0][d] - prior_x0[W * 7 + d]; map[6 * W + d] = m + 6 * W + d; }
    pose_dx(para_Ex_Pose, &prior_x0[W * 7 + 9], &dx[6 * W + 9]);
    for (int d = 0; d < 6; d++) map[6 * W + 9 + d] = m + 6 * W + 9 + d;
    dx[6 * W + 15] = tdv - prior_x0[W * 7 + 16];
    map[6 * W + 15] = m + 6 * W + 15;
    for (int k = 0; k < W; k++) if (!prior_present[k]) for (int d = 0; d < 6; d++) dx[6 * k + d] = 0;
    if (!prior_present[W]) for (int d = 0; d < 9; d++) dx[6 * W + d] = 0;
    if (!prior_present[W + 1]) for (int d = 0; d < 6; d++) dx[6 * W + 9 + d] = 0;
    if (!prior_present[W + 2]) dx[6 * W + 15] = 0;
    std::vector<double> r, pg;
    prior_gradient(*this, dx, pg, r);
    for (int a = 0; a < pn; a++) {
        b[map[a]] += pg[a];
        for (int c = 0; c < pn; c++) A(map[a], map[c]) += prior_hessian(*this, a, c);
    }
    marg_finish(*this, A, b, m, n);
    std::vector<double> x0(W * 7 + 17, 0.0);
    for (int k = 0; k < W - 1; k++) for (int d = 0; d < 7; d++) x0[k * 7 + d] = para_Pose[k][d];
    for (int d = 0; d < 7; d++) x0[(W - 1) * 7 + d] = para_Pose[W][d];  // slot W-1 <- pose W (absent)
    for (int d = 0; d < 9; d++) x0[W * 7 + d] = para_SpeedBias[0][d];
    for (int d = 0; d < 7; d++) x0[W * 7 + 9 + d] = para_Ex_Pose[d];
    x0[W * 7 + 16] = tdv;
    prior_x0 = x0;
    prior_present[W - 1] = 0;
}

void Estimator::optimization() {  // estimator.cpp:1161-1578
    vector2double();
    solve();
    double2vector();
    if (frame_count < W) return;
    if (marginalization_flag == 0) {
        vector2double();
        marginalize_old();
    } else {
        if (has_prior && prior_present[W - 1]) {
            vector2double();
            marginalize_second_new();
        }
    }
}

// ------------------------------------------------------------------ window management
void Estimator::slideWindow() {  // estimator.cpp:1580-1689
    if (marginalization_flag == 0) {
        const double t_0 = Headers[0];
        back_R0 = Rs[0];
        back_P0 = Ps[0];
        if (frame_count == W) {
            if (!all_image_frame.empty()) {  // :1633-1644: drop every image frame up to and including the old frame 0
                auto it_0 = all_image_frame.find(t_0);
                if (it_0 != all_image_frame.end()) {
                    for (auto it = all_image_frame.begin(); it != it_0; ++it) delete it->second.pre_integration;
                    delete it_0->second.pre_integration;
                    all_image_frame.erase(all_image_frame.begin(), std::next(it_0));
                }
            }
            for (int i = 0; i < W; i++) {
                Headers[i] = Headers[i + 1];
                std::swap(Ps[i], Ps[i + 1]);
                std::swap(Rs[i], Rs[i + 1]);
                if (cfg.use_imu) std::swap(pre_integrations[i], pre_integrations[i + 1]);
                std::swap(Vs[i], Vs[i + 1]);
                std::swap(Bas[i], Bas[i + 1]);
                std::swap(Bgs[i], Bgs[i + 1]);
            }
            Headers[W] = Headers[W - 1];
            Ps[W] = Ps[W - 1]; Rs[W] = Rs[W - 1]; Vs[W] = Vs[W - 1]; Bas[W] = Bas[W - 1]; Bgs[W] = Bgs[W - 1];
            if (cfg.use_imu) {
                delete pre_integrations[W];
                pre_integrations[W] = new Integration(cfg, acc_0, gyr_0, Bas[W], Bgs[W]);
            }
            slideWindowOld();
        }
    } else {
        if (frame_count == W) {
            Headers[W - 1] = Headers[W];
            Ps[W - 1] = Ps[W]; Rs[W - 1] = Rs[W];
            if (cfg.use_imu) {
                Integration *last = pre_integrations[W];
                for (size_t i = 0; i < last->dt_buf.size(); i++)
                    pre_integrations[W - 1]->push_back(last->dt_buf[i], last->acc_buf[i], last->gyr_buf[i]);
                Vs[W - 1] = Vs[W]; Bas[W - 1] = Bas[W]; Bgs[W - 1] = Bgs[W];
                delete pre_integrations[W];
                pre_integrations[W] = new Integration(cfg, acc_0, gyr_0, Bas[W], Bgs[W]);
            }
            slideWindowNew();
        }
    }
}
void Estimator::slideWindowNew() { removeFront(frame_count); }  // :1692-1696
void Estimator::slideWindowOld() {                               // :1699-1716
    if (solver_flag == 1) {
        M3 R0 = back_R0 * ric, R1 = Rs[0] * ric;
        V3 P0 = back_P0 + back_R0 * tic, P1 = Ps[0] + Rs[0] * tic;
        removeBackShiftDepth(R0, P0, R1, P1);
    } else
        removeBack();
}

void Estimator::movingConsistencyCheck() {  // estimator.cpp:1944-2009
    for (auto &l : feature) {
        l.used_num = (int)l.obs.size();
        if (!in_problem(l, W)) continue;
        double depth = l.estimated_depth;
        if (depth < 0) continue;
        double err = 0, err3D = 0;
        int errCnt = 0;
        int imu_i = l.start_frame;
        V3 pts_i(l.obs[0].x, l.obs[0].y, l.obs[0].z);
        for (int k = 1; k < (int)l.obs.size(); k++) {
            int imu_j = imu_i + k;
            V3 pts_j(l.obs[k].x, l.obs[k].y, l.obs[k].z);
            V3 pts_w = Rs[imu_i] * (ric * (depth * pts_i) + tic) + Ps[imu_i];
            V3 pts_cj = T(ric) * (T(Rs[imu_j]) * (pts_w - Ps[imu_j]) - tic);
            double rx = pts_cj.x / pts_cj.z - pts_j.x, ry = pts_cj.y / pts_cj.z - pts_j.y;
            err += std::sqrt(rx * rx + ry * ry);
            err3D += norm(pts_cj - pts_j) / depth;  // sic: metric point vs normalised observation (:1956-1963)
            errCnt++;
        }
        if (errCnt > 0) l.is_dynamic = (cfg.focal_length * err / errCnt > 10 || err3D / errCnt > 2.0);
    }
}
bool Estimator::failureDetection() {  // estimator.cpp:1113-1159
    if (norm(Bas[W]) > 2.5) return true;
    if (norm(Bgs[W]) > 1.0) return true;
    V3 tmp_P = Ps[W];
    if (norm(tmp_P - last_P) > 5) return true;
    if (std::fabs(tmp_P.z - last_P.z) > 1) return true;
    return false;
}

int Estimator::processImage(std::map<int, std::array<double, 7>> &image, const uint16_t *depth, double stamp) {  // :156-374
    depth_img = depth;  // FeatureManager::inputDepth
    double curTime = stamp + td;
    if (cfg.use_imu && !IMUAvailable(curTime)) return 1;  // upstream busy-waits (:178-183); the adapter keeps the wait loop
    marginalization_flag = addFeatureCheckParallax(frame_count, image, td) ? 0 : 1;
    Headers[frame_count] = stamp;
    if (cfg.use_imu) {
        // getIMUInterval :1910-1942
        std::vector<ImuSample> v;
        while (imu_head < imu_buf.size() && imu_buf[imu_head].t <= prevTime) imu_head++;
        while (imu_head < imu_buf.size() && imu_buf[imu_head].t < curTime) v.push_back(imu_buf[imu_head++]);
        v.push_back(imu_buf[imu_head]);
        if (!initFirstPoseFlag) initFirstIMUPose(v);
        for (size_t i = 0; i < v.size(); i++) {
            double dt;
            if (i == 0) dt = v[i].t - prevTime;
            else if (i == v.size() - 1) dt = curTime - v[i - 1].t;
            else dt = v[i].t - v[i - 1].t;
            processIMU(dt, v[i].acc, v[i].gyr);
        }
        prevTime = curTime;
        if (imu_head > 4096) { imu_buf.erase(imu_buf.begin(), imu_buf.begin() + imu_head); imu_head = 0; }
    }
    if (cfg.dynamic_init && solver_flag == 0) {  // :203-206 (all_image_frame is only read by the dynamic initialisation)
        ImageFrameO fr;
        for (auto &kv : image) fr.points[kv.first] = {kv.second[0], kv.second[1]};
        fr.pre_integration = tmp_pre_integration;
        all_image_frame[stamp] = fr;
        tmp_pre_integration = new Integration(cfg, acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]);
    }
    if (solver_flag == 0 && cfg.dynamic_init) {
        // dynamic initialisation :230-259
        if (frame_count == W) {
            bool result = false;
            if (cfg.estimate_extrinsic != 2 && (stamp - initial_timestamp) > 0.1) {
                init_attempts++;
                result = initialStructure();
                if (!result) init_failures++;
                initial_timestamp = stamp;
            }
            if (result) {
                solver_flag = 1;
                triangulateWithDepth();  // solveOdometry :921-933
                optimization();
                slideWindow();
                removeFailures();
                last_R = Rs[W]; last_P = Ps[W]; last_R0 = Rs[0]; last_P0 = Ps[0];
                for (auto &it : all_image_frame) delete it.second.pre_integration;  // not read again once NON_LINEAR
                all_image_frame.clear();
                delete tmp_pre_integration;
                tmp_pre_integration = nullptr;
            } else
                slideWindow();
        } else
            frame_count++;
    } else if (solver_flag == 0) {
        // static_init / depth branch :260-316
        triangulateWithDepth();
        if (!cfg.use_imu) {   // :300-310
            if (frame_count == W) {
                optimization();
                solver_flag = 1;
                slideWindow();
            }
        } else if (frame_count == W) {
            solveGyroscopeBias();
            for (int j = 0; j <= W; j++) pre_integrations[j]->repropagate(V3(), Bgs[j]);
            optimization();
            latest_Bg = Bgs[frame_count];  // updateLatestStates :1768-1788 (only latest_Bg feeds back into the hot path)
            imu_at_update = imu_buf.size();
            solver_flag = 1;
            slideWindow();
            last_R = Rs[W]; last_P = Ps[W]; last_R0 = Rs[0]; last_P0 = Ps[0];
        }
        if (frame_count < W) {
            frame_count++;
            int p = frame_count - 1;
            Ps[frame_count] = Ps[p]; Vs[frame_count] = Vs[p]; Rs[frame_count] = Rs[p]; Bas[frame_count] = Bas[p]; Bgs[frame_count] = Bgs[p];
        }
    } else {
        if (!cfg.use_imu) initFramePoseByPnP(frame_count);  // :321-322
        triangulateWithDepth();
        optimization();
        movingConsistencyCheck();
        if (failureDetection()) {
            // failure_occur = true; clearState(); setParameter();  -- clearState() resets failure_occur (:103)
            clearState();
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) ric(i, j) = cfg.ric[i * 3 + j];
            tic = V3(cfg.tic[0], cfg.tic[1], cfg.tic[2]);
            td = cfg.td;
            g = V3(0, 0, cfg.g_norm);   // setParameter(): g = G (estimator.cpp:26)
            reboot_count++;
            return 2;
        }
        slideWindow();
        removeFailures();
        last_R = Rs[W]; last_P = Ps[W]; last_R0 = Rs[0]; last_P0 = Ps[0];
        latest_Bg = Bgs[frame_count];
        imu_at_update = imu_buf.size();
    }
    return 0;
}

// ------------------------------------------------------------------ nodelet-side driver
// process_tracker for one frame (estimator_nodelet.cpp:234-393) after the stream checks: mode = outcome of the frequency control
// (0 skip before readImage, 1 readImage with PUB_THIS_FRAME false, 2 readImage + packaging).  R_in: caller-supplied relative_R or
// NULL = Estimator::predictMotion.  Returns 1 when `image` holds a feature map for processImage.
int Pipeline::track(const uint8_t *gray, double t, int mode, const double *R_in, std::map<int, std::array<double, 7>> &image, const V3 *bg, const double *td) {
    image.clear();
    if (first_image_flag) {  // estimator_nodelet.cpp:234-240
        first_image_flag = false;
        last_image_time = t;
        return 0;
    }
    if (mode == 0) return 0;  // "Skip this frame" :266-271 (before readImage: last_image_time keeps its value)
    double R[9];
    if (R_in) for (int k = 0; k < 9; k++) R[k] = R_in[k];
    else if (cfg.use_imu) est.predictMotion(last_image_time, t + (td ? *td : est.td), R, bg);  // :309-313
    else { for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0; }   // readImage(img, t): relative_R defaults to identity (and is not used)
    tracker.readImage(gray, t, R, mode == 2);
    last_image_time = t;
    tracker.updateIDs();  // :324-330
    if (mode != 2) return 0;
    for (size_t j = 0; j < tracker.ids.size(); j++)  // :336-363
        if (tracker.track_cnt[j] > 1)
            image[tracker.ids[j]] = {(double)tracker.cur_un_pts[j].x, (double)tracker.cur_un_pts[j].y, 1.0, (double)tracker.cur_pts[j].x,
                                     (double)tracker.cur_pts[j].y, (double)tracker.pts_velocity[j].x, (double)tracker.pts_velocity[j].y};
    if (!init_pub) { init_pub = true; image.clear(); return 0; }          // :365-368
    if (!init_feature) { init_feature = true; image.clear(); return 0; }  // :371-377
    return image.empty() ? 0 : 1;
}
void Pipeline::restart() {  // estimator_nodelet.cpp:243-262
    first_image_flag = true;
    last_image_time = 0;
    est.clearState();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) est.ric(i, j) = est.cfg.ric[i * 3 + j];   // setParameter() (estimator.cpp:15-41)
    est.tic = V3(est.cfg.tic[0], est.cfg.tic[1], est.cfg.tic[2]);
    est.td = est.cfg.td;
    est.g = V3(0, 0, est.cfg.g_norm);
    snap_bg = est.latest_Bg; snap_td = est.td;
}
// EstimatorNodelet::process for one queued feature frame (:462-549): inputDepth + processImage
int Pipeline::process(std::map<int, std::array<double, 7>> &image, const uint16_t *depth, double t) {
    int rc = est.processImage(image, depth, t);
    if (rc == 1) return 0;
    frames_processed++;
    return 1;
}
int Pipeline::feed(const uint8_t *gray, const uint16_t *depth, double t, int mode) {
    // tracker_lag 1: process_tracker of this frame runs while process() still optimises the previous one (the nodelet's two threads,
    // estimator_nodelet.cpp:192-459 / :462-549), so predictMotion sees latest_Bg / td as they were before the previous frame was
    // processed.  The snapshot is renewed on every call, before this frame is processed.
    const V3 bg_used = tracker_lag ? snap_bg : est.latest_Bg;
    const double td_used = tracker_lag ? snap_td : est.td;
    snap_bg = est.latest_Bg; snap_td = est.td;
    if (cfg.use_imu && !est.IMUAvailable(t + td_used)) return -1;  // caller contract: IMU pushed through t + td (upstream busy-waits, estimator.cpp:178-183)
    std::map<int, std::array<double, 7>> image;
    if (!track(gray, t, mode, nullptr, image, &bg_used, &td_used)) return 0;
    return process(image, depth, t);
}

// Frequency control + stream-discontinuity detection of process_tracker (estimator_nodelet.cpp:94-95, 234-286), restated as a
// stand-alone state machine so that replay drivers on both sides (oracle and product) take identical decisions.
std::vector<std::pair<int, int>> pair_color_depth(const std::vector<double> &color, const std::vector<double> &depth, int thrown[2]) {
    // estimator_nodelet.cpp:206-226, with the queues replaced by read positions (messages arrive in stamp order per topic)
    std::vector<std::pair<int, int>> out;
    size_t ic = 0, id = 0;
    thrown[0] = thrown[1] = 0;
    while (ic < color.size() && id < depth.size()) {
        const double time_color = color[ic], time_depth = depth[id];
        if (time_color < time_depth - 0.003) { ic++; thrown[0]++; }        // "throw color"
        else if (time_color > time_depth + 0.003) { id++; thrown[1]++; }   // "throw depth"
        else { out.push_back({(int)ic, (int)id}); ic++; id++; }
    }
    return out;
}

FrameGate::FrameGate(int freq_, int frontend_freq_) : freq(freq_ == 0 ? 100 : freq_), frontend_freq(frontend_freq_) {}  // parameters.cpp:133-134
int FrameGate::step(double t) {
    if (first_image_flag) {  // :234-240
        first_image_flag = false;
        first_image_time = t;
        last_image_time = t;
        return GATE_FIRST;
    }
    if (t - last_image_time > 1.0 || t < last_image_time) {  // :243-262
        first_image_flag = true;
        last_image_time = 0;
        pub_count = 1;
        return GATE_RESET;
    }
    if (std::round(1.0 * input_count / (t - first_image_time)) > frontend_freq) return GATE_SKIP;  // :264-271 (last_image_time untouched)
    ++input_count;
    bool pub = false;
    if (std::round(1.0 * pub_count / (t - first_image_time)) <= freq) {  // :274-284
        pub = true;
        if (std::abs(1.0 * pub_count / (t - first_image_time) - freq) < 0.01 * freq) {
            first_image_time = t;
            pub_count = 0;
            input_count = 0;
        }
    }
    last_image_time = t;  // :316
    if (pub) pub_count++;  // :333
    return pub ? GATE_PUBLISH : GATE_TRACK;
}
void FrameGate::empty_map(double t) {  // :386-392: a published frame whose feature map came out empty restarts the rate window
    first_image_time = t;
    pub_count = 0;
    input_count = 0;
}

}  // namespace ovio
