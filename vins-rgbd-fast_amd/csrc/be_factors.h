// Per-thread factor math of the sliding-window back-end (FP64). __host__ __device__ so the same functions can be
// exercised by host self tests; the HIP kernels in be_kernels.hip are the only product callers.
// Reference: vins_estimator/src/factor/integration_base.h:56-195, imu_factor.h:20-205,
//            projection_factor.cpp:22-130, projection_td_factor.cpp:34-150, pose_local_parameterization.cpp:3-28,
//            marginalization_factor.cpp:39-72 (loss correction), :374-393 (pose delta).
#pragma once
#include "vio_state.h"

namespace bf {
using namespace dm;

enum { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };

DM_HD void put33(double *M, int ld, int r, int c, const m3 &B) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[(r + i) * ld + c + j] = B.a[i * 3 + j];
}
DM_HD m3 get33(const double *M, int ld, int r, int c) {
    m3 B;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B.a[i * 3 + j] = M[(r + i) * ld + c + j];
    return B;
}

// IntegrationBase constructor (integration_base.h:13-30)
DM_HD void preint_init(PreInt &p, v3 acc0, v3 gyr0, v3 ba, v3 bg) {
    st3(p.lin_acc, acc0); st3(p.lin_gyr, gyr0); st3(p.lin_ba, ba); st3(p.lin_bg, bg);
    st3(p.acc0, acc0); st3(p.gyr0, gyr0);
    p.dp[0] = p.dp[1] = p.dp[2] = 0; p.dv[0] = p.dv[1] = p.dv[2] = 0;
    p.dq[0] = 1; p.dq[1] = p.dq[2] = p.dq[3] = 0;
    p.sum_dt = 0;
    for (int i = 0; i < 225; i++) { p.jac[i] = 0; p.cov[i] = 0; }
    for (int i = 0; i < 15; i++) p.jac[i * 16] = 1;
    p.n_buf = 0;
    p.valid = 1;
}

// midPointIntegration (integration_base.h:56-134): state update + F (15x15) and V (15x18), both row-major, zero-initialised here.
struct PreintStep { v3 dp, dv; quat dq; };
DM_HD PreintStep preint_midpoint(const PreInt &p, double dt, v3 acc_1, v3 gyr_1, double *F, double *V) {
    v3 acc_0 = ld3(p.acc0), gyr_0 = ld3(p.gyr0), lba = ld3(p.lin_ba), lbg = ld3(p.lin_bg);
    quat delta_q = mkq(p.dq[0], p.dq[1], p.dq[2], p.dq[3]);
    v3 delta_p = ld3(p.dp), delta_v = ld3(p.dv);
    v3 un_acc_0 = qrot(delta_q, sub(acc_0, lba));
    v3 un_gyr = sub(scl(0.5, add(gyr_0, gyr_1)), lbg);
    quat rq = qmul(delta_q, mkq(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2));
    v3 un_acc_1 = qrot(rq, sub(acc_1, lba));
    v3 un_acc = scl(0.5, add(un_acc_0, un_acc_1));
    PreintStep o;
    o.dp = add(add(delta_p, scl(dt, delta_v)), scl(dt * dt, scl(0.5, un_acc)));
    o.dv = add(delta_v, scl(dt, un_acc));
    o.dq = rq;
    for (int i = 0; i < 225; i++) F[i] = 0;
    for (int i = 0; i < 270; i++) V[i] = 0;
    m3 R_w_x = skew(un_gyr), R_a_0_x = skew(sub(acc_0, lba)), R_a_1_x = skew(sub(acc_1, lba));
    m3 Rq = q2R(delta_q), Rr = q2R(rq), I = eye();
    m3 ImW = sub(I, scl(dt, R_w_x));
    put33(F, 15, 0, 0, I);
    put33(F, 15, 0, 3, add(scl(-0.25 * dt * dt, mul(Rq, R_a_0_x)), scl(-0.25 * dt * dt, mul(mul(Rr, R_a_1_x), ImW))));
    put33(F, 15, 0, 6, scl(dt, I));
    put33(F, 15, 0, 9, scl(-0.25 * dt * dt, add(Rq, Rr)));
    put33(F, 15, 0, 12, scl(-0.25 * dt * dt * -dt, mul(Rr, R_a_1_x)));
    put33(F, 15, 3, 3, ImW);
    put33(F, 15, 3, 12, scl(-dt, I));
    put33(F, 15, 6, 3, add(scl(-0.5 * dt, mul(Rq, R_a_0_x)), scl(-0.5 * dt, mul(mul(Rr, R_a_1_x), ImW))));
    put33(F, 15, 6, 6, I);
    put33(F, 15, 6, 9, scl(-0.5 * dt, add(Rq, Rr)));
    put33(F, 15, 6, 12, scl(-0.5 * dt * -dt, mul(Rr, R_a_1_x)));
    put33(F, 15, 9, 9, I);
    put33(F, 15, 12, 12, I);
    m3 V03 = scl(0.25 * dt * dt * 0.5 * dt, neg(mul(Rr, R_a_1_x)));
    m3 V63 = scl(0.5 * dt * 0.5 * dt, neg(mul(Rr, R_a_1_x)));
    put33(V, 18, 0, 0, scl(0.25 * dt * dt, Rq));
    put33(V, 18, 0, 3, V03);
    put33(V, 18, 0, 6, scl(0.25 * dt * dt, Rr));
    put33(V, 18, 0, 9, V03);
    put33(V, 18, 3, 3, scl(0.5 * dt, I));
    put33(V, 18, 3, 9, scl(0.5 * dt, I));
    put33(V, 18, 6, 0, scl(0.5 * dt, Rq));
    put33(V, 18, 6, 3, V63);
    put33(V, 18, 6, 6, scl(0.5 * dt, Rr));
    put33(V, 18, 6, 9, V63);
    put33(V, 18, 9, 12, scl(dt, I));
    put33(V, 18, 12, 15, scl(dt, I));
    return o;
}

// The two halves of preint_midpoint for the pipelined propagation of be_ingest (same expressions, hence the same bits): the state
// recursion (serial, a few dozen operations per sample) and the F / V matrices of a step, which depend only on the state BEFORE it.
struct PreintPre { quat dq; v3 acc0, gyr0; };   // what a step's F / V need besides (dt, acc_1, gyr_1) and the linearisation biases
DM_HD void preint_state_step(quat &delta_q, v3 &delta_p, v3 &delta_v, v3 acc_0, v3 gyr_0, v3 lba, v3 lbg, double dt, v3 acc_1, v3 gyr_1) {
    v3 un_acc_0 = qrot(delta_q, sub(acc_0, lba));
    v3 un_gyr = sub(scl(0.5, add(gyr_0, gyr_1)), lbg);
    quat rq = qmul(delta_q, mkq(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2));
    v3 un_acc_1 = qrot(rq, sub(acc_1, lba));
    v3 un_acc = scl(0.5, add(un_acc_0, un_acc_1));
    delta_p = add(add(delta_p, scl(dt, delta_v)), scl(dt * dt, scl(0.5, un_acc)));
    delta_v = add(delta_v, scl(dt, un_acc));
    delta_q = qnormalized(rq);
}
DM_HD void preint_step_FV(const PreintPre &s, v3 lba, v3 lbg, double dt, v3 acc_1, v3 gyr_1, double *F, double *V, const bool zero = true) {
    const quat delta_q = s.dq;
    const v3 acc_0 = s.acc0, gyr_0 = s.gyr0;
    v3 un_gyr = sub(scl(0.5, add(gyr_0, gyr_1)), lbg);
    quat rq = qmul(delta_q, mkq(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2));
    if (zero) {   // (be_ingest clears the matrices with all its threads beforehand: 495 stores of one lane otherwise)
        for (int i = 0; i < 225; i++) F[i] = 0;
        for (int i = 0; i < 270; i++) V[i] = 0;
    }
    m3 R_w_x = skew(un_gyr), R_a_0_x = skew(sub(acc_0, lba)), R_a_1_x = skew(sub(acc_1, lba));
    m3 Rq = q2R(delta_q), Rr = q2R(rq), I = eye();
    m3 ImW = sub(I, scl(dt, R_w_x));
    put33(F, 15, 0, 0, I);
    put33(F, 15, 0, 3, add(scl(-0.25 * dt * dt, mul(Rq, R_a_0_x)), scl(-0.25 * dt * dt, mul(mul(Rr, R_a_1_x), ImW))));
    put33(F, 15, 0, 6, scl(dt, I));
    put33(F, 15, 0, 9, scl(-0.25 * dt * dt, add(Rq, Rr)));
    put33(F, 15, 0, 12, scl(-0.25 * dt * dt * -dt, mul(Rr, R_a_1_x)));
    put33(F, 15, 3, 3, ImW);
    put33(F, 15, 3, 12, scl(-dt, I));
    put33(F, 15, 6, 3, add(scl(-0.5 * dt, mul(Rq, R_a_0_x)), scl(-0.5 * dt, mul(mul(Rr, R_a_1_x), ImW))));
    put33(F, 15, 6, 6, I);
    put33(F, 15, 6, 9, scl(-0.5 * dt, add(Rq, Rr)));
    put33(F, 15, 6, 12, scl(-0.5 * dt * -dt, mul(Rr, R_a_1_x)));
    put33(F, 15, 9, 9, I);
    put33(F, 15, 12, 12, I);
    m3 V03 = scl(0.25 * dt * dt * 0.5 * dt, neg(mul(Rr, R_a_1_x)));
    m3 V63 = scl(0.5 * dt * 0.5 * dt, neg(mul(Rr, R_a_1_x)));
    put33(V, 18, 0, 0, scl(0.25 * dt * dt, Rq));
    put33(V, 18, 0, 3, V03);
    put33(V, 18, 0, 6, scl(0.25 * dt * dt, Rr));
    put33(V, 18, 0, 9, V03);
    put33(V, 18, 3, 3, scl(0.5 * dt, I));
    put33(V, 18, 3, 9, scl(0.5 * dt, I));
    put33(V, 18, 6, 0, scl(0.5 * dt, Rq));
    put33(V, 18, 6, 3, V63);
    put33(V, 18, 6, 6, scl(0.5 * dt, Rr));
    put33(V, 18, 6, 9, V63);
    put33(V, 18, 9, 12, scl(dt, I));
    put33(V, 18, 12, 15, scl(dt, I));
}

// sqrt_info = LLT(cov^-1).matrixL().transpose()  (imu_factor.h:66-69); out row-major 15x15 upper triangular
DM_HD void imu_sqrt_info(const double *cov, double *out) {
    double L[225], Li[225], Ci[225];
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) L[i * 15 + j] = 0.5 * (cov[i * 15 + j] + cov[j * 15 + i]);
    for (int i = 0; i < 225; i++) out[i] = 0;
    bool ok = true;
    for (int j = 0; j < 15 && ok; j++) {
        double s = L[j * 15 + j];
        for (int k = 0; k < j; k++) s -= L[j * 15 + k] * L[j * 15 + k];
        if (!(s > 0.0)) { ok = false; break; }
        double l = sqrt(s);
        L[j * 15 + j] = l;
        for (int i = j + 1; i < 15; i++) {
            double t = L[i * 15 + j];
            for (int k = 0; k < j; k++) t -= L[i * 15 + k] * L[j * 15 + k];
            L[i * 15 + j] = t / l;
        }
    }
    if (!ok) return;
    for (int c = 0; c < 15; c++) {
        double e[15];
        for (int i = 0; i < 15; i++) e[i] = (i == c) ? 1.0 : 0.0;
        for (int i = 0; i < 15; i++) {
            double s = e[i];
            for (int k = 0; k < i; k++) s -= L[i * 15 + k] * e[k];
            e[i] = s / L[i * 15 + i];
        }
        for (int i = 0; i < 15; i++) Li[i * 15 + c] = e[i];
    }
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            double s = 0;
            for (int k = 0; k < 15; k++) s += Li[k * 15 + i] * Li[k * 15 + j];
            Ci[i * 15 + j] = s;
        }
    for (int j = 0; j < 15; j++) {
        double s = Ci[j * 15 + j];
        for (int k = 0; k < j; k++) s -= Ci[j * 15 + k] * Ci[j * 15 + k];
        if (!(s > 0.0)) { for (int q = 0; q < 225; q++) out[q] = 0; return; }
        double l = sqrt(s);
        Ci[j * 15 + j] = l;
        for (int i = j + 1; i < 15; i++) {
            double t = Ci[i * 15 + j];
            for (int k = 0; k < j; k++) t -= Ci[i * 15 + k] * Ci[j * 15 + k];
            Ci[i * 15 + j] = t / l;
        }
    }
    for (int i = 0; i < 15; i++) for (int j = i; j < 15; j++) out[i * 15 + j] = Ci[j * 15 + i];
}

DM_HD void Qleft(quat q, double *M) {  // utility.h:46-54, 4x4 row-major
    M[0] = q.w; M[1] = -q.x; M[2] = -q.y; M[3] = -q.z;
    m3 S = skew(qvec(q));
    double v[3] = {q.x, q.y, q.z};
    for (int i = 0; i < 3; i++) {
        M[(1 + i) * 4] = v[i];
        for (int j = 0; j < 3; j++) M[(1 + i) * 4 + 1 + j] = (i == j ? q.w : 0.0) + S.a[i * 3 + j];
    }
}
DM_HD void Qright(quat q, double *M) {  // utility.h:56-64
    M[0] = q.w; M[1] = -q.x; M[2] = -q.y; M[3] = -q.z;
    m3 S = skew(qvec(q));
    double v[3] = {q.x, q.y, q.z};
    for (int i = 0; i < 3; i++) {
        M[(1 + i) * 4] = v[i];
        for (int j = 0; j < 3; j++) M[(1 + i) * 4 + 1 + j] = (i == j ? q.w : 0.0) - S.a[i * 3 + j];
    }
}
DM_HD m3 br33(const double *M) { m3 B; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B.a[i * 3 + j] = M[(1 + i) * 4 + 1 + j]; return B; }

// IntegrationBase::evaluate (integration_base.h:164-195): raw (un-whitened) 15-residual
DM_HD void imu_raw_residual(const PreInt &p, v3 G, const double *pi, const double *sbi, const double *pj, const double *sbj, double *r) {
    v3 Pi = ld3(pi), Vi = ld3(sbi), Bai = ld3(sbi + 3), Bgi = ld3(sbi + 6);
    v3 Pj = ld3(pj), Vj = ld3(sbj), Baj = ld3(sbj + 3), Bgj = ld3(sbj + 6);
    quat Qi = mkq(pi[6], pi[3], pi[4], pi[5]), Qj = mkq(pj[6], pj[3], pj[4], pj[5]);
    m3 dp_dba = get33(p.jac, 15, O_P, O_BA), dp_dbg = get33(p.jac, 15, O_P, O_BG), dq_dbg = get33(p.jac, 15, O_R, O_BG),
       dv_dba = get33(p.jac, 15, O_V, O_BA), dv_dbg = get33(p.jac, 15, O_V, O_BG);
    v3 dba = sub(Bai, ld3(p.lin_ba)), dbg = sub(Bgi, ld3(p.lin_bg));
    quat delta_q = mkq(p.dq[0], p.dq[1], p.dq[2], p.dq[3]);
    quat cq = qmul(delta_q, deltaQ(mul(dq_dbg, dbg)));
    v3 cv = add(add(ld3(p.dv), mul(dv_dba, dba)), mul(dv_dbg, dbg));
    v3 cp = add(add(ld3(p.dp), mul(dp_dba, dba)), mul(dp_dbg, dbg));
    quat Qi_inv = qinv(Qi);
    double sdt = p.sum_dt;
    v3 rp = sub(qrot(Qi_inv, sub(sub(add(scl(sdt * sdt, scl(0.5, G)), Pj), Pi), scl(sdt, Vi))), cp);
    v3 rq = scl(2.0, qvec(qmul(qinv(cq), qmul(Qi_inv, Qj))));
    v3 rv = sub(qrot(Qi_inv, sub(add(scl(sdt, G), Vj), Vi)), cv);
    st3(r + O_P, rp); st3(r + O_R, rq); st3(r + O_V, rv); st3(r + O_BA, sub(Baj, Bai)); st3(r + O_BG, sub(Bgj, Bgi));
}

// IMUFactor::Evaluate raw Jacobian (imu_factor.h:73-202) in tangent coordinates: J is 15x30 row-major,
// columns [pose_i(6) speedbias_i(9) pose_j(6) speedbias_j(9)], before whitening by sqrt_info.
DM_HD void imu_raw_jacobian(const PreInt &p, v3 G, const double *pi, const double *sbi, const double *pj, const double *sbj, double *J) {
    for (int i = 0; i < 450; i++) J[i] = 0;
    v3 Pi = ld3(pi), Vi = ld3(sbi), Bgi = ld3(sbi + 6);
    v3 Pj = ld3(pj), Vj = ld3(sbj);
    quat Qi = mkq(pi[6], pi[3], pi[4], pi[5]), Qj = mkq(pj[6], pj[3], pj[4], pj[5]);
    double sdt = p.sum_dt;
    m3 dp_dba = get33(p.jac, 15, O_P, O_BA), dp_dbg = get33(p.jac, 15, O_P, O_BG), dq_dbg = get33(p.jac, 15, O_R, O_BG),
       dv_dba = get33(p.jac, 15, O_V, O_BA), dv_dbg = get33(p.jac, 15, O_V, O_BG);
    quat Qi_inv = qinv(Qi), Qj_inv = qinv(Qj);
    m3 RiT = q2R(Qi_inv);
    quat delta_q = mkq(p.dq[0], p.dq[1], p.dq[2], p.dq[3]);
    quat cq = qmul(delta_q, deltaQ(mul(dq_dbg, sub(Bgi, ld3(p.lin_bg)))));
    double L4[16], R4[16], LR[16];
    // pose_i (cols 0..5)
    put33(J, 30, O_P, 0, neg(RiT));
    put33(J, 30, O_P, 3, skew(qrot(Qi_inv, sub(sub(add(scl(sdt * sdt, scl(0.5, G)), Pj), Pi), scl(sdt, Vi)))));
    Qleft(qmul(Qj_inv, Qi), L4);
    Qright(cq, R4);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += L4[i * 4 + k] * R4[k * 4 + j]; LR[i * 4 + j] = s; }
    put33(J, 30, O_R, 3, neg(br33(LR)));
    put33(J, 30, O_V, 3, skew(qrot(Qi_inv, sub(add(scl(sdt, G), Vj), Vi))));
    // speedbias_i (cols 6..14)
    put33(J, 30, O_P, 6, scl(-sdt, RiT));
    put33(J, 30, O_P, 9, neg(dp_dba));
    put33(J, 30, O_P, 12, neg(dp_dbg));
    Qleft(qmul(qmul(Qj_inv, Qi), delta_q), L4);
    put33(J, 30, O_R, 12, neg(mul(br33(L4), dq_dbg)));
    put33(J, 30, O_V, 6, neg(RiT));
    put33(J, 30, O_V, 9, neg(dv_dba));
    put33(J, 30, O_V, 12, neg(dv_dbg));
    put33(J, 30, O_BA, 9, neg(eye()));
    put33(J, 30, O_BG, 12, neg(eye()));
    // pose_j (cols 15..20)
    put33(J, 30, O_P, 15, RiT);
    Qleft(qmul(qmul(qinv(cq), Qi_inv), Qj), L4);
    put33(J, 30, O_R, 18, br33(L4));
    // speedbias_j (cols 21..29)
    put33(J, 30, O_V, 21, RiT);
    put33(J, 30, O_BA, 24, eye());
    put33(J, 30, O_BG, 27, eye());
}

// One column group of the raw IMU Jacobian, written straight into a 15 x ld row-major buffer (columns as in
// imu_raw_jacobian): part 0 = pose_i (cols 0-5), 1 = speedbias_i (6-14), 2 = pose_j (15-20), 3 = speedbias_j (21-29).
// Lets four threads share one factor.
DM_HD void imu_raw_jacobian_part(const PreInt &p, v3 G, const double *pi, const double *sbi, const double *pj, const double *sbj, int part,
                                 double *J, int ld) {
    const int c0 = part == 0 ? 0 : (part == 1 ? 6 : (part == 2 ? 15 : 21)), nc = (part & 1) ? 9 : 6;
    for (int r = 0; r < 15; r++) for (int c = 0; c < nc; c++) J[r * ld + c0 + c] = 0;
    v3 Pi = ld3(pi), Vi = ld3(sbi), Bgi = ld3(sbi + 6);
    v3 Pj = ld3(pj), Vj = ld3(sbj);
    quat Qi = mkq(pi[6], pi[3], pi[4], pi[5]), Qj = mkq(pj[6], pj[3], pj[4], pj[5]);
    double sdt = p.sum_dt;
    quat Qi_inv = qinv(Qi), Qj_inv = qinv(Qj);
    m3 RiT = q2R(Qi_inv);
    quat delta_q = mkq(p.dq[0], p.dq[1], p.dq[2], p.dq[3]);
    m3 dq_dbg = get33(p.jac, 15, O_R, O_BG);
    double L4[16], R4[16], LR[16];
    if (part == 0) {
        quat cq = qmul(delta_q, deltaQ(mul(dq_dbg, sub(Bgi, ld3(p.lin_bg)))));
        put33(J, ld, O_P, 0, neg(RiT));
        put33(J, ld, O_P, 3, skew(qrot(Qi_inv, sub(sub(add(scl(sdt * sdt, scl(0.5, G)), Pj), Pi), scl(sdt, Vi)))));
        Qleft(qmul(Qj_inv, Qi), L4);
        Qright(cq, R4);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += L4[i * 4 + k] * R4[k * 4 + j]; LR[i * 4 + j] = s; }
        put33(J, ld, O_R, 3, neg(br33(LR)));
        put33(J, ld, O_V, 3, skew(qrot(Qi_inv, sub(add(scl(sdt, G), Vj), Vi))));
    } else if (part == 1) {
        m3 dp_dba = get33(p.jac, 15, O_P, O_BA), dp_dbg = get33(p.jac, 15, O_P, O_BG), dv_dba = get33(p.jac, 15, O_V, O_BA),
           dv_dbg = get33(p.jac, 15, O_V, O_BG);
        put33(J, ld, O_P, 6, scl(-sdt, RiT));
        put33(J, ld, O_P, 9, neg(dp_dba));
        put33(J, ld, O_P, 12, neg(dp_dbg));
        Qleft(qmul(qmul(Qj_inv, Qi), delta_q), L4);
        put33(J, ld, O_R, 12, neg(mul(br33(L4), dq_dbg)));
        put33(J, ld, O_V, 6, neg(RiT));
        put33(J, ld, O_V, 9, neg(dv_dba));
        put33(J, ld, O_V, 12, neg(dv_dbg));
        put33(J, ld, O_BA, 9, neg(eye()));
        put33(J, ld, O_BG, 12, neg(eye()));
    } else if (part == 2) {
        quat cq = qmul(delta_q, deltaQ(mul(dq_dbg, sub(Bgi, ld3(p.lin_bg)))));
        put33(J, ld, O_P, 15, RiT);
        Qleft(qmul(qmul(qinv(cq), Qi_inv), Qj), L4);
        put33(J, ld, O_R, 18, br33(L4));
    } else {
        put33(J, ld, O_V, 21, RiT);
        put33(J, ld, O_BA, 24, eye());
        put33(J, ld, O_BG, 27, eye());
    }
}

// ProjectionFactor / ProjectionTdFactor::Evaluate. obs = 9 doubles (x y z u v vx vy cur_td depth).
// J (optional) = 2x20 row-major, columns [pose_i(6) pose_j(6) ex(6) td(1) inv_depth(1)] in tangent coordinates.
DM_HD void eval_projection(const vio_config &c, const double *pi, const double *pj, const double *ex, double inv_dep, double td,
                           const double *oi, const double *oj, bool use_td, double *r, double *J) {
    v3 Pi = ld3(pi), Pj = ld3(pj), tic = ld3(ex);
    quat Qi = mkq(pi[6], pi[3], pi[4], pi[5]), Qj = mkq(pj[6], pj[3], pj[4], pj[5]), qic = mkq(ex[6], ex[3], ex[4], ex[5]);
    v3 pts_i = mk(oi[0], oi[1], oi[2]), pts_j = mk(oj[0], oj[1], oj[2]);
    v3 vel_i = mk(oi[5], oi[6], 0), vel_j = mk(oj[5], oj[6], 0);
    if (use_td) {
        double ROW = (double)c.height;
        double row_i = oi[4] - ROW / 2, row_j = oj[4] - ROW / 2;
        pts_i = sub(pts_i, scl(td - oi[7] + c.tr / ROW * row_i, vel_i));
        pts_j = sub(pts_j, scl(td - oj[7] + c.tr / ROW * row_j, vel_j));
    }
    double sq = c.focal_length / 1.5;
    v3 pts_camera_i = scl(1.0 / inv_dep, pts_i);
    pts_camera_i = mk(pts_i.x / inv_dep, pts_i.y / inv_dep, pts_i.z / inv_dep);
    v3 pts_imu_i = add(qrot(qic, pts_camera_i), tic);
    v3 pts_w = add(qrot(Qi, pts_imu_i), Pi);
    v3 pts_imu_j = qrot(qinv(Qj), sub(pts_w, Pj));
    v3 pts_camera_j = qrot(qinv(qic), sub(pts_imu_j, tic));
    double dep_j = pts_camera_j.z;
    r[0] = sq * (pts_camera_j.x / dep_j - pts_j.x);
    r[1] = sq * (pts_camera_j.y / dep_j - pts_j.y);
    if (!J) return;
    m3 Ri = q2R(Qi), Rj = q2R(Qj), ric = q2R(qic);
    double red[6] = {sq / dep_j, 0, -sq * pts_camera_j.x / (dep_j * dep_j), 0, sq / dep_j, -sq * pts_camera_j.y / (dep_j * dep_j)};
    m3 ricT = tr(ric), RjT = tr(Rj);
    m3 A, Bm;
#define BF_RED(col0)                                                                             \
    for (int i = 0; i < 2; i++)                                                                  \
        for (int j = 0; j < 3; j++) {                                                            \
            double s = 0, t = 0;                                                                 \
            for (int k = 0; k < 3; k++) { s += red[i * 3 + k] * A.a[k * 3 + j]; t += red[i * 3 + k] * Bm.a[k * 3 + j]; } \
            J[i * 20 + (col0) + j] = s;                                                          \
            J[i * 20 + (col0) + 3 + j] = t;                                                      \
        }
    A = mul(ricT, RjT);
    Bm = mul(mul(mul(ricT, RjT), Ri), neg(skew(pts_imu_i)));
    BF_RED(0)
    A = mul(ricT, neg(RjT));
    Bm = mul(ricT, skew(pts_imu_j));
    BF_RED(6)
    m3 tmp_r = mul(mul(mul(ricT, RjT), Ri), ric);
    A = mul(ricT, sub(mul(RjT, Ri), eye()));
    Bm = add(add(neg(mul(tmp_r, skew(pts_camera_i))), skew(mul(tmp_r, pts_camera_i))),
             skew(mul(ricT, sub(mul(RjT, sub(add(mul(Ri, tic), Pi), Pj)), tic))));
    BF_RED(12)
#undef BF_RED
    {
        v3 v = scl(-1.0 / (inv_dep * inv_dep), mul(tmp_r, pts_i));
        for (int i = 0; i < 2; i++) J[i * 20 + 19] = red[i * 3] * v.x + red[i * 3 + 1] * v.y + red[i * 3 + 2] * v.z;
        if (use_td) {
            v3 w0 = mul(tmp_r, vel_i);
            v3 w = mk(w0.x / inv_dep * -1.0, w0.y / inv_dep * -1.0, w0.z / inv_dep * -1.0);
            for (int i = 0; i < 2; i++) J[i * 20 + 18] = red[i * 3] * w.x + red[i * 3 + 1] * w.y + red[i * 3 + 2] * w.z;
            J[18] += sq * vel_j.x;
            J[20 + 18] += sq * vel_j.y;
        } else {
            J[18] = 0;
            J[20 + 18] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Frame-pair form of the projection factors.  Every residual between frames (i, j) shares
//   A1 = ric^T Rj^T,  A2 = A1 Ri,  M = A2 ric,  t = A1 (Ri tic + Pi - Pj) - ric^T tic     (pts_camera_j = M pts_camera_i + t)
// so the per-residual work of projection_factor.cpp:22-130 / projection_td_factor.cpp:34-150 reduces to a few 3-vector products
// (same algebra, different association: results agree with eval_projection to round-off).
struct PairGeo { double A1[9], A2[9], M[9], t[3]; };

DM_HD void pair_geo(const double *pi, const double *pj, const double *ex, PairGeo &g) {
    v3 Pi = ld3(pi), Pj = ld3(pj), tic = ld3(ex);
    m3 Ri = q2R(mkq(pi[6], pi[3], pi[4], pi[5])), Rj = q2R(mkq(pj[6], pj[3], pj[4], pj[5])), ric = q2R(mkq(ex[6], ex[3], ex[4], ex[5]));
    m3 ricT = tr(ric);
    m3 A1 = mul(ricT, tr(Rj)), A2 = mul(A1, Ri), M = mul(A2, ric);
    v3 t = sub(mul(A1, add(mul(Ri, tic), sub(Pi, Pj))), mul(ricT, tic));
    stm(g.A1, A1); stm(g.A2, A2); stm(g.M, M); st3(g.t, t);
}
// row (1x3) times skew(w)
DM_HD v3 rowskew(v3 r, v3 w) { return mk(r.y * w.z - r.z * w.y, r.z * w.x - r.x * w.z, r.x * w.y - r.y * w.x); }
DM_HD v3 rowmul(v3 r, const double *A) {  // row (1x3) times a 3x3 row-major matrix
    return mk(r.x * A[0] + r.y * A[3] + r.z * A[6], r.x * A[1] + r.y * A[4] + r.z * A[7], r.x * A[2] + r.y * A[5] + r.z * A[8]);
}
DM_HD v3 matvec9(const double *A, v3 v) { return mk(A[0] * v.x + A[1] * v.y + A[2] * v.z, A[3] * v.x + A[4] * v.y + A[5] * v.z, A[6] * v.x + A[7] * v.y + A[8] * v.z); }

// ricm = ric (row-major 3x3), tic: extrinsic.  r[2]; J (optional) 2 rows as in eval_projection.  With cauchy = true the
// Jacobian is multiplied by the CauchyLoss(1) weight sqrt(1 / (1 + |r|^2)), which is returned in *wgt (r itself is left unweighted).
// Row layout: rs = 20 (default): [pose_i(6) pose_j(6) ex(6) td inv_depth]; rs = 14 ("compact", ext = false): [pose_i(6) pose_j(6)
// inv_depth -] for solves in which the extrinsic and td blocks are constant (Ceres does not evaluate Jacobians of constant blocks
// either); the inverse-depth column sits at index lcol = rs == 20 ? 19 : 12.
DM_HD void eval_projection_pair(const vio_config &c, const PairGeo &g, const double *ricm, const double *ticp, double inv_dep, double td,
                                const double *oi, const double *oj, bool use_td, double *r, double *J, bool cauchy, double *wgt,
                                const int rs = 20, const bool ext = true) {
    v3 pts_i = mk(oi[0], oi[1], oi[2]), pts_j = mk(oj[0], oj[1], oj[2]);
    v3 vel_i = mk(oi[5], oi[6], 0), vel_j = mk(oj[5], oj[6], 0);
    if (use_td) {
        double ROW = (double)c.height;
        double row_i = oi[4] - ROW / 2, row_j = oj[4] - ROW / 2;
        pts_i = sub(pts_i, scl(td - oi[7] + c.tr / ROW * row_i, vel_i));
        pts_j = sub(pts_j, scl(td - oj[7] + c.tr / ROW * row_j, vel_j));
    }
    const double sq = c.focal_length / 1.5;
    v3 pc_i = mk(pts_i.x / inv_dep, pts_i.y / inv_dep, pts_i.z / inv_dep);
    v3 tt = ld3(g.t);
    v3 pc_j = add(matvec9(g.M, pc_i), tt);
    const double dep_j = pc_j.z;
    r[0] = sq * (pc_j.x / dep_j - pts_j.x);
    r[1] = sq * (pc_j.y / dep_j - pts_j.y);
    if (!J) return;
    double s = 1.0;
    if (cauchy) { s = sqrt(1.0 / (1.0 + (r[0] * r[0] + r[1] * r[1]))); *wgt = s; }
    v3 red0 = mk(s * sq / dep_j, 0, -s * sq * pc_j.x / (dep_j * dep_j)), red1 = mk(0, s * sq / dep_j, -s * sq * pc_j.y / (dep_j * dep_j));
    v3 tic = ld3(ticp);
    v3 pim_i = add(matvec9(ricm, pc_i), tic), pim_j = add(matvec9(ricm, pc_j), tic);
    // rho = red * A2, rho1 = red * A1, rhoT = red * ric^T, rhoM = red * M
    v3 a0 = rowmul(red0, g.A1), a1 = rowmul(red1, g.A1);
    v3 b0 = rowmul(red0, g.A2), b1 = rowmul(red1, g.A2);
    // red * ric^T : (ric^T)[k][c] = ric[c][k]
    v3 c0 = mk(red0.x * ricm[0] + red0.y * ricm[1] + red0.z * ricm[2], red0.x * ricm[3] + red0.y * ricm[4] + red0.z * ricm[5],
               red0.x * ricm[6] + red0.y * ricm[7] + red0.z * ricm[8]);
    v3 c1 = mk(red1.x * ricm[0] + red1.y * ricm[1] + red1.z * ricm[2], red1.x * ricm[3] + red1.y * ricm[4] + red1.z * ricm[5],
               red1.x * ricm[6] + red1.y * ricm[7] + red1.z * ricm[8]);
    v3 npi = neg(pim_i);
    // pose_i: [red A1 | red A2 (-skew(pts_imu_i))]
    st3(J + 0, a0); st3(J + rs, a1);
    st3(J + 3, rowskew(b0, npi)); st3(J + rs + 3, rowskew(b1, npi));
    // pose_j: [-red A1 | red ric^T skew(pts_imu_j)]
    st3(J + 6, neg(a0)); st3(J + rs + 6, neg(a1));
    st3(J + 9, rowskew(c0, pim_j)); st3(J + rs + 9, rowskew(c1, pim_j));
    // inverse depth: red * M * pts_i * (-1 / lambda^2)
    {
        v3 v = scl(-1.0 / (inv_dep * inv_dep), matvec9(g.M, pts_i));
        const int lcol = ext ? 19 : 12;
        J[lcol] = dot(red0, v);
        J[rs + lcol] = dot(red1, v);
    }
    if (!ext) return;
    // extrinsic: [red (A2 - ric^T) | red (-M skew(pc_i) + skew(pc_j))]
    v3 m0 = rowmul(red0, g.M), m1 = rowmul(red1, g.M);
    st3(J + 12, sub(b0, c0)); st3(J + rs + 12, sub(b1, c1));
    st3(J + 15, sub(rowskew(red0, pc_j), rowskew(m0, pc_i))); st3(J + rs + 15, sub(rowskew(red1, pc_j), rowskew(m1, pc_i)));
    if (use_td) {
        v3 w0 = matvec9(g.M, vel_i);
        v3 w = mk(w0.x / inv_dep * -1.0, w0.y / inv_dep * -1.0, w0.z / inv_dep * -1.0);
        J[18] = dot(red0, w) + s * sq * vel_j.x;
        J[rs + 18] = dot(red1, w) + s * sq * vel_j.y;
    } else {
        J[18] = 0;
        J[rs + 18] = 0;
    }
}

// MarginalizationFactor::Evaluate pose delta (marginalization_factor.cpp:374-393)
DM_HD void pose_dx(const double *x, const double *x0, double *dx) {
    for (int k = 0; k < 3; k++) dx[k] = x[k] - x0[k];
    quat q0 = mkq(x0[6], x0[3], x0[4], x0[5]), q = mkq(x[6], x[3], x[4], x[5]);
    quat d = qmul(qinv(q0), q);
    v3 v = scl(2.0, qvec(d));
    if (!(d.w >= 0)) v = neg(v);
    dx[3] = v.x; dx[4] = v.y; dx[5] = v.z;
}
// PoseLocalParameterization::Plus
DM_HD void pose_plus(double *x, const double *d) {
    x[0] += d[0]; x[1] += d[1]; x[2] += d[2];
    quat q = mkq(x[6], x[3], x[4], x[5]);
    quat r = qnormalized(qmul(q, deltaQ(mk(d[3], d[4], d[5]))));
    x[3] = r.x; x[4] = r.y; x[5] = r.z; x[6] = r.w;
}

}  // namespace bf
