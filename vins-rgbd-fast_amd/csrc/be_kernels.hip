// Back-end kernels (gfx950, FP64): Estimator::processImage (vins_estimator/src/estimator/estimator.cpp:156-374) batched
// over S sequences, one workgroup per sequence, three launches per frame (be_finish runs inside the be_marg kernel); the block
// primitives and dense linear algebra live in be_linalg.h:
//   be_ingest  addFeatureCheckParallax (feature_manager.cpp:56-123), getIMUInterval/processIMU (estimator.cpp:118-154,
//              1910-1942, IntegrationBase::propagate), triangulateWithDepth (feature_manager.cpp:386-543)
//   be_solve   optimization() (estimator.cpp:1161-1368): the Ceres DENSE_SCHUR + traditional-DOGLEG solve replaced by a
//              bespoke trust-region (dogleg / LM-regularised Gauss-Newton) solver: batched residual/Jacobian evaluation,
//              frame-pair blocked J^T J, landmark Schur complement, dense Cholesky, step control as SURVEY.md App. B.5
//   be_marg    marginalisation (estimator.cpp:1370-1575, marginalization_factor.cpp:181-315) in the canonical layout
//   be_finish  movingConsistencyCheck, failureDetection, slideWindow, removeFailures, odometry row
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "be_factors.h"

using namespace dm;

// phase timers (debug): thread 0 of sequence 0 accumulates 100 MHz wall-clock ticks into B.timings[k]
#if VIO_TIMERS
#define PH_INIT long long ph_t0 = (s == 0 && threadIdx.x == 0) ? VIO_CLOCK() : 0
#define PH(k) do { if (s == 0 && threadIdx.x == 0) { long long n_ = VIO_CLOCK(); B.timings[k] += (float)(n_ - ph_t0); ph_t0 = n_; } } while (0)
#else
#define PH_INIT do {} while (0)
#define PH(k) do {} while (0)
#endif

namespace {

struct Ctx {
    float *timings;
    const DevCfg *C;
    int s, W, P, LW, NL, NLs, NPR;
    BeSeq *be;
    FeSeq *fe;
    PreInt *pre;
    int *lm_id, *lm_start, *lm_nobs, *lm_est, *lm_solve, *lm_dyn, *lm_order, *lm_free, *lm_tmp, *lm_pidx, *lm_aidx, *lm_relo;
    double *lm_depth, *lm_obs, *feat, *cfeat, *relo_xy, *relo_mp;
    double *H, *Sc, *Hpl, *vec, *Hll, *gl, *lvec, *res;
    int *res_lm, *res_k, *pair_start, *pair_list;
    double *pairblk, *imu_raw;
    double *prior_J, *prior_r, *prior_x0, *prior_H, *prior_rf;
    double *margA, *margB, *margV, *margW, *margE;
    int nres_cap;
};

__device__ Ctx make_ctx(const Batch &B, int s) {
    Ctx c;
    // (round 6: the dimensions come from the kernel arguments -- as fields of *B.cfg they were one dependent global round trip at the top of every
    // workgroup of every kernel before the first useful address could be formed)
    struct { int W, P, LW, NL, NP, NRES, NPRIOR, MX; } C = {B.gW, B.gP, B.gLW, B.gNL, B.gNP, B.gNRES, B.gNPRIOR, B.gMX};
    c.timings = B.timings;
    c.C = B.cfg; c.s = s; c.W = C.W; c.P = C.P; c.LW = C.LW; c.NL = C.NL; c.NLs = C.NL + 8; c.NPR = C.NPRIOR;
    c.be = B.be + s; c.fe = B.fe + s;
    c.pre = B.pre + (size_t)s * (C.W + 2);
    size_t o = (size_t)s * C.NL;
    c.lm_id = B.lm_id + o; c.lm_start = B.lm_start + o; c.lm_nobs = B.lm_nobs + o; c.lm_est = B.lm_est_flag + o;
    c.lm_solve = B.lm_solve_flag + o; c.lm_dyn = B.lm_dyn + o; c.lm_order = B.lm_order + o; c.lm_free = B.lm_free + o;
    c.lm_tmp = B.lm_tmp + o; c.lm_pidx = B.lm_pidx + o; c.lm_aidx = B.lm_aidx + o; c.lm_relo = B.lm_relo + o;
    c.relo_xy = B.relo_xy + o * 2; c.relo_mp = B.relo_mp + (size_t)s * C.NP * 3;
    c.lm_depth = B.lm_depth + o; c.feat = B.para_feat + o; c.cfeat = B.cand_feat + o;
    c.lm_obs = B.lm_obs + o * (C.W + 1) * VIO_OBS_D;
    c.H = B.H + (size_t)s * C.LW * C.LW; c.Sc = B.Sc + (size_t)s * C.LW * C.LW; c.Hpl = B.Hpl + (size_t)s * (C.NL + 8) * C.LW;
    c.vec = B.vec + (size_t)s * VEC_SLOTS * C.LW;
    c.Hll = B.Hll + (size_t)s * (C.NL + 8); c.gl = B.gl + (size_t)s * (C.NL + 8); c.lvec = B.lvec + (size_t)s * (C.NL + 8) * 8;
    c.nres_cap = C.NRES;
    c.res = B.res + (size_t)s * c.nres_cap * 42;
    c.res_lm = B.res_lm + (size_t)s * c.nres_cap; c.res_k = B.res_k + (size_t)s * c.nres_cap;
    int np = (C.W + 1) * (C.W + 1);
    c.pair_start = B.pair_start + (size_t)s * (np + 1); c.pair_list = B.pair_list + (size_t)s * c.nres_cap;
    c.pairblk = B.pairblk + (size_t)s * np * 210;
    c.imu_raw = B.imu_raw + (size_t)s * C.W * 15 * 31;
    int n = C.NPRIOR;
    c.prior_J = B.prior_J + (size_t)s * n * n; c.prior_r = B.prior_r + (size_t)s * n;
    c.prior_x0 = B.prior_x0 + (size_t)s * (C.W * 7 + 17); c.prior_H = B.prior_H + (size_t)s * n * n;
    c.prior_rf = B.prior_rf + (size_t)s * n;
    int mq = 15 + n;
    c.margA = B.margA + (size_t)s * mq * mq; c.margB = B.margB + (size_t)s * mq;
    c.margV = B.margV + (size_t)s * n * n; c.margW = B.margW + (size_t)s * (n + 16) * (n + 16);
    c.margE = C.MX > 0 ? B.margE + (size_t)s * ((size_t)3 * C.MX * C.MX + (size_t)n * C.MX) : nullptr;
    return c;
}

__device__ __forceinline__ double *obs_ptr(const Ctx &c, int slot, int frame) {
    int W1 = c.W + 1;
    int ph = (frame + c.be->ring_base) % W1;
    return c.lm_obs + ((size_t)slot * W1 + ph) * VIO_OBS_D;
}
__device__ __forceinline__ bool in_problem(const Ctx &c, int slot) {
    return !c.lm_dyn[slot] && c.lm_nobs[slot] >= 2 && c.lm_start[slot] < c.W - 2;
}

}  // namespace
#include "be_linalg.h"
namespace {


// ------------------------------------------------------------------ IntegrationBase::propagate, block-cooperative
// LDS workspace: sJ sP sFJ sFP (225 each) sF (225) sV (270)
struct PreWork { double J[225], Pm[225], FJ[225], FP[225], F[225], V[270]; };
// PI_CH (kernels.h): samples per chunk of the pipelined propagation in be_ingest (F / V of a chunk live in LDS: 8 x 495 doubles)
__device__ __forceinline__ void preint_load(const PreInt &p, PreWork &w) {
    for (int i = threadIdx.x; i < 225; i += blockDim.x) { w.J[i] = p.jac[i]; w.Pm[i] = p.cov[i]; }
    __syncthreads();
}
// Also refreshes the whitening matrix of the IMU factor: M = chol(cov)^-1 (lower triangular), M^T M = cov^-1.
// The reference uses LLT(cov^-1).L^T (imu_factor.h:66-69); both satisfy M^T M = cov^-1, so J^T J, J^T r and |r|^2 - the only
// quantities the solver and the marginalisation consume - are identical up to round-off (DESIGN.md "equivalent whitening").
__device__ __forceinline__ void preint_store(PreInt &p, PreWork &w) {
    const int t = threadIdx.x;
    __syncthreads();
    for (int i = t; i < 225; i += blockDim.x) { p.jac[i] = w.J[i]; p.cov[i] = w.Pm[i]; }
    if (t < 225) { int i = t / 15, j = t - i * 15; w.FJ[t] = 0.5 * (w.Pm[i * 15 + j] + w.Pm[j * 15 + i]); }
    __syncthreads();
    for (int j = 0; j < 15; j++) {
        if (t == 0) { double d = w.FJ[j * 15 + j]; w.FJ[j * 15 + j] = (d > 0.0 && isfinite(d)) ? sqrt(d) : 0.0; }
        __syncthreads();
        double l = w.FJ[j * 15 + j];
        if (t > j && t < 15) w.FJ[t * 15 + j] = l > 0.0 ? w.FJ[t * 15 + j] / l : 0.0;
        __syncthreads();
        if (t < 225) { int i = t / 15, k = t - i * 15; if (k > j && i >= k) w.FJ[t] -= w.FJ[i * 15 + j] * w.FJ[k * 15 + j]; }
        __syncthreads();
    }
    if (t < 225) w.FP[t] = 0;
    __syncthreads();
    if (t < 15) {
        double x[15];
        bool ok = true;
        for (int i = 0; i < 15; i++) ok = ok && w.FJ[i * 15 + i] > 0.0;
        for (int i = 0; i < 15; i++) {
            double sacc = (i == t) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) sacc -= w.FJ[i * 15 + k] * x[k];
            x[i] = ok ? sacc / w.FJ[i * 15 + i] : 0.0;
            w.FP[i * 15 + t] = x[i];
        }
    }
    __syncthreads();
    if (t < 225) p.sqrt_info[t] = w.FP[t];
    __syncthreads();
}
// one propagate(dt, acc, gyr) on the LDS-resident jacobian/covariance; p's small state is updated by thread 0
__device__ void preint_propagate(PreInt &p, PreWork &w, const vio_config &c, double dt, v3 acc1, v3 gyr1) {
    const int t = threadIdx.x;
    if (t == 0) {
        bf::PreintStep o = bf::preint_midpoint(p, dt, acc1, gyr1, w.F, w.V);
        st3(p.dp, o.dp); st3(p.dv, o.dv);
        quat q = qnormalized(o.dq);
        p.dq[0] = q.w; p.dq[1] = q.x; p.dq[2] = q.y; p.dq[3] = q.z;
        p.sum_dt += dt;
        st3(p.acc0, acc1); st3(p.gyr0, gyr1);
    }
    __syncthreads();
    if (t < 225) {
        int i = t / 15, j = t - i * 15;
        double s = 0, s2 = 0;
        for (int k = 0; k < 15; k++) { s += w.F[i * 15 + k] * w.J[k * 15 + j]; s2 += w.F[i * 15 + k] * w.Pm[k * 15 + j]; }
        w.FJ[t] = s; w.FP[t] = s2;
    }
    __syncthreads();
    if (t < 225) {
        int i = t / 15, j = t - i * 15;
        double s = 0;
        for (int k = 0; k < 15; k++) s += w.FP[i * 15 + k] * w.F[j * 15 + k];
        double nn[6] = {c.acc_n * c.acc_n, c.gyr_n * c.gyr_n, c.acc_n * c.acc_n, c.gyr_n * c.gyr_n, c.acc_w * c.acc_w, c.gyr_w * c.gyr_w};
        double tt = 0;
        for (int k = 0; k < 18; k++) tt += w.V[i * 18 + k] * nn[k / 3] * w.V[j * 18 + k];
        w.J[t] = w.FJ[t];
        w.Pm[t] = s + tt;
    }
    __syncthreads();
}

// n propagate steps on the LDS-resident jacobian / covariance (preint_load'ed into w), pipelined in chunks of PI_CH samples like the
// per-frame integration of be_ingest: thread 0 runs the short delta-state recursion of a chunk and keeps the state before every step,
// one lane per sample builds that step's F and V from it, the jacobian / covariance recursion consumes them step by step.  The
// arithmetic of n calls of preint_propagate (be_factors.h preint_state_step / preint_step_FV are the two halves of preint_midpoint)
// with the serial part of a step shrunk from the whole midpoint step to the recursion.  append: the samples are also filed into p's
// own buffers (IntegrationBase::push_back).  Used for the merge of MARGIN_SECOND_NEW (estimator.cpp:1651-1687), where the step-by-step
// version was 100 of the 190 us of that branch.
// PREINT_MANY_LDS_DOUBLES (kernels.h): F and V of a chunk, LDS the caller lends (the marginalisation's tile region is free by then)
__device__ void preint_propagate_many(PreInt &p, PreWork &w, const vio_config &cfg, int n, const double *dt_src, const double (*acc_src)[3],
                                      const double (*gyr_src)[3], bool append, double *lds_fv) {
    const int t = threadIdx.x;
    double (*pm_F)[225] = (double (*)[225])lds_fv;
    double (*pm_V)[270] = (double (*)[270])(lds_fv + PI_CH * 225);
    __shared__ double pm_dt[PI_CH], pm_acc[PI_CH][3], pm_gyr[PI_CH][3];
    __shared__ bf::PreintPre pm_pre[PI_CH];
    const int nb0 = p.n_buf;
    const v3 lba = ld3(p.lin_ba), lbg = ld3(p.lin_bg);
    quat s_dq = mkq(p.dq[0], p.dq[1], p.dq[2], p.dq[3]);
    v3 s_dp = ld3(p.dp), s_dv = ld3(p.dv), s_a0 = ld3(p.acc0), s_g0 = ld3(p.gyr0);
    double s_sum = p.sum_dt;
    __syncthreads();
    for (int q0 = 0; q0 < n; q0 += PI_CH) {
        const int m = min(PI_CH, n - q0);
        if (t < m) {
            const int q = q0 + t;
            pm_dt[t] = dt_src[q];
            for (int k = 0; k < 3; k++) { pm_acc[t][k] = acc_src[q][k]; pm_gyr[t][k] = gyr_src[q][k]; }
            const int nb = nb0 + q;
            if (append && nb < VIO_IMU_SLOT_CAP) { p.dt_buf[nb] = dt_src[q]; for (int k = 0; k < 3; k++) { p.acc_buf[nb][k] = acc_src[q][k]; p.gyr_buf[nb][k] = gyr_src[q][k]; } }
        }
        __syncthreads();
        if (t == 0)
            for (int k = 0; k < m; k++) {
                const v3 acc = ld3(pm_acc[k]), gyr = ld3(pm_gyr[k]);
                pm_pre[k].dq = s_dq; pm_pre[k].acc0 = s_a0; pm_pre[k].gyr0 = s_g0;
                bf::preint_state_step(s_dq, s_dp, s_dv, s_a0, s_g0, lba, lbg, pm_dt[k], acc, gyr);
                s_sum += pm_dt[k];
                s_a0 = acc; s_g0 = gyr;
            }
        __syncthreads();
        if (t < m) bf::preint_step_FV(pm_pre[t], lba, lbg, pm_dt[t], ld3(pm_acc[t]), ld3(pm_gyr[t]), pm_F[t], pm_V[t]);
        __syncthreads();
        for (int k = 0; k < m; k++) {
            const double *Fk = pm_F[k], *Vk = pm_V[k];
            if (t < 225) {
                int i = t / 15, j = t - i * 15;
                double s1 = 0, s2 = 0;
                for (int u = 0; u < 15; u++) { s1 += Fk[i * 15 + u] * w.J[u * 15 + j]; s2 += Fk[i * 15 + u] * w.Pm[u * 15 + j]; }
                w.FJ[t] = s1; w.FP[t] = s2;
            }
            __syncthreads();
            if (t < 225) {
                int i = t / 15, j = t - i * 15;
                double s1 = 0;
                for (int u = 0; u < 15; u++) s1 += w.FP[i * 15 + u] * Fk[j * 15 + u];
                double nn[6] = {cfg.acc_n * cfg.acc_n, cfg.gyr_n * cfg.gyr_n, cfg.acc_n * cfg.acc_n, cfg.gyr_n * cfg.gyr_n, cfg.acc_w * cfg.acc_w, cfg.gyr_w * cfg.gyr_w};
                double tt = 0;
                for (int u = 0; u < 18; u++) tt += Vk[i * 18 + u] * nn[u / 3] * Vk[j * 18 + u];
                w.J[t] = w.FJ[t];
                w.Pm[t] = s1 + tt;
            }
            __syncthreads();
        }
    }
    if (t == 0) {
        st3(p.dp, s_dp); st3(p.dv, s_dv);
        p.dq[0] = s_dq.w; p.dq[1] = s_dq.x; p.dq[2] = s_dq.y; p.dq[3] = s_dq.z;
        p.sum_dt = s_sum;
        st3(p.acc0, s_a0); st3(p.gyr0, s_g0);
        if (append) p.n_buf = min(nb0 + n, VIO_IMU_SLOT_CAP);
    }
    __syncthreads();
}

// 4x4 / small symmetric cyclic Jacobi (row-cyclic, as oracle/om.h sym_eig); A destroyed, eigenvalues unsorted in A diag
// stable compaction of the landmark order list; flags[k] = keep. Freed slots go back to the free stack.
__device__ void lm_compact(Ctx &c, int *flags, int *offs, int *scratch) {
    int n = c.be->n_lm;
    int kept = block_scan_flags(flags, n, offs, scratch);
    const int t = threadIdx.x, nt = blockDim.x;
    int nfree0 = c.be->n_free;
    for (int k = t; k < n; k += nt) c.lm_tmp[k] = c.lm_order[k];
    __syncthreads();
    for (int k = t; k < n; k += nt) {
        int slot = c.lm_tmp[k];
        if (flags[k]) c.lm_order[offs[k]] = slot;
        else c.lm_free[nfree0 + (k - offs[k])] = slot;
    }
    __syncthreads();
    if (t == 0) { c.be->n_lm = kept; c.be->n_free = nfree0 + (n - kept); }
    __syncthreads();
}

}  // namespace

// ------------------------------------------------------------------ VO mode: FeatureManager::initFramePoseByPnP
// cv::Rodrigues and its derivative (closed form, Gallego & Yezzi 2015), as the host side of the dynamic initialisation restates them
__device__ m3 pnp_rodrigues(v3 r) {
    const double th = nrm(r);
    if (th < 2.220446049250313e-16) return eye();
    const v3 k = scl(1.0 / th, r);
    const double cth = cos(th), sth = sin(th), c1 = 1 - cth;
    const m3 K = skew(k);
    const double kk[3] = {k.x, k.y, k.z};
    m3 R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.a[i * 3 + j] = (i == j ? cth : 0.0) + c1 * kk[i] * kk[j] + sth * K.a[i * 3 + j];
    return R;
}
__device__ v3 pnp_rodrigues_inv(const m3 &R) {
    v3 r = mk(R.a[7] - R.a[5], R.a[2] - R.a[6], R.a[3] - R.a[1]);
    const double sn = sqrt(dot(r, r) * 0.25);
    double cs = (R.a[0] + R.a[4] + R.a[8] - 1) * 0.5;
    cs = cs > 1 ? 1 : (cs < -1 ? -1 : cs);
    const double th = acos(cs);
    if (sn < 1e-5) {
        if (cs > 0) return mk(0, 0, 0);
        v3 v;
        v.x = sqrt(fmax((R.a[0] + 1) * 0.5, 0.0));
        v.y = sqrt(fmax((R.a[4] + 1) * 0.5, 0.0)) * (R.a[1] < 0 ? -1.0 : 1.0);
        v.z = sqrt(fmax((R.a[8] + 1) * 0.5, 0.0)) * (R.a[2] < 0 ? -1.0 : 1.0);
        if (fabs(v.x) < fabs(v.y) && fabs(v.x) < fabs(v.z) && (R.a[5] > 0) != (v.y * v.z > 0)) v.z = -v.z;
        return scl(th / nrm(v), v);
    }
    return scl(th / (2 * sn), r);
}
// The Levenberg-Marquardt core of cv::solvePnP(ITERATIVE): refines par[6] = (rvec, tvec) in LDS over n pairs pts[5 n] = (X, Y, Z, u, v).
// prev: 6 doubles of LDS, sw: >= (waves x 28) doubles of LDS.  Every thread of the block takes part.
__device__ void pnp_refine_block(const double *pts, int n, double *sh_par, double *sh_prev, double *sw) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6, nw = nt >> 6;
    // reduction of NV values per thread: wave DPP sums, then a fixed-order sum over the waves
    auto reduce = [&](double *v, int NV) {
        for (int q = 0; q < NV; q++) v[q] = wave_sum_dpp(v[q]);
        __syncthreads();
        if (lane == 0) for (int q = 0; q < NV; q++) sw[wave * 28 + q] = v[q];
        __syncthreads();
        for (int q = 0; q < NV; q++) { double r = 0; for (int w = 0; w < nw; w++) r += sw[w * 28 + q]; v[q] = r; }
    };
    // residuals (and Jacobian sums) at sh_par; every thread returns the same values
    auto project = [&](bool jac, double *acc /*27: JtJ upper 21 + JtErr 6*/) -> double {
        const v3 r = mk(sh_par[0], sh_par[1], sh_par[2]), tt = mk(sh_par[3], sh_par[4], sh_par[5]);
        const m3 Rm = pnp_rodrigues(r);
        m3 dR[3];
        if (jac) {
            const double th2 = dot(r, r);
            for (int i = 0; i < 3; i++) {
                const v3 e = mk(i == 0, i == 1, i == 2);
                if (th2 < 1e-24) { dR[i] = skew(e); continue; }
                const v3 w = cross(r, mul(sub(eye(), Rm), e));
                dR[i] = scl(1.0 / th2, mul(add(scl(get(r, i), skew(r)), skew(w)), Rm));
            }
        }
        double v[28];
        for (int q = 0; q < 28; q++) v[q] = 0;
        for (int i = t; i < n; i += nt) {
            const double *q = pts + (size_t)i * 5;
            const v3 Y = add(mul(Rm, mk(q[0], q[1], q[2])), tt);
            const double iz = 1.0 / Y.z, x = Y.x * iz, y = Y.y * iz;
            const double e0 = x - q[3], e1 = y - q[4];
            v[27] += e0 * e0 + e1 * e1;
            if (!jac) continue;
            double J0[6], J1[6];
            for (int k = 0; k < 3; k++) {
                const v3 d = mul(dR[k], mk(q[0], q[1], q[2]));
                J0[k] = iz * d.x - x * iz * d.z;
                J1[k] = iz * d.y - y * iz * d.z;
            }
            J0[3] = iz; J0[4] = 0; J0[5] = -x * iz;
            J1[3] = 0; J1[4] = iz; J1[5] = -y * iz;
            int e = 0;
            for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) v[e++] += J0[a] * J0[b] + J1[a] * J1[b];
            for (int a = 0; a < 6; a++) v[21 + a] += J0[a] * e0 + J1[a] * e1;
        }
        reduce(v, jac ? 28 : 28);
        if (jac) for (int q = 0; q < 27; q++) acc[q] = v[q];
        return sqrt(v[27]);
    };
    double acc[27];
    int lambda_lg10 = -3, iters = 0;
    double prev_err = 0;
    auto take_step = [&]() {   // thread 0: param = prev - pinv(JtJ with damped diagonal) JtErr
        if (t == 0) {
            const double lambda = exp(lambda_lg10 * 2.302585092994046);
            double N[36], V[36];
            int e = 0;
            for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) { N[a * 6 + b] = acc[e]; N[b * 6 + a] = acc[e]; e++; }
            for (int a = 0; a < 6; a++) N[a * 7] *= 1.0 + lambda;
            jacobi_small(N, V, 6);
            double wmax = 0;
            for (int k = 0; k < 6; k++) wmax = fmax(wmax, fabs(N[k * 7]));
            double d[6] = {0, 0, 0, 0, 0, 0};
            for (int k = 0; k < 6; k++) {
                const double wk = N[k * 7];
                if (!(fabs(wk) > wmax * 2 * 2.220446049250313e-16 * 6)) continue;
                double sacc = 0;
                for (int i = 0; i < 6; i++) sacc += V[i * 6 + k] * acc[21 + i];
                sacc /= wk;
                for (int i = 0; i < 6; i++) d[i] += V[i * 6 + k] * sacc;
            }
            for (int i = 0; i < 6; i++) sh_par[i] = sh_prev[i] - d[i];
        }
        __syncthreads();
    };
    for (;;) {
        const double e_at = project(true, acc);
        if (t == 0) for (int i = 0; i < 6; i++) sh_prev[i] = sh_par[i];
        __syncthreads();
        take_step();
        if (iters == 0) prev_err = e_at;
        bool done = false;
        for (;;) {
            const double e = project(false, acc);
            if (e > prev_err && ++lambda_lg10 <= 16) { take_step(); continue; }
            lambda_lg10 = max(lambda_lg10 - 1, -16);
            double dn = 0, pn = 0;
            for (int i = 0; i < 6; i++) { dn += (sh_par[i] - sh_prev[i]) * (sh_par[i] - sh_prev[i]); pn += sh_prev[i] * sh_prev[i]; }
            if (++iters >= 20 || sqrt(dn) / sqrt(pn) < 1.1920928955078125e-07) done = true;
            prev_err = e;
            break;
        }
        if (done) break;
    }
    __syncthreads();
}

// FeatureManager::initFramePoseByPnP + solvePoseByPnP (feature_manager.cpp:545-642): cv::solvePnP(SOLVEPNP_ITERATIVE,
// useExtrinsicGuess) = CvLevMarq on (rvec, tvec) with lambda = 10^k, diagonal x (1 + lambda), at most 20 iterations, FLT_EPSILON
// on the relative parameter change.  Block-cooperative: one thread per 3-D / 2-D pair evaluates its two residual rows and 2 x 6
// Jacobian, the 27 sums of J^T J / J^T e are reduced in a fixed order, thread 0 solves the damped 6 x 6 system through its
// eigen-decomposition (cv::solve DECOMP_SVD).  pts: scratch in HBM, 5 doubles per pair.  sw: >= 16 * 28 + 64 doubles of LDS.
__device__ void init_frame_pose_by_pnp(Ctx &c, int fc, double *pts, double *sw) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6, nw = nt >> 6;
    BeSeq &be = *c.be;
    if (fc <= 0) return;
    __shared__ int sh_n;
    __shared__ double sh_par[6], sh_prev[6], sh_flag[4];
    if (t == 0) sh_n = 0;
    __syncthreads();
    const m3 ric = ldm(be.ric);
    const v3 tic = ld3(be.tic);
    const int nlm = be.n_lm;
    // pairs in list order (deterministic): flags + scan
    int *flag = c.lm_pidx, *offs = c.lm_aidx;   // free outside the solve
    __shared__ int scan_scratch[2 * 256 + 8];
    for (int k = t; k < nlm; k += nt) {
        const int slot = c.lm_order[k];
        const int index = fc - c.lm_start[slot];
        flag[k] = (c.lm_depth[slot] > 0 && index >= 0 && c.lm_nobs[slot] >= index + 1) ? 1 : 0;
    }
    __syncthreads();
    const int n = block_scan_flags(flag, nlm, offs, scan_scratch);
    for (int k = t; k < nlm; k += nt) {
        if (!flag[k]) continue;
        const int slot = c.lm_order[k], st = c.lm_start[slot];
        const double *o0 = obs_ptr(c, slot, st), *oi = obs_ptr(c, slot, fc);
        const double d = c.lm_depth[slot];
        const v3 pc = add(mul(ric, mk(o0[0] * d, o0[1] * d, o0[2] * d)), tic);
        const v3 pw = add(mul(ldm(be.Rs[st]), pc), ld3(be.Ps[st]));
        double *q = pts + (size_t)offs[k] * 5;
        q[0] = (double)(float)pw.x; q[1] = (double)(float)pw.y; q[2] = (double)(float)pw.z;   // cv::Point3f / cv::Point2f
        q[3] = (double)(float)oi[0]; q[4] = (double)(float)oi[1];
    }
    __syncthreads();
    if (n < 4) return;
    if (t == 0) {
        const m3 RCam = mul(ldm(be.Rs[fc - 1]), ric);
        const v3 PCam = add(mul(ldm(be.Rs[fc - 1]), tic), ld3(be.Ps[fc - 1]));
        const m3 R0 = tr(RCam);
        const v3 P0 = neg(mul(R0, PCam));
        const v3 r = pnp_rodrigues_inv(R0);
        sh_par[0] = r.x; sh_par[1] = r.y; sh_par[2] = r.z; sh_par[3] = P0.x; sh_par[4] = P0.y; sh_par[5] = P0.z;
    }
    __syncthreads();
    pnp_refine_block(pts, n, sh_par, sh_prev, sw);
    __syncthreads();
    if (t == 0) {
        bool fin = true;
        for (int i = 0; i < 6; i++) fin = fin && isfinite(sh_par[i]);
        if (fin) {
            const m3 Rn = pnp_rodrigues(mk(sh_par[0], sh_par[1], sh_par[2]));
            const v3 tn = mk(sh_par[3], sh_par[4], sh_par[5]);
            const m3 RCam = tr(Rn);
            const v3 PCam = mul(RCam, neg(tn));
            stm(be.Rs[fc], mul(RCam, tr(ric)));
            st3(be.Ps[fc], add(neg(mul(RCam, mul(tr(ric), tic))), PCam));
        }
    }
    __syncthreads();
}

// FeatureManager::triangulateWithDepth (feature_manager.cpp:386-543), one thread per landmark of the list
__device__ void triangulate_with_depth(Ctx &c, int nlm) {
    const int t = threadIdx.x, nt = blockDim.x;
    BeSeq &be = *c.be;
    const vio_config &cfg = c.C->c;
        m3 ric = ldm(be.ric);
        v3 tic = ld3(be.tic);
        for (int k = t; k < nlm; k += nt) {
            int slot = c.lm_order[k];
            if (c.lm_depth[slot] > 0) continue;
            if (!in_problem(c, slot)) continue;
            int imu_i = c.lm_start[slot], K = c.lm_nobs[slot];
            v3 trr = add(ld3(be.Ps[imu_i]), mul(ldm(be.Rs[imu_i]), tic));
            m3 Rr = mul(ldm(be.Rs[imu_i]), ric);
            double vsum = 0, rsum = 0;
            int vn = 0, rn = 0, no_depth = 0;
            for (int a = 0; a < K; a++) {
                const double *oa = obs_ptr(c, slot, imu_i + a);
                if (oa[8] == 0) { no_depth++; continue; }
                v3 t0 = add(ld3(be.Ps[imu_i + a]), mul(ldm(be.Rs[imu_i + a]), tic));
                m3 R0 = mul(ldm(be.Rs[imu_i + a]), ric);
                v3 point0 = scl(oa[8], mk(oa[0], oa[1], oa[2]));
                point0 = mk(oa[0] * oa[8], oa[1] * oa[8], oa[2] * oa[8]);
                v3 t2r = mul(tr(Rr), sub(t0, trr));
                m3 R2r = mul(tr(Rr), R0);
                for (int b = 0; b < K; b++) {
                    if (a == b) continue;
                    const double *ob = obs_ptr(c, slot, imu_i + b);
                    v3 t1 = add(ld3(be.Ps[imu_i + b]), mul(ldm(be.Rs[imu_i + b]), tic));
                    m3 R1 = mul(ldm(be.Rs[imu_i + b]), ric);
                    v3 t20 = mul(tr(R0), sub(t1, t0));
                    m3 R20 = mul(tr(R0), R1);
                    v3 pp = sub(mul(tr(R20), point0), mul(tr(R20), t20));
                    double rx = ob[0] - pp.x / pp.z, ry = ob[1] - pp.y / pp.z;
                    if (sqrt(rx * rx + ry * ry) < 10.0 / 460) {
                        v3 pr = add(mul(R2r, point0), t2r);
                        if (oa[8] > cfg.depth_max) { rsum += pr.z; rn++; } else { vsum += pr.z; vn++; }
                    }
                }
            }
            double dep;
            int ef;
            if (vn == 0) {
                if (rn == 0) {
                    if (no_depth == K) {
                        double AtA[16], Vv[16];
                        for (int q = 0; q < 16; q++) AtA[q] = 0;
                        v3 t0 = add(ld3(be.Ps[imu_i]), mul(ldm(be.Rs[imu_i]), tic));
                        m3 R0 = mul(ldm(be.Rs[imu_i]), ric);
                        for (int a = 0; a < K; a++) {
                            const double *oa = obs_ptr(c, slot, imu_i + a);
                            v3 t1 = add(ld3(be.Ps[imu_i + a]), mul(ldm(be.Rs[imu_i + a]), tic));
                            m3 R1 = mul(ldm(be.Rs[imu_i + a]), ric);
                            v3 tt = mul(tr(R0), sub(t1, t0));
                            m3 Rt = tr(mul(tr(R0), R1));
                            v3 mt = neg(mul(Rt, tt));
                            double Pm[12];
                            for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) Pm[r * 4 + q] = Rt.a[r * 3 + q]; Pm[r * 4 + 3] = get(mt, r); }
                            v3 f = mk(oa[0], oa[1], oa[2]);
                            f = scl(1.0 / nrm(f), f);
                            f = mk(oa[0] / nrm(mk(oa[0], oa[1], oa[2])), oa[1] / nrm(mk(oa[0], oa[1], oa[2])), oa[2] / nrm(mk(oa[0], oa[1], oa[2])));
                            double row[8];
                            for (int q = 0; q < 4; q++) { row[q] = f.x * Pm[8 + q] - f.z * Pm[q]; row[4 + q] = f.y * Pm[8 + q] - f.z * Pm[4 + q]; }
                            for (int rr = 0; rr < 2; rr++)
                                for (int x = 0; x < 4; x++) for (int y = 0; y < 4; y++) AtA[x * 4 + y] += row[rr * 4 + x] * row[rr * 4 + y];
                        }
                        jacobi_small(AtA, Vv, 4);
                        int mi = 0;
                        for (int q = 1; q < 4; q++) if (AtA[q * 5] < AtA[mi * 5]) mi = q;
                        double svd_method = Vv[2 * 4 + mi] / Vv[3 * 4 + mi];
                        dep = svd_method < cfg.depth_min ? cfg.depth_max : svd_method;
                        ef = 2;
                    } else
                        continue;
                } else { dep = rsum / rn; ef = 0; }
            } else { dep = vsum / vn; ef = 1; }
            if (dep < 0.1) { dep = cfg.init_depth; ef = 0; }
            c.lm_depth[slot] = dep;
            c.lm_est[slot] = ef;
        }
}

// pre_integrations[j]->repropagate(Vector3d::Zero(), Bgs[j]) for every window slot (estimator.cpp:275-279, :829-836), block-cooperative
__device__ void repropagate_window(Ctx &c, PreWork &pw) {
    const int t = threadIdx.x, nt = blockDim.x, W = c.W;
    BeSeq &be = *c.be;
    const vio_config &cfg = c.C->c;
    for (int j = 0; j <= W; j++) {
            PreInt &p = c.pre[be.pre_idx[j]];
            if (!p.valid) continue;
            if (t == 0) {
                int nb = p.n_buf;
                v3 la = ld3(p.lin_acc), lg = ld3(p.lin_gyr);
                p.sum_dt = 0;
                st3(p.acc0, la); st3(p.gyr0, lg);
                p.dp[0] = p.dp[1] = p.dp[2] = 0; p.dv[0] = p.dv[1] = p.dv[2] = 0;
                p.dq[0] = 1; p.dq[1] = p.dq[2] = p.dq[3] = 0;
                p.lin_ba[0] = p.lin_ba[1] = p.lin_ba[2] = 0;
                st3(p.lin_bg, ld3(be.Bgs[j]));
                p.n_buf = nb;
            }
            for (int i = t; i < 225; i += nt) { pw.J[i] = ((i / 15) == (i % 15)) ? 1.0 : 0.0; pw.Pm[i] = 0; }
            __syncthreads();
            for (int q = 0; q < p.n_buf; q++) preint_propagate(p, pw, cfg, p.dt_buf[q], ld3(p.acc_buf[q]), ld3(p.gyr_buf[q]));
            preint_store(p, pw);
        }
}

// ====================================================================================================== be_ingest
// src.ids == NULL: the feature map packaged by the last vio_track / front-end of vio_feed (B.obs_id / B.obs / FeSeq);
// otherwise a caller-supplied map (vio_process_obs = Estimator::processImage(image, header), estimator.h:46): n_obs[s] < 0 skips s.
__global__ __launch_bounds__(256) void be_ingest_kernel(Batch B, const uint16_t *depth_base, size_t depth_stride, IngestSrc src) {
    const int s = blockIdx.x + B.s0, t = threadIdx.x, nt = blockDim.x;
    Ctx c = make_ctx(B, s);
    const DevCfg &C = *B.cfg;
    const vio_config &cfg = C.c;
    BeSeq &be = *c.be;
    FeSeq &fe = *c.fe;
    const int W = c.W;
    __shared__ int sh_i[8];
    __shared__ double sh_d[8];
    __shared__ int scratch[2 * 256 + 8];
    __shared__ double sred[256];
    __shared__ PreWork pw;
    PH_INIT;
    const bool ext = src.ids != nullptr;
    const int ext_n = ext ? src.n_obs[s] : 0;
    const double in_stamp = ext ? src.stamps[s] : fe.cur_time;
    if (t == 0) {
        // the state the next call's tracker sees with tracker lag 1 (that tracker runs while THIS frame is still being optimised)
        for (int k = 0; k < 3; k++) be.track_Bg[k] = be.latest_Bg[k];
        be.track_td = be.td;
        be.imu_count_ingest = be.imu_count;
        be.do_solve = 0; be.do_marg = 0; be.processed = 0; be.rebooted = 0; be.init_frame = 0; be.dyn_failed = 0;
        be.status_code = (!ext && fe.n_forw == -2) ? VIO_NEED_IMU : VIO_OK;
        be.cur_stamp = in_stamp;
        be.overflow = 0;
        if (be.imu_count - be.imu_head > C.NIMU) { be.imu_head = be.imu_count - C.NIMU; be.overflow |= 16; }  // samples overwritten in the ring before they were consumed
    }
    __syncthreads();
    if (ext ? ext_n <= 0 : (fe.n_forw < 0 || !fe.publish_ok)) return;
    // ---- IMU availability (estimator.cpp:178-183, :1882-1888)
    const double *it = B.imu_t + (size_t)s * C.NIMU;
    const double *ia = B.imu_acc + (size_t)s * C.NIMU * 3, *ig = B.imu_gyr + (size_t)s * C.NIMU * 3;
    double stamp = in_stamp, curTime = stamp + be.td;
    if (cfg.use_imu) {
        bool have = be.imu_count > be.imu_head;
        double back_t = have ? it[(be.imu_count - 1) % C.NIMU] : -1e300;
        if (!(have && curTime <= back_t)) {
            if (t == 0) be.status_code = VIO_NEED_IMU;
            return;
        }
    }
    const uint16_t *depth = depth_base + (size_t)s * depth_stride;
    const int fc = be.frame_count;
    // ---- addFeatureCheckParallax (feature_manager.cpp:56-123)
    int nobs = ext ? min(ext_n, C.NP) : fe.n_obs, nlm = be.n_lm;
    const int *o_id = ext ? src.ids + (size_t)s * src.cap : B.obs_id + (size_t)s * C.NP;
    const double *o = ext ? src.obs + (size_t)s * src.cap * 7 : B.obs + (size_t)s * C.NP * 7;
    int *flag = c.lm_tmp;         // new-landmark flags per observation (NP <= NL is checked at create)
    int *offs = c.lm_pidx;        // temporary
    int tracked = 0;
    // id -> slot lookup (feature_manager.cpp:66-67 find_if over the list).  The list is NOT sorted by id: a landmark removed by
    // outlier rejection while the tracker keeps its id is re-appended at the end.  Open-addressing hash table in LDS.
    extern __shared__ int htab[];  // [2 * HT]: keys, values
    const int HT = C.lm_hash_size;
    for (int q = t; q < HT; q += nt) htab[q] = -1;
    __syncthreads();
    for (int k = t; k < nlm; k += nt) {
        int slot = c.lm_order[k], id = c.lm_id[slot];
        unsigned hsh = ((unsigned)id * 2654435761u) & (unsigned)(HT - 1);
        while (atomicCAS(&htab[hsh], -1, id) != -1) hsh = (hsh + 1) & (unsigned)(HT - 1);
        htab[HT + hsh] = slot;
    }
    __syncthreads();
    for (int j = t; j < nobs; j += nt) {
        const double *p = o + (size_t)j * 7;
        unsigned short mm = depth[(size_t)min(max((int)p[4], 0), cfg.height - 1) * cfg.width + min(max((int)p[3], 0), cfg.width - 1)];
        double dmm = mm / 1000.0;
        int isnew = 0;
        if (!(0 < dmm && dmm < cfg.depth_min)) {
            int fid = o_id[j];
            int found = -1;
            unsigned hsh = ((unsigned)fid * 2654435761u) & (unsigned)(HT - 1);
            while (htab[hsh] != -1) {
                if (htab[hsh] == fid) { found = htab[HT + hsh]; break; }
                hsh = (hsh + 1) & (unsigned)(HT - 1);
            }
            if (found >= 0) {
                int k = c.lm_nobs[found];
                if (k <= W) {
                    double *q = obs_ptr(c, found, c.lm_start[found] + k);
                    for (int d = 0; d < 7; d++) q[d] = p[d];
                    q[7] = be.td; q[8] = dmm;
                    c.lm_nobs[found] = k + 1;
                }
                tracked++;
            } else
                isnew = 1;
        }
        flag[j] = isnew;
    }
    __syncthreads();
    PH(106);
    {
        double tr = block_sum((double)tracked, sred);
        if (t == 0) be.last_track_num = (int)tr;
    }
    int nnew = block_scan_flags(flag, nobs, offs, scratch);
    {
        int nfree = be.n_free;
        int can = min(nnew, nfree);
        for (int j = t; j < nobs; j += nt) {
            if (!flag[j] || offs[j] >= can) continue;
            int slot = c.lm_free[nfree - 1 - offs[j]];
            const double *p = o + (size_t)j * 7;
            unsigned short mm = depth[(size_t)min(max((int)p[4], 0), cfg.height - 1) * cfg.width + min(max((int)p[3], 0), cfg.width - 1)];
            c.lm_id[slot] = o_id[j]; c.lm_start[slot] = fc; c.lm_nobs[slot] = 1; c.lm_est[slot] = 0; c.lm_solve[slot] = 0;
            c.lm_dyn[slot] = 0; c.lm_depth[slot] = -1.0;
            double *q = obs_ptr(c, slot, fc);
            for (int d = 0; d < 7; d++) q[d] = p[d];
            q[7] = be.td; q[8] = mm / 1000.0;
            c.lm_order[nlm + offs[j]] = slot;
        }
        __syncthreads();
        if (t == 0) {
            be.n_lm = nlm + can; be.n_free = nfree - can;
            if (can < nnew) be.overflow |= 1;
        }
        __syncthreads();
        nlm = be.n_lm;
    }
    PH(107);
    // parallax (:100-122, :732-768)
    {
        double psum = 0, pnum = 0;
        if (!(fc < 2 || be.last_track_num < 20)) {
            for (int k = t; k < nlm; k += nt) {
                int slot = c.lm_order[k];
                int st = c.lm_start[slot], no = c.lm_nobs[slot];
                if (st <= fc - 2 && st + no - 1 >= fc - 1) {
                    const double *fi = obs_ptr(c, slot, fc - 2), *fj = obs_ptr(c, slot, fc - 1);
                    double dep_i = fi[2], u_i = fi[0] / dep_i, v_i = fi[1] / dep_i;
                    double du = u_i - fj[0], dv = v_i - fj[1];
                    psum += fmax(0.0, sqrt(fmin(du * du + dv * dv, du * du + dv * dv)));
                    pnum += 1.0;
                }
            }
        }
        psum = block_sum(psum, sred);
        pnum = block_sum(pnum, sred);
        if (t == 0) {
            bool kf;
            if (fc < 2 || be.last_track_num < 20) kf = true;
            else if (pnum == 0) kf = true;
            else kf = psum / pnum >= cfg.min_parallax_px / cfg.focal_length;
            be.marginalization_flag = kf ? 0 : 1;
            be.Headers[fc] = stamp;
        }
        __syncthreads();
    }
    PH(108);
    // ---- getIMUInterval + processIMU (estimator.cpp:185-200, 1910-1942, 118-154); VO mode (USE_IMU == 0) has neither
    if (cfg.use_imu && t < 64) {
        // head = the first sample later than prevTime, k = the first one at or after curTime: 64 samples per trip, one lane each (the two scalar
        // loops were a chain of dependent loads, ~14 of them per frame at 200 Hz)
        const int cnt_all = be.imu_count;
        const double prevT = be.prevTime;
        int head = be.imu_head;
        for (;;) {
            const int idx = head + t;
            const bool pass = idx < cnt_all && it[idx % C.NIMU] <= prevT;
            const unsigned long long stop = ~__ballot(pass);
            if (stop) { head += __ffsll((long long)stop) - 1; break; }
            head += 64;
        }
        int k = head;
        for (;;) {
            const int idx = k + t;
            const bool pass = idx < cnt_all && it[idx % C.NIMU] < curTime;
            const unsigned long long stop = ~__ballot(pass);
            if (stop) { k += __ffsll((long long)stop) - 1; break; }
            k += 64;
        }
        sh_i[6] = head; sh_i[7] = k;
    }
    if (cfg.use_imu && t == 0) {
        const int head = sh_i[6], k = sh_i[7];
        sh_i[0] = head;          // first sample of the interval
        sh_i[1] = k - head + 1;  // number of samples incl. the first one with t >= curTime
        be.imu_head = k;         // that last sample is not popped
        be.n_imu_frame = sh_i[1];
        be.imu_frame_head = head;
        if (!be.initFirstPoseFlag) {  // initFirstIMUPose :1890-1909
            v3 aver = mk(0, 0, 0);
            for (int q = 0; q < sh_i[1]; q++) aver = add(aver, ld3(ia + (size_t)((head + q) % C.NIMU) * 3));
            aver = scl(1.0 / (double)sh_i[1], aver);
            m3 R0 = g2R(aver);
            double yaw = R2ypr(R0).x;
            R0 = mul(ypr2R(mk(-yaw, 0, 0)), R0);
            stm(be.Rs[0], R0);
            be.initFirstPoseFlag = 1;
        }
    }
    __syncthreads();
    if (cfg.use_imu) {
        int head = sh_i[0], n = sh_i[1];
        PreInt &P = c.pre[be.pre_idx[fc]];
        if (t == 0) {
            v3 a0 = ld3(ia + (size_t)(head % C.NIMU) * 3), g0 = ld3(ig + (size_t)(head % C.NIMU) * 3);
            if (!be.first_imu) { be.first_imu = 1; st3(be.acc_0, a0); st3(be.gyr_0, g0); }
            if (!P.valid) bf::preint_init(P, ld3(be.acc_0), ld3(be.gyr_0), ld3(be.Bas[fc]), ld3(be.Bgs[fc]));
        }
        __syncthreads();
        if (fc != 0) {
            // IntegrationBase::push_back + Estimator::processIMU for the n samples of the frame, pipelined in chunks of PI_CH samples:
            //   A  thread 0 runs the delta-state recursion (it keeps the state before every step), thread 64 - another wavefront -
            //      the world-state recursion of Ps / Rs / Vs[fc], the remaining threads file the samples into the slot's buffer;
            //   B  one lane per sample builds that step's F (15 x 15) and V (15 x 18) from the state before it;
            //   C  the jacobian / covariance recursion J <- F J, P <- F P F^T + V Q V^T consumes them step by step.
            // The serial part of a sample shrinks from the whole midpoint step to its two short recursions; same arithmetic as
            // preint_propagate (be_factors.h preint_state_step / preint_step_FV are the two halves of preint_midpoint).
            preint_load(P, pw);
            __shared__ double pi_F[PI_CH][225], pi_V[PI_CH][270];
            __shared__ double pi_dt[PI_CH], pi_acc[PI_CH][3], pi_gyr[PI_CH][3];
            __shared__ bf::PreintPre pi_pre[PI_CH];
            const int nb0 = P.n_buf;
            const v3 lba = ld3(P.lin_ba), lbg = ld3(P.lin_bg);
            // thread 0: delta state; thread 64: world state of frame fc
            quat s_dq = mkq(P.dq[0], P.dq[1], P.dq[2], P.dq[3]);
            v3 s_dp = ld3(P.dp), s_dv = ld3(P.dv), s_a0 = ld3(P.acc0), s_g0 = ld3(P.gyr0);
            double s_sum = P.sum_dt;
            v3 w_a0 = ld3(be.acc_0), w_g0 = ld3(be.gyr_0), w_P = ld3(be.Ps[fc]), w_V = ld3(be.Vs[fc]);
            m3 w_R = ldm(be.Rs[fc]);
            const v3 w_Ba = ld3(be.Bas[fc]), w_Bg = ld3(be.Bgs[fc]), w_g = ld3(be.g);
            const double prevTime = be.prevTime;
            __syncthreads();
            PH(111);
            for (int q0 = 0; q0 < n; q0 += PI_CH) {
                const int m = min(PI_CH, n - q0);
                if (t < m) {   // stage the chunk's samples
                    const int q = q0 + t, idx = (head + q) % C.NIMU;
                    const double tq = it[idx];
                    double dt;
                    if (q == 0) dt = tq - prevTime;
                    else if (q == n - 1) dt = curTime - it[(head + q - 1) % C.NIMU];
                    else dt = tq - it[(head + q - 1) % C.NIMU];
                    pi_dt[t] = dt;
                    for (int k = 0; k < 3; k++) { pi_acc[t][k] = ia[(size_t)idx * 3 + k]; pi_gyr[t][k] = ig[(size_t)idx * 3 + k]; }
                    const int nb = nb0 + q;
                    if (nb < VIO_IMU_SLOT_CAP) { P.dt_buf[nb] = dt; for (int k = 0; k < 3; k++) { P.acc_buf[nb][k] = ia[(size_t)idx * 3 + k]; P.gyr_buf[nb][k] = ig[(size_t)idx * 3 + k]; } }
                }
                __syncthreads();
                if (t == 0) {
                    for (int k = 0; k < m; k++) {
                        const v3 acc = ld3(pi_acc[k]), gyr = ld3(pi_gyr[k]);
                        pi_pre[k].dq = s_dq; pi_pre[k].acc0 = s_a0; pi_pre[k].gyr0 = s_g0;
                        bf::preint_state_step(s_dq, s_dp, s_dv, s_a0, s_g0, lba, lbg, pi_dt[k], acc, gyr);
                        s_sum += pi_dt[k];
                        s_a0 = acc; s_g0 = gyr;
                    }
                } else if (t == 64) {
                    for (int k = 0; k < m; k++) {   // Estimator::processIMU (estimator.cpp:142-152)
                        const v3 acc = ld3(pi_acc[k]), gyr = ld3(pi_gyr[k]);
                        const double dt = pi_dt[k];
                        v3 un_acc_0 = sub(mul(w_R, sub(w_a0, w_Ba)), w_g);
                        v3 un_gyr = sub(scl(0.5, add(w_g0, gyr)), w_Bg);
                        w_R = mul(w_R, q2R(deltaQ(scl(dt, un_gyr))));
                        v3 un_acc_1 = sub(mul(w_R, sub(acc, w_Ba)), w_g);
                        v3 un_acc = scl(0.5, add(un_acc_0, un_acc_1));
                        w_P = add(add(w_P, scl(dt, w_V)), scl(dt * dt, scl(0.5, un_acc)));
                        w_V = add(w_V, scl(dt, un_acc));
                        w_a0 = acc; w_g0 = gyr;
                    }
                } else if (t >= 128) {   // (idle otherwise during the two recursions: clear the chunk's F / V for preint_step_FV)
                    for (int q = t - 128; q < m * 225; q += nt - 128) (&pi_F[0][0])[q] = 0;
                    for (int q = t - 128; q < m * 270; q += nt - 128) (&pi_V[0][0])[q] = 0;
                }
                __syncthreads();
                PH(112);
                if (t < m) bf::preint_step_FV(pi_pre[t], lba, lbg, pi_dt[t], ld3(pi_acc[t]), ld3(pi_gyr[t]), pi_F[t], pi_V[t], false);
                __syncthreads();
                PH(113);
                for (int k = 0; k < m; k++) {
                    const double *Fk = pi_F[k], *Vk = pi_V[k];
                    if (t < 225) {
                        int i = t / 15, j = t - i * 15;
                        double s1 = 0, s2 = 0;
                        for (int u = 0; u < 15; u++) { s1 += Fk[i * 15 + u] * pw.J[u * 15 + j]; s2 += Fk[i * 15 + u] * pw.Pm[u * 15 + j]; }
                        pw.FJ[t] = s1; pw.FP[t] = s2;
                    }
                    __syncthreads();
                    if (t < 225) {
                        int i = t / 15, j = t - i * 15;
                        double s1 = 0;
                        for (int u = 0; u < 15; u++) s1 += pw.FP[i * 15 + u] * Fk[j * 15 + u];
                        const double nn[6] = {cfg.acc_n * cfg.acc_n, cfg.gyr_n * cfg.gyr_n, cfg.acc_n * cfg.acc_n, cfg.gyr_n * cfg.gyr_n, cfg.acc_w * cfg.acc_w, cfg.gyr_w * cfg.gyr_w};
                        double tt = 0;
#pragma unroll
                        for (int u = 0; u < 18; u++) tt += Vk[i * 18 + u] * nn[u / 3] * Vk[j * 18 + u];
                        pw.J[t] = pw.FJ[t];
                        pw.Pm[t] = s1 + tt;
                    }
                    __syncthreads();
                }
                PH(114);
            }
            if (t == 0) {
                st3(P.dp, s_dp); st3(P.dv, s_dv);
                P.dq[0] = s_dq.w; P.dq[1] = s_dq.x; P.dq[2] = s_dq.y; P.dq[3] = s_dq.z;
                P.sum_dt = s_sum;
                st3(P.acc0, s_a0); st3(P.gyr0, s_g0);
                const int nb = nb0 + n;
                if (nb > VIO_IMU_SLOT_CAP) be.overflow |= 2;
                P.n_buf = min(nb, VIO_IMU_SLOT_CAP);
            }
            if (t == 64) {
                stm(be.Rs[fc], w_R); st3(be.Ps[fc], w_P); st3(be.Vs[fc], w_V);
                st3(be.acc_0, w_a0); st3(be.gyr_0, w_g0);
            }
            __syncthreads();
        } else if (t == 0) {   // frame 0: no pre-integration yet, acc_0 / gyr_0 follow the samples
            const int idx = (head + n - 1) % C.NIMU;
            st3(be.acc_0, ld3(ia + (size_t)idx * 3)); st3(be.gyr_0, ld3(ig + (size_t)idx * 3));
        }
        if (fc != 0) preint_store(P, pw);
        if (t == 0) be.prevTime = curTime;
        __syncthreads();
    }
    PH(109);
    // ---- triangulateWithDepth (feature_manager.cpp:386-543); the dynamic initialisation triangulates after its SfM instead (estimator.cpp:921-933)
    if (!cfg.use_imu && be.solver_flag == 1) init_frame_pose_by_pnp(c, fc, c.res, sred);   // estimator.cpp:321-322 (VO mode, NON_LINEAR only)
    if (!(cfg.dynamic_init && be.solver_flag == 0)) triangulate_with_depth(c, nlm);
    __syncthreads();
    PH(110);
    if (t == 0) {
        be.processed = 1;
        be.frames_processed++;
        if (be.solver_flag == 0) {
            // static initialisation solves once the window is full; the dynamic one is decided by the host (vio_abi.hip run_dynamic_init)
            if (fc == W && !cfg.dynamic_init) { be.do_solve = 1; be.do_marg = 1; }
        } else { be.do_solve = 1; be.do_marg = 1; }
    }
}

// ====================================================================================================== be_solve
namespace {


// prior residual row i at parameters X: r0 + J dx; dx assembled in LDS sdx (n)
__device__ __forceinline__ void prior_dx(const Ctx &c, const Params &X, double *sdx, bool sync = true) {
    const int t = threadIdx.x, W = c.W;
    const BeSeq &be = *c.be;
    if (t <= W + 2) {
        if (t < W) {
            double d[6];
            bf::pose_dx(&X.pose[t * 7], &c.prior_x0[t * 7], d);
            for (int k = 0; k < 6; k++) sdx[6 * t + k] = be.prior_present[t] ? d[k] : 0.0;
        } else if (t == W) {
            for (int k = 0; k < 9; k++) sdx[6 * W + k] = be.prior_present[W] ? X.sb[k] - c.prior_x0[W * 7 + k] : 0.0;
        } else if (t == W + 1) {
            double d[6];
            bf::pose_dx(X.ex, &c.prior_x0[W * 7 + 9], d);
            for (int k = 0; k < 6; k++) sdx[6 * W + 9 + k] = be.prior_present[W + 1] ? d[k] : 0.0;
        } else
            sdx[6 * W + 15] = be.prior_present[W + 2] ? X.td - c.prior_x0[W * 7 + 16] : 0.0;
    }
    if (sync) __syncthreads();
}
// tangent index of prior slot a
__device__ __forceinline__ int prior_map(int a, int W) {
    if (a < 6 * W) return a;                                  // pose k at 6k
    if (a < 6 * W + 9) return 6 * (W + 1) + (a - 6 * W);     // speed-bias 0
    if (a < 6 * W + 15) return 15 * (W + 1) + (a - 6 * W - 9);
    return 15 * (W + 1) + 6;
}

// evaluate every residual at X. withJ: store weighted Jacobians/residuals for assembly. Returns total cost.
// vext: the extrinsic and / or td block is a variable of this solve.  Otherwise the residual records are written in the compact
// layout (28 doubles: two rows of [pose_i(6) pose_j(6) inv_depth r]) and no Jacobian is evaluated for the constant blocks.
__device__ __forceinline__ double evaluate(const Ctx &c, const Params &X, const double *feat, bool withJ, int nres, double *sred, double *sdx, double *srp,
                                           double *geo, bool vext) {
    const int t = threadIdx.x, nt = blockDim.x, W = c.W, n = c.NPR;
    const BeSeq &be = *c.be;
    const vio_config &cfg = c.C->c;
    double cost = 0;
    const int s = c.s;
    struct { float *timings; } B = {c.timings};
    PH_INIT;
    // Stage 1 (independent small jobs, one barrier for all of them): prior tangent dx (threads 0 .. W + 2), IMU factors (upper
    // wavefronts), frame-pair geometry (threads 0 .. W1^2).  Stage 2: prior mat-vec, then the projection residuals.
    if (be.has_prior) prior_dx(c, X, sdx, false);
    PH(40);
    // IMU factors: five threads per factor (whitened residual + four Jacobian column groups). Spread over the upper lanes
    // of the block so that they do not serialise with the projection residuals handled by the low thread ids.
    v3 G = ld3(be.g);
    auto imu_item = [&](int i, int part) {
        const int j = i + 1;
        const PreInt &p = c.pre[be.pre_idx[j]];
        double *out = c.imu_raw + (size_t)i * 15 * 31;
        if (!cfg.use_imu || p.sum_dt > 10.0) { if (part == 0) for (int k = 0; k < 15; k++) out[k * 31 + 30] = 0; return; }
        if (part == 0) {
            double raw[15];
            bf::imu_raw_residual(p, G, &X.pose[i * 7], &X.sb[i * 9], &X.pose[j * 7], &X.sb[j * 9], raw);
            for (int r = 0; r < 15; r++) {
                double sacc = 0;
                for (int k = 0; k <= r; k++) sacc += p.sqrt_info[r * 15 + k] * raw[k];
                out[r * 31 + 30] = sacc;
                cost += 0.5 * sacc * sacc;
            }
        } else if (withJ)
            bf::imu_raw_jacobian_part(p, G, &X.pose[i * 7], &X.sb[i * 9], &X.pose[j * 7], &X.sb[j * 9], part - 1, out, 31);  // raw, whitened in assemble
    };
    {
        // Five work types per factor (whitened residual + four Jacobian column groups), each a different code path.  With at least
        // five wavefronts every type gets its own wavefront (the top five, factor i on lane 63 - i), so that the types run side by
        // side instead of one wavefront executing its divergent branches one after the other; the upper lanes are used because the
        // low thread ids carry the first projection residuals.
        const int nwv = nt >> 6, part = nwv - 1 - (t >> 6), i = 63 - (t & 63);
        if (nwv >= 5 && W <= 64) {
            if (part < 5 && i < W) imu_item(i, part);
        } else {
            for (int w = (nt - 1 - t); w < W * 5; w += nt) imu_item(w / 5, w % 5);
        }
    }
    PH(46);
    // projection factors, CauchyLoss(1.0): frame-pair geometry first (one thread per pair with residuals), then one thread per
    // residual with a handful of 3-vector products (be_factors.h eval_projection_pair).  The pair geometry lives in LDS (`geo`,
    // (W1^2 + 1) x 32 doubles of the work region, free whenever evaluate runs): every residual gathers 30 doubles of it, and as
    // per-lane gathers from HBM those were a third of the cache-line lookups that bound this loop.
    {
        const int W1 = W + 1;
        for (int p = t; p <= W1 * W1; p += nt) {
            if (p == W1 * W1) {
                m3 ric = q2R(mkq(X.ex[6], X.ex[3], X.ex[4], X.ex[5]));
                stm(geo + (size_t)p * 32, ric);
                continue;
            }
            int i = p / W1, j = p - i * W1;
            if (!(i < j) || c.pair_start[p + 1] == c.pair_start[p]) continue;
            bf::PairGeo g;
            bf::pair_geo(&X.pose[i * 7], &X.pose[j * 7], X.ex, g);
            double *o = geo + (size_t)p * 32;
            for (int q = 0; q < 9; q++) { o[q] = g.A1[q]; o[9 + q] = g.A2[q]; o[18 + q] = g.M[q]; }
            o[27] = g.t[0]; o[28] = g.t[1]; o[29] = g.t[2];
        }
        __syncthreads();
        PH(41);
        if (be.has_prior) {
            // The prior is kept as the quadratic form (A, b, c0) = (J^T J, J^T r, |r|^2) of the reference's linearised factor:
            // 1/2 |r + J dx|^2 = 1/2 c0 + dx^T b + 1/2 dx^T A dx.  srp receives the gradient q = b + A dx (what assemble / marg add to g).
            matvec_pass(c.prior_H, n, n, n, nullptr, sdx, nullptr, srp, nullptr);  // A dx: one wavefront per row
            for (int i = t; i < n; i += nt) {
                const double b0 = c.prior_r[i], q = b0 + srp[i];
                srp[i] = q;
                cost += 0.5 * sdx[i] * (b0 + q);
            }
            if (t == 0) cost += 0.5 * be.prior_c0;
        }
        const double *ricm = geo + (size_t)W1 * W1 * 32;
#pragma unroll 2
        for (int r = t; r < nres; r += nt) {
            int slot = c.res_lm[r], k = c.res_k[r];
            int imu_i = c.lm_start[slot], imu_j = imu_i + k;
            const bf::PairGeo &g = *(const bf::PairGeo *)(geo + (size_t)(imu_i * W1 + imu_j) * 32);
            double rr[2], wgt = 1.0;
            double sq;
            if (vext) {
                double *out = c.res + (size_t)r * 42;
                bf::eval_projection_pair(cfg, g, ricm, X.ex, feat[c.lm_pidx[slot]], X.td, obs_ptr(c, slot, imu_i), obs_ptr(c, slot, imu_j),
                                         cfg.estimate_td != 0, rr, withJ ? out : nullptr, true, &wgt);
                sq = rr[0] * rr[0] + rr[1] * rr[1];
                if (withJ) { out[40] = wgt * rr[0]; out[41] = wgt * rr[1]; }
            } else {
                double *out = c.res + (size_t)r * 28;
                bf::eval_projection_pair(cfg, g, ricm, X.ex, feat[c.lm_pidx[slot]], X.td, obs_ptr(c, slot, imu_i), obs_ptr(c, slot, imu_j),
                                         cfg.estimate_td != 0, rr, withJ ? out : nullptr, true, &wgt, 14, false);
                sq = rr[0] * rr[0] + rr[1] * rr[1];
                if (withJ) { out[13] = wgt * rr[0]; out[27] = wgt * rr[1]; }
            }
            cost += 0.5 * log(1.0 + sq);
        }
    }
    PH(44);
    cost = block_sum(cost, sred);
    PH(45);
    return cost;
}

// pair-block entry index of the packed symmetric 20x20
__device__ __forceinline__ int sym_idx(int a, int b) {
    if (a > b) { int x = a; a = b; b = x; }
    return a * 20 - a * (a - 1) / 2 + (b - a);
}
__device__ __forceinline__ int local_of(int a, int i, int j, int W) {
    // tangent index a within the vision set -> local column of pair (i,j) or -1
    int np = 6 * (W + 1);
    if (a < np) {
        int f = a / 6, d = a - f * 6;
        if (f == i) return d;
        if (f == j) return 6 + d;
        return -1;
    }
    int e = a - 15 * (W + 1);
    return 12 + e;  // ex 0..5 -> 12..17, td (6) -> 18
}

#define PB_U 8  // k groups (of 4 Jacobian rows) loaded per batch in the frame-pair block phase of assemble
// compact index of the frame pair (i < j)
__device__ __forceinline__ int pair_slot(int i, int j, int W1) { return i * W1 - i * (i + 1) / 2 + (j - i - 1); }

// One half of a landmark's coupling row (see assemble): `res` = the landmark's residual Jacobians (42 doubles each, observation k
// at res + 42 (k - 1)), `row` = its Hpl row.  The arrays never overlap; the restrict qualifiers let the loads of the next
// residual be issued ahead of the row stores of the current one (the loop is bound by the latency of those loads).
__device__ __forceinline__ void lm_row(const double *__restrict__ res, double *__restrict__ row, double *__restrict__ Hll,
                                       double *__restrict__ gl, int half, int st, int kend, int ext_off, int krelo = -1) {
    // krelo: index of the landmark's relocalisation record (rides behind the regular ones, be_phased.h) or -1.  Its pose_j group is
    // zero by construction and it has no frame st + k of its own: no pose_j store for it (which would land on whatever columns follow)
    // residual records are 336 bytes apart and 16-byte aligned: 16-byte loads halve the number of cache-line lookups of this gather
    if (half == 0) {
        double si[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 2
        for (int k = 1; k < kend; k++) {
            const double *Jr = res + (size_t)(k - 1) * 42;
            const double2 *J2 = (const double2 *)Jr;
            const double l0 = Jr[19], l1 = Jr[39];
            double a[6], b[6], e[6], f[6];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                double2 va = J2[q], vb = J2[3 + q], ve = J2[10 + q], vf = J2[13 + q];
                a[2 * q] = va.x; a[2 * q + 1] = va.y; b[2 * q] = vb.x; b[2 * q + 1] = vb.y;
                e[2 * q] = ve.x; e[2 * q + 1] = ve.y; f[2 * q] = vf.x; f[2 * q + 1] = vf.y;
            }
#pragma unroll
            for (int d = 0; d < 6; d++) {
                si[d] += a[d] * l0 + e[d] * l1;
                if (k != krelo) row[6 * (st + k) + d] = b[d] * l0 + f[d] * l1;
            }
        }
#pragma unroll
        for (int d = 0; d < 6; d++) row[6 * st + d] = si[d];
    } else {
        double se[7] = {0, 0, 0, 0, 0, 0, 0}, hll = 0, gg = 0;
#pragma unroll 2
        for (int k = 1; k < kend; k++) {
            const double *Jr = res + (size_t)(k - 1) * 42;
            const double2 *J2 = (const double2 *)Jr;
            double a[8], e[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                double2 va = J2[6 + q], ve = J2[16 + q];   // Jr[12..19], Jr[32..39]
                a[2 * q] = va.x; a[2 * q + 1] = va.y; e[2 * q] = ve.x; e[2 * q + 1] = ve.y;
            }
            const double2 rr = J2[20];                      // weighted residual (Jr[40], Jr[41])
            const double l0 = a[7], l1 = e[7];
#pragma unroll
            for (int d = 0; d < 7; d++) se[d] += a[d] * l0 + e[d] * l1;
            hll += l0 * l0 + l1 * l1;
            gg += l0 * rr.x + l1 * rr.y;
        }
#pragma unroll
        for (int d = 0; d < 7; d++) row[ext_off + d] = se[d];
        *Hll = hll;
        *gl = gg;
    }
}

// IMU factor block on the FP64 matrix cores (one wavefront): raw_l = [J_raw | r_whitened] (15 x 31, LDS), M_l = chol(cov)^-1 (15 x 15
// lower triangular, LDS).  The rows of J are whitened on the fly (x = M J_raw, one output row per lane group) and the 31 x 31 Gram
// matrix [Jw r]^T [Jw r] is accumulated with K = 16 (row 15 is padding): a00 = rows / cols 0-15, a10 = rows 16-31 x cols 0-15,
// a11 = rows / cols 16-31, in the C/D layout of v_mfma_f64_16x16x4_f64 (element r of lane (lk, li) = row lk + 4 r, column li).
__device__ __forceinline__ void imu_block_mfma(const double *raw_l, const double *M_l, int li, int lk, v4f64 &a00, v4f64 &a10, v4f64 &a11) {
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const int kk = 4 * ks + lk;
        double x0 = 0, x1 = 0;
        if (kk < 15) {
            const int c1 = 16 + li;
            for (int k = 0; k <= kk; k++) {
                double m = M_l[kk * 15 + k];
                x0 += m * raw_l[k * 31 + li];
                if (c1 < 30) x1 += m * raw_l[k * 31 + c1];
            }
            if (c1 == 30) x1 = raw_l[kk * 31 + 30];
        }
        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, a00, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x0, a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, a11, 0, 0, 0);
    }
}

// The same for the compact residual records (28 doubles, no extrinsic / td columns): half 0 = pose columns, half 1 = Hll and gl.
__device__ __forceinline__ void lm_row_compact(const double *__restrict__ res, double *__restrict__ row, double *__restrict__ Hll,
                                               double *__restrict__ gl, int half, int st, int kend) {
    if (half == 0) {
        double si[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 2
        for (int k = 1; k < kend; k++) {
            const double2 *J2 = (const double2 *)(res + (size_t)(k - 1) * 28);
            double a[6], b[6], e[6], f[6];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                double2 va = J2[q], vb = J2[3 + q], ve = J2[7 + q], vf = J2[10 + q];
                a[2 * q] = va.x; a[2 * q + 1] = va.y; b[2 * q] = vb.x; b[2 * q + 1] = vb.y;
                e[2 * q] = ve.x; e[2 * q + 1] = ve.y; f[2 * q] = vf.x; f[2 * q + 1] = vf.y;
            }
            const double l0 = J2[6].x, l1 = J2[13].x;
#pragma unroll
            for (int d = 0; d < 6; d++) {
                si[d] += a[d] * l0 + e[d] * l1;
                row[6 * (st + k) + d] = b[d] * l0 + f[d] * l1;
            }
        }
#pragma unroll
        for (int d = 0; d < 6; d++) row[6 * st + d] = si[d];
    } else {
        double hll = 0, gg = 0;
#pragma unroll 2
        for (int k = 1; k < kend; k++) {
            const double2 *J2 = (const double2 *)(res + (size_t)(k - 1) * 28);
            const double2 p0 = J2[6], p1 = J2[13];   // (inv_depth column, weighted residual) of the two rows
            hll += p0.x * p0.x + p1.x * p1.x;
            gg += p0.x * p0.y + p1.x * p1.y;
        }
        *Hll = hll;
        *gl = gg;
    }
}

// assemble H (P x P, ld LW), g (vec slot 0), Hpl / Hll / gl from the stored residual Jacobians.
// work: LDS scratch (>= max(W*450, npairs*210) doubles when it fits, see be_solve); pb = frame-pair blocks (LDS or HBM)
__device__ __forceinline__ void assemble(const Batch &B, const Ctx &c, const Params &X, int nres, int Fa, const int *alist, double *srp, double *work,
                         double *pb, bool hpl_sparse, bool vext) {
    const int s = c.s;
    PH_INIT;
    const int t = threadIdx.x, nt = blockDim.x, W = c.W, P = c.P, LW = c.LW, n = c.NPR;
    const int lane = t & 63, wave = t >> 6, nw = nt >> 6;
    const BeSeq &be = *c.be;
    double *H = c.H, *g = c.vec;
    for (int i = t; i < P * LW; i += nt) H[i] = 0;
    for (int i = t; i < LW; i += nt) g[i] = 0;
    {
        // landmark rows: with the column-aware Schur staging and mat-vecs (hpl_sparse) only the column tiles that hold pose or
        // extrinsic / td columns are ever read, the speed-bias columns in between need no zeros (the marginalisation kernel uses
        // this buffer as scratch, so they do hold garbage); the dense fall-back paths read whole rows
        // (with constant extrinsic / td blocks and the column-aware paths the extrinsic tile is never read either)
        const int w0 = hpl_sparse ? min(LW, (6 * (W + 1) + 15) & ~15) : LW, e_lo = max(w0, (15 * (W + 1)) & ~15),
                  wz = w0 + ((hpl_sparse && !vext) ? 0 : (LW - e_lo));
        for (int i = t; i < ((Fa + 3) & ~3) * wz; i += nt) {
            const int row = i / wz, cc = i - row * wz;
            c.Hpl[(size_t)row * LW + (cc < w0 ? cc : e_lo + (cc - w0))] = 0;
        }
    }
    __syncthreads();
    PH(32);
    // prior: H += J^T J (precomputed), g += J^T r
    if (be.has_prior) {
        for (int w = t; w < n * n; w += nt) {
            int a = w / n, b = w - a * n;
            H[prior_map(a, W) * LW + prior_map(b, W)] = c.prior_H[w];
        }
        for (int a = t; a < n; a += nt) g[prior_map(a, W)] = srp[a];  // prior gradient b + A dx (computed by evaluate)
    }
    __syncthreads();
    PH(33);
    // IMU: one wavefront per factor.  [J r] (15 x 31, r already whitened by evaluate) is whitened with M = chol(cov)^-1 from a
    // per-wave LDS slice and squared on the FP64 matrix cores (K = 16, three 16x16 accumulators); even factors first, then odd
    // ones (neighbouring factors share the pose / speed-bias block of the frame between them).
    {
        const int li = lane & 15, lk = lane >> 4;
        const int nconc = max(1, min(nw, (W * 450) / 704));
        for (int parity = 0; parity < 2; parity++) {
            for (int base = 0; 2 * base + parity < W; base += nconc) {
                const int i = 2 * (base + wave) + parity;
                bool act = wave < nconc && i < W;
                const PreInt *pp = act ? &c.pre[be.pre_idx[i + 1]] : nullptr;
                if (act && (!c.C->c.use_imu || pp->sum_dt > 10.0)) act = false;
                double *raw_l = work + (wave < nconc ? wave : 0) * 704, *M_l = raw_l + 472;
                if (act) {
                    const double *raw = c.imu_raw + (size_t)i * 15 * 31;
                    for (int q = lane; q < 465; q += 64) raw_l[q] = raw[q];
                    for (int q = lane; q < 225; q += 64) M_l[q] = pp->sqrt_info[q];
                }
                __syncthreads();
                if (act) {
                    v4f64 a00 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
                    imu_block_mfma(raw_l, M_l, li, lk, a00, a10, a11);
                    auto gidx = [&](int a) {
                        return a < 6 ? 6 * i + a : (a < 15 ? 6 * (W + 1) + 9 * i + (a - 6) : (a < 21 ? 6 * (i + 1) + (a - 15) : 6 * (W + 1) + 9 * (i + 1) + (a - 21)));
                    };
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = lk + 4 * r, col = li;  // C/D layout: acc[r] = element (row, col)
                        const int ic = gidx(col);
                        H[gidx(row) * LW + ic] += a00[r];
                        const int r1 = 16 + row, c1 = 16 + col;
                        if (r1 < 30) {
                            const int ir = gidx(r1);
                            H[ir * LW + ic] += a10[r];
                            H[ic * LW + ir] += a10[r];
                            if (c1 < 30) H[ir * LW + gidx(c1)] += a11[r];
                        } else if (r1 == 30) {
                            g[ic] += a10[r];
                            if (c1 < 30) g[gidx(c1)] += a11[r];
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    PH(34);
    // vision: frame-pair blocks G_p = [J19 r]^T [J19 r] (packed symmetric 20x20) on the FP64 matrix cores:
    // one wavefront per frame pair, K = 2 rows per residual, three 16x16 accumulators (cols 0-15 / 16-19 of the 20 columns)
    const int W1 = W + 1;
    {
        const int li = lane & 15, lk = lane >> 4;
        for (int p = wave; p < W1 * W1; p += nw) {
            int i = p / W1, j = p - i * W1;
            if (!(i < j)) continue;
            const int q0 = c.pair_start[p], np_ = c.pair_start[p + 1] - q0;
            double *out = pb + (size_t)pair_slot(i, j, W1) * 210;
            if (np_ == 0) { for (int e = lane; e < 210; e += 64) out[e] = 0; continue; }
            v4f64 a00 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
            const int K = 2 * np_;
            // residual indices of the pair are fetched 64 at a time (one coalesced load) and broadcast with shuffles; the
            // k loop is unrolled by PB_U with unconditional (clamped) loads so that 2 * PB_U row loads are in flight per MFMA
            // batch: the phase is bound by the L2 latency of those loads (a typical pair is one or two batches), not by the MFMAs
            if (vext) {
                for (int base = 0; base < np_; base += 64) {
                    const int nchunk = min(64, np_ - base);
                    const int myidx = c.pair_list[q0 + base + min(lane, nchunk - 1)];
                    const int Kc = 2 * nchunk;
                    for (int k0 = 0; k0 < Kc; k0 += 4 * PB_U) {
                        double x0[PB_U], x1[PB_U];
#pragma unroll
                        for (int u = 0; u < PB_U; u++) {
                            int kk = k0 + 4 * u + lk;
                            bool valid = kk < Kc;
                            int ridx = __shfl(myidx, min(kk, Kc - 1) >> 1, 64);
                            const double *Jr = c.res + (size_t)ridx * 42;
                            int sub = kk & 1, ro = sub * 20;
                            double v0 = Jr[ro + li];
                            double v1 = Jr[li < 3 ? ro + 16 + li : 40 + sub];
                            x0[u] = valid ? v0 : 0.0;
                            x1[u] = (valid && li < 4) ? v1 : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < PB_U; u++) {
                            if (k0 + 4 * u >= Kc) break;  // wavefront-uniform: the remaining k groups of this batch are all padding
                            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u], x0[u], a00, 0, 0, 0);
                            a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u], x0[u], a10, 0, 0, 0);
                            a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u], x1[u], a11, 0, 0, 0);
                        }
                    }
                }
                for (int r = 0; r < 4; r++) {
                    int row = lk + 4 * r, col = li;  // C/D layout of v_mfma_f64_16x16x4_f64
                    if (col <= row) out[sym_idx(col, row)] = a00[r];
                    if (row < 4) out[sym_idx(col, 16 + row)] = a10[r];
                    if (row < 4 && col < 4 && col <= row) out[sym_idx(16 + col, 16 + row)] = a11[r];
                }
            } else {
                // compact records: 12 pose columns + the weighted residual in one 16-wide operand (columns 13 .. 15 are padding), one
                // accumulator.  Every output element is the same dot product, in the same k order, as in the three-accumulator form.
                for (int base = 0; base < np_; base += 64) {
                    const int nchunk = min(64, np_ - base);
                    const int myidx = c.pair_list[q0 + base + min(lane, nchunk - 1)];
                    const int Kc = 2 * nchunk;
                    for (int k0 = 0; k0 < Kc; k0 += 4 * PB_U) {
                        double x0[PB_U];
#pragma unroll
                        for (int u = 0; u < PB_U; u++) {
                            int kk = k0 + 4 * u + lk;
                            bool valid = kk < Kc;
                            int ridx = __shfl(myidx, min(kk, Kc - 1) >> 1, 64);
                            const double *Jr = c.res + (size_t)ridx * 28 + (kk & 1) * 14;
                            double v0 = Jr[li < 12 ? li : 13];
                            x0[u] = (valid && li < 13) ? v0 : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < PB_U; u++) {
                            if (k0 + 4 * u >= Kc) break;
                            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u], x0[u], a00, 0, 0, 0);
                        }
                    }
                }
                for (int r = 0; r < 4; r++) {
                    int row = lk + 4 * r, col = li;
                    if (row < 12) { if (col <= row) out[sym_idx(col, row)] = a00[r]; }
                    else if (row == 12 && col <= 12) out[sym_idx(col < 12 ? col : 19, 19)] = a00[r];   // (column, residual) and (residual, residual)
                }
            }
        }
    }
    __syncthreads();
    PH(35);
    {
        // one item per unordered pair (ra <= rb) of vision rows plus one per gradient entry: H is symmetric and both mirror
        // elements receive the same sum (identical terms in identical order), so only half of the element sums are formed
        const int nv = 6 * W1 + (vext ? 7 : 0), ntri = nv * (nv + 1) / 2;   // constant extrinsic / td blocks get no entries at all
        for (int w = t; w < ntri + nv; w += nt) {
            int ra, rb;
            if (w < ntri) {
                int r = (int)((sqrtf(8.0f * (float)w + 1.0f) - 1.0f) * 0.5f);
                while (r * (r + 1) / 2 > w) r--;
                while ((r + 1) * (r + 2) / 2 <= w) r++;
                rb = r; ra = w - r * (r + 1) / 2;
            } else { ra = w - ntri; rb = nv; }
            int a = ra < 6 * W1 ? ra : 15 * W1 + (ra - 6 * W1);
            int b = rb < nv ? (rb < 6 * W1 ? rb : 15 * W1 + (rb - 6 * W1)) : -1;
            double sacc = 0;
            const int fa = ra < 6 * W1 ? ra / 6 : -1, fb = (rb < 6 * W1) ? rb / 6 : -1;
            if (fa >= 0 && fb >= 0 && fa != fb) {
                int i = min(fa, fb), j = max(fa, fb);
                sacc = pb[(size_t)pair_slot(i, j, W1) * 210 + sym_idx(local_of(a, i, j, W), local_of(b, i, j, W))];
            } else if (fa >= 0 || fb >= 0) {
                // diagonal pose block, or pose x (extrinsic / td / gradient): only the W pairs that contain that frame
                const int f = fa >= 0 ? fa : fb;
                for (int o = 0; o < W1; o++) {
                    if (o == f) continue;
                    int i = min(f, o), j = max(f, o);  // empty pairs hold a zero block (written above): no test, no HBM load here
                    int la = local_of(a, i, j, W);
                    int lb = b >= 0 ? local_of(b, i, j, W) : 19;
                    sacc += pb[(size_t)pair_slot(i, j, W1) * 210 + sym_idx(la, lb)];
                }
            } else {
                for (int i = 0; i < W1; i++)
                    for (int j = i + 1; j < W1; j++) {
                        int la = local_of(a, i, j, W);
                        int lb = b >= 0 ? local_of(b, i, j, W) : 19;
                        sacc += pb[(size_t)pair_slot(i, j, W1) * 210 + sym_idx(la, lb)];
                    }
            }
            if (b >= 0) {
                H[a * LW + b] += sacc;
                if (a != b) H[b * LW + a] += sacc;
            } else
                g[a] += sacc;
        }
    }
    PH(36);
    // landmark coupling rows (dense, zero padded above), Hll, gl: two threads per variable landmark walk its residuals; the even one
    // owns the pose columns (start frame + observing frames), the odd one the extrinsic / td columns, Hll and gl
    for (int w = t; w < 2 * Fa; w += nt) {
        const int ka = w >> 1, half = w & 1;
        const int slot = alist[ka];
        const int st = c.lm_start[slot], r0 = c.lm_tmp[slot];
        const int kend = min(c.lm_nobs[slot], nres - r0 + 1);  // residual r0 + k - 1 of observation k, capped by the residual list
        if (vext) lm_row(c.res + (size_t)r0 * 42, c.Hpl + (size_t)ka * LW, c.Hll + ka, c.gl + ka, half, st, kend, 15 * W1);
        else lm_row_compact(c.res + (size_t)r0 * 28, c.Hpl + (size_t)ka * LW, c.Hll + ka, c.gl + ka, half, st, kend);
    }
    __syncthreads();
    PH(37);
}

}  // namespace

namespace {
// Everything optimization() does before the first evaluation (estimator.cpp:1161-1212, 936-981): static-initialisation extras,
// vector2double into X, landmark / residual / frame-pair indexing, constness decisions (sh_i[0] = extrinsic variable, sh_i[1] = td
// variable).  Shared by the persistent solve kernel and the phased solver (ps_setup_kernel).
// allow_relo: the caller can carry relocalisation factors (phased solver): sh_i[4] returns whether this solve has them
__device__ __forceinline__ void solve_prologue(const Batch &B, Ctx &c, Params &X, int *scratch, PreWork &pw, int *sh_i, int &F, int &Fa, int &nres,
                                               const bool allow_relo = false, unsigned *lmkey = nullptr, const int lmkey_cap = 0) {
    const int t = threadIdx.x, nt = blockDim.x;
    const vio_config &cfg = c.C->c;
    BeSeq &be = *c.be;
    const int W = c.W, W1 = W + 1;
    const int s = c.s;
    PH_INIT;
    // ---- static initialisation extras (estimator.cpp:266-283): solveGyroscopeBias + repropagate (IMU mode only, :264)
    if (be.solver_flag == 0 && cfg.use_imu) {
        if (t == 0) {
            double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
            for (int i = 0; i < W; i++) {
                int j = i + 1;
                const PreInt &p = c.pre[be.pre_idx[j]];
                quat q_ij = R2q(mul(tr(ldm(be.Rs[i])), ldm(be.Rs[j])));
                m3 tA = bf::get33(p.jac, 15, bf::O_R, bf::O_BG);
                v3 tb = scl(2.0, qvec(qmul(qinv(mkq(p.dq[0], p.dq[1], p.dq[2], p.dq[3])), q_ij)));
                m3 AtA = mul(tr(tA), tA);
                v3 Atb = mul(tr(tA), tb);
                for (int r = 0; r < 9; r++) A[r] += AtA.a[r];
                b[0] += Atb.x; b[1] += Atb.y; b[2] += Atb.z;
            }
            double M[12];
            for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) M[r * 4 + q] = A[r * 3 + q]; M[r * 4 + 3] = b[r]; }
            for (int i = 0; i < 3; i++) {
                int p = i;
                for (int r = i + 1; r < 3; r++) if (fabs(M[r * 4 + i]) > fabs(M[p * 4 + i])) p = r;
                for (int q = 0; q < 4; q++) { double tmp = M[i * 4 + q]; M[i * 4 + q] = M[p * 4 + q]; M[p * 4 + q] = tmp; }
                if (M[i * 4 + i] == 0) continue;
                for (int r = i + 1; r < 3; r++) {
                    double f = M[r * 4 + i] / M[i * 4 + i];
                    for (int q = i; q < 4; q++) M[r * 4 + q] -= f * M[i * 4 + q];
                }
            }
            double x[3] = {0, 0, 0};
            for (int i = 2; i >= 0; i--) {
                double sacc = M[i * 4 + 3];
                for (int q = i + 1; q < 3; q++) sacc -= M[i * 4 + q] * x[q];
                x[i] = M[i * 4 + i] != 0 ? sacc / M[i * 4 + i] : 0;
            }
            for (int i = 0; i <= W; i++) { be.Bgs[i][0] += x[0]; be.Bgs[i][1] += x[1]; be.Bgs[i][2] += x[2]; }
        }
        __syncthreads();
        repropagate_window(c, pw);
    }

    // ---- vector2double (estimator.cpp:936-981)
    if (t <= W) {
        int i = t;
        X.pose[i * 7 + 0] = be.Ps[i][0]; X.pose[i * 7 + 1] = be.Ps[i][1]; X.pose[i * 7 + 2] = be.Ps[i][2];
        quat q = R2q(ldm(be.Rs[i]));
        X.pose[i * 7 + 3] = q.x; X.pose[i * 7 + 4] = q.y; X.pose[i * 7 + 5] = q.z; X.pose[i * 7 + 6] = q.w;
        for (int k = 0; k < 3; k++) { X.sb[i * 9 + k] = be.Vs[i][k]; X.sb[i * 9 + 3 + k] = be.Bas[i][k]; X.sb[i * 9 + 6 + k] = be.Bgs[i][k]; }
    }
    if (t == W + 1) {
        X.ex[0] = be.tic[0]; X.ex[1] = be.tic[1]; X.ex[2] = be.tic[2];
        quat q = R2q(ldm(be.ric));
        X.ex[3] = q.x; X.ex[4] = q.y; X.ex[5] = q.z; X.ex[6] = q.w;
        X.td = be.td;
        for (int k = 0; k < 7; k++) X.relo[k] = be.relo_Pose[k];
    }
    // constness (estimator.cpp:1187-1212), decided first because the relocalisation factors borrow the columns of a CONSTANT extrinsic
    if (t == 0) {
        double v0 = nrm(ld3(be.Vs[0]));
        int ex_active;
        if ((cfg.estimate_extrinsic && be.frame_count == W && v0 > 0.2) || be.openExEstimation) { be.openExEstimation = 1; ex_active = 1; }
        else ex_active = 0;
        int td_active = cfg.use_imu && cfg.estimate_td && !(v0 < 0.2);   // no td block without the IMU (estimator.cpp:1204)
        sh_i[0] = ex_active; sh_i[1] = td_active;
        int relo_on = be.relo_info && be.solver_flag == 1;
        if (relo_on && !allow_relo) {
            // relocalisation on the persistent solver (windows beyond the phased solver's range) is not supported: the request is dropped and
            // the frame flagged (overflow bit 64); optimization() without relocalization_info is what runs
            relo_on = 0; be.relo_info = 0; be.overflow |= 64;
        } else if (relo_on && ex_active) {
            // DEVIATION 15: relo_Pose borrows the six tangent columns of the extrinsic in the reduced system, so the extrinsic is held constant
            // in the (one) solve that carries relocalisation factors even when ESTIMATE_EXTRINSIC has opened it; openExEstimation stays
            // latched and the next solve refines it again.  The oracle mirrors this under reference_quirks bit 2 (tests).
            ex_active = 0; sh_i[0] = 0;
        }
        sh_i[4] = relo_on; sh_i[3] = 0;
    }
    __syncthreads();
    const int relo_on = sh_i[4];
    PH(100);
    // ---- landmark indexing: in-problem (para_Feature index), variable landmarks, residual list
    const int nlm = be.n_lm;
    int *tmpA = c.pair_list, *tmpB = c.pair_list + c.NL;       // scan temporaries (pair_list proper is built afterwards)
    int *alist = c.pair_list + c.nres_cap - c.NL;              // variable-landmark slots, kept for the whole solve
    int *plist = c.pair_list + c.nres_cap - 2 * c.NL;          // in-problem landmark slots in list order
    __syncthreads();
    for (int k = t; k < nlm; k += nt) tmpA[k] = in_problem(c, c.lm_order[k]) ? 1 : 0;
    __syncthreads();
    F = block_scan_flags(tmpA, nlm, tmpB, scratch);
    for (int k = t; k < nlm; k += nt) {
        int slot = c.lm_order[k];
        c.lm_pidx[slot] = tmpA[k] ? tmpB[k] : -1;
        if (tmpA[k]) { c.feat[tmpB[k]] = 1.0 / c.lm_depth[slot]; plist[tmpB[k]] = slot; }
    }
    __syncthreads();
    // variable landmarks (not SetParameterBlockConstant): estimator.cpp:1278-1298
    for (int k = t; k < nlm; k += nt) {
        int slot = c.lm_order[k];
        tmpA[k] = (c.lm_pidx[slot] >= 0 && !(c.lm_est[slot] == 1 && cfg.fix_depth)) ? 1 : 0;
    }
    __syncthreads();
    Fa = block_scan_flags(tmpA, nlm, tmpB, scratch);
    for (int k = t; k < nlm; k += nt) {
        int slot = c.lm_order[k];
        c.lm_aidx[slot] = tmpA[k] ? tmpB[k] : -1;
        if (tmpA[k]) alist[tmpB[k]] = slot;
    }
    __syncthreads();
    PH(101);
    // relocalisation factors (estimator.cpp:1307-1346): in-problem landmarks with start_frame <= relo_frame_local_index whose id is among
    // the match points (upstream walks both ascending lists with one cursor: the same set); their factor rides as one more residual
    // record behind the landmark's regular ones, marked res_k = 0
    for (int k = t; k < nlm; k += nt) {
        const int slot = c.lm_order[k];
        int hit = 0;
        if (relo_on && c.lm_pidx[slot] >= 0 && c.lm_start[slot] <= be.relo_local) {
            const int id = c.lm_id[slot];
            int lo = 0, hi = be.relo_nmatch - 1;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1, mi = (int)c.relo_mp[3 * mid + 2];
                if (mi == id) { hit = 1; c.relo_xy[2 * slot] = c.relo_mp[3 * mid]; c.relo_xy[2 * slot + 1] = c.relo_mp[3 * mid + 1]; break; }
                if (mi < id) lo = mid + 1; else hi = mid - 1;
            }
        }
        c.lm_relo[slot] = hit;
        if (hit) atomicAdd(&sh_i[3], 1);
    }
    // residual list: (nobs-1) residuals per in-problem landmark, list order (estimator.cpp:1243-1302)
    for (int k = t; k < nlm; k += nt) { int slot = c.lm_order[k]; tmpA[k] = c.lm_pidx[slot] >= 0 ? c.lm_nobs[slot] - 1 + c.lm_relo[slot] : 0; }
    __syncthreads();
    nres = block_scan_flags(tmpA, nlm, tmpB, scratch);
    const int nres_max = c.nres_cap - 2 * c.NL;
    if (nres > nres_max) { nres = nres_max; if (t == 0) be.overflow |= 8; }
    for (int k = t; k < nlm; k += nt) {
        int slot = c.lm_order[k];
        int r0 = tmpB[k], cnt = tmpA[k];
        c.lm_tmp[slot] = r0;  // first residual index of this landmark
        const int nreg = c.lm_nobs[slot] - 1;
        for (int q = 0; q < cnt; q++)
            if (r0 + q < nres) { c.res_lm[r0 + q] = slot; c.res_k[r0 + q] = q < nreg ? q + 1 : 0; }
    }
    __syncthreads();
    PH(102);
    // frame-pair lists in deterministic (landmark list) order: one wavefront per frame pair walks the in-problem landmarks
    {
        const int lane = t & 63, wave = t >> 6, nw = nt >> 6;
        for (int p = t; p <= W1 * W1; p += nt) c.pair_start[p] = 0;
        // (round 6) what the walks below need of a landmark -- start frame, observation count, relocalisation mark, first residual index -- packed
        // into one word of LDS per in-problem landmark: every frame pair walks the whole list twice, W1 W / 2 pairs x F / 64 trips x 2, and from HBM
        // each trip was a chain of two dependent loads (the slot, then its fields): 214 us of ps_setup at W = 20, 55 at W = 10
        const bool keys = lmkey != nullptr && F <= lmkey_cap && c.nres_cap < (1 << 19);
        if (keys)
            for (int k = t; k < F; k += nt) {
                const int slot = plist[k];
                lmkey[k] = (unsigned)c.lm_start[slot] | ((unsigned)c.lm_nobs[slot] << 6) | ((unsigned)(c.lm_relo[slot] ? 1 : 0) << 12) | ((unsigned)c.lm_tmp[slot] << 13);
            }
        __syncthreads();
        auto lm_fields = [&](int k, int &st_, int &no_, int &rl_, int &r0_) {
            if (keys) { const unsigned key = lmkey[k]; st_ = key & 63u; no_ = (key >> 6) & 63u; rl_ = (key >> 12) & 1u; r0_ = (int)(key >> 13); }
            else { const int slot = plist[k]; st_ = c.lm_start[slot]; no_ = c.lm_nobs[slot]; rl_ = c.lm_relo[slot]; r0_ = c.lm_tmp[slot]; }
        };
        for (int p = wave; p < W1 * W1; p += nw) {
            int i = p / W1, j = p - i * W1;
            if (!(i < j)) continue;
            int cnt = 0;
            for (int k0 = 0; k0 < F; k0 += 64) {
                int k = k0 + lane;
                bool hit = false;
                bool hit2 = false;   // relocalisation record: listed with the pair (start, start + 1), its pose_j columns are zero
                if (k < F) {
                    int st_, no_, rl_, r0_;
                    lm_fields(k, st_, no_, rl_, r0_);
                    hit = st_ == i && no_ > j - i && r0_ + (j - i - 1) < nres;
                    hit2 = relo_on && j == i + 1 && st_ == i && rl_ && r0_ + no_ - 1 < nres;
                }
                cnt += __popcll(__ballot(hit)) + __popcll(__ballot(hit2));
            }
            if (lane == 0) c.pair_start[p] = cnt;
        }
        __syncthreads();
        PH(103);
        {
            // exclusive prefix sum of the W1^2 counts (one thread walking them was 441 dependent loads at W = 20): chunks per thread, Hillis-Steele
            // over the chunk sums in `scratch` (2 nt ints), as block_scan_flags does
            const int np = W1 * W1, chunk = (np + nt - 1) / nt, b0 = min(np, t * chunk), e0 = min(np, b0 + chunk);
            int vals[4], sum = 0;   // chunk <= 4: np <= 441 at W = 20 needs nt >= 111
            if (chunk <= 4) {
#pragma unroll
                for (int q = 0; q < 4; q++) { vals[q] = b0 + q < e0 ? c.pair_start[b0 + q] : 0; sum += vals[q]; }
                int *cur = scratch, *nxt = scratch + nt;
                __syncthreads();
                cur[t] = sum;
                __syncthreads();
                for (int off = 1; off < nt; off <<= 1) {
                    int v = cur[t];
                    if (t >= off) v += cur[t - off];
                    nxt[t] = v;
                    __syncthreads();
                    int *tmp = cur; cur = nxt; nxt = tmp;
                }
                int o = cur[t] - sum;
#pragma unroll
                for (int q = 0; q < 4; q++) if (b0 + q < e0) { c.pair_start[b0 + q] = o; o += vals[q]; }
                if (t == nt - 1) c.pair_start[np] = cur[nt - 1];
            } else if (t == 0) {
                int acc = 0;
                for (int p = 0; p < np; p++) { int v = c.pair_start[p]; c.pair_start[p] = acc; acc += v; }
                c.pair_start[np] = acc;
            }
        }
        __syncthreads();
        for (int p = wave; p < W1 * W1; p += nw) {
            int i = p / W1, j = p - i * W1;
            if (!(i < j)) continue;
            int o = c.pair_start[p];
            for (int k0 = 0; k0 < F; k0 += 64) {
                int k = k0 + lane;
                bool hit = false;
                int r = 0;
                if (k < F) { int st_, no_, rl_, r0_; lm_fields(k, st_, no_, rl_, r0_); r = r0_ + (j - i - 1); hit = st_ == i && no_ > j - i && r < nres; }
                unsigned long long m = __ballot(hit);
                if (hit) c.pair_list[o + __popcll(m & ((1ULL << lane) - 1ULL))] = r;
                o += __popcll(m);
            }
            if (relo_on && j == i + 1)
                for (int k0 = 0; k0 < F; k0 += 64) {
                    int k = k0 + lane;
                    bool hit = false;
                    int r = 0;
                    if (k < F) { int st_, no_, rl_, r0_; lm_fields(k, st_, no_, rl_, r0_); r = r0_ + no_ - 1; hit = st_ == i && rl_ && r < nres; }
                    unsigned long long m = __ballot(hit);
                    if (hit) c.pair_list[o + __popcll(m & ((1ULL << lane) - 1ULL))] = r;
                    o += __popcll(m);
                }
        }
        __syncthreads();
    }
    PH(104);
    if (t == 0) {
        be.n_in_problem = F; be.n_var_landmarks = Fa; be.n_residuals = nres - sh_i[3];   // (f_m_cnt counts the regular factors)
        be.relo_factors = sh_i[3];
        be.iterations = 0; be.successful = 0;
    }
    __syncthreads();
}

// double2vector + setDepth + failureDetection after the solve (estimator.cpp:985-1111, feature_manager.cpp:197-223, estimator.cpp:345-353).
// sdx: >= 12 doubles of LDS scratch.  Shared by the persistent solve kernel and the phased solver (ps_final_kernel).
__device__ __forceinline__ void solve_epilogue(Ctx &c, const Params &X, double cost, int iters_done, int succ, long long ts0, double *sdx, double *sh_d, int *sh_i) {
    const int t = threadIdx.x, nt = blockDim.x;
    const vio_config &cfg = c.C->c;
    BeSeq &be = *c.be;
    const int W = c.W, nlm = be.n_lm;
    // ---- write back flat parameters + double2vector (estimator.cpp:985-1111)
    if (!cfg.use_imu) {
        // VO mode (estimator.cpp:1060-1067, 1093, 1109): poses straight from the parameters, no gauge fix, nothing else handed back
        if (t == 0) {
            be.final_cost = cost; be.iterations = iters_done; be.successful = succ;
            be.iter_total += iters_done; be.solve_total++;
            be.dbg[4] = (int)(VIO_CLOCK() - ts0);
        }
        if (t <= W) {
            const int i = t;
            for (int k = 0; k < 7; k++) be.para_Pose[i][k] = X.pose[i * 7 + k];
            stm(be.Rs[i], q2R(qnormalized(mkq(X.pose[i * 7 + 6], X.pose[i * 7 + 3], X.pose[i * 7 + 4], X.pose[i * 7 + 5]))));
            be.Ps[i][0] = X.pose[i * 7]; be.Ps[i][1] = X.pose[i * 7 + 1]; be.Ps[i][2] = X.pose[i * 7 + 2];
        }
    } else {
    if (t == 0) {
        be.final_cost = cost; be.iterations = iters_done; be.successful = succ;
        be.iter_total += iters_done; be.solve_total++;
        be.dbg[4] = (int)(VIO_CLOCK() - ts0);
        v3 origin_R0 = R2ypr(ldm(be.Rs[0]));
        v3 origin_P0 = ld3(be.Ps[0]);
        quat q0 = mkq(X.pose[6], X.pose[3], X.pose[4], X.pose[5]);
        v3 origin_R00 = R2ypr(q2R(q0));
        double y_diff = origin_R0.x - origin_R00.x;
        m3 rot_diff = ypr2R(mk(y_diff, 0, 0));
        if (fabs(fabs(origin_R0.y) - 90) < 1.0 || fabs(fabs(origin_R00.y) - 90) < 1.0) rot_diff = mul(ldm(be.Rs[0]), tr(q2R(q0)));
        sh_d[0] = 0;
        for (int q = 0; q < 9; q++) sdx[q] = rot_diff.a[q];
        sdx[9] = origin_P0.x; sdx[10] = origin_P0.y; sdx[11] = origin_P0.z;
    }
    __syncthreads();
    // para_Pose as the solve leaves it (estimator.h:139): what Estimator::setReloFrame copies into relo_Pose when no marginalisation
    // re-runs vector2double afterwards (marg_body overwrites it in that case, like optimization() does)
    if (t <= W) for (int k = 0; k < 7; k++) be.para_Pose[t][k] = X.pose[t * 7 + k];
    if (t <= W) {
        int i = t;
        m3 rot_diff = ldm(sdx);
        v3 origin_P0 = mk(sdx[9], sdx[10], sdx[11]);
        quat qi = qnormalized(mkq(X.pose[i * 7 + 6], X.pose[i * 7 + 3], X.pose[i * 7 + 4], X.pose[i * 7 + 5]));
        stm(be.Rs[i], mul(rot_diff, q2R(qi)));
        st3(be.Ps[i], add(mul(rot_diff, mk(X.pose[i * 7] - X.pose[0], X.pose[i * 7 + 1] - X.pose[1], X.pose[i * 7 + 2] - X.pose[2])), origin_P0));
        st3(be.Vs[i], mul(rot_diff, mk(X.sb[i * 9], X.sb[i * 9 + 1], X.sb[i * 9 + 2])));
        for (int k = 0; k < 3; k++) { be.Bas[i][k] = X.sb[i * 9 + 3 + k]; be.Bgs[i][k] = X.sb[i * 9 + 6 + k]; }
        // updateLatestStates (estimator.cpp:1768-1776): latest_Bg feeds predictMotion of the next frame, whose front-end may
        // start as soon as this kernel is done (overlapping the marginalisation)
        // (the frame that completes the DYNAMIC initialisation returns without updateLatestStates, estimator.cpp:241-252)
        if (i == W && !be.init_frame) for (int k = 0; k < 3; k++) be.latest_Bg[k] = X.sb[i * 9 + 6 + k];
    }
    if (t == W + 1) {
        be.tic[0] = X.ex[0]; be.tic[1] = X.ex[1]; be.tic[2] = X.ex[2];
        stm(be.ric, q2R(qnormalized(mkq(X.ex[6], X.ex[3], X.ex[4], X.ex[5]))));
        if (cfg.estimate_td) be.td = X.td;
    }
    __syncthreads();
    if (t == 0 && be.relo_info) {
        // "relative info between two loop frame" (estimator.cpp:1034-1056): the relocalisation pose through the same gauge fix, the drift
        // between this odometry frame and the old keyframe's world (yaw + translation), the relative pose the pose graph stores
        const m3 rot_diff = ldm(sdx);
        const v3 origin_P0 = mk(sdx[9], sdx[10], sdx[11]);
        const m3 relo_r = mul(rot_diff, q2R(qnormalized(mkq(X.relo[6], X.relo[3], X.relo[4], X.relo[5]))));
        const v3 relo_t = add(mul(rot_diff, mk(X.relo[0] - X.pose[0], X.relo[1] - X.pose[1], X.relo[2] - X.pose[2])), origin_P0);
        const m3 prev_r = ldm(be.prev_relo_r);
        const double drift_yaw = R2ypr(prev_r).x - R2ypr(relo_r).x;
        const m3 dr = ypr2R(mk(drift_yaw, 0, 0));
        stm(be.drift_correct_r, dr);
        st3(be.drift_correct_t, sub(ld3(be.prev_relo_t), mul(dr, relo_t)));
        const int li = be.relo_local;
        const m3 Rl = ldm(be.Rs[li]);
        st3(be.relo_relative_t, mul(tr(relo_r), sub(ld3(be.Ps[li]), relo_t)));
        const quat qrel = R2q(mul(tr(relo_r), Rl));
        be.relo_relative_q[0] = qrel.w; be.relo_relative_q[1] = qrel.x; be.relo_relative_q[2] = qrel.y; be.relo_relative_q[3] = qrel.z;
        const double a = R2ypr(Rl).x - R2ypr(relo_r).x;   // Utility::normalizeAngle (utility.h:131-139), degrees
        be.relo_relative_yaw = a > 0 ? a - 360.0 * floor((a + 180.0) / 360.0) : a + 360.0 * floor((-a + 180.0) / 360.0);
        for (int k = 0; k < 7; k++) be.relo_Pose[k] = X.relo[k];
        be.relo_info = 0;
    }
    }   // IMU mode
    // setDepth (feature_manager.cpp:197-223)
    for (int k = t; k < nlm; k += nt) {
        int slot = c.lm_order[k];
        int pi = c.lm_pidx[slot];
        if (pi < 0) continue;
        double d = 1.0 / c.feat[pi];
        c.lm_depth[slot] = d;
        c.lm_solve[slot] = d < 0 ? 2 : 1;
    }
    __syncthreads();
    // failureDetection + clearState() / setParameter() (estimator.cpp:345-353, 1113-1159, 43-116, 15-41).  The reference runs it
    // after optimization() (solve + marginalisation) and throws the whole state away when it fires, so nothing the marginalisation
    // produces survives a reboot: it is decided HERE, before the host records ev_solve, because the reset rewrites imu_head / td /
    // ric / latest_Bg, which the next frame's front-end (fe_begin) and the IMU scatter kernel read as soon as this kernel is done.
    if (be.solver_flag == 1 && !be.init_frame) {
        if (t == 0) {
            int fail = 0;
            if (nrm(ld3(be.Bas[W])) > 2.5) fail = 1;
            if (nrm(ld3(be.Bgs[W])) > 1.0) fail = 1;
            v3 tmpP = ld3(be.Ps[W]);
            if (nrm(sub(tmpP, ld3(be.last_P))) > 5) fail = 1;
            if (fabs(tmpP.z - be.last_P[2]) > 1) fail = 1;
            sh_i[0] = fail;
        }
        __syncthreads();
        if (sh_i[0]) {
            for (int k = t; k < c.NL; k += nt) c.lm_free[k] = c.NL - 1 - k;
            if (t == 0) {
                for (int i = 0; i <= W + 1; i++) c.pre[i].valid = 0;
                for (int i = 0; i <= W; i++) {
                    for (int k = 0; k < 3; k++) { be.Ps[i][k] = 0; be.Vs[i][k] = 0; be.Bas[i][k] = 0; be.Bgs[i][k] = 0; }
                    stm(be.Rs[i], eye());
                    be.Headers[i] = 0;
                    be.pre_idx[i] = i;
                }
                for (int k = 0; k < 9; k++) be.ric[k] = cfg.ric[k];
                for (int k = 0; k < 3; k++) { be.tic[k] = cfg.tic[k]; be.latest_Bg[k] = 0; }
                be.td = cfg.td;
                be.g[0] = 0; be.g[1] = 0; be.g[2] = cfg.g_norm;   // setParameter(): g = G (estimator.cpp:26)
                be.first_imu = 0; be.frame_count = 0; be.solver_flag = 0; be.openExEstimation = 0; be.has_prior = 0;
                be.initFirstPoseFlag = 0; be.prevTime = -1; be.n_lm = 0; be.n_free = c.NL; be.ring_base = 0;
                be.imu_head = be.imu_count_ingest;  // clearState() empties imu_buf (samples pushed for the next frame while this one was
                                                     // being optimised - tracker lag 1 - arrived after the reset)
                be.relo_info = 0;   // relocalization_info = false (estimator.cpp:96)
                be.reboot_count++;
                be.status_code = VIO_REBOOTED;
                be.rebooted = 1;
                be.do_marg = 0;
                be.overflow = 0;
            }
        }
    }
}

}  // namespace

template <bool EXACT> __device__ void marg_body(const Batch &B, int s, int *scratch, double *sred, unsigned char *smem_marg);
__device__ void finish_body(const Batch &B, int s, int *scratch, PreWork &pw, unsigned char *smem_marg);

// solve -> marginalise -> window slide for one sequence per workgroup (1024 threads); fusing the three stages makes the
// step time the maximum over sequences of the *sum* of the stage times instead of the sum of per-stage maxima.
__device__ __forceinline__ void solve_body(const Batch &B, int s, int *scratch, double *sred, unsigned char *smem, PreWork &pw);

__global__ __launch_bounds__(1024) void be_solve_kernel(Batch B) {
    const int s = blockIdx.x + B.s0;
    __shared__ int scratch[2 * 1024 + 8];
    __shared__ double sred[64];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PreWork &pw = *(PreWork *)(smem + 16 * ((B.cfg->LW * 8 + 15) / 16));  // aliases the work region (used before it)
    solve_body(B, s, scratch, sred, smem, pw);
}
// the same body compiled for 512 threads: 256 VGPRs per lane instead of 128 (no scratch spills), half the waves
__global__ __launch_bounds__(512) void be_solve_kernel_512(Batch B) {
    const int s = blockIdx.x + B.s0;
    __shared__ int scratch[2 * 1024 + 8];
    __shared__ double sred[64];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PreWork &pw = *(PreWork *)(smem + 16 * ((B.cfg->LW * 8 + 15) / 16));
    solve_body(B, s, scratch, sred, smem, pw);
}
// marginalisation + window slide: 512 threads (measured: 2.9 ms against 3.5 ms with 256; the QL phase is one wavefront either way)
__global__ __launch_bounds__(512) void be_marg_kernel(Batch B) {
    const int s = blockIdx.x + B.s0;
    __shared__ int scratch[2 * 512 + 8];
    __shared__ double sred[64];
    __shared__ PreWork pw;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    marg_body<false>(B, s, scratch, sred, smem);
    __syncthreads();
    finish_body(B, s, scratch, pw, smem);
}
// vio_config.marg_exact: the same stage with the marginalisation of marginalization_factor.cpp:281-315 followed literally (its own kernel so
// that the hot kernel's registers / LDS are untouched by the parity instrument)
__global__ __launch_bounds__(512) void be_marg_exact_kernel(Batch B) {
    const int s = blockIdx.x + B.s0;
    __shared__ int scratch[2 * 512 + 8];
    __shared__ double sred[64];
    __shared__ PreWork pw;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    marg_body<true>(B, s, scratch, sred, smem);
    __syncthreads();
    finish_body(B, s, scratch, pw, smem);
}

__device__ __forceinline__ void solve_body(const Batch &B, int s, int *scratch, double *sred, unsigned char *smem, PreWork &pw) {
    const int t = threadIdx.x, nt = blockDim.x;
    Ctx c = make_ctx(B, s);
    const DevCfg &C = *B.cfg;
    const vio_config &cfg = C.c;
    BeSeq &be = *c.be;
    if (!be.do_solve) return;
    const int W = c.W, P = c.P, LW = c.LW, W1 = W + 1;
    __shared__ Params X, Xc;
    __shared__ double sdx[6 * VIO_MAXW + 16], srp[6 * VIO_MAXW + 16];
    __shared__ int sh_i[8];
    __shared__ double sh_d[8];
    double *xs = (double *)smem;  // LW doubles: triangular-solve workspace
    double *work = xs + LW;       // LDS scratch: whitened IMU Jacobians, then the frame-pair blocks
    const int npairs = (W + 1) * W / 2;
    double *pb = (npairs * 210 <= 12288) ? work : c.pairblk;  // 55 pairs x 210 doubles = 92 KB for W = 10
    __shared__ double chol_dinv[VIO_LWMAX];  // reciprocal Cholesky diagonal (LDS-tile path)
    const bool tiles_in_lds = ((LW >> 4) * ((LW >> 4) + 1) / 2) * 256 <= 16896 && !(B.flags & 1);  // S as 66 lower tiles = 132 KB for W = 10
    // column-aware Schur staging (schur_mfma_staged) and two-range Hpl mat-vecs in use: see assemble() / the solver loop
    const bool schur_staged = tiles_in_lds && ((LW >> 4) * ((LW >> 4) + 1) / 2) <= 9 * (nt >> 6) &&
                              2 * SCH_CH * (LW + 8) + 64 <= ((LW >> 4) * ((LW >> 4) + 1) / 2) * 256 && (LW >> 4) <= 32;
    const bool hpl_sparse = schur_staged && 6 * W1 + 7 <= 128;

    PH_INIT;
    const long long ts0 = VIO_CLOCK();
    int F, Fa, nres;
    solve_prologue(B, c, X, scratch, pw, sh_i, F, Fa, nres);
    {
        // The persistent solver (VIO_SOLVE_MODE=0 and the fallback for residual counts beyond the phased solver's range) honours the inverse-depth
        // bound by clamping candidates only; Ceres' treatment of the bounds-constrained program (projection of x0, Armijo line search) lives in
        // the phased solver (be_phased.h).  A frame that meets a bounded landmark here is flagged (overflow bit 128).
        const int *al = c.pair_list + c.nres_cap - c.NL;
        bool hit = false;
        for (int k = t; k < Fa; k += nt) hit = hit || c.lm_est[al[k]] == 2;
        if (hit) atomicOr(&c.be->overflow, 128);
    }
    const int nlm = be.n_lm;
    int *alist = c.pair_list + c.nres_cap - c.NL;              // variable-landmark slots, kept for the whole solve
    const int n = c.NPR;
    PH(2);
    const int ex_active = sh_i[0], td_active = sh_i[1];
    const bool vext = ex_active || td_active;   // extrinsic / td Jacobians are only evaluated when one of the blocks is a variable
    const int ne_ext = vext ? 7 : 0;
    const int oE = 15 * W1, oT = 15 * W1 + 6;

    // vec slots
    double *g = c.vec, *sp = c.vec + 1 * LW, *dgp = c.vec + 2 * LW, *gradp = c.vec + 3 * LW, *gnp = c.vec + 4 * LW, *stp = c.vec + 5 * LW,
           *gs = c.vec + 6 * LW, *tmpv = c.vec + 7 * LW, *delta = c.vec + 8 * LW, *sgp = c.vec + 9 * LW, *hsgp = c.vec + 10 * LW,
           *yp = c.vec + 11 * LW, *up = c.vec + 12 * LW, *tmpv2 = c.vec + 13 * LW;
    double *sl = c.lvec, *dgl = c.lvec + c.NLs, *gradl = c.lvec + 2 * c.NLs, *gnl = c.lvec + 3 * c.NLs, *stl = c.lvec + 4 * c.NLs,
           *inv = c.lvec + 5 * c.NLs, *gls = c.lvec + 6 * c.NLs, *Hlls = c.lvec + 7 * c.NLs;
    const int Kpad = (Fa + 3) & ~3;
    double *hsgl = c.res + (size_t)c.nres_cap * 42 - 4 * (size_t)c.NLs;   // tail of the residual buffer (nres <= nres_cap - 2 NL)
    double *yl = hsgl + c.NLs, *ul = yl + c.NLs, *tmpl = ul + c.NLs;

    double cost = evaluate(c, X, c.feat, true, nres, sred, sdx, srp, work, vext);
    PH(4);
    assemble(B, c, X, nres, Fa, alist, srp, work, pb, hpl_sparse, vext);
    PH(5);
    if (t == 0) be.initial_cost = cost;
    // Jacobi scaling (once): 1/(1+||J_j||); constant blocks (ex / td when not estimated) get scale 0 = removed from the problem
    for (int a = t; a < LW; a += nt) {
        bool act = a < P && (a < oE ? true : (a < oT ? ex_active != 0 : td_active != 0));
        if (!cfg.use_imu && (a < 6 || (a >= 6 * W1 && a < oE))) act = false;   // VO mode: pose 0 constant, no speed-bias blocks (estimator.cpp:1178-1185)
        sp[a] = act ? 1.0 / (1.0 + sqrt(c.H[a * LW + a])) : 0.0;
    }
    for (int k = t; k < Kpad; k += nt) sl[k] = k < Fa ? 1.0 / (1.0 + sqrt(c.Hll[k])) : 0.0;
    __syncthreads();
    // H / Hpl stay unscaled in HBM; the column scaling S is applied to the vectors (and inside the Schur MFMA operand loads):
    //   Hs v = S H (S v).  Returns the max-norm of the active gradient.
    auto prepare_point = [&]() -> double {
        double m = 0;
        for (int a = t; a < P; a += nt) m = fmax(m, sp[a] != 0.0 ? fabs(g[a]) : 0.0);
        for (int k = t; k < Fa; k += nt) m = fmax(m, fabs(c.gl[k]));
        double r = block_max(m, sred);
        for (int a = t; a < LW; a += nt) {
            double hs = a < P ? sp[a] * sp[a] * c.H[a * LW + a] : 0.0;
            gs[a] = a < P ? sp[a] * g[a] : 0.0;
            dgp[a] = sqrt(fmin(fmax(hs, 1e-6), 1e32));
            gradp[a] = gs[a] / dgp[a];
            sgp[a] = gradp[a] / dgp[a];
            up[a] = sp[a] * sgp[a];
        }
        for (int k = t; k < Kpad; k += nt) {
            double hl = k < Fa ? sl[k] * sl[k] * c.Hll[k] : 0.0;
            Hlls[k] = hl;
            gls[k] = k < Fa ? sl[k] * c.gl[k] : 0.0;
            dgl[k] = sqrt(fmin(fmax(hl, 1e-6), 1e32));
            gradl[k] = gls[k] / dgl[k];
            ul[k] = sl[k] * (gradl[k] / dgl[k]);
        }
        __syncthreads();
        return r;
    };

    double radius = 1e4, mu = 1e-8, alpha = 0, dogleg_norm = 0;
    // Cauchy point, computed lazily: the dogleg only needs it when the Gauss-Newton step leaves the trust region (rare with the
    // initial radius 1e4).  It depends on H, g only, which stay valid until the next assemble(); the LDS work region is free then.
    bool cauchy_valid = false;
    auto compute_cauchy = [&]() {
        // Cauchy point: alpha = |grad|^2 / |J D^-1 grad|^2, and H_full * (D^-1 grad) is kept for the model evaluation
        matvec_pass(c.H, LW, P, P, nullptr, up, nullptr, tmpv, work);   // H (S sg_p): H is symmetric, row dots
        matvec_pass_2range(c.Hpl, LW, Fa, P, 6 * W1, 15 * W1, ne_ext, ul, up, tmpv2, tmpl, work);       // Hpl^T (Sl sg_l) and Hpl (S sg_p) in one pass
        double g2 = 0, jg2 = 0;
        for (int a = t; a < LW; a += nt) {
            double v = a < P ? sp[a] * (tmpv[a] + tmpv2[a]) : 0.0;
            hsgp[a] = v;
            if (a < P) { g2 += gradp[a] * gradp[a]; jg2 += sgp[a] * v; }
        }
        for (int k = t; k < Kpad; k += nt) {
            double sgl = k < Fa ? gradl[k] / dgl[k] : 0.0;
            double v = k < Fa ? sl[k] * tmpl[k] + Hlls[k] * sgl : 0.0;
            hsgl[k] = v;
            if (k < Fa) { g2 += gradl[k] * gradl[k]; jg2 += sgl * v; }
        }
        block_sum2(g2, jg2, sred);
        alpha = g2 / jg2;
        cauchy_valid = true;
    };

    bool reuse = false, need_eval = false, have_J = false;
    int invalid = 0;
    int iters_done = 0, succ = 0;
    if (prepare_point() > 1e-10)
    for (int iter = 1; iter <= cfg.max_iterations; iter++) {
        iters_done = iter;
        if (!reuse) {
            if (need_eval) {
                PH(13);
                if (!have_J) cost = evaluate(c, X, c.feat, true, nres, sred, sdx, srp, work, vext);
                have_J = false;
                PH(4);
                assemble(B, c, X, nres, Fa, alist, srp, work, pb, hpl_sparse, vext);
                PH(5);
                need_eval = false;
                if (prepare_point() <= 1e-10) { iters_done = iter - 1; break; }
                PH(6);
            }
            cauchy_valid = false;
            // Gauss-Newton step via the landmark Schur complement (FP64 matrix cores), regularised by mu * D^2
            bool ok = false;
            while (mu < 1.0) {
                for (int k = t; k < Kpad; k += nt) {
                    double iv = k < Fa ? 1.0 / (Hlls[k] + mu * dgl[k] * dgl[k]) : 0.0;
                    inv[k] = iv;
                    tmpl[k] = sl[k] * iv * gls[k];
                }
                __syncthreads();
                matvec_pass_2range(c.Hpl, LW, Fa, P, 6 * W1, 15 * W1, ne_ext, tmpl, nullptr, tmpv, nullptr, work);  // Hpl^T (Sl gls / hll); uses the work region before S moves in
                for (int a = t; a < LW; a += nt) xs[a] = a < P ? gs[a] - sp[a] * tmpv[a] : 0.0;
                for (int k = t; k < Kpad; k += nt) tmpl[k] = sl[k] * sl[k] * inv[k];  // per-row factor of the rank-K update
                __syncthreads();
                bool chol_ok;
                if (tiles_in_lds) {
                    {
                        // column tiles of Hpl that can be non-zero: poses (columns 0 .. 6 W1 - 1) and extrinsic / td (15 W1 .. 15 W1 + 6)
                        unsigned colmask = 0;
                        for (int cb = 0; cb < (LW >> 4); cb++) {
                            const int c0 = 16 * cb, c1 = c0 + 15;
                            if (c0 < 6 * W1 || (vext && c1 >= 15 * W1 && c0 < 15 * W1 + 7)) colmask |= 1u << cb;
                        }
                        if (schur_staged)
                            schur_mfma_staged<9>(c.H, c.Hpl, tmpl, dgp, sp, mu, Kpad, LW, LW, work, colmask);
                        else schur_mfma_lds(c.H, c.Hpl, tmpl, dgp, sp, mu, Kpad, LW, LW, work);
                    }
                    PH(8);
                    chol_ok = chol_tiles(work, LW >> 4, &sh_i[2], chol_dinv, nullptr, xs);   // with the forward substitution
                } else {
                    schur_mfma(c.H, c.Hpl, tmpl, dgp, sp, mu, Kpad, LW, LW, c.Sc);
                    PH(8);
                    chol_ok = chol_blocked(c.Sc, LW, LW, &sh_i[2], work);
                }
                PH(9);
                if (chol_ok) {
                    if (tiles_in_lds) chol_backward_tiles_wave(work, LW >> 4, xs, chol_dinv);
                    else chol_solve_blocked(c.Sc, LW, LW, xs, work);
                    PH(10);
                    double bad = 0;
                    for (int a = t; a < P; a += nt) if (!isfinite(xs[a])) bad += 1;
                    bad = block_sum(bad, sred);
                    if (bad == 0) {
                        for (int a = t; a < LW; a += nt) { yp[a] = xs[a]; gnp[a] = -xs[a] * dgp[a]; tmpv[a] = sp[a] * xs[a]; }
                        __syncthreads();
                        matvec_pass_2range(c.Hpl, LW, Fa, P, 6 * W1, 15 * W1, ne_ext, nullptr, tmpv, nullptr, tmpl, nullptr);  // Hpl (S y_p)
                        for (int k = t; k < Kpad; k += nt) {
                            double y = k < Fa ? (gls[k] - sl[k] * tmpl[k]) * inv[k] : 0.0;
                            yl[k] = y;
                            gnl[k] = -y * dgl[k];
                        }
                        __syncthreads();
                        ok = true;
                        break;
                    }
                }
                mu *= 10.0;
            }
            if (!ok) break;
            reuse = true;
        }
        // traditional dogleg in the D-scaled space
        double gnorm = 0, gnn = 0, gdot = 0;
        for (int a = t; a < P; a += nt) { gnorm += gradp[a] * gradp[a]; gnn += gnp[a] * gnp[a]; gdot += gradp[a] * gnp[a]; }
        for (int k = t; k < Fa; k += nt) { gnorm += gradl[k] * gradl[k]; gnn += gnl[k] * gnl[k]; gdot += gradl[k] * gnl[k]; }
        block_sum3(gnorm, gnn, gdot, sred);
        gnorm = sqrt(gnorm);
        gnn = sqrt(gnn);
        double ca = 0, cb = 0;  // step = ca * grad + cb * gn
        if (!(gnn <= radius) && !cauchy_valid) { PH(11); compute_cauchy(); PH(7); }
        if (gnn <= radius) { ca = 0; cb = 1; dogleg_norm = gnn; }
        else if (gnorm * alpha >= radius) { ca = -(radius / gnorm); cb = 0; dogleg_norm = radius; }
        else {
            double b_dot_a = -alpha * gdot;
            double a_sq = (alpha * gnorm) * (alpha * gnorm);
            double bma = a_sq - 2 * b_dot_a + gnn * gnn;
            double cc = b_dot_a - a_sq;
            double d = sqrt(cc * cc + bma * (radius * radius - a_sq));
            double beta = (cc <= 0) ? (d - cc) / bma : (radius * radius - a_sq) / (d + cc);
            ca = -alpha * (1.0 - beta); cb = beta;
            dogleg_norm = -1;
        }
        // step = ca * D^-1 grad - cb * y ;  H_full step = ca * H_full (D^-1 grad) - cb * (g' - mu D^2 y)   [(H_full + mu D^2) y = g']
        double n2 = 0, lin = 0, quad = 0;
        for (int a = t; a < LW; a += nt) {
            double v = ca * gradp[a] + cb * gnp[a];
            double st = a < P ? v / dgp[a] : 0.0;
            stp[a] = st;
            if (a < P) {
                n2 += v * v;
                lin += st * gs[a];
                quad += st * ((ca != 0.0 ? ca * hsgp[a] : 0.0) - cb * (gs[a] - mu * dgp[a] * dgp[a] * yp[a]));
            }
        }
        for (int k = t; k < Kpad; k += nt) {
            double v = k < Fa ? ca * gradl[k] + cb * gnl[k] : 0.0;
            double st = k < Fa ? v / dgl[k] : 0.0;
            stl[k] = st;
            if (k < Fa) {
                n2 += v * v;
                lin += st * gls[k];
                quad += st * ((ca != 0.0 ? ca * hsgl[k] : 0.0) - cb * (gls[k] - mu * dgl[k] * dgl[k] * yl[k]));
            }
        }
        block_sum3(n2, lin, quad, sred);
        if (dogleg_norm < 0) dogleg_norm = sqrt(n2);
        double model_change = -(lin + 0.5 * quad);
        PH(11);
        if (!(model_change > 0)) {
            if (++invalid >= 5) break;
            mu *= 10.0;
            reuse = false;
            continue;
        }
        invalid = 0;
        // candidate = Plus(x, step .* scale)
        for (int a = t; a < P; a += nt) delta[a] = stp[a] * sp[a];
        __syncthreads();
        if (t <= W) {
            for (int k = 0; k < 7; k++) Xc.pose[t * 7 + k] = X.pose[t * 7 + k];
            bf::pose_plus(&Xc.pose[t * 7], &delta[6 * t]);
            for (int k = 0; k < 9; k++) Xc.sb[t * 9 + k] = X.sb[t * 9 + k] + delta[6 * W1 + 9 * t + k];
        }
        if (t == W + 1) {
            for (int k = 0; k < 7; k++) Xc.ex[k] = X.ex[k];
            if (ex_active) bf::pose_plus(Xc.ex, &delta[oE]);
            Xc.td = X.td + (td_active ? delta[oT] : 0.0);
        }
        for (int k = t; k < F; k += nt) c.cfeat[k] = c.feat[k];
        __syncthreads();
        for (int k = t; k < Fa; k += nt) {
            int slot = alist[k], pi = c.lm_pidx[slot];
            double v = c.feat[pi] + stl[k] * sl[k];
            double ub = (c.lm_est[slot] == 2) ? 2.0 / cfg.depth_max : 1.7976931348623157e308;
            if (v > ub) v = ub;
            c.cfeat[pi] = v;
        }
        __syncthreads();
        // The candidate is evaluated WITH Jacobians (res / imu_raw / srp are only read by assemble(), which is not called again if the
        // step is rejected): an accepted point then goes straight to assemble() instead of being evaluated a second time.
        const bool cand_with_J = iter < cfg.max_iterations;
        double ccost = evaluate(c, Xc, c.cfeat, cand_with_J, nres, sred, sdx, srp, work, vext);
        PH(12);
        // parameter tolerance
        double xn = 0, dn = 0;
        if (t <= W) {
            for (int k = 0; k < 7; k++) { double v = X.pose[t * 7 + k]; xn += v * v; double d = v - Xc.pose[t * 7 + k]; dn += d * d; }
            for (int k = 0; k < 9; k++) { double v = X.sb[t * 9 + k]; xn += v * v; double d = v - Xc.sb[t * 9 + k]; dn += d * d; }
        }
        if (t == W + 1) {
            if (ex_active) for (int k = 0; k < 7; k++) { double v = X.ex[k]; xn += v * v; double d = v - Xc.ex[k]; dn += d * d; }
            if (td_active) { xn += X.td * X.td; dn += (X.td - Xc.td) * (X.td - Xc.td); }
        }
        for (int k = t; k < Fa; k += nt) { int pi = c.lm_pidx[alist[k]]; double v = c.feat[pi]; xn += v * v; double d = v - c.cfeat[pi]; dn += d * d; }
        block_sum2(xn, dn, sred);
        if (sqrt(dn) <= 1e-8 * (sqrt(xn) + 1e-8)) break;
        if (fabs(cost - ccost) <= 1e-6 * cost) break;
        double rel = (cost - ccost) / model_change;
        if (rel > 1e-3) {
            __syncthreads();
            if (t <= W) { for (int k = 0; k < 7; k++) X.pose[t * 7 + k] = Xc.pose[t * 7 + k]; for (int k = 0; k < 9; k++) X.sb[t * 9 + k] = Xc.sb[t * 9 + k]; }
            if (t == W + 1) { for (int k = 0; k < 7; k++) X.ex[k] = Xc.ex[k]; X.td = Xc.td; }
            for (int k = t; k < F; k += nt) c.feat[k] = c.cfeat[k];
            __syncthreads();
            cost = ccost;
            succ++;
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_norm);
            mu = fmax(1e-8, 2.0 * mu / 10.0);
            reuse = false;
            need_eval = true;
            have_J = cand_with_J;
        } else {
            radius *= 0.5;
            reuse = true;
        }
    }
    __syncthreads();
    PH(13);
    solve_epilogue(c, X, cost, iters_done, succ, ts0, sdx, sh_d, sh_i);
}

// ====================================================================================================== be_marg
// Marginalisation in the canonical layout. Landmarks seen first in frame 0 are eliminated analytically (their block of
// A_mm is diagonal), then pose_0/speedbias_0 (15) through a truncated eigen-decomposition, then the kept block is
// re-factorised as J^T J by a second (parallel Jacobi) eigen-decomposition (marginalization_factor.cpp:276-308).
// marg_exact (vio_config): elimination of the marginalised block and the new prior exactly as MarginalizationInfo::marginalize does them
// (marginalization_factor.cpp:270-315).  In: A (mq x mq over q = [md | n]) and b with NO landmark eliminated, the landmark coupling rows
// Cl (F0 x ldc, columns indexed like q), d_l = c.Hll (J_l^T J_l), c.gl (J_l^T r).  Marginalised block: m = md + F0 = [pose 0, speed-bias 0
// | inverse depths of the landmarks that start in frame 0] (md = 6 and F0 = 0 for MARGIN_SECOND_NEW).  Both eigen-decompositions are the
// threshold Jacobi of jacobi_block (the oracle's om::sym_eig is the cyclic form of the same iteration) on matrices in HBM scratch.
// LDS layout of the literal marginalisation's eigen-decompositions: the matrix (ld = n | 1: conflict-free column access), then d / e / g and
// the per-wavefront partial sums of sym_eig_tridiag_mt.  Host side: marg_exact_lds_bytes (vio_abi.hip) sizes the launch with the same formula.
#define MARG_EIG_AUX_DOUBLES (11 * EIG_LD)
// Symmetric eigen-decomposition in LDS: A (n x n, ld) -> eigenvectors in place (columns), eigenvalues in d.  Householder tridiagonalisation over
// the whole workgroup + implicit QL on one wavefront (be_linalg.h; the same pair be_prior_factor_kernel runs).  n <= 128 (tridiag_ql_wave).
__device__ __forceinline__ void sym_eig_lds(double *Al, int n, int ld, double *aux, bool one_wave) {
    double *d = aux, *e = aux + EIG_LD, *g = aux + 2 * EIG_LD, *part = aux + 3 * EIG_LD;
    if (one_wave) sym_eig_tridiag(Al, n, ld, d, e, g, part);   // one wavefront, wave-level ordering only (the others wait at its final barrier)
    else sym_eig_tridiag_mt(Al, n, ld, d, e, g, part);
    tridiag_ql_wave(Al, n, ld, d, e);
}
// The SECOND half of MarginalizationInfo::marginalize (marginalization_factor.cpp:293-315), literally: saes2(A), S = eigenvalues > 1e-8,
// linearized_jacobians = S^1/2 V^T, linearized_residuals = S^-1/2 V^T b, and from them what the solver consumes (J^T J, J^T r, |r|^2).
// Ar (n x n, HBM) / br (n) = the reduced system.  LDS-resident Householder + implicit QL when n <= MXL, else Jacobi sweeps over HBM scratch
// (scrA: n x n for the symmetrised matrix, scrV: n x n for the eigenvectors).  Returns the sweep count of the fallback (0 = LDS path).
// Not inlined: its own register allocation, one copy in the kernel for both literal modes (marg_exact 1 and 2).
__device__ __noinline__ int marg_literal_prior(const Ctx &c, BeSeq &be, const double *Ar, const double *br, int n, double *scrA, double *scrV, double *sred,
                                               unsigned char *smem) {
    const int t = threadIdx.x, nt = blockDim.x;
    const double eps = 1e-8;
    extern __shared__ __attribute__((aligned(16))) unsigned char marg_dyn_lds[];
    __shared__ double ev2[EIG_LD], vb2[EIG_LD];
    int sw2 = 0;
    if (n <= c.C->MXL) {
        const int ld = n | 1;
        double *Al = (double *)marg_dyn_lds, *aux = Al + (size_t)n * ld;
        for (int w = t; w < n * n; w += nt) { const int i = w / n, j = w - i * n; Al[i * ld + j] = 0.5 * (Ar[i * n + j] + Ar[j * n + i]); }
        __syncthreads();
        sym_eig_lds(Al, n, ld, aux, c.C->eig_one_wave != 0);   // SelfAdjointEigenSolver<MatrixXd> saes2(A) (:298)
        for (int k = t; k < n; k += nt) {
            const double ev = aux[k];
            double vb = 0;
            for (int i = 0; i < n; i++) vb += Al[i * ld + k] * br[i];
            c.prior_rf[k] = sqrt(ev > eps ? 1.0 / ev : 0.0) * vb;
            aux[EIG_LD + k] = sqrt(ev > eps ? ev : 0.0);
        }
        __syncthreads();
        // J = S^1/2 V^T: column k of the eigenvector array scaled in place, so Al[i][k] = J[k][i]
        for (int w = t; w < n * n; w += nt) { const int i = w / n, k = w - i * n; const double v = aux[EIG_LD + k] * Al[i * ld + k]; Al[i * ld + k] = v; c.prior_J[k * n + i] = v; }
        __syncthreads();
        // what MarginalizationFactor::Evaluate, the solver and the next marginalisation consume of (J, r): J^T J, J^T r and |r|^2
        for (int w = t; w < n * n; w += nt) {
            const int a = w / n, bb = w - a * n;
            double sacc = 0;
            for (int k = 0; k < n; k++) sacc += Al[a * ld + k] * Al[bb * ld + k];
            c.prior_H[w] = sacc;
        }
        for (int a = t; a < n; a += nt) {
            double sacc = 0;
            for (int k = 0; k < n; k++) sacc += Al[a * ld + k] * c.prior_rf[k];
            c.prior_r[a] = sacc;
        }
    } else {
        double *cs = (double *)smem, *sn = cs + 256;
        int *pp = (int *)(sn + 256), *qq = pp + 256;
        double *As = scrA, *V2 = scrV;
        for (int w = t; w < n * n; w += nt) { const int i = w / n, j = w - i * n; As[w] = 0.5 * (Ar[i * n + j] + Ar[j * n + i]); }
        __syncthreads();
        // round 6: Householder + implicit QL on the matrix where it lies (sym_eig_hbm) instead of cyclic Jacobi sweeps over HBM (VIO_MARG_EIG_JACOBI = 1)
        const bool hbm_ql = !c.C->eig_jacobi;
        if (hbm_ql) { sym_eig_hbm(As, n, n, (double *)marg_dyn_lds, sred); V2 = As; }
        else sw2 = jacobi_block(As, V2, n, n, cs, sn, pp, qq, sred);
        for (int k = t; k < n; k += nt) {
            ev2[k] = hbm_ql ? ((const double *)marg_dyn_lds)[k] : As[k * n + k];
            double vb = 0;
            for (int i = 0; i < n; i++) vb += V2[i * n + k] * br[i];
            vb2[k] = vb;
        }
        __syncthreads();
        for (int w = t; w < n * n; w += nt) {
            const int k = w / n, i = w - k * n;
            const double S = ev2[k] > eps ? ev2[k] : 0.0;
            c.prior_J[w] = sqrt(S) * V2[i * n + k];
        }
        for (int k = t; k < n; k += nt) {
            const double Sinv = ev2[k] > eps ? 1.0 / ev2[k] : 0.0;
            c.prior_rf[k] = sqrt(Sinv) * vb2[k];
        }
        __syncthreads();
        for (int w = t; w < n * n; w += nt) {
            const int a = w / n, bb = w - a * n;
            double sacc = 0;
            for (int k = 0; k < n; k++) sacc += c.prior_J[k * n + a] * c.prior_J[k * n + bb];
            c.prior_H[w] = sacc;
        }
        for (int a = t; a < n; a += nt) {
            double sacc = 0;
            for (int k = 0; k < n; k++) sacc += c.prior_J[k * n + a] * c.prior_rf[k];
            c.prior_r[a] = sacc;
        }
    }
    __syncthreads();
    double acc = 0;
    for (int k = t; k < n; k += nt) acc += c.prior_rf[k] * c.prior_rf[k];
    const double c0 = block_sum(acc, sred);
    __syncthreads();
    if (t == 0) be.prior_c0 = c0;
    return sw2;
}

__device__ void marg_exact_finish(const Ctx &c, BeSeq &be, double *A, const double *b, int md, int mq, int n, const double *Cl, int ldc, int F0,
                                  bool second_new, double *sred, unsigned char *smem) {
    const int t = threadIdx.x, nt = blockDim.x;
    const double eps = 1e-8;
    const int MX = c.C->MX, m = md + F0;
    double *Emm = c.margE, *EV = Emm + (size_t)MX * MX, *Einv = EV + (size_t)MX * MX, *ET1 = Einv + (size_t)MX * MX;
    // the dynamic LDS of this kernel, declared here so that the eigen-solver below is compiled for ds_* accesses (a pointer handed down
    // through two calls would be flat); same base address as the caller's smem
    extern __shared__ __attribute__((aligned(16))) unsigned char marg_dyn_lds[];
    double *cs = (double *)smem, *sn = cs + 256;
    int *pp = (int *)(sn + 256), *qq = pp + 256;
    // Amm = 0.5 (Amm + Amm^T) (:276); the landmark-landmark block is diagonal (an inverse depth only meets itself)
    auto amm = [&](int i, int j) -> double {
        if (i < md && j < md) return 0.5 * (A[i * mq + j] + A[j * mq + i]);
        if (i >= md && j >= md) return i == j ? c.Hll[i - md] : 0.0;
        return Cl[(size_t)((i >= md ? i : j) - md) * ldc + (i >= md ? j : i)];
    };
    // The first eigen-decomposition (saes(Amm), :281) runs LDS-resident when the block fits (m <= MXL): Householder + implicit QL instead of
    // Jacobi sweeps over HBM (round 5); the second one lives in marg_literal_prior.
    int sw1 = 0;
    double *Ar = c.margV, *br = c.vec;
    const long long tx0 = VIO_CLOCK();   // (timers build only: dbg[7..10] = ticks to the end of eig 1 / the Schur products / eig 2 / the prior; tools/marg_exact_probe.py)
    if (m <= c.C->MXL) {
        const int ld = m | 1;
        double *Al = (double *)marg_dyn_lds, *aux = Al + (size_t)m * ld;
        for (int w = t; w < m * m; w += nt) { const int i = w / m, j = w - i * m; Al[i * ld + j] = amm(i, j); }
        __syncthreads();
        sym_eig_lds(Al, m, ld, aux, c.C->eig_one_wave != 0);
        if (VIO_TIMERS && t == 0) be.dbg[7] = (int)(VIO_CLOCK() - tx0);
        // Amm_inv = V diag(lambda > eps ? 1 / lambda : 0) V^T (:281-283); 1 / lambda once per column
        for (int k = t; k < m; k += nt) { const double ev = aux[k]; aux[EIG_LD + k] = ev > eps ? 1.0 / ev : 0.0; }
        __syncthreads();
        for (int w = t; w < m * m; w += nt) {
            const int i = w / m, j = w - i * m;
            double sacc = 0;
            for (int k = 0; k < m; k++) sacc += Al[i * ld + k] * Al[j * ld + k] * aux[EIG_LD + k];
            Einv[w] = sacc;
        }
    } else {
        for (int w = t; w < m * m; w += nt) { const int i = w / m, j = w - i * m; Emm[w] = amm(i, j); }
        __syncthreads();
        if (!c.C->eig_jacobi) {
            // round 6: Householder + implicit QL on the matrix where it lies (be_linalg.h sym_eig_hbm; 23 ms at m = 185 against 50 - 100 ms of Jacobi sweeps)
            double *ewk = (double *)marg_dyn_lds;
            sym_eig_hbm(Emm, m, m, ewk, sred);
            if (VIO_TIMERS && t == 0) be.dbg[7] = (int)(VIO_CLOCK() - tx0);
            for (int k = t; k < m; k += nt) { const double ev = ewk[k]; ewk[SYM_EIG_HBM_MAX + k] = ev > eps ? 1.0 / ev : 0.0; }
            __syncthreads();
            for (int w = t; w < m * m; w += nt) {   // Amm_inv = V diag(lambda > eps ? 1 / lambda : 0) V^T: rows i and j of V, sixteen columns per trip
                const int i = w / m, j = w - i * m;
                const double *vi = Emm + (size_t)i * m, *vj = Emm + (size_t)j * m;
                double sacc = 0;
                for (int k0 = 0; k0 < m; k0 += 16) {
                    double a[16], b[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) { const int k = min(k0 + u, m - 1); a[u] = vi[k]; b[u] = vj[k]; }
#pragma unroll
                    for (int u = 0; u < 16; u++) if (k0 + u < m) sacc += a[u] * b[u] * ewk[SYM_EIG_HBM_MAX + k0 + u];
                }
                Einv[w] = sacc;
            }
        } else {
        sw1 = jacobi_block(Emm, EV, m, m, cs, sn, pp, qq, sred);
        if (VIO_TIMERS && t == 0) be.dbg[7] = (int)(VIO_CLOCK() - tx0);
        for (int w = t; w < m * m; w += nt) {
            const int i = w / m, j = w - i * m;
            double sacc = 0;
            for (int k = 0; k < m; k++) { const double ev = Emm[(size_t)k * m + k]; if (ev > eps) sacc += EV[(size_t)i * m + k] * EV[(size_t)j * m + k] / ev; }
            Einv[w] = sacc;
        }
        }
    }
    __syncthreads();
    // A_rm A_mm^-1 (:288-292); column k of A_rm: q-column k for the pose / speed-bias part, the coupling row of landmark k - md otherwise
    for (int w = t; w < n * m; w += nt) {
        const int i = w / m, j = w - i * m;
        double sacc = 0;
        for (int k = 0; k < md; k++) sacc += A[(md + i) * mq + k] * Einv[(size_t)k * m + j];
        for (int k0 = md; k0 < m; k0 += 16) {   // (sixteen terms' loads in flight; same order of the sum)
            double a[16], b[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const int k = min(k0 + u, m - 1); a[u] = Cl[(size_t)(k - md) * ldc + md + i]; b[u] = Einv[(size_t)k * m + j]; }
#pragma unroll
            for (int u = 0; u < 16; u++) if (k0 + u < m) sacc += a[u] * b[u];
        }
        ET1[w] = sacc;
    }
    __syncthreads();
    for (int w = t; w < n * n; w += nt) {   // A = Arr - Arm Amm_inv Amr
        const int i = w / n, j = w - i * n;
        double tt = A[(md + i) * mq + md + j];
        for (int k = 0; k < md; k++) tt -= ET1[(size_t)i * m + k] * A[k * mq + md + j];
        for (int k0 = md; k0 < m; k0 += 16) {
            double a[16], b[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const int k = min(k0 + u, m - 1); a[u] = ET1[(size_t)i * m + k]; b[u] = Cl[(size_t)(k - md) * ldc + md + j]; }
#pragma unroll
            for (int u = 0; u < 16; u++) if (k0 + u < m) tt -= a[u] * b[u];
        }
        Ar[w] = tt;
    }
    for (int i = t; i < n; i += nt) {       // b = brr - Arm Amm_inv bmm
        double sacc = b[md + i];
        for (int k = 0; k < md; k++) sacc -= ET1[(size_t)i * m + k] * b[k];
        for (int k = md; k < m; k++) sacc -= ET1[(size_t)i * m + k] * c.gl[k - md];
        br[i] = sacc;
    }
    __syncthreads();
    if (VIO_TIMERS && t == 0) be.dbg[8] = (int)(VIO_CLOCK() - tx0);
    // second eigen-decomposition and the prior (:293-315); A is free from here on (the fallback's eigenvector scratch)
    const int sw2 = marg_literal_prior(c, be, Ar, br, n, c.margW, A, sred, smem);
    if (VIO_TIMERS && t == 0) be.dbg[9] = (int)(VIO_CLOCK() - tx0);
    if (t == 0) { be.dbg[0] = sw1 * 100 + sw2; be.dbg[2] = second_new ? 1 : 0; be.dbg[4] = m; }
    if (VIO_TIMERS && t == 0) be.dbg[10] = (int)(VIO_CLOCK() - tx0);
}

template <bool EXACT> __device__ void marg_body(const Batch &B, int s, int *scratch, double *sred, unsigned char *smem_marg) {
    const int t = threadIdx.x, nt = blockDim.x;
    Ctx c = make_ctx(B, s);
    const DevCfg &C = *B.cfg;
    const vio_config &cfg = C.c;
    BeSeq &be = *c.be;
    if (!be.do_marg || be.rebooted || be.frame_count < c.W) return;
    const int W = c.W, W1 = W + 1, n = c.NPR;
    const double eps = 1e-8;
    __shared__ Params X;
    __shared__ double sdx[6 * VIO_MAXW + 16], srp[6 * VIO_MAXW + 16];
    __shared__ double cs[VIO_MAXW * 3 + 10], sn[VIO_MAXW * 3 + 10];
    __shared__ int pp[VIO_MAXW * 3 + 10], qq[VIO_MAXW * 3 + 10];
    __shared__ double A15[225], V15[225], Pinv[225];
    __shared__ int newpresent[VIO_MAXW + 3];
    const bool second_new = be.marginalization_flag != 0;
    if (second_new && !(be.has_prior && be.prior_present[W - 1])) return;
    // marg_exact: MarginalizationInfo::marginalize followed literally (marginalization_factor.cpp:281-315) -- the full m x m block of
    // pose 0, speed-bias 0 AND the landmarks that start in frame 0 goes through one truncated eigen-decomposition, and the new prior is
    // rebuilt from the truncated factors of the reduced system.  A parity instrument (one workgroup, Jacobi sweeps in HBM), not the hot path.
    const bool exact = EXACT && cfg.marg_exact == 1 && c.margE != nullptr;
    // marg_exact = 2 (round 5): the literal algorithm with its first eigen-decomposition replaced by a CERTIFIED inverse.  A_mm^+ of the
    // reference (:281-283) equals A_mm^-1 whenever no eigenvalue of A_mm is at or below the 1e-8 cut; this mode eliminates the marginalised
    // block exactly like the default one (landmarks analytically, the 15 x 15 / 6 x 6 rest by a Cholesky inverse) and PROVES per frame that
    // nothing could have been truncated: lambda_min(A_mm) >= 1 / |A_mm^-1|_F, bounded from the blocks of the inverse (below).  The second
    // half -- the factorisation of the new prior with its 1e-8 truncation, which does drop directions routinely -- is the literal one
    // (marg_literal_prior).  Frames whose certificate fails are counted (BeSeq::dbg[12]) and take the same route.
    const bool lit2 = EXACT && cfg.marg_exact == 2;
    int F0x = 0;   // landmarks in the marginalised block (exact mode)
    int F0k = 0;   // landmarks eliminated analytically (the other modes)
    PH_INIT;
    const long long tk0 = VIO_CLOCK();
    // vector2double
    if (t <= W) {
        int i = t;
        X.pose[i * 7 + 0] = be.Ps[i][0]; X.pose[i * 7 + 1] = be.Ps[i][1]; X.pose[i * 7 + 2] = be.Ps[i][2];
        quat q = R2q(ldm(be.Rs[i]));
        X.pose[i * 7 + 3] = q.x; X.pose[i * 7 + 4] = q.y; X.pose[i * 7 + 5] = q.z; X.pose[i * 7 + 6] = q.w;
        for (int k = 0; k < 3; k++) { X.sb[i * 9 + k] = be.Vs[i][k]; X.sb[i * 9 + 3 + k] = be.Bas[i][k]; X.sb[i * 9 + 6 + k] = be.Bgs[i][k]; }
    }
    if (t == W + 1) {
        X.ex[0] = be.tic[0]; X.ex[1] = be.tic[1]; X.ex[2] = be.tic[2];
        quat q = R2q(ldm(be.ric));
        X.ex[3] = q.x; X.ex[4] = q.y; X.ex[5] = q.z; X.ex[6] = q.w;
        X.td = be.td;
    }
    if (t < W + 3) newpresent[t] = 0;
    __syncthreads();
    if (t <= W) for (int k = 0; k < 7; k++) be.para_Pose[t][k] = X.pose[t * 7 + k];   // (the marginalisation's vector2double: see solve_epilogue)
    // reduced system over q = [m-block (md) | kept block (n)]
    const int md = second_new ? 6 : 15;
    const int mq = md + n;
    double *A = c.margA, *b = c.margB;
    for (int i = t; i < mq * mq; i += nt) A[i] = 0;
    for (int i = t; i < mq; i += nt) b[i] = 0;
    __syncthreads();
    // index maps (canonical kept layout: pose slots 0..W-1, sb, ex, td)
    const int rS = md + 6 * W, rE = md + 6 * W + 9, rT = md + 6 * W + 15;
    // prior
    if (be.has_prior) {
        prior_dx(c, X, sdx);
        // (round 6: the loops of this kernel that walked HBM one dependent load at a time -- a load behind a store the compiler must assume to
        //  alias, or a runtime-bound loop with the load inside -- now issue their loads in batches; same terms in the same order, same bits.
        //  At W = 20 the kernel took 1.1 ms, none of its phases more than 0.2.)
        for (int i = t; i < n; i += nt) {  // prior gradient at the current state: b + A dx (A is stored exactly symmetric: read by columns, coalesced)
            double sacc = c.prior_r[i];
            for (int j0 = 0; j0 < n; j0 += 32) {
                double hv[32];
#pragma unroll
                for (int u = 0; u < 32; u++) hv[u] = c.prior_H[(size_t)min(j0 + u, n - 1) * n + i];
#pragma unroll
                for (int u = 0; u < 32; u++) if (j0 + u < n) sacc += hv[u] * sdx[j0 + u];
            }
            srp[i] = sacc;
        }
        __syncthreads();
        auto pmap = [&](int a) -> int {
            if (a < 6 * W) {
                int k = a / 6, d = a - 6 * k;
                if (second_new) return k == W - 1 ? d : md + 6 * k + d;
                return k == 0 ? d : md + 6 * (k - 1) + d;
            }
            if (a < 6 * W + 9) return second_new ? rS + (a - 6 * W) : 6 + (a - 6 * W);
            if (a < 6 * W + 15) return rE + (a - 6 * W - 9);
            return rT;
        };
        for (int w0 = t; w0 < n * n; w0 += 8 * nt) {  // J^T J of the prior is kept next to J (prior_H, written when the prior was built); A is still zero here
            double pv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) pv[u] = c.prior_H[min(w0 + u * nt, n * n - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int w = w0 + u * nt;
                if (w < n * n) { const int a = w / n, bb = w - a * n; A[pmap(a) * mq + pmap(bb)] = 0.0 + pv[u]; }
            }
        }
        for (int a = t; a < n; a += nt) b[pmap(a)] += srp[a];
        if (t < W + 3) {
            const int pres = be.prior_present[t];
            if (t > W) { if (pres) newpresent[t] = 1; }
            else if (second_new) { if (t < W - 1 || t == W) newpresent[t] = pres; }
            else if (t >= 1 && t < W && pres) newpresent[t - 1] = 1;
        }
        __syncthreads();
    }
    PH(16);
    if (!second_new) {
        // IMU factor (0,1)
        PreInt &p1 = c.pre[be.pre_idx[1]];
        if (cfg.use_imu && p1.sum_dt < 10.0) {
            double *Jw = c.pairblk;         // 15x30 whitened
            double *raw = c.imu_raw;        // 15x31
            // five work types (whitened residual + four Jacobian column groups), each on its own wavefront, like evaluate()
            if ((t & 63) == 0) for (int part = t >> 6; part < 5; part += nt >> 6) {
                v3 G = ld3(be.g);
                if (part == 0) {
                    double r15[15];
                    bf::imu_raw_residual(p1, G, &X.pose[0], &X.sb[0], &X.pose[7], &X.sb[9], r15);
                    for (int r = 0; r < 15; r++) {
                        double sacc = 0;
                        for (int k = 0; k <= r; k++) sacc += p1.sqrt_info[r * 15 + k] * r15[k];
                        raw[r * 31 + 30] = sacc;
                    }
                } else
                    bf::imu_raw_jacobian_part(p1, G, &X.pose[0], &X.sb[0], &X.pose[7], &X.sb[9], part - 1, raw, 31);
            }
            __syncthreads();
            for (int w = t; w < 450; w += nt) {
                int r = w / 30, col = w - r * 30;
                double sacc = 0;
                for (int k = 0; k <= r; k++) sacc += p1.sqrt_info[r * 15 + k] * raw[k * 31 + col];
                Jw[w] = sacc;
            }
            __syncthreads();
            for (int w = t; w < 930; w += nt) {
                int a = w / 31, bb = w - a * 31;
                int ia = a < 15 ? a : (a < 21 ? md + (a - 15) : rS + (a - 21));
                double sacc = 0;
                if (bb < 30) {
                    int ib = bb < 15 ? bb : (bb < 21 ? md + (bb - 15) : rS + (bb - 21));
                    for (int k = 0; k < 15; k++) sacc += Jw[k * 30 + a] * Jw[k * 30 + bb];
                    A[ia * mq + ib] += sacc;
                } else {
                    for (int k = 0; k < 15; k++) sacc += Jw[k * 30 + a] * raw[k * 31 + 30];
                    b[ia] += sacc;
                }
            }
            if (t == 0) { newpresent[0] = 1; newpresent[W] = 1; }
            __syncthreads();
        }
        PH(17);
        // projection factors of landmarks first observed in frame 0; each landmark is eliminated on the fly:
        // contribution = J_q^T J_q - c c^T / d  with c = J_q^T J_l, d = J_l^T J_l (pseudo-inverse: dropped if d <= eps)
        int nlm = be.n_lm;
        int *flag = c.lm_tmp, *offs = c.res_k;
        for (int k = t; k < nlm; k += nt) { int slot = c.lm_order[k]; flag[k] = (in_problem(c, slot) && c.lm_start[slot] == 0) ? 1 : 0; }
        __syncthreads();
        int F0 = block_scan_flags(flag, nlm, offs, scratch);
        int *list0 = c.pair_list;
        for (int k = t; k < nlm; k += nt) if (flag[k]) list0[offs[k]] = c.lm_order[k];
        __syncthreads();
        // per landmark: evaluate residuals with loss correction, store J (2x20) r (2) per residual into c.res (cap checked)
        const int per = W;  // max residuals per landmark
        int F0c = min(F0, c.nres_cap / per);
        if (exact) { F0c = min(F0c, min(C.MX - 15, 480)); F0x = F0c; }
        F0k = F0c;
        // frame-pair form like the solver (be_factors.h eval_projection_pair): the geometry of the pairs (0, k) once, in LDS
        __shared__ double mgeo[(VIO_MAXW + 1) * 32 + 16];
        if (t >= 1 && t <= W) {
            bf::PairGeo g;
            bf::pair_geo(&X.pose[0], &X.pose[t * 7], X.ex, g);
            double *o = mgeo + (size_t)(t - 1) * 32;
            for (int q = 0; q < 9; q++) { o[q] = g.A1[q]; o[9 + q] = g.A2[q]; o[18 + q] = g.M[q]; }
            o[27] = g.t[0]; o[28] = g.t[1]; o[29] = g.t[2];
        }
        if (t == 0) stm(mgeo + (size_t)W * 32, q2R(mkq(X.ex[6], X.ex[3], X.ex[4], X.ex[5])));
        __syncthreads();
        for (int w = t; w < F0c * per; w += nt) {
            int li = w / per, k = w - li * per + 1;
            int slot = list0[li];
            double *out = c.res + (size_t)w * 42;
            if (k >= c.lm_nobs[slot]) { out[40] = 0; out[41] = 0; for (int q = 0; q < 40; q++) out[q] = 0; continue; }
            // CauchyLoss(1.0): rho'' < 0, so Ceres' corrector reduces to the scaling sqrt(rho') of residual and Jacobian (corrector.cc)
            double rr[2], wgt = 1.0;
            const double inv_dep = 1.0 / c.lm_depth[slot];
            const bf::PairGeo &g = *(const bf::PairGeo *)(mgeo + (size_t)(k - 1) * 32);
            bf::eval_projection_pair(cfg, g, mgeo + (size_t)W * 32, X.ex, inv_dep, X.td, obs_ptr(c, slot, 0), obs_ptr(c, slot, k), cfg.estimate_td != 0, rr, out,
                                     true, &wgt);
            out[40] = wgt * rr[0];
            out[41] = wgt * rr[1];
        }
        __syncthreads();
        PH(18);
        // column map of a residual's 19 non-landmark columns into q: pose0 -> 0..5, pose_k -> md+6(k-1), ex, td
        // accumulate A_qq and b_q: thread per (a,b) over the union index set is irregular; use per-landmark dense rows:
        // c_l (mq), d_l, b_l, then A -= c c^T/d after adding the plain J^T J per residual.
        double *Cl = c.Hpl;  // F0c x LW' rows (reuse): needs mq + 2 <= LW  (mq = 15 + 6W + 16 = 6W+31 <= 15W+22 = P)
        const int ldc = c.LW;
        for (int w = t; w < F0c * ldc; w += nt) Cl[w] = 0;
        __syncthreads();
        // per landmark: c_l = J_q^T J_l over its residuals, d_l = J_l^T J_l, b_l = J_l^T r.  Sixteen lanes per landmark, lane = residual
        // (frame k = lane + 1): the columns of the lane's own frame are written directly, the columns every residual shares
        // (oldest pose, extrinsic, td) and d_l / b_l are reduced over the sixteen lanes.
        {
            const int grp = t >> 4, gln = t & 15, ngrp = nt >> 4;
            for (int li0 = 0; li0 < F0c; li0 += ngrp) {
                const int li = li0 + grp;
                const bool act = li < F0c;
                const int no = act ? c.lm_nobs[list0[li]] : 0;
                double cm[13], dsum = 0, bsum = 0;
#pragma unroll
                for (int q = 0; q < 13; q++) cm[q] = 0;
                for (int k = gln + 1; k < no; k += 16) {
                    const double *Jg = c.res + (size_t)(li * per + k - 1) * 42;
                    double Jr[42];   // the whole record first: the stores to Cl below would otherwise sit between its loads
#pragma unroll
                    for (int q = 0; q < 42; q++) Jr[q] = Jg[q];
                    const double jl0 = Jr[19], jl1 = Jr[39];
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        cm[q] += Jr[q] * jl0 + Jr[20 + q] * jl1;
                        cm[6 + q] += Jr[12 + q] * jl0 + Jr[32 + q] * jl1;
                        Cl[(size_t)li * ldc + md + 6 * (k - 1) + q] = Jr[6 + q] * jl0 + Jr[26 + q] * jl1;
                    }
                    if (cfg.estimate_td) cm[12] += Jr[18] * jl0 + Jr[38] * jl1;
                    dsum += jl0 * jl0 + jl1 * jl1;
                    bsum += jl0 * Jr[40] + jl1 * Jr[41];
                }
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) {
#pragma unroll
                    for (int q = 0; q < 13; q++) cm[q] += __shfl_xor(cm[q], off, 16);
                    dsum += __shfl_xor(dsum, off, 16);
                    bsum += __shfl_xor(bsum, off, 16);
                }
                if (act && gln == 0) {
#pragma unroll
                    for (int q = 0; q < 6; q++) { Cl[(size_t)li * ldc + q] = cm[q]; Cl[(size_t)li * ldc + rE + q] = cm[6 + q]; }
                    if (cfg.estimate_td) Cl[(size_t)li * ldc + rT] = cm[12];
                    c.Hll[li] = dsum;
                    c.gl[li] = bsum;
                }
            }
        }
        __syncthreads();
        PH(19);
        // frame blocks G_j = sum over the residuals observing frame j of [J19 r]^T [J19 r] (packed symmetric 20x20) on the FP64
        // matrix cores: one wavefront per frame, K = 2 rows per landmark (rows of landmarks that do not see frame j are zero)
        {
            const int lane = t & 63, wave = t >> 6, nw = nt >> 6, li = lane & 15, lk = lane >> 4;
            for (int j = 1 + wave; j <= W; j += nw) {
                v4f64 a00 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
                const int K = 2 * F0c;
                for (int k0 = 0; k0 < K; k0 += 64) {   // four 16-row trips' loads in flight at once, the matrix-core steps in the order of the one-trip loop
                    double x0[16], x1[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        int kk = k0 + 4 * u + lk;
                        bool valid = kk < K;
                        int lm = min(kk, K - 1) >> 1, sub = kk & 1;
                        const double *Jr = c.res + (size_t)(lm * per + j - 1) * 42;
                        double v0 = Jr[sub * 20 + li];
                        double v1 = Jr[li < 3 ? sub * 20 + 16 + li : 40 + sub];
                        x0[u] = valid ? v0 : 0.0;
                        x1[u] = (valid && li < 4) ? v1 : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        if (k0 + 16 * (u >> 2) >= K) break;
                        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u], x0[u], a00, 0, 0, 0);
                        a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u], x0[u], a10, 0, 0, 0);
                        a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u], x1[u], a11, 0, 0, 0);
                    }
                }
                double *out = c.pairblk + (size_t)(j - 1) * 210;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int row = lk + 4 * r, col = li;
                    if (col <= row) out[sym_idx(col, row)] = a00[r];
                    if (row < 4) out[sym_idx(col, 16 + row)] = a10[r];
                    if (row < 4 && col < 4 && col <= row) out[sym_idx(16 + col, 16 + row)] = a11[r];
                }
            }
        }
        if (!exact) for (int li = t; li < F0c; li += nt) { double d = c.Hll[li]; c.Hll[li] = d > eps ? 1.0 / d : 0.0; }  // pseudo-inverse of the diagonal block
        __syncthreads();
        PH(24);
        // A_qq += sum_j G_j (scattered) - C^T D^+ C ; b_q likewise.  Per q-column (and the right-hand side, index mq): the frame whose
        // residuals carry it (0 = every residual: oldest pose, extrinsic, td, rhs; -1 = none) and its local column in a residual record
        __shared__ signed char qfr[6 * VIO_MAXW + 40], qlc[6 * VIO_MAXW + 40];
        __shared__ double sub_part[4][6 * VIO_MAXW + 40];
        for (int a = t; a <= mq; a += nt) {
            int fr = -1, lc = 0;
            if (a == mq) { fr = 0; lc = 19; }
            else if (a < 6) { fr = 0; lc = a; }
            else if (a >= md && a < md + 6 * W) { fr = (a - md) / 6 + 1; lc = 6 + (a - md) % 6; }
            else if (a >= rE && a < rE + 6) { fr = 0; lc = 12 + (a - rE); }
            else if (a == rT && cfg.estimate_td) { fr = 0; lc = 18; }
            qfr[a] = (signed char)fr; qlc[a] = (signed char)lc;
        }
        // b_q -= C^T D^+ b_l: four interleaved partial sums per column, added in a fixed order
        for (int w = t; w < 4 * mq; w += nt) {
            const int ch = w / mq, a = w - ch * mq;
            double acc = 0;
            if (!exact) for (int li = ch; li < F0c; li += 4) acc += Cl[(size_t)li * ldc + a] * (c.gl[li] * c.Hll[li]);
            sub_part[ch][a] = acc;
        }
        __syncthreads();
        for (int w0 = t; w0 < mq * (mq + 1); w0 += 4 * nt) {   // four entries per thread and trip: their gathers and the entries of A they add to in flight together
            bool use[4], multi[4], rhs[4];
            int dst[4], row[4], sidx[4];
            double pv[4], av[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int w = min(w0 + u * nt, mq * (mq + 1) - 1);
                const int a = w / (mq + 1), bb = w - a * (mq + 1);
                const int fa = qfr[a], fb = qfr[bb];
                const int la = qlc[a], lb = qlc[bb];
                use[u] = w0 + u * nt < mq * (mq + 1) && fa >= 0 && fb >= 0 && !(fa > 0 && fb > 0 && fa != fb);
                multi[u] = use[u] && fa == 0 && fb == 0;
                rhs[u] = bb == mq; row[u] = a; dst[u] = a * mq + bb;
                sidx[u] = sym_idx(la, lb);
                const int f1 = fa > 0 ? fa : (fb > 0 ? fb : 1);
                pv[u] = use[u] ? c.pairblk[(size_t)(f1 - 1) * 210 + sidx[u]] : 0.0;
                av[u] = use[u] ? (rhs[u] ? b[a] : A[dst[u]]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!use[u]) continue;
                double sacc = pv[u];
                if (multi[u]) {   // a column every residual carries against another such: one term per frame, all loads first
                    double fv[VIO_MAXW];
#pragma unroll
                    for (int j = 0; j < VIO_MAXW; j++) fv[j] = c.pairblk[(size_t)min(j, W - 1) * 210 + sidx[u]];
                    sacc = 0;
#pragma unroll
                    for (int j = 0; j < VIO_MAXW; j++) if (j < W) sacc += fv[j];
                }
                if (!rhs[u]) A[dst[u]] = av[u] + sacc;
                else b[row[u]] = av[u] + (sacc - ((sub_part[0][row[u]] + sub_part[1][row[u]]) + (sub_part[2][row[u]] + sub_part[3][row[u]])));
            }
        }
        __syncthreads();
        PH(25);
        // A_qq -= C^T D^+ C : rank-F0c update on the FP64 matrix cores, lower 16x16 tiles mirrored into the upper triangle
        if (!exact) {
            const int lane = t & 63, wave = t >> 6, nw = nt >> 6, li = lane & 15, lk = lane >> 4;
            const int nbq = (mq + 15) >> 4, ntile = nbq * (nbq + 1) / 2;
            for (int tile = wave; tile < ntile; tile += nw) {
                int ti, tj;
                tri_decode(tile, ti, tj);
                v4f64 acc = {0, 0, 0, 0};
                const int ca = min(16 * ti + li, ldc - 1), cb = min(16 * tj + li, ldc - 1);
                for (int k0 = 0; k0 < F0c; k0 += 64) {   // (four trips' loads in flight, as above)
                    double xa[16], xb[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        int kk = k0 + 4 * u + lk;
                        bool valid = kk < F0c;
                        int kc = min(kk, F0c - 1);
                        double va = Cl[(size_t)kc * ldc + ca], vb = Cl[(size_t)kc * ldc + cb], dinv = c.Hll[kc];
                        xa[u] = valid ? -(va * dinv) : 0.0;
                        xb[u] = valid ? vb : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        if (k0 + 16 * (u >> 2) >= F0c) break;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u], xb[u], acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int row = 16 * ti + lk + 4 * r, col = 16 * tj + li;
                    if (row < mq && col < mq && (ti != tj || col <= row)) {
                        A[row * mq + col] += acc[r];
                        if (row != col) A[col * mq + row] += acc[r];
                    }
                }
            }
        }
        for (int li = t; li < F0c; li += nt) {   // every writer stores the same value
            int no = c.lm_nobs[list0[li]];
            for (int k = 1; k < no; k++) newpresent[k - 1] = 1;
            if (no > 1) { newpresent[W + 1] = 1; if (cfg.estimate_td) newpresent[W + 2] = 1; }
        }
        __syncthreads();
    }
    if (VIO_TIMERS && s == 0 && t == 0 && !second_new) B.timings[31] += 1.0f;
    PH(20);
    if (exact) marg_exact_finish(c, be, A, b, md, mq, n, c.Hpl, c.LW, F0x, second_new, sred, smem_marg);
    else {
    // ---- eliminate the m-block (md x md) with a truncated eigen-decomposition
    for (int w = t; w < md * md; w += nt) { int i = w / md, j = w - i * md; A15[w] = 0.5 * (A[i * mq + j] + A[j * mq + i]); }
    __syncthreads();
    // Pseudo-inverse with the eigenvalues <= 1e-8 dropped.  Fast path: when a Cholesky factorisation proves every eigenvalue above 1e-6
    // (lambda_min >= 1 / |A^-1|_F) nothing is dropped and the pseudo-inverse is the inverse, a few microseconds of one wavefront instead
    // of ~100 Jacobi rounds; otherwise the eigen-decomposition decides.
    __shared__ int pinv_direct;
    __shared__ double L15[225];
    if (t < 64) {
        const bool okc = spd_inverse_wave16(A15, md, 1e-6, L15, V15, Pinv);
        if (t == 0) pinv_direct = okc ? 1 : 0;
    }
    __syncthreads();
    if (!pinv_direct) {
        if (t < 64) jacobi_wave16(A15, V15, md, md, cs, sn, pp, qq);   // md <= 15: one wavefront, no workgroup barriers inside
        __syncthreads();
        for (int w = t; w < md * md; w += nt) {
            int i = w / md, j = w - i * md;
            double sacc = 0;
            for (int k = 0; k < md; k++) { double ev = A15[k * md + k]; if (ev > eps) sacc += V15[i * md + k] * V15[j * md + k] / ev; }
            Pinv[w] = sacc;
        }
        __syncthreads();
    }
    if (VIO_TIMERS && s == 0 && t == 0 && pinv_direct) B.timings[26] += 1.0f;
    PH(36);
    // T1 = A_rm A_mm^+ (n x md) and the block A_mr (md x n) staged in LDS (the tile region, free until the constant term is formed): the
    // n^2 entries of A_r = A_rr - T1 A_mr read each of them n times
    double *T1 = (double *)smem_marg, *Amr = T1 + (size_t)n * md;
    for (int w = t; w < n * md; w += nt) {
        int i = w / md, j = w - i * md;
        double sacc = 0;
        for (int k = 0; k < md; k++) sacc += A[(md + i) * mq + k] * Pinv[k * md + j];
        T1[w] = sacc;
        const int k2 = w / n, j2 = w - k2 * n;      // the same index range covers A_mr (md x n)
        Amr[w] = A[k2 * mq + md + j2];
    }
    __syncthreads();
    PH(37);
    double *Ar = c.margV;             // reuse as A_r first (n x n), eigenvectors go to margW+...
    double *br = c.vec;               // n
    for (int w0 = t; w0 < n * n; w0 += 8 * nt) {
        double av[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int w = min(w0 + u * nt, n * n - 1), i = w / n, j = w - i * n; av[u] = A[(md + i) * mq + md + j]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int w = w0 + u * nt;
            if (w >= n * n) break;
            const int i = w / n, j = w - i * n;
            double tt = av[u];
            for (int k = 0; k < md; k++) tt -= T1[i * md + k] * Amr[k * n + j];
            Ar[w] = tt;
        }
    }
    for (int i = t; i < n; i += nt) {
        double sacc = b[md + i];
        for (int k = 0; k < md; k++) sacc -= T1[i * md + k] * b[k];
        br[i] = sacc;
    }
    __syncthreads();
    PH(21);
    if (lit2) {
        // certificate: |A_mm^-1|_F <= |S^-1|_F + 2 |T|_F + |D^-1|_F + |D^-1 B|_F |T|_F with A_mm = [[P, B^T], [B, D]], S = P - B^T D^-1 B (the
        // block A15 above), T = S^-1 B^T D^-1; it needs D invertible (every landmark's d > eps) and the direct inverse of S
        __shared__ int cert_bad;
        if (t == 0) cert_bad = pinv_direct ? 0 : 1;
        __syncthreads();
        double n_t = 0, n_d = 0, n_db = 0;
        if (!second_new) {
            const double *Cl = c.Hpl;
            const int ldc = c.LW;
            for (int li = t; li < F0k; li += nt) {
                const double dinv = c.Hll[li];           // 1 / d, or 0 where d <= eps (pseudo-inverse of the diagonal block)
                if (!(dinv > 0.0)) cert_bad = 1;         // (benign race: every writer stores 1)
                n_d += dinv * dinv;
                for (int a = 0; a < md; a++) {
                    double tt = 0;
                    for (int bq = 0; bq < md; bq++) tt += Pinv[a * md + bq] * Cl[(size_t)li * ldc + bq];
                    tt *= dinv;
                    n_t += tt * tt;
                    const double db = dinv * Cl[(size_t)li * ldc + a];
                    n_db += db * db;
                }
            }
        }
        block_sum3(n_t, n_d, n_db, sred);
        double n_s = 0;
        for (int w = t; w < md * md; w += nt) n_s += Pinv[w] * Pinv[w];
        n_s = block_sum(n_s, sred);
        __syncthreads();
        if (t == 0) {
            const double bound = sqrt(n_s) + 2.0 * sqrt(n_t) + sqrt(n_d) + sqrt(n_db) * sqrt(n_t);
            const bool certified = !cert_bad && isfinite(bound) && bound > 0.0 && 1.0 / bound > 1e-7;   // ten times the cut
            be.dbg[11] = certified ? 1 : 0;
            if (!certified) be.dbg[12] += 1;
        }
        __syncthreads();
        const int sw2 = marg_literal_prior(c, be, Ar, br, n, c.margW, A, sred, smem_marg);
        if (t == 0) { be.dbg[0] = sw2; be.dbg[2] = second_new ? 1 : 0; be.dbg[4] = md; }
    } else {
    // ---- the new prior, kept as the quadratic form the solver consumes: A = sym(A_r), b = b_r, c0 = b^T A^+ b.
    // The reference factors A = V S V^T, drops eigenvalues <= 1e-8 and stores J = S^1/2 V^T, r = S^-1/2 V^T b
    // (marginalization_factor.cpp:293-315); J^T J, J^T r and |r|^2 are all the solver and the next marginalisation ever use, and
    // they equal (A, b, c0) up to the dropped directions: measured on the canonical workload b has a 1e-12 relative component
    // there, A changes by < 1e-8 absolute (1e-16 relative) and the weakly observed directions contribute < 1e-5 of c0
    // (DESIGN.md deviation 13).  The factored form is produced on demand by be_prior_factor_kernel (vio_get_prior).
    for (int w0 = t; w0 < n * n; w0 += 8 * nt) {
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int w = min(w0 + u * nt, n * n - 1), i = w / n, j = w - i * n; x[u] = Ar[i * n + j]; y[u] = Ar[j * n + i]; }
#pragma unroll
        for (int u = 0; u < 8; u++) if (w0 + u * nt < n * n) c.prior_H[w0 + u * nt] = 0.5 * (x[u] + y[u]);
    }
    for (int i = t; i < n; i += nt) c.prior_r[i] = br[i];
    PH(38);
    {
        // c0 = |L^-1 b|^2 with L L^T = A + delta I on 16x16 LDS tiles (delta lifts the gauge directions off the round-off floor)
        __shared__ __attribute__((aligned(16))) double cq_x[EIG_LD + 16];   // read and written in 16-byte pairs by chol_tiles
        __shared__ double cq_dinv[EIG_LD + 16];
        __shared__ int cq_flag;
        const int nbq = (n + 15) >> 4;
        double *T = (double *)smem_marg;
        double dmax = 0;
        for (int i = t; i < n; i += nt) dmax = fmax(dmax, fabs(Ar[i * n + i]));
        dmax = block_max(dmax, sred);
        const double delta = 64.0 * 2.220446049250313e-16 * (double)n * dmax;
        const int ntl = nbq * (nbq + 1) / 2 * 256;
        for (int w0 = t; w0 < ntl; w0 += 4 * nt) {
            double x[4], y[4];
            int ii[4], jj[4], dd[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int w = min(w0 + u * nt, ntl - 1);
                int tile = w >> 8, e = w & 255, r = e >> 4, cc = e & 15, ti, tj;
                tri_decode(tile, ti, tj);
                const int i = 16 * ti + r, j = 16 * tj + cc, ic = min(i, n - 1), jc = min(j, n - 1);
                ii[u] = i; jj[u] = j; dd[u] = tl_idx(ti, tj, r, cc);
                x[u] = Ar[ic * n + jc]; y[u] = Ar[jc * n + ic];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (w0 + u * nt >= ntl) break;
                const int i = ii[u], j = jj[u];
                T[dd[u]] = (i < n && j < n) ? 0.5 * (x[u] + y[u]) + (i == j ? delta : 0.0) : (i == j ? 1.0 : 0.0);
            }
        }
        for (int i = t; i < 16 * nbq; i += nt) cq_x[i] = i < n ? br[i] : 0.0;
        __syncthreads();
        PH(39);
        double c0 = 0;
        if (dmax > 0 && chol_tiles(T, nbq, &cq_flag, cq_dinv, nullptr, cq_x)) {   // c0 needs the forward substitution only
            double acc = 0;
            for (int i = t; i < n; i += nt) acc += cq_x[i] * cq_x[i];
            c0 = block_sum(acc, sred);
            if (!isfinite(c0)) c0 = 0;
        }
        __syncthreads();
        if (t == 0) { be.prior_c0 = c0; be.dbg[0] = 0; be.dbg[2] = second_new ? 1 : 0; }
    }
    }
    }
    PH(22);
    // keep_block_data in the shifted (canonical) layout
    if (t < W) {
        int src = second_new ? (t == W - 1 ? W : t) : t + 1;
        for (int d = 0; d < 7; d++) c.prior_x0[t * 7 + d] = X.pose[src * 7 + d];
    }
    if (t == W) for (int d = 0; d < 9; d++) c.prior_x0[W * 7 + d] = X.sb[(second_new ? 0 : 1) * 9 + d];
    if (t == W + 1) { for (int d = 0; d < 7; d++) c.prior_x0[W * 7 + 9 + d] = X.ex[d]; c.prior_x0[W * 7 + 16] = X.td; }
    __syncthreads();
    if (t < W + 3) be.prior_present[t] = newpresent[t];
    if (t == 0) { be.has_prior = 1; be.dbg[3] = (int)(VIO_CLOCK() - tk0); }
    PH(23);
}

// Estimator::setReloFrame (estimator.cpp:1728-1747) for sequence seq; the match points are already in B.relo_mp.  par = stamp, index,
// n, relo_t(3), relo_r(9).  relo_Pose is copied from para_Pose[i] AS THE LAST optimization() LEFT IT, like upstream: after a MARGIN_OLD
// slide that array still holds the pre-slide window (the copy is then the pose of the frame one slot older), an upstream quirk that the
// first iterations of the relocalisation solve absorb.
__global__ void be_set_relo_kernel(Batch B, int seq, const double *par) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    BeSeq &be = B.be[seq];
    const int W = B.cfg->W;
    be.relo_stamp = par[0];
    be.relo_index = (int)par[1];
    be.relo_nmatch = (int)par[2];
    for (int k = 0; k < 3; k++) be.prev_relo_t[k] = par[3 + k];
    for (int k = 0; k < 9; k++) be.prev_relo_r[k] = par[6 + k];
    for (int i = 0; i < W; i++)
        if (be.relo_stamp == be.Headers[i]) {
            be.relo_local = i;
            be.relo_info = 1;
            for (int j = 0; j < 7; j++) be.relo_Pose[j] = be.para_Pose[i][j];
        }
}

// ====================================================================================================== prior factorisation
// linearized_jacobians / linearized_residuals of the current prior (marginalization_factor.cpp:293-315), on demand: eigen-
// decomposition of prior_H with the 1e-8 cut-off, J = S^1/2 V^T -> prior_J, r = S^-1/2 V^T b -> prior_rf.
__global__ __launch_bounds__(512) void be_prior_factor_kernel(Batch B, int seq) {
    const int s = seq, t = threadIdx.x, nt = blockDim.x;
    __shared__ double sred[64];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_marg[];
    Ctx c = make_ctx(B, s);
    BeSeq &be = *c.be;
    if (!be.has_prior) return;
    const int n = c.NPR;
    const double eps = 1e-8;
    const double *br = c.prior_r;
    const long long tj0 = VIO_CLOCK();
    // symmetrised A_r and its eigenvectors live in LDS when they fit (n <= 96), else in HBM scratch
    const bool in_lds = n <= 96;
    const int ldj = in_lds ? (n | 1) : n;  // odd leading dimension: conflict-free 64-bit LDS column access
    double *As = in_lds ? (double *)smem_marg : c.margA;
    double *Vv = in_lds ? (double *)smem_marg + n * ldj : c.margW + (size_t)n * 16;
    for (int w = t; w < n * n; w += nt) { int i = w / n, j = w - i * n; As[i * ldj + j] = c.prior_H[w]; }
    __syncthreads();
    __shared__ double ev_d[6 * VIO_MAXW + 16], ev_e[6 * VIO_MAXW + 16], ev_g[6 * VIO_MAXW + 16];
    __shared__ double ev_part[8 * EIG_LD];
    // two call sites so that the eigen-solver is specialised for the address space of the matrix (ds_* for LDS, global_* for HBM)
    // instead of falling back to flat loads on a pointer that could be either
    if (in_lds) {
        double *Al = (double *)smem_marg;
        if ((nt >> 6) <= 8) sym_eig_tridiag_mt(Al, n, ldj, ev_d, ev_e, ev_g, ev_part);
        else sym_eig_tridiag(Al, n, ldj, ev_d, ev_e, ev_g, sred);
        if (t == 0) be.dbg[5] = (int)(VIO_CLOCK() - tj0);
        tridiag_ql_wave(Al, n, ldj, ev_d, ev_e);
    } else {
        double *Ag = c.margA;
        if ((nt >> 6) <= 8) sym_eig_tridiag_mt(Ag, n, ldj, ev_d, ev_e, ev_g, ev_part);
        else sym_eig_tridiag(Ag, n, ldj, ev_d, ev_e, ev_g, sred);
        if (t == 0) be.dbg[5] = (int)(VIO_CLOCK() - tj0);
        tridiag_ql_wave(Ag, n, ldj, ev_d, ev_e);
    }
    if (t == 0) be.dbg[6] = (int)(VIO_CLOCK() - tj0);
    Vv = As;  // eigenvectors overwrite the matrix
    if (t == 0) be.dbg[1] = (int)(VIO_CLOCK() - tj0);
    // linearized_jacobians = sqrt(S) V^T ; linearized_residuals = S^-1/2 V^T b
    for (int w = t; w < n * n; w += nt) {
        int k = w / n, i = w - k * n;
        double wv = ev_d[k];
        double S = wv > eps ? wv : 0.0;
        c.prior_J[w] = sqrt(S) * Vv[i * ldj + k];
    }
    for (int k = t; k < n; k += nt) {
        double wv = ev_d[k];
        double Sinv = wv > eps ? 1.0 / wv : 0.0;
        double vb = 0;
        for (int i = 0; i < n; i++) vb += Vv[i * ldj + k] * br[i];
        c.prior_rf[k] = sqrt(Sinv) * vb;
    }
}

// ====================================================================================================== be_finish
__device__ void finish_body(const Batch &B, int s, int *scratch, PreWork &pw, unsigned char *smem_marg) {   // smem_marg: the marginalisation's dynamic LDS, free again (>= PREINT_MANY_LDS_DOUBLES doubles)
    const int t = threadIdx.x, nt = blockDim.x;
    Ctx c = make_ctx(B, s);
    const DevCfg &C = *B.cfg;
    const vio_config &cfg = C.c;
    BeSeq &be = *c.be;
    const int W = c.W, W1 = W + 1;
    __shared__ int sh_i[4];
    double *od = B.odom + (size_t)s * 11;
    if (!be.processed || be.rebooted) return;
    PH_INIT;
    int nlm = be.n_lm;
    int *flag = c.lm_pidx, *offs = c.lm_aidx;  // free after the solve
    const int fc = be.frame_count;
    const int sflag0 = be.solver_flag;
    __syncthreads();
    // dynamic initialisation (static_init: 0, estimator.cpp:230-259): while INITIAL the frame counter just advances; with a full
    // window and no (successful) attempt the window slides with INITIAL semantics (removeBack, no depth transfer)
    const bool dyn_initial = sflag0 == 0 && cfg.dynamic_init;
    if (dyn_initial) {
        if (fc < W) {
            if (t == 0) be.frame_count = fc + 1;
            return;
        }
    } else if (sflag0 == 0) {
        if (fc == W && be.do_solve) {
            if (t == 0) be.solver_flag = 1;
            __syncthreads();
        } else {
            // frame_count < WINDOW_SIZE: copy the state forward (estimator.cpp:306-315)
            if (t == 0 && fc < W) {
                int n = fc + 1;
                for (int k = 0; k < 3; k++) { be.Ps[n][k] = be.Ps[fc][k]; be.Vs[n][k] = be.Vs[fc][k]; be.Bas[n][k] = be.Bas[fc][k]; be.Bgs[n][k] = be.Bgs[fc][k]; }
                for (int k = 0; k < 9; k++) be.Rs[n][k] = be.Rs[fc][k];
                be.frame_count = n;
            }
            return;
        }
    } else if (!be.init_frame) {
        // movingConsistencyCheck (estimator.cpp:1965-2009); not on the frame that completed the dynamic initialisation (:243-251)
        m3 ric = ldm(be.ric);
        v3 tic = ld3(be.tic);
        for (int k = t; k < nlm; k += nt) {
            int slot = c.lm_order[k];
            if (!(c.lm_nobs[slot] >= 2 && c.lm_start[slot] < W - 2)) continue;
            double depth = c.lm_depth[slot];
            if (depth < 0) continue;
            int imu_i = c.lm_start[slot], no = c.lm_nobs[slot];
            const double *oi = obs_ptr(c, slot, imu_i);
            v3 pts_i = mk(oi[0], oi[1], oi[2]);
            double err = 0, err3 = 0;
            int cnt = 0;
            for (int q = 1; q < no; q++) {
                int imu_j = imu_i + q;
                const double *oj = obs_ptr(c, slot, imu_j);
                v3 pts_j = mk(oj[0], oj[1], oj[2]);
                v3 pts_w = add(mul(ldm(be.Rs[imu_i]), add(mul(ric, scl(depth, pts_i)), tic)), ld3(be.Ps[imu_i]));
                v3 pts_cj = mul(tr(ric), sub(mul(tr(ldm(be.Rs[imu_j])), sub(pts_w, ld3(be.Ps[imu_j]))), tic));
                double rx = pts_cj.x / pts_cj.z - pts_j.x, ry = pts_cj.y / pts_cj.z - pts_j.y;
                err += sqrt(rx * rx + ry * ry);
                err3 += nrm(sub(pts_cj, pts_j)) / depth;
                cnt++;
            }
            if (cnt > 0) c.lm_dyn[slot] = (cfg.focal_length * err / cnt > 10 || err3 / cnt > 2.0) ? 1 : 0;
        }
        __syncthreads();
        // failureDetection (estimator.cpp:1113-1159) ran at the end of be_solve (see there); a sequence that rebooted never gets here
    }
    PH(27);
    // ---- slideWindow (estimator.cpp:1580-1689)
    if (be.marginalization_flag == 0) {
        if (t == 0) {
            for (int k = 0; k < 9; k++) be.back_R0[k] = be.Rs[0][k];
            for (int k = 0; k < 3; k++) be.back_P0[k] = be.Ps[0][k];
            int first = be.pre_idx[0];
            for (int i = 0; i < W; i++) {
                be.Headers[i] = be.Headers[i + 1];
                for (int k = 0; k < 3; k++) { be.Ps[i][k] = be.Ps[i + 1][k]; be.Vs[i][k] = be.Vs[i + 1][k]; be.Bas[i][k] = be.Bas[i + 1][k]; be.Bgs[i][k] = be.Bgs[i + 1][k]; }
                for (int k = 0; k < 9; k++) be.Rs[i][k] = be.Rs[i + 1][k];
                be.pre_idx[i] = be.pre_idx[i + 1];
            }
            be.pre_idx[W] = first;
            // slot W keeps the newest state; its pre-integration restarts from acc_0 / gyr_0 (:1614-1630)
            bf::preint_init(c.pre[first], ld3(be.acc_0), ld3(be.gyr_0), ld3(be.Bas[W]), ld3(be.Bgs[W]));
        }
        __syncthreads();
        PH(28);
        // slideWindowOld -> removeBackShiftDepth (feature_manager.cpp:660-691); solver_flag is NON_LINEAR here
        m3 R0 = mul(ldm(be.back_R0), ldm(be.ric)), R1 = mul(ldm(be.Rs[0]), ldm(be.ric));
        v3 P0 = add(ld3(be.back_P0), mul(ldm(be.back_R0), ld3(be.tic))), P1 = add(ld3(be.Ps[0]), mul(ldm(be.Rs[0]), ld3(be.tic)));
        for (int k = t; k < nlm; k += nt) {
            int slot = c.lm_order[k];
            int keep = 1;
            if (c.lm_start[slot] != 0) c.lm_start[slot]--;
            else {
                const double *o0 = obs_ptr(c, slot, 0);
                v3 uv_i = mk(o0[0], o0[1], o0[2]);
                int no = c.lm_nobs[slot] - 1;
                c.lm_nobs[slot] = no;
                if (dyn_initial) keep = no > 0;     // FeatureManager::removeBack (feature_manager.cpp:693-708): solver_flag == INITIAL
                else if (no < 2) keep = 0;
                else {
                    v3 pts_i = scl(c.lm_depth[slot], uv_i);
                    v3 w_pts_i = add(mul(R0, pts_i), P0);
                    v3 pts_j = mul(tr(R1), sub(w_pts_i, P1));
                    c.lm_depth[slot] = pts_j.z > 0 ? pts_j.z : cfg.init_depth;
                }
            }
            flag[k] = keep;
        }
        __syncthreads();
        if (t == 0) be.ring_base = (be.ring_base + 1) % W1;  // frame f becomes frame f-1 without moving observations
        __syncthreads();
        lm_compact(c, flag, offs, scratch);
    } else {
        // MARGIN_SECOND_NEW: merge the IMU samples of frame W into W-1 (:1651-1687)
        PreInt &dst = c.pre[be.pre_idx[W - 1]];
        PreInt &src = c.pre[be.pre_idx[W]];
        if (t == 0) {
            be.Headers[W - 1] = be.Headers[W];
            for (int k = 0; k < 3; k++) { be.Ps[W - 1][k] = be.Ps[W][k]; be.Vs[W - 1][k] = be.Vs[W][k]; be.Bas[W - 1][k] = be.Bas[W][k]; be.Bgs[W - 1][k] = be.Bgs[W][k]; }
            for (int k = 0; k < 9; k++) be.Rs[W - 1][k] = be.Rs[W][k];
        }
        preint_load(dst, pw);
        preint_propagate_many(dst, pw, cfg, src.n_buf, src.dt_buf, src.acc_buf, src.gyr_buf, true, (double *)smem_marg);
        preint_store(dst, pw);
        if (t == 0) bf::preint_init(src, ld3(be.acc_0), ld3(be.gyr_0), ld3(be.Bas[W]), ld3(be.Bgs[W]));
        __syncthreads();
        // slideWindowNew -> removeFront(frame_count) (feature_manager.cpp:710-730)
        for (int k = t; k < nlm; k += nt) {
            int slot = c.lm_order[k];
            int keep = 1;
            int st = c.lm_start[slot], no = c.lm_nobs[slot];
            if (st == W) {
                double *d = obs_ptr(c, slot, W - 1);
                const double *sr = obs_ptr(c, slot, W);
                for (int q = 0; q < VIO_OBS_D; q++) d[q] = sr[q];
                c.lm_start[slot] = st - 1;
            } else {
                int endf = st + no - 1;
                if (endf >= W - 1) {
                    if (endf == W) {
                        double *d = obs_ptr(c, slot, W - 1);
                        const double *sr = obs_ptr(c, slot, W);
                        for (int q = 0; q < VIO_OBS_D; q++) d[q] = sr[q];
                    }
                    c.lm_nobs[slot] = no - 1;
                    if (no - 1 == 0) keep = 0;
                }
            }
            flag[k] = keep;
        }
        __syncthreads();
        lm_compact(c, flag, offs, scratch);
    }
    PH(29);
    if (dyn_initial) return;   // still INITIAL: nothing to publish
    // ---- removeFailures (feature_manager.cpp:225-233); not called on the STATIC initialisation frame (estimator.cpp:282-290)
    if (sflag0 == 1) {
        nlm = be.n_lm;
        for (int k = t; k < nlm; k += nt) flag[k] = c.lm_solve[c.lm_order[k]] == 2 ? 0 : 1;
        __syncthreads();
        lm_compact(c, flag, offs, scratch);
    }
    if (t == 0) {
        for (int k = 0; k < 9; k++) { be.last_R[k] = be.Rs[W][k]; be.last_R0[k] = be.Rs[0][k]; }
        for (int k = 0; k < 3; k++) { be.last_P[k] = be.Ps[W][k]; be.last_P0[k] = be.Ps[0][k]; }
        // CSV row of visualization.cpp:214-225
        quat q = R2q(ldm(be.Rs[W]));
        od[0] = be.Headers[W];
        od[1] = be.Ps[W][0]; od[2] = be.Ps[W][1]; od[3] = be.Ps[W][2];
        od[4] = q.w; od[5] = q.x; od[6] = q.y; od[7] = q.z;
        od[8] = be.Vs[W][0]; od[9] = be.Vs[W][1]; od[10] = be.Vs[W][2];
        int hc = B.odom_count[s];
        double *hrow = B.odom_hist + ((size_t)s * B.hist_cap + (hc % B.hist_cap)) * 11;  // ring: the getter un-rotates it
        for (int k = 0; k < 11; k++) hrow[k] = od[k];
        B.odom_count[s] = hc + 1;
        if (be.overflow) be.overflow_frames++;
    }
    PH(30);
}

// ====================================================================================================== dynamic init hand-over
// After a successful host-side initialStructure (dyninit_host.cpp) the host has written Ps / Rs / Vs / Bgs / Bas / g of sequence
// `seq`: re-propagate the window pre-integrations at the new gyroscope bias (estimator.cpp:829-836) and triangulate with the new
// poses (solveOdometry -> triangulateWithDepth, :921-933).  The regular be_solve / be_marg launches follow.
// `samples` = (dt, acc[3], gyr[3]) of every IMU step of window slot j in [offs[j], offs[j + 1]), rebuilt by the host from its mirror of
// all_image_frame: slots that absorbed dropped frames (MARGIN_SECOND_NEW while INITIAL, estimator.cpp:1651-1687) hold more steps than
// the per-slot sample buffer on the device keeps (the reference's buffers are unbounded vectors).
__global__ __launch_bounds__(256) void be_dyn_finalize_kernel(Batch B, int seq, const double *samples, const int *offs) {
    __shared__ PreWork pw;
    Ctx c = make_ctx(B, seq);
    BeSeq &be = *c.be;
    const vio_config &cfg = c.C->c;
    const int t = threadIdx.x, nt = blockDim.x, W = c.W;
    for (int j = 0; j <= W; j++) {
        PreInt &p = c.pre[be.pre_idx[j]];
        if (!p.valid) continue;
        if (t == 0) {
            v3 la = ld3(p.lin_acc), lg = ld3(p.lin_gyr);
            p.sum_dt = 0;
            st3(p.acc0, la); st3(p.gyr0, lg);
            p.dp[0] = p.dp[1] = p.dp[2] = 0; p.dv[0] = p.dv[1] = p.dv[2] = 0;
            p.dq[0] = 1; p.dq[1] = p.dq[2] = p.dq[3] = 0;
            p.lin_ba[0] = p.lin_ba[1] = p.lin_ba[2] = 0;
            st3(p.lin_bg, ld3(be.Bgs[j]));
        }
        for (int i = t; i < 225; i += nt) { pw.J[i] = ((i / 15) == (i % 15)) ? 1.0 : 0.0; pw.Pm[i] = 0; }
        __syncthreads();
        for (int q = offs[j]; q < offs[j + 1]; q++) {
            const double *sm = samples + (size_t)q * 7;
            preint_propagate(p, pw, cfg, sm[0], ld3(sm + 1), ld3(sm + 4));
        }
        preint_store(p, pw);
    }
    __syncthreads();
    triangulate_with_depth(c, be.n_lm);
}

// stage test of the device solvePnP: par6 = (rvec, tvec) in / out
__global__ __launch_bounds__(256) void be_stage_pnp_kernel(const double *pts, int n, double *par6) {
    __shared__ double sh_par[6], sh_prev[6], sw[256];
    if (threadIdx.x < 6) sh_par[threadIdx.x] = par6[threadIdx.x];
    __syncthreads();
    pnp_refine_block(pts, n, sh_par, sh_prev, sw);
    if (threadIdx.x < 6) par6[threadIdx.x] = sh_par[threadIdx.x];
}

#include "be_phased.h"

// ====================================================================================================== stage tests
// IntegrationBase::push_back x n followed by IMUFactor::Evaluate, through the same device code the pipeline uses.
__global__ __launch_bounds__(256) void be_stage_imu_kernel(vio_config cfg, PreInt *P, int n, const double *dt, const double *acc,
                                                            const double *gyr, const double *par /*pi7 sbi9 pj7 sbj9*/, double g_norm,
                                                            double *preint_out, double *r15, double *J480) {
    __shared__ PreWork pw;
    const int t = threadIdx.x;
    preint_load(*P, pw);
    for (int q = 0; q < n; q++) preint_propagate(*P, pw, cfg, dt[q], ld3(acc + 3 * q), ld3(gyr + 3 * q));
    preint_store(*P, pw);
    if (t == 0) {
        PreInt &p = *P;
        preint_out[0] = p.dp[0]; preint_out[1] = p.dp[1]; preint_out[2] = p.dp[2];
        for (int k = 0; k < 4; k++) preint_out[3 + k] = p.dq[k];
        preint_out[7] = p.dv[0]; preint_out[8] = p.dv[1]; preint_out[9] = p.dv[2];
        preint_out[10] = p.sum_dt;
        for (int k = 0; k < 225; k++) { preint_out[11 + k] = p.jac[k]; preint_out[236 + k] = p.cov[k]; }
        double raw[15], Jr[450];
        v3 G = mk(0, 0, g_norm);
        bf::imu_raw_residual(p, G, par, par + 7, par + 16, par + 23, raw);
        bf::imu_raw_jacobian(p, G, par, par + 7, par + 16, par + 23, Jr);
        for (int r = 0; r < 15; r++) {
            double sacc = 0;
            for (int k = 0; k <= r; k++) sacc += p.sqrt_info[r * 15 + k] * raw[k];
            r15[r] = sacc;
        }
        // whitened Jacobians in the reference's global layout: 15x7, 15x9, 15x7, 15x9 (7th pose column = 0)
        for (int r = 0; r < 15; r++)
            for (int col = 0; col < 30; col++) {
                double sacc = 0;
                for (int k = 0; k <= r; k++) sacc += p.sqrt_info[r * 15 + k] * Jr[k * 30 + col];
                if (col < 6) J480[r * 7 + col] = sacc;
                else if (col < 15) J480[105 + r * 9 + (col - 6)] = sacc;
                else if (col < 21) J480[240 + r * 7 + (col - 15)] = sacc;
                else J480[345 + r * 9 + (col - 21)] = sacc;
            }
        for (int r = 0; r < 15; r++) { J480[r * 7 + 6] = 0; J480[240 + r * 7 + 6] = 0; }
    }
}
__global__ void be_stage_projection_kernel(vio_config cfg, const double *in /*pi7 pj7 ex7 inv_dep td oi9 oj9*/, int use_td, int form, double *r2, double *J46) {
    if (threadIdx.x != 0) return;
    // form 0: the frame-pair formulation the solver uses (evaluate()); form 1: bf::eval_projection, the per-residual form used by the
    // marginalisation and by outlier rejection
    double J[40], wgt;
    if (form == 0) {
        bf::PairGeo g;
        bf::pair_geo(in, in + 7, in + 14, g);
        double ricm[9];
        stm(ricm, q2R(mkq(in[14 + 6], in[14 + 3], in[14 + 4], in[14 + 5])));
        bf::eval_projection_pair(cfg, g, ricm, in + 14, in[21], in[22], in + 23, in + 32, use_td != 0, r2, J, false, &wgt);
    } else
        bf::eval_projection(cfg, in, in + 7, in + 14, in[21], in[22], in + 23, in + 32, use_td != 0, r2, J);
    for (int a = 0; a < 2; a++) {
        for (int d = 0; d < 6; d++) { J46[a * 7 + d] = J[a * 20 + d]; J46[14 + a * 7 + d] = J[a * 20 + 6 + d]; J46[28 + a * 7 + d] = J[a * 20 + 12 + d]; }
        J46[a * 7 + 6] = 0; J46[14 + a * 7 + 6] = 0; J46[28 + a * 7 + 6] = 0;
        J46[42 + a] = J[a * 20 + 19];
        J46[44 + a] = J[a * 20 + 18];
    }
}
// The IMU factor exactly as evaluate() + assemble() process it in the solver: raw residual whitened by thread 0, the four raw
// Jacobian column groups by imu_raw_jacobian_part, then [Jw r]^T [Jw r] on the matrix cores (imu_block_mfma).  One wavefront.
// G961: the 31 x 31 Gram matrix, row-major, columns = pose_i(6) speedbias_i(9) pose_j(6) speedbias_j(9) | r.
__global__ __launch_bounds__(64) void be_stage_imu_block_kernel(const PreInt *P, const double *par /*pi7 sbi9 pj7 sbj9*/, double g_norm, double *G961) {
    __shared__ double raw_l[472], M_l[232];
    const int lane = threadIdx.x;
    const PreInt &p = *P;
    v3 G = mk(0, 0, g_norm);
    if (lane == 0) {
        double raw[15];
        bf::imu_raw_residual(p, G, par, par + 7, par + 16, par + 23, raw);
        for (int r = 0; r < 15; r++) {
            double sacc = 0;
            for (int k = 0; k <= r; k++) sacc += p.sqrt_info[r * 15 + k] * raw[k];
            raw_l[r * 31 + 30] = sacc;
        }
    } else if (lane <= 4)
        bf::imu_raw_jacobian_part(p, G, par, par + 7, par + 16, par + 23, lane - 1, raw_l, 31);
    for (int q = lane; q < 225; q += 64) M_l[q] = p.sqrt_info[q];
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    v4f64 a00 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
    imu_block_mfma(raw_l, M_l, li, lk, a00, a10, a11);
    for (int r = 0; r < 4; r++) {
        const int row = lk + 4 * r, col = li;
        G961[row * 31 + col] = a00[r];
        if (16 + row < 31) {
            G961[(16 + row) * 31 + col] = a10[r];
            G961[col * 31 + 16 + row] = a10[r];
            if (16 + col < 31) G961[(16 + row) * 31 + 16 + col] = a11[r];
        }
    }
}
