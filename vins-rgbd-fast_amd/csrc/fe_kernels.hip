// Front-end kernels (gfx950): FeatureTracker::readImage (vins_estimator/src/feature_tracker/feature_tracker.cpp:263-439)
// batched over S sequences.  One launch chain per camera frame:
//   fe_begin   predictMotion (estimator.cpp:1790-1860) + first-image / IMU-availability gating
//   fe_pyramid cv::pyrDown levels of the new frame (+ level-0 copy into the ping-pong buffer)
//   fe_lk      calcOpticalFlowPyrLK 21x21, one wavefront per feature, patches staged in LDS, wave reductions
//   fe_select  status/border cull, track_cnt++, rejectWithF (7-point RANSAC), setMask, per-grid counts
//   fe_fast    grid-FAST: FAST-9/16 score + NMS per deficit cell in an LDS tile (gridDetect :105-171, first half)
//   fe_add     mask filter + top-k + addPoints per cell in order, undistortedPoints, updateID, feature-map packaging
// All arithmetic follows SURVEY.md Appendix B so results are bit-identical to oracle/frontend.cpp
// (integer sums are exact, float/double ops are IEEE with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include "kernels.h"

// in-kernel phase timers of sequence 0 (thread 0, 100 MHz ticks into Batch::timings, slots 64..; tools/phase_profile.py)
#if VIO_TIMERS
#define FE_PH_INIT long long fe_t0 = (s == 0 && threadIdx.x == 0) ? VIO_CLOCK() : 0
#define FE_PH(k) do { if (s == 0 && threadIdx.x == 0) { long long n_ = VIO_CLOCK(); B.timings[k] += (float)(n_ - fe_t0); fe_t0 = n_; } } while (0)
#else
#define FE_PH_INIT do {} while (0)
#define FE_PH(k) do {} while (0)
#endif

namespace {

__device__ __forceinline__ int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int cv_floor(float v) { return (int)floorf(v); }
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// 64-bit integer wavefront sum on DPP moves + v_readlane (exact, so the order does not matter): four DPP steps leave the sum of
// each 16-lane row in all its lanes, the four row sums are combined through SGPRs.  Replaces 12 ds_bpermute round trips.
__device__ __forceinline__ long long dpp_i64(long long v, const int ctrl_tag) {
    int lo = (int)(v & 0xffffffffLL), hi = (int)(v >> 32);
    switch (ctrl_tag) {
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break;
    }
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long readlane_i64(long long v, int src_lane) {
    int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), src_lane), hi = __builtin_amdgcn_readlane((int)(v >> 32), src_lane);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    v += dpp_i64(v, 0);
    v += dpp_i64(v, 1);
    v += dpp_i64(v, 2);
    v += dpp_i64(v, 3);
    return (readlane_i64(v, 0) + readlane_i64(v, 16)) + (readlane_i64(v, 32) + readlane_i64(v, 48));
}

// Exact wavefront sum of int32 partials whose 8-lane sums still fit int32 (the LK sums below: 7 products of at most 8160 * 4080 per
// lane).  Three DPP adds in 32 bits leave the sum of every 8-lane half row in its lanes; the eight half-row sums are combined in 64
// bits on the scalar unit.  Integer arithmetic: the order does not matter, the result is the same number the CPU code accumulates.
__device__ __forceinline__ long long wave_sum_i32x(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    long long a = (long long)__builtin_amdgcn_readlane(v, 0) + (long long)__builtin_amdgcn_readlane(v, 8);
    long long b = (long long)__builtin_amdgcn_readlane(v, 16) + (long long)__builtin_amdgcn_readlane(v, 24);
    long long c = (long long)__builtin_amdgcn_readlane(v, 32) + (long long)__builtin_amdgcn_readlane(v, 40);
    long long d = (long long)__builtin_amdgcn_readlane(v, 48) + (long long)__builtin_amdgcn_readlane(v, 56);
    return (a + b) + (c + d);
}
// bilinear blend of four samples with the 14-bit fixed-point weights: every factor fits 24 bits (samples are u8 or Scharr sums of
// at most 4080 in magnitude, weights at most 2^14), so the full-rate 24-bit multiplier gives the exact 32-bit products
__device__ __forceinline__ int blend4(int p00, int p01, int p10, int p11, int w00, int w01, int w10, int w11) {
    return __mul24(p00, w00) + __mul24(p01, w01) + __mul24(p10, w10) + __mul24(p11, w11);
}

// camera (camera_model/src/camera_models/PinholeCamera.cc:449-542,645-662)
__device__ __forceinline__ void cam_distortion(const vio_config &c, double x, double y, double &dx, double &dy) {
    double mx2 = x * x, my2 = y * y, mxy = x * y;
    double rho2 = mx2 + my2;
    double rad = c.k1 * rho2 + c.k2 * rho2 * rho2;
    dx = x * rad + 2.0 * c.p1 * mxy + c.p2 * (rho2 + 2.0 * mx2);
    dy = y * rad + 2.0 * c.p2 * mxy + c.p1 * (rho2 + 2.0 * my2);
}
__device__ void cam_lift(const vio_config &c, double u, double v, double &x, double &y) {
    double inv_K11 = 1.0 / c.fx, inv_K13 = -c.cx / c.fx, inv_K22 = 1.0 / c.fy, inv_K23 = -c.cy / c.fy;
    double mx_d = inv_K11 * u + inv_K13, my_d = inv_K22 * v + inv_K23, dx, dy;
    cam_distortion(c, mx_d, my_d, dx, dy);
    double mx_u = mx_d - dx, my_u = my_d - dy;
    for (int i = 1; i < 8; i++) {
        cam_distortion(c, mx_u, my_u, dx, dy);
        mx_u = mx_d - dx;
        my_u = my_d - dy;
    }
    x = mx_u;
    y = my_u;
}
__device__ void cam_project(const vio_config &c, double X, double Y, double Z, double &u, double &v) {
    double px = X / Z, py = Y / Z, dx, dy;
    cam_distortion(c, px, py, dx, dy);
    u = c.fx * (px + dx) + c.cx;
    v = c.fy * (py + dy) + c.cy;
}

// mask(p) == 0 <=> p lies in the cv::circle(filled, r = MIN_DIST) raster of some accepted centre
__device__ __forceinline__ bool in_disk(const int *hw, int r, int px, int py, int cx, int cy) {
    int dy = py - cy;
    dy = dy < 0 ? -dy : dy;
    if (dy > r) return false;
    int dx = px - cx;
    dx = dx < 0 ? -dx : dx;
    return dx <= hw[dy];
}

}  // namespace

// ------------------------------------------------------------------------------------------------ fe_begin
// Estimator::predictMotion(t0, t1) (estimator.cpp:1790-1860): gyro integration over the IMU ring, nothing is consumed.  The rotation
// into the camera frame uses the CONFIGURED extrinsic (the global RIC.back(), :1852), which stays at its yaml value while
// ESTIMATE_EXTRINSIC refines Estimator::ric.
__device__ dm::m3 predict_motion(const Batch &B, int s, double t0, double t1) {
    const DevCfg &C = *B.cfg;
    const BeSeq &be = B.be[s];
    int imu_head = be.imu_head;
    if (be.imu_count - imu_head > C.NIMU) imu_head = be.imu_count - C.NIMU;  // ring bookkeeping: samples that were overwritten
    const double *it = B.imu_t + (size_t)s * C.NIMU;
    const double *ig = B.imu_gyr + (size_t)s * C.NIMU * 3;
    const bool have = be.imu_count > imu_head;
    const double back_t = have ? it[(be.imu_count - 1) % C.NIMU] : -1e300;
    dm::m3 rel = dm::eye();
    if (!(have && t1 <= back_t)) return rel;
    int k = imu_head;
    while (k < be.imu_count && it[k % C.NIMU] <= t0) k++;
    bool first = true;
    double prev_t = 0;
    dm::v3 prev_gyr = dm::mk(0, 0, 0);
    dm::m3 ricT = dm::tr(dm::ldm(C.c.ric));
    dm::v3 bg = dm::ld3(B.tracker_lag ? be.track_Bg : be.latest_Bg);
    while (k < be.imu_count && it[k % C.NIMU] <= t1) {
        double tk = it[k % C.NIMU];
        dm::v3 w = dm::ld3(ig + (size_t)(k % C.NIMU) * 3);
        k++;
        if (first) { prev_t = tk; first = false; prev_gyr = w; continue; }
        double dt = tk - prev_t;
        prev_t = tk;
        dm::v3 un_gyr = dm::sub(dm::scl(0.5, dm::add(prev_gyr, w)), bg);
        prev_gyr = w;
        dm::v3 aa = dm::scl(dt, dm::mul(ricT, un_gyr));
        double ang = dm::nrm(aa);
        dm::m3 Rk = dm::eye();
        if (ang > 0) {
            dm::v3 ax = dm::mk(aa.x / ang, aa.y / ang, aa.z / ang);   // angle_axis.normalized(): a division per coefficient (Eigen 3.3), like the oracle
            double sn, cs;
            dm::sincos_det(ang, &sn, &cs);   // same bits as the CPU restatement (dmath.h)
            dm::m3 K = dm::skew(ax);
            Rk = dm::add(dm::add(dm::eye(), dm::scl(sn, K)), dm::scl(1 - cs, dm::mul(K, K)));
        }
        rel = dm::mul(rel, dm::tr(Rk));
    }
    return rel;
}
// The same integration by one wavefront (fe_begin): the scalar version walks the IMU ring with one dependent global load per sample
// and a sin / cos per step (65 us per frame on the front-end's critical path).  Here 64 ring entries are read at once, the first
// sample beyond t0 is found with a ballot, every lane forms the rotation increment of its own sample (the neighbour's stamp and
// rate come through a lane shift), and only the ordered product rel <- rel * Rk^T runs sequentially on broadcast matrices.
// Identical operations on identical operands, hence the identical matrix.
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ dm::m3 predict_motion_wave(const Batch &B, int s, double t0, double t1) {
    const DevCfg &C = *B.cfg;
    const BeSeq &be = B.be[s];
    const int lane = threadIdx.x & 63;
    int imu_head = be.imu_head;
    const int imu_count = be.imu_count;
    if (imu_count - imu_head > C.NIMU) imu_head = imu_count - C.NIMU;  // ring bookkeeping: samples that were overwritten
    const double *it = B.imu_t + (size_t)s * C.NIMU;
    const double *ig = B.imu_gyr + (size_t)s * C.NIMU * 3;
    const bool have = imu_count > imu_head;
    const double back_t = have ? it[(imu_count - 1) % C.NIMU] : -1e300;
    dm::m3 rel = dm::eye();
    if (!(have && t1 <= back_t)) return rel;
    // first sample with t > t0
    int k = imu_head;
    while (k < imu_count) {
        const int j = k + lane;
        const bool le = j < imu_count && it[j % C.NIMU] <= t0;
        const unsigned long long bal = __ballot(le);
        if (bal == ~0ULL) { k += 64; continue; }
        k += __builtin_ctzll(~bal);   // stamps ascend: the lanes with t <= t0 are a prefix
        break;
    }
    const dm::m3 ricT = dm::tr(dm::ldm(C.c.ric));
    const dm::v3 bg = dm::ld3(B.tracker_lag ? be.track_Bg : be.latest_Bg);
    bool first = true;
    double prev_t = 0;
    dm::v3 prev_gyr = dm::mk(0, 0, 0);
    while (k < imu_count) {
        const int j = k + lane;
        const bool in = j < imu_count;
        const double tk = in ? it[j % C.NIMU] : 1e300;
        const dm::v3 w = in ? dm::ld3(ig + (size_t)(j % C.NIMU) * 3) : dm::mk(0, 0, 0);
        const unsigned long long bal = __ballot(in && tk <= t1);
        const int nv = bal == ~0ULL ? 64 : __builtin_ctzll(~bal);   // samples of this chunk inside (t0, t1], a prefix
        if (nv == 0) break;
        // the sample before mine: the neighbour lane's, or the last one of the previous chunk for lane 0
        double pt = __shfl_up(tk, 1, 64);
        dm::v3 pw = dm::mk(__shfl_up(w.x, 1, 64), __shfl_up(w.y, 1, 64), __shfl_up(w.z, 1, 64));
        if (lane == 0) { pt = prev_t; pw = prev_gyr; }
        const bool step = lane < nv && !(first && lane == 0);   // the very first sample only seeds (prev_t, prev_gyr)
        dm::m3 RkT = dm::eye();
        if (step) {
            const double dt = tk - pt;
            const dm::v3 un_gyr = dm::sub(dm::scl(0.5, dm::add(pw, w)), bg);
            const dm::v3 aa = dm::scl(dt, dm::mul(ricT, un_gyr));
            const double ang = dm::nrm(aa);
            dm::m3 Rk = dm::eye();
            if (ang > 0) {
                const dm::v3 ax = dm::mk(aa.x / ang, aa.y / ang, aa.z / ang);
                double sn, cs;
                dm::sincos_det(ang, &sn, &cs);
                const dm::m3 K = dm::skew(ax);
                Rk = dm::add(dm::add(dm::eye(), dm::scl(sn, K)), dm::scl(1 - cs, dm::mul(K, K)));
            }
            RkT = dm::tr(Rk);
        }
        for (int q = (first ? 1 : 0); q < nv; q++) {
            dm::m3 Mq;
#pragma unroll
            for (int e = 0; e < 9; e++) Mq.a[e] = lane_bcast(RkT.a[e], q);
            rel = dm::mul(rel, Mq);
        }
        first = false;
        prev_t = lane_bcast(tk, nv - 1);
        prev_gyr = dm::mk(lane_bcast(w.x, nv - 1), lane_bcast(w.y, nv - 1), lane_bcast(w.z, nv - 1));
        if (nv < 64) break;
        k += 64;
    }
    return rel;
}
// Estimator::predict (estimator.cpp:1862-1880) applied from the newest window state through every IMU sample that arrived after it:
// what pubLatestOdometry publishes at IMU rate (inputIMU, :1749-1766, after updateLatestStates :1768-1788).  out11 = t, P(3),
// Q(w, x, y, z), V(3); returns the window state itself when no newer sample is in the ring.  Output only: nothing is modified.
// Default: every sample is applied with its own values and becomes acc_0 / gyr_0 of the next step.  vio_config.reference_quirks bit 0
// (VIO_QUIRK_LATEST_FRONT) reproduces Estimator::updateLatestStates literally (estimator.cpp:1779-1786): the loop walks
// the buffered stamps but hands predict() the values of the queue's FRONT sample every time, and predict() never advances acc_0 / gyr_0
// (they stay as processIMU left them, :1862-1880).
__global__ void be_latest_odometry_kernel(Batch B, int seq, double *out11) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const DevCfg &C = *B.cfg;
    const BeSeq &be = B.be[seq];
    const int fc = be.frame_count;
    double latest_time = be.Headers[fc] + be.td;
    dm::v3 P = dm::ld3(be.Ps[fc]), V = dm::ld3(be.Vs[fc]), Ba = dm::ld3(be.Bas[fc]), Bg = dm::ld3(be.Bgs[fc]), g = dm::ld3(be.g);
    dm::m3 R = dm::ldm(be.Rs[fc]);
    dm::v3 acc_0 = dm::ld3(be.acc_0), gyr_0 = dm::ld3(be.gyr_0);
    const double *it = B.imu_t + (size_t)seq * C.NIMU, *ia = B.imu_acc + (size_t)seq * C.NIMU * 3, *ig = B.imu_gyr + (size_t)seq * C.NIMU * 3;
    int k = be.imu_head;
    if (be.imu_count - k > C.NIMU) k = be.imu_count - C.NIMU;
    const bool front = (C.c.reference_quirks & VIO_QUIRK_LATEST_FRONT) != 0;
    const int idx_front = k % C.NIMU;
    const int n_front = be.imu_count_ingest;
    if (be.solver_flag == 1 && C.c.use_imu)
        for (; k < be.imu_count; k++) {
            const int idx = k % C.NIMU;
            const double t = it[idx];
            if (!(t > latest_time)) continue;
            const double dt = t - latest_time;
            latest_time = t;
            // quirk mode: front values only for the samples that were in the ring when the frame was taken (updateLatestStates'
            // replay); samples pushed later went through inputIMU -> predict with their own values (estimator.cpp:1758-1764)
            const int iv = (front && k < n_front) ? idx_front : idx;
            const dm::v3 a1 = dm::ld3(ia + (size_t)iv * 3), w1 = dm::ld3(ig + (size_t)iv * 3);
            const dm::v3 un_acc_0 = dm::sub(dm::mul(R, dm::sub(acc_0, Ba)), g);
            const dm::v3 un_gyr = dm::sub(dm::scl(0.5, dm::add(gyr_0, w1)), Bg);
            R = dm::mul(R, dm::q2R(dm::deltaQ(dm::scl(dt, un_gyr))));
            const dm::v3 un_acc_1 = dm::sub(dm::mul(R, dm::sub(a1, Ba)), g);
            const dm::v3 un_acc = dm::scl(0.5, dm::add(un_acc_0, un_acc_1));
            P = dm::add(dm::add(P, dm::scl(dt, V)), dm::scl(0.5 * dt * dt, un_acc));
            V = dm::add(V, dm::scl(dt, un_acc));
            if (!front) { acc_0 = a1; gyr_0 = w1; }
        }
    const dm::quat q = dm::R2q(R);
    out11[0] = latest_time; out11[1] = P.x; out11[2] = P.y; out11[3] = P.z;
    out11[4] = q.w; out11[5] = q.x; out11[6] = q.y; out11[7] = q.z; out11[8] = V.x; out11[9] = V.y; out11[10] = V.z;
}

// vio_predict_motion: one sequence, result to out9 (row-major)
__global__ void fe_predict_motion_kernel(Batch B, int seq, double t0, double t1, double *out9) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    dm::stm(out9, predict_motion(B, seq, t0, t1));
}

// grid S, 64 threads.  gate: 1 = vio_feed (nodelet gating: IMU availability, first-image skip, init_pub / init_feature), 0 = vio_track.
// modes: per-sequence frame mode (VIO_FRAME_SKIP / TRACK / PUBLISH) or NULL = `publish` for every sequence.
// R_rel: caller-supplied relative rotations [S][9] (readImage(img, t, relative_R), feature_tracker.h:36-37) or NULL; a NaN in the
// first element of a sequence's matrix means "predict on the device" for that sequence.
__global__ __launch_bounds__(64) void fe_begin_kernel(Batch B, const double *stamps, int gate, int publish, const uint8_t *modes, const double *R_rel) {
    int s = blockIdx.x + B.s0;
    const bool w0 = threadIdx.x == 0;   // every lane follows the same (uniform) control flow, lane 0 does the stores
    const DevCfg &C = *B.cfg;
    FeSeq &fe = B.fe[s];
    const BeSeq &be = B.be[s];  // read-only here: the previous frame's marginalisation may still be running on the other stream
    double t = stamps[s];
    const double td = B.tracker_lag ? be.track_td : be.td;
    const int first_image = fe.first_image_flag;
    const double last_image_time = fe.last_image_time;
    if (w0) {
        fe.n_deficit = 0;
        fe.n_obs = 0;
        fe.publish_ok = 0;
        fe.overflow = 0;
    }
    const int mode = modes ? (int)modes[s] : (publish ? VIO_FRAME_PUBLISH : VIO_FRAME_TRACK);
    if (w0) fe.pub_req = mode == VIO_FRAME_PUBLISH;
    if (gate) {
        int imu_head = be.imu_head;
        if (be.imu_count - imu_head > C.NIMU) imu_head = be.imu_count - C.NIMU;
        const double *it = B.imu_t + (size_t)s * C.NIMU;
        bool have = be.imu_count > imu_head;
        double back_t = have ? it[(be.imu_count - 1) % C.NIMU] : -1e300;
        // caller contract of vio_feed: IMU pushed through stamp + td (upstream busy-waits, estimator.cpp:178-183); VO mode has no IMU
        if (C.c.use_imu && !(have && t + td <= back_t)) {
            if (w0) fe.n_forw = -2;  // nothing consumed; tells the later kernels to skip this sequence (be_ingest reports VIO_NEED_IMU)
            return;
        }
        if (first_image) {  // estimator_nodelet.cpp:234-240
            if (w0) {
                fe.first_image_flag = 0;
                fe.last_image_time = t;
                fe.n_forw = -1;
            }
            return;
        }
    }
    if (mode == VIO_FRAME_SKIP) {  // frequency control dropped the frame before readImage (estimator_nodelet.cpp:264-271): no state changes
        if (w0) fe.n_forw = -1;
        return;
    }
    // Estimator::predictMotion(last_image_time, t + td), unless the caller handed in relative_R
    const double *Rc = R_rel ? R_rel + (size_t)s * 9 : nullptr;
    if (!C.c.use_imu) {   // readImage(img, t) (estimator_nodelet.cpp:315): no prediction, LK starts at the old positions
        if (w0) { dm::stm(fe.R_rel, dm::eye()); fe.use_R_rel = 0; }
    } else if (Rc && Rc[0] == Rc[0]) {
        if (w0) { for (int k = 0; k < 9; k++) fe.R_rel[k] = Rc[k]; fe.use_R_rel = 1; }
    } else {
        const dm::m3 R = predict_motion_wave(B, s, last_image_time, t + td);
        if (w0) { dm::stm(fe.R_rel, R); fe.use_R_rel = 0; }
    }
    if (w0) {
        fe.last_image_time = t;
        fe.cur_time = t;
        fe.n_forw = 0;
        fe.n_unstable = 0;
    }
}

// ------------------------------------------------------------------------------------------------ fe_pyramid
// cv::pyrDown: [1 4 6 4 1]^2/256, BORDER_REFLECT_101, (sum+128)>>8.  grid (tiles_x, tiles_y, S), 256 threads.
// Each workgroup produces a 64x16 tile of dst from a (2*64+3)x(2*16+3) LDS tile of src (coalesced row loads);
// when copy0 != NULL the 128x32 source pixels owned by the tile are also copied to the level-0 ping-pong buffer.
#define PD_TW 64
#define PD_TH 16
#define PD_PITCH (2 * PD_TW + 8)   // LDS pitch of the source tile: it is staged from the 4-byte aligned column 2 ox - 4
__device__ void pyrdown_tile(const uint8_t *src, int sw, int sh, uint8_t *dst, uint8_t *l0) {
    int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    __shared__ __attribute__((aligned(16))) uint8_t tile[2 * PD_TH + 3][PD_PITCH];
    __shared__ int hrow[2 * PD_TH + 3][PD_TW];
    int ox = blockIdx.x * PD_TW, oy = blockIdx.y * PD_TH;
    int sx0 = 2 * ox - 2, sy0 = 2 * oy - 2;
    const int TWS = 2 * PD_TW + 3, THS = 2 * PD_TH + 3;
    const int sxa = sx0 - 2;   // multiple of 4; tile column of source pixel gx = gx - sxa
    if (sxa >= 0 && sxa + PD_PITCH <= sw && sy0 >= 0 && sy0 + THS <= sh && !(sw & 3) && !(((size_t)src | (size_t)l0) & 3)) {
        // interior tile: aligned 4-byte loads (and 4-byte stores of the level-0 copy)
        const uint32_t *s32 = (const uint32_t *)(src + (size_t)sy0 * sw + sxa);
        const int sw4 = sw >> 2;
        for (int q = threadIdx.x; q < THS * (PD_PITCH / 4); q += 256) {
            const int ty = q / (PD_PITCH / 4), c4 = q - ty * (PD_PITCH / 4);
            ((uint32_t *)&tile[ty][0])[c4] = s32[(size_t)ty * sw4 + c4];
        }
        __syncthreads();
        if (l0) {   // the tile owns source pixels [2 ox, 2 ox + 128) x [2 oy, 2 oy + 32): tile columns 4 .., rows 2 ..
            uint32_t *d32 = (uint32_t *)(l0 + (size_t)(2 * oy) * sw + 2 * ox);
            for (int q = threadIdx.x; q < 2 * PD_TH * (2 * PD_TW / 4); q += 256) {
                const int ty = q / (2 * PD_TW / 4), c4 = q - ty * (2 * PD_TW / 4);
                d32[(size_t)ty * sw4 + c4] = ((const uint32_t *)&tile[ty + 2][4])[c4];
            }
        }
    } else {
        for (int q = threadIdx.x; q < TWS * THS; q += 256) {
            int ty = q / TWS, tx = q - ty * TWS;
            int gx = sx0 + tx, gy = sy0 + ty;
            // pixels further than one reflection outside the image feed no valid output: clamp only keeps the address legal
            int ry = min(max(reflect101(gy, sh), 0), sh - 1), rx = min(max(reflect101(gx, sw), 0), sw - 1);
            uint8_t v = src[(size_t)ry * sw + rx];
            tile[ty][tx + 2] = v;
            if (l0 && tx >= 2 && tx < 2 + 2 * PD_TW && ty >= 2 && ty < 2 + 2 * PD_TH && gx < sw && gy < sh) l0[(size_t)gy * sw + gx] = v;
        }
        __syncthreads();
    }
    for (int q = threadIdx.x; q < THS * PD_TW; q += 256) {
        int ty = q / PD_TW, x = q - ty * PD_TW;
        const uint8_t *r = &tile[ty][2 * x + 2];
        hrow[ty][x] = r[0] + 4 * r[1] + 6 * r[2] + 4 * r[3] + r[4];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < PD_TH * PD_TW; q += 256) {
        int y = q / PD_TW, x = q - y * PD_TW;
        int gx = ox + x, gy = oy + y;
        if (gx < dw && gy < dh) {
            int acc = hrow[2 * y][x] + 4 * hrow[2 * y + 1][x] + 6 * hrow[2 * y + 2][x] + 4 * hrow[2 * y + 3][x] + hrow[2 * y + 4][x];
            dst[(size_t)gy * dw + gx] = (uint8_t)((acc + 128) >> 8);
        }
    }
}
__global__ __launch_bounds__(256) void fe_pyrdown_kernel(Batch B, const uint8_t *src_base, size_t src_stride, int sw, int sh,
                                                         int dst_level, int write_level0) {
    const DevCfg &C = *B.cfg;
    int s = blockIdx.z + B.s0;
    FeSeq &fe = B.fe[s];
    if (fe.n_forw < 0) return;
    int forw = fe.has_img ? (fe.cur_buf ^ 1) : fe.cur_buf;
    const uint8_t *src = src_base ? src_base + (size_t)s * src_stride
                                  : B.pyr + ((size_t)s * 2 + forw) * C.pyr_bytes + C.lvl_off[dst_level - 1];
    uint8_t *dst = B.pyr + ((size_t)s * 2 + forw) * C.pyr_bytes + C.lvl_off[dst_level];
    uint8_t *l0 = write_level0 ? B.img + ((size_t)s * 2 + forw) * (size_t)sw * sh : nullptr;
    pyrdown_tile(src, sw, sh, dst, l0);
}
__global__ __launch_bounds__(256) void fe_pyrdown_stage_kernel(const uint8_t *src, int sw, int sh, uint8_t *dst) {
    pyrdown_tile(src, sw, sh, dst, nullptr);
}

// ------------------------------------------------------------------------------------------------ fe_clahe
// EQUALIZE: cv::createCLAHE(3.0, Size(8, 8))->apply (feature_tracker.cpp:269-275), two launches per frame.
// fe_clahe_lut: grid (64 tiles, n sequences), 256 threads = the 256 bins: histogram of the tile in LDS (integer atomics, exact in any
// order), clip at int(3 * area / 256), even redistribution + one count on every (256 / residual)-th bin, inclusive scan, LUT byte.
// A size that is not a multiple of 8 is padded bottom / right with BORDER_REFLECT_101 by 8 - (size % 8) in both directions.
struct ClaheGeo { int tw, th, clip; float lut_scale; };
__device__ __forceinline__ ClaheGeo clahe_geo(int W, int H) {
    int We = W, He = H;
    if ((W & 7) || (H & 7)) { We = W + 8 - (W & 7); He = H + 8 - (H & 7); }
    ClaheGeo g;
    g.tw = We >> 3; g.th = He >> 3;
    const int area = g.tw * g.th;
    g.clip = max((int)(3.0 * area / 256), 1);
    g.lut_scale = (float)255 / area;
    return g;
}
__device__ void clahe_lut_tile(const uint8_t *src, int W, int H, int k, uint8_t *lut) {
    __shared__ int hist[256];
    __shared__ int wsum[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const ClaheGeo g = clahe_geo(W, H);
    const int ty = k >> 3, tx = k & 7;
    hist[t] = 0;
    __syncthreads();
    for (int i = t; i < g.tw * g.th; i += 256) {
        const int y = i / g.tw, x = i - y * g.tw;
        atomicAdd(&hist[src[(size_t)reflect101(ty * g.th + y, H) * W + reflect101(tx * g.tw + x, W)]], 1);
    }
    __syncthreads();
    int hv = hist[t];
    int excess = max(hv - g.clip, 0);
    hv = min(hv, g.clip);
    int e = excess;
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    if (lane == 0) wsum[wv] = e;
    __syncthreads();
    const int clipped = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const int batch = clipped >> 8, residual = clipped - (batch << 8);
    hv += batch;
    if (residual != 0) {
        const int step = max(256 / residual, 1);
        if (t % step == 0 && t / step < residual) hv++;
    }
    __syncthreads();
    // inclusive scan over the 256 bins
    int v = hv;
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    for (int w = 0; w < wv; w++) v += wsum[w];
    const int r = (int)rintf(v * g.lut_scale);
    lut[(size_t)k * 256 + t] = (uint8_t)min(max(r, 0), 255);
}
// interpolation between the LUTs of the four nearest tile centres, the float expression of the CPU code operation by operation
__device__ __forceinline__ uint8_t clahe_pixel(const uint8_t *lut, const ClaheGeo &g, int x, int v, const uint8_t *p1, const uint8_t *p2, float ya,
                                               float ya1, float inv_tw) {
    const float txf = x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
    const float xa = txf - tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0); tx2 = min(tx2, 7);
    const float res = (p1[tx1 * 256 + v] * xa1 + p1[tx2 * 256 + v] * xa) * ya1 + (p2[tx1 * 256 + v] * xa1 + p2[tx2 * 256 + v] * xa) * ya;
    return (uint8_t)min(max((int)rintf(res), 0), 255);
}
__device__ void clahe_apply_rows(const uint8_t *src, int W, int H, const uint8_t *lut, uint8_t *dst) {
    const ClaheGeo g = clahe_geo(W, H);
    const float inv_tw = 1.0f / g.tw, inv_th = 1.0f / g.th;
    const int y = blockIdx.y;
    const float tyf = y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0); ty2 = min(ty2, 7);
    const uint8_t *p1 = lut + (size_t)ty1 * 8 * 256, *p2 = lut + (size_t)ty2 * 8 * 256;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (x0 >= W) return;
    const uint8_t *sr = src + (size_t)y * W;
    uint8_t *dr = dst + (size_t)y * W;
    if ((W & 3) == 0) {
        const uint32_t px = *(const uint32_t *)(sr + x0);
        uint32_t o = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) o |= (uint32_t)clahe_pixel(lut, g, x0 + j, (px >> (8 * j)) & 255, p1, p2, ya, ya1, inv_tw) << (8 * j);
        *(uint32_t *)(dr + x0) = o;
    } else {
        for (int j = 0; j < 4 && x0 + j < W; j++) dr[x0 + j] = clahe_pixel(lut, g, x0 + j, sr[x0 + j], p1, p2, ya, ya1, inv_tw);
    }
}
__global__ __launch_bounds__(256) void fe_clahe_lut_kernel(Batch B, const uint8_t *src_base, size_t stride) {
    const DevCfg &C = *B.cfg;
    const int s = blockIdx.y + B.s0;
    if (B.fe[s].n_forw < 0) return;
    clahe_lut_tile(src_base + (size_t)s * stride, C.c.width, C.c.height, blockIdx.x, B.clahe_lut + (size_t)s * 64 * 256);
}
// grid (ceil(W / 1024), H, n sequences), four pixels per thread
__global__ __launch_bounds__(256) void fe_clahe_apply_kernel(Batch B, const uint8_t *src_base, size_t stride) {
    const DevCfg &C = *B.cfg;
    const int s = blockIdx.z + B.s0;
    if (B.fe[s].n_forw < 0) return;
    clahe_apply_rows(src_base + (size_t)s * stride, C.c.width, C.c.height, B.clahe_lut + (size_t)s * 64 * 256, B.clahe_img + (size_t)s * stride);
}
__global__ __launch_bounds__(256) void fe_clahe_lut_stage_kernel(const uint8_t *src, int W, int H, uint8_t *lut) {
    clahe_lut_tile(src, W, H, blockIdx.x, lut);
}
__global__ __launch_bounds__(256) void fe_clahe_apply_stage_kernel(const uint8_t *src, int W, int H, const uint8_t *lut, uint8_t *dst) {
    clahe_apply_rows(src, W, H, lut, dst);
}

// ------------------------------------------------------------------------------------------------ fe_predict
// predictPtsInNextFrame (feature_tracker.cpp:595-608). grid (ceil(NP/256), S)
__global__ void fe_predict_kernel(Batch B) {
    const DevCfg &C = *B.cfg;
    int s = blockIdx.y + B.s0;
    FeSeq &fe = B.fe[s];
    if (fe.n_forw < 0) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fe.n_pts) return;
    float2 p = B.cur_pts[(size_t)s * C.NP + i];
    if (!C.c.use_imu) { B.forw_pts[(size_t)s * C.NP + i] = p; return; }   // feature_tracker.cpp:307-311: nextPts start at prevPts
    double x, y;
    cam_lift(C.c, p.x, p.y, x, y);
    const double *R = fe.R_rel;
    double X = R[0] * x + R[1] * y + R[2], Y = R[3] * x + R[4] * y + R[5], Z = R[6] * x + R[7] * y + R[8];
    double u, v;
    cam_project(C.c, X, Y, Z, u, v);
    B.forw_pts[(size_t)s * C.NP + i] = make_float2((float)u, (float)v);
}

// ------------------------------------------------------------------------------------------------ fe_lk
// One wavefront (64 lanes) per feature.  Lane L < 63 owns seven horizontally adjacent pixels of the 21 x 21 window: row L / 3,
// columns 7 (L % 3) .. 7 (L % 3) + 6, so every LDS address of the inner loops is one per-lane base plus a compile-time offset and
// neighbouring pixels share their samples (8 + 8 byte reads for 7 bilinear taps instead of 28).  All window sums are integers
// (exact in any order); the float / double steps are the CPU code's, operation by operation.

#define LK_MARGIN 5
#define LK_REG (22 + 2 * LK_MARGIN)   // 32 x 32 search region of the next image
#define LK_RP (LK_REG + 4)            // its LDS pitch: the region is staged from 4-byte aligned columns (up to 3 bytes of lead-in)
#define LK_WP 28                      // LDS pitch of the 24 x 24 template window, same reason
#define LK_WIN_BYTES (24 * LK_WP)
#define LK_JW_BYTES (LK_REG * LK_RP)
// BORDER_REFLECT_101 index, clamped: region pixels further than one reflection outside the image are never part of a window that
// passes the bounds test, the clamp only keeps their address legal
__device__ __forceinline__ int reflect101c(int i, int n) { return min(max(reflect101(i, n), 0), n - 1); }
// a pointer every lane holds the same value of, moved to scalar registers
template <typename T> __device__ __forceinline__ const T *uniform_ptr(const T *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const T *)(((unsigned long long)hi << 32) | lo);
}
typedef const __attribute__((address_space(1))) uint8_t *gmem_u8;    // known-global pointers: global_load, not flat_load
typedef const __attribute__((address_space(1))) uint32_t *gmem_u32;
// Stage the ROWS x (4 NDW)-byte block of `img` whose top-left pixel is (x0a, y0), x0a a multiple of 4, into LDS (pitch 4 NDW).
// Inside the image: one aligned 4-byte load per lane and trip (the row pitch is a multiple of 4 there); at the image border: byte
// loads with BORDER_REFLECT_101 (`clamp` = the clamped variant).  Same bytes either way.
template <int ROWS, int NDW, bool CLAMP>
__device__ __forceinline__ void lk_stage(gmem_u8 img, int w, int h, int x0a, int y0, uint8_t *dst, int lane) {
    if (x0a >= 0 && x0a + 4 * NDW <= w && y0 >= 0 && y0 + ROWS <= h && !(w & 3) && !((size_t)img & 3)) {
        int row = lane / NDW, col = lane - row * NDW;
        const gmem_u32 src = (gmem_u32)(img + (unsigned)(y0 * w + x0a));
        const int w4 = w >> 2;
        uint32_t *d32 = (uint32_t *)dst;
#pragma unroll
        for (int q = lane; q < ROWS * NDW; q += 64) {
            d32[q] = src[(unsigned)(row * w4 + col)];
            row += 64 / NDW; col += 64 % NDW;
            if (col >= NDW) { col -= NDW; row++; }
        }
    } else {
        int row = lane / (4 * NDW), col = lane - row * (4 * NDW);
#pragma unroll 4
        for (int q = lane; q < ROWS * 4 * NDW; q += 64) {
            const int ry = CLAMP ? reflect101c(y0 + row, h) : reflect101(y0 + row, h), rx = CLAMP ? reflect101c(x0a + col, w) : reflect101(x0a + col, w);
            // (columns of the alignment lead-in / tail may reflect further than any window pixel does: only their address must be legal)
            dst[q] = img[(unsigned)(min(max(ry, 0), h - 1) * w + min(max(rx, 0), w - 1))];
            row += 64 / (4 * NDW); col += 64 % (4 * NDW);
            if (col >= 4 * NDW) { col -= 4 * NDW; row++; }
        }
    }
}
__device__ void lk_one_point(const LkImages &im, int maxLevel, float2 prevPtIn, float2 &nextPtIO, uint8_t &statusOut,
                             uint8_t *win /*LK_WIN_BYTES, 4-aligned*/, short2 *der /*22*22*/, uint8_t *jw /*LK_JW_BYTES, 4-aligned*/,
                             float *stats = nullptr /*debug: [0] iterations, [1] levels, [2] ticks before the iterations, [3] ticks in them*/) {
    const int WIN = VIO_WIN;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int lane = threadIdx.x & 63;
    const int prow = lane / 3, pcol = 7 * (lane - 3 * prow);   // first of this lane's seven window pixels
    const bool act = lane < 63;
    float2 nextPts = nextPtIO;
    uint8_t status = 1;
    int Ipk[4];    // I of the seven pixels, two 16-bit values per register
    int Dpk[7];    // (Ix, Iy) of the seven pixels
    for (int level = maxLevel; level >= 0; level--) {
        const gmem_u8 I = (gmem_u8)uniform_ptr(im.prev[level]), J = (gmem_u8)uniform_ptr(im.next[level]);
        const int w = __builtin_amdgcn_readfirstlane(im.w[level]), h = __builtin_amdgcn_readfirstlane(im.h[level]);
        float sc = (float)(1. / (1 << level));
        float2 prevPt = make_float2(prevPtIn.x * sc, prevPtIn.y * sc);
        float2 nextPt;
        if (level == maxLevel) nextPt = make_float2(nextPts.x * sc, nextPts.y * sc);  // OPTFLOW_USE_INITIAL_FLOW
        else nextPt = make_float2(nextPts.x * 2.f, nextPts.y * 2.f);
        nextPts = nextPt;
        const float halfWin = 10.f;
        prevPt.x -= halfWin; prevPt.y -= halfWin;
        const int ipx = cv_floor(prevPt.x), ipy = cv_floor(prevPt.y);
        if (ipx < -WIN || ipx >= w || ipy < -WIN || ipy >= h) {
            if (level == 0) status = 0;
            continue;
        }
        float a = prevPt.x - ipx, b = prevPt.y - ipy;
        int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        nextPt.x -= halfWin; nextPt.y -= halfWin;
        // the 22x22 window of J comes out of a (22 + 2 LK_MARGIN)^2 region cached in LDS around the first position of this level
        // (re-centred -- one more pass over global memory -- only when the iteration walks out of it).  Its loads are issued together
        // with the template's so that the two global round trips overlap each other and the derivative / template arithmetic
        int rx0 = cv_floor(nextPt.x) - LK_MARGIN, ry0 = cv_floor(nextPt.y) - LK_MARGIN;
        bool reg_ok;
        {
            const int inx = rx0 + LK_MARGIN, iny = ry0 + LK_MARGIN;
            reg_ok = !(inx < -WIN || inx >= w || iny < -WIN || iny >= h);
        }
        const int wx0a = (ipx - 1) & ~3, wxo = (ipx - 1) - wx0a;   // aligned first column of the template block, lead-in bytes
        long long lt0 = stats ? VIO_CLOCK() : 0;
        __syncthreads();
        lk_stage<24, LK_WP / 4, false>(I, w, h, wx0a, ipy - 1, win, lane);
        if (reg_ok) lk_stage<LK_REG, LK_RP / 4, true>(J, w, h, rx0 & ~3, ry0, jw, lane);
        __syncthreads();
        if (VIO_TIMERS && stats && lane == 0) stats[4] += (float)(VIO_CLOCK() - lt0);
        if (lane < 44) {
            // Scharr derivatives of the 22 x 22 block, separably: lane -> (row lane / 2, 11 columns).  Per window column c of the three
            // rows: s[c] = 3 a + 10 b + 3 c (vertical smoothing), v[c] = c - a (vertical difference); then
            // Ix[x] = s[x + 2] - s[x], Iy[x] = 3 v[x] + 10 v[x + 1] + 3 v[x + 2] -- the same integers as the 3 x 3 stencil
            const int dy = lane >> 1, x0 = 11 * (lane & 1);
            const uint8_t *r0 = win + dy * LK_WP + wxo + x0, *r1 = r0 + LK_WP, *r2 = r1 + LK_WP;
            const int gy = ipy + dy;
            const bool yin = gy >= 0 && gy < h;
            int s0 = 3 * r0[0] + 10 * r1[0] + 3 * r2[0], v0 = r2[0] - r0[0];
            int s1 = 3 * r0[1] + 10 * r1[1] + 3 * r2[1], v1 = r2[1] - r0[1];
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const int a_ = r0[k + 2], c_ = r2[k + 2];
                const int s2 = 3 * a_ + 10 * r1[k + 2] + 3 * c_, v2 = c_ - a_;
                const int gx = ipx + x0 + k;
                short2 d = make_short2(0, 0);
                if (yin && gx >= 0 && gx < w) {   // derivative buffer has a constant-0 border
                    d.x = (short)(s2 - s0);
                    d.y = (short)(3 * v0 + 10 * v1 + 3 * v2);
                }
                der[dy * 22 + x0 + k] = d;
                s0 = s1; s1 = s2; v0 = v1; v1 = v2;
            }
        }
        __syncthreads();
        if (VIO_TIMERS && stats && lane == 0) stats[5] += (float)(VIO_CLOCK() - lt0);
        // per-lane partial sums stay in 32 bits: |ix|, |iy| <= 4080 (convex blends of Scharr sums), 7 products per lane
        int pA11 = 0, pA12 = 0, pA22 = 0;
        if (act) {
            const uint8_t *r = win + (prow + 1) * LK_WP + wxo + (pcol + 1);
            const short2 *d = der + prow * 22 + pcol;
            int t0 = r[0], t1 = r[LK_WP];
            short2 d0 = d[0], d1 = d[22];
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int u0 = r[k + 1], u1 = r[LK_WP + k + 1];
                const short2 e0 = d[k + 1], e1 = d[22 + k + 1];
                const int iv = descale(blend4(t0, u0, t1, u1, iw00, iw01, iw10, iw11), W_BITS - 5);
                const int ix = descale(blend4(d0.x, e0.x, d1.x, e1.x, iw00, iw01, iw10, iw11), W_BITS);
                const int iy = descale(blend4(d0.y, e0.y, d1.y, e1.y, iw00, iw01, iw10, iw11), W_BITS);
                pA11 += __mul24(ix, ix);
                pA12 += __mul24(ix, iy);
                pA22 += __mul24(iy, iy);
                if (k & 1) Ipk[k >> 1] |= iv << 16; else Ipk[k >> 1] = iv & 0xFFFF;
                Dpk[k] = (ix & 0xFFFF) | (iy << 16);
                t0 = u0; t1 = u1; d0 = e0; d1 = e1;
            }
        }
        const long long sA11 = wave_sum_i32x(pA11), sA12 = wave_sum_i32x(pA12), sA22 = wave_sum_i32x(pA22);
        float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * WIN * WIN);
        if (minEig < 1e-4f || D < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        float2 prevDelta = make_float2(0.f, 0.f);
        if (VIO_TIMERS && stats && lane == 0) { long long n_ = VIO_CLOCK(); stats[2] += (float)(n_ - lt0); stats[1] += 1.f; lt0 = n_; }
        for (int j = 0; j < 30; j++) {
            if (VIO_TIMERS && stats && lane == 0) stats[0] += 1.f;
            int inx = cv_floor(nextPt.x), iny = cv_floor(nextPt.y);
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                if (level == 0) status = 0;
                break;
            }
            a = nextPt.x - inx; b = nextPt.y - iny;
            iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            if (!(reg_ok && inx >= rx0 && iny >= ry0 && inx <= rx0 + 2 * LK_MARGIN && iny <= ry0 + 2 * LK_MARGIN)) {
                rx0 = inx - LK_MARGIN; ry0 = iny - LK_MARGIN;
                __syncthreads();
                lk_stage<LK_REG, LK_RP / 4, true>(J, w, h, rx0 & ~3, ry0, jw, lane);
                __syncthreads();
                reg_ok = true;
            }
            // |diff| <= 8160 (both patches are u8 * 32), |Ix|, |Iy| <= 4080: 7 products per lane and 8 lanes stay below 2^31
            int pb1 = 0, pb2 = 0;
            if (act) {
                const uint8_t *r = jw + (iny - ry0 + prow) * LK_RP + (inx - (rx0 & ~3) + pcol);
                int t0 = r[0], t1 = r[LK_RP];
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    const int u0 = r[k + 1], u1 = r[LK_RP + k + 1];
                    const int iv = (k & 1) ? (Ipk[k >> 1] >> 16) : (int)(short)(Ipk[k >> 1] & 0xFFFF);
                    const int diff = descale(blend4(t0, u0, t1, u1, iw00, iw01, iw10, iw11), W_BITS - 5) - iv;
                    pb1 += __mul24(diff, (int)(short)(Dpk[k] & 0xFFFF));
                    pb2 += __mul24(diff, Dpk[k] >> 16);
                    t0 = u0; t1 = u1;
                }
            }
            const long long sb1 = wave_sum_i32x(pb1), sb2 = wave_sum_i32x(pb2);
            float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            float2 delta = make_float2((float)((A12 * b2 - A22 * b1) * D), (float)((A12 * b1 - A11 * b2) * D));
            nextPt.x += delta.x; nextPt.y += delta.y;
            nextPts = make_float2(nextPt.x + halfWin, nextPt.y + halfWin);
            if ((double)delta.x * delta.x + (double)delta.y * delta.y <= 0.01 * 0.01) break;
            if (j > 0 && fabsf(delta.x + prevDelta.x) < 0.01 && fabsf(delta.y + prevDelta.y) < 0.01) {
                nextPts.x -= delta.x * 0.5f;
                nextPts.y -= delta.y * 0.5f;
                break;
            }
            prevDelta = delta;
        }
        if (VIO_TIMERS && stats && lane == 0) stats[3] += (float)(VIO_CLOCK() - lt0);
    }
    nextPtIO = nextPts;
    statusOut = status;
}

// grid (NP, S), 64 threads
#ifndef LK_WAVES
#define LK_WAVES 4
#endif
// grid: (nblk, S) -- or, XCD-aware (B.xcd_nb = nblk > 0), 1-D with 8 * ceil(S / 8) * nblk blocks: workgroup L runs on XCD L % 8 (observed
// placement, MI355X_MICROARCH.md), so sequence 8 * ((L / 8) / nblk) + L % 8 keeps ALL its feature blocks on one XCD and its two images
// (0.77 MB with the pyramid level) are fetched into ONE L2 instead of all of them; eight sequences are in flight at a time, one per XCD
__global__ __launch_bounds__(64, LK_WAVES) void fe_lk_kernel(Batch B) {
    const DevCfg &C = *B.cfg;
    int s, blk0, nblk;
    if (B.xcd_nb > 0) {
        const int L = (int)blockIdx.x, q = L >> 3;
        nblk = B.xcd_nb;
        const int sl = (q / nblk) * 8 + (L & 7);
        if (sl >= B.ns) return;
        s = sl + B.s0; blk0 = q % nblk;
    } else { s = blockIdx.y + B.s0; blk0 = blockIdx.x; nblk = gridDim.x; }
    FeSeq &fe = B.fe[s];
    if (fe.n_forw < 0 || blk0 >= fe.n_pts) return;
    __shared__ __attribute__((aligned(16))) uint8_t win[LK_WIN_BYTES];
    __shared__ short2 der[22 * 22];
    __shared__ __attribute__((aligned(16))) uint8_t jw[LK_JW_BYTES];
    int cur = fe.cur_buf, forw = fe.has_img ? (cur ^ 1) : cur;
    // the pyramid table lives in LDS: as a local struct indexed by the (run-time) level it was 104 bytes of scratch per lane, written by
    // every lane of every block (the stage kernel gets the table as a kernel argument and never had that)
    __shared__ LkImages im;
    size_t hw = (size_t)C.c.width * C.c.height;
    if (threadIdx.x < 4) {
        const int l = threadIdx.x;
        if (l == 0) {
            im.prev[0] = B.img + ((size_t)s * 2 + cur) * hw;
            im.next[0] = B.img + ((size_t)s * 2 + forw) * hw;
            im.w[0] = C.c.width; im.h[0] = C.c.height;
        } else if (l <= C.c.lk_max_level) {
            im.prev[l] = B.pyr + ((size_t)s * 2 + cur) * C.pyr_bytes + C.lvl_off[l];
            im.next[l] = B.pyr + ((size_t)s * 2 + forw) * C.pyr_bytes + C.lvl_off[l];
            im.w[l] = C.lvl_w[l]; im.h[l] = C.lvl_h[l];
        }
    }
    __syncthreads();
    const long long lk_t0 = (s == 0 && blk0 == 0 && threadIdx.x == 0) ? VIO_CLOCK() : 0;
    // the block count per sequence is capped (most of the NP track slots are empty): a block walks its features with stride nblk
    for (int i = blk0; i < fe.n_pts; i += nblk) {
        float2 np = B.forw_pts[(size_t)s * C.NP + i];
        uint8_t st;
        lk_one_point(im, C.c.lk_max_level, B.cur_pts[(size_t)s * C.NP + i], np, st, win, der, jw, (s == 0 && blk0 == 0) ? B.timings + 92 : nullptr);
        if (threadIdx.x == 0) {
            B.forw_pts[(size_t)s * C.NP + i] = np;
            B.lk_status[(size_t)s * C.NP + i] = st;
        }
        __syncthreads();
    }
    if (VIO_TIMERS && s == 0 && blk0 == 0 && threadIdx.x == 0) { B.timings[90] += (float)(VIO_CLOCK() - lk_t0); B.timings[91] += 1.f; }
}

// stand-alone variant for the stage test: explicit images, points from arrays
__global__ __launch_bounds__(64) void fe_lk_stage_kernel(LkImages im, int maxLevel, int n, const float2 *prevPts, float2 *nextPts,
                                                         uint8_t *status) {
    int i = blockIdx.x;
    if (i >= n) return;
    __shared__ __attribute__((aligned(16))) uint8_t win[LK_WIN_BYTES];
    __shared__ short2 der[22 * 22];
    __shared__ __attribute__((aligned(16))) uint8_t jw[LK_JW_BYTES];
    float2 np = nextPts[i];
    uint8_t st;
    lk_one_point(im, maxLevel, prevPts[i], np, st, win, der, jw);
    if (threadIdx.x == 0) { nextPts[i] = np; status[i] = st; }
}

// ------------------------------------------------------------------------------------------------ RANSAC pieces
namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double det3(const double *F) {
    return F[0] * (F[4] * F[8] - F[5] * F[7]) - F[1] * (F[3] * F[8] - F[5] * F[6]) + F[2] * (F[3] * F[7] - F[4] * F[6]);
}
__device__ int solve_cubic_det(double c3, double c2, double c1, double c0, double roots[3]) {
    double mx = fmax(fmax(fabs(c3), fabs(c2)), fmax(fabs(c1), fabs(c0)));
    if (mx == 0.0) return 0;
    if (fabs(c3) < 1e-12 * mx) {
        if (fabs(c2) < 1e-12 * mx) {
            if (fabs(c1) < 1e-12 * mx) return 0;
            roots[0] = -c0 / c1;
            return 1;
        }
        double disc = c1 * c1 - 4 * c2 * c0;
        if (disc < 0) return 0;
        double sq = sqrt(disc);
        roots[0] = (-c1 + sq) / (2 * c2);
        roots[1] = (-c1 - sq) / (2 * c2);
        return 2;
    }
    double a = c2 / c3, b = c1 / c3, c = c0 / c3;
    double Bd = 1.0 + fmax(fabs(a), fmax(fabs(b), fabs(c)));
    double lo = -Bd, hi = Bd;
    for (int i = 0; i < 100; i++) {
        double mid = 0.5 * (lo + hi);
        double f = ((mid + a) * mid + b) * mid + c;
        if (f > 0) hi = mid; else lo = mid;
    }
    double r = 0.5 * (lo + hi);
    for (int i = 0; i < 2; i++) {
        double f = ((r + a) * r + b) * r + c, fp = (3 * r + 2 * a) * r + b;
        if (fp != 0.0) r -= f / fp;
    }
    roots[0] = r;
    double p = a + r, q = b + r * p;
    double disc = p * p - 4 * q;
    if (disc < 0) return 1;
    double sq = sqrt(disc);
    roots[1] = (-p + sq) * 0.5;
    roots[2] = (-p - sq) * 0.5;
    return 3;
}
// The 7-point solver (cv::findFundamentalMat's minimal solver) for ONE minimal sample on ONE wavefront.
// Lane e = 9 r + c (e < 63) owns element (r, c) of the 7 x 9 epipolar system and goes through the Gauss-Jordan elimination with
// full pivoting element-wise: the pivot is a wave-wide arg-max (first maximum in row-major order, like the scalar scan), row and
// column exchanges are one LDS round trip, and every element performs exactly the scalar algorithm's operations on it
// (a *= inv in the pivot row, a -= f * (pivot-row element * inv) elsewhere) -- same roundings, hence the same null space bit for bit,
// in 7 short steps instead of a 120 us scalar elimination on private (scratch) arrays.
// ws: 96 doubles of LDS private to the wavefront.  Fout: 27 doubles.  Returns the number of models (uniform over the wavefront).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ int seven_point_wave(double a, double *ws, double *Fout) {
    const int lane = threadIdx.x & 63;
    const int r = lane / 9, c = lane - 9 * r;   // lane 63: r = 7, never eligible
    double *As = ws, *f1s = ws + 64, *f2s = ws + 73;
    double amax = lane < 63 ? fabs(a) : 0.0;
    for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
    const double tol = 1e-12 * amax;
    unsigned long long perm = 0x876543210ULL;   // nibble i = perm[i]
    int rank = 0;
    for (int i = 0; i < 7; i++) {
        double v = (lane < 63 && r >= i && c >= i) ? fabs(a) : -1.0;
        int idx = lane;
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(v, off, 64);
            const int oi = __shfl_xor(idx, off, 64);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        if (!(v > tol)) break;   // rank-deficient sample: the remaining columns are free
        const int pr = idx / 9, pc = idx - 9 * pr;
        wave_lds_sync();
        As[lane] = a;
        wave_lds_sync();
        const int sr = r == i ? pr : (r == pr ? i : r);
        const int scol = c == i ? pc : (c == pc ? i : c);
        if (pc != i) {
            const unsigned long long ni = (perm >> (4 * i)) & 15ULL, np_ = (perm >> (4 * pc)) & 15ULL;
            perm = (perm & ~((15ULL << (4 * i)) | (15ULL << (4 * pc)))) | (np_ << (4 * i)) | (ni << (4 * pc));
        }
        if (lane < 63) {
            const double ap = As[sr * 9 + scol];          // my element after the two exchanges
            const double inv = 1.0 / As[pr * 9 + pc];     // the pivot
            const double prow = As[pr * 9 + scol] * inv;  // scaled pivot-row element of my column
            const double f = As[sr * 9 + pc];             // my row's entry in the pivot column
            if (r == i) a = prow;
            else if (f != 0.0) a = ap - f * prow;
            else a = ap;
        }
        rank = i + 1;
    }
    wave_lds_sync();
    As[lane] = a;
    wave_lds_sync();
    if (lane < 9) {   // f1[perm[i]] = -A[i][7], f2[perm[i]] = -A[i][8] (i < rank); f1[perm[7]] = 1, f2[perm[8]] = 1; other free columns 0
        int i = 0;
        for (int k = 0; k < 9; k++) if ((int)((perm >> (4 * k)) & 15ULL) == lane) i = k;
        double v1 = 0.0, v2 = 0.0;
        if (i < rank) { v1 = -As[i * 9 + 7]; v2 = -As[i * 9 + 8]; }
        if (i == 7) v1 = 1.0;
        if (i == 8) v2 = 1.0;
        f1s[lane] = v1; f2s[lane] = v2;
    }
    wave_lds_sync();
    double f1[9], f2[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { f1[k] = f1s[k]; f2[k] = f2s[k]; }
    // det(f1 + l f2) = c0 + c1 l + c2 l^2 + c3 l^3 (every lane computes the same numbers)
    double c0 = det3(f1), c3 = det3(f2), c1 = 0, c2 = 0;
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
        double t[9];
#pragma unroll
        for (int q = 0; q < 9; q++) t[q] = (q / 3 == rr) ? f2[q] : f1[q];
        c1 += det3(t);
#pragma unroll
        for (int q = 0; q < 9; q++) t[q] = (q / 3 == rr) ? f1[q] : f2[q];
        c2 += det3(t);
    }
    double roots[3] = {0.0, 0.0, 0.0};
    const int nr = solve_cubic_det(c3, c2, c1, c0, roots);
    if (lane < 27) {
        const int k = lane / 9, q = lane - 9 * k;
        double fa = f1[0], fb = f2[0];
#pragma unroll
        for (int j = 1; j < 9; j++) if (q == j) { fa = f1[j]; fb = f2[j]; }
        const double rt = k == 0 ? roots[0] : (k == 1 ? roots[1] : roots[2]);
        if (k < nr) Fout[lane] = fa + rt * fb;
    }
    return nr;
}
__device__ __forceinline__ bool f_inlier(const double *f, double x1, double y1, double x2, double y2, double thr2) {
    double a = f[0] * x1 + f[1] * y1 + f[2], b = f[3] * x1 + f[4] * y1 + f[5], cc = f[6] * x1 + f[7] * y1 + f[8];
    double s2 = 1.0 / (a * a + b * b), d2 = x2 * a + y2 * b + cc;
    double a1 = f[0] * x2 + f[3] * y2 + f[6], b1 = f[1] * x2 + f[4] * y2 + f[7], c1 = f[2] * x2 + f[5] * y2 + f[8];
    double s1 = 1.0 / (a1 * a1 + b1 * b1), d1 = x1 * a1 + y1 * b1 + c1;
    double err = fmax(d1 * d1 * s1, d2 * d2 * s2);
    return err <= thr2;
}
__device__ int ransac_update_iters(double p, double ep, int modelPoints, int maxIters) {
    p = fmin(fmax(p, 0.), 1.);
    ep = fmin(fmax(ep, 0.), 1.);
    double num = fmax(1. - p, 2.2250738585072014e-308);
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)rint(num / denom);
}

// block-wide helpers (256 threads)
__device__ int block_exclusive_scan(const int *flags, int n, int *offs, int *scratch /*blockDim+1*/) {
    // each thread owns a contiguous chunk; returns total
    int nt = blockDim.x, t = threadIdx.x;
    int chunk = (n + nt - 1) / nt;
    int b = t * chunk, e = min(n, b + chunk);
    int sum = 0;
    for (int i = b; i < e; i++) sum += flags[i] ? 1 : 0;
    scratch[t] = sum;
    __syncthreads();
    if (t < 64) {  // exclusive scan of the per-thread partials by one wavefront (nt / 64 partials per lane, shuffle scan across lanes)
        const int per = (nt + 63) >> 6, base = t * per;
        int s0 = 0;
        for (int q = 0; q < per; q++) if (base + q < nt) s0 += scratch[base + q];
        int inc = s0;
        for (int off = 1; off < 64; off <<= 1) { int v = __shfl_up(inc, off, 64); if (t >= off) inc += v; }
        int acc = inc - s0;
        for (int q = 0; q < per; q++) if (base + q < nt) { int v = scratch[base + q]; scratch[base + q] = acc; acc += v; }
        if (t == 63) scratch[nt] = inc;
    }
    __syncthreads();
    int o = scratch[t];
    for (int i = b; i < e; i++) { offs[i] = o; o += flags[i] ? 1 : 0; }
    int total = scratch[nt];
    __syncthreads();
    return total;
}

// RANSAC over normalised correspondences held in LDS; writes status flags.  All threads of the block participate (4 wavefronts).
// The iterations are the sequential algorithm's (sample `it` is a function of `it` alone, the iteration bound adapts after every
// accepted model in iteration order); they are evaluated in rounds: every wavefront solves `per` minimal samples (one in the first
// round -- with mostly inliers the bound falls below the round size at once --, up to 4 while many iterations remain), the inliers of
// all their models are counted by all threads (integer counts), thread 0 then replays the acceptance logic in iteration order.
#define RS_MAXB 16   // samples per round at most
struct RansacShared {
    double F[RS_MAXB * 27];
    double ws[4][96];
    int nm[RS_MAXB];
    int cnt[RS_MAXB * 3];
    double bestF[9];
    int niters, maxGood, base, batch;
};
__device__ void ransac_block(const vio_config &c, int N, const double *X1, const double *Y1, const double *X2, const double *Y2,
                             int *status, RansacShared &R, int *iters_out, float *tm = nullptr) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, nw = blockDim.x >> 6;
    long long rt0 = (tm && t == 0) ? VIO_CLOCK() : 0;
#define RS_PH(k) do { if (VIO_TIMERS && tm && t == 0) { long long n_ = VIO_CLOCK(); tm[k] += (float)(n_ - rt0); rt0 = n_; } } while (0)
    const double thr = c.f_threshold / c.focal_length, thr2 = thr * thr;
    if (t == 0) { R.niters = c.ransac_max_iters; R.maxGood = 0; R.base = 0; R.batch = min(nw, 4); }
    __syncthreads();
    while (true) {
        const int base = R.base, niters = R.niters, batch = R.batch;
        if (base >= niters) break;
        for (int p = t; p < batch * 3; p += blockDim.x) R.cnt[p] = 0;
        if (wv < 4)
            for (int hh = wv; hh < batch; hh += min(nw, 4)) {
                const int it = base + hh;
                int nm = 0;
                if (it < niters) {
                    uint64_t sd = 0x5649464D41545258ULL + (uint64_t)it * 0xD1B54A32D192ED03ULL;
                    int idx[7];
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        bool dup;
                        int rr;
                        do {
                            rr = (int)(splitmix64(sd) % (uint64_t)N);
                            dup = false;
#pragma unroll
                            for (int j = 0; j < k; j++) dup |= (idx[j] == rr);
                        } while (dup);
                        idx[k] = rr;
                    }
                    const int r = lane / 9, cc = lane - 9 * r;
                    int my = idx[0];
#pragma unroll
                    for (int k = 1; k < 7; k++) if (r == k) my = idx[k];
                    const double x1 = X1[my], y1 = Y1[my], x2 = X2[my], y2 = Y2[my];
                    // A[i] = (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1)
                    const double u = cc < 3 ? x2 : (cc < 6 ? y2 : 1.0);
                    const int c3 = cc - 3 * (cc / 3);
                    const double w_ = c3 == 0 ? x1 : (c3 == 1 ? y1 : 1.0);
                    const double a = cc < 6 ? (c3 == 2 ? u : u * w_) : w_;
                    nm = seven_point_wave(a, R.ws[wv], &R.F[hh * 27]);
                }
                if (lane == 0) R.nm[hh] = nm;
            }
        __syncthreads();
        RS_PH(0);
        for (int i0 = 0; i0 < N; i0 += blockDim.x) {
            const int i = i0 + t;
            const bool have = i < N;
            const double x1 = have ? X1[i] : 0.0, y1 = have ? Y1[i] : 0.0, x2 = have ? X2[i] : 0.0, y2 = have ? Y2[i] : 0.0;
            for (int hh = 0; hh < batch; hh++) {
                const int nm = R.nm[hh];
                for (int m = 0; m < nm; m++) {
                    const bool good = have && f_inlier(&R.F[hh * 27 + m * 9], x1, y1, x2, y2, thr2);
                    const unsigned long long bal = __ballot(good);
                    if (lane == 0 && bal) atomicAdd(&R.cnt[hh * 3 + m], __popcll(bal));
                }
            }
        }
        __syncthreads();
        RS_PH(1);
        if (t == 0) {
            int ni = R.niters, mg = R.maxGood;
            for (int hh = 0; hh < batch; hh++) {
                int it = base + hh;
                if (it >= ni) break;
                for (int m = 0; m < R.nm[hh]; m++) {
                    int good = R.cnt[hh * 3 + m];
                    if (good > max(mg, 6)) {
                        mg = good;
                        for (int q = 0; q < 9; q++) R.bestF[q] = R.F[hh * 27 + m * 9 + q];
                        ni = ransac_update_iters(0.99, (double)(N - good) / N, 7, ni);
                    }
                }
            }
            R.niters = ni;
            R.maxGood = mg;
            R.base = base + batch;
            const int left = ni - (base + batch);
            R.batch = left >= 4 * RS_MAXB ? RS_MAXB : (left > 8 ? 8 : 4);
        }
        __syncthreads();
        RS_PH(2);
    }
    int mg = R.maxGood;
    for (int i = t; i < N; i += blockDim.x) status[i] = (mg > 0 && f_inlier(R.bestF, X1[i], Y1[i], X2[i], Y2[i], thr2)) ? 1 : 0;
    if (t == 0 && iters_out) *iters_out = R.niters;
    __syncthreads();
    RS_PH(3);
#undef RS_PH
}

}  // namespace

// stage test entry: RANSAC on float pixel correspondences (virtual pinhole), grid 1, 256 threads, dynamic LDS 4*8*n + 4*n
__global__ __launch_bounds__(256) void fe_ransac_stage_kernel(vio_config c, int n, const float2 *p1, const float2 *p2, uint8_t *status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *X1 = (double *)smem, *Y1 = X1 + n, *X2 = Y1 + n, *Y2 = X2 + n;
    int *st = (int *)(Y2 + n);
    __shared__ RansacShared R;
    double hc = c.width / 2.0, hr = c.height / 2.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        X1[i] = ((double)p1[i].x - hc) / c.focal_length; Y1[i] = ((double)p1[i].y - hr) / c.focal_length;
        X2[i] = ((double)p2[i].x - hc) / c.focal_length; Y2[i] = ((double)p2[i].y - hr) / c.focal_length;
        st[i] = 0;
    }
    __syncthreads();
    if (n >= 8) ransac_block(c, n, X1, Y1, X2, Y2, st, R, nullptr);
    for (int i = threadIdx.x; i < n; i += blockDim.x) status[i] = (uint8_t)st[i];
}

// ------------------------------------------------------------------------------------------------ fe_select
// grid S, 256 threads, dynamic LDS:  per point: cur(8) forw(8) un(8) id(4) cnt(4) flag(4) offs(4) perm(4) + 4 doubles
__global__ __launch_bounds__(256) void fe_select_kernel(Batch B) {
    const DevCfg &C = *B.cfg;
    const vio_config &c = C.c;
    const int s = blockIdx.x + B.s0, t = threadIdx.x, NP = C.NP;
    FeSeq &fe = B.fe[s];
    if (fe.n_forw < 0) return;
    const int publish = fe.pub_req;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *X1 = (double *)smem, *Y1 = X1 + NP, *X2 = Y1 + NP, *Y2 = X2 + NP;
    float2 *cur = (float2 *)(Y2 + NP), *forw = cur + NP, *un = forw + NP, *tmp2 = un + NP;
    int *id = (int *)(tmp2 + NP), *cnt = id + NP, *flag = cnt + NP, *offs = flag + NP, *perm = offs + NP, *tmpi = perm + NP;
    int *scratch = tmpi + NP;                 // 257
    int2 *acc = (int2 *)(scratch + 260);      // NP + NP (accepted + unstable)
    __shared__ RansacShared R;
    __shared__ int hw_s[64];  // cv::circle half-widths out of the global config
    for (int k = t; k < 64; k += blockDim.x) hw_s[k] = k <= c.min_dist ? C.circle_hw[k] : 0;
    __shared__ int sh_n, sh_nacc, sh_nun;
    __shared__ int gcount[VIO_MAX_CELLS];

    float2 *g_cur = B.cur_pts + (size_t)s * NP, *g_forw = B.forw_pts + (size_t)s * NP, *g_un = B.cur_un_pts + (size_t)s * NP;
    int *g_id = B.ids + (size_t)s * NP, *g_cnt = B.track_cnt + (size_t)s * NP;
    float2 *g_unst = B.unstable_pts + (size_t)s * NP;
    int n = fe.n_pts;
    FE_PH_INIT;
    const long long wg_t0 = t == 0 ? VIO_CLOCK() : 0;
    for (int i = t; i < n; i += blockDim.x) {
        cur[i] = g_cur[i]; forw[i] = g_forw[i]; un[i] = g_un[i]; id[i] = g_id[i]; cnt[i] = g_cnt[i];
    }
    if (t == 0) { sh_nun = 0; sh_nacc = 0; }
    __syncthreads();
    // ---- status / border culling (feature_tracker.cpp:313-329)
    if (n > 0) {
        const uint8_t *st = B.lk_status + (size_t)s * NP;
        for (int i = t; i < n; i += blockDim.x) {
            int ix = cv_round(forw[i].x), iy = cv_round(forw[i].y);
            bool inb = 1 <= ix && ix < c.width - 1 && 1 <= iy && iy < c.height - 1;
            bool ok = st[i] != 0;
            if (!ok && inb) { int k = atomicAdd(&sh_nun, 1); g_unst[k] = forw[i]; }  // order irrelevant (mask circles only)
            flag[i] = (ok && inb) ? 1 : 0;
        }
        __syncthreads();
        int total = block_exclusive_scan(flag, n, offs, scratch);
        // stable compaction through temporaries
        float2 c0, f0, u0; int i0, k0;
        for (int b = 0; b < n; b += blockDim.x) {
            int i = b + t;
            bool v = i < n && flag[i];
            if (v) { c0 = cur[i]; f0 = forw[i]; u0 = un[i]; i0 = id[i]; k0 = cnt[i]; }
            int o = v ? offs[i] : 0;
            __syncthreads();
            if (v) { cur[o] = c0; forw[o] = f0; un[o] = u0; id[o] = i0; cnt[o] = k0; }
            __syncthreads();
        }
        n = total;
    }
    for (int i = t; i < n; i += blockDim.x) cnt[i]++;  // :348-349
    __syncthreads();
    FE_PH(64);

    int n_deficit = 0;
    if (publish) {
        // ---- rejectWithF (:441-473)
        if (n >= 8) {
            double hc = c.width / 2.0, hr = c.height / 2.0;
            for (int i = t; i < n; i += blockDim.x) {
                double x, y;
                cam_lift(c, cur[i].x, cur[i].y, x, y);
                float ux = (float)(c.focal_length * x + hc), uy = (float)(c.focal_length * y + hr);
                X1[i] = ((double)ux - hc) / c.focal_length; Y1[i] = ((double)uy - hr) / c.focal_length;
                cam_lift(c, forw[i].x, forw[i].y, x, y);
                ux = (float)(c.focal_length * x + hc); uy = (float)(c.focal_length * y + hr);
                X2[i] = ((double)ux - hc) / c.focal_length; Y2[i] = ((double)uy - hr) / c.focal_length;
            }
            __syncthreads();
            FE_PH(65);
            ransac_block(c, n, X1, Y1, X2, Y2, flag, R, &fe.ransac_iters, s == 0 ? B.timings + 66 : nullptr);
#if VIO_TIMERS
            if (s == 0 && t == 0) fe_t0 = VIO_CLOCK();
#endif
            int total = block_exclusive_scan(flag, n, offs, scratch);
            float2 c0, f0, u0; int i0, k0;
            for (int b = 0; b < n; b += blockDim.x) {
                int i = b + t;
                bool v = i < n && flag[i];
                if (v) { c0 = cur[i]; f0 = forw[i]; u0 = un[i]; i0 = id[i]; k0 = cnt[i]; }
                int o = v ? offs[i] : 0;
                __syncthreads();
                if (v) { cur[o] = c0; forw[o] = f0; un[o] = u0; id[o] = i0; cnt[o] = k0; }
                __syncthreads();
            }
            n = total;
            FE_PH(70);
        }
        // ---- setMask (:173-208): sort by track_cnt desc (ties: original order), greedy keep with MIN_DIST circles
        for (int i = t; i < n; i += blockDim.x) {
            int ci = cnt[i], r = 0;
            for (int j = 0; j < n; j++) r += (cnt[j] > ci || (cnt[j] == ci && j < i)) ? 1 : 0;
            perm[r] = i;
        }
        __syncthreads();
        FE_PH(71);
        // Greedy keep in sorted order, 64 candidates at a time.  For a block of 64: all threads test every candidate against the
        // centres accepted by earlier blocks (4 threads per candidate) and build the 64 x 64 matrix "candidate j lies in the disk of
        // candidate i < j" of the block; wavefront 0 then replays the sequential decision on the scalar unit -- candidate j is kept
        // iff it is not blocked from before and no KEPT candidate i < j of its block covers it -- and appends the survivors in order.
        // Same decisions as the one-by-one walk over the mask image (mask(p) == 0 <=> p lies in the disk of an accepted centre).
        {
            int *blk = scratch;                          // [64] blocked by an earlier block's centre
            unsigned *cm = (unsigned *)(scratch + 64);   // [64][2] conflict bits within the block
            const int r = c.min_dist;
            int nacc = 0;
            for (int q0 = 0; q0 < n; q0 += 64) {
                const int nb = min(64, n - q0);
                const int j = t & 63, part = t >> 6;     // 4 threads per candidate (blockDim 256)
                if (t < 64) { blk[t] = 0; cm[2 * t] = 0; cm[2 * t + 1] = 0; }
                __syncthreads();
                int px = 0, py = 0;
                if (j < nb) { const int i = perm[q0 + j]; px = cv_round(forw[i].x); py = cv_round(forw[i].y); }
                if (j < nb) {
                    bool hit = false;
                    for (int k = part; k < nacc; k += 4) hit |= in_disk(hw_s, r, px, py, acc[k].x, acc[k].y);
                    if (part == 0 && B.fisheye) hit |= B.fisheye[(size_t)py * c.width + px] != 255;   // FISHEYE: mask starts as fisheye_mask (:175-176)
                    if (hit) atomicOr(&blk[j], 1);
                    unsigned bits = 0;
                    const int i0 = 16 * part;
                    for (int q = 0; q < 16; q++) {
                        const int i2 = i0 + q;
                        if (i2 < j) {
                            const int ii = perm[q0 + i2];
                            if (in_disk(hw_s, r, px, py, cv_round(forw[ii].x), cv_round(forw[ii].y))) bits |= 1u << q;
                        }
                    }
                    if (bits) atomicOr(&cm[2 * j + (part >> 1)], bits << (16 * (part & 1)));
                }
                __syncthreads();
                if (t < 64) {
                    const int bj = j < nb ? blk[j] : 1;
                    const unsigned c_lo = cm[2 * j], c_hi = cm[2 * j + 1];
                    unsigned long long keep = 0;   // uniform: bit j = candidate j kept
                    for (int q = 0; q < nb; q++) {
                        const unsigned lo = __builtin_amdgcn_readlane(c_lo, q), hi = __builtin_amdgcn_readlane(c_hi, q);
                        const int b_ = __builtin_amdgcn_readlane(bj, q);
                        const unsigned long long cq = ((unsigned long long)hi << 32) | lo;
                        if (!b_ && !(cq & keep)) keep |= 1ULL << q;
                    }
                    if (j < nb && ((keep >> j) & 1ULL)) {
                        const int pos = nacc + __popcll(keep & ((1ULL << j) - 1ULL));
                        acc[pos] = make_int2(px, py);
                        tmpi[pos] = perm[q0 + j];
                    }
                    nacc += __popcll(keep);
                    if (t == 0) sh_nacc = nacc;
                }
                __syncthreads();
                nacc = sh_nacc;
            }
            if (n == 0 && t == 0) sh_nacc = 0;
        }
        __syncthreads();
        FE_PH(72);
        int nk = sh_nacc;
        {
            float2 f0; int i0, k0;
            for (int b = 0; b < nk; b += blockDim.x) {
                int q = b + t;
                bool v = q < nk;
                if (v) { int i = tmpi[q]; f0 = forw[i]; i0 = id[i]; k0 = cnt[i]; }
                __syncthreads();
                if (v) { tmp2[q] = f0; offs[q] = i0; flag[q] = k0; }
                __syncthreads();
            }
            for (int q = t; q < nk; q += blockDim.x) { forw[q] = tmp2[q]; id[q] = offs[q]; cnt[q] = flag[q]; }
            __syncthreads();
        }
        n = nk;
        // circles at unstable points (:204-207)
        int nun = sh_nun;
        for (int k = t; k < nun; k += blockDim.x) acc[n + k] = make_int2(cv_round(g_unst[k].x), cv_round(g_unst[k].y));
        __syncthreads();
        // ---- per-grid tracked counts and deficit cells (:360-395)
        int n_max_cnt = c.max_cnt - n;
        if (n_max_cnt > 0) {
            for (int k = t; k < C.ncells; k += blockDim.x) gcount[k] = 0;
            __syncthreads();
            for (int i = t; i < n; i += blockDim.x) {
                int col = (int)forw[i].x / C.grid_w, row = (int)forw[i].y / C.grid_h;
                if (col == c.grid_cols) --col;
                if (row == c.grid_rows) --row;
                atomicAdd(&gcount[col + c.grid_cols * row], 1);
            }
            __syncthreads();
            if (t == 0) {
                for (int k = 0; k < C.ncells; k++) {
                    fe.grids_track_num[k] = gcount[k];
                    if (gcount[k] < C.grids_threshold && fe.grids_texture_status[k]) { fe.deficit_cells[n_deficit++] = k; fe.cell_ncand[k] = 0; }
                    else { fe.grids_texture_status[k] = 1; fe.cell_ncand[k] = -1; }
                }
            }
        } else if (t == 0) {
            for (int k = 0; k < C.ncells; k++) fe.cell_ncand[k] = -1;
        }
        // accepted centres to HBM for fe_add
        int2 *g_acc = B.accept_xy + (size_t)s * 2 * NP;
        for (int k = t; k < n + nun; k += blockDim.x) g_acc[k] = acc[k];
        if (t == 0) fe.n_accept = n + nun;
    } else if (t == 0) {
        for (int k = 0; k < C.ncells; k++) fe.cell_ncand[k] = -1;
        fe.n_accept = 0;
    }
    __syncthreads();
    for (int i = t; i < n; i += blockDim.x) { g_forw[i] = forw[i]; g_id[i] = id[i]; g_cnt[i] = cnt[i]; }
    if (t == 0) { fe.n_forw = n; fe.n_deficit = n_deficit; fe.n_unstable = sh_nun; if (VIO_TIMERS) B.fe_ticks[s * 4 + 0] = (float)(VIO_CLOCK() - wg_t0); }
    FE_PH(73);
}

// ------------------------------------------------------------------------------------------------ fe_fast
// FAST-9/16 (threshold 10) + 3x3 NMS on one grid cell ROI staged in LDS. grid (ncells, S), 256 threads.
namespace {
__device__ __forceinline__ int fast_score_lds(const uint8_t *p, int stride) {
    const int thr = 10;
    int v = p[0];
    int d[25];
    // quick reject first, on the four compass pixels alone: any 9-arc contains at least two of them
    d[0] = v - p[3 * stride]; d[4] = v - p[3]; d[8] = v - p[-3 * stride]; d[12] = v - p[-3];
    int nb = (d[0] < -thr) + (d[4] < -thr) + (d[8] < -thr) + (d[12] < -thr);
    int nd = (d[0] > thr) + (d[4] > thr) + (d[8] > thr) + (d[12] > thr);
    if (nb < 2 && nd < 2) return 0;
                                    d[1] = v - p[3 * stride + 1];   d[2] = v - p[2 * stride + 2];   d[3] = v - p[stride + 3];
                                    d[5] = v - p[-stride + 3];      d[6] = v - p[-2 * stride + 2];  d[7] = v - p[-3 * stride + 1];
                                    d[9] = v - p[-3 * stride - 1];  d[10] = v - p[-2 * stride - 2]; d[11] = v - p[-stride - 3];
                                    d[13] = v - p[stride - 3];      d[14] = v - p[2 * stride - 2];  d[15] = v - p[3 * stride - 1];
#pragma unroll
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int best = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
#pragma unroll
        for (int j = 1; j < 9; j++) { mn = min(mn, d[k + j]); mx = max(mx, d[k + j]); }
        best = max(best, max(mn, -mx));
    }
    return best > thr ? best - 1 : 0;
}

// FAST-9/16 + 3x3 non-maximum suppression on the ROI r of img, survivors to out[] in row-major order (cv::FAST's order).
// LDS (fast_lds_bytes): the ROI staged from 4-byte aligned columns (pitch tp), the score plane (pitch rw), one 64-bit word of NMS
// results per 64 interior pixels.  The ordered emission needs no serial pass: the interior pixels, in row-major order, are dealt to
// the four wavefronts in contiguous quarters; a wavefront writes the ballot word of each of its 64-pixel chunks and counts its
// survivors, and after one barrier re-walks its words with the running offset of the quarters before it.
__device__ int fast_cell(const uint8_t *img, int W, GridRect r, uint8_t *smem, uint32_t *out, int cap) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int rw = r.w, rh = r.h;
    const int xa = r.x & ~3, xo = r.x - xa, tp = (xo + rw + 3) & ~3;   // aligned first column, lead-in, tile pitch
    uint8_t *tile = smem, *score = tile + ((tp * rh + 15) & ~15);
    unsigned long long *words = (unsigned long long *)(score + ((rw * rh + 15) & ~15));
    __shared__ int wtot[4];
    if (!(W & 3) && !((size_t)img & 3)) {
        const uint32_t *s32 = (const uint32_t *)(img + (size_t)r.y * W + xa);
        const int W4 = W >> 2, tp4 = tp >> 2;
        for (int q = t; q < tp4 * rh; q += blockDim.x) {
            const int y = q / tp4, c4 = q - y * tp4;
            ((uint32_t *)tile)[q] = s32[(size_t)y * W4 + c4];
        }
    } else {
        for (int q = t; q < rw * rh; q += blockDim.x) {
            int y = q / rw, x = q - y * rw;
            tile[y * tp + xo + x] = img[(size_t)(r.y + y) * W + r.x + x];
        }
    }
    for (int q = t; q < ((rw * rh + 3) >> 2); q += blockDim.x) ((uint32_t *)score)[q] = 0;
    __syncthreads();
    const int iw = rw - 6, ih = rh - 6, npx = iw * ih;
    for (int q = t; q < npx; q += blockDim.x) {
        int y = q / iw, x = q - y * iw;
        score[(y + 3) * rw + x + 3] = (uint8_t)fast_score_lds(tile + (y + 3) * tp + xo + x + 3, tp);
    }
    __syncthreads();
    const int nchunk = (npx + 63) >> 6, cpw = (nchunk + 3) >> 2;   // 64-pixel chunks, chunks per wavefront
    int mine = 0;
    for (int ch = wv * cpw; ch < min(nchunk, (wv + 1) * cpw); ch++) {
        const int q = ch * 64 + lane;
        bool mx = false;
        if (q < npx) {
            const int y = q / iw, x = q - y * iw;
            const uint8_t *cpt = score + (y + 3) * rw + x + 3;
            const int sc = cpt[0];
            mx = sc && sc > cpt[-1] && sc > cpt[1] && sc > cpt[-rw - 1] && sc > cpt[-rw] && sc > cpt[-rw + 1] && sc > cpt[rw - 1] &&
                 sc > cpt[rw] && sc > cpt[rw + 1];
        }
        const unsigned long long bal = __ballot(mx);
        if (lane == 0) words[ch] = bal;
        mine += __popcll(bal);
    }
    if (lane == 0) wtot[wv] = mine;
    __syncthreads();
    int o = 0;
    for (int k = 0; k < wv; k++) o += wtot[k];
    const int total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    for (int ch = wv * cpw; ch < min(nchunk, (wv + 1) * cpw); ch++) {
        const unsigned long long bal = words[ch];
        if ((bal >> lane) & 1ULL) {
            const int q = ch * 64 + lane, y = q / iw, x = q - y * iw;
            const int pos = o + __popcll(bal & ((1ULL << lane) - 1ULL));
            if (pos < cap) out[pos] = (uint32_t)(x + 3) | ((uint32_t)(y + 3) << 12) | ((uint32_t)score[(y + 3) * rw + x + 3] << 24);
        }
        o += __popcll(bal);
    }
    __syncthreads();
    return total;
}
}  // namespace

__global__ __launch_bounds__(256) void fe_fast_kernel(Batch B) {
    const DevCfg &C = *B.cfg;
    int s = blockIdx.y + B.s0, cell = blockIdx.x;
    FeSeq &fe = B.fe[s];
    if (fe.n_forw < 0 || !fe.pub_req || fe.cell_ncand[cell] < 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GridRect r = C.rect[cell];
    int forw = fe.has_img ? (fe.cur_buf ^ 1) : fe.cur_buf;
    const uint8_t *img = B.img + ((size_t)s * 2 + forw) * (size_t)C.c.width * C.c.height;
    uint32_t *out = B.cand + ((size_t)s * C.ncells + cell) * VIO_FAST_CAP;
    int total = fast_cell(img, C.c.width, r, smem, out, VIO_FAST_CAP);
    if (threadIdx.x == 0) {
        if (total > VIO_FAST_CAP) { total = VIO_FAST_CAP; fe.overflow |= 4; }
        fe.cell_ncand[cell] = total;
    }
}

__global__ __launch_bounds__(256) void fe_fast_stage_kernel(const uint8_t *img, int W, GridRect r, uint32_t *out, int cap, int *count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int total = fast_cell(img, W, r, smem, out, cap);
    if (threadIdx.x == 0) *count = total;
}

// ------------------------------------------------------------------------------------------------ fe_add
#define FE_NEAR_CAP 192
// Per sequence, cells in order: mask filter (runByPixelsMask), top-k by response (replace-min scan), addPoints greedy;
// then cur <- forw, undistortedPoints, velocity, updateID, feature-map packaging in ascending id.
// grid S, 256 threads, dynamic LDS.
__global__ __launch_bounds__(256) void fe_add_kernel(Batch B, int gate) {
    const DevCfg &C = *B.cfg;
    const vio_config &c = C.c;
    const int s = blockIdx.x + B.s0, t = threadIdx.x, NP = C.NP, lane = t & 63, wv = t >> 6;
    FeSeq &fe = B.fe[s];
    if (fe.n_forw < 0) return;
    const int publish = fe.pub_req;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int hw_s[64];                          // cv::circle half-widths (min_dist <= 63) out of the global config
    for (int k = t; k < 64; k += blockDim.x) hw_s[k] = k <= c.min_dist ? C.circle_hw[k] : 0;
    int2 *acc = (int2 *)smem;                         // 2*NP accepted mask centres
    unsigned long long *pre = (unsigned long long *)(acc + 2 * NP);   // [4][FE_NEAR_CAP] int2: old centres near the cell a wavefront works on
    int *flag = (int *)(pre + 4 * FE_NEAR_CAP);       // VIO_FAST_CAP
    int *offs = flag + VIO_FAST_CAP;                  // VIO_FAST_CAP
    uint32_t *filt = (uint32_t *)(offs + VIO_FAST_CAP);  // VIO_FAST_CAP
    int *scratch = (int *)(filt + VIO_FAST_CAP);      // 260
    uint32_t *keep = (uint32_t *)(scratch + 260);     // 64
    int *pid_l = (int *)(keep + 64);                  // NP   previous frame's id -> undistorted point map, staged for the velocity lookup
    float2 *pun_l = (float2 *)(pid_l + NP);           // NP
    __shared__ int sh_n;
    __shared__ int mcell[VIO_MAX_CELLS];              // candidates of a deficit cell that clear the old centres

    float2 *g_forw = B.forw_pts + (size_t)s * NP, *g_cur = B.cur_pts + (size_t)s * NP, *g_un = B.cur_un_pts + (size_t)s * NP,
           *g_vel = B.pts_velocity + (size_t)s * NP, *g_pun = B.prev_un_pt + (size_t)s * NP;
    int *g_id = B.ids + (size_t)s * NP, *g_cnt = B.track_cnt + (size_t)s * NP, *g_pid = B.prev_un_id + (size_t)s * NP;
    int n = fe.n_forw;
    FE_PH_INIT;
    const long long wg_t0 = t == 0 ? VIO_CLOCK() : 0;
    const int nprev = fe.n_prev_map;
    for (int k = t; k < nprev; k += blockDim.x) { pid_l[k] = g_pid[k]; pun_l[k] = g_pun[k]; }
    if (publish && fe.n_deficit > 0) {
        // gridDetect + addPoints, cell after cell in order (feature_tracker.cpp:105-171, 220-233, 397-409).  The mask a cell's detection
        // is filtered with = the centres accepted before the cell loop (setMask survivors + unstable points) + the points added by the
        // earlier cells.  Step 1, all wavefronts, a deficit cell each: drop the FAST candidates an OLD centre covers (the bulk of the
        // disk tests).  Step 2, wavefront 0 alone (no workgroup barriers), the cells in order: candidates that also clear
        // the NEW centres, compacted in row-major order (runByPixelsMask), top-k by response with the replace-min scan's slot semantics,
        // then addPoints among the survivors -- which by construction clear every centre accepted before them, so only their mutual
        // conflicts remain: a K x K bit matrix resolved in slot order.
        const int2 *g_acc = B.accept_xy + (size_t)s * 2 * NP;
        const int nacc0 = fe.n_accept, ndef = fe.n_deficit;
        for (int k = t; k < nacc0; k += blockDim.x) acc[k] = g_acc[k];
        if (t == 0) sh_n = n;
        __syncthreads();
        const int r = c.min_dist;
        // step 1: wavefront wv takes the deficit cells dc = wv, wv + 4, ...: the old centres whose disk can reach into the cell's
        // rectangle are gathered into a short list first (a cell sees about a tenth of them), the cell's candidates are tested
        // against that list and the survivors are compacted IN PLACE at the head of the cell's candidate array (row-major order kept:
        // a survivor never moves past its own position), their number goes to mcell[dc]
        {
            int2 *nearl = (int2 *)pre + wv * FE_NEAR_CAP;   // [4][FE_NEAR_CAP]
            for (int dc = wv; dc < ndef; dc += 4) {
                const int cell = fe.deficit_cells[dc];
                const GridRect rc = C.rect[cell];
                const int nc = fe.cell_ncand[cell];
                uint32_t *cand = B.cand + ((size_t)s * C.ncells + cell) * VIO_FAST_CAP;
                const int x_lo = rc.x - r, x_hi = rc.x + rc.w - 1 + r, y_lo = rc.y - r, y_hi = rc.y + rc.h - 1 + r;
                int nnear = 0;
                for (int a0 = 0; a0 < nacc0; a0 += 64) {
                    const int a = a0 + lane;
                    int2 q = make_int2(0, 0);
                    bool rel = false;
                    if (a < nacc0) { q = acc[a]; rel = q.x >= x_lo && q.x <= x_hi && q.y >= y_lo && q.y <= y_hi; }
                    const unsigned long long bal = __ballot(rel);
                    const int pos = nnear + __popcll(bal & ((1ULL << lane) - 1ULL));
                    if (rel && pos < FE_NEAR_CAP) nearl[pos] = q;
                    nnear += __popcll(bal);
                }
                wave_lds_sync();
                const bool use_all = nnear > FE_NEAR_CAP;   // (a crowd of centres around one cell: test against all of them)
                const int2 *lst = use_all ? acc : nearl;
                const int nl = use_all ? nacc0 : nnear;
                int m = 0;
                for (int ch = 0; ch * 64 < nc; ch++) {
                    const int k = ch * 64 + lane;
                    bool pass = false;
                    uint32_t v = 0;
                    if (k < nc) {
                        v = cand[k];
                        const int px = rc.x + (int)(v & 0xFFF), py = rc.y + (int)((v >> 12) & 0xFFF);
                        bool hit = false;
                        for (int a = 0; a < nl; a++) hit |= in_disk(hw_s, r, px, py, lst[a].x, lst[a].y);
                        if (B.fisheye) hit |= B.fisheye[(size_t)py * c.width + px] == 0;   // runByPixelsMask keeps mask != 0
                        pass = !hit;
                    }
                    const unsigned long long bal = __ballot(pass);
                    if (pass) cand[m + __popcll(bal & ((1ULL << lane) - 1ULL))] = v;
                    m += __popcll(bal);
                }
                if (lane == 0) mcell[dc] = m;
                wave_lds_sync();
            }
        }
        __threadfence_block();
        __syncthreads();
        FE_PH(80);
        if (t < 64) {
            int na = nacc0, nn = n;
            for (int dc = 0; dc < ndef; dc++) {
                const int cell = fe.deficit_cells[dc];
                const GridRect rc = C.rect[cell];
                const int nc = mcell[dc];
                const uint32_t *cand = B.cand + ((size_t)s * C.ncells + cell) * VIO_FAST_CAP;
                // ---- KeyPointsFilter::runByPixelsMask, second half: the centres added by the earlier cells
                int nf = 0;
                for (int ch = 0; ch * 64 < nc; ch++) {
                    const int k = ch * 64 + lane;
                    bool pass = k < nc;
                    uint32_t v = 0;
                    if (pass) {
                        v = cand[k];
                        const int px = rc.x + (int)(v & 0xFFF), py = rc.y + (int)((v >> 12) & 0xFFF);
                        bool hit = false;
                        for (int a = nacc0; a < na; a++) hit |= in_disk(hw_s, r, px, py, acc[a].x, acc[a].y);
                        pass = !hit;
                    }
                    const unsigned long long bal = __ballot(pass);
                    if (pass) filt[nf + __popcll(bal & ((1ULL << lane) - 1ULL))] = v;
                    nf += __popcll(bal);
                }
                wave_lds_sync();
                if (nf == 0) {  // :120-124
                    if (t == 0) fe.grids_texture_status[cell] = 0;
                    continue;
                }
                // ---- top-k by response, replace-min scan (:127-167); survivors stay in slot order
                const int K = C.grids_threshold - fe.grids_track_num[cell] + 2;
                int nk;
                uint32_t mine = 0xFFFFFFFFu;   // lane k owns slot k
                if (nf <= K) {
                    nk = nf;
                    if (nk > 64) {   // more survivors than a wavefront has lanes (never with grids_threshold + 2 <= 64): serial addPoints below
                        nk = -nf;
                    } else if (lane < nf) mine = filt[lane];
                } else if (K > 64) {
                    // never the case for the supported configurations (grids_threshold + 2 <= 64): serial replay of the scan, in place
                    if (t == 0) {
                        int min_id = 0;
                        for (int j = 0; j < nf; j++) {
                            uint32_t v = filt[j];
                            int resp = (int)(v >> 24);
                            if (j < K) { filt[j] = v; if (resp < (int)(filt[min_id] >> 24)) min_id = j; }
                            else if (resp > (int)(filt[min_id] >> 24)) {
                                filt[min_id] = v;
                                for (int k = 0; k < K; k++) if ((int)(filt[k] >> 24) < (int)(filt[min_id] >> 24)) min_id = k;
                            }
                        }
                    }
                    wave_lds_sync();
                    nk = -K;
                } else {
                    nk = K;
                    mine = lane < K ? filt[lane] : 0xFFFFFFFFu;
                    // key = response * 64 + slot: the minimum key is the first slot with the smallest response
                    auto wave_min_key = [&](uint32_t val) -> int {
                        int key = lane < K ? ((int)(val >> 24) << 6) | lane : 0x7FFFFFFF;
                        for (int off = 32; off > 0; off >>= 1) key = min(key, __shfl_xor(key, off, 64));
                        return key;
                    };
                    int mk = wave_min_key(mine);
                    int min_id = mk & 63, min_resp = mk >> 6;
                    // the scan only acts on candidates whose response beats the current minimum: find the next one 64 at a time
                    for (int j = K; j < nf;) {
                        const int jj = j + lane;
                        const uint32_t v = jj < nf ? filt[jj] : 0u;
                        const unsigned long long bal = __ballot(jj < nf && (int)(v >> 24) > min_resp);
                        if (!bal) { j += 64; continue; }
                        const int first = __builtin_ctzll(bal);
                        const uint32_t vv = (uint32_t)__builtin_amdgcn_readlane((int)v, first);
                        const int resp = (int)(vv >> 24);
                        if (lane == min_id) mine = vv;
                        const int nk2 = wave_min_key(mine);
                        int cand_id = nk2 & 63;
                        const int cand_resp = nk2 >> 6;
                        // the rescan starts at the replaced slot and only moves on a strictly smaller response
                        if (resp == cand_resp) cand_id = min_id;
                        min_id = cand_id; min_resp = cand_resp;
                        j += first + 1;
                    }
                }
                // ---- addPoints (:220-233): every survivor clears all centres accepted so far, only their mutual conflicts remain
                if (nk >= 0) {
                    const int px = lane < nk ? rc.x + (int)(mine & 0xFFF) : 0, py = lane < nk ? rc.y + (int)((mine >> 12) & 0xFFF) : 0;
                    unsigned long long cf = 0;   // bit i: slot i < my slot covers me
                    for (int i = 0; i < nk; i++) {
                        const int xi = __builtin_amdgcn_readlane(px, i), yi = __builtin_amdgcn_readlane(py, i);
                        if (i < lane && in_disk(hw_s, r, px, py, xi, yi)) cf |= 1ULL << i;
                    }
                    const unsigned c_lo = (unsigned)cf, c_hi = (unsigned)(cf >> 32);
                    // FISHEYE: addPoints wants mask == 255, the filter above only mask != 0
                    const unsigned long long grey = B.fisheye ? __ballot(lane < nk && B.fisheye[(size_t)py * c.width + px] != 255) : 0ULL;
                    unsigned long long kept = 0;
                    int room = NP - nn;
                    for (int q = 0; q < nk; q++) {
                        const unsigned long long cq = ((unsigned long long)__builtin_amdgcn_readlane(c_hi, q) << 32) | __builtin_amdgcn_readlane(c_lo, q);
                        if (!(cq & kept) && !((grey >> q) & 1ULL) && room > 0) { kept |= 1ULL << q; room--; }
                    }
                    if (lane < nk && ((kept >> lane) & 1ULL)) {
                        const int pos = __popcll(kept & ((1ULL << lane) - 1ULL));
                        acc[na + pos] = make_int2(px, py);
                        g_forw[nn + pos] = make_float2((float)px, (float)py); g_id[nn + pos] = -1; g_cnt[nn + pos] = 1;
                    }
                    const int added = __popcll(kept);
                    na += added; nn += added;
                } else {
                    // more than 64 survivors: the one-by-one walk (lanes test the accepted centres in parallel)
                    const int cnt = -nk;
                    for (int q = 0; q < cnt; q++) {
                        const uint32_t v = filt[q];
                        const int px = rc.x + (int)(v & 0xFFF), py = rc.y + (int)((v >> 12) & 0xFFF);
                        bool hit = false;
                        for (int k = nacc0 + lane; k < na; k += 64) hit |= in_disk(hw_s, r, px, py, acc[k].x, acc[k].y);
                        if (B.fisheye) hit |= B.fisheye[(size_t)py * c.width + px] != 255;
                        if (!__any(hit) && nn < NP) {
                            if (t == 0) { acc[na] = make_int2(px, py); g_forw[nn] = make_float2((float)px, (float)py); g_id[nn] = -1; g_cnt[nn] = 1; }
                            na++; nn++;
                        }
                        wave_lds_sync();
                    }
                }
                wave_lds_sync();
            }
            if (t == 0) sh_n = nn;
        }
        __syncthreads();
        n = sh_n;
    }
    __syncthreads();
    FE_PH(84);
    // ---- cur <- forw; undistortedPoints (:542-593); updateID (:485-495)
    double dt = fe.cur_time - fe.prev_time;
    int *newflag = flag, *newoff = offs;  // n <= NP <= VIO_FAST_CAP is checked at create time
    for (int i = t; i < n; i += blockDim.x) {
        float2 p = g_forw[i];
        g_cur[i] = p;
        double x, y;
        cam_lift(c, p.x, p.y, x, y);
        float2 u = make_float2((float)x, (float)y);
        g_un[i] = u;
        float2 vel = make_float2(0.f, 0.f);
        int idv = g_id[i];
        if (nprev > 0 && idv != -1) {
            for (int k = 0; k < nprev; k++)
                if (pid_l[k] == idv) {
                    double vx = (u.x - pun_l[k].x) / dt, vy = (u.y - pun_l[k].y) / dt;
                    vel = make_float2((float)vx, (float)vy);
                    break;
                }
        }
        g_vel[i] = vel;
        newflag[i] = idv == -1 ? 1 : 0;
    }
    __syncthreads();
    FE_PH(85);
    // prev_un_pts_map = cur_un_pts_map: ids as they are *before* updateID
    for (int i = t; i < n; i += blockDim.x) { g_pid[i] = g_id[i]; g_pun[i] = g_un[i]; }
    __syncthreads();
    int nnew = block_exclusive_scan(newflag, n, newoff, scratch);
    int nid0 = fe.n_id;
    for (int i = t; i < n; i += blockDim.x) if (newflag[i]) g_id[i] = nid0 + newoff[i];
    __syncthreads();
    FE_PH(86);
    // ---- feature-map packaging (estimator_nodelet.cpp:336-363): track_cnt > 1, ascending id (std::map order)
    int nobs = 0;
    if (publish) {
        for (int i = t; i < n; i += blockDim.x) newflag[i] = g_cnt[i] > 1 ? 1 : 0;
        __syncthreads();
        nobs = block_exclusive_scan(newflag, n, newoff, scratch);
        int *o_id = B.obs_id + (size_t)s * NP;
        double *o = B.obs + (size_t)s * NP * 7;
        // ids of the published features staged in LDS (the accepted-point list is no longer needed): the rank sort then runs on LDS
        // instead of n dependent global loads per thread
        int *ids_l = (int *)acc;
        for (int i = t; i < n; i += blockDim.x) ids_l[i] = newflag[i] ? g_id[i] : 0x7FFFFFFF;
        __syncthreads();
        for (int i = t; i < n; i += blockDim.x) {
            if (!newflag[i]) continue;
            int my = ids_l[i], rank = 0;
            for (int j = 0; j < n; j++) rank += (ids_l[j] < my) ? 1 : 0;
            o_id[rank] = my;
            double *q = o + (size_t)rank * 7;
            q[0] = g_un[i].x; q[1] = g_un[i].y; q[2] = 1.0; q[3] = g_cur[i].x; q[4] = g_cur[i].y; q[5] = g_vel[i].x; q[6] = g_vel[i].y;
        }
    }
    FE_PH(87);
    if (t == 0) {
        if (VIO_TIMERS) B.fe_ticks[s * 4 + 1] = (float)(VIO_CLOCK() - wg_t0);
        B.fe_ticks[s * 4 + 2] = (float)fe.n_deficit;
        B.fe_ticks[s * 4 + 3] = (float)fe.ransac_iters;
        fe.n_pts = n;
        fe.n_forw = n;
        fe.n_prev_map = n;
        fe.n_id = nid0 + nnew;
        fe.prev_time = fe.cur_time;
        if (fe.has_img) fe.cur_buf ^= 1;
        fe.has_img = 1;
        fe.n_obs = nobs;
        int ok = 0;
        if (publish) {
            if (!gate) ok = nobs > 0;
            else if (!fe.init_pub) fe.init_pub = 1;              // estimator_nodelet.cpp:365-368
            else if (!fe.init_feature) fe.init_feature = 1;      // :371-377
            else ok = nobs > 0;
        }
        fe.publish_ok = ok;
    }
}
