// HBM-resident state of a batch of S independent VIO sequences (DESIGN.md "Data layout in HBM").
// One instance of each per-sequence record per sequence; kernels index them with blockIdx.
// Reference state being mirrored: FeatureTracker members (feature_tracker.h:67-96), Estimator members
// (estimator.h:117-201), FeatureManager::feature (feature_manager.h:143), IntegrationBase (integration_base.h:197-216),
// MarginalizationInfo (marginalization_factor.h:51-76).
#pragma once
// In-kernel phase timers (tools/phase_profile.py, tools/fe_profile.py): compiled OUT of the default build.  `make -C csrc timers` builds
// libvio_hip_timers.so with -DVIO_PHASE_TIMERS; the tools load it through VIO_HIP_LIB.  VIO_CLOCK() is the only spelling of the clock read in
// the product kernels, and every accumulation into Batch::timings is behind VIO_TIMERS, so the default build carries neither.
#ifdef VIO_PHASE_TIMERS
#define VIO_TIMERS 1
#define VIO_CLOCK() ((long long)wall_clock64())
#else
#define VIO_TIMERS 0
#define VIO_CLOCK() 0LL
#endif
#include <cstddef>
#include <stdint.h>
#include "../../include/vio_abi.h"
#include "dmath.h"

#define VIO_MAXW 20
#define VIO_IMU_SLOT_CAP 64  // IMU samples kept per window slot for repropagate() (only read until initialisation)
#define VIO_OBS_D 9          // x y z u v vx vy cur_td depth  (FeaturePerFrame)
#define VIO_MAX_CELLS 64
#define VIO_FAST_CAP 1024    // NMS survivors kept per grid cell (before mask filter)
#define VIO_WIN 21           // LK window

struct GridRect { int x, y, w, h; };

// device-side copy of the configuration + derived constants
struct DevCfg {
    vio_config c;
    int W;              // window size
    int NP;             // capacity of the tracker point arrays
    int NL;             // landmark table capacity
    int lm_hash_size;   // power of two >= 2 * NL: id -> slot hash table built in LDS by be_ingest
    int NRES;           // residual slots per sequence: W * NP in-problem observations + 2 * NL list scratch at the tail
    int NIMU;           // IMU ring capacity
    int ncells;
    int grids_threshold;
    int grid_w, grid_h;
    GridRect rect[VIO_MAX_CELLS];
    int circle_hw[64];  // cv::circle(filled) half-widths for radius min_dist (min_dist <= 63)
    int P;              // tangent dimension 15(W+1)+7
    int NPRIOR;         // prior dimension 6W+16
    int LW;             // dense row stride of the landmark coupling matrix (= P rounded up to 16)
    int lvl_w[4], lvl_h[4], lvl_off[4];  // pyramid levels >= 1 packed in one buffer
    int pyr_bytes;
    int MX;             // marg_exact: largest marginalised block (15 + landmarks starting in frame 0) the scratch margE is sized for (0 = off)
    int MXL;            // marg_exact: largest block whose eigen-decomposition runs LDS-resident (Householder + implicit QL); larger ones use the HBM Jacobi
    int eig_jacobi;     // VIO_MARG_EIG_JACOBI: the eigen-decompositions of the literal marginalisation that do not fit LDS by cyclic Jacobi sweeps over HBM (rounds 3 - 5) instead of sym_eig_hbm
    int eig_one_wave;   // VIO_EIG_ONE_WAVE: the Householder tridiagonalisation of the LDS-resident eigen-decompositions on ONE wavefront (no workgroup barriers) instead of the whole workgroup
};

// IntegrationBase (integration_base.h)
struct PreInt {
    double lin_acc[3], lin_gyr[3], lin_ba[3], lin_bg[3];
    double acc0[3], gyr0[3];
    double dp[3], dq[4], dv[3], sum_dt;  // dq = (w,x,y,z)
    double jac[225];
    double sqrt_info[225];               // whitening matrix M (M^T M = cov^-1), refreshed whenever the covariance changes
    // ---- everything above (VIO_PREINT_HDR doubles) is what evaluating the IMU factor reads: ps_eval stages exactly that in LDS
    double cov[225];
    double dt_buf[VIO_IMU_SLOT_CAP], acc_buf[VIO_IMU_SLOT_CAP][3], gyr_buf[VIO_IMU_SLOT_CAP][3];
    int n_buf, valid;
};

#define VIO_PREINT_HDR 479
static_assert(offsetof(PreInt, cov) == VIO_PREINT_HDR * sizeof(double), "PreInt header layout");

// per-sequence tracker record (scalars); arrays live in TrackerArrays
struct FeSeq {
    double cur_time, prev_time;
    double R_rel[9];
    int n_pts;        // size of cur_pts / ids / track_cnt
    int n_forw;       // working count during a frame
    int n_unstable;
    int n_id;
    int has_img;
    int cur_buf;      // which ping-pong image buffer holds cur_img
    int n_prev_map;   // entries of prev_un_pts_map
    int first_image_flag, init_pub, init_feature;
    double last_image_time;
    int n_obs;        // packaged features for the back-end (track_cnt > 1), ascending id
    int publish_ok;   // 1 if the packaged features should be handed to processImage
    int n_deficit;
    int deficit_cells[VIO_MAX_CELLS];
    int grids_track_num[VIO_MAX_CELLS];
    int grids_texture_status[VIO_MAX_CELLS];
    int cell_ncand[VIO_MAX_CELLS];
    int n_accept;     // accepted mask centres (setMask survivors + unstable + added)
    int ransac_iters; // diagnostics
    int pub_req;      // PUB_THIS_FRAME of the current frame (estimator_nodelet.cpp:274-286), set by fe_begin from the caller's frame mode
    int overflow;     // front-end capacity flags of the current frame (bit 2: FAST candidates of a cell truncated)
    int use_R_rel;    // 1: R_rel was supplied by the caller (readImage(img, t, relative_R)), 0: predictMotion on the device
};

struct BeSeq {
    double Ps[VIO_MAXW + 1][3], Vs[VIO_MAXW + 1][3], Bas[VIO_MAXW + 1][3], Bgs[VIO_MAXW + 1][3];
    double Rs[VIO_MAXW + 1][9];
    double Headers[VIO_MAXW + 1];
    double acc_0[3], gyr_0[3], g[3], ric[9], tic[3], td;
    double latest_Bg[3];
    // what the tracker of the NEXT vio_feed call reads when the handle runs with tracker lag 1 (vio_set_tracker_lag): latest_Bg / td as
    // they were when this frame was ingested, and the IMU count a reboot decided by this frame's solve falls back to
    double track_Bg[3], track_td;
    int imu_count_ingest;
    double last_R[9], last_R0[9], last_P[3], last_P0[3], back_R0[9], back_P0[3];
    double para_Pose[VIO_MAXW + 1][7], para_SB[VIO_MAXW + 1][9], para_Ex[7], para_Td;
    double prevTime, cur_stamp;
    double initial_cost, final_cost;
    int first_imu, initFirstPoseFlag, openExEstimation, frame_count, solver_flag, marginalization_flag;
    int has_prior, prior_present[VIO_MAXW + 3];
    int n_lm, n_free, last_track_num, ring_base;
    int status_code, processed, reboot_count, frames_processed;
    int iterations, successful, n_in_problem, n_residuals, n_var_landmarks;
    int imu_head, imu_count;        // ring read index / number of samples ever pushed (absolute counters)
    int pre_idx[VIO_MAXW + 1];      // window slot -> physical PreInt (pointer swaps of slideWindow)
    int do_solve, do_marg;          // decisions of the ingest stage for the later kernels
    int n_imu_frame;                // samples consumed for the current frame
    int overflow;                   // capacity flags of the LAST processed frame (1 landmark table, 2 IMU slot, 8 residual list,
                                    // 16 IMU ring overwritten before it was consumed); cleared by be_ingest, see overflow_frames
    int overflow_frames;            // sticky diagnostic: number of frames that raised any capacity flag since the last reset / reboot
    int iter_total, solve_total;    // solver iterations / solves since vio_create (bench.py averages them over its timed steps)
    int rebooted;                   // failureDetection fired in this frame's solve: be_marg / be_finish skip the sequence
    int init_frame;                 // this frame completed the dynamic initialisation (set by the host): no failureDetection / movingConsistencyCheck
    int dyn_failed;                 // a dynamic initialisation attempt failed on this frame (diagnostics)
    int imu_frame_head;             // first ring sample consumed for the current frame (absolute index), with n_imu_frame
    double prior_c0;                // |r|^2 of the prior at its linearisation point (constant cost offset)
    // relocalisation (estimator.h:173-186): set by vio_set_relo_frame, consumed by the next solve
    int relo_info, relo_local, relo_index, relo_nmatch, relo_factors;
    double relo_stamp, relo_Pose[7], prev_relo_t[3], prev_relo_r[9];
    double relo_relative_t[3], relo_relative_q[4], relo_relative_yaw, drift_correct_t[3], drift_correct_r[9];
    int dbg[16];                    // debug counters (sweeps, ticks)
    // bounds-constrained solves (estimator.cpp:1282-1297), sticky since vio_create / vio_reset: inverse depths cut by the bound while a point was
    // formed, bounded landmarks that entered solves, trial evaluations and shortened steps of the projected Armijo line search (vio_get_bound_stats)
    int bound_clamps, bounded_solves, ls_evals, ls_contractions;
};

// flat parameter arrays of one solve (estimator.h para_Pose / para_SpeedBias / para_Ex_Pose / para_Td): pose = p(3) q(x,y,z,w)
// relo: relo_Pose (estimator.h:179), the copy of the matched window frame's pose that the relocalisation factors act on
struct Params { double pose[(VIO_MAXW + 1) * 7], sb[(VIO_MAXW + 1) * 9], ex[7], td, relo[7]; };

// Per-sequence state of the PHASED solver (be_phased.h): the trust-region loop of optimization() cut into kernels that each fill the
// whole GPU (evaluate / assemble / Schur) or run one workgroup per sequence (accept, Cholesky + dogleg); everything that the
// persistent kernel keeps in registers / LDS across phases lives here.
enum { PS_IDLE = 0, PS_EVAL_X0, PS_ASM, PS_SCHUR, PS_STEP, PS_EVAL_C, PS_DONE };
#define PS_MAX_EVAL_BLOCKS 64   // (ps_accept sums the partial costs one per lane of a wavefront)
// fused evaluate + assemble kernel (be_phased.h ps_evalf_kernel): residuals per projection workgroup, most workgroups / frame pairs per sequence
#define PS_FUSE_CAP 256
#define PS_FUSE_MAXBLK 12
#define PS_FUSE_MAXW 10
#define PS_FUSE_MAXPAIRS ((PS_FUSE_MAXW + 1) * PS_FUSE_MAXW / 2)
struct SolveSt {
    Params X, Xc;
    double sdx[6 * VIO_MAXW + 16], srp[6 * VIO_MAXW + 16];   // prior tangent / gradient at the last evaluated point
    double part[PS_MAX_EVAL_BLOCKS];                           // per-block partial costs of the last evaluation (block 0: prior + IMU)
    double cost, ccost, radius, mu, alpha, dogleg_norm, model_change;
    double step_xn2, step_dn2;   // |x|^2 and |candidate - x|^2 over the variable blocks (parameter tolerance), from ps_serial
    long long ts0;
    int stage;            // PS_*: what the sequence needs next
    int F, Fa, nres, ex_active, td_active, vext;
    int iter, iters_done, succ, invalid;
    int point_new;        // H / g were re-assembled since the last prepare_point
    int scale_pending;    // the Jacobi column scaling (sp, sl) is still to be fixed from the first linearisation
    int retry;            // the next step continues the same iteration (Cholesky failed, mu was raised)
    int cauchy_valid, eval_with_J, n_eval_blocks;
    int eval_done;        // blocks of the running evaluation that have published their partial cost (device-scope counter)
    int relo;             // this solve carries relocalisation factors: the (constant) extrinsic's six columns are lent to relo_Pose
    int test_fail;        // test hook (VIO_TEST_CHOL_FAIL_SHIFT): Cholesky factorisations of this solve still to be reported as failed
    // fused evaluate + assemble (ps_evalf_kernel): the projection residuals in landmark-aligned chunks of at most PS_FUSE_CAP, one workgroup each
    int fused;            // this solve runs it (compact records: extrinsic / td constant, no relocalisation factors; W <= PS_FUSE_MAXW)
    int rowbuf;           // which half of the double-buffered landmark rows (Hpl / Hll / gl) belongs to the CURRENT point: the fused kernel
                          // builds the rows of the point it evaluates into the other half, ps_accept flips when that point is taken
    int nblk;             // projection workgroups
    int blk_p[PS_FUSE_MAXBLK + 1], blk_r[PS_FUSE_MAXBLK + 1], blk_ka[PS_FUSE_MAXBLK + 1];   // chunk b: in-problem landmarks, residuals, variable landmarks [b], [b + 1])
    int fi[PS_FUSE_MAXBLK][PS_FUSE_MAXPAIRS];   // chunk b, frame pair p: first index into pair_list | residuals in the chunk << 16 | rank among the pair's chunks << 26
    int pm_np[PS_FUSE_MAXPAIRS];    // chunks that hold residuals of frame pair p (> 1: partial Gram blocks, summed by the last chunk to finish)
    int chunk_done;                 // chunks of the running evaluation whose Gram blocks are written (device-scope counter; the last one sums the partial blocks)
    int constrained;      // Program::IsBoundsConstrained(): a variable landmark carries the inverse-depth bound -> Ceres' projected line search
    int ls_pending;       // ps_serial has formed the alpha = 1 candidate of a constrained solve: ps_ls_kernel runs the search before the next ps_eval
};

// all HBM pointers of a batch; passed to kernels by value
struct Batch {
    DevCfg *cfg;  // device copy
    int S;
    int s0;               // first sequence of the launching group: kernels use s = blockIdx + s0
    int ns, xcd_nb, xcd_n; // sequences of this launch; blocks per sequence under the XCD-aware block map of the ps_* kernels (0: plain 2-D grid) -- be_phased.h ps_blk
    int tracker_lag;      // 0: fe_begin reads the live estimator state; 1: the snapshot be_ingest took one frame earlier
    int eval_rpt;         // projection residuals per thread of ps_eval (VIO_EVAL_RPT: 1 = 256 per workgroup; 2 halves the workgroups of the launch)
    FeSeq *fe;
    BeSeq *be;
    PreInt *pre;          // [S][W+2]
    // ---- tracker arrays, stride NP per sequence
    uint8_t *img;         // [S][2][H*W] ping-pong level 0
    uint8_t *pyr;         // [S][2][pyr_bytes] levels >= 1
    uint8_t *clahe_lut;   // vio_config.equalize only: [S][64 tiles][256] and the equalised frame [S][H*W]
    uint8_t *clahe_img;
    const uint8_t *fisheye;   // vio_set_fisheye_mask: [H*W] initial value of the feature mask (FISHEYE), nullptr = all 255
    float2 *cur_pts, *forw_pts, *cur_un_pts, *pts_velocity, *prev_un_pt, *unstable_pts;
    float2 *tmp_pts;
    int *ids, *track_cnt, *prev_un_id, *tmp_i0, *tmp_i1;
    uint8_t *lk_status;
    int2 *accept_xy;      // rounded mask circle centres
    uint32_t *cand;       // [S][ncells][VIO_FAST_CAP] packed x | y<<12 | score<<24 (ROI-relative)
    int *obs_id;          // [S][NP]
    double *obs;          // [S][NP][7]
    // ---- IMU ring [S][NIMU]
    double *imu_t, *imu_acc, *imu_gyr;
    // ---- landmark table, stride NL
    int *lm_id, *lm_start, *lm_nobs, *lm_est_flag, *lm_solve_flag, *lm_dyn, *lm_order, *lm_free, *lm_tmp;
    int *lm_pidx;         // index among in-problem landmarks (para_Feature index) or -1
    int *lm_aidx;         // index among variable landmarks or -1
    double *lm_depth;     // estimated_depth
    double *lm_obs;       // [S][NL][W+1][VIO_OBS_D], ring-indexed per frame
    int *lm_relo;         // [S][NL] 1: the landmark has a relocalisation factor in the current solve
    double *relo_xy;      // [S][NL][2] its matched point in the old keyframe
    double *relo_mp;      // [S][NP][3] match_points of vio_set_relo_frame: (x, y, feature id), ascending id
    double *para_feat;    // [S][NL] inverse depths (para_Feature)
    double *cand_feat;    // [S][NL]
    // ---- prior (canonical layout) per sequence
    // prior as a quadratic form: prior_H = A (n*n), prior_r = b (n), prior_x0 (W*7+17); prior_J / prior_rf receive the factored
    // form (linearized_jacobians / linearized_residuals) only when vio_get_prior asks for it
    double *prior_J, *prior_r, *prior_x0, *prior_H, *prior_rf;
    // ---- solver scratch per sequence
    double *H, *Sc, *Hpl;     // P*P, P*P, NL*LW
    double *vec;              // [S][VEC_SLOTS][LW]
    double *Hll, *gl, *lvec;  // [S][NL], [S][NL], [S][8][NL]
    double *res;              // per residual: weighted J (2x20) and r (2): [S][NRES][42]
    int *res_pair, *res_lm, *res_k;   // pair id / landmark slot / obs index per residual
    int *pair_start, *pair_list;      // counting sort by frame pair
    double *pairblk;                  // [S][npairs][210] packed symmetric 20x20
    unsigned char *ls_scratch;        // ps_ls_kernel: [S][ps_eval_lds_bytes(W)] staging region of the evaluation roles
    double *pairpart;                 // fused kernel: [S][PS_FUSE_MAXPAIRS][PS_FUSE_MAXBLK][210] partial Gram blocks of frame pairs whose residuals span chunks
    // the dimensions make_ctx needs, as kernel arguments (scalar registers) instead of a dependent load from *cfg at the top of every workgroup
    int gW, gP, gLW, gNL, gNP, gNRES, gNPRIOR, gMX;
    int gn_ext, n_schur;   // one more workgroup of the Schur launch (behind its n_schur tiles) forms the landmark term of the Gauss-Newton right-hand side (VIO_GN_EXT)
    int form_s;   // ps_asm_b_schur forms S = Sp (H - U) Sp + mu D^2 itself (windows on ps_serial_big, VIO_FORM_S)
    int fuse;                         // VIO_FUSE (default 1): solves that qualify run ps_evalf_kernel instead of ps_eval + ps_asm_a; value = chunks the grid covers
    int fuse_only;                    // this launch sequence carries no ps_eval / ps_asm_a (every solve of the handle's configuration qualifies)
    double *imu_raw;                  // [S][W][15*31] raw / whitened IMU Jacobians + residual
    double *margA, *margB, *margV, *margW;  // marginalisation workspaces
    double *margE;                          // marg_exact only: [S][3 MX^2 + NPRIOR MX] (A_mm, its eigenvectors, A_mm^+, A_rm A_mm^+)
    // ---- outputs
    double *odom;         // [S][11]
    double *odom_hist;    // [S][hist_cap][11]: one CSV row per processed NON_LINEAR frame (visualization.cpp:214-225)
    int *odom_count;      // [S]
    int hist_cap;
    int flags;            // debug switches (VIO_FLAGS): 1 = keep the Schur complement in HBM instead of LDS tiles
    float *timings;
    float *fe_ticks;      // [S][4] debug: 100 MHz ticks fe_select / fe_add / fe_fast(max cell) of each sequence spent in the last frame
    SolveSt *sst;         // [S] phased solver state
};

#define VEC_SLOTS 24
#define NRES_PER_LM VIO_MAXW
