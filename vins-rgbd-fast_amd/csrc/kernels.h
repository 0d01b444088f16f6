// Kernel entry points shared between the translation units of libvio_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include "vio_state.h"
#include "../../include/vio_synth.h"

// pipelined pre-integration (be_kernels.hip): samples per chunk, and the LDS doubles preint_propagate_many borrows from its caller
// (F 15x15 and V 15x18 of every sample of a chunk); the host sizes the marginalisation's dynamic LDS with the same macro
#define PI_CH 8
#define PREINT_MANY_LDS_DOUBLES (PI_CH * (225 + 270))
// dynamic LDS of fast_cell (fe_kernels.hip) for a rw x rh region: tile (pitch up to rw + 6) + score plane + NMS ballot words
static inline size_t fast_lds_bytes(int rw, int rh) {
    return (size_t)(((rw + 6) * rh + 15) & ~15) + (size_t)((rw * rh + 15) & ~15) + (size_t)(((rw - 6) * (rh - 6) + 63) / 64 + 2) * 8 + 16;
}
// dynamic LDS of one evaluation role of ps_eval (be_phased.h): frame-pair geometry / staged pre-integration headers
__host__ __device__ static inline size_t ps_eval_lds_bytes(int W) {
    const size_t W1 = (size_t)W + 1, a = (W1 * W1 + 1) * 32 * 8, b = (size_t)W * (VIO_PREINT_HDR + 1) * 8;
    return (a > b ? a : b) + 64;
}
// dynamic LDS of ps_ls_kernel: evaluation point + scalar workspace + partial costs (PS_LS_HEAD_DOUBLES); the evaluation roles' region is in HBM (Batch::ls_scratch)
static inline size_t ps_ls_lds_bytes(int W) { (void)W; return ((((sizeof(Params) / sizeof(double)) + 1) & ~(size_t)1) + 176) * 8 + 16; }
struct LkImages {
    const uint8_t *prev[4];
    const uint8_t *next[4];
    int w[4], h[4];
};

__global__ void fe_begin_kernel(Batch B, const double *stamps, int gate, int publish, const uint8_t *modes, const double *R_rel);
__global__ void be_latest_odometry_kernel(Batch B, int seq, double *out11);
__global__ void fe_predict_motion_kernel(Batch B, int seq, double t0, double t1, double *out9);
__global__ void fe_pyrdown_kernel(Batch B, const uint8_t *src_base, size_t src_stride, int sw, int sh, int dst_level, int write_level0);
__global__ void fe_pyrdown_stage_kernel(const uint8_t *src, int sw, int sh, uint8_t *dst);
__global__ void fe_clahe_lut_kernel(Batch B, const uint8_t *src_base, size_t stride);
__global__ void fe_clahe_apply_kernel(Batch B, const uint8_t *src_base, size_t stride);
__global__ void fe_clahe_lut_stage_kernel(const uint8_t *src, int W, int H, uint8_t *lut);
__global__ void fe_clahe_apply_stage_kernel(const uint8_t *src, int W, int H, const uint8_t *lut, uint8_t *dst);
__global__ void fe_predict_kernel(Batch B);
__global__ void fe_lk_kernel(Batch B);
__global__ void fe_lk_stage_kernel(LkImages im, int maxLevel, int n, const float2 *prevPts, float2 *nextPts, uint8_t *status);
__global__ void fe_ransac_stage_kernel(vio_config c, int n, const float2 *p1, const float2 *p2, uint8_t *status);
__global__ void fe_select_kernel(Batch B);
__global__ void fe_fast_kernel(Batch B);
__global__ void fe_fast_stage_kernel(const uint8_t *img, int W, GridRect r, uint32_t *out, int cap, int *count);
__global__ void fe_add_kernel(Batch B, int gate);
// feature-map source of be_ingest: ids == NULL selects the map packaged on the device by the front-end
struct IngestSrc {
    const int *n_obs;       // [S] entries per sequence (< 0: skip the sequence)
    const int *ids;         // [S][cap] ascending feature ids
    const double *obs;      // [S][cap][7] x y z u v vx vy
    const double *stamps;   // [S] header stamps
    int cap;
};
__global__ void be_ingest_kernel(Batch B, const uint16_t *depth_base, size_t depth_stride, IngestSrc src);
__global__ void be_solve_kernel(Batch B);
__global__ void be_solve_kernel_512(Batch B);
__global__ void be_marg_kernel(Batch B);
__global__ void be_marg_exact_kernel(Batch B);
// phased solver (be_phased.h)
__global__ void ps_setup_kernel(Batch B);
__global__ void ps_eval_kernel(Batch B);
__global__ void ps_eval_kernel_occ3(Batch B);
__global__ void ps_eval_kernel_occ4(Batch B);
__global__ void ps_asm_a_kernel(Batch B);
__global__ void ps_asm_a_kernel_occ4(Batch B);
__global__ void ps_asm_b_schur_kernel(Batch B, int nb_b, int by_blocks);
__global__ void ps_serial_big_kernel(Batch B);
__global__ void ps_serial_kernel_512(Batch B);
__global__ void ps_final_kernel(Batch B);
__global__ void ps_ls_kernel(Batch B);
__global__ void ps_evalf_kernel(Batch B);
// dynamic LDS of ps_evalf_kernel: a chunk's records + the frame-pair geometry (by pair slot), (+ the chunk's landmark table), or (workgroup 1) staged pre-integration headers + raw IMU
// Jacobians, or (workgroup 0) the prior's vectors
static inline size_t ps_evalf_lds_bytes(int W) {
    const size_t W1 = (size_t)W + 1, a = ((size_t)PS_FUSE_CAP * 28 + (W1 * W / 2 + 1) * 32) * 8 + (size_t)PS_FUSE_CAP * 12, b = ((size_t)W * (VIO_PREINT_HDR + 1) + (size_t)W * 465) * 8;
    return (a > b ? a : b) + 16;
}
__global__ void be_prior_factor_kernel(Batch B, int seq);
__global__ void be_set_relo_kernel(Batch B, int seq, const double *par);
__global__ void be_stage_pnp_kernel(const double *pts, int n, double *par6);
__global__ void be_dyn_finalize_kernel(Batch B, int seq, const double *samples, const int *offs);
__global__ void be_stage_imu_kernel(vio_config cfg, PreInt *P, int n, const double *dt, const double *acc, const double *gyr,
                                    const double *par, double g_norm, double *preint_out, double *r15, double *J480);
__global__ void be_stage_projection_kernel(vio_config cfg, const double *in, int use_td, int form, double *r2, double *J46);
__global__ void be_stage_imu_block_kernel(const PreInt *P, const double *par, double g_norm, double *G961);
__global__ void imu_scatter_kernel(Batch B, int total, const int *seq_of, const double *t, const double *acc, const double *gyr);
__global__ void imu_commit_kernel(Batch B, int total, const int *seq_of);
__global__ void synth_render_kernel(vio_synth_config c, int S, uint64_t seq0, const float *rays, const float *poses, uint8_t *gray, uint16_t *depth);
