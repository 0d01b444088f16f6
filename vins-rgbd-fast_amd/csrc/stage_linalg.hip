// Stage entry points of the dense FP64 linear algebra (be_linalg.h) on its own: the LDS-tile Cholesky + triangular solves of ps_serial
// (be_phased.h) on a caller-supplied matrix, with the in-kernel time of each part -- the parity anchor (numpy) and the tuning harness of
// that code.  Replaces nothing of the reference by itself: the reference reaches this arithmetic through Ceres' DENSE_SCHUR
// (estimator.cpp:1251-1263, Eigen LLT underneath).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/vio_abi.h"
#include "be_linalg.h"

namespace {

__global__ __launch_bounds__(512) void be_stage_chol_kernel(int nb, int reps, int blocks_mode, const double *S, const double *rhs, double *Lout, double *xout,
                                                             float *ticks /*[3]: factor, backward, wall ticks per microsecond*/) {
    extern __shared__ __attribute__((aligned(16))) double T[];
    const int t = threadIdx.x, nt = blockDim.x, n = 16 * nb, ntile = nb * (nb + 1) / 2;
    double *xs = T + (size_t)ntile * 256, *dinv = xs + n;
    __shared__ int flag;
    __shared__ float tm[4];
    if (t < 4) tm[t] = 0;
    long long tf = 0, tb = 0;
    bool ok = true;
    if (blocks_mode > 0) {
        // micro modes (tuning): 1 = chol_diag_tile alone on wavefront 0, 2 = the same with wavefront 4 (its SIMD mate) running trailing
        // updates beside it, 3 = one chol_panel_tile per wavefront, 4 = 8 tile updates per wavefront
        const int wave = t >> 6, lane = t & 63, li = lane & 15, lk = lane >> 4;
        for (int q = t; q < ntile * 256; q += nt) {
            int ti, tj;
            tri_decode(q >> 8, ti, tj);
            T[tl_idx(ti, tj, (q >> 4) & 15, q & 15)] = S[(size_t)(16 * ti + ((q >> 4) & 15)) * n + 16 * tj + (q & 15)];
        }
        __syncthreads();
        long long acc_t = 0;
        for (int rep = 0; rep < reps; rep++) {
            if (wave == 0 && (blocks_mode == 1 || blocks_mode == 2))
                for (int q = lane; q < 256; q += 64) T[q] = S[(size_t)(q >> 4) * n + (q & 15)] + (q >> 4 == (q & 15) ? 1.0 : 0.0);
            __syncthreads();
            const long long t0 = (long long)wall_clock64();
            if (blocks_mode == 1) { if (wave == 0) { const long long c0 = (long long)clock64(); chol_diag_tile(T, dinv, &flag); if (lane == 0) ticks[1] = (float)((long long)clock64() - c0); } }
            else if (blocks_mode == 2) {
                if (wave == 0) chol_diag_tile(T, dinv, &flag);
                else if (wave == 4 && nb >= 3)
                    for (int it = 0; it < 6; it++) {
                        v4f64 a;
                        for (int r = 0; r < 4; r++) a[r] = T[tl_idx(2, 1, lk + 4 * r, li)];
                        for (int kk = 0; kk < 4; kk++) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[tl_idx(2, 0, li, 4 * kk + lk)], T[tl_idx(1, 0, li, 4 * kk + lk)], a, 0, 0, 0);
                        for (int r = 0; r < 4; r++) T[tl_idx(2, 1, lk + 4 * r, li)] = a[r];
                    }
            } else if (blocks_mode == 4) {
                // 256 dependent FP64 fused multiply-adds, then 64 dependent v_rsq_f64, on wavefront 0; clock64 ticks into ticks[1..2]
                if (wave == 0) {
                    double v = T[lane];
                    const long long c0 = (long long)clock64();
#pragma unroll
                    for (int k = 0; k < 256; k++) v = __builtin_fma(v, 1.0000001, 1e-9);
                    const long long c1 = (long long)clock64();
#pragma unroll
                    for (int k = 0; k < 64; k++) v = __builtin_amdgcn_rsq(v + 2.0);
                    const long long c2 = (long long)clock64();
                    T[lane] = v;
                    if (lane == 0) { ticks[1] = (float)(c1 - c0); ticks[2] = (float)(c2 - c1); }
                }
            } else if (blocks_mode == 6) {
                // FP64 matrix-core timing on wavefront 0 (clock64 ticks): [1] 64 MFMAs chained on one accumulator, [2] 64 alternating
                // between two accumulators, [3] 32 x (MFMA -> v_mul on its result -> MFMA), [4] 64 dependent v_rsq_f64
                if (wave == 0) {
                    v4f64 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
                    double x = T[lane], y = T[64 + lane];
                    const long long c0 = (long long)clock64();
#pragma unroll
                    for (int k = 0; k < 64; k++) a = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    double s0 = a[0] + a[1] + a[2] + a[3];
                    const long long c1 = (long long)clock64();
#pragma unroll
                    for (int k = 0; k < 32; k++) { a = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a, 0, 0, 0); b = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, b, 0, 0, 0); }
                    __builtin_amdgcn_sched_barrier(0);
                    s0 += a[0] + b[0];
                    const long long c2 = (long long)clock64();
#pragma unroll
                    for (int k = 0; k < 32; k++) { a = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a, 0, 0, 0); x = a[0] * 1e-3; }
                    __builtin_amdgcn_sched_barrier(0);
                    s0 += a[0];
                    const long long c3 = (long long)clock64();
                    double z = s0 * 1e-300 + 2.0;
#pragma unroll
                    for (int k = 0; k < 64; k++) z = __builtin_amdgcn_rsq(z) + 1.5;
                    __builtin_amdgcn_sched_barrier(0);
                    const long long c4 = (long long)clock64();
                    T[lane] = s0 + z;
                    if (lane == 0) { ticks[1] = (float)(c1 - c0); ticks[2] = (float)(c2 - c1); ticks[3] = (float)(c3 - c2); ticks[4] = (float)(c4 - c3); }
                }
            } else if (blocks_mode == 3) { if (nb >= 4 && wave < 3) chol_panel_tile(T + ((size_t)((wave + 1) * (wave + 2) / 2) << 8), T, dinv); }
            __syncthreads();
            acc_t += (long long)wall_clock64() - t0;
        }
        if (t == 0 && blockIdx.x == 0) { ticks[0] = (float)acc_t / reps; if (blocks_mode != 4 && blocks_mode != 1 && blocks_mode != 6) { ticks[1] = ticks[2] = 0; } if (blocks_mode != 6) { ticks[3] = ticks[4] = 0; } }
        return;
    }
    for (int rep = 0; rep < reps; rep++) {
        for (int q = t; q < ntile * 256; q += nt) {
            int ti, tj;
            tri_decode(q >> 8, ti, tj);
            const int r = (q >> 4) & 15, c = q & 15;
            T[tl_idx(ti, tj, r, c)] = S[(size_t)(16 * ti + r) * n + 16 * tj + c];
        }
        for (int q = t; q < n; q += nt) xs[q] = rhs[q];
        __syncthreads();
        const long long t0 = (long long)wall_clock64();
        ok = chol_tiles(T, nb, &flag, dinv, tm, xs) && ok;
        __syncthreads();
        const long long t1 = (long long)wall_clock64();
        chol_backward_tiles_wave(T, nb, xs, dinv);
        __syncthreads();
        const long long t2 = (long long)wall_clock64();
        tf += t1 - t0; tb += t2 - t1;
    }
    if (blockIdx.x != 0) return;
    for (int q = t; q < ntile * 256; q += nt) {
        int ti, tj;
        tri_decode(q >> 8, ti, tj);
        const int r = (q >> 4) & 15, c = q & 15;
        if (ti == tj && c > r) continue;   // (the strict upper triangle of a diagonal tile holds L^-1 of the block)
        Lout[(size_t)(16 * ti + r) * n + 16 * tj + c] = T[tl_idx(ti, tj, r, c)];
    }
    for (int q = t; q < n; q += nt) xout[q] = ok ? xs[q] : nan("");
    if (t == 0) { ticks[0] = (float)tf / reps; ticks[1] = (float)tb / reps; for (int k = 0; k < 3; k++) ticks[2 + k] = tm[k] / reps; }
}

// the streaming factorisation of ps_serial_big (windows beyond 10 keyframes): tiles in HBM / L2, one block column at a time through LDS
__global__ __launch_bounds__(512) void be_stage_chol_stream_kernel(int nb, int reps, const double *S, const double *rhs, double *Tg, double *Lout, double *xout,
                                                                    float *ticks, int getenv_mode) {
    extern __shared__ __attribute__((aligned(16))) double colbuf[];
    const int t = threadIdx.x, nt = blockDim.x, n = 16 * nb, ntile = nb * (nb + 1) / 2;
    double *T = Tg + (size_t)blockIdx.x * ntile * 256;
    double *xs = colbuf + ((size_t)2 * nb + 1) * 256, *dinv = xs + n;   // two block columns + the look-ahead tile (chol_tiles_stream)
    __shared__ int flag;
    __shared__ float tm[4];
    if (t < 4) tm[t] = 0;
    long long tf = 0, tb = 0;
    bool ok = true;
    for (int rep = 0; rep < reps; rep++) {
        for (int q = t; q < ntile * 256; q += nt) {
            int ti, tj;
            tri_decode(q >> 8, ti, tj);
            T[tl_idx(ti, tj, (q >> 4) & 15, q & 15)] = S[(size_t)(16 * ti + ((q >> 4) & 15)) * n + 16 * tj + (q & 15)];
        }
        for (int q = t; q < n; q += nt) xs[q] = rhs[q];
        __threadfence_block();
        __syncthreads();
        const long long t0 = (long long)wall_clock64();
        ok = chol_tiles_stream(T, nb, colbuf, &flag, dinv, xs, tm) && ok;
        __syncthreads();
        const long long t1 = (long long)wall_clock64();
        if (ok) { if (getenv_mode) chol_backward_tiles(T, nb, xs, dinv); else chol_backward_tiles_wave(T, nb, xs, dinv); }   // (VIO_STAGE_BACKWARD_BLOCK: the all-wavefront walk, for comparison)
        __syncthreads();
        const long long t2 = (long long)wall_clock64();
        tf += t1 - t0; tb += t2 - t1;
    }
    if (blockIdx.x != 0) return;
    for (int q = t; q < ntile * 256; q += nt) {
        int ti, tj;
        tri_decode(q >> 8, ti, tj);
        const int r = (q >> 4) & 15, c = q & 15;
        if (ti == tj && c > r) continue;
        Lout[(size_t)(16 * ti + r) * n + 16 * tj + c] = T[tl_idx(ti, tj, r, c)];
    }
    for (int q = t; q < n; q += nt) xout[q] = ok ? xs[q] : nan("");
    __syncthreads();
    if (t == 0) { ticks[0] = (float)tf / reps; ticks[1] = (float)tb / reps; ticks[2] = tm[0] / reps; ticks[3] = tm[1] / reps; ticks[4] = tm[2] / reps; }
}

}  // namespace

#define ST_CHK(x) do { if ((x) != hipSuccess) { rc = VIO_EDEVICE; goto done; } } while (0)

// the streaming path (ps_serial_big): tiles in HBM, one block column in LDS; blocks = -7 forces it for nb <= 11 (bit-for-bit against the LDS path)
static int stage_chol_stream(int nb, int reps, int blocks, const double *S, const double *rhs, double *L_out, double *x_out, double *usec2) {
    int rc = VIO_OK;
    const size_t n = 16 * (size_t)nb, ntile = (size_t)nb * (nb + 1) / 2, lds = (((size_t)2 * nb + 1) * 256 + 2 * n) * sizeof(double);
    const int nblk = blocks < 0 ? 1 : blocks;
    double *dS = nullptr, *dr = nullptr, *dL = nullptr, *dx = nullptr, *dT = nullptr;
    float *dt = nullptr, ht[5] = {0, 0, 0, 0, 0};
    int rate_khz = 100000, dev = 0;
    ST_CHK(hipFuncSetAttribute((const void *)be_stage_chol_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ST_CHK(hipMalloc((void **)&dS, n * n * 8)); ST_CHK(hipMalloc((void **)&dr, n * 8)); ST_CHK(hipMalloc((void **)&dL, n * n * 8));
    ST_CHK(hipMalloc((void **)&dx, n * 8)); ST_CHK(hipMalloc((void **)&dt, 32)); ST_CHK(hipMalloc((void **)&dT, (size_t)nblk * ntile * 256 * 8));
    ST_CHK(hipMemcpy(dS, S, n * n * 8, hipMemcpyHostToDevice)); ST_CHK(hipMemcpy(dr, rhs, n * 8, hipMemcpyHostToDevice));
    ST_CHK(hipMemcpy(dL, L_out, n * n * 8, hipMemcpyHostToDevice));
    be_stage_chol_stream_kernel<<<nblk, 512, lds>>>(nb, reps, dS, dr, dT, dL, dx, dt, getenv("VIO_STAGE_BACKWARD_BLOCK") ? 1 : 0);
    ST_CHK(hipDeviceSynchronize());
    ST_CHK(hipMemcpy(L_out, dL, n * n * 8, hipMemcpyDeviceToHost)); ST_CHK(hipMemcpy(x_out, dx, n * 8, hipMemcpyDeviceToHost));
    ST_CHK(hipMemcpy(ht, dt, 20, hipMemcpyDeviceToHost));
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
    if (usec2) for (int k = 0; k < 5; k++) usec2[k] = ht[k] / (rate_khz * 1e-3);
done:
    if (dS) (void)hipFree(dS); if (dr) (void)hipFree(dr); if (dL) (void)hipFree(dL); if (dx) (void)hipFree(dx); if (dt) (void)hipFree(dt); if (dT) (void)hipFree(dT);
    return rc;
}

// Harness of sym_eig_hbm (be_linalg.h): one workgroup decomposes the symmetric n x n matrix A (row-major, n <= 512) in place in HBM.  evals[n] (unsorted),
// evecs[n * n] (evecs[i * n + k] = component i of the eigenvector of evals[k]), usec = the decomposition's time on the device.
namespace {
__global__ __launch_bounds__(512) void be_stage_sym_eig_kernel(double *A, int n, double *evals, float *ticks) {
    __shared__ double sred[64];
    extern __shared__ __attribute__((aligned(16))) double eig_wk[];
    const long long t0 = (long long)wall_clock64();
    sym_eig_hbm(A, n, n, eig_wk, sred, ticks);
    for (int j = threadIdx.x; j < n; j += blockDim.x) evals[j] = eig_wk[j];
    __syncthreads();
    if (threadIdx.x == 0) ticks[0] = (float)((long long)wall_clock64() - t0);
}
}  // namespace
extern "C" int vio_stage_sym_eig(int n, const double *A, double *evals, double *evecs, double *usec) {   // usec[4]: total, tridiagonalisation, accumulation, QL
    if (n < 2 || n > SYM_EIG_HBM_MAX || !A || !evals || !evecs) return VIO_EINVAL;
    int rc = VIO_OK;
    const size_t lds = (size_t)SYM_EIG_HBM_LDS_DOUBLES * sizeof(double);
    double *dA = nullptr, *dv = nullptr;
    float *dt = nullptr, ht[4] = {0, 0, 0, 0};
    int rate_khz = 100000, dev = 0;
    ST_CHK(hipMalloc((void **)&dA, (size_t)n * n * 8)); ST_CHK(hipMalloc((void **)&dv, (size_t)n * 8)); ST_CHK(hipMalloc((void **)&dt, 16));
    ST_CHK(hipMemcpy(dA, A, (size_t)n * n * 8, hipMemcpyHostToDevice));
    be_stage_sym_eig_kernel<<<1, 512, lds>>>(dA, n, dv, dt);
    ST_CHK(hipDeviceSynchronize());
    ST_CHK(hipMemcpy(evecs, dA, (size_t)n * n * 8, hipMemcpyDeviceToHost)); ST_CHK(hipMemcpy(evals, dv, (size_t)n * 8, hipMemcpyDeviceToHost));
    ST_CHK(hipMemcpy(ht, dt, 16, hipMemcpyDeviceToHost));
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
    if (usec) for (int k = 0; k < 4; k++) usec[k] = ht[k] / (rate_khz * 1e-3);
done:
    if (dA) (void)hipFree(dA); if (dv) (void)hipFree(dv); if (dt) (void)hipFree(dt);
    return rc;
}

// S: [16 nb][16 nb] row-major symmetric positive definite, rhs: [16 nb].  L_out (row-major, lower triangle written, the rest left as
// passed in), x_out = S^-1 rhs, usec5 = {factorisation + forward substitution, backward substitution, and of the factorisation as
// thread 0 sees it: panel phases, diagonal block + trailing update, barrier wait} in microseconds of one workgroup (mean over reps;
// `blocks` identical workgroups run side by side).
extern "C" int vio_stage_chol(int nb, int reps, int blocks, const double *S, const double *rhs, double *L_out, double *x_out, double *usec5) {
    double *usec2 = usec5;
    if (nb < 1 || nb > 24 || reps < 1 || blocks == 0 || blocks < -7 || !S || !rhs || !L_out || !x_out) return VIO_EINVAL;
    if (nb > 11 && blocks < 0 && blocks != -7) return VIO_EINVAL;
    int rc = VIO_OK;
    if (nb > 11 || blocks == -7) return stage_chol_stream(nb, reps, blocks, S, rhs, L_out, x_out, usec5);
    const size_t n = 16 * (size_t)nb, ntile = (size_t)nb * (nb + 1) / 2, lds = (ntile * 256 + 2 * n) * sizeof(double);
    double *dS = nullptr, *dr = nullptr, *dL = nullptr, *dx = nullptr;
    float *dt = nullptr, ht[5] = {0, 0, 0, 0, 0};
    int rate_khz = 100000, dev = 0;
    ST_CHK(hipFuncSetAttribute((const void *)be_stage_chol_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ST_CHK(hipMalloc((void **)&dS, n * n * 8)); ST_CHK(hipMalloc((void **)&dr, n * 8)); ST_CHK(hipMalloc((void **)&dL, n * n * 8));
    ST_CHK(hipMalloc((void **)&dx, n * 8)); ST_CHK(hipMalloc((void **)&dt, 32));
    ST_CHK(hipMemcpy(dS, S, n * n * 8, hipMemcpyHostToDevice)); ST_CHK(hipMemcpy(dr, rhs, n * 8, hipMemcpyHostToDevice));
    ST_CHK(hipMemcpy(dL, L_out, n * n * 8, hipMemcpyHostToDevice));
    be_stage_chol_kernel<<<blocks < 0 ? 1 : blocks, 512, lds>>>(nb, reps, blocks < 0 ? -blocks : 0, dS, dr, dL, dx, dt);
    ST_CHK(hipDeviceSynchronize());
    ST_CHK(hipMemcpy(L_out, dL, n * n * 8, hipMemcpyDeviceToHost)); ST_CHK(hipMemcpy(x_out, dx, n * 8, hipMemcpyDeviceToHost));
    ST_CHK(hipMemcpy(ht, dt, 20, hipMemcpyDeviceToHost));
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
    if (usec2) for (int k = 0; k < 5; k++) usec2[k] = ht[k] / (rate_khz * 1e-3);
done:
    if (dS) (void)hipFree(dS); if (dr) (void)hipFree(dr); if (dL) (void)hipFree(dL); if (dx) (void)hipFree(dx); if (dt) (void)hipFree(dt);
    return rc;
}
