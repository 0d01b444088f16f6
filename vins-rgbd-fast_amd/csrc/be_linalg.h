// Block / wavefront primitives and the dense FP64 linear algebra of the back-end kernels (included by be_kernels.hip only):
//   * DPP + v_readlane wavefront reductions, block sums / scans
//   * symmetric eigen-solvers (Jacobi for the 15x15 block, Householder tridiagonalisation + implicit QL for the prior)
//   * landmark Schur complement on v_mfma_f64_16x16x4_f64 (HBM, LDS-tile and LDS-staged variants)
//   * blocked Cholesky + triangular solves (LDS 16x16 tiles with XOR swizzle, HBM fallback for large windows)
//   * one-pass mat-vec helpers
// Everything is __device__ code for gfx950; no host fallbacks.
#pragma once
#include <hip/hip_runtime.h>
#include "vio_state.h"

namespace {

// ---- wavefront reductions on DPP + readlane instead of ds_bpermute shuffles (each __shfl_xor costs an LDS-crossbar round trip).
// Steps: quad_perm xor 1, xor 2, row_half_mirror, row_mirror leave the 16-lane row sum in every lane of the row; the four row
// sums are then read through SGPRs.  Fixed summation order -> deterministic.
__device__ __forceinline__ double dpp_f64(double v, const int ctrl_tag) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_tag) {
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break; // row_half_mirror
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break; // row_mirror
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rl_f64(double v, int src_lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_f64(v, 0);
    v += dpp_f64(v, 1);
    v += dpp_f64(v, 2);
    v += dpp_f64(v, 3);
    return (rl_f64(v, 0) + rl_f64(v, 16)) + (rl_f64(v, 32) + rl_f64(v, 48));
}
__device__ __forceinline__ double wave_max_dpp(double v) {
    v = fmax(v, dpp_f64(v, 0));
    v = fmax(v, dpp_f64(v, 1));
    v = fmax(v, dpp_f64(v, 2));
    v = fmax(v, dpp_f64(v, 3));
    return fmax(fmax(rl_f64(v, 0), rl_f64(v, 16)), fmax(rl_f64(v, 32), rl_f64(v, 48)));
}

// deterministic block-wide sum; all threads get the result. sred: >= 16 doubles of LDS.
// wavefront shuffle reduction, then a fixed-order sum over the per-wave partials (2 barriers).
__device__ __forceinline__ double block_sum(double v, double *sred) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum_dpp(v);
    __syncthreads();
    if (lane == 0) sred[wave] = v;
    __syncthreads();
    double r = 0;
    for (int k = 0; k < nw; k++) r += sred[k];
    return r;
}
// Two / three sums behind one pair of barriers (sred holds 64 doubles, at most 16 wavefronts): each value is reduced in exactly
// the order block_sum uses, so fusing reductions does not change a single bit.
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *sred) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = (blockDim.x + 63) >> 6;
    a = wave_sum_dpp(a); b = wave_sum_dpp(b); c = wave_sum_dpp(c);
    __syncthreads();
    if (lane == 0) { sred[wave] = a; sred[16 + wave] = b; sred[32 + wave] = c; }
    __syncthreads();
    double ra = 0, rb = 0, rc = 0;
    for (int k = 0; k < nw; k++) { ra += sred[k]; rb += sred[16 + k]; rc += sred[32 + k]; }
    a = ra; b = rb; c = rc;
}
__device__ __forceinline__ void block_sum2(double &a, double &b, double *sred) {
    double c = 0;
    block_sum3(a, b, c, sred);
}
__device__ __forceinline__ double block_max(double v, double *sred) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max_dpp(v);
    __syncthreads();
    if (lane == 0) sred[wave] = v;
    __syncthreads();
    double r = sred[0];
    for (int k = 1; k < nw; k++) r = fmax(r, sred[k]);
    return r;
}
__device__ int block_scan_flags(const int *flags, int n, int *offs, int *scratch /* 2*blockDim + 2 ints */) {
    int nt = blockDim.x, t = threadIdx.x;
    int chunk = (n + nt - 1) / nt;
    int b = t * chunk, e = min(n, b + chunk);
    int sum = 0;
    for (int i = b; i < e; i++) sum += flags[i];
    __syncthreads();
    int *cur = scratch, *nxt = scratch + nt;
    cur[t] = sum;
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {  // Hillis-Steele inclusive scan
        int v = cur[t];
        if (t >= off) v += cur[t - off];
        nxt[t] = v;
        __syncthreads();
        int *tmp = cur; cur = nxt; nxt = tmp;
    }
    int incl = cur[t];
    int total = cur[nt - 1];
    int o = incl - sum;
    for (int i = b; i < e; i++) { offs[i] = o; o += flags[i]; }
    __syncthreads();
    return total;
}

__device__ void jacobi_small(double *A, double *V, int n) {
    for (int i = 0; i < n * n; i++) V[i] = 0;
    for (int i = 0; i < n; i++) V[i * n + i] = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-30 * (diag + 1e-300) || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                double tau = (aqq - app) / (2.0 * apq);
                double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                double cs = 1.0 / sqrt(1.0 + tt * tt), sn = tt * cs;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = cs * akp - sn * akq;
                    A[k * n + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = cs * apk - sn * aqk;
                    A[q * n + k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = cs * vkp - sn * vkq;
                    V[k * n + q] = sn * vkp + cs * vkq;
                }
            }
    }
}

// Parallel two-sided Jacobi (round-robin ordering) on a symmetric n x n matrix A (ld = n), V = eigenvectors (columns).
// All threads of the block participate. cs/sn: LDS arrays of n/2+1 doubles; sred: blockDim doubles.
__device__ int jacobi_block(double *A, double *V, int n, int ld, double *cs, double *sn, int *pp, int *qq, double *sred) {
    int sweeps_done = 0;
    const int t = threadIdx.x, nt = blockDim.x;
    for (int i = t; i < n * n; i += nt) V[(i / n) * ld + (i % n)] = ((i / n) == (i % n)) ? 1.0 : 0.0;
    __syncthreads();
    const int m = (n + 1) & ~1;  // even number of players (one bye if n is odd)
    const int half = m / 2;
    for (int sweep = 0; sweep < 30; sweep++) {
        // threshold Jacobi: a pair is rotated only if |a_pq| > 1e-15 sqrt(|a_pp a_qq|) and > 1e-18 max|a_ii|;
        // the sweep loop ends when a whole sweep applied no rotation
        double dmax = 0;
        for (int i = t; i < n; i += nt) dmax = fmax(dmax, fabs(A[i * ld + i]));
        dmax = block_max(dmax, sred);
        __syncthreads();
        const double absfloor = 1e-18 * dmax;
        double nrot = 0;
        for (int round = 0; round < m - 1; round++) {
            // chess-tournament pairing: player m-1 fixed, others rotate
            if (t < half) {
                int a = (t == 0) ? m - 1 : (round + t) % (m - 1);
                int b = (round + m - 1 - t) % (m - 1);
                if (t == 0) b = round % (m - 1);
                int p = min(a, b), q = max(a, b);
                double c1 = 1.0, s1 = 0.0;
                if (q < n) {
                    double apq = A[p * ld + q];
                    double app = A[p * ld + p], aqq = A[q * ld + q];
                    if (fabs(apq) > absfloor && fabs(apq) > 1e-15 * sqrt(fabs(app * aqq))) {
                        nrot += 1.0;
                        double tau = (aqq - app) / (2.0 * apq);
                        double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c1 = 1.0 / sqrt(1.0 + tt * tt);
                        s1 = tt * c1;
                    }
                } else { p = -1; }
                pp[t] = p; qq[t] = q; cs[t] = c1; sn[t] = s1;
            }
            __syncthreads();
            // columns: A <- A J, V <- V J
            for (int w = t; w < half * n; w += nt) {
                int k = w / n, r = w - k * n;
                int p = pp[k], q = qq[k];
                if (p < 0 || sn[k] == 0.0) continue;
                double c1 = cs[k], s1 = sn[k];
                double akp = A[r * ld + p], akq = A[r * ld + q];
                A[r * ld + p] = c1 * akp - s1 * akq;
                A[r * ld + q] = s1 * akp + c1 * akq;
                double vkp = V[r * ld + p], vkq = V[r * ld + q];
                V[r * ld + p] = c1 * vkp - s1 * vkq;
                V[r * ld + q] = s1 * vkp + c1 * vkq;
            }
            __syncthreads();
            // rows: A <- J^T A
            for (int w = t; w < half * n; w += nt) {
                int k = w / n, cc = w - k * n;
                int p = pp[k], q = qq[k];
                if (p < 0 || sn[k] == 0.0) continue;
                double c1 = cs[k], s1 = sn[k];
                double apk = A[p * ld + cc], aqk = A[q * ld + cc];
                A[p * ld + cc] = c1 * apk - s1 * aqk;
                A[q * ld + cc] = s1 * apk + c1 * aqk;
            }
            __syncthreads();
        }
        nrot = block_sum(nrot, sred);
        sweeps_done = sweep + 1;
        if (nrot == 0.0) break;
    }
    __syncthreads();
    return sweeps_done;
}

// The same Jacobi iteration (same pairing, thresholds and rotation order as jacobi_block, hence the same result) run by ONE wavefront
// for n <= 16, A and V in LDS: a round is three dependent phases, and wave-level ordering instead of workgroup barriers is what
// makes the ~100 rounds of a 15 x 15 block cheap.  Call from a single wavefront; the caller synchronises the block afterwards.
#define JW_SYNC() do { __builtin_amdgcn_wave_barrier(); __threadfence_block(); } while (0)
__device__ int jacobi_wave16(double *A, double *V, int n, int ld, double *cs, double *sn, int *pp, int *qq) {
    const int lane = threadIdx.x & 63;
    int sweeps_done = 0;
    for (int i = lane; i < n * n; i += 64) V[(i / n) * ld + (i % n)] = ((i / n) == (i % n)) ? 1.0 : 0.0;
    JW_SYNC();
    const int m = (n + 1) & ~1, half = m / 2;
    for (int sweep = 0; sweep < 30; sweep++) {
        double dmax = lane < n ? fabs(A[lane * ld + lane]) : 0.0;
        dmax = wave_max_dpp(dmax);   // uniform over the wavefront
        const double absfloor = 1e-18 * dmax;
        double nrot = 0;
        for (int round = 0; round < m - 1; round++) {
            if (lane < half) {
                int a = (lane == 0) ? m - 1 : (round + lane) % (m - 1);
                int b = (round + m - 1 - lane) % (m - 1);
                if (lane == 0) b = round % (m - 1);
                int p = min(a, b), q = max(a, b);
                double c1 = 1.0, s1 = 0.0;
                if (q < n) {
                    double apq = A[p * ld + q];
                    double app = A[p * ld + p], aqq = A[q * ld + q];
                    if (fabs(apq) > absfloor && fabs(apq) > 1e-15 * sqrt(fabs(app * aqq))) {
                        nrot += 1.0;
                        double tau = (aqq - app) / (2.0 * apq);
                        double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c1 = 1.0 / sqrt(1.0 + tt * tt);
                        s1 = tt * c1;
                    }
                } else { p = -1; }
                pp[lane] = p; qq[lane] = q; cs[lane] = c1; sn[lane] = s1;
            }
            JW_SYNC();
            for (int w = lane; w < half * n; w += 64) {
                int k = w / n, r = w - k * n;
                int p = pp[k], q = qq[k];
                if (p < 0 || sn[k] == 0.0) continue;
                double c1 = cs[k], s1 = sn[k];
                double akp = A[r * ld + p], akq = A[r * ld + q];
                A[r * ld + p] = c1 * akp - s1 * akq;
                A[r * ld + q] = s1 * akp + c1 * akq;
                double vkp = V[r * ld + p], vkq = V[r * ld + q];
                V[r * ld + p] = c1 * vkp - s1 * vkq;
                V[r * ld + q] = s1 * vkp + c1 * vkq;
            }
            JW_SYNC();
            for (int w = lane; w < half * n; w += 64) {
                int k = w / n, cc = w - k * n;
                int p = pp[k], q = qq[k];
                if (p < 0 || sn[k] == 0.0) continue;
                double c1 = cs[k], s1 = sn[k];
                double apk = A[p * ld + cc], aqk = A[q * ld + cc];
                A[p * ld + cc] = c1 * apk - s1 * aqk;
                A[q * ld + cc] = s1 * apk + c1 * aqk;
            }
            JW_SYNC();
        }
        nrot = wave_sum_dpp(nrot);
        sweeps_done = sweep + 1;
        if (nrot == 0.0) break;
    }
    return sweeps_done;
}

// Inverse of a symmetric positive definite n x n matrix (n <= 16, LDS) by one wavefront: A = L L^T, L^-1 column by column, A^-1 = L^-T L^-1.
// Returns true only if the factorisation succeeded AND every eigenvalue of A is provably above `floor` (lambda_min >= 1 / |A^-1|_F):
// the caller wants the pseudo-inverse that drops eigenvalues <= 1e-8 (marginalization_factor.cpp:281-291), which IS the inverse in that
// case; otherwise it falls back to the eigen-decomposition.  L / Linv: n x n scratch in LDS, Ainv: result.
__device__ bool spd_inverse_wave16(const double *A, int n, double floor, double *L, double *Linv, double *Ainv) {
    const int lane = threadIdx.x & 63;
    // Factorisation in registers: lane i holds row i (n <= 16), the pivot and the entries L[k][j] travel through SGPRs (v_readlane).
    // The operations and their order per entry are those of the LDS version it replaces (three wavefront syncs per pivot: 42 us of
    // the 360 us marginalisation; now ~3): sqrt of the pivot, the column divided by it, a[i][k] -= L[i][j] * L[k][j] for k > j, i >= k.
    const int ri = lane < n ? lane : 0;
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = k < n ? A[ri * n + k] : 0.0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (j < n && ok) {
            const double d = rl_f64(a[j], j);
            if (!(d > 0.0) || !isfinite(d)) ok = false;   // wave-uniform
            else {
                const double l = sqrt(d);
                a[j] = lane > j ? a[j] / l : (lane == j ? l : a[j]);
#pragma unroll
                for (int k = j + 1; k < 16; k++) {
                    if (k < n) {
                        const double lkj = rl_f64(a[j], k);
                        const double prod = a[j] * lkj;
                        if (lane >= k) a[k] -= prod;
                    }
                }
            }
        }
    }
    if (!ok) return false;
    if (lane < n) {
#pragma unroll
        for (int k = 0; k < 16; k++) if (k < n) L[lane * n + k] = a[k];
    }
    JW_SYNC();
    if (lane < n) {   // column `lane` of L^-1
        // (fully unrolled with the loop bounds as predicates: x[] indexed by runtime loop variables lived in scratch memory, a
        // global-memory round trip per access -- most of the 40 us this routine took)
        const int c0 = lane;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            x[i] = 0.0;
            if (i < n) {
                double sacc = (i == c0) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; k++) if (k >= c0) sacc -= L[i * n + k] * x[k];
                x[i] = i < c0 ? 0.0 : sacc / L[i * n + i];
                Linv[i * n + c0] = x[i];
            }
        }
    }
    JW_SYNC();
    double fro = 0;
    for (int q = lane; q < n * n; q += 64) {
        const int i = q / n, j = q - i * n;
        double sacc = 0;
        for (int k = (i > j ? i : j); k < n; k++) sacc += Linv[k * n + i] * Linv[k * n + j];
        Ainv[q] = sacc;
        fro += sacc * sacc;
    }
    fro = wave_sum_dpp(fro);
    JW_SYNC();
    return isfinite(fro) && fro > 0.0 && 1.0 / sqrt(fro) > floor;
}

// Symmetric eigen-decomposition (Householder tridiagonalisation + implicit-shift QL, the algorithm class of
// Eigen::SelfAdjointEigenSolver used at marginalization_factor.cpp:277,298).  V (n x n, ld) holds A on entry (lower
// triangle is read) and the eigenvectors (columns) on exit; d = eigenvalues (unsorted), e / gtmp = workspaces; all in LDS.
// Everything runs in ONE wavefront: the algorithm is a chain of O(n) dependent steps, and wave-level ordering (no
// workgroup barriers) is what makes it fast; the other waves of the block wait at the final barrier.
__device__ __forceinline__ double wave_sum(double v) {
    return wave_sum_dpp(v);
}
#define WAVE_SYNC() do { __builtin_amdgcn_wave_barrier(); __threadfence_block(); } while (0)
__device__ __forceinline__ void sym_eig_tridiag(double *__restrict__ V, int n, int ld, double *__restrict__ d, double *__restrict__ e, double *__restrict__ gtmp, double *sred) {
    (void)sred;
    if (threadIdx.x < 64) {
        const int t = threadIdx.x, nt = 64;
        for (int j = t; j < n; j += nt) d[j] = V[(n - 1) * ld + j];
        WAVE_SYNC();
        for (int i = n - 1; i > 0; i--) {
            double sc = 0;
            for (int k = t; k < i; k += nt) sc += fabs(d[k]);
            sc = wave_sum(sc);
            if (sc == 0.0) {
                if (t == 0) e[i] = d[i - 1];
                WAVE_SYNC();
                for (int j = t; j < i; j += nt) { d[j] = V[(i - 1) * ld + j]; V[i * ld + j] = 0.0; V[j * ld + i] = 0.0; }
                if (t == 0) d[i] = 0.0;
                WAVE_SYNC();
                continue;
            }
            double h = 0;
            for (int k = t; k < i; k += nt) { double v = d[k] / sc; d[k] = v; h += v * v; }
            h = wave_sum(h);
            WAVE_SYNC();
            double f = d[i - 1];
            double g = sqrt(h);
            if (f > 0) g = -g;
            h -= f * g;
            WAVE_SYNC();
            if (t == 0) { e[i] = sc * g; d[i - 1] = f - g; }
            WAVE_SYNC();
            // e[0..i) = A_sub * d using the lower triangle; V[j][i] = d[j]
            for (int j = t; j < i; j += nt) {
                double acc = 0;
#pragma unroll 8
                for (int k = 0; k < i; k++) acc += (k <= j ? V[j * ld + k] : V[k * ld + j]) * d[k];
                V[j * ld + i] = d[j];
                e[j] = acc / h;
            }
            WAVE_SYNC();
            double ff = 0;
            for (int j = t; j < i; j += nt) ff += e[j] * d[j];
            ff = wave_sum(ff);
            double hh = ff / (h + h);
            for (int j = t; j < i; j += nt) e[j] -= hh * d[j];
            WAVE_SYNC();
            for (int k = t; k < i; k += nt) {  // row k of the lower triangle
                double dk = d[k], ek = e[k];
#pragma unroll 8
                for (int j = 0; j <= k; j++) V[k * ld + j] -= (d[j] * ek + e[j] * dk);
            }
            WAVE_SYNC();
            for (int j = t; j < i; j += nt) { d[j] = V[(i - 1) * ld + j]; V[i * ld + j] = 0.0; }
            if (t == 0) d[i] = h;
            WAVE_SYNC();
        }
        // accumulate the Householder transformations
        for (int i = 0; i < n - 1; i++) {
            if (t == 0) { V[(n - 1) * ld + i] = V[i * ld + i]; V[i * ld + i] = 1.0; }
            double h = d[i + 1];
            WAVE_SYNC();
            if (h != 0.0) {
                for (int k = t; k <= i; k += nt) d[k] = V[k * ld + i + 1] / h;
                WAVE_SYNC();
                for (int j = t; j <= i; j += nt) {
                    double g = 0;
#pragma unroll 8
                    for (int k = 0; k <= i; k++) g += V[k * ld + i + 1] * V[k * ld + j];
                    gtmp[j] = g;
                }
                WAVE_SYNC();
                for (int k = t; k <= i; k += nt) {
                    double dk = d[k];
#pragma unroll 8
                    for (int j = 0; j <= i; j++) V[k * ld + j] -= gtmp[j] * dk;
                }
                WAVE_SYNC();
            }
            for (int k = t; k <= i; k += nt) V[k * ld + i + 1] = 0.0;
            WAVE_SYNC();
        }
        for (int j = t; j < n; j += nt) { d[j] = V[(n - 1) * ld + j]; V[(n - 1) * ld + j] = 0.0; }
        WAVE_SYNC();
        if (t == 0) { V[(n - 1) * ld + n - 1] = 1.0; e[0] = 0.0; }
        WAVE_SYNC();
    }
    __syncthreads();
}

// Householder tridiagonalisation with accumulation (same recurrences as sym_eig_tridiag), spread over the whole workgroup:
// the O(i) reductions are computed redundantly by every wavefront (no writes, no barrier), the two O(i^2) pieces of each step --
// the symmetric matrix-vector product and the rank-2 update -- are split over all threads.  part: LDS [nw * EIG_LD].
#define EIG_LD (6 * VIO_MAXW + 16)
__device__ __forceinline__ void sym_eig_tridiag_mt(double *V, int n, int ld, double *d, double *e, double *gtmp, double *part) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6, nw = nt >> 6;
    for (int j = t; j < n; j += nt) d[j] = V[(n - 1) * ld + j];
    __syncthreads();
    for (int i = n - 1; i > 0; i--) {
        double sc = 0;
        for (int k = lane; k < i; k += 64) sc += fabs(d[k]);
        sc = wave_sum(sc);
        if (sc == 0.0) {
            __syncthreads();
            if (t == 0) { e[i] = d[i - 1]; }
            __syncthreads();
            for (int j = t; j < i; j += nt) { d[j] = V[(i - 1) * ld + j]; V[i * ld + j] = 0.0; V[j * ld + i] = 0.0; }
            if (t == 0) d[i] = 0.0;
            __syncthreads();
            continue;
        }
        double h = 0;
        for (int k = lane; k < i; k += 64) { double v = d[k] / sc; h += v * v; }
        h = wave_sum(h);
        const double f = d[i - 1] / sc;
        double g = sqrt(h);
        if (f > 0) g = -g;
        h -= f * g;
        // u[k] = scaled Householder vector (not yet stored): d[k] / sc, last entry f - g
        // partial products: lanes own rows j (and j + 64, ...), wavefronts split the k range
        {
            const int kc = (i + nw - 1) / nw, kb = wave * kc, ke = min(i, kb + kc);
            for (int j = lane; j < i; j += 64) {
                double acc = 0;
#pragma unroll 4
                for (int k = kb; k < ke; k++) {
                    double uk = (k == i - 1) ? (f - g) : d[k] / sc;
                    acc += (k <= j ? V[j * ld + k] : V[k * ld + j]) * uk;
                }
                part[wave * EIG_LD + j] = acc;
            }
        }
        __syncthreads();
        for (int j = t; j < i; j += nt) {
            double acc = 0;
            for (int q = 0; q < nw; q++) acc += part[q * EIG_LD + j];
            double uj = (j == i - 1) ? (f - g) : d[j] / sc;
            e[j] = acc / h;
            d[j] = uj;
            V[j * ld + i] = uj;
        }
        if (t == 0) e[i] = sc * g;
        __syncthreads();
        double ff = 0;
        for (int j = lane; j < i; j += 64) ff += e[j] * d[j];
        ff = wave_sum(ff);
        const double hh = ff / (h + h);
        __syncthreads();  // everybody has read e[] before it is updated
        for (int j = t; j < i; j += nt) e[j] -= hh * d[j];
        __syncthreads();
        for (int idx = t; idx < i * i; idx += nt) {
            int k = idx / i, j = idx - k * i;
            if (j <= k) V[k * ld + j] -= (d[j] * e[k] + e[j] * d[k]);
        }
        __syncthreads();
        for (int j = t; j < i; j += nt) { d[j] = V[(i - 1) * ld + j]; V[i * ld + j] = 0.0; }
        if (t == 0) d[i] = h;
        __syncthreads();
    }
    // accumulate the Householder transformations
    for (int i = 0; i < n - 1; i++) {
        if (t == 0) { V[(n - 1) * ld + i] = V[i * ld + i]; V[i * ld + i] = 1.0; }
        const double h = d[i + 1];
        __syncthreads();
        if (h != 0.0) {
            {
                const int m = i + 1, kc = (m + nw - 1) / nw, kb = wave * kc, ke = min(m, kb + kc);
                for (int j = lane; j < m; j += 64) {
                    double g = 0;
#pragma unroll 4
                    for (int k = kb; k < ke; k++) g += V[k * ld + i + 1] * V[k * ld + j];
                    part[wave * EIG_LD + j] = g;
                }
            }
            __syncthreads();
            for (int j = t; j <= i; j += nt) {
                double g = 0;
                for (int q = 0; q < nw; q++) g += part[q * EIG_LD + j];
                gtmp[j] = g;
            }
            __syncthreads();
            {
                const int m = i + 1;
                for (int idx = t; idx < m * m; idx += nt) {
                    int k = idx / m, j = idx - k * m;
                    V[k * ld + j] -= gtmp[j] * (V[k * ld + i + 1] / h);
                }
            }
            __syncthreads();
        }
        for (int k = t; k <= i; k += nt) V[k * ld + i + 1] = 0.0;
        __syncthreads();
    }
    for (int j = t; j < n; j += nt) { d[j] = V[(n - 1) * ld + j]; V[(n - 1) * ld + j] = 0.0; }
    __syncthreads();
    if (t == 0) { V[(n - 1) * ld + n - 1] = 1.0; e[0] = 0.0; }
    __syncthreads();
}

// implicit QL on the tridiagonal (d, e) with eigenvector accumulation into V; one wavefront, lanes own rows k, k+64, ...
__device__ __forceinline__ void tridiag_ql_wave(double *V, int n, int ld, double *d, double *e) {
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        for (int i = 1 + lane; i < n; i += 64) { double v = e[i]; __builtin_amdgcn_wave_barrier(); e[i - 1] = v; }
        // (shifting through LDS lane-parallel: read all first, then write)
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (lane == 0) e[n - 1] = 0.0;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        double f = 0.0, tst1 = 0.0;
        const double eps = 2.220446049250313e-16;
        for (int l = 0; l < n; l++) {
            tst1 = fmax(tst1, fabs(d[l]) + fabs(e[l]));
            int m = l;
            while (m < n) { if (fabs(e[m]) <= eps * tst1) break; m++; }
            if (m > l) {
                int iter = 0;
                do {
                    iter++;
                    double g = d[l];
                    double p = (d[l + 1] - g) / (2.0 * e[l]);
                    double r = sqrt(p * p + 1.0);
                    if (p < 0) r = -r;
                    double dl = e[l] / (p + r), dl1 = e[l] * (p + r);
                    double h = g - dl;
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) { d[l] = dl; d[l + 1] = dl1; }
                    for (int i = l + 2 + lane; i < n; i += 64) d[i] -= h;
                    __builtin_amdgcn_wave_barrier();
                    __threadfence_block();
                    f += h;
                    p = d[m];
                    double c = 1.0, c2 = 1.0, c3 = 1.0, el1 = e[l + 1], s = 0.0, s2 = 0.0;
                    const int r0 = lane < n ? lane : 0, r1 = lane + 64 < n ? lane + 64 : r0;  // n <= 128; duplicates write equal values
                    double car0 = V[r0 * ld + m], car1 = V[r1 * ld + m];
                    // software pipeline: e[i-1], d[i-1] and the next V column are fetched before this rotation's stores
                    double ei = e[m - 1], di = d[m - 1];
                    double vi0 = V[r0 * ld + m - 1], vi1 = V[r1 * ld + m - 1];
                    for (int i = m - 1; i >= l; i--) {
                        c3 = c2; c2 = c; s2 = s;
                        const int in = i > l ? i - 1 : i;
                        const double ei_n = e[in], di_n = d[in];
                        const double vn0 = V[r0 * ld + in], vn1 = V[r1 * ld + in];
                        g = c * ei;
                        h = c * p;
                        const double rr2 = p * p + ei * ei;
                        const double rinv = rr2 > 0.0 ? rsqrt(rr2) : 0.0;   // one slow op on the serial chain instead of sqrt + divide
                        r = rr2 * rinv;
                        double e_ip1 = s * r;
                        s = ei * rinv;
                        c = p * rinv;
                        p = c * di - s * g;
                        double d_ip1 = h + s * (c * g + s * di);
                        if (lane == 0) { e[i + 1] = e_ip1; d[i + 1] = d_ip1; }  // read again only after the sweep's fence
                        // rows lane and lane+64 of V: the rotated column i is carried in registers to the next rotation
                        V[r0 * ld + i + 1] = s * vi0 + c * car0;
                        V[r1 * ld + i + 1] = s * vi1 + c * car1;
                        car0 = c * vi0 - s * car0;
                        car1 = c * vi1 - s * car1;
                        ei = ei_n; di = di_n; vi0 = vn0; vi1 = vn1;
                    }
                    V[r0 * ld + l] = car0;
                    V[r1 * ld + l] = car1;
                    p = -s * s2 * c3 * el1 * e[l] / dl1;
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) { e[l] = s * p; d[l] = c * p; }
                    __builtin_amdgcn_wave_barrier();
                    __threadfence_block();
                } while (fabs(e[l]) > eps * tst1 && iter < 60);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) { d[l] = d[l] + f; e[l] = 0.0; }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
    }
    __syncthreads();
}

// in-place lower Cholesky of the n x n matrix A (ld), right-looking; returns false if a pivot is not positive
// Symmetric eigen-decomposition of a matrix that lives in HBM / L2, by the whole workgroup, n <= SYM_EIG_HBM_MAX (round 6: the blocks of the literal
// marginalisation, marg_exact = 1, that do not fit LDS -- m = 155 .. 197 on the canonical workload -- went through cyclic Jacobi sweeps over HBM, 50 - 100 ms
// each).  Householder tridiagonalisation with accumulation + implicit QL (EISPACK tred2 / tql2, the recurrences of sym_eig_tridiag / tridiag_ql_wave
// above), organised for a matrix that is NOT in LDS:
//   * every O(i^2) piece of a step is ONE THREAD PER COLUMN walking down the rows, thirty-two loads in flight per trip: a wavefront's load is then one
//     contiguous 512-byte row segment (a thread per row touches 64 cache lines per load instruction, and a runtime-bound loop with the load inside waits
//     for every L2 round trip in turn).  For that the active block is kept SYMMETRIC in both triangles (tred2 only needs the lower one; the mirror image
//     costs the rank-2 update twice the arithmetic and makes the matrix-vector product a column walk too; mirrored entries are the same bits);
//   * a QL sweep does not interleave the rotation recurrence with the eigenvector update: wavefront 0 runs the scalar recurrence of the whole sweep (d, e in
//     LDS) and leaves the rotations (c_i, s_i) in LDS, then every thread applies the sweep's rotations to ITS component of all eigenvectors -- on the
//     TRANSPOSED eigenvector array (row i = vector i), again a walk down the rows with the rotated element carried in a register.
// V: n x n row-major, leading dimension ld, the COMPLETE symmetric matrix on entry, the eigenvectors on exit (V[i * ld + k] = component i of vector k).
// wk: LDS, SYM_EIG_HBM_LDS_DOUBLES = 7 * SYM_EIG_HBM_MAX doubles; the eigenvalues (unsorted) are left in wk[0 .. n).  sred: 64 doubles of LDS.  blockDim >= 128.
#define SYM_EIG_HBM_MAX 512
#define SYM_EIG_HBM_LDS_DOUBLES (7 * SYM_EIG_HBM_MAX)
__device__ __forceinline__ void sym_eig_hbm_transpose(double *V, int n, int ld) {
    // in place: one thread per pair of 16-element row segments would be the coalesced way; at n <= 512 the plain element swap is 0.1 ms and runs twice
    const int t = threadIdx.x, nt = blockDim.x;
    for (int w0 = t; w0 < n * n; w0 += 4 * nt) {
        double a[4], b[4];
        int ia[4], ib[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int w = min(w0 + u * nt, n * n - 1), i = w / n, j = w - i * n;
            ia[u] = i * ld + j; ib[u] = j * ld + i;
            a[u] = V[ia[u]]; b[u] = V[ib[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int w = w0 + u * nt;
            if (w < n * n && ia[u] < ib[u]) { V[ia[u]] = b[u]; V[ib[u]] = a[u]; }   // (i < j: each pair once)
        }
    }
    __syncthreads();
}
__device__ __noinline__ void sym_eig_hbm(double *V, int n, int ld, double *wk, double *sred, float *tm = nullptr) {   // tm (harness): ticks of the tridiagonalisation, the accumulation, the QL phase
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63;
    long long tm0 = tm ? (long long)wall_clock64() : 0;
    double *d = wk, *e = wk + SYM_EIG_HBM_MAX, *gt = wk + 2 * SYM_EIG_HBM_MAX, *cs = wk + 3 * SYM_EIG_HBM_MAX, *sn = wk + 4 * SYM_EIG_HBM_MAX;
    for (int j = t; j < n; j += nt) d[j] = V[(size_t)(n - 1) * ld + j];
    __syncthreads();
    for (int i = n - 1; i > 0; i--) {
        double sc = 0;
        for (int k = t; k < i; k += nt) sc += fabs(d[k]);
        sc = block_sum(sc, sred);
        if (sc == 0.0) {
            const double dim1 = d[i - 1];
            __syncthreads();
            if (t == 0) e[i] = dim1;
            for (int j = t; j < i; j += nt) { d[j] = V[(size_t)(i - 1) * ld + j]; V[(size_t)i * ld + j] = 0.0; V[(size_t)j * ld + i] = 0.0; }
            if (t == 0) d[i] = 0.0;
            __syncthreads();
            continue;
        }
        double h = 0;
        for (int k = t; k < i; k += nt) { const double v = d[k] / sc; d[k] = v; h += v * v; }
        h = block_sum(h, sred);
        __syncthreads();
        const double f = d[i - 1];
        double g = sqrt(h);
        if (f > 0) g = -g;
        h -= f * g;
        __syncthreads();
        if (t == 0) { e[i] = sc * g; d[i - 1] = f - g; }
        __syncthreads();
        // e[j] = (A_sub d)[j] / h: column j of the symmetric active block, rows 0 .. i - 1
        for (int j = t; j < i; j += nt) {
            double acc = 0;
            for (int k0 = 0; k0 < i; k0 += 32) {
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; u++) v[u] = V[(size_t)min(k0 + u, i - 1) * ld + j];
#pragma unroll
                for (int u = 0; u < 32; u++) if (k0 + u < i) acc += v[u] * d[k0 + u];
            }
            e[j] = acc / h;
        }
        __syncthreads();
        double ff = 0;
        for (int j = t; j < i; j += nt) ff += e[j] * d[j];
        ff = block_sum(ff, sred);
        const double hh = ff / (h + h);
        __syncthreads();
        for (int j = t; j < i; j += nt) e[j] -= hh * d[j];
        __syncthreads();
        // rank-2 update of the whole active block (both triangles), column j per thread; row i - 1 is read back as the next d below
        for (int j = t; j < i; j += nt) {
            const double dj = d[j], ej = e[j];
            for (int k0 = 0; k0 < i; k0 += 32) {
                double v[32];
#pragma unroll
                for (int u = 0; u < 32; u++) v[u] = V[(size_t)min(k0 + u, i - 1) * ld + j];
#pragma unroll
                for (int u = 0; u < 32; u++) if (k0 + u < i) V[(size_t)(k0 + u) * ld + j] = v[u] - (dj * e[k0 + u] + ej * d[k0 + u]);
            }
        }
        __syncthreads();
        // the Householder vector goes to column i (rows j < i), row i is cleared, the next d is row i - 1 of the updated block
        for (int j = t; j < i; j += nt) { const double dj = d[j]; d[j] = V[(size_t)(i - 1) * ld + j]; V[(size_t)j * ld + i] = dj; V[(size_t)i * ld + j] = 0.0; }
        __syncthreads();
        if (t == 0) d[i] = h;
        __syncthreads();
    }
    if (tm && t == 0) { const long long n_ = (long long)wall_clock64(); tm[1] = (float)(n_ - tm0); tm0 = n_; }
    // accumulate the Householder transformations
    for (int i = 0; i < n - 1; i++) {
        if (t == 0) { V[(size_t)(n - 1) * ld + i] = V[(size_t)i * ld + i]; V[(size_t)i * ld + i] = 1.0; }
        const double h = d[i + 1];
        __syncthreads();
        if (h != 0.0) {
            for (int k = t; k <= i; k += nt) { const double u = V[(size_t)k * ld + i + 1]; cs[k] = u; d[k] = u / h; }   // (cs: the vector itself, free until the QL phase)
            __syncthreads();
            for (int j = t; j <= i; j += nt) {
                double g = 0;
                for (int k0 = 0; k0 <= i; k0 += 32) {
                    double b[32];
#pragma unroll
                    for (int u = 0; u < 32; u++) b[u] = V[(size_t)min(k0 + u, i) * ld + j];
#pragma unroll
                    for (int u = 0; u < 32; u++) if (k0 + u <= i) g += cs[k0 + u] * b[u];
                }
                // V[k][j] -= g d[k] down the same column
                for (int k0 = 0; k0 <= i; k0 += 32) {
                    double v[32];
#pragma unroll
                    for (int u = 0; u < 32; u++) v[u] = V[(size_t)min(k0 + u, i) * ld + j];
#pragma unroll
                    for (int u = 0; u < 32; u++) if (k0 + u <= i) V[(size_t)(k0 + u) * ld + j] = v[u] - g * d[k0 + u];
                }
            }
            __syncthreads();
        }
        for (int k = t; k <= i; k += nt) V[(size_t)k * ld + i + 1] = 0.0;
        __syncthreads();
    }
    for (int j = t; j < n; j += nt) { d[j] = V[(size_t)(n - 1) * ld + j]; V[(size_t)(n - 1) * ld + j] = 0.0; }
    __syncthreads();
    if (t == 0) { V[(size_t)(n - 1) * ld + n - 1] = 1.0; e[0] = 0.0; }
    __syncthreads();
    if (tm && t == 0) { const long long n_ = (long long)wall_clock64(); tm[2] = (float)(n_ - tm0); tm0 = n_; }
    // implicit QL on the transposed array: row i = basis vector i, thread k = component k
    sym_eig_hbm_transpose(V, n, ld);
    {
        double sh[(SYM_EIG_HBM_MAX + 63) / 64];   // e shifted down by one: read all, then write
#pragma unroll
        for (int q = 0; q < (SYM_EIG_HBM_MAX + 63) / 64; q++) sh[q] = 0;
        if (t < 64) {
#pragma unroll
            for (int q = 0; q < (SYM_EIG_HBM_MAX + 63) / 64; q++) { const int i = 1 + lane + 64 * q; sh[q] = i < n ? e[i] : 0.0; }
        }
        __syncthreads();
        if (t < 64) {
#pragma unroll
            for (int q = 0; q < (SYM_EIG_HBM_MAX + 63) / 64; q++) { const int i = 1 + lane + 64 * q; if (i < n) e[i - 1] = sh[q]; }
            if (lane == 0) e[n - 1] = 0.0;
        }
        __syncthreads();
    }
    // Wavefront 0 PRODUCES sweeps -- the whole QL control flow and the rotation recurrences, which touch only d and e -- and the other wavefronts APPLY them
    // to the eigenvector array one sweep behind (rotations double-buffered in LDS, one workgroup barrier per sweep: it publishes sweep k and, the appliers
    // arriving there after sweep k - 1, frees that sweep's buffer).  A sweep then costs max(recurrence, application) instead of their sum; the recurrence
    // (a dependent chain of ~400 cycles per rotation: reciprocal square root, eight multiply-adds) is the longer one.
    __shared__ int ql_desc[2][2];   // per buffer: l (-1 = no more sweeps), m
    const int wave = t >> 6;
    if (wave == 0) {
        int pb = 0;
        double f = 0.0, tst1 = 0.0;
        const double eps = 2.220446049250313e-16;
        for (int l = 0; l < n; l++) {
            tst1 = fmax(tst1, fabs(d[l]) + fabs(e[l]));
            int m = l;
            while (m < n) { if (fabs(e[m]) <= eps * tst1) break; m++; }
            if (m > l) {
                int iter = 0;
                bool again;
                do {
                    iter++;
                    const double g0 = d[l];
                    double p = (d[l + 1] - g0) / (2.0 * e[l]);
                    double r = sqrt(p * p + 1.0);
                    if (p < 0) r = -r;
                    const double dl = e[l] / (p + r), dl1 = e[l] * (p + r);
                    const double h = g0 - dl;
                    WAVE_SYNC();
                    if (lane == 0) { d[l] = dl; d[l + 1] = dl1; }
                    for (int i = l + 2 + lane; i < n; i += 64) d[i] -= h;
                    WAVE_SYNC();
                    f += h;
                    double *csb = cs + (size_t)pb * 2 * SYM_EIG_HBM_MAX, *snb = csb + SYM_EIG_HBM_MAX;
                    // e[l .. m) and d[l .. m) staged in registers (element l + q + 64 rg in lane q of register rg) and handed to the recurrence by lane
                    // broadcasts: no LDS read sits on the chain
                    const int len = m - l;
                    double er[SYM_EIG_HBM_MAX / 64], dr[SYM_EIG_HBM_MAX / 64];
#pragma unroll
                    for (int rg = 0; rg < SYM_EIG_HBM_MAX / 64; rg++) { const int q = lane + 64 * rg; er[rg] = q < len ? e[l + q] : 0.0; dr[rg] = q < len ? d[l + q] : 0.0; }
                    p = d[m];
                    double c = 1.0, c2 = 1.0, c3 = 1.0, s = 0.0, s2 = 0.0;
                    const double el1 = e[l + 1];
#pragma unroll
                    for (int rg = SYM_EIG_HBM_MAX / 64 - 1; rg >= 0; rg--) {
                        if (64 * rg >= len) continue;
                        for (int q = min(63, len - 1 - 64 * rg); q >= 0; q--) {
                            const int i = l + 64 * rg + q;
                            c3 = c2; c2 = c; s2 = s;
                            const double ei = rl_f64(er[rg], q), di = rl_f64(dr[rg], q);
                            const double g = c * ei;
                            const double hc = c * p;
                            const double rr2 = p * p + ei * ei;
                            const double rinv = rr2 > 0.0 ? rsqrt(rr2) : 0.0;
                            r = rr2 * rinv;
                            const double e_ip1 = s * r;
                            s = ei * rinv;
                            c = p * rinv;
                            p = c * di - s * g;
                            const double d_ip1 = hc + s * (c * g + s * di);
                            if (lane == 0) { e[i + 1] = e_ip1; d[i + 1] = d_ip1; csb[i] = c; snb[i] = s; }
                        }
                    }
                    p = -s * s2 * c3 * el1 * rl_f64(er[0], 0) / dl1;
                    WAVE_SYNC();
                    if (lane == 0) { e[l] = s * p; d[l] = c * p; ql_desc[pb][0] = l; ql_desc[pb][1] = m; }
                    WAVE_SYNC();
                    again = fabs(e[l]) > eps * tst1 && iter < 60;
                    __syncthreads();   // sweep published; the appliers are done with the other buffer
                    pb ^= 1;
                } while (again);
            }
            WAVE_SYNC();
            if (lane == 0) { d[l] = d[l] + f; e[l] = 0.0; }
            WAVE_SYNC();
        }
        if (lane == 0) ql_desc[pb][0] = -1;
        __syncthreads();
    } else {
        int cb = 0;
        const int na = nt - 64;
        for (;;) {
            __syncthreads();
            const int l = ql_desc[cb][0], m = ql_desc[cb][1];
            if (l < 0) break;
            const double *csb = cs + (size_t)cb * 2 * SYM_EIG_HBM_MAX, *snb = csb + SYM_EIG_HBM_MAX;
            // the sweep's rotations applied to component k of the vectors m .. l (rows of the transposed array)
            for (int k = t - 64; k < n; k += na) {
                double car = V[(size_t)m * ld + k];
                for (int i0 = m - 1; i0 >= l; i0 -= 32) {
                    double v[32];
#pragma unroll
                    for (int u = 0; u < 32; u++) v[u] = V[(size_t)max(i0 - u, l) * ld + k];
#pragma unroll
                    for (int u = 0; u < 32; u++) {
                        const int ii = i0 - u;
                        if (ii < l) break;
                        const double c = csb[ii], s = snb[ii];
                        V[(size_t)(ii + 1) * ld + k] = s * v[u] + c * car;
                        car = c * v[u] - s * car;
                    }
                }
                V[(size_t)l * ld + k] = car;
            }
            cb ^= 1;
        }
    }
    __syncthreads();
    sym_eig_hbm_transpose(V, n, ld);
    if (tm && t == 0) { const long long n_ = (long long)wall_clock64(); tm[3] = (float)(n_ - tm0); }
}

__device__ bool chol_block(double *A, int n, int ld, int *sh_flag) {
    const int t = threadIdx.x, nt = blockDim.x;
    if (t == 0) *sh_flag = 1;
    __syncthreads();
    for (int j = 0; j < n; j++) {
        if (t == 0) {
            double d = A[j * ld + j];
            if (!(d > 0.0) || !isfinite(d)) *sh_flag = 0; else A[j * ld + j] = sqrt(d);
        }
        __syncthreads();
        if (!*sh_flag) return false;
        double l = A[j * ld + j];
        for (int i = j + 1 + t; i < n; i += nt) A[i * ld + j] /= l;
        __syncthreads();
        int m = n - j - 1;
        for (int w = t; w < m * m; w += nt) {
            int r = w / m, cc = w - r * m;
            if (cc > r) continue;
            int i = j + 1 + r, k = j + 1 + cc;
            A[i * ld + k] -= A[i * ld + j] * A[k * ld + j];
        }
        __syncthreads();
    }
    return true;
}
// solve L L^T x = b with one wavefront (lanes own strided entries of x held in LDS xs)
__device__ void chol_solve_wave(const double *L, int n, int ld, double *xs) {
    const int t = threadIdx.x;
    if (t < 64) {
        for (int j = 0; j < n; j++) {
            double xj = xs[j] / L[j * ld + j];
            __builtin_amdgcn_wave_barrier();
            if (t == 0) xs[j] = xj;
            for (int i = j + 1 + t; i < n; i += 64) xs[i] -= L[i * ld + j] * xj;
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
        for (int j = n - 1; j >= 0; j--) {
            double xj = xs[j] / L[j * ld + j];
            __builtin_amdgcn_wave_barrier();
            if (t == 0) xs[j] = xj;
            for (int i = t; i < j; i += 64) xs[i] -= L[j * ld + i] * xj;
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ dense linear algebra on the scaled system
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define VIO_LWMAX 336  // >= LW for W = 20

// lower-triangle tile index -> (ti, tj), ti >= tj
__device__ __forceinline__ void tri_decode(int idx, int &ti, int &tj) {
    int r = 0;
    while (idx >= r + 1) { idx -= r + 1; r++; }
    ti = r; tj = idx;
}

// S = S_p H S_p + mu*diag(dgp^2) - sum_k inv[k] (S_p Hpl[k][:])^T (S_p Hpl[k][:])   (lower tiles only), v_mfma_f64_16x16x4_f64.
// Hs / Ws: the UNSCALED H and landmark coupling rows (ld); the Jacobi column scaling sp is applied while loading the MFMA
// operands. Kpad rows (multiple of 4, rows >= Fa are zero), inv[k] = sl[k]^2 / hllr[k].
__device__ __forceinline__ void schur_mfma(const double *Hs, const double *Ws, const double *inv, const double *dgp, const double *sp, double mu,
                           int Kpad, int n /*multiple of 16*/, int ld, double *Sc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nb = n >> 4, ntile = nb * (nb + 1) / 2;
    const int li = lane & 15, lk = lane >> 4;
    for (int tile = wave; tile < ntile; tile += nw) {
        int ti, tj;
        tri_decode(tile, ti, tj);
        v4f64 acc;
        for (int r = 0; r < 4; r++) {
            int row = 16 * ti + lk + 4 * r, col = 16 * tj + li;
            double v = sp[row] * sp[col] * Hs[(size_t)row * ld + col];
            if (row == col) { v += mu * dgp[row] * dgp[row]; if (sp[row] == 0.0) v = 1.0; }
            acc[r] = v;
        }
        const double *wa = Ws + 16 * ti + li, *wb = Ws + 16 * tj + li;
        const double spa = sp[16 * ti + li], spb = sp[16 * tj + li];
        for (int k0 = 0; k0 < Kpad; k0 += 4) {
            int kk = k0 + lk;
            double a = -(wa[(size_t)kk * ld] * spa * inv[kk]);
            double b = wb[(size_t)kk * ld] * spb;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; r++) Sc[(size_t)(16 * ti + lk + 4 * r) * ld + 16 * tj + li] = acc[r];
    }
    __syncthreads();
}

// ------------------------------------------------------------------ Schur complement + Cholesky with S resident in LDS
// S (n x n, n = 16 nb) is kept as its nb(nb+1)/2 lower 16x16 tiles in LDS; element (r, c) of tile (ti, tj) lives at
// tile_base + r*16 + (c ^ r): the XOR swizzle makes both row-wise and column-wise 64-bit accesses bank-conflict free.
__device__ __forceinline__ int tl_idx(int ti, int tj, int r, int c) { return ((ti * (ti + 1) / 2 + tj) << 8) + (r << 4) + (c ^ r); }

__device__ __forceinline__ void schur_mfma_lds(const double *Hs, const double *Ws, const double *inv, const double *dgp, const double *sp, double mu,
                               int Kpad, int n, int ld, double *T) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nb = n >> 4, ntile = nb * (nb + 1) / 2;
    const int li = lane & 15, lk = lane >> 4;
    for (int tile = wave; tile < ntile; tile += nw) {
        int ti, tj;
        tri_decode(tile, ti, tj);
        v4f64 acc;
        for (int r = 0; r < 4; r++) {
            int row = 16 * ti + lk + 4 * r, col = 16 * tj + li;
            double v = sp[row] * sp[col] * Hs[(size_t)row * ld + col];
            if (row == col) { v += mu * dgp[row] * dgp[row]; if (sp[row] == 0.0) v = 1.0; }
            acc[r] = v;
        }
        const double *wa = Ws + 16 * ti + li, *wb = Ws + 16 * tj + li;
        const double spa = sp[16 * ti + li], spb = sp[16 * tj + li];
        for (int k0 = 0; k0 < Kpad; k0 += 4) {
            int kk = k0 + lk;
            double a = -(wa[(size_t)kk * ld] * spa * inv[kk]);
            double b = wb[(size_t)kk * ld] * spb;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; r++) T[tl_idx(ti, tj, lk + 4 * r, li)] = acc[r];
    }
    __syncthreads();
}

// Same Schur complement, landmark rows staged through LDS once: every wavefront keeps the accumulators of its (<= MAXT)
// tiles in registers for the whole k loop, the 16-row chunks of Hpl (scaled by sp on the way in) are double-buffered in the
// tile region itself (it is only written at the end).  Global traffic drops from 2 * ntile * Kpad * 16 doubles (every
// tile re-reading its two column panels) to Kpad * n doubles.
// colmask: bit c set = column tile c of Ws can be non-zero.  The landmark rows only couple to pose and extrinsic / td columns, the
// speed-bias columns (more than half of the row) are identically zero: tiles with an all-zero panel keep their H values and get no
// MFMAs, their columns are not staged, and the active tiles are dealt round-robin over the wavefronts so that the work stays
// balanced.  Skipping them is exact (they would only add zeros).
#define SCH_CH 16
template <int MAXT>
__device__ __forceinline__ void schur_mfma_staged(const double *Hs, const double *Ws, const double *inv, const double *dgp, const double *sp, double mu,
                                  int Kpad, int n, int ld, double *T, unsigned colmask) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6, nw = nt >> 6;
    const int nb = n >> 4, ntile = nb * (nb + 1) / 2;
    const int li = lane & 15, lk = lane >> 4;
    const int lds = n + 8;  // padded row stride of the staged chunk (bank spread of the 4 k-rows)
    double *buf0 = T, *buf1 = T + SCH_CH * lds;
    int *order = (int *)(T + 2 * SCH_CH * lds);  // [ntile] tiles, active ones first; [ntile] = their count; then the active column tiles
    int *ctab = order + ntile + 1;
    if (t == 0) {
        int na = 0;
        for (int pass = 0; pass < 2; pass++)
            for (int tile = 0; tile < ntile; tile++) {
                int ti, tj;
                tri_decode(tile, ti, tj);
                bool act = ((colmask >> ti) & 1u) && ((colmask >> tj) & 1u);
                if (act == (pass == 0)) order[na++] = tile;
                if (pass == 0 && tile == ntile - 1) order[ntile] = na;
            }
        int nc = 0;
        for (int cb = 0; cb < nb; cb++) if ((colmask >> cb) & 1u) ctab[nc++] = cb;
    }
    __syncthreads();
    const int nact = order[ntile], nac = __popc(colmask & ((1u << nb) - 1u));
    v4f64 acc[MAXT];
    int tis[MAXT], tjs[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        int slot = wave + i * nw;
        int ti = 0, tj = 0;
        if (slot < ntile) tri_decode(order[slot], ti, tj);
        tis[i] = ti; tjs[i] = tj;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int row = 16 * ti + lk + 4 * r, col = 16 * tj + li;
            double v = sp[row] * sp[col] * Hs[(size_t)row * ld + col];
            if (row == col) { v += mu * dgp[row] * dgp[row]; if (sp[row] == 0.0) v = 1.0; }
            acc[i][r] = v;
        }
    }
    const int nchunk = (Kpad + SCH_CH - 1) / SCH_CH;
    const int ncol = 16 * nac;
    auto stage = [&](int ch, double *buf) {
        for (int q = t; q < SCH_CH * ncol; q += nt) {
            int r = q / ncol, c2 = q - r * ncol, cc = 16 * ctab[c2 >> 4] + (c2 & 15), kk = ch * SCH_CH + r;
            buf[r * lds + cc] = kk < Kpad ? Ws[(size_t)kk * ld + cc] * sp[cc] : 0.0;
        }
    };
    stage(0, buf0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ch++) {
        double *cur = (ch & 1) ? buf1 : buf0, *nxt = (ch & 1) ? buf0 : buf1;
        if (ch + 1 < nchunk) stage(ch + 1, nxt);
        double iv[SCH_CH / 4];
#pragma unroll
        for (int ks = 0; ks < SCH_CH / 4; ks++) { int kk = ch * SCH_CH + 4 * ks + lk; iv[ks] = kk < Kpad ? inv[kk] : 0.0; }
#pragma unroll
        for (int i = 0; i < MAXT; i++) {
            if (wave + i * nw < nact) {
                const double *pa = cur + lk * lds + 16 * tis[i] + li, *pb = cur + lk * lds + 16 * tjs[i] + li;
#pragma unroll
                for (int ks = 0; ks < SCH_CH / 4; ks++) {
                    double a = -(pa[4 * ks * lds] * iv[ks]);
                    double b = pb[4 * ks * lds];
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MAXT; i++)
        if (wave + i * nw < ntile) {
#pragma unroll
            for (int r = 0; r < 4; r++) T[tl_idx(tis[i], tjs[i], lk + 4 * r, li)] = acc[i][r];
        }
    __syncthreads();
}

// broadcast of one lane's double through SGPRs (v_readlane_b32 x 2): far lower latency than the ds_bpermute behind __shfl.
// The lane index must be wave-uniform (here: a compile-time constant of an unrolled loop).
__device__ __forceinline__ double bcast_lane(double v, int src_lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// ---- pieces shared by the tile Cholesky variants.  D = one 16 x 16 tile (256 doubles, element (r, c) at (r << 4) + (c ^ r)).
// Diagonal block by ONE wavefront on the matrix core: the tile is held SYMMETRIC in the accumulator layout of v_mfma_f64_16x16x4
// (acc[r] = S[lk + 4 r][li]) and every pivot step is one rank-1 MFMA S -= l l^T whose operands need no cross-lane traffic: row j of the
// symmetric tile -- which is column j -- already sits in the lanes lk == j % 4, exactly the k-slot an operand lane feeds.  Only the
// pivot itself is broadcast (v_readlane).  Previously the rank-1 update was 14 readlane pairs + 14 multiply-subtracts per pivot
// (~370 cycles); this is one readlane pair, the reciprocal square root, one multiply and one MFMA.
// Input: the COMPLETE symmetric tile (every producer writes both triangles: the tile loaders, the MFMA trailing updates).
// Output: L in the lower triangle of D (diagonal included), 1 / l_jj in dinv16, and -- computed right after by the same wavefront --
// the strictly lower part of M = L^-1 in the strictly UPPER triangle of D (M[i][c], i > c, at row c, column i; M[c][c] = dinv16[c]).
// With M the panel below the block is a small matrix product on the matrix core (X = A M^T) instead of a 16-step substitution per
// row, and the triangular solves of the substitution phases are dot products.
// Round 4: the sixteen pivots run as FOUR blocks of four.  A pivot of the rank-1 form costs ~350 cycles -- an MFMA for S, one for F, the
// reciprocal square root and the operand selection, all on one dependent chain -- and sixteen of them per diagonal tile were two thirds of the
// whole factorisation's critical path.  Per block: every lane fetches the four panel entries of ITS row (and the four of F) from the four
// lane groups (one selection MFMA each), the 4 x 4 diagonal block is read through ten lane broadcasts and factored redundantly by all lanes
// (uniform arithmetic, four reciprocal square roots), every lane finishes its own row of the 16 x 4 panel with that factor, and ONE rank-4
// MFMA per accumulator (all four k-slots used: slot k = column 4 kb + k, which by symmetry already sits in lane group k) applies the block.
// Same algorithm, different association of the subtractions inside a block: results agree with the rank-1 form to round-off.
__device__ __forceinline__ void chol_diag_tile(double *D, double *dinv16, int *sh_flag) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    v4f64 acc, fcc;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = lk + 4 * r;
        acc[r] = D[(row << 4) + (li ^ row)];     // (diagonal tiles are stored complete: both triangles, bitwise symmetric)
        fcc[r] = row == li ? 1.0 : 0.0;
    }
    bool ok = true;
    double lcol[4], mrow[4];
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        const int c0 = 4 * kb;
        // p[m] = S[c0 + m][li] = panel entry (row li, column m); f[m] = F[c0 + m][li]
        // (gathered by the matrix core itself: with A[i][k] = [k == i / 4] and B[k][j] = S[c0 + k][j] -- which is this lane's own acc[kb] as a
        // B operand -- the product puts S[c0 + m][j] into register m of EVERY lane group; exact (ones and zeros), one MFMA instead of
        // four cross-row lane shuffles)
        const v4f64 zero4 = {0, 0, 0, 0};
        const double sel = lk == (li >> 2) ? 1.0 : 0.0;     // A operand of lane (k = lk, i = li): A[li][lk]
        const v4f64 pg = __builtin_amdgcn_mfma_f64_16x16x4f64(sel, acc[kb], zero4, 0, 0, 0);
        const v4f64 fg = __builtin_amdgcn_mfma_f64_16x16x4f64(sel, fcc[kb], zero4, 0, 0, 0);
        double p[4], f[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { p[m] = pg[m]; f[m] = fg[m]; }
        // the 4 x 4 diagonal block (lower triangle): D[a][b] = panel entry (row c0 + a, column b) = p[b] of lane li = c0 + a
        const double d00 = bcast_lane(p[0], c0), d10 = bcast_lane(p[0], c0 + 1), d20 = bcast_lane(p[0], c0 + 2), d30 = bcast_lane(p[0], c0 + 3);
        const double d11 = bcast_lane(p[1], c0 + 1), d21 = bcast_lane(p[1], c0 + 2), d31 = bcast_lane(p[1], c0 + 3);
        const double d22 = bcast_lane(p[2], c0 + 2), d32 = bcast_lane(p[2], c0 + 3), d33 = bcast_lane(p[3], c0 + 3);
        const double rl0 = rsqrt(d00);
        const double l10 = d10 * rl0, l20 = d20 * rl0, l30 = d30 * rl0;
        const double t11 = d11 - l10 * l10;
        const double rl1 = rsqrt(t11);
        const double l21 = (d21 - l20 * l10) * rl1, l31 = (d31 - l30 * l10) * rl1;
        const double t22 = d22 - l20 * l20 - l21 * l21;
        const double rl2 = rsqrt(t22);
        const double l32 = (d32 - l30 * l20 - l31 * l21) * rl2;
        const double t33 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
        const double rl3 = rsqrt(t33);
        ok = ok && (d00 > 0.0) && (t11 > 0.0) && (t22 > 0.0) && (t33 > 0.0) && isfinite(d00) && isfinite(t11) && isfinite(t22) && isfinite(t33);
        // my row of the panel and of E
        const double x0 = p[0] * rl0;
        const double x1 = (p[1] - x0 * l10) * rl1;
        const double x2 = (p[2] - x0 * l20 - x1 * l21) * rl2;
        const double x3 = (p[3] - x0 * l30 - x1 * l31 - x2 * l32) * rl3;
        const double e0 = f[0] * rl0;
        const double e1 = (f[1] - e0 * l10) * rl1;
        const double e2 = (f[2] - e0 * l20 - e1 * l21) * rl2;
        const double e3 = (f[3] - e0 * l30 - e1 * l31 - e2 * l32) * rl3;
        const double xk = lk == 0 ? x0 : (lk == 1 ? x1 : (lk == 2 ? x2 : x3));   // L[li][c0 + lk]: the k-slot this lane feeds
        const double ek = lk == 0 ? e0 : (lk == 1 ? e1 : (lk == 2 ? e2 : e3));   // M[c0 + lk][li]
        lcol[kb] = xk; mrow[kb] = ek;
        if (lane < 4) dinv16[c0 + lane] = lane == 0 ? rl0 : (lane == 1 ? rl1 : (lane == 2 ? rl2 : rl3));
        if (kb < 3) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xk, xk, acc, 0, 0, 0);
            fcc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xk, ek, fcc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int j = lk + 4 * q;
        if (li >= j) D[(li << 4) + (j ^ li)] = lcol[q];          // L[li][j]
        else D[(li << 4) + (j ^ li)] = mrow[q];                   // M[j][li], li < j, at (row li, column j)
    }
    if (!ok && lane == 0) *sh_flag = 0;
    WAVE_SYNC();
}
// the rank-1 form of rounds 2 / 3 (one MFMA pair per pivot), kept for the harness (tools/chol_bench.py micro modes compare the two)
__device__ __forceinline__ void chol_diag_tile_rank1(double *D, double *dinv16, int *sh_flag) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    // acc = the symmetric tile S; fcc = F = E^T where E starts as the identity "panel" below the block: eliminating S from
    // [[S, I], [I, 0]] leaves E = L^-T, i.e. the finished column j of E is row j of M = L^-1 (the panel operation applied to the identity).
    // In the transposed form row j of F sits in the lanes lk == j % 4 like row j of S, so its update is the same kind of rank-1 MFMA.
    v4f64 acc, fcc;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = lk + 4 * r;
        acc[r] = D[(row << 4) + (li ^ row)];     // (diagonal tiles are stored complete: both triangles, bitwise symmetric)
        fcc[r] = row == li ? 1.0 : 0.0;
    }
    bool ok = true;
    double lcol[4], mrow[4];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double ajj = bcast_lane(acc[j >> 2], j + 16 * (j & 3));   // S[j][j]: lane li = j, lk = j % 4, register j / 4
        ok = ok && (ajj > 0.0) && isfinite(ajj);
        const double rl = rsqrt(ajj);
        const bool own = lk == (j & 3);
        const double lj = own ? acc[j >> 2] * rl : 0.0;    // L[li][j] in the k-slot j % 4, zeros in the other three
        const double ej = own ? fcc[j >> 2] * rl : 0.0;    // M[j][li]
        if (own) { lcol[j >> 2] = lj; mrow[j >> 2] = ej; }
        if (lane == j + 16 * (j & 3)) dinv16[j] = rl;
        if (j < 15) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lj, lj, acc, 0, 0, 0);
            fcc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lj, ej, fcc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int j = lk + 4 * q;
        if (li >= j) D[(li << 4) + (j ^ li)] = lcol[q];          // L[li][j]
        else D[(li << 4) + (j ^ li)] = mrow[q];                   // M[j][li], li < j, at (row li, column j)
    }
    if (!ok && lane == 0) *sh_flag = 0;
    WAVE_SYNC();
}
// element (n, k) of M = L^-1 from a factored diagonal tile (chol_diag_tile)
__device__ __forceinline__ double chol_minv(const double *D, const double *dinv16, int n, int k) {
    return k < n ? D[(k << 4) + (n ^ k)] : (k == n ? dinv16[n] : 0.0);
}
// panel tile X = A M^T on the matrix core, in place (one wavefront): X[m][n] = sum_k A[m][k] M[n][k]
__device__ __forceinline__ void chol_panel_tile(double *A, const double *D, const double *dinv16) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    v4f64 x = {0, 0, 0, 0};
    double a[4], b[4];
    // (every load unconditional -- the entry of M is picked afterwards: a load under a condition is a branch with its own s_waitcnt)
    const double dv = dinv16[li];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) { const int k = 4 * kk + lk; a[kk] = A[(li << 4) + (k ^ li)]; b[kk] = D[(k << 4) + (li ^ k)]; }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) { const int k = 4 * kk + lk; b[kk] = k < li ? b[kk] : (k == li ? dv : 0.0); }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) x = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], x, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) A[((lk + 4 * r) << 4) + (li ^ (lk + 4 * r))] = x[r];
}
// y = M b for one 16-vector (forward substitution step of the right-hand side), lanes 0 .. 15 of the calling wavefront
__device__ __forceinline__ void chol_rhs_block(double *b16, const double *D, const double *dinv16) {
    const int lane = threadIdx.x & 63, n = lane & 15;
    double m[16], bv[16];
    const double dv = dinv16[n];
#pragma unroll
    for (int k = 0; k < 16; k++) { m[k] = D[(k << 4) + (n ^ k)]; bv[k] = b16[k]; }
    double y = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) y += (k < n ? m[k] : (k == n ? dv : 0.0)) * bv[k];
    WAVE_SYNC();
    if (lane < 16) b16[lane] = y;
    WAVE_SYNC();
}

// Right-looking tile Cholesky of S resident in LDS.  Look-ahead: while the other wavefronts run the trailing update of step p,
// wavefront 0 updates tile (p+1, p+1) first and factors it straight away (chol_diag_tile), so the diagonal factorisation is off the
// critical path.  dinv[16 nb] receives 1 / l_jj.  rhs (optional, 16 nb doubles in LDS): the forward substitution L y = rhs rides along.
__device__ __forceinline__ bool chol_tiles(double *T, int nb, int *sh_flag, double *dinv, float *tm = nullptr, double *rhs = nullptr) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6, nw = nt >> 6;
    const int li = lane & 15, lk = lane >> 4;
    auto diag = [&](int p) -> double * { return T + ((p * (p + 1) / 2 + p) << 8); };
    auto update_tile = [&](int ti, int tj, int p) {
        // (all twelve loads first: written as an expression inside the MFMA chain they were issued pair by pair, one LDS round trip
        // per k-step)
        v4f64 acc;
        double a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = T[tl_idx(ti, tj, lk + 4 * r, li)];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { a[kk] = T[tl_idx(ti, p, li, 4 * kk + lk)]; b[kk] = T[tl_idx(tj, p, li, 4 * kk + lk)]; }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[kk], b[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) T[tl_idx(ti, tj, lk + 4 * r, li)] = acc[r];
    };
    // right-hand side rows below tile row p: b_i -= L_ip y_p
    auto update_rhs = [&](int p) {
        for (int q = lane; q < 16 * (nb - 1 - p); q += 64) {
            const int ti = p + 1 + (q >> 4), r = q & 15;
            double sacc = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) sacc += T[tl_idx(ti, p, r, k)] * rhs[16 * p + k];
            rhs[16 * ti + r] -= sacc;
        }
    };
    // the trailing tiles are handed out through a counter in LDS: wavefront 0 joins the others as soon as the next diagonal block is
    // factored (a fixed assignment left it idle behind the short factorisation of the last steps and the others behind the 54 tiles of
    // the first)
    __shared__ int ctr_s, pdone_s;
    int *ctr = &ctr_s;
    volatile int *pdone = &pdone_s;
    if (t == 0) { *sh_flag = 1; pdone_s = 0; }
    __syncthreads();
    if (wave == 0) chol_diag_tile(diag(0), dinv, sh_flag);
    __syncthreads();
    long long tm0 = (tm && t == 0) ? VIO_CLOCK() : 0;
    for (int p = 0; p < nb; p++) {
        if (!*sh_flag) return false;
        if (nw > 1 && t == 0) *ctr = 1;   // (everybody is behind the barrier that ended step p - 1; published with wavefront 0's panel count below)
        // (b) panel: the tiles below the diagonal block become A M^T (one wavefront per tile); the right-hand side block becomes M b
        for (int ti = p + 1 + wave; ti < nb; ti += nw) chol_panel_tile(T + ((ti * (ti + 1) / 2 + p) << 8), diag(p), dinv + 16 * p);
        if (rhs && wave == nw - 1) chol_rhs_block(rhs + 16 * p, diag(p), dinv + 16 * p);
        // (c) trailing update S22 -= L21 L21^T on the FP64 matrix cores; tile 0 = (p+1, p+1) belongs to wavefront 0
        const int m = nb - 1 - p, ntile = m * (m + 1) / 2;
        if (nw > 1) {
            // Round 6: no workgroup barrier between the panel and the trailing update.  Every wavefront counts its panel tiles in (pdone, LDS) and the
            // trailing update waits for the count -- except wavefront 0's first piece: tile (p+1, p+1) takes column p's term from panel tile (p+1, p),
            // which wavefront 0 has just solved ITSELF, so it goes on to that update and to the factorisation of the next diagonal block at once
            // (2.5 us, the longest piece of a step) instead of first waiting for everybody's panel tiles.  Same tile operations, same order per tile.
            WAVE_SYNC();
            if (lane == 0) atomicAdd((int *)&pdone_s, 1);
            if (ntile > 0 && wave == 0) {
                update_tile(p + 1, p + 1, p);
                WAVE_SYNC();
                chol_diag_tile(diag(p + 1), dinv + 16 * (p + 1), sh_flag);
            }
            while (*pdone < nw * (p + 1)) __builtin_amdgcn_s_sleep(1);
            __threadfence_block();
            if (VIO_TIMERS && tm && t == 0) { long long n_ = VIO_CLOCK(); tm[0] += (float)(n_ - tm0); tm0 = n_; }
            if (rhs && wave == nw - 1) update_rhs(p);
            for (;;) {
                int tile = 0;
                if (lane == 0) tile = atomicAdd(ctr, 1);
                tile = __builtin_amdgcn_readfirstlane(tile);
                if (tile >= ntile) break;
                int ti, tj;
                tri_decode(tile, ti, tj);
                update_tile(ti + p + 1, tj + p + 1, p);
            }
        } else {
            WAVE_SYNC();
            if (rhs) update_rhs(p);
            for (int tile = 0; tile < ntile; tile++) {
                int ti, tj;
                tri_decode(tile, ti, tj);
                update_tile(ti + p + 1, tj + p + 1, p);
            }
            WAVE_SYNC();
            if (ntile > 0) chol_diag_tile(diag(p + 1), dinv + 16 * (p + 1), sh_flag);
        }
        if (VIO_TIMERS && tm && t == 0) { long long n_ = VIO_CLOCK(); tm[1] += (float)(n_ - tm0); tm0 = n_; }
        __syncthreads();
        if (VIO_TIMERS && tm && t == 0) { long long n_ = VIO_CLOCK(); tm[2] += (float)(n_ - tm0); tm0 = n_; }
    }
    return *sh_flag != 0;
}

// Cholesky of a matrix whose lower 16x16 tiles (tl_idx layout) live in HBM / L2: windows whose Schur complement does not fit LDS (P = 322
// at W = 20 is 231 tiles = 473 KB).  Left-looking by block column: while column p is gathered into LDS (col: nb tiles) every wavefront
// subtracts the contributions of the p finished columns from its row tiles -- MFMA operands straight from global memory, the loads of
// two finished columns in flight per trip -- then the column is factored in LDS with the same pieces as chol_tiles (diagonal block by
// the wavefront that owns it, as soon as it has it; panel tiles X = A M^T; the right-hand side block riding along) and written back.
// Per element the same operations in the same order as chol_tiles (updates in ascending column order, each a chain of four MFMAs), so
// the two factorisations agree bit for bit.  Traffic: nb^3 / 6 tile reads (3 MB at nb = 21), all of it L2 hits.
// Left-looking update of N row tiles i0, i0 + step, ... of block column p at once (one wavefront): tile(i, p) - sum_j L(i, j) L(p, j)^T over the
// finished columns j < p, straight from HBM / L2 into `col` (LDS).  The tiles share the L(p, j) operand, their loads are in flight together and
// their MFMA chains interleave; per tile the operations and their order are those of the one-tile walk (ascending j, four MFMAs each).
template <int N> __device__ __forceinline__ void chol_stream_update(double *G, int p, int i0, int step, double *col) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    auto gtile = [&](int ti, int tj) -> double * { return G + ((size_t)(ti * (ti + 1) / 2 + tj) << 8); };
    v4f64 acc[N];
    const double *ai[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double *gi = gtile(i0 + k * step, p);
        ai[k] = gtile(i0 + k * step, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) acc[k][r] = gi[((lk + 4 * r) << 4) + (li ^ (lk + 4 * r))];
    }
    const double *bp = gtile(p, 0);
    int j = 0;
    for (; j + 2 <= p; j += 2) {
        double a[N][8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int off = ((j + (u >> 2)) << 8) + (li << 4) + ((4 * (u & 3) + lk) ^ li);
            b[u] = bp[off];
#pragma unroll
            for (int k = 0; k < N; k++) a[k][u] = ai[k][off];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int k = 0; k < N; k++) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[k][u], b[u], acc[k], 0, 0, 0);
    }
    if (j < p) {
        double a[N][4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int off = (j << 8) + (li << 4) + ((4 * u + lk) ^ li);
            b[u] = bp[off];
#pragma unroll
            for (int k = 0; k < N; k++) a[k][u] = ai[k][off];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < N; k++) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[k][u], b[u], acc[k], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
        double *ct = col + ((i0 + k * step - p) << 8);
#pragma unroll
        for (int r = 0; r < 4; r++) ct[((lk + 4 * r) << 4) + (li ^ (lk + 4 * r))] = acc[k][r];
    }
}

__device__ __forceinline__ bool chol_tiles_stream(double *G, int nb, double *col2, int *sh_flag, double *dinv, double *rhs, float *tm = nullptr) {
    // col2: TWO block columns of LDS plus one tile (2 nb + 1 tiles; the last one carries the look-ahead diagonal tile).  Column p is built in buffer p & 1, and every tile goes back to HBM from the wavefront that
    // finished it (the diagonal tile after its factorisation, a panel tile after its solve): no separate write-back phase, two barriers per
    // column instead of three, and the right-hand side update of column p (last wavefront, reading buffer p & 1) runs beside the other
    // wavefronts' updates of column p + 1, which fill the other buffer.
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6, nw = nt >> 6;
    const int li = lane & 15, lk = lane >> 4;
    auto gtile = [&](int ti, int tj) -> double * { return G + ((size_t)(ti * (ti + 1) / 2 + tj) << 8); };
    auto store_tile = [&](double *g, const double *ct) {   // one wavefront: 256 doubles
#pragma unroll
        for (int r = 0; r < 4; r++) g[lane + 64 * r] = ct[lane + 64 * r];
    };
    if (t == 0) *sh_flag = 1;
    __syncthreads();
    long long tm1 = tm ? VIO_CLOCK() : 0;   // timing harness only (stage_linalg.hip), as wavefront 1 sees the phases: [0] update of the
                                                          // column incl. the wait for the diagonal block, [1] panels
    // Look-ahead on the diagonal: wavefront 1 takes no row tiles; during the update phase of column p it accumulates tile (p + 1, p + 1) over the
    // finished columns 0 .. p - 1 into `ahead` (LDS), so that wavefront 0 -- which owns the diagonal tile -- only adds column p's own term
    // (from the LDS buffer) before the factorisation.  The two serial pieces of a column (the length-p chain of the diagonal tile, 2.4 us on
    // average at nb = 21, and chol_diag_tile, 2.3 us) run side by side instead of one after the other.  Per element the terms arrive in
    // ascending column order, four MFMAs each, as in chol_tiles: bit-identical.
    double *ahead = col2 + ((size_t)(2 * nb) << 8);
    const int la_wave = nw > 2 ? 1 : 0;   // (fewer than three wavefronts: wavefront 0 looks ahead itself after its factorisation)
    if (wave == la_wave) {
#pragma unroll
        for (int r = 0; r < 4; r++) ahead[((lk + 4 * r) << 4) + (li ^ (lk + 4 * r))] = gtile(0, 0)[((lk + 4 * r) << 4) + (li ^ (lk + 4 * r))];
    }
    __syncthreads();
    v4f64 ahead_next = {0, 0, 0, 0};
    for (int p = 0; p < nb; p++) {
        double *col = col2 + (size_t)(p & 1) * ((size_t)nb << 8);
        // the right-hand side rows below block p - 1 take their term b_i -= L_i,p-1 y_p-1 (previous column's buffer); on the look-ahead
        // wavefront, which has the shortest update phase (block p itself is solved by the last wavefront in the panel phase, behind a barrier)
        if (rhs && wave == la_wave && p > 0) {
            const double *pc = col2 + (size_t)((p - 1) & 1) * ((size_t)nb << 8);
            for (int q = lane; q < 16 * (nb - p); q += 64) {
                const int tl = 1 + (q >> 4), r = q & 15;
                double sacc = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) sacc += pc[(tl << 8) + (r << 4) + (k ^ r)] * rhs[16 * (p - 1) + k];
                rhs[16 * (p - 1 + tl) + r] -= sacc;
            }
            WAVE_SYNC();
        }
        // (a) column p minus the finished columns, into LDS: wavefront 0 takes the diagonal tile alone (its update, then chol_diag_tile),
        // wavefronts 1 .. nw - 1 share the row tiles below it and walk all of theirs at once (chol_stream_update)
        auto look_ahead = [&]() {
            // (reads `ahead` of this column happen in wavefront 0 right after the previous barrier; the new value is written behind the barrier
            //  that ends this phase -- see below -- so the two never overlap)
            v4f64 nx = {0, 0, 0, 0};
            if (p + 1 < nb) {
                const double *row = gtile(p + 1, 0);   // tiles (p + 1, j) are consecutive
#pragma unroll
                for (int r = 0; r < 4; r++) nx[r] = gtile(p + 1, p + 1)[((lk + 4 * r) << 4) + (li ^ (lk + 4 * r))];
                int j = 0;
                for (; j + 4 <= p; j += 4) {
                    double a[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) a[u] = row[((j + (u >> 2)) << 8) + (li << 4) + ((4 * (u & 3) + lk) ^ li)];
#pragma unroll
                    for (int u = 0; u < 16; u++) nx = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[u], a[u], nx, 0, 0, 0);
                }
                for (; j < p; j++) {
                    double a[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) a[u] = row[(j << 8) + (li << 4) + ((4 * u + lk) ^ li)];
#pragma unroll
                    for (int u = 0; u < 4; u++) nx = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[u], a[u], nx, 0, 0, 0);
                }
            }
            ahead_next = nx;
        };
        if (wave == 0) {
            v4f64 acc;
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] = ahead[((lk + 4 * r) << 4) + (li ^ (lk + 4 * r))];
            if (p > 0) {   // column p - 1's term: tile (p, p - 1) sits in the previous buffer
                const double *pt = col2 + (size_t)((p - 1) & 1) * ((size_t)nb << 8) + (1 << 8);
                double a[4];
#pragma unroll
                for (int u = 0; u < 4; u++) a[u] = pt[(li << 4) + ((4 * u + lk) ^ li)];
#pragma unroll
                for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[u], a[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) col[((lk + 4 * r) << 4) + (li ^ (lk + 4 * r))] = acc[r];
            WAVE_SYNC(); chol_diag_tile(col, dinv + 16 * p, sh_flag); WAVE_SYNC(); store_tile(gtile(p, p), col);
            if (la_wave == 0) look_ahead();
            if (nw == 1) for (int i = p + 1; i < nb; i++) chol_stream_update<1>(G, p, i, 1, col);   // (a lone wavefront does everything)
        } else if (wave == la_wave) {
            look_ahead();
        } else {
            const int st7 = la_wave == 1 ? nw - 2 : nw - 1;   // the look-ahead wavefront takes no row tiles
            int i = p + (la_wave == 1 ? wave - 1 : wave);
            for (; i + 3 * st7 < nb; i += 4 * st7) chol_stream_update<4>(G, p, i, st7, col);
            const int left = i < nb ? (nb - 1 - i) / st7 + 1 : 0;
            if (left == 3) chol_stream_update<3>(G, p, i, st7, col);
            else if (left == 2) chol_stream_update<2>(G, p, i, st7, col);
            else if (left == 1) chol_stream_update<1>(G, p, i, st7, col);
        }
        if (VIO_TIMERS && tm && t == 64) { long long n_ = VIO_CLOCK(); tm[0] += (float)(n_ - tm1); }
        __syncthreads();
        if (VIO_TIMERS && tm && t == 64) tm1 = VIO_CLOCK();
        if (!*sh_flag) return false;
        if (wave == la_wave) {   // wavefront 0 is past its read of `ahead`: publish the next tile (visible after this column's second barrier)
#pragma unroll
            for (int r = 0; r < 4; r++) ahead[((lk + 4 * r) << 4) + (li ^ (lk + 4 * r))] = ahead_next[r];
        }
        // (b) panel tiles (each back to HBM from the wavefront that solved it) and the right-hand side block
        for (int i = p + 1 + wave; i < nb; i += nw) {
            double *ct = col + ((i - p) << 8);
            chol_panel_tile(ct, col, dinv + 16 * p);
            WAVE_SYNC();
            store_tile(gtile(i, p), ct);
        }
        if (rhs && wave == nw - 1) chol_rhs_block(rhs + 16 * p, col, dinv + 16 * p);
        __syncthreads();
        if (VIO_TIMERS && tm && t == 64) { long long n_ = VIO_CLOCK(); tm[1] += (float)(n_ - tm1); tm1 = n_; }
    }
    return *sh_flag != 0;
}

// backward half of the solve: L^T x = y (y from the forward substitution that rode along with chol_tiles).  The diagonal blocks are
// applied as x_p = M_p^T b_p with M = L^-1 from the strictly upper triangle of the factored tile (chol_diag_tile).
__device__ __forceinline__ void chol_backward_tiles(const double *T, int nb, double *xs, const double *dinv) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6;
    for (int p = nb - 1; p >= 0; p--) {
        if (wave == 0) {
            const double *D = T + ((size_t)(p * (p + 1) / 2 + p) << 8);
            double x = 0;
            if (lane < 16) {
                const int c = lane;
#pragma unroll
                for (int k = 0; k < 16; k++) x += (k > c ? D[(c << 4) + (k ^ c)] : (k == c ? dinv[16 * p + c] : 0.0)) * xs[16 * p + k];   // M[k][c]
            }
            WAVE_SYNC();
            if (lane < 16) xs[16 * p + lane] = x;
        }
        __syncthreads();
        for (int q = t; q < 16 * p; q += nt) {
            int tj = q >> 4, cc = q & 15;
            double sacc = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) sacc += T[tl_idx(p, tj, k, cc)] * xs[16 * p + k];
            xs[q] -= sacc;
        }
        __syncthreads();
    }
}

// The same backward half by wavefront 0 alone for tiles resident in LDS: no workgroup barrier inside (the 2 x nb barriers and the
// conditional loads of chol_backward_tiles were 20 us of a 75 us solve at nb = 11), every load of a step unconditional and in flight
// at once, per element the operations of chol_backward_tiles in the same order (bit-identical results).  Ends with one barrier.
__device__ __forceinline__ void chol_backward_tiles_wave(const double *T, int nb, double *xs, const double *dinv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const int c = lane & 15;
        for (int p = nb - 1; p >= 0; p--) {
            const double *D = T + ((size_t)(p * (p + 1) / 2 + p) << 8);
            double m[16], b[16];
#pragma unroll
            for (int k = 0; k < 16; k++) { m[k] = D[(c << 4) + (k ^ c)]; b[k] = xs[16 * p + k]; }
            const double dv = dinv[16 * p + c];
            double x = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) x += (k > c ? m[k] : (k == c ? dv : 0.0)) * b[k];   // M[k][c]
            WAVE_SYNC();
            if (lane < 16) xs[16 * p + lane] = x;
            WAVE_SYNC();
#pragma unroll
            for (int k = 0; k < 16; k++) b[k] = xs[16 * p + k];
            for (int q = lane; q < 16 * p; q += 64) {
                const int tj = q >> 4, cc = q & 15;
#pragma unroll
                for (int k = 0; k < 16; k++) m[k] = T[tl_idx(p, tj, k, cc)];
                double sacc = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) sacc += m[k] * b[k];
                xs[q] -= sacc;
            }
            WAVE_SYNC();
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void chol_solve_tiles(const double *T, int nb, double *xs, const double *dinv) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6;
    for (int p = 0; p < nb; p++) {  // forward: L y = b
        if (wave == 0) {
            const int row = lane & 15;
            double b = xs[16 * p + row];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                double xj = bcast_lane(b, j) * dinv[16 * p + j];
                if (row == j) b = xj; else if (row > j) b -= T[tl_idx(p, p, row, j)] * xj;
            }
            if (lane < 16) xs[16 * p + row] = b;
        }
        __syncthreads();
        for (int q = t; q < 16 * (nb - 1 - p); q += nt) {
            int ti = p + 1 + (q >> 4), r = q & 15;
            double sacc = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) sacc += T[tl_idx(ti, p, r, k)] * xs[16 * p + k];
            xs[16 * ti + r] -= sacc;
        }
        __syncthreads();
    }
    for (int p = nb - 1; p >= 0; p--) {  // backward: L^T x = y
        if (wave == 0) {
            const int row = lane & 15;
            double b = xs[16 * p + row];
#pragma unroll
            for (int j = 15; j >= 0; j--) {
                double xj = bcast_lane(b, j) * dinv[16 * p + j];
                if (row == j) b = xj; else if (row < j) b -= T[tl_idx(p, p, j, row)] * xj;
            }
            if (lane < 16) xs[16 * p + row] = b;
        }
        __syncthreads();
        for (int q = t; q < 16 * p; q += nt) {
            int tj = q >> 4, cc = q & 15;
            double sacc = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) sacc += T[tl_idx(p, tj, k, cc)] * xs[16 * p + k];
            xs[q] -= sacc;
        }
        __syncthreads();
    }
}

// forward substitution only: xs <- L^-1 xs (used for the constant |L^-1 b|^2 = b^T A^-1 b of the prior)
__device__ __forceinline__ void chol_forward_tiles(const double *T, int nb, double *xs, const double *dinv) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6;
    for (int p = 0; p < nb; p++) {
        if (wave == 0) {
            const int row = lane & 15;
            double b = xs[16 * p + row];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                double xj = bcast_lane(b, j) * dinv[16 * p + j];
                if (row == j) b = xj; else if (row > j) b -= T[tl_idx(p, p, row, j)] * xj;
            }
            if (lane < 16) xs[16 * p + row] = b;
        }
        __syncthreads();
        for (int q = t; q < 16 * (nb - 1 - p); q += nt) {
            int ti = p + 1 + (q >> 4), r = q & 15;
            double sacc = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) sacc += T[tl_idx(ti, p, r, k)] * xs[16 * p + k];
            xs[16 * ti + r] -= sacc;
        }
        __syncthreads();
    }
}

// Blocked (16) right-looking Cholesky of the lower triangle of A (n x n, n multiple of 16): diagonal block by one
// wavefront in LDS, panel solve one row per thread, trailing update L21 L21^T on the FP64 matrix cores.
__device__ __forceinline__ bool chol_blocked(double *A, int n, int ld, int *sh_flag, double *Lpp) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6, nw = nt >> 6;
    const int nb = n >> 4;
    if (t == 0) *sh_flag = 1;
    __syncthreads();
    for (int p = 0; p < nb; p++) {
        const int o = 16 * p;
        if (t < 256) Lpp[t] = A[(size_t)(o + (t >> 4)) * ld + o + (t & 15)];
        __syncthreads();
        if (wave == 0) {
            double *L = Lpp;
            bool ok = true;
            for (int j = 0; j < 16 && ok; j++) {
                double d = L[j * 16 + j];
                if (!(d > 0.0) || !isfinite(d)) { ok = false; break; }
                double l = sqrt(d);
                if (lane > j && lane < 16) L[lane * 16 + j] = L[lane * 16 + j] / l;
                if (lane == j) L[j * 16 + j] = l;
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
                for (int q = lane; q < 256; q += 64) {
                    int i = q >> 4, k = q & 15;
                    if (k > j && i >= k) L[q] = L[q] - L[i * 16 + j] * L[k * 16 + j];
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
            }
            if (!ok && lane == 0) *sh_flag = 0;
        }
        __syncthreads();
        if (!*sh_flag) return false;
        if (t < 256) { int i = t >> 4, k = t & 15; if (k <= i) A[(size_t)(o + i) * ld + o + k] = Lpp[t]; }
        for (int r = o + 16 + t; r < n; r += nt) {
            double x[16];
            double *row = A + (size_t)r * ld + o;
#pragma unroll
            for (int cc = 0; cc < 16; cc++) {
                double s = row[cc];
#pragma unroll
                for (int k = 0; k < cc; k++) s -= x[k] * Lpp[cc * 16 + k];
                x[cc] = s / Lpp[cc * 16 + cc];
            }
#pragma unroll
            for (int cc = 0; cc < 16; cc++) row[cc] = x[cc];
        }
        __syncthreads();
        const int m = nb - 1 - p, ntile = m * (m + 1) / 2;
        const int li = lane & 15, lk = lane >> 4;
        for (int tile = wave; tile < ntile; tile += nw) {
            int ti, tj;
            tri_decode(tile, ti, tj);
            ti += p + 1; tj += p + 1;
            v4f64 acc;
            for (int r = 0; r < 4; r++) acc[r] = A[(size_t)(16 * ti + lk + 4 * r) * ld + 16 * tj + li];
            const double *pa = A + (size_t)(16 * ti + li) * ld + o, *pb = A + (size_t)(16 * tj + li) * ld + o;
            for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * kk + lk], pb[4 * kk + lk], acc, 0, 0, 0);
            for (int r = 0; r < 4; r++) A[(size_t)(16 * ti + lk + 4 * r) * ld + 16 * tj + li] = acc[r];
        }
        __syncthreads();
    }
    return true;
}
// Solve L L^T x = b in place (xs in LDS), block-wise: 16x16 triangular solves by one wavefront, updates by all threads.
__device__ __forceinline__ void chol_solve_blocked(const double *L, int n, int ld, double *xs, double *Lpp) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6;
    const int nb = n >> 4;
    for (int p = 0; p < nb; p++) {  // forward: L y = b
        const int o = 16 * p;
        if (t < 256) Lpp[t] = L[(size_t)(o + (t >> 4)) * ld + o + (t & 15)];
        __syncthreads();
        if (wave == 0) {
            double *x = xs;
            for (int j = 0; j < 16; j++) {
                double xj = x[o + j] / Lpp[j * 16 + j];
                __builtin_amdgcn_wave_barrier();
                if (lane == j) x[o + j] = xj;
                if (lane > j && lane < 16) x[o + lane] = x[o + lane] - Lpp[lane * 16 + j] * xj;
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
            }
        }
        __syncthreads();
        for (int r = o + 16 + t; r < n; r += nt) {
            const double *row = L + (size_t)r * ld + o;
            double s = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) s += row[k] * xs[o + k];
            xs[r] -= s;
        }
        __syncthreads();
    }
    for (int p = nb - 1; p >= 0; p--) {  // backward: L^T x = y
        const int o = 16 * p;
        if (t < 256) Lpp[t] = L[(size_t)(o + (t >> 4)) * ld + o + (t & 15)];
        __syncthreads();
        if (wave == 0) {
            double *x = xs;
            for (int j = 15; j >= 0; j--) {
                double xj = x[o + j] / Lpp[j * 16 + j];
                __builtin_amdgcn_wave_barrier();
                if (lane == j) x[o + j] = xj;
                if (lane < j) x[o + lane] = x[o + lane] - Lpp[j * 16 + lane] * xj;
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
            }
        }
        __syncthreads();
        for (int r = t; r < o; r += nt) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) s += L[(size_t)(o + k) * ld + r] * xs[o + k];
            xs[r] -= s;
        }
        __syncthreads();
    }
}

// out[a] = sum_b M[b][a] * v[b] for a < n (M symmetric or "column sum" of a row-major matrix with nrows rows), split over
// blockDim/256 row groups and combined through LDS part[(blockDim/256)*VIO_LWMAX]
__device__ __forceinline__ void colsum(const double *M, int ld, int nrows, const double *v, int n, double *out, double *part) {
    const int t = threadIdx.x, nt = blockDim.x;
    const int groups = nt >> 8, g = t >> 8, a0 = t & 255;
    for (int a = a0; a < n; a += 256) {
        double s = 0;
        for (int b = g; b < nrows; b += groups) s += M[(size_t)b * ld + a] * v[b];
        part[g * VIO_LWMAX + a] = s;
    }
    __syncthreads();
    for (int a = t; a < n; a += nt) {
        double s = 0;
        for (int q = 0; q < groups; q++) s += part[q * VIO_LWMAX + a];
        out[a] = s;
    }
    __syncthreads();
}
// out[k] = sum_a M[k][a] * v[a], one wavefront per row
__device__ __forceinline__ void rowdot(const double *M, int ld, int nrows, const double *v, int n, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int k = wave; k < nrows; k += nw) {
        double s = 0;
        for (int a = lane; a < n; a += 64) s += M[(size_t)k * ld + a] * v[a];
        s = wave_sum_dpp(s);
        if (lane == 0) out[k] = s;
    }
    __syncthreads();
}

// One pass over a row-major matrix M (nrows x n, leading dimension ld) producing either or both of
//   out_row[k] = M[k][:] . v          (v != nullptr)
//   out_col[a] = sum_k u[k] M[k][a]   (u != nullptr)
// One wavefront per row (lanes own columns lane, lane + 64, ...): row dots through shuffles, column sums in registers and
// combined over the wavefronts through LDS part[nw * VIO_LWMAX].  Ends with a block barrier.
template <int NC, int RB = 4>
__device__ __forceinline__ void matvec_pass_t(const double *M, int ld, int nrows, int n, const double *u, const double *v, double *out_col, double *out_row,
                              double *part) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6, nw = nt >> 6;
    double cs[NC], vv[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) { int a = lane + 64 * j; cs[j] = 0; vv[j] = (v && a < n) ? v[a] : 0.0; }
    // RB rows per trip: all their loads are issued before the first use (one wavefront has nothing else to hide the latency with; a
    // trip is one global-memory round trip, so the 512-thread serial phase with its 256 VGPRs takes eight rows per trip)
    for (int k0 = wave; k0 < nrows; k0 += RB * nw) {
        double m[RB][NC], uk[RB];
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const int k = k0 + b * nw;
            const double *r = M + (size_t)min(k, nrows - 1) * ld;
#pragma unroll
            for (int j = 0; j < NC; j++) { int a = lane + 64 * j; m[b][j] = a < n ? r[a] : 0.0; }
            uk[b] = (u && k < nrows) ? u[k] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const int k = k0 + b * nw;
            if (k >= nrows) break;
            if (v) {
                double rd = 0;
#pragma unroll
                for (int j = 0; j < NC; j++) rd += m[b][j] * vv[j];
                rd = wave_sum_dpp(rd);
                if (lane == 0) out_row[k] = rd;
            }
            if (u) {
#pragma unroll
                for (int j = 0; j < NC; j++) cs[j] += uk[b] * m[b][j];
            }
        }
    }
    if (u) {
#pragma unroll
        for (int j = 0; j < NC; j++) { int a = lane + 64 * j; if (a < n) part[wave * VIO_LWMAX + a] = cs[j]; }
        __syncthreads();
        for (int a = t; a < n; a += nt) {
            double sacc = 0;
            for (int q = 0; q < nw; q++) sacc += part[q * VIO_LWMAX + a];
            out_col[a] = sacc;
        }
    }
    __syncthreads();
}
// Same pass for a matrix whose rows are non-zero only in columns [0, n0) and [e0, e0 + ne) (the landmark coupling rows: pose
// columns and extrinsic / td columns, the speed-bias columns in between are identically zero): lanes own the compacted columns,
// a third of the loads of the dense pass.  out_col is still written for all n columns (zeros outside the two ranges).
template <int NC, int RB = 4>
__device__ __forceinline__ void matvec_pass_2range_t(const double *M, int ld, int nrows, int n, int n0, int e0, int ne, const double *u, const double *v,
                                                     double *out_col, double *out_row, double *part) {
    const int t = threadIdx.x, nt = blockDim.x, lane = t & 63, wave = t >> 6, nw = nt >> 6;
    const int ncomp = n0 + ne;
    double cs[NC], vv[NC];
    int col[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) {
        int q = lane + 64 * j;
        col[j] = q < n0 ? q : (q < ncomp ? e0 + (q - n0) : -1);
        cs[j] = 0;
        vv[j] = (v && col[j] >= 0) ? v[col[j]] : 0.0;
    }
    for (int k0 = wave; k0 < nrows; k0 += RB * nw) {   // RB rows per trip, loads first (see matvec_pass_t)
        double m[RB][NC], uk[RB];
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const int k = k0 + b * nw;
            const double *r = M + (size_t)min(k, nrows - 1) * ld;
#pragma unroll
            for (int j = 0; j < NC; j++) m[b][j] = col[j] >= 0 ? r[col[j]] : 0.0;
            uk[b] = (u && k < nrows) ? u[k] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const int k = k0 + b * nw;
            if (k >= nrows) break;
            if (v) {
                double rd = 0;
#pragma unroll
                for (int j = 0; j < NC; j++) rd += m[b][j] * vv[j];
                rd = wave_sum_dpp(rd);
                if (lane == 0) out_row[k] = rd;
            }
            if (u) {
#pragma unroll
                for (int j = 0; j < NC; j++) cs[j] += uk[b] * m[b][j];
            }
        }
    }
    if (u) {
#pragma unroll
        for (int j = 0; j < NC; j++) if (col[j] >= 0) part[wave * VIO_LWMAX + col[j]] = cs[j];
        __syncthreads();
        for (int a = t; a < n; a += nt) {
            double sacc = 0;
            if (a < n0 || (a >= e0 && a < e0 + ne))
                for (int q = 0; q < nw; q++) sacc += part[q * VIO_LWMAX + a];
            out_col[a] = sacc;
        }
    }
    __syncthreads();
}
template <int RB = 4>
__device__ __forceinline__ void matvec_pass_2range(const double *M, int ld, int nrows, int n, int n0, int e0, int ne, const double *u, const double *v,
                                                   double *out_col, double *out_row, double *part) {
    if (n0 + ne <= 128) matvec_pass_2range_t<2, RB>(M, ld, nrows, n, n0, e0, ne, u, v, out_col, out_row, part);
    else matvec_pass_t<6>(M, ld, nrows, n, u, v, out_col, out_row, part);  // larger windows: dense pass
}
template <int RB = 4>
__device__ __forceinline__ void matvec_pass(const double *M, int ld, int nrows, int n, const double *u, const double *v, double *out_col, double *out_row,
                            double *part) {
    if (n <= 192) matvec_pass_t<3, RB>(M, ld, nrows, n, u, v, out_col, out_row, part);
    else matvec_pass_t<6>(M, ld, nrows, n, u, v, out_col, out_row, part);
}

}  // namespace
