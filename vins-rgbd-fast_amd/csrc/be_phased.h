// PHASED solver (included at the end of be_kernels.hip): the trust-region loop of optimization() (estimator.cpp:1161-1368) cut into
// kernels.  The persistent kernel (be_solve_kernel_512) gives one workgroup = one CU to a sequence for the whole solve: at S = 128 half
// of the 256 CUs idle and the other half wait on their own dependent phases.  Here every data-parallel phase is a launch that covers
// all sequences with many workgroups each -- residual evaluation (one thread per residual), frame-pair / IMU Gram blocks on the FP64
// matrix cores (one wavefront per block), landmark rows, H assembly (one thread per entry), landmark Schur complement (one wavefront
// per 16 x 16 tile) -- and only the inherently serial part (Cholesky, triangular solves, dogleg, step control) runs one workgroup per
// sequence.  A "slot" = EVAL (+ accept), ASM_A, ASM_B, SCHUR, SERIAL; the host enqueues max_iterations + 2 slots back to back (no
// synchronisation); sequences that have converged fall through.  State between kernels: SolveSt (vio_state.h) + the vectors that
// already live in HBM.  Same mathematics and the same iteration / acceptance logic as solve_body; sums are formed in a fixed order,
// so results are deterministic and independent of the batch a sequence runs in.
#define PS_LDS_TILE_DOUBLES 16896   // 66 tiles (LW = 176, W = 10): the most ps_serial keeps resident in LDS

namespace {

#define PS_LS_HEAD_DOUBLES (((int)(sizeof(Params) / sizeof(double)) + 1 & ~1) + 176)   // ps_line_search: evaluation point + scalar workspace + partial costs ahead of the roles' region
#define PS_LVEC 14   // P-sized vectors ps_serial keeps in LDS (kernels.h ps_serial_lds_bytes sizes the launch with it)
__device__ __forceinline__ bool ps_active(const SolveSt &st) { return st.stage != PS_IDLE && st.stage != PS_DONE; }
__device__ __forceinline__ double *ps_imu_blk(const Ctx &c) { return c.pairblk + (size_t)((c.W + 1) * c.W / 2) * 210; }  // W x 768 doubles behind the pair blocks
__device__ __forceinline__ unsigned ps_colmask(int W1, int LW, bool vext) {
    unsigned m = 0;
    for (int cb = 0; cb < (LW >> 4); cb++) {
        const int c0 = 16 * cb, c1 = c0 + 15;
        if (c0 < 6 * W1 || (vext && c1 >= 15 * W1 && c0 < 15 * W1 + 7)) m |= 1u << cb;
    }
    return m;
}

// The landmark rows (Hpl, Hll, gl) are double-buffered: the fused evaluate + assemble kernel builds the rows of the point it evaluates -- a
// candidate that may still be rejected -- into the half that is NOT current (st.rowbuf ^ 1), ps_accept flips st.rowbuf when the point is taken.
// Second half = the same arrays one handle-size further (vio_abi.hip allocates them twice).  Everything else uses half 0 / st.rowbuf.
__device__ __forceinline__ int fis_early(const SolveSt &st, int chunk, int pslot) { return st.fi[chunk][pslot]; }
__device__ __forceinline__ void ps_sel_rows(const Batch &B, Ctx &c, int buf) {
    if (!buf) return;
    const size_t S = (size_t)B.S, n8 = (size_t)c.NL + 8;
    c.Hpl += S * n8 * c.LW; c.Hll += S * n8; c.gl += S * n8;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- setup
__global__ __launch_bounds__(512) void ps_setup_kernel(Batch B) {
    const int s = blockIdx.x + B.s0, t = threadIdx.x;
    __shared__ int scratch[2 * 512 + 8];
    __shared__ int sh_i[8];
    __shared__ Params X;
    __shared__ PreWork pw;
    __shared__ double setup_red[64];
    Ctx c = make_ctx(B, s);
    SolveSt &st = B.sst[s];
    BeSeq &be = *c.be;
    if (!be.do_solve) { if (t == 0) st.stage = PS_IDLE; return; }
    const long long ts0 = VIO_CLOCK();
    int F, Fa, nres;
    __shared__ unsigned lmkey[2048];
    solve_prologue(B, c, X, scratch, pw, sh_i, F, Fa, nres, /*allow_relo=*/true, lmkey, 2048);
    // Bounds (estimator.cpp:1282-1297: SetParameterUpperBound(para_Feature, 0, 2 / DEPTH_MAX_DIST) on landmarks triangulated without a depth
    // measurement).  Ceres: Program::IsBoundsConstrained() of the reduced program -- any VARIABLE block with a finite bound -- switches the
    // trust-region loop to its constrained form: TrustRegionMinimizer::IterationZero sends x through Plus(x, 0) before the first evaluation
    // (every local parameterisation once: the quaternions are re-normalised; ParameterBlock::Plus projects onto the box), and every step goes
    // through the projected Armijo line search (ps_eval).  reference_quirks bit 3 keeps rounds 1 - 5's clamp-only treatment (tests).
    {
        const int *alist = c.pair_list + c.nres_cap - c.NL;
        const vio_config &cfg = c.C->c;
        const double ub = 2.0 / cfg.depth_max;
        double nb = 0, ncl = 0;
        for (int k = t; k < Fa; k += blockDim.x) if (c.lm_est[alist[k]] == 2) nb += 1;
        nb = block_sum(nb, setup_red);
        const bool constrained = nb > 0 && !(cfg.reference_quirks & VIO_QUIRK_BOUND_CLAMP_ONLY);
        if (constrained) {
            const double zero6[6] = {0, 0, 0, 0, 0, 0};
            if (t <= c.W && (cfg.use_imu || t > 0)) bf::pose_plus(&X.pose[t * 7], zero6);
            if (t == c.W + 1 && sh_i[0]) bf::pose_plus(X.ex, zero6);
            if (t == c.W + 2 && sh_i[4]) bf::pose_plus(X.relo, zero6);
            for (int k = t; k < Fa; k += blockDim.x) {
                const int slot = alist[k], pi = c.lm_pidx[slot];
                if (c.lm_est[slot] == 2 && c.feat[pi] > ub) { c.feat[pi] = ub; ncl += 1; }
            }
            ncl = block_sum(ncl, setup_red);
        }
        __syncthreads();
        if (t == 0) { st.constrained = constrained ? 1 : 0; st.ls_pending = 0; be.bounded_solves += (int)nb; be.bound_clamps += (int)ncl; }
    }
    // Fused evaluate + assemble (ps_evalf_kernel): the projection residuals are cut into chunks of whole landmarks -- chunk b starts at the first
    // in-problem landmark whose first residual index is >= b C, C = PS_FUSE_CAP - W (a landmark has at most W residuals, so no chunk exceeds
    // PS_FUSE_CAP); every boundary is an independent binary search.  A frame pair whose residuals fall into several chunks gets one partial Gram
    // block per chunk (pm_np); the chunk that publishes the last one sums them in chunk order.
    {
        const int *plist = c.pair_list + c.nres_cap - 2 * c.NL;
        const int Cc = PS_FUSE_CAP - c.W, nblk = (nres + Cc - 1) / Cc, W1f = c.W + 1;
        const bool fused = B.fuse > 0 && !(sh_i[0] || sh_i[1] || sh_i[4]) && c.W <= PS_FUSE_MAXW && nblk <= PS_FUSE_MAXBLK && c.C->c.use_imu;   // (B.fuse = chunk workgroups per sequence in the grid; they loop when there are more chunks)
        if (fused) {
            if (t <= nblk) {
                int lo = 0, hi = F;   // first in-problem landmark (list position) whose first residual index is >= t C
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (c.lm_tmp[plist[mid]] >= t * Cc) hi = mid; else lo = mid + 1; }
                if (t == nblk) lo = F;
                int k = lo;
                while (k < F && c.lm_aidx[plist[k]] < 0) k++;   // (constant landmarks carry no row)
                st.blk_p[t] = lo;
                st.blk_r[t] = lo < F ? min(c.lm_tmp[plist[lo]], nres) : nres;
                st.blk_ka[t] = k < F ? c.lm_aidx[plist[k]] : Fa;
            }
            __syncthreads();
            // per frame pair (one wavefront each): which part of its residual list (ascending indices) falls into which chunk -- fixed for the
            // whole solve, so the fused kernel finds its work in a table instead of scanning lists every iteration
            const int lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
            int br[PS_FUSE_MAXBLK + 1];
#pragma unroll
            for (int b2 = 0; b2 <= PS_FUSE_MAXBLK; b2++) br[b2] = ((volatile int *)st.blk_r)[min(b2, nblk)];
            for (int p = wave; p < W1f * W1f; p += nw) {
                const int i = p / W1f, j = p - i * W1f;
                if (!(i < j)) continue;
                const int q0 = c.pair_start[p], np_ = c.pair_start[p + 1] - q0, ps_ = pair_slot(i, j, W1f);
                int lo[PS_FUSE_MAXBLK + 1];
#pragma unroll
                for (int b2 = 0; b2 <= PS_FUSE_MAXBLK; b2++) lo[b2] = 0;
                for (int base = 0; base < np_; base += 64) {
                    const bool valid = base + lane < np_;
                    const int r = c.pair_list[q0 + min(base + lane, np_ - 1)];
#pragma unroll
                    for (int b2 = 0; b2 <= PS_FUSE_MAXBLK; b2++) lo[b2] += __popcll(__ballot(valid && b2 <= nblk && r < br[b2]));
                }
                int rank = 0;
#pragma unroll
                for (int b2 = 0; b2 < PS_FUSE_MAXBLK; b2++) {
                    const int cnt = b2 < nblk ? lo[b2 + 1] - lo[b2] : 0;
                    if (lane == 0) st.fi[b2][ps_] = (q0 + lo[b2]) | (cnt << 16) | (rank << 26);
                    rank += cnt > 0 ? 1 : 0;
                }
                if (lane == 0) st.pm_np[ps_] = rank;
            }
        }
        __syncthreads();
        // (B.fuse_only: the handle launches ONLY the fused kernel -- its configuration cannot produce anything else; should a solve still not
        // qualify it is flagged, overflow bit 256, rather than left waiting for kernels that never come)
        // -- and left out of this frame's optimisation (PS_DONE at once: ps_final hands the unoptimised window back)
        if (t == 0 && !fused && B.fuse_only) be.overflow |= 256;
        if (t == 0) { st.fused = fused ? 1 : 0; st.nblk = fused ? nblk : 0; st.rowbuf = 0; st.chunk_done = 0; }
    }
    for (int k = t; k < (int)(sizeof(Params) / sizeof(double)); k += blockDim.x) ((double *)&st.X)[k] = ((const double *)&X)[k];
    if (t == 0) {
        st.F = F; st.Fa = Fa; st.nres = nres;
        // relocalisation (sh_i[4]): relo_Pose takes over the six tangent columns of the extrinsic, which is constant in such a solve
        // (solve_prologue drops the request otherwise) -- the column block is active, the records use the full 42-double layout
        st.relo = sh_i[4];
        st.ex_active = sh_i[0] || sh_i[4]; st.td_active = sh_i[1]; st.vext = sh_i[0] || sh_i[1] || sh_i[4];
        st.cost = 0; st.ccost = 0; st.radius = 1e4; st.mu = 1e-8; st.alpha = 0; st.dogleg_norm = 0; st.model_change = 0;
        st.iter = 0; st.iters_done = 0; st.succ = 0; st.invalid = 0;
        st.point_new = 0; st.scale_pending = 1; st.retry = 0; st.cauchy_valid = 0; st.eval_with_J = 1;
        st.test_fail = (c.C->c.reference_quirks >> VIO_TEST_CHOL_FAIL_SHIFT) & VIO_TEST_CHOL_FAIL_MASK;
        // (fused: workgroup 0 = prior, 1 = IMU factors, then one per chunk of projection residuals)
        st.n_eval_blocks = st.fused ? 2 + min(st.nblk, B.fuse) : 3 + (nres + 256 * B.eval_rpt - 1) / (256 * B.eval_rpt);
        st.eval_done = 0;
        st.ts0 = ts0;
        // (a solve that cannot take the fused path on a handle that launches nothing else: closed at once with the point it started from)
        st.stage = (!st.fused && B.fuse_only) ? PS_DONE : PS_EVAL_X0;
    }
}

// ---------------------------------------------------------------------------------------------------------------- ACCEPT
// Run by one wavefront once every block of the evaluation has published its partial cost: sums them in block order and takes the
// step-acceptance decision of the loop (parameter / function tolerance, rho test, radius and mu updates: Ceres TrustRegionMinimizer
// semantics as solve_body)
__device__ void ps_accept(const Batch &B, int s) {
    const int t = threadIdx.x;   // one wavefront (the first 64 threads of the block that finished the evaluation last)
    SolveSt &st = B.sst[s];
    Ctx c = make_ctx(B, s);
    const vio_config &cfg = c.C->c;
    const int W = c.W;
    // total cost: the per-block partial sums, one per lane (n_eval_blocks <= PS_MAX_EVAL_BLOCKS <= 64), reduced in a fixed tree order
    // (a device-scope read-modify-write as the load: it returns the value at the coherence point, whatever sits in this XCD's L2 -- no cache-wide
    // invalidate needed, see the tail of ps_eval_body)
    double total = t < st.n_eval_blocks ? __longlong_as_double((long long)atomicOr((unsigned long long *)&st.part[t], 0ull)) : 0.0;
    total = wave_sum_dpp(total);
    if (st.stage == PS_EVAL_X0) {
        if (t == 0) { st.cost = total; c.be->initial_cost = total; st.point_new = 1; st.stage = PS_ASM; if (st.fused) st.rowbuf ^= 1; }
        return;
    }
    const double ccost = total, cost = st.cost;
    const int *alist = c.pair_list + c.nres_cap - c.NL;
    // parameter tolerance |dx| <= 1e-8 (|x| + 1e-8): the two norms were accumulated by ps_serial when it formed the candidate
    const double xn = st.step_xn2, dn = st.step_dn2;
    bool done = false, accept = false;
    if (sqrt(dn) <= 1e-8 * (sqrt(xn) + 1e-8)) done = true;
    else if (fabs(cost - ccost) <= 1e-6 * cost) done = true;
    const double rel = (cost - ccost) / st.model_change;
    if (!done && rel > 1e-3) accept = true;
    if (accept) {
        // candidate -> current point, eight loads in flight per lane and trip (the plain loops waited for every load before the next)
        constexpr int NPAR = (int)(sizeof(Params) / sizeof(double));
        for (int k0 = t; k0 < NPAR; k0 += 8 * 64) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = ((const double *)&st.Xc)[min(k0 + 64 * u, NPAR - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) if (k0 + 64 * u < NPAR) ((double *)&st.X)[k0 + 64 * u] = v[u];
        }
        const int Fp = st.F;
        for (int k0 = t; k0 < Fp; k0 += 8 * 64) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = c.cfeat[min(k0 + 64 * u, Fp - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) if (k0 + 64 * u < Fp) c.feat[k0 + 64 * u] = v[u];
        }
    }
    if (t == 0) {
        if (done) st.stage = PS_DONE;
        else if (accept) {
            st.cost = ccost;
            st.succ++;
            if (rel < 0.25) st.radius *= 0.5;
            if (rel > 0.75) st.radius = fmax(st.radius, 3.0 * st.dogleg_norm);
            st.mu = fmax(1e-8, 2.0 * st.mu / 10.0);
            st.point_new = 1;
            if (st.fused && st.eval_with_J) st.rowbuf ^= 1;   // the rows the fused kernel built for this candidate become the current ones
            st.stage = st.iter >= cfg.max_iterations ? PS_DONE : PS_ASM;   // the candidate of the last iteration carries no Jacobians
        } else {
            st.radius *= 0.5;
            st.stage = st.iter >= cfg.max_iterations ? PS_DONE : PS_STEP;
        }
    }
}

// candidate = Plus(x, alpha * delta): every variable block through its local parameterisation (PoseLocalParameterization::Plus) and PROJECTED
// onto the box (ParameterBlock::Plus: the inverse depth of a depth-less landmark is cut at 2 / DEPTH_MAX_DIST) -> st.Xc, c.cfeat and the two
// norms of Ceres' parameter tolerance.  alpha = 1 is the trust-region step itself (1.0 * delta is exact); the line search of a
// bounds-constrained solve calls it again with shorter steps (LineSearchFunction::Evaluate: scaled_direction = alpha * direction, then Plus).
// delta: the P tangent entries of the unscaled step (LDS in ps_serial, c.vec slot 8 in ps_eval); stl, sl: landmark step and column scaling.
__device__ __forceinline__ void ps_form_candidate(const Ctx &c, SolveSt &st, const double alpha, const double *delta, const double *stl, const double *sl, double *sred) {
    const int t = threadIdx.x, nt = blockDim.x;
    const vio_config &cfg = c.C->c;
    const int W = c.W, W1 = W + 1, oE = 15 * W1, oT = 15 * W1 + 6;
    const int F = st.F, Fa = st.Fa, ex_active = st.ex_active, td_active = st.td_active;
    const int *alist = c.pair_list + c.nres_cap - c.NL;
    const Params &X = st.X;
    Params &Xc = st.Xc;
    double xn2 = 0, dn2 = 0, ncl = 0;   // |x|^2 and |candidate - x|^2 over the variable blocks (Ceres' parameter tolerance, checked by ps_accept)
    if (t <= W) {
        double pc[7], d6[6];
        for (int k = 0; k < 7; k++) pc[k] = X.pose[t * 7 + k];
        for (int k = 0; k < 6; k++) d6[k] = alpha * delta[6 * t + k];
        bf::pose_plus(pc, d6);
        for (int k = 0; k < 7; k++) { const double v = X.pose[t * 7 + k], d = v - pc[k]; xn2 += v * v; dn2 += d * d; Xc.pose[t * 7 + k] = pc[k]; }
        for (int k = 0; k < 9; k++) {
            const double v = X.sb[t * 9 + k], vc = v + alpha * delta[6 * W1 + 9 * t + k], d = v - vc;
            xn2 += v * v; dn2 += d * d;
            Xc.sb[t * 9 + k] = vc;
        }
    }
    if (t == W + 1) {
        // the block at oE: the extrinsic, or relo_Pose when the solve carries relocalisation factors (the extrinsic is constant then)
        const bool relo = st.relo != 0;
        double ec[7], d6[6];
        for (int k = 0; k < 7; k++) ec[k] = relo ? X.relo[k] : X.ex[k];
        for (int k = 0; k < 6; k++) d6[k] = alpha * delta[oE + k];
        if (ex_active) bf::pose_plus(ec, d6);
        const double tdc = X.td + (td_active ? alpha * delta[oT] : 0.0);
        if (ex_active) for (int k = 0; k < 7; k++) { const double v = relo ? X.relo[k] : X.ex[k], d = v - ec[k]; xn2 += v * v; dn2 += d * d; }
        if (td_active) { xn2 += X.td * X.td; dn2 += (X.td - tdc) * (X.td - tdc); }
        for (int k = 0; k < 7; k++) { Xc.ex[k] = relo ? X.ex[k] : ec[k]; Xc.relo[k] = relo ? ec[k] : X.relo[k]; }
        Xc.td = tdc;
    }
    for (int k = t; k < F; k += nt) c.cfeat[k] = c.feat[k];
    __syncthreads();
    for (int k = t; k < Fa; k += nt) {
        int slot = alist[k], pi = c.lm_pidx[slot];
        const double v0 = c.feat[pi];
        double v = v0 + alpha * (stl[k] * sl[k]);
        double ub = (c.lm_est[slot] == 2) ? 2.0 / cfg.depth_max : 1.7976931348623157e308;
        if (v > ub) { v = ub; ncl += 1; }
        c.cfeat[pi] = v;
        xn2 += v0 * v0; dn2 += (v0 - v) * (v0 - v);
    }
    block_sum3(xn2, dn2, ncl, sred);
    if (t == 0) { st.step_xn2 = xn2; st.step_dn2 = dn2; if (ncl > 0) c.be->bound_clamps += (int)ncl; }
}

// ---------------------------------------------------------------------------------------------------------------- EVAL
// grid (3 + ceil(max residuals / 256), S), 256 threads, dynamic LDS = pair geometry.  Block 0: prior; blocks 1, 2: IMU factors (one
// work type per wavefront: whitened residual and the four Jacobian column groups are different code paths); block b >= 3: projection
// residuals [256 (b - 3), 256 (b - 2)).  Evaluates X (first point) or the candidate Xc.  The block of a sequence that finishes last
// (device-scope counter) sums the partial costs in block order and takes the step-acceptance decision (ps_accept).
// Block -> (sequence, block of the sequence).  B.xcd_nb == 0: grid (blocks per sequence, sequences).  B.xcd_nb > 0: one-dimensional grid
// of X * ceil(ns / X) * xcd_nb blocks, X = B.xcd_n = the XCDs the launching stream may use (its CU mask covers whole XCDs), laid out so
// that EVERY block of sequence s0 + r lands on the r % X-th of them (observed dispatch: block L of a grid runs on the L % X-th enabled
// XCD; a placement for speed, never for correctness): the one-block-per-sequence kernels of the same stream (ps_setup, ps_serial,
// be_ingest, be_marg) put sequence r there as well, so the records one phase leaves behind are read by the next from the same L2.
__device__ __forceinline__ bool ps_blk(const Batch &B, int &s, int &b) {
    if (B.xcd_nb == 0) { s = (int)blockIdx.y + B.s0; b = (int)blockIdx.x; return true; }
    const int L = (int)blockIdx.x, X = B.xcd_n, idx = L / X, sl = idx / B.xcd_nb, r = sl * X + (L - idx * X);
    b = idx - sl * B.xcd_nb;
    s = r + B.s0;
    return r < B.ns;
}

// ONE evaluation role of a sequence ("block b" of ps_eval's grid) on 256 threads: b = 0 the prior, 1 / 2 the IMU factors (one work type per
// wavefront), b >= 3 the projection residuals [256 rpt (b - 3), 256 rpt (b - 2)).  Returns the calling thread's share of the cost (the caller
// block-sums it).  t = thread index within the role, act = the thread takes part: ps_eval runs a role per workgroup (all 256 threads active);
// the line search of a bounds-constrained solve (ps_line_search, inside ps_serial's 512-thread workgroup) runs the roles one after the other on
// its first 256 threads -- the barriers are unconditional, inactive threads only pass through them -- and gets the same bits.  LS = true adds
// the directional derivative gradient . delta of the role's factors to gd (the IMU factors' share is formed by the caller from c.imu_raw)
// and leaves the prior's vectors st.srp / st.sdx alone (they belong to the point ps_asm_b assembles).
template <bool LS>
__device__ __forceinline__ double ps_eval_role(const Batch &B, const int s, const Ctx &c, SolveSt &st, const Params &X, const double *feat, const bool withJ,
                                               const int b, const int t, const bool act, unsigned char *smem, const double *dlt, const double *lstl,
                                               const double *lsl, double &gd) {
    const int nt = 256;
    const BeSeq &be = *c.be;
    const vio_config &cfg = c.C->c;
    const bool vext = st.vext != 0;
    const int W = c.W, W1 = W + 1, n = c.NPR;
    double cost = 0;
    PH_INIT;
    if (b == 0) {
        if (be.has_prior) {
            // prior gradient q = b + A dx and cost dx^T b + 1/2 dx^T A dx: one thread per row, A read by columns (it is stored exactly
            // symmetric), dx from LDS
            // (three threads per row, each over a third of the columns, partial sums added in a fixed order)
            double *dxs = (double *)smem, *pacc = dxs + ((n + 1) & ~1);
            PH(45);
            prior_dx(c, X, dxs, true);
            PH(40);
            // four threads per row when they fit (n <= 64), every load of a thread in flight at once: the partials took 28 us of a 57 us
            // launch as a runtime-bound loop with one load per iteration, 15 us in batches of eight
            const int nch = max(1, min(4, nt / n));
            if (act && t < nch * n) {
                const int ch = t / n, row = t - ch * n;
                const int per = (n + nch - 1) / nch, j0 = ch * per, j1 = min(n, j0 + per);
                constexpr int MAXJ = 32;   // columns per thread on the fast path (26 at W = 10 with three threads per row)
                double acc = 0;
                if (j1 - j0 <= MAXJ) {
                    double hv[MAXJ];
#pragma unroll
                    for (int u = 0; u < MAXJ; u++) hv[u] = c.prior_H[(size_t)min(j0 + u, n - 1) * n + row];
#pragma unroll
                    for (int u = 0; u < MAXJ; u++) if (j0 + u < j1) acc += hv[u] * dxs[j0 + u];
                } else
                    for (int j = j0; j < j1; j += MAXJ) {   // larger windows: the same batches, several trips (eight loads per trip made 17 dependent round trips of the 136 columns at W = 20: 21 us)
                        double hv[MAXJ];
#pragma unroll
                        for (int u = 0; u < MAXJ; u++) hv[u] = c.prior_H[(size_t)min(j + u, j1 - 1) * n + row];
#pragma unroll
                        for (int u = 0; u < MAXJ; u++) if (j + u < j1) acc += hv[u] * dxs[j + u];
                    }
                pacc[ch * n + row] = acc;
            }
            __syncthreads();
            PH(44);
            if (act && t < n) {
                double acc = pacc[t];
                for (int ch = 1; ch < nch; ch++) acc += pacc[ch * n + t];
                const double b0 = c.prior_r[t], d = dxs[t], q = b0 + acc;
                cost += 0.5 * d * (b0 + q);
                if (LS) {
                    // (relocalisation solve: the extrinsic's columns are lent to relo_Pose, the prior does not act on them -- ps_asm_b)
                    const int a = prior_map(t, W);
                    if (!(st.relo && a >= 15 * W1 && a < 15 * W1 + 6)) gd += q * dlt[a];
                } else { st.srp[t] = q; st.sdx[t] = d; }
            }
            if (act && t == 0) cost += 0.5 * be.prior_c0;
        }
    } else if (b <= 2) {
        // IMU factors: pre-integration headers (state, Jacobian, whitening matrix: the first VIO_PREINT_HDR doubles of PreInt) staged in
        // LDS, four loads in flight per thread
        double *pl = (double *)smem;
        constexpr int PH_LD = VIO_PREINT_HDR + 1;
        // (ten loads in flight per thread and trip: the 19 loads a thread owns at W = 10 are two round trips instead of five)
        for (int q0 = act ? t : W * PH_LD; q0 < W * PH_LD; q0 += 10 * nt) {
            double v[10];
#pragma unroll
            for (int u = 0; u < 10; u++) {
                const int q = min(q0 + u * nt, W * PH_LD - 1), i = q / PH_LD, e = min(q - i * PH_LD, VIO_PREINT_HDR - 1);
                v[u] = ((const double *)&c.pre[be.pre_idx[i + 1]])[e];
            }
#pragma unroll
            for (int u = 0; u < 10; u++) if (q0 + u * nt < W * PH_LD) pl[q0 + u * nt] = v[u];
        }
        __syncthreads();
        if (b == 1) PH(46);
        const v3 G = ld3(be.g);
        const int part_ = (b == 1 ? 0 : 4) + (t >> 6), i0 = t & 63;   // block 1: types 0 .. 3 on its four wavefronts, block 2: type 4
        if (act && part_ <= 4 && !(b == 2 && (t >> 6) > 0))
            for (int i = i0; i < W; i += 64) {
                const int j = i + 1;
                const PreInt &p = *(const PreInt *)(pl + (size_t)i * PH_LD);   // only the header fields are read
                double *__restrict__ out = c.imu_raw + (size_t)i * 15 * 31;
                if (!cfg.use_imu || p.sum_dt > 10.0) { if (part_ == 0) for (int k = 0; k < 15; k++) out[k * 31 + 30] = 0; continue; }
                if (part_ == 0) {
                    double raw[15];
                    bf::imu_raw_residual(p, G, &X.pose[i * 7], &X.sb[i * 9], &X.pose[j * 7], &X.sb[j * 9], raw);
                    for (int r = 0; r < 15; r++) {
                        double sacc = 0;
                        for (int k = 0; k <= r; k++) sacc += p.sqrt_info[r * 15 + k] * raw[k];
                        out[r * 31 + 30] = sacc;
                        cost += 0.5 * sacc * sacc;
                    }
                } else if (withJ)
                    bf::imu_raw_jacobian_part(p, G, &X.pose[i * 7], &X.sb[i * 9], &X.pose[j * 7], &X.sb[j * 9], part_ - 1, out, 31);
            }
    } else {
        double *geo = (double *)smem;
        const int rpt = B.eval_rpt, r0 = 256 * rpt * (b - 3), nres = st.nres;
        for (int p = act ? t : W1 * W1 + 1; p <= W1 * W1; p += nt) {
            if (p == W1 * W1) { stm(geo + (size_t)p * 32, q2R(mkq(X.ex[6], X.ex[3], X.ex[4], X.ex[5]))); continue; }
            const int i = p / W1, j = p - i * W1;
            if (!(i < j) || c.pair_start[p + 1] == c.pair_start[p]) continue;
            bf::PairGeo g;
            bf::pair_geo(&X.pose[i * 7], &X.pose[j * 7], X.ex, g);
            double *o = geo + (size_t)p * 32;
            for (int q = 0; q < 9; q++) { o[q] = g.A1[q]; o[9 + q] = g.A2[q]; o[18 + q] = g.M[q]; }
            o[27] = g.t[0]; o[28] = g.t[1]; o[29] = g.t[2];
        }
        __syncthreads();
        const double *ricm = geo + (size_t)W1 * W1 * 32;
        // (round 5: rpt residuals per thread -- with two, the 7 projection workgroups of a sequence become 4 and a 64-sequence launch fits the
        // device's wave slots at this kernel's 216 VGPRs in one round instead of two)
        for (int u = 0; u < rpt; u++) {
            const int r = r0 + t + 256 * u;
            if (!act || r >= nres) break;
            const int slot = c.res_lm[r], k = c.res_k[r];   // k = 0: the landmark's relocalisation factor
            const int imu_i = c.lm_start[slot], imu_j = imu_i + (k > 0 ? k : 1);
            const bf::PairGeo &g = *(const bf::PairGeo *)(geo + (size_t)(imu_i * W1 + imu_j) * 32);
            double rr[2], wgt = 1.0, sq;
            if (k == 0) {
                // relocalisation factor (estimator.cpp:1336-1340): ProjectionFactor(first observation, matched point of the old keyframe)
                // on (para_Pose[start], relo_Pose, ex, inverse depth).  Its record carries d r / d relo_Pose in the extrinsic's columns
                // (lent to relo_Pose for this solve) and zeros in the pose_j columns.
                double *out = c.res + (size_t)r * 42;
                double J[40], oj[VIO_OBS_D];
                const double *oi = obs_ptr(c, slot, imu_i);
                for (int q = 0; q < VIO_OBS_D; q++) oj[q] = oi[q];
                oj[0] = c.relo_xy[2 * slot]; oj[1] = c.relo_xy[2 * slot + 1]; oj[2] = 1.0;
                bf::eval_projection(cfg, &X.pose[imu_i * 7], X.relo, X.ex, feat[c.lm_pidx[slot]], X.td, oi, oj, false, rr, withJ ? J : nullptr);
                sq = rr[0] * rr[0] + rr[1] * rr[1];
                wgt = sqrt(1.0 / (1.0 + sq));
                if (withJ) {
                    for (int a = 0; a < 2; a++) {
                        for (int d = 0; d < 6; d++) { out[a * 20 + d] = wgt * J[a * 20 + d]; out[a * 20 + 6 + d] = 0.0; out[a * 20 + 12 + d] = wgt * J[a * 20 + 6 + d]; }
                        out[a * 20 + 18] = 0.0;
                        out[a * 20 + 19] = wgt * J[a * 20 + 19];
                    }
                    out[40] = wgt * rr[0]; out[41] = wgt * rr[1];
                }
            } else if (vext) {
                double *out = c.res + (size_t)r * 42;
                bf::eval_projection_pair(cfg, g, ricm, X.ex, feat[c.lm_pidx[slot]], X.td, obs_ptr(c, slot, imu_i), obs_ptr(c, slot, imu_j),
                                         cfg.estimate_td != 0, rr, withJ ? out : nullptr, true, &wgt);
                sq = rr[0] * rr[0] + rr[1] * rr[1];
                if (withJ) {
                    out[40] = wgt * rr[0]; out[41] = wgt * rr[1];
                    // the extrinsic itself is constant while its columns serve relo_Pose: no Jacobian for it (Ceres evaluates none either)
                    if (st.relo) for (int d = 0; d < 6; d++) { out[12 + d] = 0.0; out[32 + d] = 0.0; }
                }
            } else {
                double *out = c.res + (size_t)r * 28;
                bf::eval_projection_pair(cfg, g, ricm, X.ex, feat[c.lm_pidx[slot]], X.td, obs_ptr(c, slot, imu_i), obs_ptr(c, slot, imu_j),
                                         cfg.estimate_td != 0, rr, withJ ? out : nullptr, true, &wgt, 14, false);
                sq = rr[0] * rr[0] + rr[1] * rr[1];
                if (withJ) { out[13] = wgt * rr[0]; out[27] = wgt * rr[1]; }
            }
            cost += 0.5 * log(1.0 + sq);
            if (LS) {
                // directional derivative of this factor from the record just written: (w r)^T (w J) delta over its parameter blocks
                const int oE = 15 * W1, oT = 15 * W1 + 6, ka = c.lm_aidx[slot];
                const double dl = ka >= 0 ? lstl[ka] * lsl[ka] : 0.0;
                if (k == 0 || vext) {
                    const double *out = c.res + (size_t)r * 42;
                    for (int a = 0; a < 2; a++) {
                        double jd = out[a * 20 + 18] * dlt[oT] + out[a * 20 + 19] * dl;
                        for (int d = 0; d < 6; d++) jd += out[a * 20 + d] * dlt[6 * imu_i + d] + out[a * 20 + 6 + d] * dlt[6 * imu_j + d] + out[a * 20 + 12 + d] * dlt[oE + d];
                        gd += out[40 + a] * jd;
                    }
                } else {
                    const double *out = c.res + (size_t)r * 28;
                    for (int a = 0; a < 2; a++) {
                        double jd = out[a * 14 + 12] * dl;
                        for (int d = 0; d < 6; d++) jd += out[a * 14 + d] * dlt[6 * imu_i + d] + out[a * 14 + 6 + d] * dlt[6 * imu_j + d];
                        gd += out[a * 14 + 13] * jd;
                    }
                }
            }
        }
    }
    if (b == 0) PH(62); else if (b == 1) PH(63); else if (b == 3) PH(14);
    return cost;
}

__device__ __forceinline__ void ps_eval_body(const Batch &B) {
    int s, b;
    if (!ps_blk(B, s, b)) return;
    const int t = threadIdx.x, nt = blockDim.x;
    SolveSt &st = B.sst[s];
    if (st.stage != PS_EVAL_X0 && st.stage != PS_EVAL_C) return;
    if (st.fused || b >= st.n_eval_blocks) return;   // (fused solves are evaluated by ps_evalf_kernel)
    Ctx c = make_ctx(B, s);
    const BeSeq &be = *c.be;
    const vio_config &cfg = c.C->c;
    const bool cand = st.stage == PS_EVAL_C;
    const double *feat = cand ? c.cfeat : c.feat;
    const bool withJ = st.eval_with_J != 0, vext = st.vext != 0;
    const int W = c.W, W1 = W + 1, n = c.NPR;
    __shared__ double sred[64];
    // the evaluation point in LDS: the factor code reads its parameters many times between stores to HBM (which the compiler must
    // assume to alias them), and every such re-read would be a dependent global load
    __shared__ Params X;
    {
        const double *src = (const double *)(cand ? &st.Xc : &st.X);
        for (int k = t; k < (int)(sizeof(Params) / sizeof(double)); k += nt) ((double *)&X)[k] = src[k];
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double cost = 0;
    PH_INIT;
    __syncthreads();
    if (b == 0) {
        if (be.has_prior) {
            // prior gradient q = b + A dx and cost dx^T b + 1/2 dx^T A dx: one thread per row, A read by columns (it is stored exactly
            // symmetric), dx from LDS
            // (three threads per row, each over a third of the columns, partial sums added in a fixed order)
            double *dxs = (double *)smem, *pacc = dxs + ((n + 1) & ~1);
            PH(45);
            prior_dx(c, X, dxs, true);
            PH(40);
            // four threads per row when they fit (n <= 64), every load of a thread in flight at once: the partials took 28 us of a 57 us
            // launch as a runtime-bound loop with one load per iteration, 15 us in batches of eight
            const int nch = max(1, min(4, nt / n));
            if (t < nch * n) {
                const int ch = t / n, row = t - ch * n;
                const int per = (n + nch - 1) / nch, j0 = ch * per, j1 = min(n, j0 + per);
                constexpr int MAXJ = 32;   // columns per thread on the fast path (26 at W = 10 with three threads per row)
                double acc = 0;
                if (j1 - j0 <= MAXJ) {
                    double hv[MAXJ];
#pragma unroll
                    for (int u = 0; u < MAXJ; u++) hv[u] = c.prior_H[(size_t)min(j0 + u, n - 1) * n + row];
#pragma unroll
                    for (int u = 0; u < MAXJ; u++) if (j0 + u < j1) acc += hv[u] * dxs[j0 + u];
                } else
                    for (int j = j0; j < j1; j += MAXJ) {   // larger windows: the same batches, several trips (eight loads per trip made 17 dependent round trips of the 136 columns at W = 20: 21 us)
                        double hv[MAXJ];
#pragma unroll
                        for (int u = 0; u < MAXJ; u++) hv[u] = c.prior_H[(size_t)min(j + u, j1 - 1) * n + row];
#pragma unroll
                        for (int u = 0; u < MAXJ; u++) if (j + u < j1) acc += hv[u] * dxs[j + u];
                    }
                pacc[ch * n + row] = acc;
            }
            __syncthreads();
            PH(44);
            if (t < n) {
                double acc = pacc[t];
                for (int ch = 1; ch < nch; ch++) acc += pacc[ch * n + t];
                const double b0 = c.prior_r[t], d = dxs[t], q = b0 + acc;
                st.srp[t] = q;
                st.sdx[t] = d;
                cost += 0.5 * d * (b0 + q);
            }
            if (t == 0) cost += 0.5 * be.prior_c0;
        }
    } else if (b <= 2) {
        // IMU factors: pre-integration headers (state, Jacobian, whitening matrix: the first VIO_PREINT_HDR doubles of PreInt) staged in
        // LDS, four loads in flight per thread
        double *pl = (double *)smem;
        constexpr int PH_LD = VIO_PREINT_HDR + 1;
        // (ten loads in flight per thread and trip: the 19 loads a thread owns at W = 10 are two round trips instead of five)
        for (int q0 = t; q0 < W * PH_LD; q0 += 10 * nt) {
            double v[10];
#pragma unroll
            for (int u = 0; u < 10; u++) {
                const int q = min(q0 + u * nt, W * PH_LD - 1), i = q / PH_LD, e = min(q - i * PH_LD, VIO_PREINT_HDR - 1);
                v[u] = ((const double *)&c.pre[be.pre_idx[i + 1]])[e];
            }
#pragma unroll
            for (int u = 0; u < 10; u++) if (q0 + u * nt < W * PH_LD) pl[q0 + u * nt] = v[u];
        }
        __syncthreads();
        if (b == 1) PH(46);
        const v3 G = ld3(be.g);
        const int part_ = (b == 1 ? 0 : 4) + (t >> 6), i0 = t & 63;   // block 1: types 0 .. 3 on its four wavefronts, block 2: type 4
        if (part_ <= 4 && !(b == 2 && (t >> 6) > 0))
            for (int i = i0; i < W; i += 64) {
                const int j = i + 1;
                const PreInt &p = *(const PreInt *)(pl + (size_t)i * PH_LD);   // only the header fields are read
                double *__restrict__ out = c.imu_raw + (size_t)i * 15 * 31;
                if (!cfg.use_imu || p.sum_dt > 10.0) { if (part_ == 0) for (int k = 0; k < 15; k++) out[k * 31 + 30] = 0; continue; }
                if (part_ == 0) {
                    double raw[15];
                    bf::imu_raw_residual(p, G, &X.pose[i * 7], &X.sb[i * 9], &X.pose[j * 7], &X.sb[j * 9], raw);
                    for (int r = 0; r < 15; r++) {
                        double sacc = 0;
                        for (int k = 0; k <= r; k++) sacc += p.sqrt_info[r * 15 + k] * raw[k];
                        out[r * 31 + 30] = sacc;
                        cost += 0.5 * sacc * sacc;
                    }
                } else if (withJ)
                    bf::imu_raw_jacobian_part(p, G, &X.pose[i * 7], &X.sb[i * 9], &X.pose[j * 7], &X.sb[j * 9], part_ - 1, out, 31);
            }
    } else {
        double *geo = (double *)smem;
        const int rpt = B.eval_rpt, r0 = 256 * rpt * (b - 3), nres = st.nres;
        for (int p = t; p <= W1 * W1; p += nt) {
            if (p == W1 * W1) { stm(geo + (size_t)p * 32, q2R(mkq(X.ex[6], X.ex[3], X.ex[4], X.ex[5]))); continue; }
            const int i = p / W1, j = p - i * W1;
            if (!(i < j) || c.pair_start[p + 1] == c.pair_start[p]) continue;
            bf::PairGeo g;
            bf::pair_geo(&X.pose[i * 7], &X.pose[j * 7], X.ex, g);
            double *o = geo + (size_t)p * 32;
            for (int q = 0; q < 9; q++) { o[q] = g.A1[q]; o[9 + q] = g.A2[q]; o[18 + q] = g.M[q]; }
            o[27] = g.t[0]; o[28] = g.t[1]; o[29] = g.t[2];
        }
        __syncthreads();
        const double *ricm = geo + (size_t)W1 * W1 * 32;
        // (round 5: rpt residuals per thread -- with two, the 7 projection workgroups of a sequence become 4 and a 64-sequence launch fits the
        // device's wave slots at this kernel's 216 VGPRs in one round instead of two)
        for (int u = 0; u < rpt; u++) {
            const int r = r0 + t + 256 * u;
            if (r >= nres) break;
            const int slot = c.res_lm[r], k = c.res_k[r];   // k = 0: the landmark's relocalisation factor
            const int imu_i = c.lm_start[slot], imu_j = imu_i + (k > 0 ? k : 1);
            const bf::PairGeo &g = *(const bf::PairGeo *)(geo + (size_t)(imu_i * W1 + imu_j) * 32);
            double rr[2], wgt = 1.0, sq;
            if (k == 0) {
                // relocalisation factor (estimator.cpp:1336-1340): ProjectionFactor(first observation, matched point of the old keyframe)
                // on (para_Pose[start], relo_Pose, ex, inverse depth).  Its record carries d r / d relo_Pose in the extrinsic's columns
                // (lent to relo_Pose for this solve) and zeros in the pose_j columns.
                double *out = c.res + (size_t)r * 42;
                double J[40], oj[VIO_OBS_D];
                const double *oi = obs_ptr(c, slot, imu_i);
                for (int q = 0; q < VIO_OBS_D; q++) oj[q] = oi[q];
                oj[0] = c.relo_xy[2 * slot]; oj[1] = c.relo_xy[2 * slot + 1]; oj[2] = 1.0;
                bf::eval_projection(cfg, &X.pose[imu_i * 7], X.relo, X.ex, feat[c.lm_pidx[slot]], X.td, oi, oj, false, rr, withJ ? J : nullptr);
                sq = rr[0] * rr[0] + rr[1] * rr[1];
                wgt = sqrt(1.0 / (1.0 + sq));
                if (withJ) {
                    for (int a = 0; a < 2; a++) {
                        for (int d = 0; d < 6; d++) { out[a * 20 + d] = wgt * J[a * 20 + d]; out[a * 20 + 6 + d] = 0.0; out[a * 20 + 12 + d] = wgt * J[a * 20 + 6 + d]; }
                        out[a * 20 + 18] = 0.0;
                        out[a * 20 + 19] = wgt * J[a * 20 + 19];
                    }
                    out[40] = wgt * rr[0]; out[41] = wgt * rr[1];
                }
            } else if (vext) {
                double *out = c.res + (size_t)r * 42;
                bf::eval_projection_pair(cfg, g, ricm, X.ex, feat[c.lm_pidx[slot]], X.td, obs_ptr(c, slot, imu_i), obs_ptr(c, slot, imu_j),
                                         cfg.estimate_td != 0, rr, withJ ? out : nullptr, true, &wgt);
                sq = rr[0] * rr[0] + rr[1] * rr[1];
                if (withJ) {
                    out[40] = wgt * rr[0]; out[41] = wgt * rr[1];
                    // the extrinsic itself is constant while its columns serve relo_Pose: no Jacobian for it (Ceres evaluates none either)
                    if (st.relo) for (int d = 0; d < 6; d++) { out[12 + d] = 0.0; out[32 + d] = 0.0; }
                }
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
    if (b == 0) PH(62); else if (b == 1) PH(63); else if (b == 3) PH(14);
    cost = block_sum(cost, sred);
    // Hand-over of the partial cost to the block that finishes last WITHOUT __threadfence(): on gfx950 that fence is `buffer_wbl2 sc1` +
    // `buffer_inv sc1` -- a write-back AND an invalidate of the XCD's whole L2, 448 + 64 of them per 64-sequence launch, each one throwing out
    // the lines every other workgroup on the XCD was about to hit (round 6: found when a fence per partial Gram block made the fused kernel's
    // workgroups seven times slower).  The only data that crosses workgroups inside this kernel is part[b] (everything else ps_accept reads was
    // written by earlier kernels): it is published by a returning device-scope atomic exchange, the counter increment depends on the value that
    // exchange returns (so it is issued after the exchange has been performed at the coherence point), and ps_accept reads part[] with
    // device-scope read-modify-writes.
    __shared__ int last;
    if (t == 0) {
        const unsigned long long old = atomicExch((unsigned long long *)&st.part[b], (unsigned long long)__double_as_longlong(cost));
        int zero;   // 0, computed from the exchange's return value in a way the compiler cannot fold: the increment below waits for that value
        asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"((int)(old >> 32)));
        last = atomicAdd(&st.eval_done, 1 + zero) == st.n_eval_blocks - 1;
    }
    __syncthreads();
    if (last && t < 64) {
        if (t == 0) st.eval_done = 0;
        const long long ta = (s == 0 && t == 0) ? VIO_CLOCK() : 0;
        ps_accept(B, s);
        if (VIO_TIMERS && s == 0 && t == 0) B.timings[15] += (float)(VIO_CLOCK() - ta);
    }
}

__global__ __launch_bounds__(256) void ps_eval_kernel(Batch B) { ps_eval_body(B); }
// VIO_EVAL_OCC = 3 / 4: the same kernel held to 168 / 128 VGPRs (scratch spills in the IMU factor code) for three / four workgroups per
// SIMD instead of two -- an experiment switch for the occupancy question of DESIGN.md 9
__global__ __launch_bounds__(256, 3) void ps_eval_kernel_occ3(Batch B) { ps_eval_body(B); }
__global__ __launch_bounds__(256, 4) void ps_eval_kernel_occ4(Batch B) { ps_eval_body(B); }

// ---------------------------------------------------------------------------------------------------------------- ASM_A
// grid (NB, S), 512 threads = 8 wavefronts per block; wavefront item w = 8 blockIdx.x + wave over the whole sequence:
//   [0, W1^2)            frame-pair Gram block G_p = [J r]^T [J r] on the FP64 matrix cores -> c.pairblk (items with i >= j idle)
//   [W1^2, W1^2 + W)     IMU factor Gram block (imu_block_mfma) -> ps_imu_blk
//   the rest             landmark coupling rows, Hll, gl: 32 landmarks (two threads each) per wavefront
#define PS_ROW_WAVES 32   // wavefronts of a sequence that build landmark rows (8 landmarks each per trip)
__device__ __forceinline__ void ps_asm_a_body(const Batch &B) {
    int s, bq;
    if (!ps_blk(B, s, bq)) return;
    const int t = threadIdx.x;
    const SolveSt &st = B.sst[s];
    if (st.stage != PS_ASM || st.fused) return;   // (the fused kernel has already built a fused solve's blocks and rows)
    Ctx c = make_ctx(B, s);
    ps_sel_rows(B, c, st.rowbuf);
    const BeSeq &be = *c.be;
    const int W = c.W, W1 = W + 1, LW = c.LW;
    const int lane = t & 63, wave = t >> 6, li = lane & 15, lk = lane >> 4;
    const int item = 8 * bq + wave;
    const bool vext = st.vext != 0;
    const int nres = st.nres, Fa = st.Fa;
    __shared__ double imu_lds[8 * 704];
    const long long tk0 = (s == 0 && lane == 0) ? VIO_CLOCK() : 0;
    auto tick = [&](int slot) { if (VIO_TIMERS && s == 0 && lane == 0) B.timings[slot] += (float)(VIO_CLOCK() - tk0); };
    const int npairs = W1 * (W1 - 1) / 2;
    if (item < npairs) {
        // item = pair_slot(i, j): the frame pairs i < j in row-major order
        int i = 0, rem = item;
        while (rem >= W1 - 1 - i) { rem -= W1 - 1 - i; i++; }
        const int j = i + 1 + rem, p = i * W1 + j;
        const int q0 = c.pair_start[p], np_ = c.pair_start[p + 1] - q0;
        double *out = c.pairblk + (size_t)pair_slot(i, j, W1) * 210;
        if (np_ == 0) { for (int e = lane; e < 210; e += 64) out[e] = 0; return; }
        v4f64 a00 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
        // Straight-line trips: the record layout (42 or 28 doubles) is decided OUTSIDE the loops and the MFMAs of a trip are not guarded
        // (rows beyond the end are zero operands).  With either test inside, every gather of a trip sat in its own basic block behind an
        // s_waitcnt vmcnt(0): the PB_U loads of a trip were serialised (8 us per trip instead of ~1).
        if (vext) {
            for (int base = 0; base < np_; base += 64) {
                const int nchunk = min(64, np_ - base);
                const int myidx = c.pair_list[q0 + base + min(lane, nchunk - 1)];
                const int Kc = 2 * nchunk;
                for (int k0 = 0; k0 < Kc; k0 += 4 * PB_U) {
                    double x0[PB_U], x1[PB_U];
#pragma unroll
                    for (int u = 0; u < PB_U; u++) {
                        const int kk = k0 + 4 * u + lk;
                        const bool valid = kk < Kc;
                        const int ridx = __shfl(myidx, min(kk, Kc - 1) >> 1, 64);
                        const double *Jr = c.res + (size_t)ridx * 42;
                        const int sub = kk & 1, ro = sub * 20;
                        const double v0 = Jr[ro + li], v1 = Jr[li < 3 ? ro + 16 + li : 40 + sub];
                        x0[u] = valid ? v0 : 0.0;
                        x1[u] = (valid && li < 4) ? v1 : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < PB_U; u++) {
                        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u], x0[u], a00, 0, 0, 0);
                        a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u], x0[u], a10, 0, 0, 0);
                        a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[u], x1[u], a11, 0, 0, 0);
                    }
                }
            }
        } else {
            const int lcol = li < 12 ? li : 13;
            for (int base = 0; base < np_; base += 64) {
                const int nchunk = min(64, np_ - base);
                const int myidx = c.pair_list[q0 + base + min(lane, nchunk - 1)];
                const int Kc = 2 * nchunk;
                for (int k0 = 0; k0 < Kc; k0 += 4 * PB_U) {
                    double x0[PB_U];
#pragma unroll
                    for (int u = 0; u < PB_U; u++) {
                        const int kk = k0 + 4 * u + lk;
                        const int ridx = __shfl(myidx, min(kk, Kc - 1) >> 1, 64);
                        const double v0 = c.res[(size_t)ridx * 28 + (kk & 1) * 14 + lcol];
                        x0[u] = (kk < Kc && li < 13) ? v0 : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < PB_U; u++) {
                        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u], x0[u], a00, 0, 0, 0);
                    }
                }
            }
        }
        for (int r = 0; r < 4; r++) {
            const int row = lk + 4 * r, col = li;
            if (vext) {
                if (col <= row) out[sym_idx(col, row)] = a00[r];
                if (row < 4) out[sym_idx(col, 16 + row)] = a10[r];
                if (row < 4 && col < 4 && col <= row) out[sym_idx(16 + col, 16 + row)] = a11[r];
            } else {
                if (row < 12) { if (col <= row) out[sym_idx(col, row)] = a00[r]; }
                else if (row == 12 && col <= 12) out[sym_idx(col < 12 ? col : 19, 19)] = a00[r];
            }
        }
        if (item == 0) tick(32); else if (item == W - 1) tick(33);
        return;
    }
    if (item < npairs + W) {
        const int i = item - npairs;
        const PreInt &pp = c.pre[be.pre_idx[i + 1]];
        double *G = ps_imu_blk(c) + (size_t)i * 768;
        if (!c.C->c.use_imu || pp.sum_dt > 10.0) { for (int e = lane; e < 768; e += 64) G[e] = 0; return; }
        double *raw_l = imu_lds + wave * 704, *M_l = raw_l + 472;
        const double *raw = c.imu_raw + (size_t)i * 15 * 31;
        for (int q = lane; q < 465; q += 64) raw_l[q] = raw[q];
        for (int q = lane; q < 225; q += 64) M_l[q] = pp.sqrt_info[q];
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        v4f64 a00 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
        imu_block_mfma(raw_l, M_l, li, lk, a00, a10, a11);
        for (int r = 0; r < 4; r++) {   // three 16 x 16 tiles, element (row, col) at row * 16 + col
            const int e = (lk + 4 * r) * 16 + li;
            G[e] = a00[r]; G[256 + e] = a10[r]; G[512 + e] = a11[r];
        }
        if (i == 0) tick(34);
        return;
    }
    // landmark rows: the part of the row the solver reads (zeros included), Hll and gl
    const int Kpad = (Fa + 3) & ~3;
    const int w0 = min(LW, (6 * W1 + 15) & ~15), e_lo = max(w0, (15 * W1) & ~15);
    const int *alist = c.pair_list + c.nres_cap - c.NL;
    if (!vext) {
        // compact records: eight lanes per landmark, lane = frame (f = l8, l8 + 8, l8 + 16): every 6-column block of the row is written
        // exactly once -- the coupling with frame f from the one residual that observes it, the start-frame block / Hll / gl from
        // sums over the residuals that are reduced across the eight lanes, zeros elsewhere -- with one batch of loads per lane
        const int rw = item - (npairs + W);                      // row wavefront 0 .. PS_ROW_WAVES - 1
        if (rw >= PS_ROW_WAVES) return;
        const int l8 = lane & 7;
        for (int ka = rw * 8 + (lane >> 3); ka < Kpad; ka += PS_ROW_WAVES * 8) {   // (every lane group of a wavefront makes the same number of trips or one fewer: the shuffles below stay inside a group)
        const bool live = true, real = ka < Fa;
        int stf = 0, kend = 0;
        const double *rec = c.res;
        if (real) {
            const int slot = alist[ka];
            const int r0 = c.lm_tmp[slot];
            stf = c.lm_start[slot];
            kend = min(c.lm_nobs[slot], nres - r0 + 1);
            rec = c.res + (size_t)r0 * 28;
        }
        double *row = c.Hpl + (size_t)(live ? ka : 0) * LW;
        double si[6] = {0, 0, 0, 0, 0, 0}, hll = 0, gg = 0;
        constexpr int NH = (VIO_MAXW + 1 + 7) / 8;
        double blk[NH][6];
        bool has[NH];
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const int f = l8 + 8 * h, k = f - stf;
            has[h] = real && f < W1 && k >= 1 && k < kend;
            const double2 *J2 = (const double2 *)(rec + (size_t)(has[h] ? k - 1 : 0) * 28);
            double a[6], b[6], e[6], g6[6];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const double2 va = J2[q], vb = J2[3 + q], ve = J2[7 + q], vf = J2[10 + q];
                a[2 * q] = va.x; a[2 * q + 1] = va.y; b[2 * q] = vb.x; b[2 * q + 1] = vb.y;
                e[2 * q] = ve.x; e[2 * q + 1] = ve.y; g6[2 * q] = vf.x; g6[2 * q + 1] = vf.y;
            }
            const double2 p0 = J2[6], p1 = J2[13];   // (inv_depth column, weighted residual) of the two rows
            const double l0 = has[h] ? p0.x : 0.0, l1 = has[h] ? p1.x : 0.0;
#pragma unroll
            for (int d = 0; d < 6; d++) {
                si[d] += a[d] * l0 + e[d] * l1;
                blk[h][d] = b[d] * l0 + g6[d] * l1;
            }
            hll += l0 * l0 + l1 * l1;
            gg += l0 * (has[h] ? p0.y : 0.0) + l1 * (has[h] ? p1.y : 0.0);
        }
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
#pragma unroll
            for (int d = 0; d < 6; d++) si[d] += __shfl_xor(si[d], off, 8);
            hll += __shfl_xor(hll, off, 8);
            gg += __shfl_xor(gg, off, 8);
        }
        if (live) {
#pragma unroll
            for (int h = 0; h < NH; h++) {
                const int f = l8 + 8 * h;
                if (f < W1) {
                    const bool start = real && f == stf;
#pragma unroll
                    for (int d = 0; d < 6; d++) row[6 * f + d] = start ? si[d] : (has[h] ? blk[h][d] : 0.0);
                }
            }
            for (int q = 6 * W1 + l8; q < w0; q += 8) row[q] = 0;
            if (l8 == 0) { c.Hll[ka] = hll; c.gl[ka] = gg; }
        }
        }
        if (rw == 0) tick(35);
        return;
    }
    const int rwx = item - (npairs + W);
    if (rwx >= PS_ROW_WAVES) return;
    for (int w = rwx * 64 + lane; w < 2 * Kpad; w += PS_ROW_WAVES * 64) {
        const int ka = w >> 1, half = w & 1;
        double *row = c.Hpl + (size_t)ka * LW;
        if (half == 0) for (int q = 0; q < w0; q++) row[q] = 0;
        else for (int q = e_lo; q < LW; q++) row[q] = 0;
        if (ka < Fa) {
            const int slot = alist[ka];
            const int stf = c.lm_start[slot], r0 = c.lm_tmp[slot];
            const int kend = min(c.lm_nobs[slot] + c.lm_relo[slot], nres - r0 + 1);   // (+ the landmark's relocalisation record, if any)
            lm_row(c.res + (size_t)r0 * 42, row, c.Hll + ka, c.gl + ka, half, stf, kend, 15 * W1, c.lm_relo[slot] ? c.lm_nobs[slot] : -1);
        } else if (half == 1) { c.Hll[ka] = 0; c.gl[ka] = 0; }
    }
    if (item == W1 * W1 + W) tick(35);
}

__global__ __launch_bounds__(512) void ps_asm_a_kernel(Batch B) { ps_asm_a_body(B); }
// VIO_ASM_A_OCC = 4: the same kernel held to 128 VGPRs (20 bytes of scratch) so that two workgroups share a compute unit
__global__ __launch_bounds__(512, 4) void ps_asm_a_kernel_occ4(Batch B) { ps_asm_a_body(B); }

// ---------------------------------------------------------------------------------------------------------------- EVAL + ASM_A fused
// Round 6: evaluation and the first half of the assembly in ONE kernel, so that the residual records (28 doubles per residual: 0.42 MB per
// sequence and iteration, written by ps_eval and read back twice by ps_asm_a) never leave the compute unit.  Grid (2 + chunks, S), 256
// threads, two workgroups per CU (78 KB of LDS each).  Workgroup 0 of a sequence: the prior (ps_eval's block 0).  Workgroup 1: the IMU factors,
// one work type per wavefront, raw Jacobians into LDS, then the ten IMU Gram blocks on the matrix cores.  Workgroup b >= 2: chunk b - 2 of the
// projection residuals (whole landmarks, at most PS_FUSE_CAP residuals, cut by ps_setup): thread = residual, record into LDS; then, from LDS,
//   * the frame-pair Gram blocks [J r]^T [J r] of the pairs the chunk holds (one wavefront per pair, the MFMA sequence of ps_asm_a; a pair
//     whose residuals span chunks gets one partial block per chunk and the chunk that publishes the last one sums them in chunk order:
//     deterministic, whichever finishes last), and
//   * the landmark rows of the chunk's landmarks (eight lanes per landmark, ps_asm_a's code on LDS records) -- into the half of the
//     double-buffered rows that is NOT current: the evaluated point may still be rejected (ps_sel_rows, ps_accept).
// Same mathematics as ps_eval + ps_asm_a: identical rows and IMU blocks, identical pair blocks where a pair sits in one chunk, a different
// association of the same sums where it is split.  Only solves with compact records take this path (extrinsic / td constant, no
// relocalisation factors, W <= PS_FUSE_MAXW: st.fused, ps_setup); the others keep ps_eval + ps_asm_a, whose launches idle for fused solves.
__global__ __launch_bounds__(256) void ps_evalf_kernel(Batch B) {
    // Block -> (sequence, role).  ROLE-major over the launch (every sequence's IMU workgroup first -- the longest --, then the priors, then chunk
    // 0 of every sequence, chunk 1, ...): with two workgroups per CU a 64-sequence launch has 576 workgroups for 512 slots, and in sequence-
    // major order the last eight sequences started -- IMU workgroup included -- when the first ones had finished (kernel = two full rounds,
    // 82 us median); now the overflow is the last chunk of every sequence, which many sequences do not even have.  The stride between roles is a
    // multiple of the XCD count, so every workgroup of sequence r still runs on XCD r % X (ps_blk).
    int s, b;
    if (B.xcd_nb == 0) { s = (int)blockIdx.y + B.s0; b = (int)blockIdx.x; }
    else {
        const int X = B.xcd_n, S8 = X * ((B.ns + X - 1) / X), L = (int)blockIdx.x, role = L / S8, r = L - role * S8;
        if (r >= B.ns) return;
        s = r + B.s0;
        b = role == 0 ? 1 : (role == 1 ? 0 : role);
    }
    const int t = threadIdx.x, nt = 256;
    constexpr int NW = 4;   // wavefronts
    SolveSt &st = B.sst[s];
    if (st.stage != PS_EVAL_X0 && st.stage != PS_EVAL_C) return;
    if (!st.fused || b >= st.n_eval_blocks) return;
    Ctx c = make_ctx(B, s);
    ps_sel_rows(B, c, st.rowbuf ^ 1);   // the rows of the point under evaluation
    const BeSeq &be = *c.be;
    const vio_config &cfg = c.C->c;
    const bool cand = st.stage == PS_EVAL_C;
    const double *feat = cand ? c.cfeat : c.feat;
    const bool withJ = st.eval_with_J != 0;
    const int W = c.W, W1 = W + 1, n = c.NPR, LW = c.LW;
    const int lane = t & 63, wave = t >> 6, li = lane & 15, lk = lane >> 4;
    __shared__ double sred[64];
    __shared__ Params X;
    {
        const double *src = (const double *)(cand ? &st.Xc : &st.X);
        for (int k = t; k < (int)(sizeof(Params) / sizeof(double)); k += nt) ((double *)&X)[k] = src[k];
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double cost = 0;
    PH_INIT;
    __syncthreads();
    if (b == 0) {
        // ---- prior gradient q = b + A dx and cost dx^T b + 1/2 dx^T A dx (the code of ps_eval's block 0), and the zero Gram blocks of frame pairs
        // without residuals (ps_asm_b reads every pair)
        double *dxs = (double *)smem, *pacc = dxs + ((n + 1) & ~1);
        if (withJ)
            for (int p = wave; p < W1 * W1; p += NW) {
                const int i = p / W1, j = p - i * W1;
                if (i < j && c.pair_start[p + 1] == c.pair_start[p]) { double *o = c.pairblk + (size_t)pair_slot(i, j, W1) * 210; for (int e = lane; e < 210; e += 64) o[e] = 0; }
            }
        if (be.has_prior) {
            prior_dx(c, X, dxs, true);
            const int nch = max(1, min(4, nt / n));
            if (t < nch * n) {
                // up to four threads per row, each over a share of the columns, every load of a thread in flight at once
                const int ch = t / n, row = t - ch * n;
                const int per = (n + nch - 1) / nch, j0 = ch * per, j1 = min(n, j0 + per);
                constexpr int MAXJ = 32;
                double acc = 0;
                if (j1 - j0 <= MAXJ) {
                    double hv[MAXJ];
#pragma unroll
                    for (int u = 0; u < MAXJ; u++) hv[u] = c.prior_H[(size_t)min(j0 + u, n - 1) * n + row];
#pragma unroll
                    for (int u = 0; u < MAXJ; u++) if (j0 + u < j1) acc += hv[u] * dxs[j0 + u];
                } else
                    for (int j = j0; j < j1; j += 8) {
                        double hv[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) hv[u] = c.prior_H[(size_t)min(j + u, j1 - 1) * n + row];
#pragma unroll
                        for (int u = 0; u < 8; u++) if (j + u < j1) acc += hv[u] * dxs[j + u];
                    }
                pacc[ch * n + row] = acc;
            }
            __syncthreads();
            if (t < n) {
                double acc = pacc[t];
                for (int ch = 1; ch < nch; ch++) acc += pacc[ch * n + t];
                const double b0 = c.prior_r[t], d = dxs[t], q = b0 + acc;
                st.srp[t] = q;
                st.sdx[t] = d;
                cost += 0.5 * d * (b0 + q);
            }
            if (t == 0) cost += 0.5 * be.prior_c0;
        }
        PH(62);
    } else if (b == 1) {
        // ---- IMU factors: headers staged in LDS, one work type per wavefront (whitened residual | d / d pose_i | d / d speed-bias_i | d / d pose_j and
        // d / d speed-bias_j -- the last one is three copies of two matrices), lane = factor, raw Jacobians into LDS; then the Gram blocks
        double *pl = (double *)smem;
        constexpr int PH_LD = VIO_PREINT_HDR + 1;
        double *rawl = pl + (size_t)W * PH_LD;                               // [W][15 x 31] raw Jacobians | whitened residual
        for (int q0 = t; q0 < W * PH_LD; q0 += 10 * nt) {   // (ten loads in flight per thread and trip)
            double v[10];
#pragma unroll
            for (int u = 0; u < 10; u++) {
                const int q = min(q0 + u * nt, W * PH_LD - 1), i = q / PH_LD, e = min(q - i * PH_LD, VIO_PREINT_HDR - 1);
                v[u] = ((const double *)&c.pre[be.pre_idx[i + 1]])[e];
            }
#pragma unroll
            for (int u = 0; u < 10; u++) if (q0 + u * nt < W * PH_LD) pl[q0 + u * nt] = v[u];
        }
        __syncthreads();
        {
            const v3 G = ld3(be.g);
            const int wt = wave;
            for (int i = lane; i < W; i += 64) {
                const int j = i + 1;
                const PreInt &p = *(const PreInt *)(pl + (size_t)i * PH_LD);   // only the header fields are read
                double *out = rawl + (size_t)i * 465;
                if (p.sum_dt > 10.0) { if (wt == 0) for (int k = 0; k < 15; k++) out[k * 31 + 30] = 0; continue; }
                if (wt == 0) {
                    double raw[15];
                    bf::imu_raw_residual(p, G, &X.pose[i * 7], &X.sb[i * 9], &X.pose[j * 7], &X.sb[j * 9], raw);
                    for (int r = 0; r < 15; r++) {
                        double sacc = 0;
                        for (int k = 0; k <= r; k++) sacc += p.sqrt_info[r * 15 + k] * raw[k];
                        out[r * 31 + 30] = sacc;
                        cost += 0.5 * sacc * sacc;
                    }
                } else if (withJ) {
                    bf::imu_raw_jacobian_part(p, G, &X.pose[i * 7], &X.sb[i * 9], &X.pose[j * 7], &X.sb[j * 9], wt - 1, out, 31);
                    if (wt == 3) bf::imu_raw_jacobian_part(p, G, &X.pose[i * 7], &X.sb[i * 9], &X.pose[j * 7], &X.sb[j * 9], 3, out, 31);
                }
            }
        }
        __syncthreads();
        if (withJ)
            for (int i = wave; i < W; i += NW) {   // IMU Gram blocks [Jw r]^T [Jw r] on the matrix cores, one wavefront per factor
                const PreInt &p = *(const PreInt *)(pl + (size_t)i * PH_LD);
                double *G = ps_imu_blk(c) + (size_t)i * 768;
                if (p.sum_dt > 10.0) { for (int e = lane; e < 768; e += 64) G[e] = 0; continue; }
                v4f64 a00 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
                imu_block_mfma(rawl + (size_t)i * 465, p.sqrt_info, li, lk, a00, a10, a11);
                for (int r = 0; r < 4; r++) {
                    const int e = (lk + 4 * r) * 16 + li;
                    G[e] = a00[r]; G[256 + e] = a10[r]; G[512 + e] = a11[r];
                }
            }
        PH(63);
    } else {
        // ---- chunks b - 2, b - 2 + G, ... of the projection residuals (G = B.fuse chunk workgroups per sequence in the grid: one chunk each in
        // the steady state, a second pass for the longer residual lists right after the initialisation): evaluate into LDS records, then Gram
        // blocks and landmark rows from LDS
        const int nblk = st.nblk;
        __shared__ int fis[PS_FUSE_MAXPAIRS];                      // this chunk's row of the pair table
        auto chunk = [&](const int pb) {
        const int r0 = st.blk_r[pb], r1 = st.blk_r[pb + 1], nin = r1 - r0, nres = st.nres;
        double *rec = (double *)smem;                              // [PS_FUSE_CAP][28]
        double *geo = rec + (size_t)PS_FUSE_CAP * 28;              // [W1 W / 2 + 1][32] frame-pair geometry by pair slot (i < j), then ric
        int (*lmm)[3] = (int (*)[3])(geo + (size_t)(W1 * W / 2 + 1) * 32);   // per variable landmark of the chunk (each has a residual: at most PS_FUSE_CAP): first record, start frame, observations
        const int Fa_ = st.Fa, ka_lo = st.blk_ka[pb], ka_n = min(min(st.blk_ka[pb + 1], Fa_) - ka_lo, PS_FUSE_CAP);
        if (t < W1 * W / 2) fis[t] = st.fi[pb][t];
        if (withJ && t < ka_n) {
            const int slot = (c.pair_list + c.nres_cap - c.NL)[ka_lo + t];
            lmm[t][0] = c.lm_tmp[slot]; lmm[t][1] = c.lm_start[slot]; lmm[t][2] = c.lm_nobs[slot];
        }
        // this thread's residual: its indices and inputs are fetched before the barrier (a chain of three dependent loads hidden behind the staging)
        int slot_r = 0, k_r = 1, imu_i = 0;
        double inv_dep = 1.0;
        const double *oi_p = nullptr, *oj_p = nullptr;
        if (t < nin) {
            slot_r = c.res_lm[r0 + t]; k_r = c.res_k[r0 + t];
            imu_i = c.lm_start[slot_r];
            inv_dep = feat[c.lm_pidx[slot_r]];
            oi_p = obs_ptr(c, slot_r, imu_i); oj_p = obs_ptr(c, slot_r, imu_i + k_r);
        }
        const int npairs = W1 * W / 2;
        for (int q = t; q <= npairs; q += nt) {
            if (q == npairs) { stm(geo + (size_t)q * 32, q2R(mkq(X.ex[6], X.ex[3], X.ex[4], X.ex[5]))); continue; }
            if (((fis_early(st, pb, q) >> 16) & 0x3FF) == 0) continue;   // (only the pairs this chunk evaluates)
            int i = 0, rem = q;
            while (rem >= W1 - 1 - i) { rem -= W1 - 1 - i; i++; }
            const int j = i + 1 + rem;
            bf::PairGeo g;
            bf::pair_geo(&X.pose[i * 7], &X.pose[j * 7], X.ex, g);
            double *o = geo + (size_t)q * 32;
            for (int e = 0; e < 9; e++) { o[e] = g.A1[e]; o[9 + e] = g.A2[e]; o[18 + e] = g.M[e]; }
            o[27] = g.t[0]; o[28] = g.t[1]; o[29] = g.t[2];
        }
        __syncthreads();
        const double *ricm = geo + (size_t)npairs * 32;
        if (t < nin) {
            const int imu_j = imu_i + k_r;
            const bf::PairGeo &g = *(const bf::PairGeo *)(geo + (size_t)pair_slot(imu_i, imu_j, W1) * 32);
            double rr[2], wgt = 1.0;
            double *out = rec + (size_t)t * 28;
            bf::eval_projection_pair(cfg, g, ricm, X.ex, inv_dep, X.td, oi_p, oj_p, cfg.estimate_td != 0, rr, withJ ? out : nullptr, true, &wgt, 14, false);
            const double sq = rr[0] * rr[0] + rr[1] * rr[1];
            if (withJ) { out[13] = wgt * rr[0]; out[27] = wgt * rr[1]; }
            cost += 0.5 * log(1.0 + sq);
        }
        PH(14);
        __syncthreads();
        if (withJ) {
            // (1) frame-pair Gram blocks of the pairs that have residuals in this chunk (table from ps_setup, staged in LDS before the evaluation);
            // one wavefront per pair
            {
                for (int slot_p = wave; slot_p < W1 * W / 2; slot_p += NW) {
                    const int e_ = fis[slot_p], cnt = (e_ >> 16) & 0x3FF;
                    if (cnt == 0) continue;
                    const int qlo = e_ & 0xFFFF, rank = (e_ >> 26) & 15;
                    v4f64 a00 = {0, 0, 0, 0};
                    const int lcol = li < 12 ? li : 13;
                    for (int base = 0; base < cnt; base += 64) {
                        const int nchunk = min(64, cnt - base);
                        const int myidx = c.pair_list[qlo + base + min(lane, nchunk - 1)] - r0;   // record index within the chunk
                        const int Kc = 2 * nchunk;
                        for (int k0 = 0; k0 < Kc; k0 += 4 * PB_U) {
                            double x0[PB_U];
#pragma unroll
                            for (int u = 0; u < PB_U; u++) {
                                const int kk = k0 + 4 * u + lk;
                                const int ridx = __shfl(myidx, min(kk, Kc - 1) >> 1, 64);
                                const double v0 = rec[(size_t)ridx * 28 + (kk & 1) * 14 + lcol];
                                x0[u] = (kk < Kc && li < 13) ? v0 : 0.0;
                            }
#pragma unroll
                            for (int u = 0; u < PB_U; u++) a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[u], x0[u], a00, 0, 0, 0);
                        }
                    }
                    const int npart = st.pm_np[slot_p];
                    double *part0 = B.pairpart + ((size_t)s * PS_FUSE_MAXPAIRS + slot_p) * PS_FUSE_MAXBLK * 210;
                    double *out = npart <= 1 ? c.pairblk + (size_t)slot_p * 210 : part0 + (size_t)rank * 210;
                    // (a pair whose residuals span chunks: this is one of its partial blocks; the chunk that finishes last adds them up, below.  Partial
                    // blocks cross workgroups inside this kernel: device-scope atomic stores -- write-through to the coherence point -- instead of a
                    // cache-wide fence, see the tail)
                    for (int r = 0; r < 4; r++) {
                        const int row = lk + 4 * r, col = li;
                        int e = -1;
                        if (row < 12) { if (col <= row) e = sym_idx(col, row); }
                        else if (row == 12 && col <= 12) e = sym_idx(col < 12 ? col : 19, 19);
                        if (e >= 0) { if (npart > 1) __hip_atomic_store(&out[e], a00[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else out[e] = a00[r]; }
                    }
                }
            }
            PH(32);
            // (2) landmark rows of the chunk's variable landmarks (the last chunk also writes the zero rows that pad to a multiple of four):
            // eight lanes per landmark, lane = frame (f = l8, l8 + 8) -- the coupling with frame f from the one residual that observes it, the
            // start-frame block / Hll / gl from sums over the residuals that are reduced across the eight lanes, zeros elsewhere
            {
                const int Fa = Fa_, Kpad = (Fa + 3) & ~3;
                const int ka_hi = pb == nblk - 1 ? Kpad : st.blk_ka[pb + 1];
                const int w0 = min(LW, (6 * W1 + 15) & ~15);
                const int l8 = lane & 7;
                for (int ka = ka_lo + wave * 8 + (lane >> 3); ka < ka_hi; ka += 8 * NW) {
                    const bool real = ka < Fa;
                    int stf = 0, kend = 0;
                    const double *lrec = rec;
                    if (real) {
                        const int rl = lmm[ka - ka_lo][0];
                        stf = lmm[ka - ka_lo][1];
                        kend = min(lmm[ka - ka_lo][2], nres - rl + 1);
                        lrec = rec + (size_t)(rl - r0) * 28;
                    }
                    double *row = c.Hpl + (size_t)ka * LW;
                    double si[6] = {0, 0, 0, 0, 0, 0}, hll = 0, gg = 0;
                    constexpr int NH = (PS_FUSE_MAXW + 1 + 7) / 8;
                    double blk[NH][6];
                    bool has[NH];
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        const int f = l8 + 8 * h, k = f - stf;
                        has[h] = real && f < W1 && k >= 1 && k < kend;
                        const double2 *J2 = (const double2 *)(lrec + (size_t)(has[h] ? k - 1 : 0) * 28);
                        double a[6], bb[6], e[6], g6[6];
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            const double2 va = J2[q], vb = J2[3 + q], ve = J2[7 + q], vf = J2[10 + q];
                            a[2 * q] = va.x; a[2 * q + 1] = va.y; bb[2 * q] = vb.x; bb[2 * q + 1] = vb.y;
                            e[2 * q] = ve.x; e[2 * q + 1] = ve.y; g6[2 * q] = vf.x; g6[2 * q + 1] = vf.y;
                        }
                        const double2 p0 = J2[6], p1 = J2[13];   // (inv_depth column, weighted residual) of the two rows
                        const double l0 = has[h] ? p0.x : 0.0, l1 = has[h] ? p1.x : 0.0;
#pragma unroll
                        for (int d = 0; d < 6; d++) {
                            si[d] += a[d] * l0 + e[d] * l1;
                            blk[h][d] = bb[d] * l0 + g6[d] * l1;
                        }
                        hll += l0 * l0 + l1 * l1;
                        gg += l0 * (has[h] ? p0.y : 0.0) + l1 * (has[h] ? p1.y : 0.0);
                    }
#pragma unroll
                    for (int off = 4; off >= 1; off >>= 1) {
#pragma unroll
                        for (int d = 0; d < 6; d++) si[d] += __shfl_xor(si[d], off, 8);
                        hll += __shfl_xor(hll, off, 8);
                        gg += __shfl_xor(gg, off, 8);
                    }
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        const int f = l8 + 8 * h;
                        if (f < W1) {
                            const bool start = real && f == stf;
#pragma unroll
                            for (int d = 0; d < 6; d++) row[6 * f + d] = start ? si[d] : (has[h] ? blk[h][d] : 0.0);
                        }
                    }
                    for (int q = 6 * W1 + l8; q < w0; q += 8) row[q] = 0;
                    if (l8 == 0) { c.Hll[ka] = hll; c.gl[ka] = gg; }
                }
            }
            PH(35);
        }
        };
        // (the first chunk as straight-line code: as a general loop the kernel was 6 % slower)
        chunk(b - 2);
        for (int pb = b - 2 + B.fuse; pb < nblk; pb += B.fuse) { __syncthreads(); chunk(pb); }   // (the previous chunk's records are still being read)
    }
    cost = block_sum(cost, sred);
    // No cache-wide fence (on gfx950 __threadfence() is buffer_wbl2 sc1 + buffer_inv sc1: the XCD's whole L2 written back AND invalidated -- a
    // fence per published partial block made every workgroup on the device seven times slower, and even one per workgroup costs 10 % of the
    // whole pipeline, see ps_eval_body).  What crosses workgroups inside this kernel: the partial cost (published by a returning device-scope
    // exchange; the counter increments depend on its return value) and the partial Gram blocks (device-scope atomic stores above, drained by
    // the barriers of block_sum before thread 0 gets here; read back with device-scope atomic loads).  Rows and blocks go to later kernels.
    __shared__ int last, lastc;
    if (t == 0) {
        const unsigned long long old = atomicExch((unsigned long long *)&st.part[b], (unsigned long long)__double_as_longlong(cost));
        int zero;
        asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"((int)(old >> 32)));
        const int mine = b >= 2 ? (st.nblk - (b - 2) + B.fuse - 1) / B.fuse : 0;   // chunks this workgroup took
        lastc = (mine > 0 && withJ) ? (atomicAdd(&st.chunk_done, mine + zero) == st.nblk - mine) : 0;
        last = atomicAdd(&st.eval_done, 1 + zero) == st.n_eval_blocks - 1;
    }
    __syncthreads();
    if (lastc) {
        // the chunk that finished last: frame pairs whose residuals span chunks have one partial Gram block per chunk -- added up in chunk
        // order (deterministic whichever chunk gets here), one wavefront per pair
        if (t == 0) st.chunk_done = 0;
        // (this is a serial tail of the kernel: every load of a pair in flight at once -- as a chain of volatile loads it took 25 us)
        for (int slot_p = wave; slot_p < W1 * W / 2; slot_p += NW) {
            const int npart = st.pm_np[slot_p];
            if (npart <= 1) continue;
            const double *part0 = B.pairpart + ((size_t)s * PS_FUSE_MAXPAIRS + slot_p) * PS_FUSE_MAXBLK * 210;
            double *dst = c.pairblk + (size_t)slot_p * 210;
            double v[PS_FUSE_MAXBLK][4];
#pragma unroll
            for (int q = 0; q < PS_FUSE_MAXBLK; q++)
#pragma unroll
                for (int u = 0; u < 4; u++) v[q][u] = __hip_atomic_load(&part0[(size_t)min(q, npart - 1) * 210 + min(lane + 64 * u, 209)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                double acc = v[0][u];
#pragma unroll
                for (int q = 1; q < PS_FUSE_MAXBLK; q++) if (q < npart) acc += v[q][u];
                if (lane + 64 * u < 210) dst[lane + 64 * u] = acc;
            }
        }
    }
    if (last && t < 64) {
        if (t == 0) st.eval_done = 0;
        ps_accept(B, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------- ASM_B
// grid (NB, S), 256 threads: one thread per entry (a, b) of H (both triangles, every entry of the P x LW block is written, so no
// zero-fill pass) and per entry of g: prior block, the (at most two) IMU Gram blocks that contain both columns, then the
// frame-pair sums -- the same terms in the same order as assemble().  The thread that owns a diagonal entry fixes the Jacobi
// column scaling the first time round.
// tri = true (VIO_ASM_B_MODE = 2, round 6): only the entries a >= b are formed (and the gradient); every thread stores its entry and the mirror
// image.  H's terms are symmetric source by source -- prior_H is stored exactly symmetric, the Gram blocks are read through sym_idx -- and are added
// in the same order for (a, b) and (b, a), so the mirrored H is the same bits with half the index arithmetic and half the gathers.
// One entry of H (bcol < LW) or of the gradient (bcol == LW): the prior block, the (at most two) IMU Gram blocks that contain both columns, then the
// frame-pair sums -- the same terms in the same order as assemble().  Shared by ps_asm_b_body and by the Schur tiles that form S themselves
// (ps_schur_body, B.form_s).  imu_ok: ps_imu_ok_mask.
__device__ __forceinline__ unsigned ps_imu_ok_mask(const Ctx &c, const BeSeq &be) {
    // IMU factors that exist (estimator.cpp:1212-1220 skips pre-integrations longer than 10 s): one lane per factor, then a mask -- looked up per
    // entry it was two dependent global loads (pre_idx, then sum_dt) in front of every IMU term
    const int W = c.W;
    const bool f_ok = (int)(threadIdx.x & 63) < W && c.C->c.use_imu && !(c.pre[be.pre_idx[min((int)(threadIdx.x & 63), W - 1) + 1]].sum_dt > 10.0);
    return (unsigned)__ballot(f_ok);
}
__device__ __forceinline__ double ps_h_entry(const Ctx &c, const SolveSt &st, const BeSeq &be, const unsigned imu_ok, const int a, const int bcol) {
    const int W = c.W, W1 = W + 1, LW = c.LW, n = c.NPR;
    const bool vext = st.vext != 0;
    const double *pb = c.pairblk, *ib = ps_imu_blk(c);
    const int oE = 15 * W1, oT = 15 * W1 + 6;
    // IMU local column of tangent index a in factor i (-1 if absent)
    auto imu_local = [&](int a, int i) -> int {
        if (a < 6 * W1) { const int f = a / 6, d = a - 6 * f; return f == i ? d : (f == i + 1 ? 15 + d : -1); }
        if (a < 15 * W1) { const int q = a - 6 * W1, f = q / 9, d = q - 9 * f; return f == i ? 6 + d : (f == i + 1 ? 21 + d : -1); }
        return -1;
    };
    auto imu_get = [&](int i, int la, int lb) -> double {   // Gram entry (la, lb), la, lb in 0 .. 30 (30 = residual column)
        const double *G = ib + (size_t)i * 768;
        if (la < lb) { const int x = la; la = lb; lb = x; }
        if (la < 16) return G[la * 16 + lb];
        if (lb < 16) return G[256 + (la - 16) * 16 + lb];
        return G[512 + (la - 16) * 16 + (lb - 16)];
    };
    auto prior_inv = [&](int a) -> int {   // tangent index -> prior index (-1: not in the prior layout)
        if (a < 6 * W) return a;
        if (a < 6 * W1) return -1;
        if (a < 6 * W1 + 9) return 6 * W + (a - 6 * W1);
        if (a < oE) return -1;
        if (a < oE + 6) return 6 * W + 9 + (a - oE);
        if (a == oT) return 6 * W + 15;
        return -1;
    };
    const bool grad = bcol == LW;
    const int b = grad ? -1 : bcol;
    double v = 0;
    // prior
    if (be.has_prior) {
        // (relocalisation solve: the extrinsic's columns belong to relo_Pose, which the prior knows nothing about)
        const bool lent_a = st.relo && a >= oE && a < oE + 6, lent_b = st.relo && b >= oE && b < oE + 6;
        const int pa = lent_a ? -1 : prior_inv(a);
        if (pa >= 0) {
            if (grad) v = st.srp[pa];
            else { const int pb_ = lent_b ? -1 : prior_inv(b); if (pb_ >= 0) v = c.prior_H[pa * n + pb_]; }
        }
    }
    // IMU factors: even ones first, then odd ones (the order assemble() adds them in)
    if (a < 15 * W1 && (grad || b < 15 * W1)) {
        const int fa = a < 6 * W1 ? a / 6 : (a - 6 * W1) / 9;
        for (int parity = 0; parity < 2; parity++)
            for (int i = fa - 1; i <= fa; i++) {
                if (i < 0 || i >= W || (i & 1) != parity) continue;
                if (!((imu_ok >> i) & 1u)) continue;
                const int la = imu_local(a, i);
                if (la < 0) continue;
                const int lb = grad ? 30 : imu_local(b, i);
                if (lb < 0) continue;
                v += imu_get(i, la, lb);
            }
    }
    // vision: a (and b) must be a pose column or, when they are variables, an extrinsic / td column
    const int ra = a < 6 * W1 ? a : ((vext && a >= oE && a < oE + 7) ? 6 * W1 + (a - oE) : -1);
    const int rb = grad ? 0 : (b < 6 * W1 ? b : ((vext && b >= oE && b < oE + 7) ? 6 * W1 + (b - oE) : -1));
    if (ra >= 0 && rb >= 0) {
        const int fa = ra < 6 * W1 ? ra / 6 : -1, fb = grad ? -1 : (rb < 6 * W1 ? rb / 6 : -1);
        double sacc = 0;
        if (!grad && fa >= 0 && fb >= 0 && fa != fb) {
            const int i = min(fa, fb), j = max(fa, fb);
            sacc = pb[(size_t)pair_slot(i, j, W1) * 210 + sym_idx(local_of(a, i, j, W), local_of(b, i, j, W))];
        } else if (fa >= 0 || fb >= 0) {
            // one term per other frame of the window: all loads issued first (a runtime-bound loop with the load inside waits for
            // every L2 round trip in turn -- ten of them per entry at W = 10), then summed in the same order
            const int f = fa >= 0 ? fa : fb;
            double pv[VIO_MAXW + 1];
#pragma unroll
            for (int o = 0; o <= VIO_MAXW; o++) {
                const bool use = o < W1 && o != f;
                const int oo = use ? o : (f == 0 ? 1 : 0);     // unused slots read a term that exists (and drop it)
                const int i = min(f, oo), j = max(f, oo);
                const int la = local_of(a, i, j, W), lb = grad ? 19 : local_of(b, i, j, W);
                pv[o] = pb[(size_t)pair_slot(i, j, W1) * 210 + sym_idx(la, lb)];
            }
#pragma unroll
            for (int o = 0; o <= VIO_MAXW; o++) if (o < W1 && o != f) sacc += pv[o];
        } else {
            for (int i = 0; i < W1; i++)
                for (int j = i + 1; j < W1; j++) {
                    const int la = local_of(a, i, j, W), lb = grad ? 19 : local_of(b, i, j, W);
                    sacc += pb[(size_t)pair_slot(i, j, W1) * 210 + sym_idx(la, lb)];
                }
        }
        v += sacc;
    }
    return v;
}

__device__ __forceinline__ void ps_asm_b_body(const Batch &B, int s, int blk, int nb_b, const bool tri = false) {
    const SolveSt &st = B.sst[s];
    if (st.stage != PS_ASM) return;
    Ctx c = make_ctx(B, s);
    const BeSeq &be = *c.be;
    const int W = c.W, W1 = W + 1, P = c.P, LW = c.LW, n = c.NPR;
    const bool vext = st.vext != 0;
    const double *pb = c.pairblk, *ib = ps_imu_blk(c);
    const int oE = 15 * W1, oT = 15 * W1 + 6;
    const int total = P * (LW + 1);   // column LW stands for the gradient entry of the row
    const unsigned imu_ok = ps_imu_ok_mask(c, be);
    const int ntri = P * (P + 1) / 2, total_t = ntri + P + P * (LW - P);   // tri: lower triangle, gradient, zero padding columns
    // B.form_s (round 6, windows whose Schur complement stays in HBM): once the column scaling is fixed, S = Sp (H - U) Sp + mu D^2 is formed HERE,
    // entry by entry as H is summed, instead of by ps_serial's one workgroup per sequence (its "tile load": 39 of 242 us per iteration at W = 20).
    // The tiles the landmark rows touch (ps_colmask) belong to their Schur block, which sums their entries of H itself and subtracts U; every
    // other tile of S is written below.  Same expressions on the same operands as ps_serial's load, hence the same bits.
    const bool form_s = tri && B.form_s && !st.scale_pending;
    const unsigned colmask = ps_colmask(W1, LW, vext);
    const double mu = st.mu;
    const double *spv = c.vec + LW;
    for (int w = blk * blockDim.x + threadIdx.x; w < (tri ? total_t : total); w += nb_b * blockDim.x) {
        int a, bcol;
        if (tri) {
            if (w < ntri) {   // w = a (a + 1) / 2 + b, b <= a: closed form with a one-step correction of the float square root
                a = (int)((sqrtf(8.0f * (float)w + 1.0f) - 1.0f) * 0.5f);
                if (a * (a + 1) / 2 > w) a--;
                if ((a + 1) * (a + 2) / 2 <= w) a++;
                bcol = w - a * (a + 1) / 2;
            }
            else if (w < ntri + P) { a = w - ntri; bcol = LW; }
            else { const int q = w - ntri - P; a = q / (LW - P); bcol = P + (q - a * (LW - P)); }
        } else { a = w / (LW + 1); bcol = w - a * (LW + 1); }
        const bool grad = bcol == LW;
        const int b = grad ? -1 : bcol;
        if (!grad && b >= P) { c.H[(size_t)a * LW + b] = 0; continue; }
        const int ti = a >> 4, tj = b >> 4;
        if (form_s && !grad && ((colmask >> ti) & 1u) && ((colmask >> tj) & 1u)) continue;   // (its Schur block's)
        const double v = ps_h_entry(c, st, be, imu_ok, a, bcol);
        if (grad) c.vec[a] = v;
        else {
            c.H[(size_t)a * LW + b] = v;
            if (tri && a != b) c.H[(size_t)b * LW + a] = v;
            if (form_s) {
                const double sa = spv[a], sb = spv[b];
                double sv = sa * sb * (v - 0.0);
                if (a == b) {
                    const double hs = sa * sa * v, dg = sqrt(fmin(fmax(hs, 1e-6), 1e32));
                    sv += mu * dg * dg;
                    if (sa == 0.0) sv = 1.0;
                }
                c.Sc[tl_idx(ti, tj, a & 15, b & 15)] = sv;
                if (ti == tj && a != b) c.Sc[tl_idx(ti, ti, b & 15, a & 15)] = sv;
            }
            if (a == b && st.scale_pending) {
                bool act = a < oE ? true : (a < oT ? st.ex_active != 0 : st.td_active != 0);
                if (!c.C->c.use_imu && (a < 6 || a >= 6 * W1)) act = false;   // VO mode: pose 0 constant, no speed-bias blocks
                c.vec[1 * LW + a] = act ? 1.0 / (1.0 + sqrt(v)) : 0.0;   // sp
            }
        }
    }
    // padding of g / sp beyond P and the landmark scaling
    if (blk == 0) {
        for (int a = P + threadIdx.x; a < LW; a += blockDim.x) { c.vec[a] = 0; if (st.scale_pending) c.vec[1 * LW + a] = 0; }
        if (form_s && P < LW) {
            // rows / columns P .. LW - 1 of the last row of tiles: unit diagonal, zeros elsewhere (what ps_serial's load makes of sp = 0)
            const int nbt = LW >> 4, tl = nbt - 1;
            for (int q = threadIdx.x; q < nbt * 256; q += blockDim.x) {
                const int tj = q >> 8, r = (q >> 4) & 15, cc = q & 15, row = 16 * tl + r, col = 16 * tj + cc;
                if ((row < P && col < P) || (((colmask >> tl) & 1u) && ((colmask >> tj) & 1u))) continue;
                c.Sc[tl_idx(tl, tj, r, cc)] = row == col ? 1.0 : 0.0;
            }
        }
    }
}

// Round 5: the same sums organised by BLOCK PAIRS of the tangent space instead of by entries.  The per-entry version above spends ~330 vector
// instructions per entry on finding out which prior / IMU / frame-pair blocks contain it (9.9 M VALU instructions per launch,
// profiles/round4_pmc_sq.json) -- 55 % of the solver's thread-time at large batches.  Here one wavefront takes one pair of parameter blocks
// (pose k: 6 columns, speed-bias k: 9, extrinsic: 6, td: 1; lower triangle + one gradient item per block = 324 items at W = 10): which sources
// contribute is decided once per item on the scalar unit, a lane only adds (row, column) within the block, and the upper triangle is written
// as the mirror image.  Same terms in the same order as ps_asm_b_body, hence the same H and g.
__device__ __forceinline__ void ps_tblock(int q, int W1, int &off, int &sz, int &kind, int &frame) {
    if (q < W1) { off = 6 * q; sz = 6; kind = 0; frame = q; }
    else if (q < 2 * W1) { off = 6 * W1 + 9 * (q - W1); sz = 9; kind = 1; frame = q - W1; }
    else if (q == 2 * W1) { off = 15 * W1; sz = 6; kind = 2; frame = -1; }
    else { off = 15 * W1 + 6; sz = 1; kind = 3; frame = -1; }
}
__device__ __forceinline__ void ps_asm_b_blocks(const Batch &B, int s, int blk, int nb_b) {
    const SolveSt &st = B.sst[s];
    if (st.stage != PS_ASM) return;
    Ctx c = make_ctx(B, s);
    const BeSeq &be = *c.be;
    const int W = c.W, W1 = W + 1, P = c.P, LW = c.LW, n = c.NPR;
    const bool vext = st.vext != 0, relo = st.relo != 0, has_prior = be.has_prior != 0;
    const double *pb = c.pairblk, *ib = ps_imu_blk(c);
    const int oE = 15 * W1, oT = oE + 6;
    const int NBLK = 2 * W1 + 2, NTRI = NBLK * (NBLK + 1) / 2, NITEM = NTRI + NBLK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // IMU factors that exist (estimator.cpp:1212-1220 skips pre-integrations longer than 10 s): one lane per factor, then a mask
    const bool f_ok = lane < W && c.C->c.use_imu && !(c.pre[be.pre_idx[min(lane, W - 1) + 1]].sum_dt > 10.0);
    const unsigned imu_ok = (unsigned)__ballot(f_ok);
    const bool scale_pending = st.scale_pending != 0;
    for (int item0 = blk * nw + wave; item0 < NITEM; item0 += nb_b * nw) {
        const int item = __builtin_amdgcn_readfirstlane(item0);
        const bool grad = item >= NTRI;
        int br, bc;
        if (grad) { br = item - NTRI; bc = 0; } else tri_decode(item, br, bc);   // br >= bc
        int ro, rs, kr, fr, co, cs, kc, fc;
        ps_tblock(br, W1, ro, rs, kr, fr);
        ps_tblock(bc, W1, co, cs, kc, fc);
        if (grad) { cs = 1; kc = 4; fc = -1; }
        // prior (relocalisation solve: the extrinsic's columns belong to relo_Pose, which the prior knows nothing about)
        auto prior_base = [&](int kind, int f) -> int {
            if (kind == 0) return f < W ? 6 * f : -1;
            if (kind == 1) return f == 0 ? 6 * W : -1;
            if (kind == 2) return relo ? -1 : 6 * W + 9;
            return 6 * W + 15;
        };
        const int pr = has_prior ? prior_base(kr, fr) : -1, pc = grad ? 0 : (has_prior ? prior_base(kc, fc) : -1);
        // IMU factors that contain both blocks, even factors first (the order assemble() adds them in)
        // (of the two factors fr - 1 and fr that touch frame fr one is even, one odd: two fixed slots in that order, no indexed arrays)
        const int fi0 = (fr & 1) ? fr - 1 : fr, fi1 = (fr & 1) ? fr : fr - 1;
        auto imu_slot = [&](int i, int &la0, int &lb0) -> bool {
            if (!(kr <= 1 && (grad || kc <= 1)) || i < 0 || i >= W || !((imu_ok >> i) & 1u)) return false;
            lb0 = 30;
            if (!grad) { if (fc != i && fc != i + 1) return false; lb0 = (kc == 0 ? 0 : 6) + (fc == i ? 0 : 15); }
            la0 = (kr == 0 ? 0 : 6) + (fr == i ? 0 : 15);
            return true;
        };
        int la_0 = 0, lb_0 = 0, la_1 = 0, lb_1 = 0;
        const bool imu0 = imu_slot(fi0, la_0, lb_0), imu1 = imu_slot(fi1, la_1, lb_1);
        auto imu_get = [&](int i, int la, int lb) -> double {   // Gram entry (la, lb) of factor i, la, lb in 0 .. 30 (30 = residual column)
            const double *G = ib + (size_t)i * 768;
            if (la < lb) { const int x = la; la = lb; lb = x; }
            return la < 16 ? G[la * 16 + lb] : (lb < 16 ? G[256 + (la - 16) * 16 + lb] : G[512 + (la - 16) * 16 + (lb - 16)]);
        };
        // vision: 0 none, 1 one frame pair, 2 one term per other frame of the window, 3 every frame pair
        const bool vr = kr == 0 || (vext && kr >= 2), vc = grad || kc == 0 || (vext && (kc == 2 || kc == 3));
        int vmode = 0, vf = -1;
        if (vr && vc) {
            if (!grad && kr == 0 && kc == 0 && fr != fc) vmode = 1;
            else if (kr == 0 || kc == 0) { vmode = 2; vf = kr == 0 ? fr : fc; }
            else vmode = 3;
        }
        const int la_fix = kr == 2 ? 12 : 18, lb_fix = grad ? 19 : (kc == 2 ? 12 : 18);   // local column of a non-pose block in any frame pair
        for (int e = lane; e < rs * cs; e += 64) {
            const int r = cs == 6 ? e / 6 : (cs == 9 ? e / 9 : e), cc = e - r * cs, a = ro + r, b = co + cc;   // (constant divisors)
            double v = 0;
            if (pr >= 0 && pc >= 0) v = grad ? st.srp[pr + r] : c.prior_H[(pr + r) * n + pc + cc];
            if (imu0) v += imu_get(fi0, la_0 + r, lb_0 + (grad ? 0 : cc));
            if (imu1) v += imu_get(fi1, la_1 + r, lb_1 + (grad ? 0 : cc));
            if (vmode == 1) {
                const int i = min(fr, fc), j = max(fr, fc);
                v += pb[(size_t)pair_slot(i, j, W1) * 210 + sym_idx((fr == i ? 0 : 6) + r, (fc == i ? 0 : 6) + cc)];
            } else if (vmode == 2) {
                // one term per other frame of the window: all loads issued first, then summed in frame order
                double pv[VIO_MAXW + 1];
#pragma unroll
                for (int o = 0; o <= VIO_MAXW; o++) {
                    const bool use = o < W1 && o != vf;
                    const int oo = use ? o : (vf == 0 ? 1 : 0);     // unused slots read a term that exists (and drop it)
                    const int i = min(vf, oo), j = max(vf, oo), lf = vf == i ? 0 : 6;
                    const int la = (kr == 0 ? lf : la_fix) + r, lb = (grad ? 19 : (kc == 0 ? lf : lb_fix) + cc);
                    pv[o] = pb[(size_t)pair_slot(i, j, W1) * 210 + sym_idx(la, lb)];
                }
                double sacc = 0;
#pragma unroll
                for (int o = 0; o <= VIO_MAXW; o++) if (o < W1 && o != vf) sacc += pv[o];
                v += sacc;
            } else if (vmode == 3) {
                double sacc = 0;
                const int li = sym_idx(la_fix + r, lb_fix + (grad ? 0 : cc));
                for (int q = 0; q < W1 * W / 2; q++) sacc += pb[(size_t)q * 210 + li];   // pair slots in (i, j) order
                v += sacc;
            }
            if (grad) c.vec[a] = v;
            else {
                c.H[(size_t)a * LW + b] = v;
                if (br != bc) c.H[(size_t)b * LW + a] = v;
                else if (a == b && scale_pending) {
                    bool act = a < oE ? true : (a < oT ? st.ex_active != 0 : st.td_active != 0);
                    if (!c.C->c.use_imu && (a < 6 || a >= 6 * W1)) act = false;   // VO mode: pose 0 constant, no speed-bias blocks
                    c.vec[1 * LW + a] = act ? 1.0 / (1.0 + sqrt(v)) : 0.0;   // sp
                }
            }
        }
    }
    // padding: columns P .. LW - 1 of H, g / sp beyond P
    if (blk == 0) {
        for (int w = threadIdx.x; w < P * (LW - P); w += blockDim.x) { const int a = w / (LW - P), b = P + (w - a * (LW - P)); c.H[(size_t)a * LW + b] = 0; }
        for (int a = P + threadIdx.x; a < LW; a += blockDim.x) { c.vec[a] = 0; if (scale_pending) c.vec[1 * LW + a] = 0; }
    }
}

// ---------------------------------------------------------------------------------------------------------------- SCHUR
// grid (active tiles, S), 64 threads: one wavefront forms one 16 x 16 lower tile of S = S_p H S_p + mu D^2 - sum_k w_k h_k h_k^T
// over all landmark rows, operands straight from HBM / L2 in batches of 8 k-steps, into c.Sc in the LDS tile layout.
#define PS_SCH_U 8
// U = sum_k hpl_k^T hpl_k w_k, the landmark part of the Schur complement S = Sp (H - U) Sp + mu D^2, for one active 16 x 16 tile on the
// FP64 matrix cores (one wavefront).  It needs nothing ps_asm_b produces (neither H nor the column scaling Sp), which is why the two run
// side by side in one launch; ps_serial combines H, U, Sp and mu D^2 while it loads the tiles into LDS.
__device__ __forceinline__ void ps_schur_body(const Batch &B, int s, int tile_index, double *wk_s) {
    const SolveSt &st = B.sst[s];
    if (st.stage != PS_ASM && st.stage != PS_SCHUR) return;
    Ctx c = make_ctx(B, s);
    ps_sel_rows(B, c, st.rowbuf);
    const int W1 = c.W + 1, LW = c.LW, nb = LW >> 4;
    const unsigned colmask = ps_colmask(W1, LW, st.vext != 0);
    // the tile_index-th active tile (ti >= tj, both column tiles active)
    int ti = -1, tj = -1, cnt = 0;
    for (int a = 0; a < nb && ti < 0; a++)
        for (int b = 0; b <= a; b++)
            if (((colmask >> a) & 1u) && ((colmask >> b) & 1u)) { if (cnt == tile_index) { ti = a; tj = b; break; } cnt++; }
    if (ti < 0) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6, li = lane & 15, lk = lane >> 4;
    const double *Ws = c.Hpl;
    const double mu = st.mu;
    const int Fa = st.Fa, Kpad = (Fa + 3) & ~3;
    const bool first = st.scale_pending != 0;
    v4f64 acc = {0, 0, 0, 0};
    const double *wa = Ws + 16 * ti + li, *wb = Ws + 16 * tj + li;
    // B.form_s: this block also sums its tile's entries of H (one per thread; ps_asm_b_body skips them) and stores S instead of U -- see
    // ps_asm_b_body.  The gathers are issued ahead of the landmark-row walk below.
    const bool form_s = B.form_s && st.stage == PS_ASM && !first;
    double *hbuf = wk_s + ((Kpad + 7) & ~7) + 3 * 256;
    if (form_s) {
        const BeSeq &be = *c.be;
        const unsigned imu_ok = ps_imu_ok_mask(c, be);
        const int r = t >> 4, cc = t & 15, row = 16 * ti + r, col = 16 * tj + cc, P = c.P;
        const int a = max(row, col), b = min(row, col);
        const double h = a < P ? ps_h_entry(c, st, be, imu_ok, a, b) : 0.0;
        hbuf[t] = h;
        if (a < P) { c.H[(size_t)row * LW + col] = h; if (ti != tj) c.H[(size_t)col * LW + row] = h; }
    }
    // per-row factor sl^2 / (sl^2 Hll + mu dgl^2), sl = 1 / (1 + sqrt(Hll)) at the first linearisation: once per row into LDS (a square
    // root and two divisions each), not once per lane and trip
    for (int kc = t; kc < Kpad; kc += blockDim.x) {
        double hll = c.Hll[kc], slk = first ? (kc < Fa ? 1.0 / (1.0 + sqrt(hll)) : 0.0) : c.lvec[kc];
        const double hl = kc < Fa ? slk * slk * hll : 0.0;
        const double dl = sqrt(fmin(fmax(hl, 1e-6), 1e32));
        const double iv = kc < Fa ? 1.0 / (hl + mu * dl * dl) : 0.0;
        wk_s[kc] = slk * slk * iv;
    }
    __syncthreads();
    // the k range is dealt to the wavefronts of the block trip by trip (a single wavefront walking all of it was a chain of ~7 dependent
    // L2 round trips); the partial tiles are added in wavefront order
    for (int k0 = 4 * PS_SCH_U * wave; k0 < Kpad; k0 += 4 * PS_SCH_U * nw) {
        double a[PS_SCH_U], b[PS_SCH_U];
#pragma unroll
        for (int u = 0; u < PS_SCH_U; u++) {
            const int kk = k0 + 4 * u + lk;
            const bool valid = kk < Kpad;
            const int kc = min(kk, Kpad - 1);
            const double wk = wk_s[kc];
            const double va = wa[(size_t)kc * LW], vb = wb[(size_t)kc * LW];
            a[u] = valid ? va * wk : 0.0;
            b[u] = valid ? vb : 0.0;
        }
#pragma unroll
        for (int u = 0; u < PS_SCH_U; u++) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
    }
    double *part = wk_s + ((Kpad + 7) & ~7);   // [nw - 1][256] partial tiles of wavefronts 1 ..
    if (wave > 0)
        for (int r = 0; r < 4; r++) part[(wave - 1) * 256 + r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < nw; w++)
            for (int r = 0; r < 4; r++) acc[r] += part[(w - 1) * 256 + r * 64 + lane];
        if (form_s) {
            const double *spv = c.vec + LW;
            const double sc = spv[16 * tj + li];
            for (int r = 0; r < 4; r++) {
                const int rl = lk + 4 * r, row = 16 * ti + rl, col = 16 * tj + li;
                const double sr = spv[row], h = hbuf[rl * 16 + li];
                double sv = sr * sc * (h - acc[r]);
                if (row == col) {
                    const double hs = sr * sr * h, dg = sqrt(fmin(fmax(hs, 1e-6), 1e32));
                    sv += mu * dg * dg;
                    if (sr == 0.0) sv = 1.0;
                }
                acc[r] = sv;
            }
        }
        for (int r = 0; r < 4; r++) c.Sc[tl_idx(ti, tj, lk + 4 * r, li)] = acc[r];
    }
}

// Column sums out[a] = sum_k u[k] M[k][a] of the landmark rows, with the BITS of matvec_pass_2range / matvec_pass_t run by ps_serial's eight wavefronts:
// wavefront w of this 256-thread block plays wavefronts w and w + 4 of that pass one after the other (rows k = w', w' + 8, ..., RB of them per trip,
// accumulated in that order), then the eight partial sums are added in wavefront order.  Only the columns [0, n0) and [e0, e0 + ne) are walked (the
// others of a landmark row are identically zero: the dense pass ps_serial falls back to beyond 128 such columns sums exact zeros there, so its result
// is the same bits too); the partial rows are kept compact, 64 NC doubles each.
template <int NC, int RB>
__device__ __forceinline__ void ps_colsum_as_eight_waves(const double *M, int ld, int nrows, int n, int n0, int e0, int ne, const double *u, double *out_col, double *part) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ncomp = n0 + ne;
    int col[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) { const int q = lane + 64 * j; col[j] = q < n0 ? q : (q < ncomp ? e0 + (q - n0) : -1); }
    for (int vw = wave; vw < 8; vw += 4) {
        double cs[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) cs[j] = 0;
        for (int k0 = vw; k0 < nrows; k0 += RB * 8) {
            double m[RB][NC], uk[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) {
                const int k = k0 + b * 8;
                const double *r = M + (size_t)min(k, nrows - 1) * ld;
#pragma unroll
                for (int j = 0; j < NC; j++) m[b][j] = col[j] >= 0 ? r[col[j]] : 0.0;
                uk[b] = k < nrows ? u[k] : 0.0;
            }
#pragma unroll
            for (int b = 0; b < RB; b++) {
                if (k0 + b * 8 >= nrows) break;
#pragma unroll
                for (int j = 0; j < NC; j++) cs[j] += uk[b] * m[b][j];
            }
        }
#pragma unroll
        for (int j = 0; j < NC; j++) if (col[j] >= 0) part[vw * (64 * NC) + lane + 64 * j] = cs[j];
    }
    __syncthreads();
    for (int a = t; a < n; a += blockDim.x) {
        double sacc = 0;
        const int qa = a < n0 ? a : ((a >= e0 && a < e0 + ne) ? n0 + (a - e0) : -1);
        if (qa >= 0)
            for (int q = 0; q < 8; q++) sacc += part[q * (64 * NC) + qa];
        out_col[a] = sacc;
    }
}
// B.gn_ext (round 6): the landmark term of the Gauss-Newton right-hand side, Hpl^T (sl inv gls), needs nothing of this launch's H or of ps_serial's
// prepare_point beyond what the Schur tiles use (Hll, gl, the landmark scaling and mu) -- one more workgroup of the Schur launch forms it (into the
// tmpv slot of c.vec) beside the tiles instead of ps_serial's one workgroup per sequence in front of its Cholesky (7 of 93 us per iteration).
__device__ __forceinline__ void ps_gn_rhs_body(const Batch &B, int s, double *wk_s) {
    const SolveSt &st = B.sst[s];
    if (st.stage != PS_ASM && st.stage != PS_SCHUR) return;
    Ctx c = make_ctx(B, s);
    ps_sel_rows(B, c, st.rowbuf);
    const int W1 = c.W + 1, LW = c.LW, P = c.P, t = threadIdx.x;
    const double mu = st.mu;
    const int Fa = st.Fa, Kpad = (Fa + 3) & ~3;
    const bool first = st.scale_pending != 0;
    for (int kc = t; kc < Kpad; kc += blockDim.x) {   // (the expressions of ps_serial's prepare_point and Gauss-Newton step)
        const double hll = c.Hll[kc], slk = first ? (kc < Fa ? 1.0 / (1.0 + sqrt(hll)) : 0.0) : c.lvec[kc];
        const double hl = kc < Fa ? slk * slk * hll : 0.0;
        const double gls = kc < Fa ? slk * c.gl[kc] : 0.0;
        const double dl = sqrt(fmin(fmax(hl, 1e-6), 1e32));
        const double iv = kc < Fa ? 1.0 / (hl + mu * dl * dl) : 0.0;
        wk_s[kc] = slk * iv * gls;
    }
    __syncthreads();
    double *part = wk_s + ((Kpad + 7) & ~7);
    double *out = c.vec + 7 * LW;   // tmpv (ps_serial_body's slot 7)
    const int n0 = 6 * W1, e0 = 15 * W1, ne = st.vext ? 7 : 0;
    if (n0 + ne <= 128) ps_colsum_as_eight_waves<2, 8>(c.Hpl, LW, Fa, P, n0, e0, ne, wk_s, out, part);   // matvec_pass_2range<8>
    else ps_colsum_as_eight_waves<3, 8>(c.Hpl, LW, Fa, P, n0, e0, ne, wk_s, out, part);                  // (ps_serial: the dense matvec_pass_t<6>; 6 (W + 1) + 7 <= 133 columns here)
}

// one launch: blocks [0, nb_b) sum the entries of H and the gradient, the blocks behind them form the landmark part of the
// Schur complement tile by tile (and, B.gn_ext, one more the landmark term of the Gauss-Newton right-hand side)
// five workgroups per CU (96 VGPRs, two spilled; 30 KB of LDS each): the launch is 2 944 workgroups at S = 64 and ran in three rounds of 1 024 at 112 VGPRs
__global__ __launch_bounds__(256, 5) void ps_asm_b_schur_kernel(Batch B, int nb_b, int by_blocks) {
    int s, b;
    if (!ps_blk(B, s, b)) return;
    extern __shared__ double ps_wk_s[];
    if (B.gn_ext && b == nb_b + B.n_schur) ps_gn_rhs_body(B, s, ps_wk_s);
    else if (b < nb_b) { if (by_blocks == 1) ps_asm_b_blocks(B, s, b, nb_b); else if (by_blocks == 2) ps_asm_b_body(B, s, b, nb_b, true); else ps_asm_b_body(B, s, b, nb_b, false); }
    else ps_schur_body(B, s, b - nb_b, ps_wk_s);
}

// ---------------------------------------------------------------------------------------------------------------- SERIAL
// grid S, 512 threads, dynamic LDS = xs + the 16 x 16 tiles of S: prepare_point, Cholesky, triangular solves, landmark
// back-substitution, dogleg, model decrease, candidate -- the serial spine of one trust-region iteration.
// TB: tiles a thread keeps in flight per trip of the tile load (4 at 1024 threads = 128 VGPRs, 8 at 512 threads)
template <bool BIG, int TB> __device__ __forceinline__ void ps_serial_body(const Batch &B) {
    const int s = blockIdx.x + B.s0, t = threadIdx.x, nt = blockDim.x;
    SolveSt &st = B.sst[s];
    if (st.stage != PS_ASM && st.stage != PS_SCHUR && st.stage != PS_STEP) return;
    Ctx c = make_ctx(B, s);
    ps_sel_rows(B, c, st.rowbuf);
    const vio_config &cfg = c.C->c;
    const int W = c.W, W1 = W + 1, P = c.P, LW = c.LW;
    __shared__ double sred[64];
    __shared__ int sh_i[8];
    __shared__ double chol_dinv[VIO_LWMAX];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *xs = (double *)smem, *work = xs + LW;
    const int F = st.F, Fa = st.Fa;
    const int ex_active = st.ex_active, td_active = st.td_active;
    const int ne_ext = st.vext ? 7 : 0;
    const int oE = 15 * W1, oT = 15 * W1 + 6;
    const int *alist = c.pair_list + c.nres_cap - c.NL;
    // The fourteen P-sized vectors of the trust-region step live in LDS for the duration of the kernel: as HBM arrays every small vector
    // phase below paid a dependent global round trip (about 25 per iteration).  One coalesced load here, one write-back in finish();
    // same layout as c.vec (slot 0 = g and slot 1 = sp come from ps_asm_b, the others persist between the slots of a solve).
    const int nbk = LW >> 4;
    const int wk_d = max(BIG ? (2 * nbk + 1) * 256 : nbk * (nbk + 1) / 2 * 256, 16 * 336);   // doubles of the work region (host: lds_serial); BIG: two block columns + the look-ahead tile (chol_tiles_stream)
    double *lv = work + wk_d + 2;
    double *g = lv, *sp = lv + 1 * LW, *dgp = lv + 2 * LW, *gradp = lv + 3 * LW, *gnp = lv + 4 * LW, *stp = lv + 5 * LW,
           *gs = lv + 6 * LW, *tmpv = lv + 7 * LW, *delta = lv + 8 * LW, *sgp = lv + 9 * LW, *hsgp = lv + 10 * LW,
           *yp = lv + 11 * LW, *up = lv + 12 * LW, *tmpv2 = lv + 13 * LW;
    double *sl = c.lvec, *dgl = c.lvec + c.NLs, *gradl = c.lvec + 2 * c.NLs, *gnl = c.lvec + 3 * c.NLs, *stl = c.lvec + 4 * c.NLs,
           *inv = c.lvec + 5 * c.NLs, *gls = c.lvec + 6 * c.NLs, *Hlls = c.lvec + 7 * c.NLs;
    const int Kpad = (Fa + 3) & ~3;
    double *hsgl = c.res + (size_t)c.nres_cap * 42 - 4 * (size_t)c.NLs;
    double *yl = hsgl + c.NLs, *ul = yl + c.NLs, *tmpl = ul + c.NLs;
    const int stage0 = st.stage;
    // (B.form_s: ps_asm_b_schur has left S itself in c.Sc -- whenever it ran at PS_ASM with the column scaling already fixed)
    const bool s_formed = B.form_s && stage0 == PS_ASM && st.scale_pending == 0;
    double radius = st.radius, mu = st.mu, alpha = st.alpha, dogleg_norm = st.dogleg_norm;
    bool cauchy_valid = st.cauchy_valid != 0;
    int invalid = st.invalid;
    int iter = st.iter;
    if (!st.retry) iter++;
    PH_INIT;
    const bool have_lv = !(iter > cfg.max_iterations);   // (the slot that only closes a solve at the iteration cap touches none of them)
    if (have_lv) for (int i = t; i < PS_LVEC * LW; i += nt) lv[i] = c.vec[i];
    __syncthreads();
    auto finish = [&](int new_stage) {
        __syncthreads();
        if (have_lv) for (int i = 2 * LW + t; i < PS_LVEC * LW; i += nt) c.vec[i] = lv[i];
        if (t == 0) {
            st.radius = radius; st.mu = mu; st.alpha = alpha; st.dogleg_norm = dogleg_norm; st.cauchy_valid = cauchy_valid ? 1 : 0;
            st.invalid = invalid; st.iter = iter; st.stage = new_stage;
        }
    };
    if (iter > cfg.max_iterations) { if (t == 0) st.iters_done = cfg.max_iterations; iter = cfg.max_iterations; finish(PS_DONE); return; }
    if (t == 0) { st.iters_done = iter; st.retry = 0; }
    if (stage0 == PS_ASM || stage0 == PS_SCHUR) {
        if (st.scale_pending) {
            for (int k = t; k < Kpad; k += nt) sl[k] = k < Fa ? 1.0 / (1.0 + sqrt(c.Hll[k])) : 0.0;
            __syncthreads();
            if (t == 0) st.scale_pending = 0;
        }
        if (st.point_new) {
            // prepare_point: gradient max-norm, scaled gradient, trust-region diagonal
            double m = 0;
            for (int a = t; a < P; a += nt) m = fmax(m, sp[a] != 0.0 ? fabs(g[a]) : 0.0);
            for (int k = t; k < Fa; k += nt) m = fmax(m, fabs(c.gl[k]));
            const double gmax = block_max(m, sred);
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
            if (t == 0) st.point_new = 0;
            if (gmax <= 1e-10) { if (t == 0) st.iters_done = iter - 1; finish(PS_DONE); return; }
        }
        PH(48);
        cauchy_valid = false;
        // Gauss-Newton step through the Schur complement (its landmark part U comes from ps_asm_b_schur_kernel at this mu)
        for (int k = t; k < Kpad; k += nt) {
            double iv = k < Fa ? 1.0 / (Hlls[k] + mu * dgl[k] * dgl[k]) : 0.0;
            inv[k] = iv;
            if (!B.gn_ext) tmpl[k] = sl[k] * iv * gls[k];
        }
        __syncthreads();
        // (B.gn_ext: tmpv = Hpl^T (sl inv gls) came in with the vectors, formed by ps_gn_rhs_body in the Schur launch at this mu)
        if (!B.gn_ext) matvec_pass_2range<TB>(c.Hpl, LW, Fa, P, 6 * W1, 15 * W1, ne_ext, tmpl, nullptr, tmpv, nullptr, work);
        for (int a = t; a < LW; a += nt) xs[a] = a < P ? gs[a] - sp[a] * tmpv[a] : 0.0;
        __syncthreads();
        PH(49);
        // windows whose Schur complement does not fit LDS as tiles (W > 10): S stays in HBM / L2 (in place in c.Sc) and the Cholesky
        // streams it through LDS one block column at a time (chol_tiles_stream)
        constexpr bool big = BIG;
        double *Stiles = big ? c.Sc : work;
        if (s_formed) {
            // S is in c.Sc already (tile layout): the LDS-resident Cholesky only needs it copied, 16 bytes per load, eight loads in flight per thread
            if (!big) {
                const int nd2 = (LW >> 4) * ((LW >> 4) + 1) / 2 * 128;   // double2 elements
                const double2 *src = (const double2 *)c.Sc;
                double2 *dst = (double2 *)work;
                for (int i0 = t; i0 < nd2; i0 += 8 * nt) {
                    double2 v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = src[min(i0 + u * nt, nd2 - 1)];
#pragma unroll
                    for (int u = 0; u < 8; u++) if (i0 + u * nt < nd2) dst[i0 + u * nt] = v[u];
                }
                __syncthreads();
            }
        } else {
            const unsigned colmask = ps_colmask(W1, LW, st.vext != 0);
            const int nb = LW >> 4, ntile = nb * (nb + 1) / 2;
            // thread = the element pair (r, c2), (r, c2 + 1) of every (nt / 128)-th tile, TB tiles per trip: 16-byte loads of H (row-major) and
            // of U (tile layout: the XOR swizzle keeps an even-aligned pair adjacent, swapped in odd rows), branch-free so that all of a
            // trip's loads are in flight together; 66 tiles = three trips of 512 threads
            const int e = t & 127, r = e >> 3, c2 = (e & 7) << 1, tstep = nt >> 7;
            const bool swp = (r & 1) != 0;
            for (int tile0 = t >> 7; tile0 < ntile; tile0 += TB * tstep) {
                double2 hv[TB], uv[TB];
                double sr[TB], sc0[TB], sc1[TB], dg[TB];
                int widx[TB];
                bool msk[TB], dia0[TB], dia1[TB];
#pragma unroll
                for (int b = 0; b < TB; b++) {
                    const int tile = min(tile0 + b * tstep, ntile - 1);
                    int ti, tj;
                    tri_decode(tile, ti, tj);
                    const int row = 16 * ti + r, col = 16 * tj + c2;
                    widx[b] = tl_idx(ti, tj, r, c2) & ~1;   // slot of the pair
                    msk[b] = ((colmask >> ti) & 1u) && ((colmask >> tj) & 1u);
                    dia0[b] = row == col; dia1[b] = row == col + 1;
                    hv[b] = row < P ? *(const double2 *)&c.H[(size_t)row * LW + col] : make_double2(0.0, 0.0);   // rows >= P are padding
                    uv[b] = msk[b] ? *(const double2 *)&c.Sc[widx[b]] : make_double2(0.0, 0.0);
                    sr[b] = sp[row]; sc0[b] = sp[col]; sc1[b] = sp[col + 1]; dg[b] = dgp[row];
                }
#pragma unroll
                for (int b = 0; b < TB; b++) {
                    if (tile0 + b * tstep >= ntile) break;
                    // S = Sp (H - U) Sp + mu D^2 (U = 0 outside the tiles the landmark rows touch); unit diagonal for constant parameters
                    const double u0 = swp ? uv[b].y : uv[b].x, u1 = swp ? uv[b].x : uv[b].y;
                    double v0 = sr[b] * sc0[b] * (hv[b].x - u0), v1 = sr[b] * sc1[b] * (hv[b].y - u1);
                    if (dia0[b]) { v0 += mu * dg[b] * dg[b]; if (sr[b] == 0.0) v0 = 1.0; }
                    if (dia1[b]) { v1 += mu * dg[b] * dg[b]; if (sr[b] == 0.0) v1 = 1.0; }
                    *(double2 *)&Stiles[widx[b]] = swp ? make_double2(v1, v0) : make_double2(v0, v1);   // (big: c.Sc is read -- its U part -- and rewritten by the same thread)
                }
            }
            __syncthreads();
        }
        PH(50);
        bool ok = big ? chol_tiles_stream(c.Sc, LW >> 4, work, &sh_i[2], chol_dinv, xs)
                      : chol_tiles(work, LW >> 4, &sh_i[2], chol_dinv, s == 0 ? B.timings + 59 : nullptr, xs);   // with the forward substitution
        PH(51);
        if (st.test_fail > 0) { ok = false; __syncthreads(); if (t == 0) st.test_fail--; }   // test hook: walk the retry ladder
        if (ok) {
            // (one wavefront without workgroup barriers also wins on the HBM-resident tiles of the big windows: 64 -> 33 us at nb = 21)
            chol_backward_tiles_wave(big ? c.Sc : work, LW >> 4, xs, chol_dinv);
            PH(56);
            double bad = 0;
            for (int a = t; a < P; a += nt) if (!isfinite(xs[a])) bad += 1;
            bad = block_sum(bad, sred);
            ok = bad == 0;
        }
        if (!ok) {
            mu *= 10.0;
            if (mu < 1.0) { if (t == 0) st.retry = 1; finish(PS_SCHUR); return; }   // same iteration, Schur complement again at the new mu
            finish(PS_DONE);   // "if (!ok) break"
            return;
        }
        PH(57);
        for (int a = t; a < LW; a += nt) { yp[a] = xs[a]; gnp[a] = -xs[a] * dgp[a]; tmpv[a] = sp[a] * xs[a]; }
        __syncthreads();
        matvec_pass_2range<TB>(c.Hpl, LW, Fa, P, 6 * W1, 15 * W1, ne_ext, nullptr, tmpv, nullptr, tmpl, nullptr);
        PH(58);
        for (int k = t; k < Kpad; k += nt) {
            double y = k < Fa ? (gls[k] - sl[k] * tmpl[k]) * inv[k] : 0.0;
            yl[k] = y;
            gnl[k] = -y * dgl[k];
        }
        __syncthreads();
    }
    PH(52);
    // traditional dogleg in the D-scaled space
    double gnorm = 0, gnn = 0, gdot = 0;
    for (int a = t; a < P; a += nt) { gnorm += gradp[a] * gradp[a]; gnn += gnp[a] * gnp[a]; gdot += gradp[a] * gnp[a]; }
    for (int k = t; k < Fa; k += nt) { gnorm += gradl[k] * gradl[k]; gnn += gnl[k] * gnl[k]; gdot += gradl[k] * gnl[k]; }
    block_sum3(gnorm, gnn, gdot, sred);
    gnorm = sqrt(gnorm);
    gnn = sqrt(gnn);
    double ca = 0, cb = 0;
    if (!(gnn <= radius) && !cauchy_valid) {
        matvec_pass<TB>(c.H, LW, P, P, nullptr, up, nullptr, tmpv, work);
        matvec_pass_2range<TB>(c.Hpl, LW, Fa, P, 6 * W1, 15 * W1, ne_ext, ul, up, tmpv2, tmpl, work);
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
    }
    PH(53);
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
    double n2 = 0, lin = 0, quad = 0;
    for (int a = t; a < LW; a += nt) {
        double v = ca * gradp[a] + cb * gnp[a];
        double stv = a < P ? v / dgp[a] : 0.0;
        stp[a] = stv;
        if (a < P) {
            n2 += v * v;
            lin += stv * gs[a];
            quad += stv * ((ca != 0.0 ? ca * hsgp[a] : 0.0) - cb * (gs[a] - mu * dgp[a] * dgp[a] * yp[a]));
        }
    }
    for (int k = t; k < Kpad; k += nt) {
        double v = k < Fa ? ca * gradl[k] + cb * gnl[k] : 0.0;
        double stv = k < Fa ? v / dgl[k] : 0.0;
        stl[k] = stv;
        if (k < Fa) {
            n2 += v * v;
            lin += stv * gls[k];
            quad += stv * ((ca != 0.0 ? ca * hsgl[k] : 0.0) - cb * (gls[k] - mu * dgl[k] * dgl[k] * yl[k]));
        }
    }
    block_sum3(n2, lin, quad, sred);
    if (dogleg_norm < 0) dogleg_norm = sqrt(n2);
    const double model_change = -(lin + 0.5 * quad);
    if (!(model_change > 0)) {
        if (++invalid >= 5) { finish(PS_DONE); return; }
        mu *= 10.0;
        finish(PS_SCHUR);   // "reuse = false; continue": the next iteration re-forms the Schur complement at the larger mu
        return;
    }
    invalid = 0;
    PH(54);
    // candidate = Plus(x, step .* scale)
    for (int a = t; a < LW; a += nt) delta[a] = a < P ? stp[a] * sp[a] : 0.0;
    __syncthreads();
    ps_form_candidate(c, st, 1.0, delta, stl, sl, sred);
    // a bounds-constrained solve: this candidate is only the first trial of Ceres' projected line search (ps_ls_kernel, next launch)
    if (t == 0) st.ls_pending = st.constrained;
    if (t == 0) { st.model_change = model_change; st.eval_with_J = iter < cfg.max_iterations ? 1 : 0; }
    finish(PS_EVAL_C);
    PH(55);
}

// 8 wavefronts, 256 VGPRs per lane, no scratch (the 1024-thread / 128-VGPR build of rounds 2 - 5 was removed in round 6: it had gone wrong unnoticed)
__global__ __launch_bounds__(512) void ps_serial_kernel_512(Batch B) { ps_serial_body<false, 8>(B); }
// the same serial phase for windows whose Schur complement stays in HBM / L2 (its own kernel: the streaming Cholesky's registers must not
// weigh on the 128-VGPR budget of the resident version); 512 threads = 256 VGPRs per lane
__global__ __launch_bounds__(512) void ps_serial_big_kernel(Batch B) { ps_serial_body<true, 8>(B); }

// Ceres' projected Armijo line search of a bounds-constrained solve (TrustRegionMinimizer::DoLineSearch -> ArmijoLineSearch::DoSearch,
// line_search.cc; Solver::Options defaults: CUBIC interpolation, sufficient decrease 1e-4, contraction in [1e-3, 0.6], at most 20 iterations,
// min step 1e-9).  Its own kernel, grid S x 256 threads, launched behind ps_serial in every slot: one workgroup per sequence, which exits at once
// unless ps_serial has just formed the alpha = 1 candidate of a CONSTRAINED solve (st.ls_pending).  Every trial evaluates cost and slope
// gradient(x_a) . delta at Plus(x, a delta) -- the evaluation roles of ps_eval one after the other, same sums as the multi-block evaluation --
// until f(x_a) <= f(x) + 1e-4 a g^T delta; the step is then shortened to a (the candidate stays in st.Xc / c.cfeat), or, when the search fails
// (20 trials, or a |delta|_inf < 1e-9), left as it was.  The next ps_eval evaluates whatever candidate stands, as Ceres evaluates the candidate
// again after its search; model_change, dogleg_norm and the radius logic keep the FULL step's values.  (Kept out of ps_serial / ps_eval on
// purpose: inlined there its stack objects gave the two hottest kernels of the solve a private segment.)
__global__ __launch_bounds__(256, 2) void ps_ls_kernel(Batch B) {   // (held to 256 VGPRs -- the search itself spills, it is the rare path: an idle workgroup of 400 VGPRs per lane waits for register space on every CU it lands on; 128 measured the same when idle and slower when it runs)
    const int s = blockIdx.x + B.s0, t = threadIdx.x, nt = blockDim.x;
    SolveSt &st = B.sst[s];
    if (st.stage != PS_EVAL_C || !st.ls_pending) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ double sred[64];
    Ctx c = make_ctx(B, s);
    ps_sel_rows(B, c, st.rowbuf);
    const BeSeq &be = *c.be;
    const vio_config &cfg = c.C->c;
    const int W = c.W, W1 = W + 1, P = c.P, Fa = st.Fa, nblk = st.n_eval_blocks;
    constexpr int XD = ((int)(sizeof(Params) / sizeof(double)) + 1) & ~1;
    Params &X = *(Params *)lds;
    double *lsw = (double *)lds + XD, *lpart = lsw + 104;                 // scalar workspace (96) + broadcast slot, partial costs (64)
    // the evaluation roles' staging region (frame-pair geometry / pre-integration headers, 38 KB) lives in HBM / L2 here, not in LDS: an IDLE
    // launch of this kernel -- every slot of every unconstrained solve -- must ask for next to nothing, or its workgroups queue behind the LDS
    // the other stream group's kernels hold (measured with the region in LDS: 9 us per idle launch at S = 128, 30 us at S = 512)
    unsigned char *role_smem = B.ls_scratch + (size_t)s * ps_eval_lds_bytes(c.W);
    const double *stl = c.lvec + 4 * (size_t)c.NLs, *sl = c.lvec;
    const double *delta = c.vec + 8 * (size_t)c.LW;   // the unscaled tangent step ps_serial left in slot 8 of the step vectors
    const bool act = true;
    double g0 = 0, dmax = 0;
    for (int a = t; a < P; a += nt) { const double d = delta[a]; g0 += d * c.vec[a]; dmax = fmax(dmax, fabs(d)); }   // initial_gradient = gradient . delta
    for (int k = t; k < Fa; k += nt) { const double d = stl[k] * sl[k]; g0 += d * c.gl[k]; dmax = fmax(dmax, fabs(d)); }
    g0 = block_sum(g0, sred);
    dmax = block_max(dmax, sred);   // LineSearchFunction::DirectionInfinityNorm
    LsSample lower = {0.0, st.cost, g0, 1}, previous = {0, 0, 0, 0}, current = {0, 0, 0, 0};
    double alpha = 1.0;
    int it = 0, evals = 0;
    for (;;) {
        __syncthreads();
        for (int k = t; k < (int)(sizeof(Params) / sizeof(double)); k += nt) ((double *)&X)[k] = ((const double *)&st.Xc)[k];
        __syncthreads();
        double gd = 0;
        for (int vb = 0; vb < nblk; vb++) {
            double cost = ps_eval_role<true>(B, s, c, st, X, c.cfeat, true, vb, t, act, role_smem, delta, stl, sl, gd);
            cost = block_sum(cost, sred);
            if (t == 0) lpart[vb] = cost;
            __threadfence_block();
            __syncthreads();   // the next role reuses the region; the IMU pass below reads c.imu_raw back
        }
        {
            // the IMU factors' share of the slope: whitened residual . M (J_raw delta), raw Jacobians and residuals from c.imu_raw
            double *vj = (double *)role_smem;   // [W][15]
            for (int q = t; q < 15 * W; q += nt) {
                const int i = q / 15, r = q - 15 * i;
                const PreInt &p = c.pre[be.pre_idx[i + 1]];
                double acc = 0;
                if (cfg.use_imu && !(p.sum_dt > 10.0)) {
                    const double *Jr = c.imu_raw + (size_t)i * 15 * 31 + r * 31;
                    for (int d = 0; d < 6; d++) acc += Jr[d] * delta[6 * i + d] + Jr[15 + d] * delta[6 * (i + 1) + d];
                    for (int d = 0; d < 9; d++) acc += Jr[6 + d] * delta[6 * W1 + 9 * i + d] + Jr[21 + d] * delta[6 * W1 + 9 * (i + 1) + d];
                }
                vj[q] = acc;
            }
            __syncthreads();
            for (int q = t; q < 15 * W; q += nt) {
                const int i = q / 15, r = q - 15 * i;
                const PreInt &p = c.pre[be.pre_idx[i + 1]];
                if (!cfg.use_imu || p.sum_dt > 10.0) continue;
                double mu_r = 0;
                for (int k = 0; k <= r; k++) mu_r += p.sqrt_info[r * 15 + k] * vj[15 * i + k];
                gd += c.imu_raw[(size_t)i * 15 * 31 + r * 31 + 30] * mu_r;
            }
        }
        gd = block_sum(gd, sred);
        double total = (t & 63) < nblk ? lpart[t & 63] : 0.0;   // the sum ps_accept forms from the partial costs, on every wavefront
        total = wave_sum_dpp(total);
        evals++;
        current.x = alpha; current.value = total; current.gradient = gd; current.valid = (isfinite(total) && isfinite(gd)) ? 1 : 0;
        // one turn of ArmijoLineSearch::DoSearch's loop (uniform over the workgroup: every thread holds the same samples)
        if (current.valid && !(current.value > st.cost + 1e-4 * g0 * current.x)) break;   // sufficient decrease: delta *= alpha, the candidate stands
        bool fail = ++it >= 20;
        double a_next = 1.0;
        if (!fail) {
            __syncthreads();
            if (t == 0) lsw[96] = ls_next_step(lower, previous, current, 1e-3 * current.x, 0.6 * current.x, lsw);   // run-time indexed arrays in LDS
            __syncthreads();
            a_next = lsw[96];
            if (a_next * dmax < 1e-9) fail = true;
        }
        if (fail) {
            // "Line search failed": the step goes to the trust-region test unshortened
            if (alpha != 1.0) { __syncthreads(); ps_form_candidate(c, st, 1.0, delta, stl, sl, sred); }
            break;
        }
        previous = current;
        alpha = a_next;
        __syncthreads();
        ps_form_candidate(c, st, alpha, delta, stl, sl, sred);
        __threadfence_block();
    }
    if (t == 0) { st.ls_pending = 0; c.be->ls_evals += evals; c.be->ls_contractions += evals - 1; }
}

// ---------------------------------------------------------------------------------------------------------------- FINAL
__global__ __launch_bounds__(256) void ps_final_kernel(Batch B) {
    const int s = blockIdx.x + B.s0, t = threadIdx.x;
    SolveSt &st = B.sst[s];
    if (st.stage == PS_IDLE) return;
    Ctx c = make_ctx(B, s);
    __shared__ Params X;
    __shared__ double sdx[16], sh_d[8];
    __shared__ int sh_i[8];
    for (int k = t; k < (int)(sizeof(Params) / sizeof(double)); k += blockDim.x) ((double *)&X)[k] = ((const double *)&st.X)[k];
    // the host enqueues max_iterations + 2 slots; more than one Cholesky retry / invalid step in one solve can use them up before the
    // trust-region loop has finished: the last accepted point is written back (a valid, less converged estimate) and the frame is flagged
    if (t == 0 && st.stage != PS_DONE) c.be->overflow |= 32;
    __syncthreads();
    solve_epilogue(c, X, st.cost, st.iters_done, st.succ, st.ts0, sdx, sh_d, sh_i);
    if (t == 0) st.stage = PS_IDLE;
}
