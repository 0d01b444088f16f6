// Host side of the C ABI (include/vio_abi.h): owns the HBM state of a batch and launches the kernel chain.
// There is no CPU fallback: every entry point fails with VIO_EDEVICE when HIP is unavailable.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <chrono>
#include <string>
#include <string.h>
#include <vector>
#include "kernels.h"
#include "dyninit_host.h"

thread_local std::string g_err;   // last failure of the calling thread (vio_last_error); also set by pg_kernels.hip / posegraph_host.cpp

namespace {

#define HIPCHK(x)                                                                                        \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return VIO_EDEVICE; } \
    } while (0)

void circle_halfwidths(int radius, int *hw) {  // cv::circle(filled), OpenCV drawing.cpp Circle()
    for (int i = 0; i <= radius; i++) hw[i] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > hw[dy]) hw[dy] = dx;
        if (dy > hw[dx]) hw[dx] = dy;
        dy++;
        err += plus;
        plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

}  // namespace

struct vio_batch {
    DevCfg hc;  // host copy
    Batch B;
    int S;
    int device = -1;   // the HIP device this handle's memory, streams and events live on (vio_create: the caller's current device; vio_create_on_device:
                       // the one asked for).  Every entry point binds the calling thread to it for the duration of the call (DevGuard below).
    // Sequences are split into groups of contiguous sequences; every group has its own pair of streams, so the chain
    // track -> ingest -> solve -> marginalise of one group never waits for the slowest sequence of another group.
    struct Group {
        int s0 = 0, n = 0;
        hipStream_t stream = nullptr;     // back-end (and uploads that feed it)
        hipStream_t fe_stream = nullptr;  // front-end: frame k+1 tracks while frame k is still being marginalised
        hipEvent_t ev_solve = nullptr, ev_fe = nullptr, ev_be = nullptr, ev_ingest = nullptr;
        hipStream_t copy_stream = nullptr;   // host -> HBM uploads of vio_feed (on_device == 0), beside the kernels of the previous frame
        hipStream_t copy_stream2 = nullptr;  // the depth images on a stream of their own: two DMA engines per group (one stream moves ~25 GB/s from page-locked memory)
        hipGraphExec_t solve_graph = nullptr;   // VIO_GRAPH: setup + iteration slots + final of this group as one graph launch
        uint64_t solve_graph_key = 0;           // hash of the arguments the capture baked in (Batch by value + launch knobs)
        // vio_feed uploads from host buffers: one event pair per staging buffer (g.flip), so that TWO uploads may be in flight -- the call that
        // reuses a staging buffer waits for the upload of two calls ago, not for the previous one (round 5; vio_host_buffers_done)
        hipEvent_t ev_up_gray[2] = {nullptr, nullptr}, ev_up_depth[2] = {nullptr, nullptr};
        bool up_used[2] = {false, false};
        // the staging images of vio_feed are double-buffered: frame n uploads into buffer n & 1 while frame n-1's kernels still read the
        // other one, so an upload only waits for the readers of frame n-2 (ev_rd_gray / ev_rd_depth of its buffer)
        int flip = 0;
        hipEvent_t ev_rd_gray[2] = {nullptr, nullptr}, ev_rd_depth[2] = {nullptr, nullptr};
        bool have_rd_gray[2] = {false, false}, have_rd_depth[2] = {false, false};
        // host -> HBM uploads enqueued on fe_stream / stream by the non-overlap entry points (vio_track, vio_process, vio_process_obs*):
        // asynchronous when the caller's buffers are page-locked, so the next call that takes host buffers waits for them first
        hipEvent_t ev_host_fe = nullptr, ev_host_be = nullptr;
        bool host_fe_pending = false, host_be_pending = false;
        bool have_solve_ev = false, have_ingest_ev = false;
        // vio_feed: stamps / frame modes of a call bounce through a library-owned page-locked ring, so the caller's arrays are free when the
        // call returns although the copies only run when fe_stream gets to them (tracker lag 1 holds that stream behind be_ingest)
        static constexpr int kSideRing = 16;
        unsigned char *side_ring = nullptr;          // [kSideRing][n * 9]: n doubles then n mode bytes
        hipEvent_t side_ev[kSideRing] = {};
        bool side_used[kSideRing] = {};
        int side_pos = 0;
    };
    std::vector<Group> groups;
    int tracker_lag = 0;              // vio_set_tracker_lag
    int extra_slots = 2;              // VIO_EXTRA_SLOTS: iteration slots beyond max_iterations (1 carries the last evaluation, the second absorbs one Cholesky retry / invalid step)
    int xcd_n = 0;                    // VIO_XCD_N: override of the XCD count the map assumes (0: 8)
    bool use_graph = false;           // VIO_GRAPH: replay the solve chain of a group as a hipGraph (launch_backend)
    int xcd_map = 1;                  // VIO_XCD_MAP: XCD-aware block map of the multi-block ps_* kernels (be_phased.h ps_blk)
    int fe_xcd_map = 1;               // VIO_FE_XCD_MAP: the same idea for fe_lk (needs the front-end on every XCD: off under a CU partition)
    bool fe_partitioned = false;      // the front-end streams carry a CU mask (VIO_FE_CUS > 0 with tracker lag 1)
    int ps_asm_b_blocks = 24;         // workgroups per sequence that sum the entries of H (VIO_ASM_B_BLOCKS)
    int asm_b_by_blocks = 2;          // VIO_ASM_B_MODE: 2 (default since round 6) = one thread per entry a >= b of H, mirror image stored too (same bits as 0, +5 % frames/s); 0 = one thread per entry of H; 1 = H summed by pairs of parameter blocks (round 5: same bits, 1 - 2 % slower)
    int serial_threads = 512;         // ps_serial block size (VIO_SERIAL_THREADS: 512 or 1024).  Round 3: equal speed (36.2 k vs 36.4 k frames/s); the 512-thread
                                      // build has 256 VGPRs per lane and no scratch, the 1024-thread one spills 21 registers since the matrix-core diagonal block
    hipStream_t stream = nullptr;     // = groups[0].stream (returned by vio_get_stream; IMU scatter runs here)
    hipStream_t fe_stream = nullptr;  // = groups[0].fe_stream
    hipEvent_t ev[4];
    std::vector<void *> allocs;
    uint8_t *d_fisheye = nullptr;                                    // vio_set_fisheye_mask
    uint8_t *d_gray_stage = nullptr, *d_gray_stage1 = nullptr;       // [S][H][W] staging images of host-buffer calls; the second one only for vio_feed
    uint16_t *d_depth_stage = nullptr, *d_depth_stage1 = nullptr;
    double *d_stamps = nullptr;
    uint8_t *d_modes = nullptr;       // [S] frame modes of the current vio_feed_modes / vio_track_ex call
    double *d_rrel = nullptr;         // [S][9] caller-supplied relative rotations (vio_track_ex)
    // caller-supplied feature maps (vio_process_obs): [S] counts / stamps, [S][NP] ids, [S][NP][7] observations
    int *d_in_n = nullptr, *d_in_ids = nullptr;
    double *d_in_obs = nullptr, *d_in_stamps = nullptr;
    double *d_r9 = nullptr;           // vio_predict_motion result
    // ---- dynamic initialisation (static_init: 0): host mirror of Estimator::all_image_frame per sequence while it is INITIAL
    struct DynSeq {
        std::vector<vinit::ImageFrame> frames;
        double initial_timestamp = 0;
        bool nonlinear = false;       // host view of solver_flag (refreshed from h_state)
        int attempts = 0, failures = 0, last_stage = 0;
    };
    std::vector<DynSeq> dyn;          // [S], only used when cfg.dynamic_init
    bool dyn_active = false;          // some sequence is still INITIAL: the back-end runs in two halves with the host in between
    int *h_state = nullptr;           // pinned [S]: solver_flag of every sequence after the last be_solve (async copy per frame)
    int *d_state = nullptr;
    hipEvent_t ev_state = nullptr;
    bool state_pending = false;
    double *d_dyn_samples = nullptr;  // IMU steps of the window slots handed to be_dyn_finalize_kernel (grown on demand)
    int *d_dyn_offs = nullptr;
    size_t dyn_samples_cap = 0;
    // pending IMU samples (host staging)
    std::mutex imu_mu;
    std::vector<int> p_seq;
    std::vector<double> p_t, p_acc, p_gyr;
    // IMU upload: two pinned host staging sets + device sets used alternately, each guarded by an event, so that vio_push_imu
    // between frames never forces a device-wide synchronisation (the scatter kernel is ordered on the streams instead)
    struct ImuStage {
        int *h_seq = nullptr, *d_seq = nullptr;
        double *h_t = nullptr, *h_acc = nullptr, *h_gyr = nullptr, *d_t = nullptr, *d_acc = nullptr, *d_gyr = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
    } imu_stage[2];
    int imu_stage_cur = 0;
    hipEvent_t ev_imu = nullptr;
    std::vector<double> last_imu_t;
    size_t lds_select = 0, lds_add = 0, lds_fast = 0, lds_solve = 0, lds_serial = 0, lds_marg = 0, lds_factor = 0, lds_ps_ls = 0, lds_ps_evalf = 0;
    int ps_evalf_blocks = 0;           // workgroups per sequence of ps_evalf_kernel (2 + B.fuse)
    bool feed_throttle = true;         // VIO_FEED_THROTTLE: host-fed vio_feed waits for the back-end of two feeds ago before it enqueues (stage_inputs)
    int uploads_in_flight = 2;         // VIO_UPLOADS_IN_FLIGHT: page-locked image uploads of vio_feed that may be pending when a call returns
    int relo_frames = 0;               // frames for which the two-kernel solver path is launched beside the fused kernel (armed by vio_set_relo_frame)
    bool line_search = true;           // ps_ls_kernel behind every ps_serial (Ceres' projected line search on bounds-constrained solves)
    // VIO_BE_THREADS / VIO_MARG_THREADS, read at vio_create.  The marginalisation kernel runs next to the following frame's front-end:
    // with 6 instead of 8 wavefronts (256 VGPRs each) two SIMDs per CU keep half of their register file free and the LK wavefronts can
    // co-reside (be_marg 1.4 -> 1.6 ms, fe_lk 0.77 -> 0.60 ms; the front-end is the longer of the two, so the step gets shorter).
    int be_threads = 512, marg_threads = 384;
    // VIO_SOLVE_MODE: 0 = persistent kernel (one workgroup per sequence for the whole solve), 1 = phased solver (be_phased.h, default)
    int solve_mode = 1;
    bool asm_a_occ4 = false;          // VIO_ASM_A_OCC=4
    int eval_occ = 0;                 // VIO_EVAL_OCC=3|4
    bool serial_big = false;          // the window's Schur complement does not fit LDS: ps_serial_big_kernel (HBM-resident tiles, streaming Cholesky)
    size_t lds_ps_eval = 0;
    int ps_eval_blocks = 0, ps_asm_a_blocks = 0, ps_schur_tiles = 0;
    bool timing_valid = false;
    // per-kernel event pool (vio_profile_begin / vio_profile_end)
    std::vector<hipEvent_t> pev;
    int prof_steps = 0, prof_cur = -1;
    bool prof_fe_only = false;        // the profiled steps were vio_track calls: only the front-end events exist
};
#define VIO_NK 10  // kernels per vio_feed: fe_begin pyrdown predict lk select fast add | be_ingest solve marg(+finish)
#define VIO_NEV 12 // events per step: 0..7 bracket the front-end kernels on fe_stream, 8..11 the back-end kernels on stream
#define PEV(h, k) do { if (g.s0 == 0 && (h)->prof_cur >= 0 && (h)->prof_cur < (h)->prof_steps) (void)hipEventRecord((h)->pev[(size_t)(h)->prof_cur * VIO_NEV + (k)], (k) <= 7 ? g.fe_stream : g.stream); } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel, not of a handle: with several handles of different
// configurations alive, keep the largest value ever requested (monotonic), otherwise the handle created last would shrink the
// limit under the others.
static int raise_lds_limit(const void *fn, size_t bytes) {
    // the attribute belongs to the function ON THE CURRENT DEVICE (a handle per GPU in one process sets it once per device)
    struct Seen { const void *fn; int dev; size_t bytes; };
    static std::mutex mu;
    static std::vector<Seen> seen;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    for (auto &e : seen)
        if (e.fn == fn && e.dev == dev) {
            if (bytes <= e.bytes) return 0;
            e.bytes = bytes;
            return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
        }
    seen.push_back({fn, dev, bytes});
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}

// static + dynamic LDS of a kernel this handle will launch against what a workgroup may own: a configuration that does not fit fails at
// vio_create, not with an aborted launch in the middle of a frame
static bool lds_fits(const void *fn, size_t dynamic_bytes, const char *name) {
    hipFuncAttributes a;
    int dev = 0, cap = 0;
    if (hipFuncGetAttributes(&a, fn) != hipSuccess) return true;   // (cannot tell: let the launch decide)
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || cap <= 0) cap = 160 * 1024;
    if (a.sharedSizeBytes + dynamic_bytes <= (size_t)cap) return true;
    g_err = std::string("configuration needs more LDS than a workgroup may own: ") + name;
    return false;
}

// dynamic LDS of be_marg_exact_kernel's LDS-resident eigen-decomposition of an m x m block (be_kernels.hip marg_exact_finish)
static size_t marg_exact_lds_bytes(int m) { return ((size_t)m * (m | 1) + 11 * (size_t)(6 * VIO_MAXW + 16)) * 8 + 64; }

// A handle owns its device: entry points may be called from any host thread with any device current (one host thread per GPU
// in one process, or a caller that moves between devices); the guard makes h->device current for the call and restores the caller's on return.
struct DevGuard {
    int prev = -1;
    bool switched = false;
    bool failed = false;   // the handle's device could not be made current: the entry point must not run on the caller's device instead
    explicit DevGuard(const vio_batch *h) {
        if (!h || h->device < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { failed = true; return; }
        if (prev != h->device) { switched = hipSetDevice(h->device) == hipSuccess; failed = !switched; }
    }
    ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard &) = delete;
    DevGuard &operator=(const DevGuard &) = delete;
};
static int sync_all(vio_batch *h) {
    for (auto &g : h->groups) {
        if (g.copy_stream) HIPCHK(hipStreamSynchronize(g.copy_stream));
        if (g.copy_stream2) HIPCHK(hipStreamSynchronize(g.copy_stream2));
        HIPCHK(hipStreamSynchronize(g.fe_stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        g.host_fe_pending = g.host_be_pending = false;
    }
    return VIO_OK;
}

// Pending IMU samples arrive grouped by sequence (counting sort on the host keeps the push order inside a sequence): sample i of
// sequence s = seq_of[i] is the (i - off[s])-th new sample of that sequence.  One thread per sample; the ring counters are advanced
// by imu_commit_kernel afterwards (same stream), so every thread of the scatter sees the old count.
__global__ void imu_scatter_kernel(Batch B, int total, const int *seq_of, const double *t, const double *acc, const double *gyr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const DevCfg &C = *B.cfg;
    const int *off = seq_of + total;  // [S + 1] offsets of each sequence's run, stored behind the sequence ids
    const int s = seq_of[i];
    const int slot = (B.be[s].imu_count + (i - off[s])) % C.NIMU;
    B.imu_t[(size_t)s * C.NIMU + slot] = t[i];
    for (int k = 0; k < 3; k++) {
        B.imu_acc[((size_t)s * C.NIMU + slot) * 3 + k] = acc[3 * i + k];
        B.imu_gyr[((size_t)s * C.NIMU + slot) * 3 + k] = gyr[3 * i + k];
    }
}
__global__ void imu_commit_kernel(Batch B, int total, const int *seq_of) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B.S) return;
    const int *off = seq_of + total;
    B.be[s].imu_count += off[s + 1] - off[s];
}

namespace {

template <class T> int dalloc(vio_batch *h, T **p, size_t n) {
    void *q = nullptr;
    size_t bytes = n * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIPCHK(hipMalloc(&q, bytes));
    HIPCHK(hipMemset(q, 0, bytes));
    h->allocs.push_back(q);
    *p = (T *)q;
    return VIO_OK;
}

// (re)initialise the state of sequences [s_lo, s_hi).  what & VIO_RESET_ESTIMATOR: Estimator::clearState() + setParameter()
// (estimator.cpp:15-116: window, landmarks, prior, pre-integrations, pending IMU) and the nodelet's first_image_flag / last_image_time
// (estimator_nodelet.cpp:246-247); what & VIO_RESET_TRACKER: a fresh FeatureTracker (points, ids, images) and init_pub / init_feature.
// The stream-discontinuity branch of the nodelet (estimator_nodelet.cpp:243-262) resets ONLY the estimator side: trackerData keeps its
// points, ids and previous image, init_pub / init_feature keep their values.
enum { VIO_RESET_ESTIMATOR = 1, VIO_RESET_TRACKER = 2 };
int init_state(vio_batch *h, int s_lo, int s_hi, int what = VIO_RESET_ESTIMATOR | VIO_RESET_TRACKER) {
    const DevCfg &C = h->hc;
    int n = s_hi - s_lo, W = C.W;
    std::vector<FeSeq> fe(n);
    if (what & VIO_RESET_TRACKER) {
        memset(fe.data(), 0, sizeof(FeSeq) * n);
        for (int s = 0; s < n; s++) {
            FeSeq &f = fe[s];
            f.first_image_flag = 1;
            for (int k = 0; k < C.ncells; k++) f.grids_texture_status[k] = 1;
            f.R_rel[0] = f.R_rel[4] = f.R_rel[8] = 1;
        }
    } else {
        HIPCHK(hipMemcpy(fe.data(), h->B.fe + s_lo, sizeof(FeSeq) * n, hipMemcpyDeviceToHost));
        for (int s = 0; s < n; s++) { fe[s].first_image_flag = 1; fe[s].last_image_time = 0; fe[s].n_obs = 0; fe[s].publish_ok = 0; }
    }
    HIPCHK(hipMemcpy(h->B.fe + s_lo, fe.data(), sizeof(FeSeq) * n, hipMemcpyHostToDevice));
    if (!(what & VIO_RESET_ESTIMATOR)) return VIO_OK;
    std::vector<BeSeq> be(n);
    memset(be.data(), 0, sizeof(BeSeq) * n);
    for (int s = 0; s < n; s++) {
        BeSeq &b = be[s];
        for (int i = 0; i <= VIO_MAXW; i++) { b.Rs[i][0] = b.Rs[i][4] = b.Rs[i][8] = 1; b.pre_idx[i] = i; }
        for (int k = 0; k < 9; k++) b.ric[k] = C.c.ric[k];
        for (int k = 0; k < 3; k++) b.tic[k] = C.c.tic[k];
        b.td = C.c.td;
        b.track_td = C.c.td;
        b.g[2] = C.c.g_norm;
        b.prevTime = -1;
        b.n_free = C.NL;
    }
    if (what != (VIO_RESET_ESTIMATOR | VIO_RESET_TRACKER)) {
        // diagnostics that count over the life of the handle survive an estimator restart
        std::vector<BeSeq> old(n);
        HIPCHK(hipMemcpy(old.data(), h->B.be + s_lo, sizeof(BeSeq) * n, hipMemcpyDeviceToHost));
        for (int s = 0; s < n; s++) { be[s].iter_total = old[s].iter_total; be[s].solve_total = old[s].solve_total; be[s].reboot_count = old[s].reboot_count; }
    }
    HIPCHK(hipMemcpy(h->B.be + s_lo, be.data(), sizeof(BeSeq) * n, hipMemcpyHostToDevice));
    std::vector<int> fr((size_t)n * C.NL);
    for (int s = 0; s < n; s++) for (int k = 0; k < C.NL; k++) fr[(size_t)s * C.NL + k] = C.NL - 1 - k;
    HIPCHK(hipMemcpy(h->B.lm_free + (size_t)s_lo * C.NL, fr.data(), fr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(h->B.lm_order + (size_t)s_lo * C.NL, 0, sizeof(int) * (size_t)n * C.NL));
    HIPCHK(hipMemset(h->B.pre + (size_t)s_lo * (W + 2), 0, sizeof(PreInt) * (size_t)n * (W + 2)));
    HIPCHK(hipMemset(h->B.odom + (size_t)s_lo * 11, 0, sizeof(double) * (size_t)n * 11));
    if (what & VIO_RESET_TRACKER) HIPCHK(hipMemset(h->B.odom_count + s_lo, 0, sizeof(int) * (size_t)n));   // (an estimator restart keeps the CSV rows written so far)
    {
        if (!h->dyn.empty()) {
            for (int s = s_lo; s < s_hi; s++) h->dyn[s] = vio_batch::DynSeq();
            h->dyn_active = true;
            if (s_lo == 0 && s_hi == h->S) h->state_pending = false;   // a partial reset must not swallow another sequence's pending reboot notice
        }
        // Estimator::clearState() empties imu_buf: drop what is still staged on the host for these sequences
        std::lock_guard<std::mutex> lk(h->imu_mu);
        for (int s = s_lo; s < s_hi; s++) h->last_imu_t[s] = -1e300;
        size_t w = 0;
        for (size_t i = 0; i < h->p_seq.size(); i++) {
            if (h->p_seq[i] >= s_lo && h->p_seq[i] < s_hi) continue;
            h->p_seq[w] = h->p_seq[i]; h->p_t[w] = h->p_t[i];
            for (int k = 0; k < 3; k++) { h->p_acc[3 * w + k] = h->p_acc[3 * i + k]; h->p_gyr[3 * w + k] = h->p_gyr[3 * i + k]; }
            w++;
        }
        h->p_seq.resize(w); h->p_t.resize(w); h->p_acc.resize(3 * w); h->p_gyr.resize(3 * w);
    }
    return VIO_OK;
}

// Moves the samples accepted by vio_push_imu into the per-sequence HBM rings.  The scatter kernel runs on `st`; the caller has
// already ordered `st` after the last reader of the rings (be_ingest of the previous frame, via ev_solve / stream order) and orders
// the next readers (fe_begin, be_ingest) after it.  No host-device synchronisation unless a staging set is still in flight.
int flush_imu(vio_batch *h, hipStream_t st, bool *launched) {
    if (launched) *launched = false;
    std::lock_guard<std::mutex> lk(h->imu_mu);
    size_t n = h->p_seq.size();
    if (n == 0) return VIO_OK;
    vio_batch::ImuStage &sg = h->imu_stage[h->imu_stage_cur];
    h->imu_stage_cur ^= 1;
    if (sg.busy) { HIPCHK(hipEventSynchronize(sg.done)); sg.busy = false; }
    if (n > sg.cap) {
        size_t cap = n * 2 + 1024;
        if (sg.h_seq) {
            (void)hipHostFree(sg.h_seq); (void)hipHostFree(sg.h_t); (void)hipHostFree(sg.h_acc); (void)hipHostFree(sg.h_gyr);
            (void)hipFree(sg.d_seq); (void)hipFree(sg.d_t); (void)hipFree(sg.d_acc); (void)hipFree(sg.d_gyr);
        }
        HIPCHK(hipHostMalloc((void **)&sg.h_seq, (cap + h->S + 1) * sizeof(int), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&sg.h_t, cap * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&sg.h_acc, cap * 3 * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&sg.h_gyr, cap * 3 * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&sg.d_seq, (cap + h->S + 1) * sizeof(int)));
        HIPCHK(hipMalloc((void **)&sg.d_t, cap * sizeof(double)));
        HIPCHK(hipMalloc((void **)&sg.d_acc, cap * 3 * sizeof(double)));
        HIPCHK(hipMalloc((void **)&sg.d_gyr, cap * 3 * sizeof(double)));
        sg.cap = cap;
        if (!sg.done) HIPCHK(hipEventCreateWithFlags(&sg.done, hipEventDisableTiming));
    }
    {
        // stable counting sort by sequence into the pinned staging set; the S + 1 run offsets travel behind the sequence ids
        const int S = h->S;
        int *off = sg.h_seq + n;
        for (int s = 0; s <= S; s++) off[s] = 0;
        for (size_t i = 0; i < n; i++) off[h->p_seq[i] + 1]++;
        for (int s = 0; s < S; s++) off[s + 1] += off[s];
        std::vector<int> cur(off, off + S);
        for (size_t i = 0; i < n; i++) {
            const int s = h->p_seq[i], q = cur[s]++;
            sg.h_seq[q] = s;
            sg.h_t[q] = h->p_t[i];
            for (int k = 0; k < 3; k++) { sg.h_acc[3 * q + k] = h->p_acc[3 * i + k]; sg.h_gyr[3 * q + k] = h->p_gyr[3 * i + k]; }
        }
    }
    h->p_seq.clear(); h->p_t.clear(); h->p_acc.clear(); h->p_gyr.clear();
    HIPCHK(hipMemcpyAsync(sg.d_seq, sg.h_seq, (n + h->S + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(sg.d_t, sg.h_t, n * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(sg.d_acc, sg.h_acc, n * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(sg.d_gyr, sg.h_gyr, n * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    imu_scatter_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(h->B, (int)n, sg.d_seq, sg.d_t, sg.d_acc, sg.d_gyr);
    imu_commit_kernel<<<(h->S + 63) / 64, 64, 0, st>>>(h->B, (int)n, sg.d_seq);
    HIPCHK(hipEventRecord(sg.done, st));
    sg.busy = true;
    if (launched) *launched = true;
    return VIO_OK;
}

// vio_feed / vio_track: pending IMU goes in on group 0's front-end stream after every group's previous solve (its be_ingest, the
// last reader of the rings, precedes the solve), and every group's front-end waits for it.
int flush_imu_frontend(vio_batch *h) {
    {
        std::lock_guard<std::mutex> lk(h->imu_mu);
        if (h->p_seq.empty()) return VIO_OK;
    }
    hipStream_t st = h->groups[0].fe_stream;
    for (auto &g : h->groups) {
        if (h->tracker_lag) { if (g.have_ingest_ev) HIPCHK(hipStreamWaitEvent(st, g.ev_ingest, 0)); }   // be_ingest is the last reader of the rings
        else if (g.have_solve_ev) HIPCHK(hipStreamWaitEvent(st, g.ev_solve, 0));
        HIPCHK(hipStreamWaitEvent(st, g.ev_be, 0));
    }
    bool launched = false;
    int rc = flush_imu(h, st, &launched);
    if (rc != VIO_OK || !launched) return rc;
    if (h->groups.size() > 1) {
        HIPCHK(hipEventRecord(h->ev_imu, st));
        for (size_t k = 1; k < h->groups.size(); k++) HIPCHK(hipStreamWaitEvent(h->groups[k].fe_stream, h->ev_imu, 0));
    }
    return VIO_OK;
}

// vio_process (IMU that arrived between vio_track and vio_process): on group 0's back-end stream after every group's front-end
int flush_imu_backend(vio_batch *h) {
    {
        std::lock_guard<std::mutex> lk(h->imu_mu);
        if (h->p_seq.empty()) return VIO_OK;
    }
    hipStream_t st = h->groups[0].stream;
    for (auto &g : h->groups) HIPCHK(hipStreamWaitEvent(st, g.ev_fe, 0));
    bool launched = false;
    int rc = flush_imu(h, st, &launched);
    if (rc != VIO_OK || !launched) return rc;
    if (h->groups.size() > 1) {
        HIPCHK(hipEventRecord(h->ev_imu, st));
        for (size_t k = 1; k < h->groups.size(); k++) HIPCHK(hipStreamWaitEvent(h->groups[k].stream, h->ev_imu, 0));
    }
    return VIO_OK;
}

int launch_frontend(vio_batch *h, vio_batch::Group &g, const uint8_t *d_gray, int publish, int gate, const uint8_t *d_modes = nullptr,
                    const double *d_rrel = nullptr) {
    const DevCfg &C = h->hc;
    const int S = g.n, Wd = C.c.width, Ht = C.c.height;
    hipStream_t st = g.fe_stream;
    Batch Bg = h->B;
    Bg.s0 = g.s0;
    PEV(h, 0);
    fe_begin_kernel<<<S, 64, 0, st>>>(Bg, h->d_stamps, gate, publish, d_modes, d_rrel);
    PEV(h, 1);
    if (C.c.equalize) {   // EQUALIZE (feature_tracker.cpp:269-275): the tracker sees the CLAHE image
        const size_t HW = (size_t)Wd * Ht;
        fe_clahe_lut_kernel<<<dim3(64, S), 256, 0, st>>>(Bg, d_gray, HW);
        fe_clahe_apply_kernel<<<dim3((Wd + 1023) / 1024, Ht, S), 256, 0, st>>>(Bg, d_gray, HW);
        d_gray = h->B.clahe_img;
    }
    // pyramid: level 1 from the new frame (+ level-0 copy), further levels from the previous one
    {
        int sw = Wd, sh = Ht;
        for (int l = 1; l <= C.c.lk_max_level; l++) {
            int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
            dim3 grid((dw + 63) / 64, (dh + 15) / 16, S);
            fe_pyrdown_kernel<<<grid, 256, 0, st>>>(Bg, l == 1 ? d_gray : nullptr, (size_t)Wd * Ht, sw, sh, l, l == 1 ? 1 : 0);
            sw = dw; sh = dh;
        }
    }
    PEV(h, 2);
    fe_predict_kernel<<<dim3((C.NP + 255) / 256, S), 256, 0, st>>>(Bg);
    PEV(h, 3);
    {
        const int nblk = std::min(C.NP, C.c.max_cnt + C.c.max_cnt / 2 + 32);   // ~1.5 x max_cnt blocks per sequence, strided over n_pts
        if (h->fe_xcd_map && !h->fe_partitioned && S >= 8) {   // VIO_FE_XCD_MAP: all feature blocks of a sequence on one XCD (fe_lk_kernel)
            Batch Bl = Bg;
            Bl.ns = S; Bl.xcd_nb = nblk;
            fe_lk_kernel<<<dim3((unsigned)(8 * ((S + 7) / 8) * nblk)), 64, 0, st>>>(Bl);
        } else {
            Batch Bl = Bg;
            Bl.xcd_nb = 0;
            fe_lk_kernel<<<dim3(nblk, S), 64, 0, st>>>(Bl);
        }
    }
    PEV(h, 4);
    fe_select_kernel<<<S, 256, h->lds_select, st>>>(Bg);
    PEV(h, 5);
    if (publish || d_modes) fe_fast_kernel<<<dim3(C.ncells, S), 256, h->lds_fast, st>>>(Bg);
    PEV(h, 6);
    fe_add_kernel<<<S, 256, h->lds_add, st>>>(Bg, gate);
    PEV(h, 7);
    HIPCHK(hipGetLastError());
    return VIO_OK;
}
__global__ void state_gather_kernel(Batch B, int *state) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < B.S) state[s] = B.be[s].solver_flag;
}

// Host half of the dynamic initialisation for sequence s, called with the group's stream drained right after be_ingest: mirrors
// Estimator::all_image_frame (estimator.cpp:203-206), runs initialStructure when the window is full (:230-259) and hands the result
// to the device.  Returns through `finalize` whether be_dyn_finalize_kernel has to run for the sequence.
int dynamic_init_step(vio_batch *h, int s, const IngestSrc &src, bool *finalize) {
    *finalize = false;
    const DevCfg &C = h->hc;
    const int W = C.W, W1 = W + 1, NL = C.NL, NP = C.NP;
    vio_batch::DynSeq &D = h->dyn[s];
    static thread_local BeSeq be;
    HIPCHK(hipMemcpy(&be, h->B.be + s, sizeof(BeSeq), hipMemcpyDeviceToHost));
    if (be.solver_flag != 0) { D.nonlinear = true; return VIO_OK; }
    if (D.nonlinear) { D = vio_batch::DynSeq(); h->dyn_active = true; }   // rebooted: clearState() dropped all_image_frame
    if (!be.processed) return VIO_OK;
    const int fc = be.frame_count;
    // ---- the image frame just ingested: feature points + the pre-integration of its interval (tmp_pre_integration)
    {
        vinit::ImageFrame f;
        f.stamp = be.cur_stamp;
        int n = 0;
        const int *d_ids;
        const double *d_obs;
        if (src.ids) {
            HIPCHK(hipMemcpy(&n, src.n_obs + s, sizeof(int), hipMemcpyDeviceToHost));
            d_ids = src.ids + (size_t)s * src.cap; d_obs = src.obs + (size_t)s * src.cap * 7;
        } else {
            static thread_local FeSeq fe;
            HIPCHK(hipMemcpy(&fe, h->B.fe + s, sizeof(FeSeq), hipMemcpyDeviceToHost));
            n = fe.n_obs;
            d_ids = h->B.obs_id + (size_t)s * NP; d_obs = h->B.obs + (size_t)s * NP * 7;
        }
        n = std::max(0, std::min(n, NP));
        std::vector<double> o((size_t)n * 7);
        f.ids.resize(n);
        if (n > 0) {
            HIPCHK(hipMemcpy(f.ids.data(), d_ids, sizeof(int) * n, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(o.data(), d_obs, sizeof(double) * 7 * n, hipMemcpyDeviceToHost));
        }
        f.xy.resize((size_t)2 * n);
        for (int k = 0; k < n; k++) { f.xy[2 * k] = o[7 * k]; f.xy[2 * k + 1] = o[7 * k + 1]; }
        static thread_local PreInt p;
        HIPCHK(hipMemcpy(&p, h->B.pre + (size_t)s * (W + 2) + be.pre_idx[fc], sizeof(PreInt), hipMemcpyDeviceToHost));
        if (p.valid && fc != 0) {
            for (int k = 0; k < 3; k++) { f.lin_acc[k] = p.lin_acc[k]; f.lin_gyr[k] = p.lin_gyr[k]; }
            f.dt.assign(p.dt_buf, p.dt_buf + p.n_buf);
            f.acc.resize((size_t)3 * p.n_buf); f.gyr.resize((size_t)3 * p.n_buf);
            for (int k = 0; k < p.n_buf; k++) for (int q = 0; q < 3; q++) { f.acc[3 * k + q] = p.acc_buf[k][q]; f.gyr[3 * k + q] = p.gyr_buf[k][q]; }
        }
        for (int k = 0; k < 3; k++) f.bg_lin[k] = be.Bgs[fc][k];
        D.frames.push_back(std::move(f));
    }
    if (fc < W) return VIO_OK;
    // ---- window full: attempt (estimator.cpp:232-240), at most every 0.1 s
    bool changed = false;
    vinit::Result res;
    if (be.cur_stamp - D.initial_timestamp > 0.1) {
        D.attempts++;
        std::vector<int> order(NL), id(NL), st(NL), no(NL);
        const size_t o = (size_t)s * NL;
        HIPCHK(hipMemcpy(order.data(), h->B.lm_order + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(id.data(), h->B.lm_id + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(st.data(), h->B.lm_start + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(no.data(), h->B.lm_nobs + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
        std::vector<double> obs((size_t)NL * W1 * VIO_OBS_D);
        HIPCHK(hipMemcpy(obs.data(), h->B.lm_obs + o * W1 * VIO_OBS_D, sizeof(double) * obs.size(), hipMemcpyDeviceToHost));
        std::vector<vinit::Landmark> lms((size_t)be.n_lm);
        for (int k = 0; k < be.n_lm; k++) {
            const int slot = order[k];
            lms[k].id = id[slot]; lms[k].start = st[slot];
            lms[k].obs.resize(no[slot]);
            for (int a = 0; a < no[slot]; a++) {
                const double *q = &obs[((size_t)slot * W1 + (st[slot] + a + be.ring_base) % W1) * VIO_OBS_D];
                lms[k].obs[a] = {q[0], q[1], q[8]};
            }
        }
        vinit::run(C.c, W, be.Headers, be.Bgs[0], be.ric, be.tic, D.frames, lms, res);
        D.last_stage = res.stage;
        if (!res.ok) D.failures++;
        D.initial_timestamp = be.cur_stamp;
        if (res.ok || res.stage == 4) {   // solveGyroscopeBias already moved Bgs when the alignment rejects the attempt
            for (int i = 0; i <= W; i++) for (int k = 0; k < 3; k++) be.Bgs[i][k] += res.delta_bg[k];
            changed = true;
        }
        if (res.force_margin_old) { be.marginalization_flag = 0; changed = true; }
    }
    if (res.ok) {
        for (int i = 0; i <= W; i++) {
            for (int k = 0; k < 3; k++) { be.Ps[i][k] = res.Ps[i][k]; be.Vs[i][k] = res.Vs[i][k]; if (res.set_ba) be.Bas[i][k] = res.Ba[k]; }
            for (int k = 0; k < 9; k++) be.Rs[i][k] = res.Rs[i][k];
        }
        for (int k = 0; k < 3; k++) be.g[k] = res.g[k];
        be.solver_flag = 1; be.init_frame = 1; be.do_solve = 1; be.do_marg = 1;
        {
            // IMU steps of every window slot from the image-frame mirror: slot j spans the image frames in (Headers[j - 1], Headers[j]]
            std::vector<double> samp;
            std::vector<int> offs(VIO_MAXW + 3, 0);
            for (int j = 1; j <= W; j++) {
                for (const auto &f : D.frames)
                    if (f.stamp > be.Headers[j - 1] && f.stamp <= be.Headers[j])
                        for (size_t k = 0; k < f.dt.size(); k++) {
                            samp.push_back(f.dt[k]);
                            for (int q = 0; q < 3; q++) samp.push_back(f.acc[3 * k + q]);
                            for (int q = 0; q < 3; q++) samp.push_back(f.gyr[3 * k + q]);
                        }
                offs[j + 1] = (int)(samp.size() / 7);
            }
            offs[1] = 0;
            if (samp.size() > h->dyn_samples_cap) {
                if (h->d_dyn_samples) (void)hipFree(h->d_dyn_samples);
                h->dyn_samples_cap = samp.size() * 2 + 7 * 1024;
                HIPCHK(hipMalloc((void **)&h->d_dyn_samples, h->dyn_samples_cap * sizeof(double)));
            }
            if (!h->d_dyn_offs) HIPCHK(hipMalloc((void **)&h->d_dyn_offs, (VIO_MAXW + 3) * sizeof(int)));
            if (!samp.empty()) HIPCHK(hipMemcpy(h->d_dyn_samples, samp.data(), samp.size() * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(h->d_dyn_offs, offs.data(), (VIO_MAXW + 3) * sizeof(int), hipMemcpyHostToDevice));
        }
        D.frames.clear();
        D.nonlinear = true;
        *finalize = true;
        changed = true;
    } else {
        be.dyn_failed = 1;
        changed = true;
        if (be.marginalization_flag == 0) {   // slideWindow(MARGIN_OLD) drops every image frame up to the old frame 0 (estimator.cpp:1632-1650)
            const double t0 = be.Headers[0];
            size_t k = 0;
            while (k < D.frames.size() && D.frames[k].stamp <= t0) k++;
            bool found = false;
            for (const auto &f : D.frames) if (f.stamp == t0) found = true;
            if (found) D.frames.erase(D.frames.begin(), D.frames.begin() + k);
        }
    }
    if (changed) HIPCHK(hipMemcpy(h->B.be + s, &be, sizeof(BeSeq), hipMemcpyHostToDevice));
    return VIO_OK;
}

// one_seq >= 0: only that sequence (vio_process_obs), otherwise every sequence of the group
int launch_backend(vio_batch *h, vio_batch::Group &g, const uint16_t *d_depth, const IngestSrc &src, int one_seq = -1) {
    const DevCfg &C = h->hc;
    const int S = one_seq >= 0 ? 1 : g.n;
    hipStream_t st = g.stream;
    Batch Bg = h->B;
    Bg.s0 = one_seq >= 0 ? one_seq : g.s0;
    const bool prof = one_seq < 0;
    if (prof) PEV(h, 8);
    be_ingest_kernel<<<S, 256, (size_t)C.lm_hash_size * 8, st>>>(Bg, d_depth, (size_t)C.c.width * C.c.height, src);
    if (prof) PEV(h, 9);
    if (one_seq < 0) { (void)hipEventRecord(g.ev_ingest, st); g.have_ingest_ev = true; }  // tracker lag 1: the next frame's front-end starts here
    if (h->dyn_active || (C.c.dynamic_init && one_seq >= 0)) {
        // some sequence of this handle is still INITIAL under static_init: 0: the host collects its image frame / runs the
        // initialisation between the two halves of the back-end (once per sequence; the steady state below never synchronises)
        HIPCHK(hipStreamSynchronize(st));
        for (int s = Bg.s0; s < Bg.s0 + S; s++) {
            if (h->dyn[s].nonlinear && one_seq < 0) continue;
            bool fin = false;
            int rc = dynamic_init_step(h, s, src, &fin);
            if (rc != VIO_OK) return rc;
            if (fin) { be_dyn_finalize_kernel<<<1, 256, 0, st>>>(h->B, s, h->d_dyn_samples, h->d_dyn_offs); HIPCHK(hipStreamSynchronize(st)); }  // (the staging buffers are shared by all sequences)
        }
        bool any = false;
        for (int s = 0; s < h->S; s++) any = any || !h->dyn[s].nonlinear;
        h->dyn_active = any;
    }
    const int be_threads = h->be_threads;
    if (h->solve_mode == 1) {
        // phased solver: every data-parallel phase of the trust-region loop covers all sequences with many workgroups; max_iterations
        // + 1 slots carry the iterations, two more absorb Cholesky retries (a converged sequence falls through the remaining launches)
        // The two-kernel path (ps_eval + ps_asm_a) is launched beside the fused kernel only where a solve of this handle can need it: the
        // extrinsic / td blocks may open (42-double records), the handle has ever been handed a relocalisation request (vio_set_relo_frame; the request waits on the device for its sequence's next solve), or the
        // residual list can plausibly outgrow the fused kernel's PS_FUSE_MAXBLK chunks (a solve that does so anyway on a fused-only handle is skipped and
        // flagged, overflow bit 256).  Decided on the host from the configuration: deterministic, no device feedback.
        // (VIO_FUSE, default 0: measured on the canonical workload the fused kernel moves 28 % less data -- 0.80 against 1.11 GB per 64-sequence solve --
        // and wins where the device is throughput-bound (S = 256: +2.5 %, S = 512: +2 %), but at 64 sequences per stream group, where the chain of
        // dependent launches bounds the step, the two-kernel path is 2 % faster: its smaller workgroups share the CUs better with the other group's
        // kernels.  One path per handle whatever the batch size, so that a sequence's result never depends on the batch it runs in.)
        const bool fuse_only = Bg.fuse > 0 && !C.c.estimate_extrinsic && !C.c.estimate_td && h->relo_frames <= 0 &&
                               (size_t)(C.W + 1) * C.c.max_cnt * 12 / 10 <= (size_t)PS_FUSE_MAXBLK * (PS_FUSE_CAP - C.W);
        Bg.fuse_only = fuse_only ? 1 : 0;
        auto launch_phased = [&]() {
            ps_setup_kernel<<<S, 512, 0, st>>>(Bg);
            const int slots = C.c.max_iterations + h->extra_slots;
            // XCD-aware block map (ps_blk): every block of a sequence on the XCD its one-block kernels run on
            const bool xm = h->xcd_map && S >= 8;
            const int XN = h->xcd_n > 0 ? h->xcd_n : 8;   // (measured with the back-end streams masked to six XCDs: 8 -> +2 %, 6 -> +0.3 %, 3 / 12 -> -1 %: the block -> XCD rotation ignores the mask)
            const int nb_e = h->ps_eval_blocks, nb_a = h->ps_asm_a_blocks, nb_bs = h->ps_asm_b_blocks + h->ps_schur_tiles + (Bg.gn_ext ? 1 : 0), S8 = XN * ((S + XN - 1) / XN);
            Batch Be = Bg, Ba = Bg, Bb = Bg;
            Be.ns = Ba.ns = Bb.ns = S;
            Be.xcd_n = Ba.xcd_n = Bb.xcd_n = XN;
            Be.xcd_nb = xm ? nb_e : 0; Ba.xcd_nb = xm ? nb_a : 0; Bb.xcd_nb = xm ? nb_bs : 0;
            const dim3 g_e = xm ? dim3(S8 * nb_e) : dim3(nb_e, S), g_a = xm ? dim3(S8 * nb_a) : dim3(nb_a, S), g_b = xm ? dim3(S8 * nb_bs) : dim3(nb_bs, S);
            Batch Bf = Bg;
            Bf.ns = S; Bf.xcd_n = XN; Bf.xcd_nb = xm ? h->ps_evalf_blocks : 0;
            const dim3 g_f = xm ? dim3(S8 * h->ps_evalf_blocks) : dim3(h->ps_evalf_blocks, S);
            for (int k = 0; k < slots; k++) {
                // fused evaluate + assemble for the solves that qualify (st.fused); ps_eval / ps_asm_a serve the others and idle for these
                if (Bg.fuse) ps_evalf_kernel<<<g_f, 256, h->lds_ps_evalf, st>>>(Bf);
                if (!fuse_only) {
                    // (idle launches are not free here: their workgroups ask for 40 KB of LDS each and queue behind the other stream group's
                    // fused workgroups, which fill the CUs' LDS -- 15 us per idle launch, measured)
                    if (h->eval_occ == 4) ps_eval_kernel_occ4<<<g_e, 256, h->lds_ps_eval, st>>>(Be);
                    else if (h->eval_occ == 3) ps_eval_kernel_occ3<<<g_e, 256, h->lds_ps_eval, st>>>(Be);
                    else ps_eval_kernel<<<g_e, 256, h->lds_ps_eval, st>>>(Be);
                    if (h->asm_a_occ4) ps_asm_a_kernel_occ4<<<g_a, 512, 0, st>>>(Ba);
                    else ps_asm_a_kernel<<<g_a, 512, 0, st>>>(Ba);
                }
                ps_asm_b_schur_kernel<<<g_b, 256, (size_t)(C.NL + 16 + (6 * (C.W + 1) + 7 <= 128 ? 8 * 128 : 8 * 192) /* eight compact partial rows of ps_gn_rhs_body (>= the 4 x 256 of a Schur tile) */) * sizeof(double), st>>>(Bb, h->ps_asm_b_blocks, h->asm_b_by_blocks);   // per-row factors + the partial tiles of wavefronts 1 .. 3 + the tile of H (form_s)
                if (h->serial_big) ps_serial_big_kernel<<<S, 512, h->lds_serial, st>>>(Bg);
                else ps_serial_kernel_512<<<S, 512, h->lds_serial, st>>>(Bg);
                // Ceres' projected line search of bounds-constrained solves (one workgroup per sequence, idle otherwise); the candidate of the
                // last slot is never evaluated, so no search follows it
                if (h->line_search && k + 1 < slots) ps_ls_kernel<<<S, 256, h->lds_ps_ls, st>>>(Bg);
            }
            ps_final_kernel<<<S, 256, 0, st>>>(Bg);
        };
        // VIO_GRAPH: the 43 launches of one solve replayed as a hipGraph captured on the group's stream at its first use (the kernel
        // arguments -- the Batch of the group -- normally do not change from frame to frame)
        if (h->use_graph && one_seq < 0) {
            // the capture bakes in the Batch (by value) and the launch knobs: key it on their bytes, so that any setter that touches h->B
            // (vio_set_tracker_lag, vio_set_fisheye_mask, ...) or a knob makes the next solve re-capture instead of replaying stale arguments
            uint64_t key = 1469598103934665603ULL;
            auto mixin = [&](const void *p, size_t nbytes) { const unsigned char *q = (const unsigned char *)p; for (size_t i = 0; i < nbytes; i++) { key ^= q[i]; key *= 1099511628211ULL; } };
            mixin(&Bg, sizeof(Bg));
            mixin(&h->line_search, sizeof(h->line_search));
            mixin(&fuse_only, sizeof(fuse_only));
            const int knobs[12] = {C.c.max_iterations + h->extra_slots, h->eval_occ, h->asm_a_occ4 ? 1 : 0, h->serial_big ? 1 : 0, h->serial_threads, h->ps_eval_blocks, h->ps_asm_a_blocks,
                                   h->ps_asm_b_blocks + 1000 * h->asm_b_by_blocks, h->ps_schur_tiles, h->xcd_map, h->xcd_n, (int)h->lds_serial};
            mixin(knobs, sizeof(knobs));
            mixin(&h->lds_ps_eval, sizeof(h->lds_ps_eval));
            if (g.solve_graph && g.solve_graph_key != key) { (void)hipGraphExecDestroy(g.solve_graph); g.solve_graph = nullptr; }
            g.solve_graph_key = key;
            if (!g.solve_graph) {
                hipGraph_t gr = nullptr;
                HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                launch_phased();
                HIPCHK(hipStreamEndCapture(st, &gr));
                HIPCHK(hipGraphInstantiate(&g.solve_graph, gr, nullptr, nullptr, 0));
                (void)hipGraphDestroy(gr);
            }
            HIPCHK(hipGraphLaunch(g.solve_graph, st));
        } else
            launch_phased();
    } else if (be_threads <= 512) be_solve_kernel_512<<<S, be_threads, h->lds_solve, st>>>(Bg);
    else be_solve_kernel<<<S, be_threads, h->lds_solve, st>>>(Bg);
    if (prof) PEV(h, 10);
    (void)hipEventRecord(g.ev_solve, st);  // the next frame's front-end may start here (it only needs latest_Bg / td / imu_head, and a
    g.have_solve_ev = true;                // reboot - which rewrites them - is decided at the end of be_solve)
    if (C.c.dynamic_init && one_seq < 0 && &g == &h->groups.back()) {
        // the host must learn about reboots (a rebooted sequence is INITIAL again and needs its image frames collected from its very
        // next frame): solver flags after this solve travel to pinned memory and are looked at when the next frame is fed
        state_gather_kernel<<<(h->S + 255) / 256, 256, 0, st>>>(h->B, h->d_state);
        HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(int) * h->S, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(h->ev_state, st));
        h->state_pending = true;
    }
    if (C.MX > 0) be_marg_exact_kernel<<<S, h->marg_threads, h->lds_marg, st>>>(Bg);   // vio_config.marg_exact (parity instrument)
    else be_marg_kernel<<<S, h->marg_threads, h->lds_marg, st>>>(Bg);  // marginalisation + window slide (be_finish is fused into it)
    if (prof) PEV(h, 11);
    HIPCHK(hipGetLastError());
    return VIO_OK;
}

// dynamic initialisation: before a frame is fed, pick up the solver flags of the previous one (see launch_backend)
int refresh_dynamic_state(vio_batch *h) {
    if (!h->hc.c.dynamic_init) return VIO_OK;
    if (h->state_pending) {
        HIPCHK(hipEventSynchronize(h->ev_state));
        h->state_pending = false;
        bool any = false;
        for (int s = 0; s < h->S; s++) {
            if (h->h_state[s] == 0 && h->dyn[s].nonlinear) h->dyn[s] = vio_batch::DynSeq();   // rebooted
            if (h->h_state[s] != 0) h->dyn[s].nonlinear = true;
            any = any || !h->dyn[s].nonlinear;
        }
        h->dyn_active = any;
    }
    return VIO_OK;
}

}  // namespace

extern "C" {

const char *vio_last_error(void) { return g_err.c_str(); }

void vio_config_default(vio_config *c) {
    memset(c, 0, sizeof(*c));
    c->width = 640; c->height = 480;
    c->max_cnt = 150; c->min_dist = 15;
    c->grid_rows = 5; c->grid_cols = 6;
    c->window_size = 10;
    c->max_landmarks = 1000;
    c->fix_depth = 1;
    c->estimate_extrinsic = 0;
    c->estimate_td = 0;
    c->max_iterations = 8;
    c->ransac_max_iters = 1000;
    c->lk_max_level = 1;
    c->use_imu = 1;
    c->fx = 604.5821781259577; c->fy = 604.2544712985845; c->cx = 321.2638233484251; c->cy = 239.70969315130674;
    c->k1 = 0.13387871564774004; c->k2 = -0.2731913133377051; c->p1 = 0.0020296263577681264; c->p2 = -0.00044384544608203714;
    c->focal_length = 460.0;
    c->f_threshold = 1.0;
    c->depth_min = 0.3; c->depth_max = 6.0;
    c->acc_n = 0.1; c->acc_w = 0.001; c->gyr_n = 0.01; c->gyr_w = 0.0001; c->g_norm = 9.805;
    const double ric[9] = {0.02629567, -0.00713751, 0.99962873, -0.99934346, 0.02474397, 0.02646484, -0.02492368, -0.99966834, -0.00648216};
    const double tic[3] = {0.17336835, 0.049596, -0.10574841};
    memcpy(c->ric, ric, sizeof(ric));
    memcpy(c->tic, tic, sizeof(tic));
    c->td = 0.0; c->tr = 0.0;
    c->min_parallax_px = 10.0;
    c->init_depth = 5.0;
}

static int build_devcfg(const vio_config *cfg, int imu_capacity, DevCfg &C) {
    memset(&C, 0, sizeof(C));
    C.c = *cfg;
    const vio_config &c = C.c;
    if (c.width < 64 || c.height < 64 || c.width > 4095 || c.height > 4095) { g_err = "image size out of range"; return VIO_EINVAL; }
    if (c.window_size < 4 || c.window_size > VIO_MAXW) { g_err = "window_size must be 4..20"; return VIO_EINVAL; }
    if (c.dynamic_init < 0 || c.dynamic_init > 1) { g_err = "dynamic_init must be 0 or 1"; return VIO_EINVAL; }
    if (c.use_imu < 0 || c.use_imu > 1) { g_err = "use_imu must be 0 or 1"; return VIO_EINVAL; }
    if (!c.use_imu && c.dynamic_init) { g_err = "the dynamic initialisation needs the IMU (estimator.cpp:231)"; return VIO_EINVAL; }
    if (c.grid_rows < 1 || c.grid_cols < 1 || c.grid_rows * c.grid_cols > VIO_MAX_CELLS) { g_err = "too many grid cells"; return VIO_EINVAL; }
    if (c.min_dist < 1 || c.min_dist > 63) { g_err = "min_dist must be 1..63"; return VIO_EINVAL; }
    if (c.lk_max_level < 0 || c.lk_max_level > 3) { g_err = "lk_max_level must be 0..3"; return VIO_EINVAL; }
    if (c.estimate_extrinsic < 0 || c.estimate_extrinsic > 1) { g_err = "estimate_extrinsic must be 0 or 1"; return VIO_EINVAL; }
    {   // readParameters() re-orthonormalises the extrinsic rotation through a normalised quaternion (parameters.cpp:202-209)
        dm::m3 Rc = dm::q2R(dm::qnormalized(dm::R2q(dm::ldm(c.ric))));
        dm::stm(C.c.ric, Rc);
    }
    C.W = c.window_size;
    C.ncells = c.grid_rows * c.grid_cols;
    C.grids_threshold = c.max_cnt / C.ncells;
    if (C.grids_threshold <= 0) { g_err = "max_cnt must exceed the number of grid cells (feature_tracker.cpp:88-93)"; return VIO_EINVAL; }
    // initGridsDetector (feature_tracker.cpp:33-94)
    C.grid_h = c.height / c.grid_rows;
    C.grid_w = c.width / c.grid_cols;
    int res_h = c.height - (c.grid_rows - 1) * C.grid_h, res_w = c.width - (c.grid_cols - 1) * C.grid_w;
    for (int i = 0; i < c.grid_rows; i++)
        for (int j = 0; j < c.grid_cols; j++) {
            GridRect r;
            r.x = j == 0 ? 0 : j * C.grid_w - 3;
            r.y = i == 0 ? 0 : i * C.grid_h - 3;
            int gw = (j == c.grid_cols - 1) ? res_w : C.grid_w, gh = (i == c.grid_rows - 1) ? res_h : C.grid_h;
            r.w = gw + ((j > 0 && j < c.grid_cols - 1) ? 6 : 3);
            r.h = gh + ((i > 0 && i < c.grid_rows - 1) ? 6 : 3);
            if (c.grid_cols == 1) r.w = gw;
            if (c.grid_rows == 1) r.h = gh;
            C.rect[i * c.grid_cols + j] = r;
        }
    circle_halfwidths(c.min_dist, C.circle_hw);
    C.NP = (c.max_cnt + C.ncells * (C.grids_threshold + 2) + 8 + 7) & ~7;
    if (C.NP > VIO_FAST_CAP) { g_err = "max_cnt too large for this build"; return VIO_EINVAL; }
    C.NL = c.max_landmarks < C.NP ? C.NP : c.max_landmarks;
    C.NL = (C.NL + 7) & ~7;
    C.lm_hash_size = 256;
    while (C.lm_hash_size < 2 * C.NL) C.lm_hash_size <<= 1;
    C.NRES = std::max(4 * C.NL, C.W * C.NP + 2 * C.NL);
    C.NRES = (C.NRES + 63) & ~63;
    C.NIMU = imu_capacity < 256 ? 256 : imu_capacity;
    C.P = 15 * (C.W + 1) + 7;
    C.NPRIOR = 6 * C.W + 16;
    C.LW = (C.P + 15) & ~15;
    int off = 0, sw = c.width, sh = c.height;
    C.lvl_w[0] = sw; C.lvl_h[0] = sh; C.lvl_off[0] = 0;
    for (int l = 1; l <= 3; l++) {
        sw = (sw + 1) / 2; sh = (sh + 1) / 2;
        C.lvl_w[l] = sw; C.lvl_h[l] = sh; C.lvl_off[l] = off;
        off += (sw * sh + 63) & ~63;
    }
    C.pyr_bytes = off;
    if (c.marg_exact < 0 || c.marg_exact > 2) { g_err = "marg_exact must be 0, 1 or 2"; return VIO_EINVAL; }
    if (c.equalize < 0 || c.equalize > 1) { g_err = "equalize must be 0 or 1"; return VIO_EINVAL; }
    // marg_exact 1: scratch for the full marginalised block (a landmark that starts in frame 0 was packaged by the tracker in that frame: at most NP
    // of them); marg_exact 2 never forms that block (MX = 16 only selects be_marg_exact_kernel and its LDS budget)
    C.MX = c.marg_exact == 1 ? std::min(15 + C.NP, 495) : (c.marg_exact == 2 ? 16 : 0);
    return VIO_OK;
}

// The two streams of a stream group.  partitioned (tracker lag 1): compute-unit partition by hipExtStreamCreateWithCUMask with
// contiguous ranges = whole XCDs.  The wide front-end kernels of frame f+1 then overlap the latency-bound solver kernels of frame f, and
// keeping them on their own XCDs leaves the solver's L2 slices and dispatchers alone (measured: +5 %; stream priorities and an
// interleaved mask did nothing).  VIO_FE_CUS = n (default 64, 0 = no partition): front-end streams use CUs [0, n), back-end streams
// the rest; VIO_BE_CU_SPLIT = 1: the back-end streams of the groups share [n, 256) in equal contiguous parts; VIO_BE_CU_ALL = 1: back-end
// on all CUs.  With lag 0 the front-end is on the critical path itself and gets the whole device.
static int create_group_streams(vio_batch *h, vio_batch::Group &g, bool partitioned) {
    if (g.solve_graph) { (void)hipGraphExecDestroy(g.solve_graph); g.solve_graph = nullptr; }   // captured on the stream that goes away
    if (g.stream) { (void)hipStreamDestroy(g.stream); g.stream = nullptr; }
    if (g.fe_stream) { (void)hipStreamDestroy(g.fe_stream); g.fe_stream = nullptr; }
    // the partition is sized from the device: a quarter of its compute units (64 of the MI355X's 256 = two XCDs) for the front-end by
    // default; parts with a different CU count get the same proportion, and a split that leaves either side empty falls back to
    // unpartitioned streams
    int n_cu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 0;
    const char *fe_cus_env = getenv("VIO_FE_CUS");
    // Round 4: no partition by default.  The front-end of a 64-sequence group now takes 0.8 ms on the whole device (2.7 ms on its 64-CU
    // partition in round 3) and no longer slows the solver chain measurably (43.4 k frames/s either way), and on every XCD the XCD-aware
    // block map of fe_lk cuts its HBM-side traffic to a third.  VIO_FE_CUS = n > 0 restores the partition (and switches that map off: under a
    // CU mask the workgroup -> XCD rotation it relies on does not hold).
    // Round 6 (second half): with the solver chain a fifth shorter the wide front-end kernels of the next frame disturb it again where the device is
    // NOT full -- up to 128 sequences per handle a quarter of the CUs (two XCDs) for the front-end streams is back as the default: 54.3 -> 56.5 k
    // frames/s at 128 sequences, +4 % at 64, nothing at 16 or 1, nothing at tracker lag 0; beyond 128 sequences the partition costs throughput
    // (192: -2.3 %, 256: -3.7 %, 512: -6.7 %) and stays off.  Scheduling only: the results do not depend on it.
    const int fe_cus = fe_cus_env ? atoi(fe_cus_env) : (h->S <= 128 ? n_cu / 4 : 0);
    hipError_t e_fe, e_be;
    h->fe_partitioned = false;
    if (partitioned && n_cu >= 8 && n_cu <= 1024 && fe_cus > 0 && fe_cus < n_cu) {
        h->fe_partitioned = true;
        const int nw = (n_cu + 31) / 32;
        std::vector<uint32_t> mfe(nw, 0u), mbe(nw, 0u);
        const int ng = (int)h->groups.size(), gi = (int)(&g - &h->groups[0]);
        const bool split = getenv("VIO_BE_CU_SPLIT") && atoi(getenv("VIO_BE_CU_SPLIT")) != 0 && ng > 1 && (n_cu - fe_cus) / ng >= 1;
        const int per = (n_cu - fe_cus) / ng;
        const int b0 = split ? fe_cus + gi * per : fe_cus, b1 = split ? (gi == ng - 1 ? n_cu : b0 + per) : n_cu;
        for (int q = 0; q < n_cu; q++) {
            if (q < fe_cus) mfe[q >> 5] |= 1u << (q & 31);
            if (q >= b0 && q < b1) mbe[q >> 5] |= 1u << (q & 31);
        }
        e_fe = hipExtStreamCreateWithCUMask(&g.fe_stream, (uint32_t)nw, mfe.data());
        e_be = getenv("VIO_BE_CU_ALL") ? hipStreamCreate(&g.stream) : hipExtStreamCreateWithCUMask(&g.stream, (uint32_t)nw, mbe.data());
    } else {
        e_fe = hipStreamCreate(&g.fe_stream);
        e_be = hipStreamCreate(&g.stream);
    }
    if (e_be != hipSuccess || e_fe != hipSuccess) { g_err = "stream create failed"; return VIO_EDEVICE; }
    return VIO_OK;
}

vio_batch *vio_create(const vio_config *cfg, int n_seq, int imu_capacity) { return vio_create_on_device(cfg, n_seq, imu_capacity, -1); }
int vio_get_device(vio_batch *h) { return h ? h->device : VIO_EINVAL; }

vio_batch *vio_create_on_device(const vio_config *cfg, int n_seq, int imu_capacity, int device) {
    if (!cfg || n_seq < 1) { g_err = "bad arguments"; return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_err = "no HIP device (the product path has no CPU fallback)"; return nullptr; }
    if (device >= ndev) { g_err = "vio_create_on_device: no such HIP device"; return nullptr; }
    int caller_dev = 0;
    if (hipGetDevice(&caller_dev) != hipSuccess) { g_err = "hipGetDevice failed"; return nullptr; }
    vio_batch *h = new vio_batch();
    h->device = device < 0 ? caller_dev : device;
    DevGuard dev_guard(h);   // every allocation, stream and event below is created on h->device; the caller's device is current again on return
    if (dev_guard.failed) { g_err = "vio_create_on_device: hipSetDevice failed"; delete h; return nullptr; }
    if (build_devcfg(cfg, imu_capacity, h->hc) != VIO_OK) { delete h; return nullptr; }
    h->hc.MXL = 0;
    h->hc.eig_one_wave = getenv("VIO_EIG_ONE_WAVE") ? (atoi(getenv("VIO_EIG_ONE_WAVE")) != 0) : 0;
    h->hc.eig_jacobi = getenv("VIO_MARG_EIG_JACOBI") ? (atoi(getenv("VIO_MARG_EIG_JACOBI")) != 0) : 0;
    if (h->hc.MX > 0) {
        // marg_exact: the eigen-decompositions of the literal marginalisation run LDS-resident for blocks up to MXL -- whatever the kernel's
        // static LDS leaves of the workgroup's share (matrix with an odd leading dimension + the solver's vectors, be_kernels.hip
        // MARG_EIG_AUX_DOUBLES), and never beyond the 128 rows tridiag_ql_wave covers
        hipFuncAttributes fa;
        int cap = 0;
        if (hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) != hipSuccess || cap <= 0) cap = 160 * 1024;
        if (hipFuncGetAttributes(&fa, (const void *)be_marg_exact_kernel) == hipSuccess) {
            const long avail = (long)cap - (long)fa.sharedSizeBytes - 256;
            int m = 0;
            while (m < 128 && (long)marg_exact_lds_bytes(m + 1) <= avail) m++;
            if (getenv("VIO_MARG_EIG_LDS")) m = std::min(m, std::max(0, atoi(getenv("VIO_MARG_EIG_LDS"))));   // 0: the round-4 HBM Jacobi everywhere (A/B timing)
            h->hc.MXL = m;
        }
    }
    h->S = n_seq;
    const DevCfg &C = h->hc;
    const size_t S = (size_t)n_seq, NP = C.NP, NL = C.NL, W1 = C.W + 1, HW = (size_t)C.c.width * C.c.height;
    Batch &B = h->B;
    memset(&B, 0, sizeof(B));
    B.S = n_seq;
    int rc = VIO_OK;
#define DA(ptr, n) if (rc == VIO_OK) rc = dalloc(h, &ptr, (size_t)(n))
    DA(B.cfg, 1); DA(B.fe, S); DA(B.be, S); DA(B.pre, S * (C.W + 2));
    DA(B.img, S * 2 * HW); DA(B.pyr, S * 2 * (size_t)C.pyr_bytes);
    if (C.c.equalize) { DA(B.clahe_lut, S * 64 * 256); DA(B.clahe_img, S * HW); }
    DA(B.cur_pts, S * NP); DA(B.forw_pts, S * NP); DA(B.cur_un_pts, S * NP); DA(B.pts_velocity, S * NP); DA(B.prev_un_pt, S * NP);
    DA(B.unstable_pts, S * NP); DA(B.tmp_pts, S * NP);
    DA(B.ids, S * NP); DA(B.track_cnt, S * NP); DA(B.prev_un_id, S * NP); DA(B.tmp_i0, S * NP); DA(B.tmp_i1, S * NP);
    DA(B.lk_status, S * NP); DA(B.accept_xy, S * 2 * NP); DA(B.cand, S * C.ncells * VIO_FAST_CAP);
    DA(B.obs_id, S * NP); DA(B.obs, S * NP * 7);
    DA(B.imu_t, S * C.NIMU); DA(B.imu_acc, S * C.NIMU * 3); DA(B.imu_gyr, S * C.NIMU * 3);
    DA(B.lm_id, S * NL); DA(B.lm_start, S * NL); DA(B.lm_nobs, S * NL); DA(B.lm_est_flag, S * NL); DA(B.lm_solve_flag, S * NL);
    DA(B.lm_dyn, S * NL); DA(B.lm_order, S * NL); DA(B.lm_free, S * NL); DA(B.lm_tmp, S * NL); DA(B.lm_pidx, S * NL); DA(B.lm_aidx, S * NL);
    DA(B.lm_depth, S * NL); DA(B.lm_obs, S * NL * W1 * VIO_OBS_D); DA(B.para_feat, S * NL); DA(B.cand_feat, S * NL);
    DA(B.lm_relo, S * NL); DA(B.relo_xy, S * NL * 2); DA(B.relo_mp, S * NP * 3);
    const size_t n = C.NPRIOR, LW = C.LW, nres = C.NRES, npair = W1 * W1, mq = 15 + n;
    DA(B.prior_J, S * n * n); DA(B.prior_r, S * n); DA(B.prior_x0, S * (C.W * 7 + 17)); DA(B.prior_H, S * n * n); DA(B.prior_rf, S * n);
    // (the landmark rows Hpl / Hll / gl twice: be_phased.h ps_sel_rows -- the fused evaluate + assemble kernel builds a candidate's rows beside the current ones)
    DA(B.H, S * LW * LW); DA(B.Sc, S * LW * LW); DA(B.Hpl, 2 * S * (NL + 8) * LW); DA(B.vec, S * VEC_SLOTS * LW);
    DA(B.Hll, 2 * S * (NL + 8)); DA(B.gl, 2 * S * (NL + 8)); DA(B.lvec, S * (NL + 8) * 8);
    if (C.W <= PS_FUSE_MAXW) DA(B.pairpart, S * PS_FUSE_MAXPAIRS * PS_FUSE_MAXBLK * 210);
    DA(B.ls_scratch, S * ps_eval_lds_bytes(C.W));
    DA(B.res, S * nres * 42); DA(B.res_lm, S * nres); DA(B.res_k, S * nres); DA(B.res_pair, S);
    DA(B.pair_start, S * (npair + 1)); DA(B.pair_list, S * nres); DA(B.pairblk, S * npair * 210);
    DA(B.imu_raw, S * C.W * 15 * 31);
    DA(B.margA, S * mq * mq); DA(B.margB, S * mq); DA(B.margV, S * n * n); DA(B.margW, S * (n + 16) * (n + 16));
    if (C.MX > 0) DA(B.margE, S * ((size_t)3 * C.MX * C.MX + n * (size_t)C.MX));
    DA(B.odom, S * 11); DA(B.timings, 128); DA(B.fe_ticks, S * 4);
    B.hist_cap = 2048;
    B.s0 = 0;
    B.ns = 0; B.xcd_nb = 0; B.xcd_n = 8;
    if (getenv("VIO_BE_THREADS")) h->be_threads = std::min(1024, std::max(64, atoi(getenv("VIO_BE_THREADS")) & ~63));
    if (getenv("VIO_ASM_B_MODE")) h->asm_b_by_blocks = std::max(0, std::min(2, atoi(getenv("VIO_ASM_B_MODE"))));
    // (round 6: 12 for windows up to W = 10 -- with the mirrored assembly and the Schur tiles summing their own entries a launch of 24 left two entries per
    //  thread, and fewer, longer workgroups fill the back-end's CUs in fewer rounds: 56.5 -> 57.1 k at 128 sequences, 76.5 -> 78.0 k at 512; W = 20: 24 stays,
    //  12 / 16 / 48 measured 11.8 / 12.0 / 12.2 k against 12.2 k)
    h->ps_asm_b_blocks = h->hc.W <= 10 ? 12 : 24;
    if (getenv("VIO_ASM_B_BLOCKS")) h->ps_asm_b_blocks = std::max(1, std::min(256, atoi(getenv("VIO_ASM_B_BLOCKS"))));
    if (getenv("VIO_EVAL_OCC")) h->eval_occ = atoi(getenv("VIO_EVAL_OCC"));
    if (getenv("VIO_GRAPH")) h->use_graph = atoi(getenv("VIO_GRAPH")) != 0;
    if (getenv("VIO_XCD_MAP")) h->xcd_map = atoi(getenv("VIO_XCD_MAP"));
    if (getenv("VIO_FE_XCD_MAP")) h->fe_xcd_map = atoi(getenv("VIO_FE_XCD_MAP"));
    if (getenv("VIO_XCD_N")) h->xcd_n = atoi(getenv("VIO_XCD_N"));
    if (getenv("VIO_FEED_THROTTLE")) h->feed_throttle = atoi(getenv("VIO_FEED_THROTTLE")) != 0;
    if (getenv("VIO_UPLOADS_IN_FLIGHT")) h->uploads_in_flight = atoi(getenv("VIO_UPLOADS_IN_FLIGHT")) == 1 ? 1 : 2;
    if (getenv("VIO_EXTRA_SLOTS")) h->extra_slots = std::max(1, atoi(getenv("VIO_EXTRA_SLOTS")));
    if (getenv("VIO_ASM_A_OCC")) h->asm_a_occ4 = atoi(getenv("VIO_ASM_A_OCC")) >= 4;
    // (VIO_SERIAL_THREADS = 1024 is gone: the 1024-thread build of ps_serial had been producing wrong steps since round 4 -- 2.1 iterations per solve,
    // metres of ATE, found in round 6 by a knob sweep -- while bench.py's `valid` only looked at the solver flags; never the default, no test ran it)
    if (getenv("VIO_MARG_THREADS")) h->marg_threads = std::min(512, std::max(64, atoi(getenv("VIO_MARG_THREADS")) & ~63));
    B.flags = getenv("VIO_FLAGS") ? atoi(getenv("VIO_FLAGS")) : 0;
    DA(B.odom_hist, S * (size_t)B.hist_cap * 11); DA(B.odom_count, S);
    DA(B.sst, S);
    DA(h->d_stamps, S); DA(h->d_modes, S); DA(h->d_rrel, S * 9); DA(h->d_r9, 16);
    DA(h->d_in_n, S); DA(h->d_in_stamps, S); DA(h->d_in_ids, S * NP); DA(h->d_in_obs, S * NP * 7);
#undef DA
    B.gW = h->hc.W; B.gP = h->hc.P; B.gLW = h->hc.LW; B.gNL = h->hc.NL; B.gNP = h->hc.NP; B.gNRES = h->hc.NRES; B.gNPRIOR = h->hc.NPRIOR; B.gMX = h->hc.MX;
    if (rc == VIO_OK && hipMemcpy(B.cfg, &h->hc, sizeof(DevCfg), hipMemcpyHostToDevice) != hipSuccess) { g_err = "cfg upload failed"; rc = VIO_EDEVICE; }
    if (rc == VIO_OK) {
        // group size: VIO_GROUP_SEQS sequences per group (at most 16 groups).  Default: one group.  More groups only pay off when
        // the runtime has a hardware queue per stream (GPU_MAX_HW_QUEUES >= 2 x groups, default 4): measured +1.7 % at 8 groups.
        int per = getenv("VIO_GROUP_SEQS") ? atoi(getenv("VIO_GROUP_SEQS")) : n_seq;
        if (per < 1) per = 1;
        if (C.c.dynamic_init) per = n_seq;   // the host half of the dynamic initialisation looks at one solver-flag snapshot per frame
        int ng = (n_seq + per - 1) / per;
        if (ng > 16) { ng = 16; per = (n_seq + ng - 1) / ng; ng = (n_seq + per - 1) / per; }
        h->groups.resize(ng);
        for (int k = 0; k < ng && rc == VIO_OK; k++) {
            vio_batch::Group &g = h->groups[k];
            g.s0 = k * per;
            g.n = std::min(per, n_seq - g.s0);
            if (create_group_streams(h, g, /*partitioned=*/false) != VIO_OK) { rc = VIO_EDEVICE; break; }
            if (hipEventCreateWithFlags(&g.ev_solve, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g.ev_fe, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&g.ev_be, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g.ev_ingest, hipEventDisableTiming) != hipSuccess) { g_err = "event create failed"; rc = VIO_EDEVICE; }
        }
        if (rc == VIO_OK) { h->stream = h->groups[0].stream; h->fe_stream = h->groups[0].fe_stream; }
        if (rc == VIO_OK && hipEventCreateWithFlags(&h->ev_imu, hipEventDisableTiming) != hipSuccess) { g_err = "event create failed"; rc = VIO_EDEVICE; }
    }
    if (rc == VIO_OK && C.c.dynamic_init) {
        h->dyn.assign(n_seq, vio_batch::DynSeq());
        h->dyn_active = true;
        if (hipHostMalloc((void **)&h->h_state, sizeof(int) * n_seq, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&h->d_state, sizeof(int) * n_seq) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_state, hipEventDisableTiming) != hipSuccess) { g_err = "dynamic-init state allocation failed"; rc = VIO_EDEVICE; }
    }
    for (int i = 0; i < 4 && rc == VIO_OK; i++)
        if (hipEventCreate(&h->ev[i]) != hipSuccess) { g_err = "event create failed"; rc = VIO_EDEVICE; }
    if (rc == VIO_OK) { h->last_imu_t.assign(n_seq, -1e300); rc = init_state(h, 0, n_seq); }
    if (rc == VIO_OK) for (auto &g : h->groups) (void)hipEventRecord(g.ev_be, g.stream);
    if (rc == VIO_OK) {
        h->lds_select = (size_t)C.NP * 104 + 260 * 4 + 64;
        h->lds_add = (size_t)C.NP * 16 + (size_t)4 * 192 * 8 /* FE_NEAR_CAP */ + 3 * VIO_FAST_CAP * 4 + 260 * 4 + 64 * 4 + (size_t)C.NP * 12 + 64;
        h->lds_fast = 0;
        for (int k = 0; k < C.ncells; k++) h->lds_fast = std::max(h->lds_fast, fast_lds_bytes(C.rect[k].w, C.rect[k].h));
        {
            size_t npairs = (size_t)(C.W + 1) * C.W / 2;
            size_t workd = std::max((size_t)C.W * 450, npairs * 210 <= 12288 ? npairs * 210 : (size_t)0);
            workd = std::max(workd, (size_t)1400 /* PreWork */);
            workd = std::max(workd, (size_t)4 * 336);
            workd = std::max(workd, ((size_t)(C.W + 1) * (C.W + 1) + 1) * 32);  // frame-pair geometry during evaluate()
            if (C.NPRIOR <= 96) workd = std::max(workd, (size_t)C.NPRIOR * (C.NPRIOR | 1) + 2);
            {
                size_t nb = (size_t)C.LW >> 4, tiles = nb * (nb + 1) / 2 * 256;
                if (tiles <= 16896) workd = std::max(workd, tiles);  // Schur complement / Cholesky tiles resident in LDS
            }
            h->lds_solve = ((size_t)C.LW + 2 + workd) * 8 + 16;
            (void)raise_lds_limit((const void *)be_solve_kernel, (size_t)(h->lds_solve));
            (void)raise_lds_limit((const void *)be_solve_kernel_512, (size_t)(h->lds_solve));
            {
                // ps_serial: xs + the work region = the resident tiles (windows up to W = 10) or one block column of them (larger
                // windows, chol_tiles_stream), never less than the scratch of the mat-vec passes (one row of VIO_LWMAX per wavefront)
                const size_t nb = (size_t)C.LW >> 4, tiles = nb * (nb + 1) / 2 * 256;
                const size_t wk = std::max(tiles <= 16896 ? tiles : (2 * nb + 1) * 256, (size_t)16 * 336);   // (streaming: two block columns + the look-ahead tile, chol_tiles_stream)
                h->serial_big = tiles > 16896;
                // the Schur-complement launch leaves S itself behind (ps_asm_b_body; VIO_FORM_S = 0: ps_serial combines H, U, Sp and mu D^2 as in rounds 2 - 5).
                // W = 20: ps_serial_big 242 -> 213 us per iteration; W = 10: +0.6 % frames/s at 128 sequences, -0.5 % at 512
                h->B.form_s = (getenv("VIO_FORM_S") ? atoi(getenv("VIO_FORM_S")) != 0 : true) && h->asm_b_by_blocks == 2;
                h->lds_serial = ((size_t)C.LW + 2 + wk + 2 + (size_t)14 /* PS_LVEC */ * C.LW) * 8 + 16;   // + the step's vectors (be_phased.h)
                if (getenv("VIO_SERIAL_LDS") && (size_t)atol(getenv("VIO_SERIAL_LDS")) > h->lds_serial) h->lds_serial = (size_t)atol(getenv("VIO_SERIAL_LDS"));   // experiment: a larger request keeps other workgroups off the CU
                (void)raise_lds_limit(h->serial_big ? (const void *)ps_serial_big_kernel : (const void *)ps_serial_kernel_512, h->lds_serial);
            }
            {
                // phased solver: needs the Schur complement as LDS tiles and the column-aware (pose + extrinsic) landmark rows
                const size_t nb = (size_t)C.LW >> 4, tiles = nb * (nb + 1) / 2 * 256, W1 = (size_t)C.W + 1;
                // (larger windows run the same kernels with the Schur complement in HBM / L2 and chol_tiles_stream: W = 20 is 231 tiles)
                const bool eligible = !(B.flags & 1) && (size_t)C.W * C.NP / 256 + 4 <= PS_MAX_EVAL_BLOCKS &&
                                      (size_t)C.W * 768 <= (W1 * W1 - W1 * C.W / 2) * 210 && nb <= 32;
                h->solve_mode = getenv("VIO_SOLVE_MODE") ? atoi(getenv("VIO_SOLVE_MODE")) : 1;   // phased by default where it applies (windows up to ~10 keyframes)
                if (!eligible) h->solve_mode = 0;
                h->lds_ps_eval = ps_eval_lds_bytes(C.W);   // pair geometry / staged pre-integration headers
                (void)raise_lds_limit((const void *)ps_eval_kernel, h->lds_ps_eval);
                // fused evaluate + assemble (VIO_FUSE = 1; default off, see launch_backend): workgroups 0 / 1 (prior, IMU) + one workgroup per chunk of at most PS_FUSE_CAP - W residuals; a
                // frame observes at most max_cnt landmarks, so a window holds at most (W + 1) max_cnt observations.  B.fuse = chunks the grid covers
                // (ps_setup sends solves that need more, or whose records are not compact, down the ps_eval + ps_asm_a path)
                h->lds_ps_evalf = ps_evalf_lds_bytes(C.W);
                {
                    const int want = getenv("VIO_FUSE") ? atoi(getenv("VIO_FUSE")) : 0;
                    // (the tracker holds a little over max_cnt features per frame -- every grid cell may add k + 2 -- hence the 10 %; residual lists that
                    // need more chunks than the grid has workgroups make them loop, up to PS_FUSE_MAXBLK chunks)
                    int chunks = (int)std::min<size_t>(PS_FUSE_MAXBLK, (W1 * (size_t)C.c.max_cnt * 11 / 10 + (PS_FUSE_CAP - C.W) - 1) / (PS_FUSE_CAP - C.W));
                    if (want > 1) chunks = std::min(PS_FUSE_MAXBLK, want);
                    B.fuse = (want && C.W <= PS_FUSE_MAXW && C.c.use_imu && B.pairpart) ? std::max(1, chunks) : 0;
                    if (B.fuse && raise_lds_limit((const void *)ps_evalf_kernel, h->lds_ps_evalf) != 0) B.fuse = 0;
                    h->ps_evalf_blocks = 2 + B.fuse;
                }
                h->lds_ps_ls = ps_ls_lds_bytes(C.W);
                (void)raise_lds_limit((const void *)ps_ls_kernel, h->lds_ps_ls);
                // reference_quirks bit 3 (tests) / VIO_LINE_SEARCH=0 (measurement): the clamp-only treatment of the inverse-depth bound, no search launches
                h->line_search = !(C.c.reference_quirks & VIO_QUIRK_BOUND_CLAMP_ONLY) && !(getenv("VIO_LINE_SEARCH") && atoi(getenv("VIO_LINE_SEARCH")) == 0);
                (void)raise_lds_limit((const void *)ps_eval_kernel_occ3, h->lds_ps_eval);
                (void)raise_lds_limit((const void *)ps_eval_kernel_occ4, h->lds_ps_eval);
                B.eval_rpt = getenv("VIO_EVAL_RPT") ? std::max(1, std::min(4, atoi(getenv("VIO_EVAL_RPT")))) : 2;
                h->ps_eval_blocks = (int)std::min<size_t>(PS_MAX_EVAL_BLOCKS, (size_t)C.W * C.NP / (256 * (size_t)B.eval_rpt) + 4);
                h->ps_asm_a_blocks = (int)((W1 * (W1 - 1) / 2 + C.W + 32 + 7) / 8);   // pair items (i < j), IMU items, PS_ROW_WAVES = 32 landmark-row wavefronts
                int nact = 0;
                for (size_t a = 0; a < nb; a++) for (size_t b2 = 0; b2 <= a; b2++) {
                    auto act = [&](size_t cb) { const size_t c0 = 16 * cb, c1 = c0 + 15; return c0 < 6 * W1 || (c1 >= 15 * W1 && c0 < 15 * W1 + 7); };
                    if (act(a) && act(b2)) nact++;
                }
                h->ps_schur_tiles = nact;
                h->B.n_schur = nact;
                h->B.gn_ext = getenv("VIO_GN_EXT") ? (atoi(getenv("VIO_GN_EXT")) != 0) : 1;
            }
        }
        {
            size_t nbq = ((size_t)C.NPRIOR + 15) >> 4;
            h->lds_marg = std::max(std::max((size_t)nbq * (nbq + 1) / 2 * 2048, (size_t)2 * C.NPRIOR * 15 * 8), (size_t)PREINT_MANY_LDS_DOUBLES * 8) + 64;   // (the last: F / V of a chunk of the pre-integration merge)  // lower 16x16 tiles of the new prior (Cholesky for its constant term); before that T1 and A_mr
            if (C.MXL > 0) h->lds_marg = std::max(h->lds_marg, marg_exact_lds_bytes(C.MXL));
            if (C.MX > 0) h->lds_marg = std::max(h->lds_marg, (size_t)7 * 512 * 8 + 64);   // sym_eig_hbm's vectors (SYM_EIG_HBM_LDS_DOUBLES)
            h->lds_factor = C.NPRIOR <= 96 ? (size_t)C.NPRIOR * (C.NPRIOR | 1) * 8 + 64 : 64;  // on-demand eigen-decomposition (vio_get_prior)
            (void)raise_lds_limit((const void *)be_prior_factor_kernel, h->lds_factor);
        }
        (void)raise_lds_limit((const void *)be_marg_kernel, (size_t)(h->lds_marg));
        (void)raise_lds_limit((const void *)be_marg_exact_kernel, (size_t)(h->lds_marg));
        (void)raise_lds_limit((const void *)be_ingest_kernel, (size_t)(C.lm_hash_size * 8));

        (void)raise_lds_limit((const void *)fe_select_kernel, (size_t)(h->lds_select));
        (void)raise_lds_limit((const void *)fe_add_kernel, (size_t)(h->lds_add));
        (void)raise_lds_limit((const void *)fe_fast_kernel, (size_t)(h->lds_fast));
        bool fits = lds_fits(C.MX > 0 ? (const void *)be_marg_exact_kernel : (const void *)be_marg_kernel, h->lds_marg, "be_marg") &&
                    lds_fits((const void *)be_ingest_kernel, (size_t)C.lm_hash_size * 8, "be_ingest") &&
                    lds_fits((const void *)fe_select_kernel, h->lds_select, "fe_select") && lds_fits((const void *)fe_add_kernel, h->lds_add, "fe_add") &&
                    lds_fits((const void *)fe_fast_kernel, h->lds_fast, "fe_fast");
        if (fits && h->solve_mode == 1)
            fits = lds_fits((const void *)ps_eval_kernel, h->lds_ps_eval, "ps_eval") && lds_fits((const void *)ps_ls_kernel, h->lds_ps_ls, "ps_ls") &&
                   lds_fits(h->serial_big ? (const void *)ps_serial_big_kernel : (const void *)ps_serial_kernel_512, h->lds_serial, "ps_serial");
        else if (fits)
            fits = lds_fits(h->be_threads <= 512 ? (const void *)be_solve_kernel_512 : (const void *)be_solve_kernel, h->lds_solve, "be_solve");
        if (!fits) rc = VIO_ECAPACITY;
    }
    if (rc != VIO_OK) { vio_destroy(h); return nullptr; }
    return h;
}

void vio_destroy(vio_batch *h) {
    if (!h) return;
    int caller_dev = -1;   // (not a DevGuard: the handle dies inside this function)
    const bool dev_switched = h->device >= 0 && hipGetDevice(&caller_dev) == hipSuccess && caller_dev != h->device && hipSetDevice(h->device) == hipSuccess;
    (void)hipDeviceSynchronize();
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->d_gray_stage) (void)hipFree(h->d_gray_stage);
    if (h->d_fisheye) (void)hipFree(h->d_fisheye);
    if (h->d_depth_stage) (void)hipFree(h->d_depth_stage);
    if (h->d_gray_stage1) (void)hipFree(h->d_gray_stage1);
    if (h->d_depth_stage1) (void)hipFree(h->d_depth_stage1);
    for (auto &sg : h->imu_stage) {
        if (sg.h_seq) {
            (void)hipHostFree(sg.h_seq); (void)hipHostFree(sg.h_t); (void)hipHostFree(sg.h_acc); (void)hipHostFree(sg.h_gyr);
            (void)hipFree(sg.d_seq); (void)hipFree(sg.d_t); (void)hipFree(sg.d_acc); (void)hipFree(sg.d_gyr);
        }
        if (sg.done) (void)hipEventDestroy(sg.done);
    }
    if (h->ev_imu) (void)hipEventDestroy(h->ev_imu);
    if (h->ev_state) (void)hipEventDestroy(h->ev_state);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_dyn_samples) (void)hipFree(h->d_dyn_samples);
    if (h->d_dyn_offs) (void)hipFree(h->d_dyn_offs);
    for (hipEvent_t e : h->pev) (void)hipEventDestroy(e);
    for (auto &g : h->groups) {
        if (g.stream) (void)hipStreamDestroy(g.stream);
        if (g.fe_stream) (void)hipStreamDestroy(g.fe_stream);
        if (g.ev_solve) (void)hipEventDestroy(g.ev_solve);
        if (g.ev_fe) (void)hipEventDestroy(g.ev_fe);
        if (g.ev_be) (void)hipEventDestroy(g.ev_be);
        if (g.ev_ingest) (void)hipEventDestroy(g.ev_ingest);
        if (g.solve_graph) (void)hipGraphExecDestroy(g.solve_graph);
        for (int p = 0; p < 2; p++) {
            if (g.ev_up_gray[p]) (void)hipEventDestroy(g.ev_up_gray[p]);
            if (g.ev_up_depth[p]) (void)hipEventDestroy(g.ev_up_depth[p]);
        }
        for (int p = 0; p < 2; p++) {
            if (g.ev_rd_gray[p]) (void)hipEventDestroy(g.ev_rd_gray[p]);
            if (g.ev_rd_depth[p]) (void)hipEventDestroy(g.ev_rd_depth[p]);
        }
        if (g.ev_host_fe) (void)hipEventDestroy(g.ev_host_fe);
        if (g.side_ring) { (void)hipHostFree(g.side_ring); for (auto e : g.side_ev) if (e) (void)hipEventDestroy(e); }
        if (g.ev_host_be) (void)hipEventDestroy(g.ev_host_be);
        if (g.copy_stream) (void)hipStreamDestroy(g.copy_stream);
        if (g.copy_stream2) (void)hipStreamDestroy(g.copy_stream2);
    }
    for (int i = 0; i < 4; i++) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    delete h;
    if (dev_switched) (void)hipSetDevice(caller_dev);
}

int vio_reset(vio_batch *h) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    return init_state(h, 0, h->S);
}

int vio_reset_seq(vio_batch *h, int seq) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    { int rc_ = refresh_dynamic_state(h); if (rc_ != VIO_OK) return rc_; }   // a reboot of ANOTHER sequence decided by the last solve must not be lost
    return init_state(h, seq, seq + 1, VIO_RESET_ESTIMATOR);
}

int vio_reset_tracker_seq(vio_batch *h, int seq) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    return init_state(h, seq, seq + 1, VIO_RESET_TRACKER);
}

int vio_push_imu(vio_batch *h, int seq, int n, const double *t, const double *acc, const double *gyr) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S || n < 0) return VIO_EINVAL;
    std::lock_guard<std::mutex> lk(h->imu_mu);
    for (int i = 0; i < n; i++) {
        if (!(t[i] > h->last_imu_t[seq])) continue;  // "imu message in disorder" (estimator_nodelet.cpp:110-114)
        h->last_imu_t[seq] = t[i];
        h->p_seq.push_back(seq);
        h->p_t.push_back(t[i]);
        for (int k = 0; k < 3; k++) { h->p_acc.push_back(acc[3 * i + k]); h->p_gyr.push_back(gyr[3 * i + k]); }
    }
    return VIO_OK;
}

// the last reader of a staging buffer has been enqueued on `st`: uploads into that buffer wait for this point
static int note_stage_read(hipEvent_t &ev, bool &have, hipStream_t st) {
    if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev, st));
    have = true;
    return VIO_OK;
}

static int stage_inputs(vio_batch *h, vio_batch::Group &g, const uint8_t *gray, const uint16_t *depth, const double *stamps, int on_device,
                        const uint8_t **dg, const uint16_t **dd, bool overlap = false) {
    // every group uploads its own slice on its own streams: in order with the kernels that consume it, no cross-group hazard
    const DevCfg &C = h->hc;
    size_t HW = (size_t)C.c.width * C.c.height, S = h->S, s0 = g.s0, n = g.n;
    if (stamps) HIPCHK(hipMemcpyAsync(h->d_stamps + s0, stamps + s0, n * sizeof(double), hipMemcpyHostToDevice, g.fe_stream));
    if (on_device) { *dg = gray; *dd = depth; return VIO_OK; }
    // overlap (vio_feed): the uploads run on the group's copy stream into staging buffer g.flip as soon as that buffer is free (the
    // front-end of two frames ago has read the grey image, its be_ingest the depth image), i.e. beside the previous frame's kernels,
    // and the consumers wait for them through events.  Otherwise the copies sit in the consumer's own stream and use buffer 0.
    const int p = overlap ? g.flip : 0;
    if (overlap && !g.copy_stream) {
        HIPCHK(hipStreamCreate(&g.copy_stream));   // (one per group: a copy stream shared by the groups measured 27.9 k against 33.9 k frames/s from page-locked buffers)
        // The depth images go on a second copy stream per group (VIO_COPY_STREAMS = 1: one stream for both).  Rounds 4 - 5 measured no gain (26.4 k against
        // 27.4 k frames/s from page-locked buffers); with the paced feed and the shorter device step of round 6 the second stream is worth 14 %:
        // 38.0 -> 43.4 k from page-locked buffers, 47.2 -> 49.3 k from pageable ones (tools/pcie_sweep.sh, 60 host-fed steps)
        if (!(getenv("VIO_COPY_STREAMS") && atoi(getenv("VIO_COPY_STREAMS")) == 1)) HIPCHK(hipStreamCreate(&g.copy_stream2));
        for (int q = 0; q < 2; q++) {
            HIPCHK(hipEventCreateWithFlags(&g.ev_up_gray[q], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&g.ev_up_depth[q], hipEventDisableTiming));
        }
    }
    static const bool si_trace = getenv("VIO_FEED_TRACE") && atoi(getenv("VIO_FEED_TRACE")) != 0;
    auto si_now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double si0 = si_trace ? si_now() : 0;
    if (overlap && g.up_used[p]) {
        // caller contract (include/vio_abi.h "Host buffers"): the page-locked buffers of a vio_feed call are free once vio_host_buffers_done
        // says so, and at the latest when the SECOND next vio_feed has returned -- this call is about to reuse staging buffer p, whose last
        // upload came from the call before the previous one.  The previous call's upload (buffer p ^ 1) may still be in flight: two uploads
        // overlap with the enqueueing of the kernels (until round 4 every call waited for its predecessor's upload: 28 k against 36 k frames/s
        // from pageable memory)
        HIPCHK(hipEventSynchronize(g.ev_up_gray[p]));
        HIPCHK(hipEventSynchronize(g.ev_up_depth[p]));
    }
    if (overlap && h->uploads_in_flight == 1 && g.up_used[p ^ 1]) {
        // VIO_UPLOADS_IN_FLIGHT=1: the round-4 contract (the previous call's images are free when this call returns)
        HIPCHK(hipEventSynchronize(g.ev_up_gray[p ^ 1]));
        HIPCHK(hipEventSynchronize(g.ev_up_depth[p ^ 1]));
    }
    if (overlap && h->feed_throttle && g.have_rd_depth[p]) {
        // Host-fed feeds are THROTTLED to two frames of lead: wait until the back-end of the frame that last used staging buffer p (two host feeds
        // ago) has finished.  Without it the host runs many frames ahead, the runtime's copy path eventually pushes back -- hipMemcpyAsync blocks
        // for ~7 ms at a time (VIO_FEED_TRACE) -- and the device idles while the host refills its queues: 26 k frames/s from page-locked buffers
        // against 44 k resident.  With one frame of work always queued behind the wait the device never runs dry.
        HIPCHK(hipEventSynchronize(g.ev_rd_depth[p]));
    }
    if (overlap) g.up_used[p] = true;
    const double si1 = si_trace ? si_now() : 0;
    if (gray) {
        uint8_t *&buf = p ? h->d_gray_stage1 : h->d_gray_stage;
        if (!buf) HIPCHK(hipMalloc((void **)&buf, S * HW));
        if (overlap) {
            if (g.have_rd_gray[p]) HIPCHK(hipStreamWaitEvent(g.copy_stream, g.ev_rd_gray[p], 0));
            HIPCHK(hipMemcpyAsync(buf + s0 * HW, gray + s0 * HW, n * HW, hipMemcpyHostToDevice, g.copy_stream));
            HIPCHK(hipEventRecord(g.ev_up_gray[p], g.copy_stream));
            HIPCHK(hipStreamWaitEvent(g.fe_stream, g.ev_up_gray[p], 0));
        } else
            HIPCHK(hipMemcpyAsync(buf + s0 * HW, gray + s0 * HW, n * HW, hipMemcpyHostToDevice, g.fe_stream));
        *dg = buf;
    }
    const double si2 = si_trace ? si_now() : 0;
    if (depth) {
        uint16_t *&buf = p ? h->d_depth_stage1 : h->d_depth_stage;
        if (!buf) HIPCHK(hipMalloc((void **)&buf, S * HW * 2));
        if (overlap) {
            hipStream_t cs2 = g.copy_stream2 ? g.copy_stream2 : g.copy_stream;
            if (g.have_rd_depth[p]) HIPCHK(hipStreamWaitEvent(cs2, g.ev_rd_depth[p], 0));
            HIPCHK(hipMemcpyAsync(buf + s0 * HW, depth + s0 * HW, n * HW * 2, hipMemcpyHostToDevice, cs2));
            HIPCHK(hipEventRecord(g.ev_up_depth[p], cs2));
            HIPCHK(hipStreamWaitEvent(g.stream, g.ev_up_depth[p], 0));
        } else
            HIPCHK(hipMemcpyAsync(buf + s0 * HW, depth + s0 * HW, n * HW * 2, hipMemcpyHostToDevice, g.stream));
        *dd = buf;
    }
    if (si_trace && !on_device) fprintf(stderr, "  stage_inputs group %d: wait for the staging buffer's last upload %.0f us, grey copy call %.0f us, depth copy call %.0f us\n", (int)s0, si1 - si0, si2 - si1, si_now() - si2);
    return VIO_OK;
}

// front-end of this frame may overlap the marginalisation of the previous one: it waits only for the previous solve
// (tracker lag 1: only for the previous be_ingest -- the tracker reads the state snapshot that kernel took, see vio_set_tracker_lag)
static int fe_wait(vio_batch *h, vio_batch::Group &g) {
    if (h->tracker_lag) { if (g.have_ingest_ev) HIPCHK(hipStreamWaitEvent(g.fe_stream, g.ev_ingest, 0)); }
    else if (g.have_solve_ev) HIPCHK(hipStreamWaitEvent(g.fe_stream, g.ev_solve, 0));
    return VIO_OK;
}
static int be_wait(vio_batch::Group &g) {
    HIPCHK(hipEventRecord(g.ev_fe, g.fe_stream));
    HIPCHK(hipStreamWaitEvent(g.stream, g.ev_fe, 0));
    return VIO_OK;
}

static const IngestSrc kTrackerMap = {nullptr, nullptr, nullptr, nullptr, 0};

// Host buffers handed to a previous non-overlap call are free again once this returns (include/vio_abi.h "Host buffers").
static int wait_host_uploads(vio_batch *h) {
    for (auto &g : h->groups) {
        if (g.host_fe_pending) { HIPCHK(hipEventSynchronize(g.ev_host_fe)); g.host_fe_pending = false; }
        if (g.host_be_pending) { HIPCHK(hipEventSynchronize(g.ev_host_be)); g.host_be_pending = false; }
    }
    return VIO_OK;
}
static int note_host_upload(vio_batch::Group &g, bool fe, bool be) {
    if (fe) {
        if (!g.ev_host_fe) HIPCHK(hipEventCreateWithFlags(&g.ev_host_fe, hipEventDisableTiming));
        HIPCHK(hipEventRecord(g.ev_host_fe, g.fe_stream)); g.host_fe_pending = true;
    }
    if (be) {
        if (!g.ev_host_be) HIPCHK(hipEventCreateWithFlags(&g.ev_host_be, hipEventDisableTiming));
        HIPCHK(hipEventRecord(g.ev_host_be, g.stream)); g.host_be_pending = true;
    }
    return VIO_OK;
}

// vio_feed's stamps / modes through the group's page-locked ring (see Group::side_ring): a slot is reused only after its copies ran
static int stage_side_ring(vio_batch *h, vio_batch::Group &g, const double *stamps, const uint8_t *modes) {
    const size_t n = (size_t)g.n, slot_bytes = n * 9;
    if (!g.side_ring) {
        HIPCHK(hipHostMalloc((void **)&g.side_ring, slot_bytes * vio_batch::Group::kSideRing, hipHostMallocDefault));
        for (int k = 0; k < vio_batch::Group::kSideRing; k++) HIPCHK(hipEventCreateWithFlags(&g.side_ev[k], hipEventDisableTiming));
    }
    const int k = g.side_pos;
    g.side_pos = (k + 1) % vio_batch::Group::kSideRing;
    if (g.side_used[k]) HIPCHK(hipEventSynchronize(g.side_ev[k]));   // sixteen calls ago: long done unless the caller runs that far ahead
    unsigned char *slot = g.side_ring + (size_t)k * slot_bytes;
    memcpy(slot, stamps + g.s0, n * sizeof(double));
    HIPCHK(hipMemcpyAsync(h->d_stamps + g.s0, slot, n * sizeof(double), hipMemcpyHostToDevice, g.fe_stream));
    if (modes) {
        memcpy(slot + n * 8, modes + g.s0, n);
        HIPCHK(hipMemcpyAsync(h->d_modes + g.s0, slot + n * 8, n, hipMemcpyHostToDevice, g.fe_stream));
    }
    HIPCHK(hipEventRecord(g.side_ev[k], g.fe_stream));
    g.side_used[k] = true;
    return VIO_OK;
}

// per-call side inputs of the front-end (frame modes, caller-supplied relative rotations): group slices, on the group's fe_stream
static int stage_side_inputs(vio_batch *h, vio_batch::Group &g, const uint8_t *modes, const double *R_rel) {
    if (modes) HIPCHK(hipMemcpyAsync(h->d_modes + g.s0, modes + g.s0, (size_t)g.n, hipMemcpyHostToDevice, g.fe_stream));
    if (R_rel) HIPCHK(hipMemcpyAsync(h->d_rrel + (size_t)g.s0 * 9, R_rel + (size_t)g.s0 * 9, (size_t)g.n * 9 * sizeof(double), hipMemcpyHostToDevice, g.fe_stream));
    return VIO_OK;
}

int vio_feed_modes(vio_batch *h, const uint8_t *gray, const uint16_t *depth_mm, const double *stamps, const uint8_t *modes, int on_device) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || !gray || !depth_mm || !stamps) return VIO_EINVAL;
    int rc = wait_host_uploads(h);   // (pending uploads of an earlier vio_track / vio_process call; vio_feed itself leaves none, see stage_side_ring)
    if (rc != VIO_OK) return rc;
    rc = refresh_dynamic_state(h);
    if (rc != VIO_OK) return rc;
    rc = flush_imu_frontend(h);
    if (rc != VIO_OK) return rc;
    // VIO_FEED_TRACE=1 (diagnostic): host time of every section of the call on stderr -- where an "asynchronous" feed blocks
    static const bool feed_trace = getenv("VIO_FEED_TRACE") && atoi(getenv("VIO_FEED_TRACE")) != 0;
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (auto &g : h->groups) {
        const double tt0 = feed_trace ? now_us() : 0;
        if ((rc = fe_wait(h, g)) != VIO_OK) return rc;
        const uint8_t *dg = nullptr;
        const uint16_t *dd = nullptr;
        rc = stage_inputs(h, g, gray, depth_mm, nullptr, on_device, &dg, &dd, /*overlap=*/true);
        if (rc != VIO_OK) return rc;
        const double tt1 = feed_trace ? now_us() : 0;
        if ((rc = stage_side_ring(h, g, stamps, modes)) != VIO_OK) return rc;
        if (g.s0 == 0) HIPCHK(hipEventRecord(h->ev[0], g.fe_stream));
        rc = launch_frontend(h, g, dg, 1, 1, modes ? h->d_modes : nullptr, nullptr);
        if (rc != VIO_OK) return rc;
        if (g.s0 == 0) HIPCHK(hipEventRecord(h->ev[1], g.fe_stream));
        if (!on_device && (rc = note_stage_read(g.ev_rd_gray[g.flip], g.have_rd_gray[g.flip], g.fe_stream)) != VIO_OK) return rc;
        if ((rc = be_wait(g)) != VIO_OK) return rc;
        rc = launch_backend(h, g, dd, kTrackerMap);
        if (rc != VIO_OK) return rc;
        if (!on_device) {
            if ((rc = note_stage_read(g.ev_rd_depth[g.flip], g.have_rd_depth[g.flip], g.stream)) != VIO_OK) return rc;
            g.flip ^= 1;
        }
        if (g.s0 == 0) HIPCHK(hipEventRecord(h->ev[2], g.stream));
        if (feed_trace) fprintf(stderr, "vio_feed group %d: stage_inputs %.0f us, rest (side ring + %s launches) %.0f us, t = %.0f\n", g.s0, tt1 - tt0, "all", now_us() - tt1, tt0);
    }
    if (h->prof_cur >= 0) h->prof_cur++;
    h->timing_valid = true;
    return VIO_OK;
}

int vio_abi_version(void) { return 6; }

// marg_exact = 2 (the literal marginalisation with a CERTIFIED first inverse): out2 = {marginalisations of sequence seq whose certificate failed
// since vio_create / vio_reset -- those frames used the block inverse WITHOUT the proof that the reference's 1e-8 cut drops nothing --, 1 if the
// last marginalisation was certified}.  A parity run asserts out2[0] == 0; a deployment that sees it grow should switch to marg_exact = 1.
int vio_get_marg_certificate(vio_batch *h, int seq, int32_t *out2) {
    DevGuard dev_guard(h);
    if (!h || !out2 || seq < 0 || seq >= h->S) return VIO_EINVAL;
    if (dev_guard.failed) { g_err = "hipSetDevice failed"; return VIO_EDEVICE; }
    HIPCHK(hipDeviceSynchronize());
    BeSeq be;
    HIPCHK(hipMemcpy(&be, h->B.be + seq, sizeof(BeSeq), hipMemcpyDeviceToHost));
    out2[0] = be.dbg[12]; out2[1] = be.dbg[11];
    return VIO_OK;
}

int vio_host_buffers_done(vio_batch *h, int calls_ago) {
    DevGuard dev_guard(h);
    if (!h || calls_ago < 0) return VIO_EINVAL;
    if (dev_guard.failed) { g_err = "hipSetDevice failed"; return VIO_EDEVICE; }
    // calls_ago counts vio_feed calls that took HOST images (on_device = 0): feeds of device-resident frames in between neither use nor wait for
    // the staging buffers.  Only the two most recent host feeds can still be uploading (the third-last was waited for when its staging
    // buffer was reused) -- and that is checked against the events, not assumed.
    if (calls_ago > 1) return 1;
    for (auto &g : h->groups) {
        if (!g.copy_stream) continue;
        const int p = (g.flip ^ 1 ^ calls_ago) & 1;     // g.flip = the buffer the NEXT host feed will use; the latest host feed used g.flip ^ 1
        if (!g.up_used[p]) continue;
        for (hipEvent_t e : {g.ev_up_gray[p], g.ev_up_depth[p]}) {
            const hipError_t q = hipEventQuery(e);
            if (q == hipErrorNotReady) return 0;
            if (q != hipSuccess) { g_err = "hipEventQuery failed"; return VIO_EDEVICE; }
        }
    }
    return 1;
}

int vio_feed(vio_batch *h, const uint8_t *gray, const uint16_t *depth_mm, const double *stamps, int on_device) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    return vio_feed_modes(h, gray, depth_mm, stamps, nullptr, on_device);
}

static int track_impl(vio_batch *h, const uint8_t *gray, const double *stamps, int publish, const uint8_t *modes, const double *R_rel, int on_device) {
    if (!h || !gray || !stamps) return VIO_EINVAL;
    int rc = wait_host_uploads(h);
    if (rc != VIO_OK) return rc;
    rc = flush_imu_frontend(h);
    if (rc != VIO_OK) return rc;
    for (auto &g : h->groups) {
        if ((rc = fe_wait(h, g)) != VIO_OK) return rc;
        HIPCHK(hipStreamWaitEvent(g.fe_stream, g.ev_be, 0));  // stand-alone use: no overlap with a pending vio_process
        const uint8_t *dg = nullptr;
        const uint16_t *dd = nullptr;
        rc = stage_inputs(h, g, gray, nullptr, stamps, on_device, &dg, &dd);
        if (rc != VIO_OK) return rc;
        if ((rc = stage_side_inputs(h, g, modes, R_rel)) != VIO_OK) return rc;
        if ((rc = note_host_upload(g, true, false)) != VIO_OK) return rc;
        rc = launch_frontend(h, g, dg, publish ? 1 : 0, 0, modes ? h->d_modes : nullptr, R_rel ? h->d_rrel : nullptr);
        if (rc != VIO_OK) return rc;
        if (!on_device && (rc = note_stage_read(g.ev_rd_gray[0], g.have_rd_gray[0], g.fe_stream)) != VIO_OK) return rc;
        if ((rc = be_wait(g)) != VIO_OK) return rc;
    }
    if (h->prof_cur >= 0) { h->prof_cur++; h->prof_fe_only = true; }
    return VIO_OK;
}

int vio_track(vio_batch *h, const uint8_t *gray, const double *stamps, int publish, int on_device) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    return track_impl(h, gray, stamps, publish, nullptr, nullptr, on_device);
}

int vio_track_ex(vio_batch *h, const uint8_t *gray, const double *stamps, const uint8_t *modes, const double *R_rel, int on_device) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    return track_impl(h, gray, stamps, 1, modes, R_rel, on_device);
}

int vio_predict_motion(vio_batch *h, int seq, double t0, double t1, double *R9) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S || !R9) return VIO_EINVAL;
    int rc = flush_imu_frontend(h);  // samples pushed so far must be in the ring (Estimator::predictMotion reads imu_buf)
    if (rc != VIO_OK) return rc;
    hipStream_t st = h->groups[0].fe_stream;
    for (auto &g : h->groups)
        if (seq >= g.s0 && seq < g.s0 + g.n) st = g.fe_stream;
    Batch Bq = h->B;
    Bq.tracker_lag = 0;   // an explicit call reads the estimator as it is now
    fe_predict_motion_kernel<<<1, 64, 0, st>>>(Bq, seq, t0, t1, h->d_r9);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(R9, h->d_r9, 9 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return VIO_OK;
}

int vio_set_fisheye_mask(vio_batch *h, const uint8_t *mask, int on_device) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h) return VIO_EINVAL;
    int rc = sync_all(h);
    if (rc != VIO_OK) return rc;
    const size_t HW = (size_t)h->hc.c.width * h->hc.c.height;
    if (!mask) { h->B.fisheye = nullptr; return VIO_OK; }
    if (!h->d_fisheye) HIPCHK(hipMalloc((void **)&h->d_fisheye, HW));
    HIPCHK(hipMemcpy(h->d_fisheye, mask, HW, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    h->B.fisheye = h->d_fisheye;
    return VIO_OK;
}

int vio_set_tracker_lag(vio_batch *h, int lag) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || (lag != 0 && lag != 1)) return VIO_EINVAL;
    if (lag && h->hc.c.dynamic_init) { g_err = "vio_set_tracker_lag: dynamic_init handles run their initialisation on the host between frames (lag 0 only)"; return VIO_EINVAL; }
    int rc = sync_all(h);
    if (rc != VIO_OK) return rc;
    if (lag != h->tracker_lag) {   // the overlapping front-end gets its own compute units (see create_group_streams)
        for (auto &g : h->groups) if ((rc = create_group_streams(h, g, lag == 1)) != VIO_OK) return rc;
        h->stream = h->groups[0].stream; h->fe_stream = h->groups[0].fe_stream;
    }
    h->tracker_lag = lag;
    h->B.tracker_lag = lag;
    return VIO_OK;
}

int vio_set_relo_frame(vio_batch *h, int seq, double frame_stamp, int frame_index, int n, const double *match_points, const double *relo_t3,
                       const double *relo_r9) {
    DevGuard dev_guard(h);
    if (!h || seq < 0 || seq >= h->S || n < 0 || (n > 0 && !match_points) || !relo_t3 || !relo_r9) return VIO_EINVAL;
    if (n > h->hc.NP) { g_err = "more match points than the tracker holds features (vio_get_capacity)"; return VIO_ECAPACITY; }
    for (int i = 1; i < n; i++)
        if (!(match_points[3 * i + 2] > match_points[3 * (i - 1) + 2])) { g_err = "match points must ascend in feature id"; return VIO_EINVAL; }
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    h->relo_frames = 1 << 30;   // relocalisation factors use the 42-double records: from now on the two-kernel solver path is launched too (launch_backend)
    double par[15];
    par[0] = frame_stamp; par[1] = frame_index; par[2] = n;
    for (int k = 0; k < 3; k++) par[3 + k] = relo_t3[k];
    for (int k = 0; k < 9; k++) par[6 + k] = relo_r9[k];
    if (n > 0) HIPCHK(hipMemcpy(h->B.relo_mp + (size_t)seq * h->hc.NP * 3, match_points, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_r9, par, sizeof(par), hipMemcpyHostToDevice));
    be_set_relo_kernel<<<1, 64, 0, h->stream>>>(h->B, seq, h->d_r9);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return VIO_OK;
}

int vio_get_relo(vio_batch *h, int seq, double *out30) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S || !out30) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    static thread_local BeSeq be;
    HIPCHK(hipMemcpy(&be, h->B.be + seq, sizeof(BeSeq), hipMemcpyDeviceToHost));
    double *o = out30;
    for (int k = 0; k < 3; k++) *o++ = be.relo_relative_t[k];
    for (int k = 0; k < 4; k++) *o++ = be.relo_relative_q[k];
    *o++ = be.relo_relative_yaw;
    for (int k = 0; k < 3; k++) *o++ = be.drift_correct_t[k];
    for (int k = 0; k < 9; k++) *o++ = be.drift_correct_r[k];
    for (int k = 0; k < 7; k++) *o++ = be.relo_Pose[k];
    *o++ = be.relo_info; *o++ = be.relo_local; *o++ = be.relo_factors;
    return VIO_OK;
}

int vio_get_latest_odometry(vio_batch *h, int seq, double *out11) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S || !out11) return VIO_EINVAL;
    int rc = flush_imu_backend(h);   // samples pushed so far must be in the ring
    if (rc != VIO_OK) return rc;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    be_latest_odometry_kernel<<<1, 64, 0, h->stream>>>(h->B, seq, h->d_r9);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out11, h->d_r9, 11 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return VIO_OK;
}

int vio_process(vio_batch *h, const uint16_t *depth_mm, int on_device) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || !depth_mm) return VIO_EINVAL;
    int rc = wait_host_uploads(h);
    if (rc != VIO_OK) return rc;
    rc = refresh_dynamic_state(h);
    if (rc != VIO_OK) return rc;
    rc = flush_imu_backend(h);
    if (rc != VIO_OK) return rc;
    for (auto &g : h->groups) {
        const uint8_t *dg = nullptr;
        const uint16_t *dd = nullptr;
        rc = stage_inputs(h, g, nullptr, depth_mm, nullptr, on_device, &dg, &dd);
        if (rc != VIO_OK) return rc;
        if (!on_device && (rc = note_host_upload(g, false, true)) != VIO_OK) return rc;
        rc = launch_backend(h, g, dd, kTrackerMap);
        if (rc != VIO_OK) return rc;
        HIPCHK(hipEventRecord(g.ev_be, g.stream));
        if ((rc = note_stage_read(g.ev_rd_depth[0], g.have_rd_depth[0], g.stream)) != VIO_OK) return rc;
    }
    return VIO_OK;
}

int vio_process_obs_batch(vio_batch *h, const int32_t *n_obs, const int32_t *ids, const double *obs, int cap, const uint16_t *depth_mm,
                          const double *stamps, int on_device) {
    DevGuard dev_guard(h);
    if (!h || !n_obs || !ids || !obs || !depth_mm || !stamps || cap < 1) return VIO_EINVAL;
    const int NP = h->hc.NP;
    for (int s = 0; s < h->S; s++)
        if (n_obs[s] > cap || n_obs[s] > NP) { g_err = "feature map larger than the tracker capacity (vio_get_capacity)"; return VIO_ECAPACITY; }
    int rc = wait_host_uploads(h);
    if (rc != VIO_OK) return rc;
    rc = refresh_dynamic_state(h);
    if (rc != VIO_OK) return rc;
    rc = flush_imu_backend(h);
    if (rc != VIO_OK) return rc;
    for (auto &g : h->groups) {
        const uint8_t *dg = nullptr;
        const uint16_t *dd = nullptr;
        rc = stage_inputs(h, g, nullptr, depth_mm, nullptr, on_device, &dg, &dd);
        if (rc != VIO_OK) return rc;
        HIPCHK(hipMemcpyAsync(h->d_in_n + g.s0, n_obs + g.s0, (size_t)g.n * sizeof(int), hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipMemcpyAsync(h->d_in_stamps + g.s0, stamps + g.s0, (size_t)g.n * sizeof(double), hipMemcpyHostToDevice, g.stream));
        for (int s = g.s0; s < g.s0 + g.n; s++) {
            if (n_obs[s] <= 0) continue;
            HIPCHK(hipMemcpyAsync(h->d_in_ids + (size_t)s * NP, ids + (size_t)s * cap, (size_t)n_obs[s] * sizeof(int), hipMemcpyHostToDevice, g.stream));
            HIPCHK(hipMemcpyAsync(h->d_in_obs + (size_t)s * NP * 7, obs + (size_t)s * cap * 7, (size_t)n_obs[s] * 7 * sizeof(double), hipMemcpyHostToDevice, g.stream));
        }
        if ((rc = note_host_upload(g, false, true)) != VIO_OK) return rc;
        IngestSrc src = {h->d_in_n, h->d_in_ids, h->d_in_obs, h->d_in_stamps, NP};
        rc = launch_backend(h, g, dd, src);
        if (rc != VIO_OK) return rc;
        HIPCHK(hipEventRecord(g.ev_be, g.stream));
        if ((rc = note_stage_read(g.ev_rd_depth[0], g.have_rd_depth[0], g.stream)) != VIO_OK) return rc;
    }
    return VIO_OK;
}

int vio_process_obs(vio_batch *h, int seq, int n, const int32_t *ids, const double *obs, const uint16_t *depth_mm, double stamp) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S || n < 0 || (n > 0 && (!ids || !obs)) || !depth_mm) return VIO_EINVAL;
    if (n == 0) return VIO_OK;  // the nodelet only queues non-empty maps (estimator_nodelet.cpp:378)
    const DevCfg &C = h->hc;
    const int NP = C.NP;
    if (n > NP) { g_err = "feature map larger than the tracker capacity (vio_get_capacity)"; return VIO_ECAPACITY; }
    int rc = wait_host_uploads(h);
    if (rc != VIO_OK) return rc;
    rc = flush_imu_backend(h);
    if (rc != VIO_OK) return rc;
    vio_batch::Group *gp = &h->groups[0];
    for (auto &g : h->groups)
        if (seq >= g.s0 && seq < g.s0 + g.n) gp = &g;
    vio_batch::Group &g = *gp;
    const size_t HW = (size_t)C.c.width * C.c.height;
    if (!h->d_depth_stage) HIPCHK(hipMalloc((void **)&h->d_depth_stage, (size_t)h->S * HW * 2));
    HIPCHK(hipMemcpyAsync(h->d_depth_stage + (size_t)seq * HW, depth_mm, HW * 2, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(h->d_in_n + seq, &n, sizeof(int), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(h->d_in_stamps + seq, &stamp, sizeof(double), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(h->d_in_ids + (size_t)seq * NP, ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(h->d_in_obs + (size_t)seq * NP * 7, obs, (size_t)n * 7 * sizeof(double), hipMemcpyHostToDevice, g.stream));
    if ((rc = note_host_upload(g, false, true)) != VIO_OK) return rc;
    IngestSrc src = {h->d_in_n, h->d_in_ids, h->d_in_obs, h->d_in_stamps, NP};
    rc = launch_backend(h, g, h->d_depth_stage, src, seq);
    if (rc != VIO_OK) return rc;
    HIPCHK(hipEventRecord(g.ev_be, g.stream));
    if ((rc = note_stage_read(g.ev_rd_depth[0], g.have_rd_depth[0], g.stream)) != VIO_OK) return rc;
    return VIO_OK;
}

int vio_get_packaged(vio_batch *h, int seq, int cap, int32_t *ids, double *obs) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    static thread_local FeSeq fe;
    HIPCHK(hipMemcpy(&fe, h->B.fe + seq, sizeof(FeSeq), hipMemcpyDeviceToHost));
    if (fe.n_forw < 0 || !fe.publish_ok) return 0;
    int n = fe.n_obs, m = n < cap ? n : cap;
    if (m > 0 && ids) HIPCHK(hipMemcpy(ids, h->B.obs_id + (size_t)seq * h->hc.NP, sizeof(int) * m, hipMemcpyDeviceToHost));
    if (m > 0 && obs) HIPCHK(hipMemcpy(obs, h->B.obs + (size_t)seq * h->hc.NP * 7, sizeof(double) * 7 * m, hipMemcpyDeviceToHost));
    return n;
}

// ---- HBM buffers for callers without their own HIP binding (the on_device = 1 paths take plain device addresses)
void *vio_device_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { g_err = "hipMalloc failed"; return nullptr; }
    return p;
}
void vio_device_free(void *p) { if (p) (void)hipFree(p); }
void *vio_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { g_err = "hipHostMalloc failed"; return nullptr; }
    return p;
}
void vio_host_free(void *p) { if (p) (void)hipHostFree(p); }
int vio_device_upload(void *dst, const void *src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return VIO_OK; }
int vio_device_download(void *dst, const void *src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return VIO_OK; }

int vio_abi_sizeof(int what) { return what == 0 ? (int)sizeof(vio_config) : (what == 1 ? (int)sizeof(vio_status) : -1); }

int vio_get_capacity(vio_batch *h, int32_t *out3) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || !out3) return VIO_EINVAL;
    out3[0] = h->hc.NP; out3[1] = h->hc.NL; out3[2] = h->hc.NIMU;
    return VIO_OK;
}

// which solver the handle runs: 0 = persistent kernel (round-1 fallback), 1 = phased with the Schur complement in LDS tiles, 2 = phased
// with the Schur complement in HBM / L2 (windows beyond W = 10)
int vio_get_solver_kind(vio_batch *h) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h) return VIO_EINVAL;
    return h->solve_mode == 0 ? 0 : (h->serial_big ? 2 : 1);
}

// bounds-constrained solves of sequence seq since vio_create / vio_reset (estimator.cpp:1282-1297): out4 = inverse depths cut by the bound
// while a point was formed, bounded landmarks that entered solves, trial evaluations and shortened steps of the projected Armijo line search
int vio_get_bound_stats(vio_batch *h, int seq, int64_t *out4) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || !out4 || seq < 0 || seq >= h->S) return VIO_EINVAL;
    HIPCHK(hipDeviceSynchronize());
    BeSeq be;
    HIPCHK(hipMemcpy(&be, h->B.be + seq, sizeof(BeSeq), hipMemcpyDeviceToHost));
    out4[0] = be.bound_clamps; out4[1] = be.bounded_solves; out4[2] = be.ls_evals; out4[3] = be.ls_contractions;
    return VIO_OK;
}

int vio_push_imu_batch(vio_batch *h, const int32_t *n, int stride, const double *t, const double *acc, const double *gyr) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || stride < 0 || !t || !acc || !gyr) return VIO_EINVAL;
    std::lock_guard<std::mutex> lk(h->imu_mu);
    for (int s = 0; s < h->S; s++) {
        const int ns = n ? n[s] : stride;
        if (ns < 0 || ns > stride) return VIO_EINVAL;
        for (int i = 0; i < ns; i++) {
            const size_t q = (size_t)s * stride + i;
            if (!(t[q] > h->last_imu_t[s])) continue;  // "imu message in disorder" (estimator_nodelet.cpp:110-114)
            h->last_imu_t[s] = t[q];
            h->p_seq.push_back(s);
            h->p_t.push_back(t[q]);
            for (int k = 0; k < 3; k++) { h->p_acc.push_back(acc[3 * q + k]); h->p_gyr.push_back(gyr[3 * q + k]); }
        }
    }
    return VIO_OK;
}

int vio_sync(vio_batch *h) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h) return VIO_EINVAL;
    return sync_all(h);
}
void *vio_get_stream(vio_batch *h) { DevGuard dev_guard(h); return h ? (void *)h->stream : nullptr; }

static void fill_status(const BeSeq &be, const FeSeq &fe, vio_status *out) {
    const int ovf = be.overflow | fe.overflow;
    out->code = (ovf && be.status_code == VIO_OK) ? VIO_ECAPACITY : be.status_code;  // a table overflowed in the last frame: results are truncated, say so
    out->overflow_flags = ovf; out->overflow_frames = be.overflow_frames;
    out->iterations_total = be.iter_total; out->solves_total = be.solve_total;
    out->solver_flag = be.solver_flag; out->frame_count = be.frame_count;
    out->marginalization_flag = be.marginalization_flag; out->n_landmarks = be.n_lm; out->last_track_num = be.last_track_num;
    out->n_tracks = fe.n_pts; out->processed = be.processed; out->iterations = be.iterations; out->successful_steps = be.successful;
    out->n_in_problem = be.n_in_problem; out->n_residuals = be.n_residuals; out->n_var_landmarks = be.n_var_landmarks;
    out->has_prior = be.has_prior; out->reboot_count = be.reboot_count; out->frames_processed = be.frames_processed;
    out->initial_cost = be.initial_cost; out->final_cost = be.final_cost; out->td = be.td;
}

int vio_get_status(vio_batch *h, int seq, vio_status *out) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S || !out) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    static thread_local BeSeq be;
    static thread_local FeSeq fe;
    HIPCHK(hipMemcpy(&be, h->B.be + seq, sizeof(BeSeq), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&fe, h->B.fe + seq, sizeof(FeSeq), hipMemcpyDeviceToHost));
    fill_status(be, fe, out);
    return VIO_OK;
}

int vio_get_status_all(vio_batch *h, vio_status *out) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || !out) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    std::vector<BeSeq> be((size_t)h->S);
    std::vector<FeSeq> fe((size_t)h->S);
    HIPCHK(hipMemcpy(be.data(), h->B.be, sizeof(BeSeq) * (size_t)h->S, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fe.data(), h->B.fe, sizeof(FeSeq) * (size_t)h->S, hipMemcpyDeviceToHost));
    for (int s = 0; s < h->S; s++) fill_status(be[s], fe[s], out + s);
    return VIO_OK;
}

int vio_get_window(vio_batch *h, int seq, double *out) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S || !out) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    static thread_local BeSeq be;
    HIPCHK(hipMemcpy(&be, h->B.be + seq, sizeof(BeSeq), hipMemcpyDeviceToHost));
    for (int i = 0; i <= h->hc.W; i++) {
        double *o = out + 17 * i;
        dm::quat q = dm::R2q(dm::ldm(be.Rs[i]));
        o[0] = be.Ps[i][0]; o[1] = be.Ps[i][1]; o[2] = be.Ps[i][2];
        o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z;
        for (int k = 0; k < 3; k++) { o[7 + k] = be.Vs[i][k]; o[10 + k] = be.Bas[i][k]; o[13 + k] = be.Bgs[i][k]; }
        o[16] = be.Headers[i];
    }
    return VIO_OK;
}

int vio_get_odometry(vio_batch *h, double *out) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || !out) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    HIPCHK(hipMemcpy(out, h->B.odom, sizeof(double) * (size_t)h->S * 11, hipMemcpyDeviceToHost));
    return VIO_OK;
}

int vio_get_odometry_history(vio_batch *h, int seq, int cap, double *out) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S || !out) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    int n = 0;
    HIPCHK(hipMemcpy(&n, h->B.odom_count + seq, sizeof(int), hipMemcpyDeviceToHost));
    const int hc = h->B.hist_cap;
    int m = std::min(std::min(n, cap), hc);  // the most recent m rows, oldest first
    const double *base = h->B.odom_hist + (size_t)seq * hc * 11;
    for (int done = 0; done < m;) {
        int row = (n - m + done) % hc, run = std::min(m - done, hc - row);
        HIPCHK(hipMemcpy(out + (size_t)done * 11, base + (size_t)row * 11, sizeof(double) * (size_t)run * 11, hipMemcpyDeviceToHost));
        done += run;
    }
    return n;
}

int vio_get_extrinsic(vio_batch *h, int seq, double *out13) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S || !out13) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    static thread_local BeSeq be;
    HIPCHK(hipMemcpy(&be, h->B.be + seq, sizeof(BeSeq), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; k++) out13[k] = be.tic[k];
    for (int k = 0; k < 9; k++) out13[3 + k] = be.ric[k];
    out13[12] = be.td;
    return VIO_OK;
}

int vio_get_tracks(vio_batch *h, int seq, int cap, int32_t *ids, int32_t *cnt, float *cur, float *un, float *vel) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    static thread_local FeSeq fe;
    HIPCHK(hipMemcpy(&fe, h->B.fe + seq, sizeof(FeSeq), hipMemcpyDeviceToHost));
    int n = fe.n_pts, m = n < cap ? n : cap;
    size_t o = (size_t)seq * h->hc.NP;
    if (m > 0) {
        HIPCHK(hipMemcpy(ids, h->B.ids + o, sizeof(int) * m, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(cnt, h->B.track_cnt + o, sizeof(int) * m, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(cur, h->B.cur_pts + o, sizeof(float2) * m, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(un, h->B.cur_un_pts + o, sizeof(float2) * m, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(vel, h->B.pts_velocity + o, sizeof(float2) * m, hipMemcpyDeviceToHost));
    }
    return n;
}

static int get_landmarks_impl(vio_batch *h, int seq, int cap, double *out, int width) {
    if (!h || seq < 0 || seq >= h->S) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    static thread_local BeSeq be;
    HIPCHK(hipMemcpy(&be, h->B.be + seq, sizeof(BeSeq), hipMemcpyDeviceToHost));
    const int NL = h->hc.NL, n = be.n_lm, W1 = h->hc.W + 1;
    const size_t o = (size_t)seq * NL;
    // the seven scalar tables are adjacent allocations but not one buffer: one copy each, then (ex only) the observation rows
    std::vector<int> tab((size_t)7 * NL);
    int *order = tab.data(), *id = order + NL, *st = id + NL, *no = st + NL, *ef = no + NL, *sf = ef + NL, *dy = sf + NL;
    std::vector<double> dep(NL);
    HIPCHK(hipMemcpy(order, h->B.lm_order + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(id, h->B.lm_id + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(st, h->B.lm_start + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(no, h->B.lm_nobs + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ef, h->B.lm_est_flag + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sf, h->B.lm_solve_flag + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dy, h->B.lm_dyn + o, sizeof(int) * NL, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dep.data(), h->B.lm_depth + o, sizeof(double) * NL, hipMemcpyDeviceToHost));
    std::vector<double> obs;
    if (width > 7) {
        obs.resize((size_t)NL * W1 * VIO_OBS_D);
        HIPCHK(hipMemcpy(obs.data(), h->B.lm_obs + o * W1 * VIO_OBS_D, sizeof(double) * obs.size(), hipMemcpyDeviceToHost));
    }
    for (int k = 0; k < n && k < cap; k++) {
        int s = order[k];
        double *q = out + (size_t)width * k;
        q[0] = id[s]; q[1] = st[s]; q[2] = no[s]; q[3] = dep[s]; q[4] = ef[s]; q[5] = sf[s]; q[6] = dy[s];
        if (width > 7) {
            // feature_per_frame[0] / .back(): the observation rows are ring-indexed per frame (vio_state.h lm_obs)
            const double *f0 = &obs[((size_t)s * W1 + (st[s] + be.ring_base) % W1) * VIO_OBS_D];
            const double *fb = &obs[((size_t)s * W1 + (st[s] + std::max(no[s], 1) - 1 + be.ring_base) % W1) * VIO_OBS_D];
            q[7] = f0[0]; q[8] = f0[1]; q[9] = f0[2]; q[10] = f0[8]; q[11] = fb[8];
        }
    }
    return n;
}
int vio_get_landmarks(vio_batch *h, int seq, int cap, double *out) { DevGuard dev_guard(h); return get_landmarks_impl(h, seq, cap, out, 7); }
int vio_get_landmarks_ex(vio_batch *h, int seq, int cap, double *out12) { DevGuard dev_guard(h); return get_landmarks_impl(h, seq, cap, out12, 12); }

int vio_get_prior(vio_batch *h, int seq, double *J, double *r, double *x0, uint8_t *present) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    static thread_local BeSeq be;
    HIPCHK(hipMemcpy(&be, h->B.be + seq, sizeof(BeSeq), hipMemcpyDeviceToHost));
    if (!be.has_prior) return 0;
    int n = h->hc.NPRIOR, W = h->hc.W;
    if (J || r) {
        // the hot path keeps the prior as a quadratic form (DESIGN.md deviation 13); the factored form the reference stores
        // (linearized_jacobians / linearized_residuals) is produced here, on the GPU, only when somebody asks for it
        be_prior_factor_kernel<<<1, 512, h->lds_factor, h->stream>>>(h->B, seq);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (J) HIPCHK(hipMemcpy(J, h->B.prior_J + (size_t)seq * n * n, sizeof(double) * n * n, hipMemcpyDeviceToHost));
    if (r) HIPCHK(hipMemcpy(r, h->B.prior_rf + (size_t)seq * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (x0) HIPCHK(hipMemcpy(x0, h->B.prior_x0 + (size_t)seq * (W * 7 + 17), sizeof(double) * (W * 7 + 17), hipMemcpyDeviceToHost));
    if (present) for (int k = 0; k < W + 3; k++) present[k] = (uint8_t)be.prior_present[k];
    return n;
}

int vio_get_timings(vio_batch *h, int cap, double *out_ms) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || !out_ms || cap < 3) return VIO_EINVAL;
    if (!h->timing_valid) return 0;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    float a = 0, b = 0;
    HIPCHK(hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
    HIPCHK(hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
    out_ms[0] = a; out_ms[1] = b; out_ms[2] = a + b;
    return 3;
}

int vio_debug_seq(vio_batch *h, int seq, int *out16) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || seq < 0 || seq >= h->S) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    static thread_local BeSeq be;
    HIPCHK(hipMemcpy(&be, h->B.be + seq, sizeof(BeSeq), hipMemcpyDeviceToHost));
    for (int k = 0; k < 16; k++) out16[k] = be.dbg[k];
    return VIO_OK;
}

// debug: accumulated in-kernel phase ticks (100 MHz) of sequence 0; reset != 0 clears them
int vio_debug_phases(vio_batch *h, float *out128, int reset) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    if (out128) HIPCHK(hipMemcpy(out128, h->B.timings, 128 * sizeof(float), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(hipMemset(h->B.timings, 0, 128 * sizeof(float)));
    return VIO_OK;
}

// debug: per-sequence in-kernel durations (100 MHz ticks) of the last frame's fe_select / fe_add: out[S][4]
int vio_debug_fe_ticks(vio_batch *h, float *out) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || !out) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    HIPCHK(hipMemcpy(out, h->B.fe_ticks, (size_t)h->S * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return VIO_OK;
}

// per-kernel HIP-event profile of the next max_steps vio_feed calls (events sit on the batch stream)
int vio_profile_begin(vio_batch *h, int max_steps) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || max_steps < 1) return VIO_EINVAL;
    size_t need = (size_t)max_steps * VIO_NEV;
    while (h->pev.size() < need) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        h->pev.push_back(e);
    }
    h->prof_steps = max_steps;
    h->prof_cur = 0;
    h->prof_fe_only = false;
    return VIO_OK;
}
// out_ms[k] = average duration of kernel k over the recorded steps (ms); returns the number of recorded steps
int vio_profile_end(vio_batch *h, int cap, double *out_ms) {
    DevGuard dev_guard(h);
    if (dev_guard.failed) { g_err = "hipSetDevice failed for the handle's device"; return VIO_EDEVICE; }
    if (!h || !out_ms || cap < VIO_NK) return VIO_EINVAL;
    { int rc_ = sync_all(h); if (rc_ != VIO_OK) return rc_; }
    int n = h->prof_cur < 0 ? 0 : (h->prof_cur < h->prof_steps ? h->prof_cur : h->prof_steps);
    static const int e0[VIO_NK] = {0, 1, 2, 3, 4, 5, 6, 8, 9, 10}, e1[VIO_NK] = {1, 2, 3, 4, 5, 6, 7, 9, 10, 11};
    for (int k = 0; k < VIO_NK; k++) out_ms[k] = 0;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < (h->prof_fe_only ? 7 : VIO_NK); k++) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, h->pev[(size_t)i * VIO_NEV + e0[k]], h->pev[(size_t)i * VIO_NEV + e1[k]]));
            out_ms[k] += ms;
        }
    for (int k = 0; k < VIO_NK; k++) out_ms[k] = n > 0 ? out_ms[k] / n : 0;
    h->prof_cur = -1;
    return n;
}

// ------------------------------------------------------------------------------------------------ stage entry points
#define STAGE_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); rc = VIO_EDEVICE; goto done; } } while (0)

int vio_stage_pyr_down(const uint8_t *src, int w, int h, uint8_t *dst) {
    int rc = VIO_OK, dw = (w + 1) / 2, dh = (h + 1) / 2;
    uint8_t *ds = nullptr, *dd = nullptr;
    STAGE_CHK(hipMalloc((void **)&ds, (size_t)w * h));
    STAGE_CHK(hipMalloc((void **)&dd, (size_t)dw * dh));
    STAGE_CHK(hipMemcpy(ds, src, (size_t)w * h, hipMemcpyHostToDevice));
    fe_pyrdown_stage_kernel<<<dim3((dw + 63) / 64, (dh + 15) / 16), 256>>>(ds, w, h, dd);
    STAGE_CHK(hipDeviceSynchronize());
    STAGE_CHK(hipMemcpy(dst, dd, (size_t)dw * dh, hipMemcpyDeviceToHost));
done:
    if (ds) (void)hipFree(ds);
    if (dd) (void)hipFree(dd);
    return rc;
}

int vio_stage_clahe(const uint8_t *src, int w, int h, uint8_t *dst) {
    int rc = VIO_OK;
    uint8_t *ds = nullptr, *dd = nullptr, *dl = nullptr;
    if (!src || !dst || w < 8 || h < 8) return VIO_EINVAL;
    STAGE_CHK(hipMalloc((void **)&ds, (size_t)w * h));
    STAGE_CHK(hipMalloc((void **)&dd, (size_t)w * h));
    STAGE_CHK(hipMalloc((void **)&dl, 64 * 256));
    STAGE_CHK(hipMemcpy(ds, src, (size_t)w * h, hipMemcpyHostToDevice));
    fe_clahe_lut_stage_kernel<<<64, 256>>>(ds, w, h, dl);
    fe_clahe_apply_stage_kernel<<<dim3((w + 1023) / 1024, h), 256>>>(ds, w, h, dl, dd);
    STAGE_CHK(hipDeviceSynchronize());
    STAGE_CHK(hipMemcpy(dst, dd, (size_t)w * h, hipMemcpyDeviceToHost));
done:
    if (ds) (void)hipFree(ds);
    if (dd) (void)hipFree(dd);
    if (dl) (void)hipFree(dl);
    return rc;
}

int vio_stage_fast_roi(const uint8_t *img, int W, int H, int rx, int ry, int rw, int rh, int cap, float *out) {
    int rc = VIO_OK, count = 0;
    uint8_t *di = nullptr;
    uint32_t *dout = nullptr;
    int *dcnt = nullptr;
    std::vector<uint32_t> hv;
    GridRect r{rx, ry, rw, rh};
    if (rw < 7 || rh < 7) return 0;
    size_t lds = fast_lds_bytes(rw, rh);
    STAGE_CHK(hipMalloc((void **)&di, (size_t)W * H));
    STAGE_CHK(hipMalloc((void **)&dout, (size_t)cap * 4));
    STAGE_CHK(hipMalloc((void **)&dcnt, 4));
    STAGE_CHK(hipMemcpy(di, img, (size_t)W * H, hipMemcpyHostToDevice));
    (void)raise_lds_limit((const void *)fe_fast_stage_kernel, (size_t)(lds));
    fe_fast_stage_kernel<<<1, 256, lds>>>(di, W, r, dout, cap, dcnt);
    STAGE_CHK(hipDeviceSynchronize());
    STAGE_CHK(hipMemcpy(&count, dcnt, 4, hipMemcpyDeviceToHost));
    hv.resize(cap);
    STAGE_CHK(hipMemcpy(hv.data(), dout, (size_t)cap * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < count && i < cap; i++) { out[3 * i] = (float)(hv[i] & 0xFFF); out[3 * i + 1] = (float)((hv[i] >> 12) & 0xFFF); out[3 * i + 2] = (float)(hv[i] >> 24); }
    rc = count;
done:
    if (di) (void)hipFree(di);
    if (dout) (void)hipFree(dout);
    if (dcnt) (void)hipFree(dcnt);
    return rc;
}

int vio_stage_lk(const uint8_t *prev, const uint8_t *next, int w, int h, int max_level, int n, const float *prev_pts, float *next_pts,
                 uint8_t *status) {
    int rc = VIO_OK;
    if (max_level < 0 || max_level > 3 || n < 0) return VIO_EINVAL;
    uint8_t *dp[4] = {0, 0, 0, 0}, *dn[4] = {0, 0, 0, 0}, *dst = nullptr;
    float2 *dpp = nullptr, *dnp = nullptr;
    LkImages im;
    memset(&im, 0, sizeof(im));
    int lw = w, lh = h;
    for (int l = 0; l <= max_level; l++) {
        STAGE_CHK(hipMalloc((void **)&dp[l], (size_t)lw * lh));
        STAGE_CHK(hipMalloc((void **)&dn[l], (size_t)lw * lh));
        im.prev[l] = dp[l]; im.next[l] = dn[l]; im.w[l] = lw; im.h[l] = lh;
        if (l == 0) {
            STAGE_CHK(hipMemcpy(dp[0], prev, (size_t)w * h, hipMemcpyHostToDevice));
            STAGE_CHK(hipMemcpy(dn[0], next, (size_t)w * h, hipMemcpyHostToDevice));
        } else {
            int pw = im.w[l - 1], ph = im.h[l - 1];
            fe_pyrdown_stage_kernel<<<dim3((lw + 63) / 64, (lh + 15) / 16), 256>>>(dp[l - 1], pw, ph, dp[l]);
            fe_pyrdown_stage_kernel<<<dim3((lw + 63) / 64, (lh + 15) / 16), 256>>>(dn[l - 1], pw, ph, dn[l]);
        }
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
    }
    STAGE_CHK(hipMalloc((void **)&dpp, sizeof(float2) * (n + 1)));
    STAGE_CHK(hipMalloc((void **)&dnp, sizeof(float2) * (n + 1)));
    STAGE_CHK(hipMalloc((void **)&dst, n + 1));
    STAGE_CHK(hipMemcpy(dpp, prev_pts, sizeof(float2) * n, hipMemcpyHostToDevice));
    STAGE_CHK(hipMemcpy(dnp, next_pts, sizeof(float2) * n, hipMemcpyHostToDevice));
    if (n > 0) fe_lk_stage_kernel<<<n, 64>>>(im, max_level, n, dpp, dnp, dst);
    STAGE_CHK(hipDeviceSynchronize());
    STAGE_CHK(hipMemcpy(next_pts, dnp, sizeof(float2) * n, hipMemcpyDeviceToHost));
    STAGE_CHK(hipMemcpy(status, dst, n, hipMemcpyDeviceToHost));
done:
    for (int l = 0; l < 4; l++) { if (dp[l]) (void)hipFree(dp[l]); if (dn[l]) (void)hipFree(dn[l]); }
    if (dpp) (void)hipFree(dpp);
    if (dnp) (void)hipFree(dnp);
    if (dst) (void)hipFree(dst);
    return rc;
}

int vio_stage_ransac(const vio_config *cfg, int n, const float *p1, const float *p2, uint8_t *status) {
    int rc = VIO_OK;
    float2 *d1 = nullptr, *d2 = nullptr;
    uint8_t *ds = nullptr;
    size_t lds = (size_t)n * 36 + 64;
    STAGE_CHK(hipMalloc((void **)&d1, sizeof(float2) * (n + 1)));
    STAGE_CHK(hipMalloc((void **)&d2, sizeof(float2) * (n + 1)));
    STAGE_CHK(hipMalloc((void **)&ds, n + 1));
    STAGE_CHK(hipMemcpy(d1, p1, sizeof(float2) * n, hipMemcpyHostToDevice));
    STAGE_CHK(hipMemcpy(d2, p2, sizeof(float2) * n, hipMemcpyHostToDevice));
    (void)raise_lds_limit((const void *)fe_ransac_stage_kernel, (size_t)(lds));
    fe_ransac_stage_kernel<<<1, 256, lds>>>(*cfg, n, d1, d2, ds);
    STAGE_CHK(hipDeviceSynchronize());
    STAGE_CHK(hipMemcpy(status, ds, n, hipMemcpyDeviceToHost));
done:
    if (d1) (void)hipFree(d1);
    if (d2) (void)hipFree(d2);
    if (ds) (void)hipFree(ds);
    return rc;
}

static int stage_imu_impl(const vio_config *cfg, int n, const double *dt, const double *acc, const double *gyr, const double *acc0,
                          const double *gyr0, const double *ba, const double *bg, const double *pose_i, const double *sb_i,
                          const double *pose_j, const double *sb_j, double *preint_out, double *r15, double *J480, double *G961) {
    int rc = VIO_OK;
    double *dG = nullptr;
    PreInt *hp = new PreInt();
    PreInt *dp = nullptr;
    double *dbuf = nullptr;
    std::vector<double> hb;
    memset(hp, 0, sizeof(PreInt));
    {
        using namespace dm;
        // IntegrationBase constructor on the host side of the test harness (plain state initialisation)
        for (int k = 0; k < 3; k++) { hp->lin_acc[k] = acc0[k]; hp->lin_gyr[k] = gyr0[k]; hp->lin_ba[k] = ba[k]; hp->lin_bg[k] = bg[k]; hp->acc0[k] = acc0[k]; hp->gyr0[k] = gyr0[k]; }
        hp->dq[0] = 1;
        for (int i = 0; i < 15; i++) hp->jac[i * 16] = 1;
        hp->valid = 1;
    }
    size_t nd = (size_t)n * 7 + 32 + 461 + 15 + 480;
    hb.assign(nd, 0.0);
    for (int i = 0; i < n; i++) { hb[i] = dt[i]; for (int k = 0; k < 3; k++) { hb[n + 3 * i + k] = acc[3 * i + k]; hb[4 * n + 3 * i + k] = gyr[3 * i + k]; } }
    for (int k = 0; k < 7; k++) { hb[7 * n + k] = pose_i[k]; hb[7 * n + 16 + k] = pose_j[k]; }
    for (int k = 0; k < 9; k++) { hb[7 * n + 7 + k] = sb_i[k]; hb[7 * n + 23 + k] = sb_j[k]; }
    STAGE_CHK(hipMalloc((void **)&dp, sizeof(PreInt)));
    STAGE_CHK(hipMalloc((void **)&dbuf, nd * sizeof(double)));
    STAGE_CHK(hipMemcpy(dp, hp, sizeof(PreInt), hipMemcpyHostToDevice));
    STAGE_CHK(hipMemcpy(dbuf, hb.data(), nd * sizeof(double), hipMemcpyHostToDevice));
    be_stage_imu_kernel<<<1, 256>>>(*cfg, dp, n, dbuf, dbuf + n, dbuf + 4 * n, dbuf + 7 * n, cfg->g_norm, dbuf + 7 * n + 32, dbuf + 7 * n + 32 + 461,
                                    dbuf + 7 * n + 32 + 461 + 15);
    STAGE_CHK(hipDeviceSynchronize());
    STAGE_CHK(hipMemcpy(hb.data(), dbuf, nd * sizeof(double), hipMemcpyDeviceToHost));
    if (preint_out) memcpy(preint_out, &hb[7 * n + 32], 461 * sizeof(double));
    if (r15) memcpy(r15, &hb[7 * n + 32 + 461], 15 * sizeof(double));
    if (J480) memcpy(J480, &hb[7 * n + 32 + 461 + 15], 480 * sizeof(double));
    if (G961) {
        STAGE_CHK(hipMalloc((void **)&dG, 961 * sizeof(double)));
        STAGE_CHK(hipMemset(dG, 0, 961 * sizeof(double)));
        be_stage_imu_block_kernel<<<1, 64>>>(dp, dbuf + 7 * n, cfg->g_norm, dG);
        STAGE_CHK(hipDeviceSynchronize());
        STAGE_CHK(hipMemcpy(G961, dG, 961 * sizeof(double), hipMemcpyDeviceToHost));
    }
done:
    if (dG) (void)hipFree(dG);
    if (dp) (void)hipFree(dp);
    if (dbuf) (void)hipFree(dbuf);
    delete hp;
    return rc;
}

int vio_stage_imu_factor(const vio_config *cfg, int n, const double *dt, const double *acc, const double *gyr, const double *acc0,
                         const double *gyr0, const double *ba, const double *bg, const double *pose_i, const double *sb_i,
                         const double *pose_j, const double *sb_j, double *preint_out, double *r15, double *J480) {
    return stage_imu_impl(cfg, n, dt, acc, gyr, acc0, gyr0, ba, bg, pose_i, sb_i, pose_j, sb_j, preint_out, r15, J480, nullptr);
}
int vio_stage_imu_block(const vio_config *cfg, int n, const double *dt, const double *acc, const double *gyr, const double *acc0,
                        const double *gyr0, const double *ba, const double *bg, const double *pose_i, const double *sb_i,
                        const double *pose_j, const double *sb_j, double *G961) {
    return stage_imu_impl(cfg, n, dt, acc, gyr, acc0, gyr0, ba, bg, pose_i, sb_i, pose_j, sb_j, nullptr, nullptr, nullptr, G961);
}

static int stage_projection_impl(const vio_config *cfg, const double *pose_i, const double *pose_j, const double *ex, double inv_dep, double td,
                                 const double *obs_i, const double *obs_j, int use_td, int form, double *r2, double *J46) {
    int rc = VIO_OK;
    double hb[41 + 2 + 46];
    double *db = nullptr;
    memcpy(hb, pose_i, 56); memcpy(hb + 7, pose_j, 56); memcpy(hb + 14, ex, 56);
    hb[21] = inv_dep; hb[22] = td;
    memcpy(hb + 23, obs_i, 72); memcpy(hb + 32, obs_j, 72);
    STAGE_CHK(hipMalloc((void **)&db, sizeof(hb)));
    STAGE_CHK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
    be_stage_projection_kernel<<<1, 64>>>(*cfg, db, use_td, form, db + 41, db + 43);
    STAGE_CHK(hipDeviceSynchronize());
    STAGE_CHK(hipMemcpy(hb, db, sizeof(hb), hipMemcpyDeviceToHost));
    memcpy(r2, hb + 41, 16);
    memcpy(J46, hb + 43, 46 * 8);
done:
    if (db) (void)hipFree(db);
    return rc;
}
// cv::solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess = 1) with K = I as FeatureManager::solvePoseByPnP calls it (feature_manager.cpp:571):
// obj[n][3], img[n][2] (both rounded to float like cv::Point3f / Point2f), rvec / tvec in and out
int vio_stage_pnp(int n, const double *obj, const double *img, double *rvec3, double *tvec3) {
    int rc = VIO_OK;
    if (n < 4) return VIO_EINVAL;
    std::vector<double> pts((size_t)n * 5), par(6);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) pts[5 * i + k] = (double)(float)obj[3 * i + k];
        for (int k = 0; k < 2; k++) pts[5 * i + 3 + k] = (double)(float)img[2 * i + k];
    }
    for (int k = 0; k < 3; k++) { par[k] = rvec3[k]; par[3 + k] = tvec3[k]; }
    double *dp = nullptr, *dq = nullptr;
    STAGE_CHK(hipMalloc((void **)&dp, pts.size() * sizeof(double)));
    STAGE_CHK(hipMalloc((void **)&dq, 6 * sizeof(double)));
    STAGE_CHK(hipMemcpy(dp, pts.data(), pts.size() * sizeof(double), hipMemcpyHostToDevice));
    STAGE_CHK(hipMemcpy(dq, par.data(), 6 * sizeof(double), hipMemcpyHostToDevice));
    be_stage_pnp_kernel<<<1, 256>>>(dp, n, dq);
    STAGE_CHK(hipDeviceSynchronize());
    STAGE_CHK(hipMemcpy(par.data(), dq, 6 * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; k++) { rvec3[k] = par[k]; tvec3[k] = par[3 + k]; }
done:
    if (dp) (void)hipFree(dp);
    if (dq) (void)hipFree(dq);
    return rc;
}

int vio_stage_projection(const vio_config *cfg, const double *pose_i, const double *pose_j, const double *ex, double inv_dep, double td,
                         const double *obs_i, const double *obs_j, int use_td, double *r2, double *J46) {
    return stage_projection_impl(cfg, pose_i, pose_j, ex, inv_dep, td, obs_i, obs_j, use_td, 0, r2, J46);
}
int vio_stage_projection_residual(const vio_config *cfg, const double *pose_i, const double *pose_j, const double *ex, double inv_dep, double td,
                                  const double *obs_i, const double *obs_j, int use_td, double *r2, double *J46) {
    return stage_projection_impl(cfg, pose_i, pose_j, ex, inv_dep, td, obs_i, obs_j, use_td, 1, r2, J46);
}

}  // extern "C"
