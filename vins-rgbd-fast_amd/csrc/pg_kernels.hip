// Loop-closure slice of pose_graph on the GPU (include/vio_posegraph.h): what KeyFrame's constructor computes for every keyframe and what
// KeyFrame::findConnection's descriptor search does (pose_graph/src/keyframe/keyframe.cpp:80-169, 530; DVision::BRIEF::compute).
//   pg_blur_kernel        cv::GaussianBlur(9x9, sigma 2) on u8, 64 x 16 tiles staged in LDS (coalesced row loads, halo 4)
//   pg_fast_score_kernel  FAST-9/16 corner score of every pixel (threshold = argument), 64 x 16 tiles with halo 3
//   pg_fast_nms_kernel    3x3 non-maximum suppression + row-major ordered compaction over the whole image (cv::FAST's output order)
//   pg_brief_kernel       one wavefront per point: lane l evaluates the pair tests l, l + 64, l + 128, l + 192 -> four ballots = the descriptor
//   pg_match_kernel       one wavefront per window descriptor: lanes stride over the old descriptors (v_bcnt popcounts), wave arg-min on
//                         (distance, index) = the sequential scan's "first smallest"
// Integer arithmetic throughout (the blur's 8-bit fixed point included), so the results are the CPU restatement's bit for bit.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/vio_posegraph.h"

extern thread_local std::string g_err;   // vio_abi.hip

namespace {

__device__ __forceinline__ int refl101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

#define PG_TW 64
#define PG_TH 16
// separable 9-tap Gaussian, 8 fractional bits per pass (52 46 32 17 7 | sum 256), one rounding: (v + 2^15) >> 16
__global__ __launch_bounds__(256) void pg_blur_kernel(const uint8_t *src, int W, int H, uint8_t *dst) {
    __shared__ uint8_t tile[PG_TH + 8][PG_TW + 8];
    __shared__ int hrow[PG_TH + 8][PG_TW];
    const int ox = blockIdx.x * PG_TW, oy = blockIdx.y * PG_TH;
    for (int q = threadIdx.x; q < (PG_TH + 8) * (PG_TW + 8); q += 256) {
        const int ty = q / (PG_TW + 8), tx = q - ty * (PG_TW + 8);
        const int gy = min(max(refl101(oy + ty - 4, H), 0), H - 1), gx = min(max(refl101(ox + tx - 4, W), 0), W - 1);
        tile[ty][tx] = src[(size_t)gy * W + gx];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < (PG_TH + 8) * PG_TW; q += 256) {
        const int ty = q / PG_TW, x = q - ty * PG_TW;
        const uint8_t *r = &tile[ty][x];
        hrow[ty][x] = 7 * (r[0] + r[8]) + 17 * (r[1] + r[7]) + 32 * (r[2] + r[6]) + 46 * (r[3] + r[5]) + 52 * r[4];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < PG_TH * PG_TW; q += 256) {
        const int y = q / PG_TW, x = q - y * PG_TW;
        if (ox + x < W && oy + y < H) {
            const int v = 7 * (hrow[y][x] + hrow[y + 8][x]) + 17 * (hrow[y + 1][x] + hrow[y + 7][x]) + 32 * (hrow[y + 2][x] + hrow[y + 6][x]) +
                          46 * (hrow[y + 3][x] + hrow[y + 5][x]) + 52 * hrow[y + 4][x];
            dst[(size_t)(oy + y) * W + ox + x] = (uint8_t)((v + (1 << 15)) >> 16);
        }
    }
}

__device__ __forceinline__ int pg_fast_score(const uint8_t *p, int stride, int thr) {
    const int v = p[0];
    int d[25];
    d[0] = v - p[3 * stride]; d[4] = v - p[3]; d[8] = v - p[-3 * stride]; d[12] = v - p[-3];
    const int nb = (d[0] < -thr) + (d[4] < -thr) + (d[8] < -thr) + (d[12] < -thr);
    const int nd = (d[0] > thr) + (d[4] > thr) + (d[8] > thr) + (d[12] > thr);
    if (nb < 2 && nd < 2) return 0;   // any 9-arc contains at least two of the four compass pixels
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
// score of every pixel at least 3 away from the border (0 elsewhere), tile (64 + 6) x (16 + 6) in LDS
__global__ __launch_bounds__(256) void pg_fast_score_kernel(const uint8_t *img, int W, int H, int thr, uint8_t *score) {
    __shared__ uint8_t tile[PG_TH + 6][PG_TW + 8];
    const int ox = blockIdx.x * PG_TW, oy = blockIdx.y * PG_TH;
    for (int q = threadIdx.x; q < (PG_TH + 6) * (PG_TW + 6); q += 256) {
        const int ty = q / (PG_TW + 6), tx = q - ty * (PG_TW + 6);
        const int gy = min(max(oy + ty - 3, 0), H - 1), gx = min(max(ox + tx - 3, 0), W - 1);
        tile[ty][tx] = img[(size_t)gy * W + gx];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < PG_TH * PG_TW; q += 256) {
        const int y = q / PG_TW, x = q - y * PG_TW;
        const int gx = ox + x, gy = oy + y;
        if (gx < W && gy < H) {
            int sc = 0;
            if (gx >= 3 && gx < W - 3 && gy >= 3 && gy < H - 3) sc = pg_fast_score(&tile[y + 3][x + 3], PG_TW + 8, thr);
            score[(size_t)gy * W + gx] = (uint8_t)sc;
        }
    }
}
// one workgroup of 1024 threads: the interior pixels in row-major order, 64 at a time per wavefront, contiguous ranges per wavefront; the
// ballot words go to `words`, the per-wavefront counts are scanned after one barrier and the survivors are written at their final positions
__global__ __launch_bounds__(1024) void pg_fast_nms_kernel(const uint8_t *score, int W, int H, unsigned long long *words, float *kp_xy, int cap, int *count) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, nw = blockDim.x >> 6;
    const int iw = W - 6, ih = H - 6, npx = iw * ih;
    const int nchunk = (npx + 63) >> 6, cpw = (nchunk + nw - 1) / nw;
    __shared__ int wtot[16];
    int mine = 0;
    for (int ch = wv * cpw; ch < min(nchunk, (wv + 1) * cpw); ch++) {
        const int q = ch * 64 + lane;
        bool mx = false;
        if (q < npx) {
            const int y = q / iw + 3, x = q - (q / iw) * iw + 3;
            const uint8_t *c = score + (size_t)y * W + x;
            const int sc = c[0];
            mx = sc && sc > c[-1] && sc > c[1] && sc > c[-W - 1] && sc > c[-W] && sc > c[-W + 1] && sc > c[W - 1] && sc > c[W] && sc > c[W + 1];
        }
        const unsigned long long bal = __ballot(mx);
        if (lane == 0) words[ch] = bal;
        mine += __popcll(bal);
    }
    if (lane == 0) wtot[wv] = mine;
    __syncthreads();
    int o = 0, total = 0;
    for (int k = 0; k < nw; k++) { if (k < wv) o += wtot[k]; total += wtot[k]; }
    for (int ch = wv * cpw; ch < min(nchunk, (wv + 1) * cpw); ch++) {
        const unsigned long long bal = words[ch];
        if ((bal >> lane) & 1ULL) {
            const int q = ch * 64 + lane, pos = o + __popcll(bal & ((1ULL << lane) - 1ULL));
            if (pos < cap) { kp_xy[2 * pos] = (float)(q - (q / iw) * iw + 3); kp_xy[2 * pos + 1] = (float)(q / iw + 3); }
        }
        o += __popcll(bal);
    }
    if (t == 0) *count = total;
}

// DVision::BRIEF::compute: bit i = I(p + (x1, y1)_i) < I(p + (x2, y2)_i) if both samples are inside the image; (int)(pt + offset) truncates
__global__ __launch_bounds__(64) void pg_brief_kernel(const uint8_t *blur, int W, int H, const float *xy, int n, const int *pat, unsigned long long *desc) {
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= n) return;
    const float px = xy[2 * p], py = xy[2 * p + 1];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = 64 * j + lane;
        const int x1 = (int)(px + (float)pat[i]), y1 = (int)(py + (float)pat[256 + i]);
        const int x2 = (int)(px + (float)pat[512 + i]), y2 = (int)(py + (float)pat[768 + i]);
        bool bit = false;
        if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H && x2 >= 0 && x2 < W && y2 >= 0 && y2 < H) bit = blur[(size_t)y1 * W + x1] < blur[(size_t)y2 * W + x2];
        const unsigned long long bal = __ballot(bit);
        if (lane == 0) desc[4 * (size_t)p + j] = bal;
    }
}

// PinholeCamera::liftProjective (camera_model/src/camera_models/PinholeCamera.cc:449-510): 8 fixed-point iterations of the radial-tangential model
__global__ void pg_lift_kernel(vio_config c, const float *xy, int n, float *nrm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double u = xy[2 * i], v = xy[2 * i + 1];
    const double mx_d = (1.0 / c.fx) * u + (-c.cx / c.fx), my_d = (1.0 / c.fy) * v + (-c.cy / c.fy);
    auto dist = [&](double x, double y, double &dx, double &dy) {
        const double mx2 = x * x, my2 = y * y, mxy = x * y, rho2 = mx2 + my2, rad = c.k1 * rho2 + c.k2 * rho2 * rho2;
        dx = x * rad + 2.0 * c.p1 * mxy + c.p2 * (rho2 + 2.0 * mx2);
        dy = y * rad + 2.0 * c.p2 * mxy + c.p1 * (rho2 + 2.0 * my2);
    };
    double dx, dy;
    dist(mx_d, my_d, dx, dy);
    double mx_u = mx_d - dx, my_u = my_d - dy;
    for (int k = 1; k < 8; k++) { dist(mx_u, my_u, dx, dy); mx_u = mx_d - dx; my_u = my_d - dy; }
    nrm[2 * i] = (float)mx_u; nrm[2 * i + 1] = (float)my_u;
}

// one wavefront per window descriptor
__global__ __launch_bounds__(64) void pg_match_kernel(const unsigned long long *wd, int n, const unsigned long long *od, int m, int *best_index, int *best_dist) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const unsigned long long a0 = wd[4 * (size_t)i], a1 = wd[4 * (size_t)i + 1], a2 = wd[4 * (size_t)i + 2], a3 = wd[4 * (size_t)i + 3];
    int key = (128 << 20) | 0xFFFFF;   // (distance << 20) | index: the minimum key = smallest distance, then smallest index
    for (int j = lane; j < m; j += 64) {
        const unsigned long long *b = od + 4 * (size_t)j;
        const int dis = __popcll(a0 ^ b[0]) + __popcll(a1 ^ b[1]) + __popcll(a2 ^ b[2]) + __popcll(a3 ^ b[3]);
        if (dis < 128) key = min(key, (dis << 20) | j);
    }
    for (int off = 32; off > 0; off >>= 1) key = min(key, __shfl_xor(key, off, 64));
    if (lane == 0) {
        const int bd = key >> 20, bi = key & 0xFFFFF;
        best_dist[i] = bd;
        best_index[i] = (bd < 80 && bi != 0xFFFFF) ? bi : -1;
    }
}

struct DevBuf {   // scope-bound device allocation
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <typename T> T *as() { return (T *)p; }
};
#define PGCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return VIO_EDEVICE; } } while (0)

int have_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_err = "no HIP device: the pose-graph kernels have no CPU fallback"; return VIO_EDEVICE; }
    return VIO_OK;
}

}  // namespace

extern "C" int vio_pg_stage_blur(const uint8_t *gray, int width, int height, uint8_t *out) {
    if (!gray || !out || width < 16 || height < 16) return VIO_EINVAL;
    if (int rc = have_device()) return rc;
    DevBuf a, b;
    const size_t hw = (size_t)width * height;
    PGCHK(a.alloc(hw)); PGCHK(b.alloc(hw));
    PGCHK(hipMemcpy(a.p, gray, hw, hipMemcpyHostToDevice));
    pg_blur_kernel<<<dim3((width + PG_TW - 1) / PG_TW, (height + PG_TH - 1) / PG_TH), 256>>>(a.as<uint8_t>(), width, height, b.as<uint8_t>());
    PGCHK(hipDeviceSynchronize());
    PGCHK(hipMemcpy(out, b.p, hw, hipMemcpyDeviceToHost));
    return VIO_OK;
}

extern "C" int vio_pg_describe(const vio_config *cfg, const uint8_t *gray, int n_win, const float *win_uv, const int32_t *pattern1024, int fast_threshold,
                               uint64_t *win_desc, int cap, float *kp_xy, uint64_t *kp_desc, float *kp_norm) {
    if (!cfg || !gray || !pattern1024 || n_win < 0 || cap < 0 || (n_win > 0 && (!win_uv || !win_desc)) || (cap > 0 && (!kp_xy || !kp_desc || !kp_norm))) return VIO_EINVAL;
    const int W = cfg->width, H = cfg->height;
    if (W < 16 || H < 16 || W > 4095 || H > 4095 || fast_threshold < 1 || fast_threshold > 254) return VIO_EINVAL;
    if (int rc = have_device()) return rc;
    const size_t hw = (size_t)W * H;
    DevBuf img, blur, score, words, pat, wuv, wdesc, kxy, kdesc, knrm, cnt;
    const int nchunk = ((W - 6) * (H - 6) + 63) / 64;
    PGCHK(img.alloc(hw)); PGCHK(blur.alloc(hw)); PGCHK(score.alloc(hw)); PGCHK(words.alloc((size_t)nchunk * 8)); PGCHK(pat.alloc(1024 * 4));
    PGCHK(wuv.alloc((size_t)n_win * 8)); PGCHK(wdesc.alloc((size_t)n_win * 32)); PGCHK(kxy.alloc((size_t)cap * 8)); PGCHK(kdesc.alloc((size_t)cap * 32));
    PGCHK(knrm.alloc((size_t)cap * 8)); PGCHK(cnt.alloc(4));
    PGCHK(hipMemcpy(img.p, gray, hw, hipMemcpyHostToDevice));
    PGCHK(hipMemcpy(pat.p, pattern1024, 1024 * 4, hipMemcpyHostToDevice));
    const dim3 tiles((W + PG_TW - 1) / PG_TW, (H + PG_TH - 1) / PG_TH);
    pg_blur_kernel<<<tiles, 256>>>(img.as<uint8_t>(), W, H, blur.as<uint8_t>());
    pg_fast_score_kernel<<<tiles, 256>>>(img.as<uint8_t>(), W, H, fast_threshold, score.as<uint8_t>());
    if (n_win > 0) {
        PGCHK(hipMemcpy(wuv.p, win_uv, (size_t)n_win * 8, hipMemcpyHostToDevice));
        pg_brief_kernel<<<n_win, 64>>>(blur.as<uint8_t>(), W, H, wuv.as<float>(), n_win, pat.as<int>(), wdesc.as<unsigned long long>());
    }
    pg_fast_nms_kernel<<<1, 1024>>>(score.as<uint8_t>(), W, H, words.as<unsigned long long>(), kxy.as<float>(), cap, cnt.as<int>());
    int total = 0;
    PGCHK(hipMemcpy(&total, cnt.p, 4, hipMemcpyDeviceToHost));
    const int m = total < cap ? total : cap;
    if (m > 0) {
        pg_brief_kernel<<<m, 64>>>(blur.as<uint8_t>(), W, H, kxy.as<float>(), m, pat.as<int>(), kdesc.as<unsigned long long>());
        pg_lift_kernel<<<(m + 255) / 256, 256>>>(*cfg, kxy.as<float>(), m, knrm.as<float>());
    }
    PGCHK(hipDeviceSynchronize());
    if (n_win > 0) PGCHK(hipMemcpy(win_desc, wdesc.p, (size_t)n_win * 32, hipMemcpyDeviceToHost));
    if (m > 0) {
        PGCHK(hipMemcpy(kp_xy, kxy.p, (size_t)m * 8, hipMemcpyDeviceToHost));
        PGCHK(hipMemcpy(kp_desc, kdesc.p, (size_t)m * 32, hipMemcpyDeviceToHost));
        PGCHK(hipMemcpy(kp_norm, knrm.p, (size_t)m * 8, hipMemcpyDeviceToHost));
    }
    return total;
}

extern "C" int vio_pg_match(const uint64_t *win_desc, int n, const uint64_t *old_desc, int m, int32_t *best_index, int32_t *best_dist) {
    if (n < 0 || m < 0 || m > 0xFFFFE || (n > 0 && (!win_desc || !best_index || !best_dist)) || (m > 0 && !old_desc)) return VIO_EINVAL;
    if (n == 0) return VIO_OK;
    if (int rc = have_device()) return rc;
    DevBuf a, b, bi, bd;
    PGCHK(a.alloc((size_t)n * 32)); PGCHK(b.alloc((size_t)m * 32)); PGCHK(bi.alloc((size_t)n * 4)); PGCHK(bd.alloc((size_t)n * 4));
    PGCHK(hipMemcpy(a.p, win_desc, (size_t)n * 32, hipMemcpyHostToDevice));
    if (m > 0) PGCHK(hipMemcpy(b.p, old_desc, (size_t)m * 32, hipMemcpyHostToDevice));
    pg_match_kernel<<<n, 64>>>(a.as<unsigned long long>(), n, b.as<unsigned long long>(), m, bi.as<int>(), bd.as<int>());
    PGCHK(hipDeviceSynchronize());
    PGCHK(hipMemcpy(best_index, bi.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    PGCHK(hipMemcpy(best_dist, bd.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return VIO_OK;
}
