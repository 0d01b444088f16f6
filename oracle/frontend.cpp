// ORACLE (test infrastructure only — never linked into or called by the product path).
// CPU restatement of the front-end: vins_estimator/src/feature_tracker/feature_tracker.cpp (all),
// camera_model/src/camera_models/PinholeCamera.cc:449-542,645-662, plus the OpenCV routines the reference
// calls (FAST, calcOpticalFlowPyrLK, findFundamentalMat, circle) restated from SURVEY.md Appendix B.
// "parity unpinned": OpenCV is not available in this image, see DESIGN.md.
#include <array>
#include "oracle.h"
#include <cfloat>
#include <cmath>

namespace ovio {

// ------------------------------------------------------------------ camera (PinholeCamera.cc)
void cam_distortion(const Config &c, double x, double y, double &dx, double &dy) {  // :645-662
    double mx2 = x * x, my2 = y * y, mxy = x * y;
    double rho2 = mx2 + my2;
    double rad = c.k1 * rho2 + c.k2 * rho2 * rho2;
    dx = x * rad + 2.0 * c.p1 * mxy + c.p2 * (rho2 + 2.0 * mx2);
    dy = y * rad + 2.0 * c.p2 * mxy + c.p1 * (rho2 + 2.0 * my2);
}
void cam_lift(const Config &c, double u, double v, double &x, double &y) {  // :449-510 (recursive model, n = 8)
    double inv_K11 = 1.0 / c.fx, inv_K13 = -c.cx / c.fx, inv_K22 = 1.0 / c.fy, inv_K23 = -c.cy / c.fy;
    double mx_d = inv_K11 * u + inv_K13;
    double my_d = inv_K22 * v + inv_K23;
    double dx, dy;
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
void cam_project(const Config &c, double X, double Y, double Z, double &u, double &v) {  // :519-542
    double px = X / Z, py = Y / Z, dx, dy;
    cam_distortion(c, px, py, dx, dy);
    u = c.fx * (px + dx) + c.cx;
    v = c.fy * (py + dy) + c.cy;
}

// ------------------------------------------------------------------ helpers
static inline int reflect101(int i, int n) {
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}
static inline int cvRoundf(float v) { return (int)lrintf(v); }  // round-half-even, as cvRound
static inline int cvFloorf(float v) { return (int)floorf(v); }

// cv::pyrDown: separable [1 4 6 4 1]/16, BORDER_REFLECT_101, (sum+128)>>8   
void pyr_down(const Image &src, Image &dst) {
    int W = src.w, H = src.h, w = (W + 1) / 2, h = (H + 1) / 2;
    dst.w = w; dst.h = h;
    dst.d.assign((size_t)w * h, 0);
    static const int k[5] = {1, 4, 6, 4, 1};
    std::vector<int> row((size_t)5 * w);
    for (int y = 0; y < h; y++) {
        for (int j = 0; j < 5; j++) {
            int sy = reflect101(2 * y + j - 2, H);
            const uint8_t *s = &src.d[(size_t)sy * W];
            for (int x = 0; x < w; x++) {
                int acc = 0;
                for (int i = 0; i < 5; i++) acc += k[i] * s[reflect101(2 * x + i - 2, W)];
                row[(size_t)j * w + x] = acc;
            }
        }
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int j = 0; j < 5; j++) acc += k[j] * row[(size_t)j * w + x];
            dst.d[(size_t)y * w + x] = (uint8_t)((acc + 128) >> 8);
        }
    }
}

// cv::createCLAHE(clip, Size(tiles, tiles))->apply of FeatureTracker::readImage (feature_tracker.cpp:269-275; EQUALIZE).  The algorithm is
// OpenCV's (modules/imgproc/src/clahe.cpp, not in the reference tree -- restated from its published form, parity unpinned like the other
// OpenCV restatements): per-tile 256-bin histogram, clip at int(clip * tileArea / 256), the excess spread evenly plus one count on every
// (256 / residual)-th bin, LUT = round(cdf * 255 / tileArea); every pixel interpolates bilinearly between the LUTs of the four nearest tile
// centres (float arithmetic in this exact association, round-half-even).  A size that is not a multiple of `tiles` is padded at the bottom /
// right by tiles - (size % tiles) with BORDER_REFLECT_101 -- in BOTH directions, as cv::copyMakeBorder is called there.
void clahe_apply(const uint8_t *src, int W, int H, uint8_t *dst, double clip, int tiles) {
    int We = W, He = H;
    if (W % tiles != 0 || H % tiles != 0) { We = W + tiles - (W % tiles); He = H + tiles - (H % tiles); }
    const int tw = We / tiles, th = He / tiles, area = tw * th;
    const float lutScale = (float)(256 - 1) / area;
    int clipLimit = 0;
    if (clip > 0.0) { clipLimit = (int)(clip * area / 256); if (clipLimit < 1) clipLimit = 1; }
    std::vector<uint8_t> lut((size_t)tiles * tiles * 256);
    for (int k = 0; k < tiles * tiles; k++) {
        const int ty = k / tiles, tx = k % tiles;
        int hist[256] = {0};
        for (int y = 0; y < th; y++) {
            const int sy = reflect101(ty * th + y, H);
            for (int x = 0; x < tw; x++) hist[src[(size_t)sy * W + reflect101(tx * tw + x, W)]]++;
        }
        if (clipLimit > 0) {
            int clipped = 0;
            for (int i = 0; i < 256; i++)
                if (hist[i] > clipLimit) { clipped += hist[i] - clipLimit; hist[i] = clipLimit; }
            const int batch = clipped / 256;
            int residual = clipped - batch * 256;
            for (int i = 0; i < 256; i++) hist[i] += batch;
            if (residual != 0) {
                const int step = std::max(256 / residual, 1);
                for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
            }
        }
        int sum = 0;
        for (int i = 0; i < 256; i++) {
            sum += hist[i];
            int v = cvRoundf(sum * lutScale);
            lut[(size_t)k * 256 + i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < H; y++) {
        const float tyf = y * inv_th - 0.5f;
        int ty1 = cvFloorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        ty1 = std::max(ty1, 0); ty2 = std::min(ty2, tiles - 1);
        for (int x = 0; x < W; x++) {
            const float txf = x * inv_tw - 0.5f;
            int tx1 = cvFloorf(txf), tx2 = tx1 + 1;
            const float xa = txf - tx1, xa1 = 1.0f - xa;
            tx1 = std::max(tx1, 0); tx2 = std::min(tx2, tiles - 1);
            const int v = src[(size_t)y * W + x];
            const uint8_t *p1 = &lut[(size_t)ty1 * tiles * 256], *p2 = &lut[(size_t)ty2 * tiles * 256];
            const float res = (p1[tx1 * 256 + v] * xa1 + p1[tx2 * 256 + v] * xa) * ya1 + (p2[tx1 * 256 + v] * xa1 + p2[tx2 * 256 + v] * xa) * ya;
            const int r = cvRoundf(res);
            dst[(size_t)y * W + x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
}

// ------------------------------------------------------------------ FAST-9/16
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

int fast_corner_score(const uint8_t *p, int stride, int thr) {
    int v = p[0];
    int d[25];
    for (int k = 0; k < 16; k++) d[k] = v - p[RING_DY[k] * stride + RING_DX[k]];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    // best arc of 9 contiguous ring pixels: min over the arc of d (ring darker) and of -d (ring brighter)
    int best = 0;
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) {
            mn = d[k + j] < mn ? d[k + j] : mn;
            mx = d[k + j] > mx ? d[k + j] : mx;
        }
        if (mn > best) best = mn;
        if (-mx > best) best = -mx;
    }
    // corner iff some arc has all |d| > thr with one sign; score = largest threshold that keeps it a corner
    return best > thr ? best - 1 : 0;
}

void fast_detect_roi(const uint8_t *img, int W, int H, int rx, int ry, int rw, int rh, std::vector<KeyPt> &out, int thr) {
    (void)H;
    out.clear();
    if (rw < 7 || rh < 7) return;
    std::vector<uint8_t> score((size_t)rw * rh, 0);
    for (int i = 3; i < rh - 3; i++)
        for (int j = 3; j < rw - 3; j++)
            score[(size_t)i * rw + j] = (uint8_t)fast_corner_score(img + (size_t)(ry + i) * W + rx + j, W, thr);
    for (int i = 3; i < rh - 3; i++)
        for (int j = 3; j < rw - 3; j++) {
            int s = score[(size_t)i * rw + j];
            if (!s) continue;
            const uint8_t *c = &score[(size_t)i * rw + j];
            if (s > c[-1] && s > c[1] && s > c[-rw - 1] && s > c[-rw] && s > c[-rw + 1] && s > c[rw - 1] && s > c[rw] &&
                s > c[rw + 1])
                out.push_back(KeyPt{(float)j, (float)i, (float)s});
        }
}

// cv::circle(filled) row half-widths (OpenCV drawing.cpp Circle())
void circle_halfwidths(int radius, std::vector<int> &hw) {
    hw.assign(radius + 1, -1);
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > hw[dy]) hw[dy] = dx;  // rows cy±dy span cx±dx
        if (dy > hw[dx]) hw[dx] = dy;  // rows cy±dx span cx±dy
        dy++;
        err += plus;
        plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

// ------------------------------------------------------------------ pyramidal LK
static inline int img_at(const Image &im, int x, int y) {
    return im.d[(size_t)reflect101(y, im.h) * im.w + reflect101(x, im.w)];
}
static inline void scharr_at(const Image &im, int x, int y, int &ix, int &iy) {
    if (x < 0 || y < 0 || x >= im.w || y >= im.h) { ix = iy = 0; return; }  // constant 0 border of the derivative buffer
    int ym = reflect101(y - 1, im.h), yp = reflect101(y + 1, im.h);
    int xm = reflect101(x - 1, im.w), xp = reflect101(x + 1, im.w);
    const uint8_t *r0 = &im.d[(size_t)ym * im.w], *r1 = &im.d[(size_t)y * im.w], *r2 = &im.d[(size_t)yp * im.w];
    ix = 3 * (r0[xp] - r0[xm]) + 10 * (r1[xp] - r1[xm]) + 3 * (r2[xp] - r2[xm]);
    iy = 3 * (r2[xm] - r0[xm]) + 10 * (r2[x] - r0[x]) + 3 * (r2[xp] - r0[xp]);
}
static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

void lk_track(const std::vector<Image> &prev, const std::vector<Image> &next, const std::vector<P2f> &prevPts,
              std::vector<P2f> &nextPts, std::vector<uint8_t> &status, int maxLevel, bool useInitialFlow) {
    const int WIN = 21;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float minEigThreshold = 1e-4f;
    size_t n = prevPts.size();
    status.assign(n, 1);
    if (!useInitialFlow) nextPts = prevPts;
    std::vector<short> Ibuf(WIN * WIN), dIx(WIN * WIN), dIy(WIN * WIN);
    for (int level = maxLevel; level >= 0; level--) {
        const Image &I = prev[level], &J = next[level];
        for (size_t pi = 0; pi < n; pi++) {
            float sc = (float)(1. / (1 << level));
            P2f prevPt{prevPts[pi].x * sc, prevPts[pi].y * sc};
            P2f nextPt;
            if (level == maxLevel) {
                if (useInitialFlow) nextPt = P2f{nextPts[pi].x * sc, nextPts[pi].y * sc};
                else nextPt = prevPt;
            } else
                nextPt = P2f{nextPts[pi].x * 2.f, nextPts[pi].y * 2.f};
            nextPts[pi] = nextPt;
            const float halfWin = 10.f;
            prevPt.x -= halfWin; prevPt.y -= halfWin;
            int ipx = cvFloorf(prevPt.x), ipy = cvFloorf(prevPt.y);
            if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
                if (level == 0) status[pi] = 0;
                continue;
            }
            float a = prevPt.x - ipx, b = prevPt.y - ipy;
            int iw00 = cvRoundf((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cvRoundf(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cvRoundf((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            long long sA11 = 0, sA12 = 0, sA22 = 0;  // exact integer sums (order independent)
            for (int y = 0; y < WIN; y++)
                for (int x = 0; x < WIN; x++) {
                    int gx = ipx + x, gy = ipy + y;
                    int ival = descale(img_at(I, gx, gy) * iw00 + img_at(I, gx + 1, gy) * iw01 +
                                           img_at(I, gx, gy + 1) * iw10 + img_at(I, gx + 1, gy + 1) * iw11,
                                       W_BITS - 5);
                    int x00, y00, x01, y01, x10, y10, x11, y11;
                    scharr_at(I, gx, gy, x00, y00);
                    scharr_at(I, gx + 1, gy, x01, y01);
                    scharr_at(I, gx, gy + 1, x10, y10);
                    scharr_at(I, gx + 1, gy + 1, x11, y11);
                    int ixval = descale(x00 * iw00 + x01 * iw01 + x10 * iw10 + x11 * iw11, W_BITS);
                    int iyval = descale(y00 * iw00 + y01 * iw01 + y10 * iw10 + y11 * iw11, W_BITS);
                    Ibuf[y * WIN + x] = (short)ival;
                    dIx[y * WIN + x] = (short)ixval;
                    dIy[y * WIN + x] = (short)iyval;
                    sA11 += (long long)ixval * ixval;
                    sA12 += (long long)ixval * iyval;
                    sA22 += (long long)iyval * iyval;
                }
            float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * WIN * WIN);
            if (minEig < minEigThreshold || D < FLT_EPSILON) {
                if (level == 0) status[pi] = 0;
                continue;
            }
            D = 1.f / D;
            nextPt.x -= halfWin; nextPt.y -= halfWin;
            P2f prevDelta{0, 0};
            for (int j = 0; j < 30; j++) {
                int inx = cvFloorf(nextPt.x), iny = cvFloorf(nextPt.y);
                if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
                    if (level == 0) status[pi] = 0;
                    break;
                }
                a = nextPt.x - inx; b = nextPt.y - iny;
                iw00 = cvRoundf((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cvRoundf(a * (1.f - b) * (1 << W_BITS));
                iw10 = cvRoundf((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                long long sb1 = 0, sb2 = 0;
                for (int y = 0; y < WIN; y++)
                    for (int x = 0; x < WIN; x++) {
                        int gx = inx + x, gy = iny + y;
                        int diff = descale(img_at(J, gx, gy) * iw00 + img_at(J, gx + 1, gy) * iw01 +
                                               img_at(J, gx, gy + 1) * iw10 + img_at(J, gx + 1, gy + 1) * iw11,
                                           W_BITS - 5) -
                                   Ibuf[y * WIN + x];
                        sb1 += (long long)diff * dIx[y * WIN + x];
                        sb2 += (long long)diff * dIy[y * WIN + x];
                    }
                float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
                P2f delta{(float)((A12 * b2 - A22 * b1) * D), (float)((A12 * b1 - A11 * b2) * D)};
                nextPt.x += delta.x; nextPt.y += delta.y;
                nextPts[pi] = P2f{nextPt.x + halfWin, nextPt.y + halfWin};
                if ((double)delta.x * delta.x + (double)delta.y * delta.y <= 0.01 * 0.01) break;
                if (j > 0 && std::abs(delta.x + prevDelta.x) < 0.01 && std::abs(delta.y + prevDelta.y) < 0.01) {
                    nextPts[pi].x -= delta.x * 0.5f;
                    nextPts[pi].y -= delta.y * 0.5f;
                    break;
                }
                prevDelta = delta;
            }
        }
    }
}

// ------------------------------------------------------------------ F-matrix RANSAC (own fixed-seed sampler)
static inline uint64_t splitmix64(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline double det3(const double *F) {
    return F[0] * (F[4] * F[8] - F[5] * F[7]) - F[1] * (F[3] * F[8] - F[5] * F[6]) + F[2] * (F[3] * F[7] - F[4] * F[6]);
}
// roots of c3 x^3 + c2 x^2 + c1 x + c0 using only + - * / sqrt (bit-reproducible on CPU and GPU)
static int solve_cubic_det(double c3, double c2, double c1, double c0, double roots[3]) {
    double mx = std::fmax(std::fmax(std::fabs(c3), std::fabs(c2)), std::fmax(std::fabs(c1), std::fabs(c0)));
    if (mx == 0.0) return 0;
    if (std::fabs(c3) < 1e-12 * mx) {
        if (std::fabs(c2) < 1e-12 * mx) {
            if (std::fabs(c1) < 1e-12 * mx) return 0;
            roots[0] = -c0 / c1;
            return 1;
        }
        double disc = c1 * c1 - 4 * c2 * c0;
        if (disc < 0) return 0;
        double sq = std::sqrt(disc);
        roots[0] = (-c1 + sq) / (2 * c2);
        roots[1] = (-c1 - sq) / (2 * c2);
        return 2;
    }
    double a = c2 / c3, b = c1 / c3, c = c0 / c3;
    double B = 1.0 + std::fmax(std::fabs(a), std::fmax(std::fabs(b), std::fabs(c)));
    double lo = -B, hi = B;
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
    double p = a + r, q = b + r * p;  // x^2 + p x + q
    double disc = p * p - 4 * q;
    if (disc < 0) return 1;
    double sq = std::sqrt(disc);
    roots[1] = (-p + sq) * 0.5;
    roots[2] = (-p - sq) * 0.5;
    return 3;
}
// 7-point: returns number of models (≤3) in F[3][9]
static int seven_point(const double *x1, const double *y1, const double *x2, const double *y2, double F[3][9]) {
    double A[7][9];
    for (int i = 0; i < 7; i++) {
        A[i][0] = x2[i] * x1[i]; A[i][1] = x2[i] * y1[i]; A[i][2] = x2[i];
        A[i][3] = y2[i] * x1[i]; A[i][4] = y2[i] * y1[i]; A[i][5] = y2[i];
        A[i][6] = x1[i];         A[i][7] = y1[i];         A[i][8] = 1.0;
    }
    int perm[9];
    for (int i = 0; i < 9; i++) perm[i] = i;
    double amax = 0;
    for (int r = 0; r < 7; r++) for (int c = 0; c < 9; c++) amax = std::fmax(amax, std::fabs(A[r][c]));
    const double tol = 1e-12 * amax;
    int rank = 0;
    for (int i = 0; i < 7; i++) {
        int pr = i, pc = i;
        double best = -1;
        for (int r = i; r < 7; r++)
            for (int c = i; c < 9; c++)
                if (std::fabs(A[r][c]) > best) { best = std::fabs(A[r][c]); pr = r; pc = c; }
        if (!(best > tol)) break;  // rank-deficient sample (e.g. zero motion): remaining columns are free
        if (pr != i) for (int c = 0; c < 9; c++) std::swap(A[pr][c], A[i][c]);
        if (pc != i) {
            for (int r = 0; r < 7; r++) std::swap(A[r][pc], A[r][i]);
            std::swap(perm[pc], perm[i]);
        }
        double inv = 1.0 / A[i][i];
        for (int c = 0; c < 9; c++) A[i][c] *= inv;
        for (int r = 0; r < 7; r++) {
            if (r == i) continue;
            double f = A[r][i];
            if (f == 0.0) continue;
            for (int c = 0; c < 9; c++) A[r][c] -= f * A[i][c];
        }
        rank = i + 1;
    }
    // two null-space vectors: free columns 7 and 8 set to (1,0) / (0,1), other free columns 0
    double f1[9], f2[9];
    for (int i = 0; i < 9; i++) f1[i] = f2[i] = 0;
    for (int i = 0; i < rank; i++) { f1[perm[i]] = -A[i][7]; f2[perm[i]] = -A[i][8]; }
    f1[perm[7]] = 1;
    f2[perm[8]] = 1;
    // det(f1 + l f2) = c0 + c1 l + c2 l^2 + c3 l^3
    double c0 = det3(f1), c3 = det3(f2), c1 = 0, c2 = 0;
    for (int r = 0; r < 3; r++) {
        double t[9];
        std::memcpy(t, f1, sizeof(t));
        for (int c = 0; c < 3; c++) t[r * 3 + c] = f2[r * 3 + c];
        c1 += det3(t);
        std::memcpy(t, f2, sizeof(t));
        for (int c = 0; c < 3; c++) t[r * 3 + c] = f1[r * 3 + c];
        c2 += det3(t);
    }
    double roots[3];
    int nr = solve_cubic_det(c3, c2, c1, c0, roots);
    for (int k = 0; k < nr; k++)
        for (int i = 0; i < 9; i++) F[k][i] = f1[i] + roots[k] * f2[i];
    return nr;
}
// test hook (tests/test_oracle_numpy_cpu.py): the 7-point solver on its own
int seven_point_models(const double *x1, const double *y1, const double *x2, const double *y2, double F[3][9]) { return seven_point(x1, y1, x2, y2, F); }
static int ransac_update_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::fmax(p, 0.); p = std::fmin(p, 1.);
    ep = std::fmax(ep, 0.); ep = std::fmin(ep, 1.);
    double num = std::fmax(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)lrint(num / denom);
}

void ransac_fundamental(const Config &c, const std::vector<P2f> &p1, const std::vector<P2f> &p2,
                        std::vector<uint8_t> &status) {
    int N = (int)p1.size();
    status.assign(N, 0);
    if (N < 8) return;
    std::vector<double> X1(N), Y1(N), X2(N), Y2(N);
    double hc = c.width / 2.0, hr = c.height / 2.0;
    for (int i = 0; i < N; i++) {
        X1[i] = ((double)p1[i].x - hc) / c.focal_length; Y1[i] = ((double)p1[i].y - hr) / c.focal_length;
        X2[i] = ((double)p2[i].x - hc) / c.focal_length; Y2[i] = ((double)p2[i].y - hr) / c.focal_length;
    }
    double thr = c.f_threshold / c.focal_length;
    double thr2 = thr * thr;
    int niters = c.ransac_max_iters, maxGood = 0;
    std::vector<uint8_t> cur(N);
    for (int it = 0; it < niters; it++) {
        uint64_t s = 0x5649464D41545258ULL + (uint64_t)it * 0xD1B54A32D192ED03ULL;
        int idx[7];
        for (int k = 0; k < 7;) {
            int r = (int)(splitmix64(s) % (uint64_t)N);
            bool dup = false;
            for (int j = 0; j < k; j++) dup |= (idx[j] == r);
            if (!dup) idx[k++] = r;
        }
        double sx1[7], sy1[7], sx2[7], sy2[7];
        for (int k = 0; k < 7; k++) { sx1[k] = X1[idx[k]]; sy1[k] = Y1[idx[k]]; sx2[k] = X2[idx[k]]; sy2[k] = Y2[idx[k]]; }
        double F[3][9];
        int nm = seven_point(sx1, sy1, sx2, sy2, F);
        for (int m = 0; m < nm; m++) {
            const double *f = F[m];
            int good = 0;
            for (int i = 0; i < N; i++) {
                double a = f[0] * X1[i] + f[1] * Y1[i] + f[2], b = f[3] * X1[i] + f[4] * Y1[i] + f[5],
                       cc = f[6] * X1[i] + f[7] * Y1[i] + f[8];
                double s2 = 1.0 / (a * a + b * b), d2 = X2[i] * a + Y2[i] * b + cc;
                double a1 = f[0] * X2[i] + f[3] * Y2[i] + f[6], b1 = f[1] * X2[i] + f[4] * Y2[i] + f[7],
                       c1 = f[2] * X2[i] + f[5] * Y2[i] + f[8];
                double s1 = 1.0 / (a1 * a1 + b1 * b1), d1 = X1[i] * a1 + Y1[i] * b1 + c1;
                double err = std::fmax(d1 * d1 * s1, d2 * d2 * s2);
                cur[i] = err <= thr2;
                good += cur[i];
            }
            if (good > std::max(maxGood, 6)) {
                maxGood = good;
                status = cur;
                niters = ransac_update_iters(0.99, (double)(N - good) / N, 7, niters);
            }
        }
    }
}

// ------------------------------------------------------------------ FeatureTracker
Tracker::Tracker(const Config &c) : cfg(c) {
    // initGridsDetector feature_tracker.cpp:33-94
    int ROW = c.height, COL = c.width, NR = c.grid_rows, NC = c.grid_cols;
    grid_height = ROW / NR;
    grid_width = COL / NC;
    grid_res_height = ROW - (NR - 1) * grid_height;
    grid_res_width = COL - (NC - 1) * grid_width;
    for (int i = 0; i < NR; i++)
        for (int j = 0; j < NC; j++) {
            Rect r;
            r.x = j == 0 ? 0 : j * grid_width - 3;
            r.y = i == 0 ? 0 : i * grid_height - 3;
            int gw = (j == NC - 1) ? grid_res_width : grid_width;
            int gh = (i == NR - 1) ? grid_res_height : grid_height;
            r.w = gw + ((j > 0 && j < NC - 1) ? 6 : 3);
            r.h = gh + ((i > 0 && i < NR - 1) ? 6 : 3);
            if (NC == 1) r.w = gw;  // single column: upstream formula would overrun; keep inside the image
            if (NR == 1) r.h = gh;
            grids_rect.push_back(r);
            grids_track_num.push_back(0);
            grids_texture_status.push_back(1);
        }
    grids_threshold = c.max_cnt / (int)grids_rect.size();
    circle_halfwidths(c.min_dist, circle_hw);
    mask.assign((size_t)ROW * COL, 255);
}

bool Tracker::inBorder(const P2f &pt) const {  // feature_tracker.cpp:96-103
    const int B = 1;
    int x = cvRoundf(pt.x), y = cvRoundf(pt.y);
    return B <= x && x < cfg.width - B && B <= y && y < cfg.height - B;
}
uint8_t Tracker::maskAt(const P2f &pt) const { return mask[(size_t)cvRoundf(pt.y) * cfg.width + cvRoundf(pt.x)]; }
void Tracker::drawCircle(const P2f &pt) {
    int cx = cvRoundf(pt.x), cy = cvRoundf(pt.y), r = cfg.min_dist;
    for (int dy = -r; dy <= r; dy++) {
        int y = cy + dy;
        if (y < 0 || y >= cfg.height) continue;
        int hw = circle_hw[dy < 0 ? -dy : dy];
        int x0 = std::max(cx - hw, 0), x1 = std::min(cx + hw, cfg.width - 1);
        for (int x = x0; x <= x1; x++) mask[(size_t)y * cfg.width + x] = 0;
    }
}

template <class T> static void reduceVector(std::vector<T> &v, const std::vector<uint8_t> &status) {  // :9-25
    int j = 0;
    for (int i = 0; i < (int)v.size(); i++)
        if (status[i]) v[j++] = v[i];
    v.resize(j);
}

void Tracker::predictPtsInNextFrame(const double R[9]) {  // :595-608
    predict_pts.resize(cur_pts.size());
    for (size_t i = 0; i < cur_pts.size(); i++) {
        double x, y;
        cam_lift(cfg, cur_pts[i].x, cur_pts[i].y, x, y);
        double X = R[0] * x + R[1] * y + R[2], Y = R[3] * x + R[4] * y + R[5], Z = R[6] * x + R[7] * y + R[8];
        double u, v;
        cam_project(cfg, X, Y, Z, u, v);
        predict_pts[i] = P2f{(float)u, (float)v};
    }
}

void Tracker::rejectWithF() {  // :441-473
    if (forw_pts.size() < 8) return;
    std::vector<P2f> un_cur(cur_pts.size()), un_forw(forw_pts.size());
    for (size_t i = 0; i < cur_pts.size(); i++) {
        double x, y;
        cam_lift(cfg, cur_pts[i].x, cur_pts[i].y, x, y);
        un_cur[i] = P2f{(float)(cfg.focal_length * x + cfg.width / 2.0), (float)(cfg.focal_length * y + cfg.height / 2.0)};
        cam_lift(cfg, forw_pts[i].x, forw_pts[i].y, x, y);
        un_forw[i] = P2f{(float)(cfg.focal_length * x + cfg.width / 2.0), (float)(cfg.focal_length * y + cfg.height / 2.0)};
    }
    std::vector<uint8_t> status;
    ransac_fundamental(cfg, un_cur, un_forw, status);
    reduceVector(cur_pts, status);
    reduceVector(forw_pts, status);
    reduceVector(cur_un_pts, status);
    reduceVector(ids, status);
    reduceVector(track_cnt, status);
}

void Tracker::setMask() {  // :173-208 (std::sort ties pinned to original order = stable)
    if (!fisheye_mask.empty()) mask = fisheye_mask;   // FISHEYE: mask = fisheye_mask.clone() (:175-176)
    else std::fill(mask.begin(), mask.end(), 255);
    std::vector<int> order(forw_pts.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return track_cnt[a] > track_cnt[b]; });
    std::vector<P2f> fp;
    std::vector<int> fid, fcnt;
    for (int i : order) {
        if (maskAt(forw_pts[i]) == 255) {
            fp.push_back(forw_pts[i]);
            fid.push_back(ids[i]);
            fcnt.push_back(track_cnt[i]);
            drawCircle(forw_pts[i]);
        }
    }
    forw_pts = fp; ids = fid; track_cnt = fcnt;
    for (auto &pt : unstable_pts) drawCircle(pt);
}

std::vector<KeyPt> Tracker::gridDetect(int g) {  // :105-171
    const Rect &r = grids_rect[g];
    std::vector<KeyPt> all, kps;
    fast_detect_roi(forw_pyr[0].d.data(), cfg.width, cfg.height, r.x, r.y, r.w, r.h, all);
    // KeyPointsFilter::runByPixelsMask on mask(rect)
    for (auto &k : all)
        if (mask[(size_t)(r.y + (int)(k.y + 0.5f)) * cfg.width + r.x + (int)(k.x + 0.5f)] != 0) kps.push_back(k);
    if (kps.empty()) {
        grids_texture_status[g] = 0;
        return {};
    }
    size_t num_to_add = (size_t)(grids_threshold - grids_track_num[g] + 2);
    if (kps.size() <= num_to_add) {
        for (auto &k : kps) { k.x += r.x; k.y += r.y; }
        return kps;
    }
    std::vector<KeyPt> keep(num_to_add, KeyPt{0, 0, 0});
    size_t min_id = 0, remaining = num_to_add;
    for (size_t j = 0; j < kps.size(); j++) {
        if (remaining > 0) {
            kps[j].x += r.x; kps[j].y += r.y;
            keep[j] = kps[j];
            --remaining;
            if (kps[j].response < keep[min_id].response) min_id = j;
        } else if (kps[j].response > keep[min_id].response) {
            kps[j].x += r.x; kps[j].y += r.y;
            keep[min_id] = kps[j];
            for (size_t k = 0; k < keep.size(); k++)
                if (keep[k].response < keep[min_id].response) min_id = k;
        }
    }
    return keep;
}

void Tracker::addPoints(const std::vector<KeyPt> &kps) {  // :220-233
    for (auto &k : kps) {
        P2f p{k.x, k.y};
        if (maskAt(p) == 255) {
            forw_pts.push_back(p);
            ids.push_back(-1);
            track_cnt.push_back(1);
            drawCircle(p);
        }
    }
}

void Tracker::undistortedPoints() {  // :542-593
    cur_un_pts.clear();
    cur_un_pts_map.clear();
    for (size_t i = 0; i < cur_pts.size(); i++) {
        double x, y;
        cam_lift(cfg, cur_pts[i].x, cur_pts[i].y, x, y);
        P2f p{(float)x, (float)y};
        cur_un_pts.push_back(p);
        cur_un_pts_map.insert(std::make_pair(ids[i], p));
    }
    pts_velocity.clear();
    if (!prev_un_pts_map.empty()) {
        double dt = cur_time - prev_time;
        for (size_t i = 0; i < cur_un_pts.size(); i++) {
            if (ids[i] != -1) {
                auto it = prev_un_pts_map.find(ids[i]);
                if (it != prev_un_pts_map.end()) {
                    double vx = (cur_un_pts[i].x - it->second.x) / dt;
                    double vy = (cur_un_pts[i].y - it->second.y) / dt;
                    pts_velocity.push_back(P2f{(float)vx, (float)vy});
                } else
                    pts_velocity.push_back(P2f{0, 0});
            } else
                pts_velocity.push_back(P2f{0, 0});
        }
    } else {
        for (size_t i = 0; i < cur_pts.size(); i++) pts_velocity.push_back(P2f{0, 0});
    }
    prev_un_pts_map = cur_un_pts_map;
}

void Tracker::readImage(const uint8_t *img, double t, const double R[9], bool publish) {  // :263-439
    cur_time = t;
    int maxLevel = cfg.lk_max_level;
    std::vector<Image> pyr(maxLevel + 1);
    pyr[0].w = cfg.width; pyr[0].h = cfg.height;
    if (cfg.equalize) {   // :269-275
        pyr[0].d.resize((size_t)cfg.width * cfg.height);
        clahe_apply(img, cfg.width, cfg.height, pyr[0].d.data(), 3.0, 8);
    } else
        pyr[0].d.assign(img, img + (size_t)cfg.width * cfg.height);
    for (int l = 1; l <= maxLevel; l++) pyr_down(pyr[l - 1], pyr[l]);
    if (!has_img) {
        cur_pyr = pyr;
        has_img = true;
    }
    forw_pyr.swap(pyr);
    forw_pts.clear();
    unstable_pts.clear();

    if (!cur_pts.empty()) {
        std::vector<uint8_t> status;
        if (cfg.use_imu) {   // feature_tracker.cpp:298-306
            predictPtsInNextFrame(R);
            forw_pts = predict_pts;
            lk_track(cur_pyr, forw_pyr, cur_pts, forw_pts, status, maxLevel, true);
        } else {             // :307-311: calcOpticalFlowPyrLK(..., Size(21, 21), 3) without OPTFLOW_USE_INITIAL_FLOW
            forw_pts = cur_pts;
            lk_track(cur_pyr, forw_pyr, cur_pts, forw_pts, status, maxLevel, false);
        }
        for (size_t i = 0; i < forw_pts.size(); i++) {
            if (!status[i] && inBorder(forw_pts[i])) unstable_pts.push_back(forw_pts[i]);
            else if (status[i] && !inBorder(forw_pts[i])) status[i] = 0;
        }
        reduceVector(cur_pts, status);
        reduceVector(forw_pts, status);
        reduceVector(ids, status);
        reduceVector(cur_un_pts, status);
        reduceVector(track_cnt, status);
    }
    for (auto &n : track_cnt) n++;

    if (publish) {
        rejectWithF();
        setMask();
        int n_max_cnt = cfg.max_cnt - (int)forw_pts.size();
        if (n_max_cnt > 0) {
            for (auto &g : grids_track_num) g = 0;
            for (auto &p : forw_pts) {
                int col = (int)p.x / grid_width, row = (int)p.y / grid_height;
                if (col == cfg.grid_cols) --col;
                if (row == cfg.grid_rows) --row;
                ++grids_track_num[col + cfg.grid_cols * row];
            }
            std::vector<int> grids_id;
            for (size_t i = 0; i < grids_rect.size(); i++) {
                if (grids_track_num[i] < grids_threshold && grids_texture_status[i]) grids_id.push_back((int)i);
                else grids_texture_status[i] = 1;
            }
            // sequential semantics: cell k detects against the mask after cells <k were added (one legal
            // interleaving of the thread-pool race at feature_tracker.cpp:397-409)
            for (int g : grids_id) {
                std::vector<KeyPt> kps = gridDetect(g);
                addPoints(kps);
            }
        }
    }
    cur_pyr = forw_pyr;
    cur_pts = forw_pts;
    undistortedPoints();
    prev_time = cur_time;
}

void Tracker::updateIDs() {  // :485-495 looped as in estimator_nodelet.cpp:324-330
    for (size_t i = 0; i < ids.size(); i++)
        if (ids[i] == -1) ids[i] = n_id++;
}

}  // namespace ovio
