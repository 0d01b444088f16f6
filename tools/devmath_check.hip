// Which double-precision operations give different bits on gfx950 and on the host?  (round 5: Estimator::predictMotion must be bit-reproducible
// between oracle/backend.cpp and fe_kernels.hip.)  sqrt, 1/x, y/x, the shared polynomial sincos_det and the angle-axis step of predictMotion, on
// 2^20 pseudo-random operands each; prints the number of operands whose results differ.  Build + run: tools/devmath_check.sh (needs a GPU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../vins-rgbd-fast_amd/csrc/dmath.h"

__host__ __device__ inline void eval(double x, double y, double *o) {
    o[0] = sqrt(x);
    o[1] = 1.0 / x;
    o[2] = y / x;
    double s, c;
    dm::sincos_det(y, &s, &c);
    o[3] = s; o[4] = c;
    // one predictMotion increment: axis-angle from a vector built of the operands
    dm::v3 aa = dm::mk(0.01 * y, -0.003 * x, 0.002 * (x - y));
    const double ang = dm::nrm(aa);
    dm::m3 Rk = dm::eye();
    if (ang > 0) {
        const dm::v3 ax = dm::scl(1.0 / ang, aa);
        double sn, cs;
        dm::sincos_det(ang, &sn, &cs);
        const dm::m3 K = dm::skew(ax);
        Rk = dm::add(dm::add(dm::eye(), dm::scl(sn, K)), dm::scl(1 - cs, dm::mul(K, K)));
    }
    for (int e = 0; e < 9; e++) o[5 + e] = Rk.a[e];
    o[14] = ang;
}
__global__ void k(const double *x, const double *y, double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double o[15];
    eval(x[i], y[i], o);
    for (int e = 0; e < 15; e++) out[(size_t)e * n + i] = o[e];
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), y(n), hd((size_t)15 * n);
    unsigned long long st = 88172645463325252ULL;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; i++) { x[i] = 0.05 + 3.0 * rnd(); y[i] = (rnd() - 0.5) * ((i & 7) == 0 ? 40.0 : 0.2); }
    double *dx, *dy, *dout;
    if (hipMalloc(&dx, n * 8) != hipSuccess) { printf("no device\n"); return 2; }
    hipMalloc(&dy, n * 8); hipMalloc(&dout, (size_t)15 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, dy, dout, n);
    hipMemcpy(hd.data(), dout, (size_t)15 * n * 8, hipMemcpyDeviceToHost);
    const char *names[15] = {"sqrt", "1/x", "y/x", "sincos_det.sin", "sincos_det.cos", "R00", "R01", "R02", "R10", "R11", "R12", "R20", "R21", "R22", "norm"};
    long bad[15] = {0};
    for (int i = 0; i < n; i++) {
        double o[15];
        eval(x[i], y[i], o);
        for (int e = 0; e < 15; e++) if (memcmp(&o[e], &hd[(size_t)e * n + i], 8) != 0) bad[e]++;
    }
    for (int e = 0; e < 15; e++) printf("%-16s differing operands: %ld of %d\n", names[e], bad[e], n);
    return 0;
}
