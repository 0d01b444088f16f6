// Synthetic workload generator — device renderer (outside any timed region). See include/vio_synth.h.
#include <hip/hip_runtime.h>
#include <vector>
#include <mutex>
#include <string.h>
#include "synth_scene.h"
#include "kernels.h"

using namespace vsyn;

// one thread per pixel; rays[H*W*2] are the undistorted normalised directions (shared by all sequences),
// poses[S*12] = world<-camera rotation (9) + camera centre (3)
__global__ __launch_bounds__(256) void synth_render_kernel(vio_synth_config c, int S, uint64_t seq0, const float *rays, const float *poses,
                                                           uint8_t *gray, uint16_t *depth) {
    int s = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int npx = c.width * c.height;
    if (i >= npx || s >= S) return;
    CamPose cp;
    for (int k = 0; k < 9; k++) cp.R[k] = poses[s * 12 + k];
    for (int k = 0; k < 3; k++) cp.p[k] = poses[s * 12 + 9 + k];
    uint32_t seed = (uint32_t)((c.seed + seq0 + (uint64_t)s) & 0xFFFFFFFFu);
    render_pixel(seed, cp, rays[2 * i], rays[2 * i + 1], gray + (size_t)s * npx + i, depth + (size_t)s * npx + i);
}

namespace {
struct RayCache {
    vio_synth_config key;
    float *d_rays = nullptr;
    bool valid = false;
};
RayCache g_rays;
std::mutex g_mu;
}  // namespace

extern "C" int vio_synth_render_device(const vio_synth_config *c, int S, uint64_t seq0, double t, uint8_t *d_gray, uint16_t *d_depth_mm,
                                       void *stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    hipStream_t st = (hipStream_t)stream;
    int npx = c->width * c->height;
    bool same = g_rays.valid && g_rays.key.width == c->width && g_rays.key.height == c->height && g_rays.key.fx == c->fx &&
                g_rays.key.fy == c->fy && g_rays.key.cx == c->cx && g_rays.key.cy == c->cy && g_rays.key.k1 == c->k1 &&
                g_rays.key.k2 == c->k2 && g_rays.key.p1 == c->p1 && g_rays.key.p2 == c->p2;
    if (!same) {
        if (g_rays.d_rays) (void)hipFree(g_rays.d_rays);
        std::vector<float> rays((size_t)npx * 2);
        for (int y = 0; y < c->height; y++)
            for (int x = 0; x < c->width; x++) {
                double rx, ry;
                syn_lift(c, (double)x, (double)y, &rx, &ry);
                rays[2 * ((size_t)y * c->width + x)] = (float)rx;
                rays[2 * ((size_t)y * c->width + x) + 1] = (float)ry;
            }
        if (hipMalloc(&g_rays.d_rays, rays.size() * sizeof(float)) != hipSuccess) return -2;
        if (hipMemcpy(g_rays.d_rays, rays.data(), rays.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -2;
        g_rays.key = *c;
        g_rays.valid = true;
    }
    std::vector<float> poses((size_t)S * 12);
    for (int s = 0; s < S; s++) {
        double p[3], R[9];
        vio_synth_pose(c, seq0 + (uint64_t)s, t, p, R, nullptr);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                double acc = 0;
                for (int q = 0; q < 3; q++) acc += R[i * 3 + q] * c->ric[q * 3 + j];
                poses[(size_t)s * 12 + i * 3 + j] = (float)acc;
            }
            poses[(size_t)s * 12 + 9 + i] = (float)(p[i] + R[i * 3 + 0] * c->tic[0] + R[i * 3 + 1] * c->tic[1] + R[i * 3 + 2] * c->tic[2]);
        }
    }
    float *d_poses = nullptr;
    if (hipMalloc(&d_poses, poses.size() * sizeof(float)) != hipSuccess) return -2;
    if (hipMemcpyAsync(d_poses, poses.data(), poses.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) return -2;
    dim3 grid((npx + 255) / 256, S);
    synth_render_kernel<<<grid, 256, 0, st>>>(*c, S, seq0, g_rays.d_rays, d_poses, d_gray, d_depth_mm);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(d_poses);
    return e == hipSuccess ? 0 : -2;
}
