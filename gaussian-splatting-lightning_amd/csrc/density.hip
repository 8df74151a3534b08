// density.hip — the per-step densification statistics of the density controller in one launch (gfx950).
//
// SURVEY.md §8 a14 / §8f rank 3.  Replaces the PyTorch lines of `VanillaDensityControllerImpl.update_states` and
// `_add_densification_stats` (internal/density_controllers/vanilla_density_controller.py:101-123): masked max of the
// screen radii, masked accumulation of the norm of the (scaled) screen-space gradient, masked count —
//     max_radii2D[v] = max(max_radii2D[v], radii[v]);  accum[v] += |grad[v, :2] * scale|_2;  denom[v] += 1
// which PyTorch runs as a dozen gather/scatter launches (boolean-mask indexing materialises index lists).  HBM-bound,
// 29 B per Gaussian; visible rows only are written.
#include "gspl_device.h"
#include "gspl_host.h"

namespace gspl {

__global__ __launch_bounds__(256) void densify_stats_kernel(int N, const float* __restrict__ grad, int grad_stride, float sx, float sy,
                                                            const float* __restrict__ scale_dev, const uint8_t* __restrict__ visible,
                                                            const int32_t* __restrict__ radii_i, const float* __restrict__ radii_f,
                                                            float* __restrict__ accum, float* __restrict__ denom, float* __restrict__ max_radii) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const float r = radii_i ? (float)radii_i[g] : (radii_f ? radii_f[g] : 0.f);
    const bool vis = visible ? visible[g] != 0 : r > 0.f;
    if (!vis) return;
    if (scale_dev) { sx = scale_dev[0]; sy = scale_dev[1]; }
    const float gx = grad[(size_t)g * grad_stride + 0] * sx, gy = grad[(size_t)g * grad_stride + 1] * sy;
    accum[g] += sqrtf(fmaf(gx, gx, gy * gy));      // spelled out: inria_preprocess_bwd_kernel applies the same update (BwdStats)
    denom[g] += 1.f;
    if (max_radii) max_radii[g] = fmaxf(max_radii[g], r);
}

// The same update for the C cameras of a Gaussian-sharded step in ONE launch (round 6): the reference's distributed controller loops
// over the step's cameras (internal/density_controllers/distributed_vanilla_density_controller.py:22-47), eight launches at W = 8.
// Thread g walks the cameras in order, so accum / denom / max_radii receive exactly the additions of C sequential single-camera
// launches, in the same order: bit-identical buffers.
struct StatsViews {
    const float* grad[GSPL_STATS_MAX_VIEWS];
    const uint8_t* visible[GSPL_STATS_MAX_VIEWS];
    const int32_t* radii[GSPL_STATS_MAX_VIEWS];
};
__global__ __launch_bounds__(256) void densify_stats_views_kernel(int N, int C, StatsViews v, int grad_stride, float sx, float sy,
                                                                  const float* __restrict__ scale_dev,
                                                                  float* __restrict__ accum, float* __restrict__ denom, float* __restrict__ max_radii) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    if (scale_dev) { sx = scale_dev[0]; sy = scale_dev[1]; }
    float a = 0.f, d = 0.f, m = 0.f;
    bool any = false;
    for (int c = 0; c < C; ++c) {
        const float r = v.radii[c] ? (float)v.radii[c][g] : 0.f;
        const bool vis = v.visible[c] ? v.visible[c][g] != 0 : r > 0.f;
        if (!vis) continue;
        if (!any) { a = accum[g]; d = denom[g]; m = max_radii ? max_radii[g] : 0.f; any = true; }
        const float gx = v.grad[c][(size_t)g * grad_stride + 0] * sx, gy = v.grad[c][(size_t)g * grad_stride + 1] * sy;
        a += sqrtf(fmaf(gx, gx, gy * gy));
        d += 1.f;
        m = fmaxf(m, r);
    }
    if (any) {
        accum[g] = a; denom[g] = d;
        if (max_radii) max_radii[g] = m;
    }
}

}  // namespace gspl

extern "C" int gspl_densify_stats_views(int N, int n_views, const float* const* grads, int grad_stride, float scale_x, float scale_y, const float* scale_dev,
                                        const uint8_t* const* visible, const int32_t* const* radii_i32,
                                        float* accum, float* denom, float* max_radii, void* stream) {
    using namespace gspl;
    if (N < 0 || grad_stride < 2 || n_views < 1 || n_views > GSPL_STATS_MAX_VIEWS) return fail_arg("densify_stats_views: bad sizes (1 .. 16 views)");
    if (N == 0) return GSPL_OK;
    if (!grads || !accum || !denom) return fail_arg("densify_stats_views: NULL required pointer");
    if (!visible && !radii_i32) return fail_arg("densify_stats_views: visibility masks or radii are needed");
    if (max_radii && !radii_i32) return fail_arg("densify_stats_views: max_radii without radii");
    StatsViews v = {};
    for (int c = 0; c < n_views; ++c) {
        if (!grads[c]) return fail_arg("densify_stats_views: NULL gradient of a view");
        v.grad[c] = grads[c];
        v.visible[c] = visible ? visible[c] : nullptr;
        v.radii[c] = radii_i32 ? radii_i32[c] : nullptr;
        if (!v.visible[c] && !v.radii[c]) return fail_arg("densify_stats_views: a view with neither mask nor radii");
        if (max_radii && !v.radii[c]) return fail_arg("densify_stats_views: max_radii without the radii of a view");
    }
    hipLaunchKernelGGL(densify_stats_views_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, n_views, v, grad_stride, scale_x, scale_y,
                       scale_dev, accum, denom, max_radii);
    return check_launch("densify_stats_views");
}

extern "C" int gspl_densify_stats(int N, const float* grad, int grad_stride, float scale_x, float scale_y, const float* scale_dev,
                                  const uint8_t* visible, const int32_t* radii_i32, const float* radii_f32,
                                  float* accum, float* denom, float* max_radii, void* stream) {
    using namespace gspl;
    if (N < 0 || grad_stride < 2) return fail_arg("densify_stats: bad sizes");
    if (N == 0) return GSPL_OK;
    if (!grad || !accum || !denom) return fail_arg("densify_stats: NULL required pointer");
    if (radii_i32 && radii_f32) return fail_arg("densify_stats: radii as int32 OR float32");
    if (!visible && !radii_i32 && !radii_f32) return fail_arg("densify_stats: a visibility mask or radii are needed");
    if (max_radii && !radii_i32 && !radii_f32) return fail_arg("densify_stats: max_radii without radii");
    hipLaunchKernelGGL(densify_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, grad, grad_stride, scale_x, scale_y,
                       scale_dev, visible, radii_i32, radii_f32, accum, denom, max_radii);
    return check_launch("densify_stats");
}
