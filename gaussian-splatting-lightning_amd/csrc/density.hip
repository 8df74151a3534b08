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

}  // namespace gspl

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
