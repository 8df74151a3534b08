// adam.hip — visibility-masked fused Adam over all per-Gaussian parameter tensors in one launch (gfx950).
//
// "Next" row of SURVEY.md §8f (rank 3).  Replaces gsplat's `SelectiveAdam` (un-vendored; reference wrapper
// internal/optimizers.py:26-58, enabled by configs/gsplat_v1-accel_more.yaml) and, with the mask absent and bias
// correction on, the per-tensor `torch.optim.Adam` steps of the default configuration
// (internal/models/vanilla_gaussian.py:266-300: one parameter group per property, eps 1e-15).
// Update restated (the published selective_adam kernel): for every element of every row n with visible[n]
//     m = b1 m + (1 - b1) g;   v = b2 v + (1 - b2) g^2;   p -= step * m / (sqrt(v) * inv_bc2 + eps)
// with step = lr, inv_bc2 = 1 (gsplat: no bias correction) or step = lr / (1 - b1^t), inv_bc2 = 1 / sqrt(1 - b2^t)
// (torch.optim.Adam).  Rows with visible[n] == 0 keep parameter AND moments untouched.
//
// HBM-bound: 28 B per updated element (p, g, m, v read; p, m, v written).  All tensors of the model go through one
// launch (blockIdx.y = tensor); lanes handle 16-byte chunks of the flat [N * row] arrays, the row's visibility is looked
// up per element (byte loads, cache resident).
#include "gspl_device.h"
#include "gspl_host.h"

namespace gspl {

struct AdamTensorDev {
    float* p;
    const float* g;
    float* m;
    float* v;
    float lr;
    int row;          // elements per Gaussian
};
struct AdamBatchDev { AdamTensorDev t[GSPL_ADAM_MAX_TENSORS]; };

// (adam_elem: gspl_device.h — shared with the backward kernels that apply the update themselves)

__global__ __launch_bounds__(256) void selective_adam_kernel(AdamBatchDev batch, int N, const uint8_t* __restrict__ visible,
                                                             float b1, float b2, float eps, float inv_bc1, float inv_bc2) {
    const AdamTensorDev T = batch.t[blockIdx.y];
    const int64_t total = (int64_t)N * T.row;
    const float step = T.lr * inv_bc1;
    const int64_t nvec = total >> 2;
#ifndef GSPL_ADAM_V1
    typedef float v4f __attribute__((ext_vector_type(4)));
    auto ntl = [](const float* base, int64_t i) { const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(base) + i); return make_float4(t.x, t.y, t.z, t.w); };
    auto nts = [](float* base, int64_t i, const float4& q) { v4f t = {q.x, q.y, q.z, q.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(base) + i); };
    // two 16-byte chunks per thread and iteration (8 loads in flight) and streaming (nontemporal) accesses of everything that is
    // not read again before the next step: 302 -> 257 us at 1 M Gaussians (1.65 GB: 6.4 TB/s); -DGSPL_ADAM_V1 = the one-chunk loop
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < nvec; i0 += 2 * stride) {
        float4 p[2], g[2], m[2], v[2];
        bool vis[2][4], any[2] = {false, false};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < nvec) {
                const int64_t e = i << 2;
#pragma unroll
                for (int k = 0; k < 4; ++k) { vis[u][k] = visible ? visible[(e + k) / T.row] != 0 : true; any[u] = any[u] || vis[u][k]; }
            }
            if (any[u]) {
                p[u] = ntl(T.p, i);
                g[u] = ntl(T.g, i);
                m[u] = ntl(T.m, i);
                v[u] = ntl(T.v, i);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!any[u]) continue;
            const int64_t i = i0 + u * stride;
            if (vis[u][0]) adam_elem(p[u].x, g[u].x, m[u].x, v[u].x, step, b1, b2, inv_bc2, eps);
            if (vis[u][1]) adam_elem(p[u].y, g[u].y, m[u].y, v[u].y, step, b1, b2, inv_bc2, eps);
            if (vis[u][2]) adam_elem(p[u].z, g[u].z, m[u].z, v[u].z, step, b1, b2, inv_bc2, eps);
            if (vis[u][3]) adam_elem(p[u].w, g[u].w, m[u].w, v[u].w, step, b1, b2, inv_bc2, eps);
            reinterpret_cast<float4*>(T.p)[i] = p[u];
            nts(T.m, i, m[u]);
            nts(T.v, i, v[u]);
        }
    }
#else
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i << 2;
        bool vis[4];
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) { vis[k] = visible ? visible[(e + k) / T.row] != 0 : true; any = any || vis[k]; }
        if (!any) continue;
        float4 p = reinterpret_cast<float4*>(T.p)[i];
        const float4 g = reinterpret_cast<const float4*>(T.g)[i];
        float4 m = reinterpret_cast<float4*>(T.m)[i];
        float4 v = reinterpret_cast<float4*>(T.v)[i];
        if (vis[0]) adam_elem(p.x, g.x, m.x, v.x, step, b1, b2, inv_bc2, eps);
        if (vis[1]) adam_elem(p.y, g.y, m.y, v.y, step, b1, b2, inv_bc2, eps);
        if (vis[2]) adam_elem(p.z, g.z, m.z, v.z, step, b1, b2, inv_bc2, eps);
        if (vis[3]) adam_elem(p.w, g.w, m.w, v.w, step, b1, b2, inv_bc2, eps);
        reinterpret_cast<float4*>(T.p)[i] = p;
        reinterpret_cast<float4*>(T.m)[i] = m;
        reinterpret_cast<float4*>(T.v)[i] = v;
    }
#endif
    // tail (total not a multiple of 4)
    if (blockIdx.x == 0) {
        for (int64_t e = (nvec << 2) + threadIdx.x; e < total; e += blockDim.x) {
            if (visible && visible[e / T.row] == 0) continue;
            float p = T.p[e], m = T.m[e], v = T.v[e];
            adam_elem(p, T.g[e], m, v, step, b1, b2, inv_bc2, eps);
            T.p[e] = p; T.m[e] = m; T.v[e] = v;
        }
    }
}

}  // namespace gspl

extern "C" int gspl_selective_adam(int n_tensors, const gspl_adam_tensor* tensors, int N, const uint8_t* visible,
                                   float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt,
                                   void* stream) {
    return gspl_selective_adam_limited(n_tensors, tensors, N, visible, beta1, beta2, eps, bias_correction1, bias_correction2_sqrt, 0, stream);
}

extern "C" int gspl_selective_adam_limited(int n_tensors, const gspl_adam_tensor* tensors, int N, const uint8_t* visible,
                                           float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt,
                                           int max_blocks, void* stream) {
    using namespace gspl;
    if (max_blocks < 0) return fail_arg("selective_adam: bad sizes");
    if (n_tensors < 0 || n_tensors > GSPL_ADAM_MAX_TENSORS || N < 0) return fail_arg("selective_adam: bad sizes");
    if (n_tensors == 0 || N == 0) return GSPL_OK;
    if (!tensors) return fail_arg("selective_adam: NULL tensor table");
    if (!(bias_correction1 > 0.f) || !(bias_correction2_sqrt > 0.f)) return fail_arg("selective_adam: bias corrections must be positive (1 = none)");
    AdamBatchDev b;
    int64_t longest = 0;
    for (int k = 0; k < n_tensors; ++k) {
        const gspl_adam_tensor& t = tensors[k];
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || t.row_elems <= 0) return fail_arg("selective_adam: bad tensor entry");
        if ((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15u) != 0)
            return fail_arg("selective_adam: tensors must be 16-byte aligned");
        b.t[k] = AdamTensorDev{t.param, t.grad, t.exp_avg, t.exp_avg_sq, t.lr, t.row_elems};
        longest = std::max<int64_t>(longest, (int64_t)N * t.row_elems);
    }
    for (int k = n_tensors; k < GSPL_ADAM_MAX_TENSORS; ++k) b.t[k] = b.t[0];
    const int64_t want = ((longest >> 2) + 255) / 256;
    // max_blocks > 0: a launch that runs NEXT TO other work (the deferred update on the colour stream) keeps to that many workgroups
    // per tensor, so that it leaves wave slots on every CU to the kernels of the other stream; the loop is grid-strided either way
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(want, max_blocks > 0 ? max_blocks : 16384));
    hipLaunchKernelGGL(selective_adam_kernel, dim3(gx, n_tensors), dim3(256), 0, (hipStream_t)stream, b, N, visible,
                       beta1, beta2, eps, 1.f / bias_correction1, 1.f / bias_correction2_sqrt);
    return check_launch("selective_adam");
}
