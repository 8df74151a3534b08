// inria.hip — Inria-convention preprocess (front half of the fused `GaussianRasterizer`), fwd + bwd.
//
// Replaces the part of `diff_gaussian_rasterization.GaussianRasterizer` that runs before the sort,
// at the reference call sites internal/renderers/vanilla_renderer.py:62-77,111-120 (forward) and
// its autograd backward.  The package is not vendored in the reference; the algorithm restated
// here is the published 3DGS preprocess with the constants of SURVEY.md Appendix B:
//   cull view-z <= 0.2 ; cov3D = (R S)(R S)^T ; EWA cov2D with x/z, y/z clamped to 1.3 tan(fov/2),
//   focal = W / (2 tan(fov/2)) ; +0.3 on the cov2D diagonal (no compensation) ; cull det == 0 ;
//   radius = ceil(3 sqrt(mid + sqrt(max(0.1, mid^2 - det)))) ; mean2D = ((ndc + 1) S - 1) / 2 with
//   ndc = (P p).xy / ((P p).w + 1e-7) ; tile rect [(p-r)/16, (p+r+15)/16) ; cull empty rect ;
//   colour = max(0, SH(dir = normalise(p - campos)) + 0.5) with the clamp recorded for backward.
// The reference hands matrices over in its transposed (row-vector) storage
// (internal/cameras/cameras.py:147-189): p_view = p V[:3,:3] + V[3,:3].
//
// The colour stage reuses the LDS-staged SH kernels of sh.hip (dirs = means, origin = campos,
// GSPL_SH_ADD_HALF_CLAMP, masked by radii); backward writes dL/d(dir) into v_means first and the
// geometry kernel then accumulates the projection/covariance terms on top.
// Roofline: HBM-bound elementwise.
// Floating-point contraction as the LANGUAGE defines it (a * b + c inside one expression), not as the back end finds it: the
// gradient-writing and the Adam-applying instantiations of the backward kernels below must produce bit-identical gradient values
// (tests/test_fused_backward_adam.py), and -ffp-contract=fast (hipcc's default) lets the DAG combiner fuse across statements
// differently in each instantiation (measured: the means' gradient differed in its last bit from the third step on).
#pragma clang fp contract(on)
#include "gspl_device.h"
#include "gspl_host.h"

namespace gspl {

struct InriaCam {
    float V[16];   // row-vector storage: p_view[c] = sum_r p[r] V[r*4+c] + V[12+c]
    float P[16];
};

__device__ __forceinline__ void load_inria_cam(const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, InriaCam& c) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { c.V[i] = viewmatrix[i]; c.P[i] = projmatrix[i]; }
}

// Block-cooperative copies of `rows` consecutive rows of W floats between a dense [*, W] array in global memory and LDS, as 16-byte
// accesses (every fetched line fully used by ONE instruction) whenever the block's first row is 16-byte aligned; the per-Gaussian
// arithmetic then reads / writes its own row in LDS.  Per-lane strided dword accesses of the AoS rows — the first version — touch
// W lines per instruction and use 1/W of each: 2.8 TB/s at 1 M Gaussians where this form streams.
template <int NT>
__device__ __forceinline__ void rows_to_lds(const float* __restrict__ g, int64_t row0, int rows, int W, float* __restrict__ s) {
    const float* base = g + row0 * W;
    const int total = rows * W;
    if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
        const int n4 = total >> 2;
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        for (int i = threadIdx.x; i < n4; i += NT) reinterpret_cast<v4f_*>(s)[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f_*>(base) + i);
        for (int i = (n4 << 2) + threadIdx.x; i < total; i += NT) s[i] = base[i];
    } else {
        for (int i = threadIdx.x; i < total; i += NT) s[i] = base[i];
    }
}
template <int NT>
__device__ __forceinline__ void lds_to_rows(float* __restrict__ g, int64_t row0, int rows, int W, const float* __restrict__ s) {
    float* base = g + row0 * W;
    const int total = rows * W;
    if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
        const int n4 = total >> 2;
        for (int i = threadIdx.x; i < n4; i += NT) reinterpret_cast<float4*>(base)[i] = reinterpret_cast<const float4*>(s)[i];
        for (int i = (n4 << 2) + threadIdx.x; i < total; i += NT) base[i] = s[i];
    } else {
        for (int i = threadIdx.x; i < total; i += NT) base[i] = s[i];
    }
}

// common geometry: returns false when culled; fills everything needed by fwd and bwd
struct InriaGeom {
    float pv[3];        // view-space mean
    float hom[4];       // P p
    float S6[6];
    float Wstd[9];      // standard-orientation rotation (row i, col j) = V[j*4+i]
    EwaCtx ctx;
    float a, b, c;      // blurred cov2D
    float det;
    float fx, fy;
};

__device__ __forceinline__ bool inria_geom(const InriaCam& cam, const float p[3], const float* S6_in,
                                           int width, int height, float tanfovx, float tanfovy, InriaGeom& G) {
#pragma unroll
    for (int c = 0; c < 3; ++c) G.pv[c] = p[0] * cam.V[0 * 4 + c] + p[1] * cam.V[1 * 4 + c] + p[2] * cam.V[2 * 4 + c] + cam.V[12 + c];
    if (G.pv[2] <= 0.2f) return false;
#pragma unroll
    for (int c = 0; c < 4; ++c) G.hom[c] = p[0] * cam.P[0 * 4 + c] + p[1] * cam.P[1 * 4 + c] + p[2] * cam.P[2 * 4 + c] + cam.P[12 + c];
#pragma unroll
    for (int k = 0; k < 6; ++k) G.S6[k] = S6_in[k];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) G.Wstd[i * 3 + j] = cam.V[j * 4 + i];
    G.fx = (float)width / (2.f * tanfovx);
    G.fy = (float)height / (2.f * tanfovy);
    float a0, b0, c0;
    ewa_fwd(G.pv, G.S6, G.Wstd, G.fx, G.fy, 1.3f * tanfovx, 1.3f * tanfovy, a0, b0, c0, G.ctx);
    G.a = a0 + 0.3f; G.b = b0; G.c = c0 + 0.3f;
    G.det = G.a * G.c - G.b * G.b;
    return G.det != 0.f;
}

// The model's activations, for the RAW-parameter form of the fused call (GSPL_INRIA_RAW_PARAMS): what the reference's model applies
// in torch before every render and differentiates after every backward — `scale_activation` = exp, `rotation_activation` =
// F.normalize (x / max(|x|, 1e-12)), `opacity_activation` = sigmoid (internal/models/vanilla_gaussian.py:345-358; ten elementwise
// launches and a reduction per step at 1 M Gaussians, profiles/r05f_loop_sequence_torch_activations.txt) — evaluated where the parameters are read.
// expf / IEEE division: the arithmetic of torch.exp / torch.sigmoid.
__device__ __forceinline__ float act_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float act_quat_norm(const float q[4]) { return fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f); }

template <bool RAW>
__global__ __launch_bounds__(256) void inria_preprocess_fwd_kernel(
    int N,
    const float* __restrict__ means, const float* __restrict__ scales, const float* __restrict__ quats,
    const float* __restrict__ cov3d_precomp,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    int width, int height, int tile_size, float tanfovx, float tanfovy, float scale_modifier,
    int32_t* __restrict__ radii, float* __restrict__ means2d, float* __restrict__ depths,
    float* __restrict__ conics, float* __restrict__ cov3d,
    const float* __restrict__ raw_opacities, float* __restrict__ opacities_out, uint4* __restrict__ zero_p, uint32_t zero_n16) {
    zero_table(zero_p, zero_n16);      // tables of the depth sort that follows (ZeroJob; nothing to do with this kernel's rows)
    // rows of this block in LDS: inputs means[3] | scales[3] quats[4] (or cov3d_precomp[6]); then, over the same memory, the outputs
    // cov3d[6] | means2d[2] | conics[3]
    __shared__ __attribute__((aligned(16))) float s_buf[256 * 11];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    const int rows = (int)min((int64_t)256, (int64_t)N - row0);
    const int g = (int)row0 + t;
    float* s_means = s_buf;
    float* s_a = s_buf + 256 * 3;            // scales, or cov3d_precomp
    float* s_q = s_buf + 256 * 6;
    rows_to_lds<256>(means, row0, rows, 3, s_means);
    if (cov3d_precomp) rows_to_lds<256>(cov3d_precomp, row0, rows, 6, s_a);
    else { rows_to_lds<256>(scales, row0, rows, 3, s_a); rows_to_lds<256>(quats, row0, rows, 4, s_q); }
    __syncthreads();

    float S6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int o_radius = 0;
    float o_xy[2] = {0.f, 0.f}, o_depth = 0.f, o_conic[3] = {0.f, 0.f, 0.f};
    if (t < rows) {
        InriaCam cam;
        load_inria_cam(viewmatrix, projmatrix, cam);
        const float p[3] = {s_means[t * 3 + 0], s_means[t * 3 + 1], s_means[t * 3 + 2]};
        if (cov3d_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) S6[k] = s_a[t * 6 + k];
        } else {
            float s[3] = {s_a[t * 3 + 0], s_a[t * 3 + 1], s_a[t * 3 + 2]};
            const float4 qv = *reinterpret_cast<const float4*>(s_q + t * 4);
            float q[4] = {qv.x, qv.y, qv.z, qv.w};
            if constexpr (RAW) {
#pragma unroll
                for (int j = 0; j < 3; ++j) s[j] = expf(s[j]);
                const float inv = 1.f / act_quat_norm(q);
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] *= inv;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) s[j] *= scale_modifier;
            float R[9];
            quat_to_rotmat(q, R);
            cov3d_from_scale_rot(s, R, S6);
        }
        InriaGeom G;
        if (inria_geom(cam, p, S6, width, height, tanfovx, tanfovy, G)) {
            const float inv_det = 1.f / G.det;
            const float mid = 0.5f * (G.a + G.c);
            const float lambda = mid + sqrtf(fmaxf(0.1f, mid * mid - G.det));
            const int radius = (int)ceilf(3.f * sqrtf(lambda));
            const float pw = 1.f / (G.hom[3] + 1e-7f);
            const float x2d = ((G.hom[0] * pw + 1.f) * (float)width - 1.f) * 0.5f;
            const float y2d = ((G.hom[1] * pw + 1.f) * (float)height - 1.f) * 0.5f;
            const int grid_x = (width + tile_size - 1) / tile_size, grid_y = (height + tile_size - 1) / tile_size;
            const float ts = (float)tile_size, rf = (float)radius;
            const int minx = min(grid_x, max(0, (int)((x2d - rf) / ts)));
            const int miny = min(grid_y, max(0, (int)((y2d - rf) / ts)));
            const int maxx = min(grid_x, max(0, (int)((x2d + rf + ts - 1.f) / ts)));
            const int maxy = min(grid_y, max(0, (int)((y2d + rf + ts - 1.f) / ts)));
            if ((maxx - minx) * (maxy - miny) > 0) {
                o_radius = radius;
                o_xy[0] = x2d; o_xy[1] = y2d;
                o_depth = G.pv[2];
                o_conic[0] = G.c * inv_det; o_conic[1] = -G.b * inv_det; o_conic[2] = G.a * inv_det;
            }
        }
        radii[g] = o_radius;            // 4-byte columns: already one full line per 32 lanes
        depths[g] = o_depth;
        if constexpr (RAW) opacities_out[g] = act_sigmoid(raw_opacities[g]);
    }
    __syncthreads();                    // every lane has read its input rows: the memory now takes the output rows
    float* s_cov = s_buf;
    float* s_xy = s_buf + 256 * 6;
    float* s_con = s_buf + 256 * 8;
    if (t < rows) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s_cov[t * 6 + k] = S6[k];
        s_xy[t * 2 + 0] = o_xy[0]; s_xy[t * 2 + 1] = o_xy[1];
        s_con[t * 3 + 0] = o_conic[0]; s_con[t * 3 + 1] = o_conic[1]; s_con[t * 3 + 2] = o_conic[2];
    }
    __syncthreads();
    lds_to_rows<256>(cov3d, row0, rows, 6, s_cov);
    lds_to_rows<256>(means2d, row0, rows, 2, s_xy);
    lds_to_rows<256>(conics, row0, rows, 3, s_con);
}

// ACCUM: v_means already holds dL/d(dir) from the SH backward and is accumulated into.
// (Measured, round 3: the LDS-staged form of this kernel — every array copied through LDS as 16-byte rows, as the forward does —
// is SLOWER, 40.5 against 34.5 us at 1 M Gaussians: 31 KB of LDS per block and three barriers cost more occupancy than the
// strided loads cost bandwidth; the kernel carries ~400 flops per Gaussian between its loads and its stores.)
// RAW: `scales` / `quats` are the raw parameters and v_scales / v_quats / v_opac_dst their gradients (chain rule of exp, normalize,
// sigmoid: v s, (v - q (q.v)) / |raw|, v o (1 - o)); `opac_act` = the activated opacities the forward stored.
// ADAM: v_means is SCRATCH (the SH backward's direction gradient, read, not written), v_scales / v_quats / v_opac_dst are not written:
// the kernel applies the Adam update to the rows of means, scales, rotations and opacities it has just produced the gradient of
// (`adam`: parameter, moments, hyper-parameters; every row, the invisible ones with a zero gradient, as torch.optim.Adam does).
// The parameters are read and written through `adam.*.p` (no __restrict__ promise on memory this kernel writes).
struct PreAdam { AdamTarget means, scales, quats, opac; };
// Adam over `total` consecutive floats of one parameter starting at float `base` (a block's rows), gradients in LDS: 16-byte chunks
// per lane, coalesced; a slice that is not 16-byte aligned, and the last partial chunk, go element by element.
__device__ __forceinline__ void pre_adam_pass(const AdamTarget& T, int64_t base, int total, const float* grad) {
    const int t = threadIdx.x;
    float* gp = T.p + base;
    float* gm = T.m + base;
    float* gv = T.v + base;
    int done = 0;
    if ((((uintptr_t)gp | (uintptr_t)gm | (uintptr_t)gv) & 15u) == 0) {
        const int n4 = total >> 2;
        for (int e4 = t; e4 < n4; e4 += 256) {
            float4 p = reinterpret_cast<const float4*>(gp)[e4], m = reinterpret_cast<const float4*>(gm)[e4], v = reinterpret_cast<const float4*>(gv)[e4];
            const float4 g = reinterpret_cast<const float4*>(grad)[e4];
            adam_elem(p.x, g.x, m.x, v.x, T.h);
            adam_elem(p.y, g.y, m.y, v.y, T.h);
            adam_elem(p.z, g.z, m.z, v.z, T.h);
            adam_elem(p.w, g.w, m.w, v.w, T.h);
            reinterpret_cast<float4*>(gp)[e4] = p;
            reinterpret_cast<float4*>(gm)[e4] = m;
            reinterpret_cast<float4*>(gv)[e4] = v;
        }
        done = n4 << 2;
    }
    for (int e = done + t; e < total; e += 256) {
        float p = gp[e], m = gm[e], v = gv[e];
        adam_elem(p, grad[e], m, v, T.h);
        gp[e] = p; gm[e] = m; gv[e] = v;
    }
}
template <bool ACCUM, bool RAW, bool ADAM = false>
__global__ __launch_bounds__(256) void inria_preprocess_bwd_kernel(
    int N,
    const float* means, const float* scales, const float* quats,
    const float* __restrict__ cov3d,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    int width, int height, float tanfovx, float tanfovy, float scale_modifier,
    const int32_t* __restrict__ radii,
    const float* __restrict__ v_means2d, const float* __restrict__ v_conics, int gs2, int gs3,
    float* __restrict__ v_means, float* __restrict__ v_scales, float* __restrict__ v_quats,
    float* __restrict__ v_cov3d_precomp, float* __restrict__ v_means2d_ndc,
    const float* __restrict__ v_opac_src, float* __restrict__ v_opac_dst, const float* __restrict__ opac_act, PreAdam adam, BwdStats stats) {
    // ADAM: the block's 256 rows of gradients meet in LDS and the update runs as FLAT, coalesced 16-byte passes over the block's
    // slice of every array (pre_adam_pass) — a lane-per-row update reads and writes 33 strided dwords per Gaussian three times over
    // (95 us at 1 M, 527 us at 6 M: 3 TB/s); every thread of the block stays for the barrier, rows past N carry nothing.
    __shared__ __attribute__((aligned(16))) float s_grad[ADAM ? 256 * 11 : 4];      // means 768 | scales 768 | quats 1024 | opacities 256
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (!ADAM && g >= N) return;
    const bool row = g < N;
    if (ADAM || v_opac_dst) {
        float v = row ? v_opac_src[(int64_t)g * gs2] : 0.f;
        if constexpr (RAW) { const float o = row ? opac_act[g] : 0.f; v *= o * (1.f - o); }
        if constexpr (ADAM) s_grad[256 * 10 + threadIdx.x] = v;
        else v_opac_dst[g] = v;
    }
    float vp[3] = {0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f};
    float G6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ndc[2] = {0.f, 0.f};
    if (row && radii[g] > 0) {
        InriaCam cam;
        load_inria_cam(viewmatrix, projmatrix, cam);
        const float p[3] = {means[g * 3 + 0], means[g * 3 + 1], means[g * 3 + 2]};
        float S6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) S6[k] = cov3d[g * 6 + k];
        InriaGeom G;
        inria_geom(cam, p, S6, width, height, tanfovx, tanfovy, G);

        // 2D mean (pixels) -> clip space
        ndc[0] = v_means2d[(int64_t)g * gs2 + 0] * 0.5f * (float)width;
        ndc[1] = v_means2d[(int64_t)g * gs2 + 1] * 0.5f * (float)height;
        // the density controller's statistics of this row (gspl_densify_stats with radii > 0 as the mask, no scale, this gradient as
        // `grad`: the same three lines, the same arithmetic — density.hip) when the caller handed its buffers to the backward
        if (stats.accum) {
            stats.accum[g] += sqrtf(fmaf(ndc[0], ndc[0], ndc[1] * ndc[1]));
            stats.denom[g] += 1.f;
            if (stats.max_radii) stats.max_radii[g] = fmaxf(stats.max_radii[g], (float)radii[g]);
        }
        const float mw = 1.f / (G.hom[3] + 1e-7f);
        const float vh0 = ndc[0] * mw, vh1 = ndc[1] * mw;
        const float vh3 = -(ndc[0] * G.hom[0] + ndc[1] * G.hom[1]) * mw * mw;
#pragma unroll
        for (int r = 0; r < 3; ++r) vp[r] = cam.P[r * 4 + 0] * vh0 + cam.P[r * 4 + 1] * vh1 + cam.P[r * 4 + 3] * vh3;

        // conic -> cov2D -> (view-space mean, cov3D)
        float va, vb, vc;
        conic_bwd(G.a, G.b, G.c, v_conics[(int64_t)g * gs3 + 0], v_conics[(int64_t)g * gs3 + 1], v_conics[(int64_t)g * gs3 + 2], va, vb, vc);
        float vpv[3] = {0.f, 0.f, 0.f};
        ewa_bwd<false>(G.pv, S6, G.Wstd, G.fx, G.fy, G.ctx, va, vb, vc, vpv, G6);
#pragma unroll
        for (int r = 0; r < 3; ++r) vp[r] += cam.V[r * 4 + 0] * vpv[0] + cam.V[r * 4 + 1] * vpv[1] + cam.V[r * 4 + 2] * vpv[2];

        if (ADAM || v_scales) {
            float s[3] = {scales[g * 3 + 0], scales[g * 3 + 1], scales[g * 3 + 2]};
            float q[4] = {quats[g * 4 + 0], quats[g * 4 + 1], quats[g * 4 + 2], quats[g * 4 + 3]};
            float act_s[3] = {1.f, 1.f, 1.f}, inv_norm = 1.f;
            if constexpr (RAW) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { act_s[j] = expf(s[j]); s[j] = act_s[j]; }
                inv_norm = 1.f / act_quat_norm(q);
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] *= inv_norm;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) s[j] *= scale_modifier;
            cov3d_bwd(s, q, G6, vs, vq);
#pragma unroll
            for (int j = 0; j < 3; ++j) vs[j] *= scale_modifier;
            if constexpr (RAW) {
#pragma unroll
                for (int j = 0; j < 3; ++j) vs[j] *= act_s[j];
                const float qv = q[0] * vq[0] + q[1] * vq[1] + q[2] * vq[2] + q[3] * vq[3];
#pragma unroll
                for (int j = 0; j < 4; ++j) vq[j] = (vq[j] - q[j] * qv) * inv_norm;
            }
        }
    }
    if constexpr (ADAM) {
        // the gradient rows never reach HBM: parameter + moments are read, updated, written (same sums, in the same order, as the
        // gradient-writing form below: bit-identical parameters for identical inputs)
        const int t = threadIdx.x;
#pragma unroll
        for (int j = 0; j < 3; ++j) s_grad[t * 3 + j] = row ? (ACCUM ? v_means[g * 3 + j] + vp[j] : vp[j]) : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) s_grad[256 * 3 + t * 3 + j] = vs[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) s_grad[256 * 6 + t * 4 + j] = vq[j];
        if (row) { v_means2d_ndc[g * 3 + 0] = ndc[0]; v_means2d_ndc[g * 3 + 1] = ndc[1]; v_means2d_ndc[g * 3 + 2] = 0.f; }
        __syncthreads();
        const int n0 = blockIdx.x * 256, rows = min(256, N - n0);
        pre_adam_pass(adam.means, (int64_t)n0 * 3, rows * 3, s_grad);
        pre_adam_pass(adam.scales, (int64_t)n0 * 3, rows * 3, s_grad + 256 * 3);
        pre_adam_pass(adam.quats, (int64_t)n0 * 4, rows * 4, s_grad + 256 * 6);
        pre_adam_pass(adam.opac, (int64_t)n0, rows, s_grad + 256 * 10);
        return;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (ACCUM) v_means[g * 3 + j] += vp[j];
        else v_means[g * 3 + j] = vp[j];
    }
    if (v_scales) {
#pragma unroll
        for (int j = 0; j < 3; ++j) v_scales[g * 3 + j] = vs[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) v_quats[g * 4 + j] = vq[j];
    }
    if (v_cov3d_precomp) {
        v_cov3d_precomp[g * 6 + 0] = G6[0]; v_cov3d_precomp[g * 6 + 1] = 2.f * G6[1]; v_cov3d_precomp[g * 6 + 2] = 2.f * G6[2];
        v_cov3d_precomp[g * 6 + 3] = G6[3]; v_cov3d_precomp[g * 6 + 4] = 2.f * G6[4]; v_cov3d_precomp[g * 6 + 5] = G6[5];
    }
    v_means2d_ndc[g * 3 + 0] = ndc[0]; v_means2d_ndc[g * 3 + 1] = ndc[1]; v_means2d_ndc[g * 3 + 2] = 0.f;
}

__global__ __launch_bounds__(256) void masked_copy3_kernel(int N, const int32_t* __restrict__ radii,
                                                           const float* __restrict__ src, int src_stride, float* __restrict__ dst,
                                                           uint8_t* __restrict__ clamped) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const bool live = radii[g] > 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dst[g * 3 + c] = live ? src[(int64_t)g * src_stride + c] : 0.f;
        if (clamped) clamped[g * 3 + c] = 0;
    }
}

// geometry phase, activated parameters (raw_opacities == NULL) or the model's raw ones (-> opacities_out [N] = sigmoid)
int inria_geometry_launch(int N, const float* means, const float* scales, const float* quats, const float* cov3d_precomp,
                          const float* viewmatrix, const float* projmatrix, int width, int height, int tile_size,
                          float tanfovx, float tanfovy, float scale_modifier,
                          int32_t* radii, float* means2d, float* depths, float* conics, float* cov3d,
                          const float* raw_opacities, float* opacities_out, hipStream_t s, ZeroJob zero) {
    const int grid = (N + 255) / 256;
    if (raw_opacities) {
        if (cov3d_precomp || !opacities_out) return fail_arg("inria_preprocess_fwd: raw parameters need scales + rotations and room for the opacities");
        hipLaunchKernelGGL(inria_preprocess_fwd_kernel<true>, dim3(grid), dim3(256), 0, s,
                           N, means, scales, quats, cov3d_precomp, viewmatrix, projmatrix, width, height, tile_size,
                           tanfovx, tanfovy, scale_modifier, radii, means2d, depths, conics, cov3d, raw_opacities, opacities_out, zero.p, zero.n16);
    } else {
        hipLaunchKernelGGL(inria_preprocess_fwd_kernel<false>, dim3(grid), dim3(256), 0, s,
                           N, means, scales, quats, cov3d_precomp, viewmatrix, projmatrix, width, height, tile_size,
                           tanfovx, tanfovy, scale_modifier, radii, means2d, depths, conics, cov3d, (const float*)nullptr, (float*)nullptr, zero.p, zero.n16);
    }
    return check_launch("inria_preprocess_fwd");
}

}  // namespace gspl

extern "C" int gspl_inria_preprocess_fwd(int N, int degree, int n_coeffs,
                                         const float* means, const float* scales, const float* quats,
                                         const float* cov3d_precomp, const float* shs, const float* shs_rest, const float* colors_precomp,
                                         const float* viewmatrix, const float* projmatrix, const float* campos,
                                         int width, int height, int tile_size,
                                         float tanfovx, float tanfovy, float scale_modifier,
                                         int32_t* radii, float* means2d, float* depths, float* conics,
                                         float* colors, uint8_t* clamped, float* cov3d, float* sh_jac, int phases, void* stream) {
    using namespace gspl;
    if (N < 0 || width <= 0 || height <= 0 || tile_size <= 0) return fail_arg("inria_preprocess_fwd: bad sizes");
    if ((phases & ~GSPL_INRIA_ALL) || phases == 0) return fail_arg("inria_preprocess_fwd: bad phases");
    if (N == 0) return GSPL_OK;
    if (!means || !viewmatrix || !projmatrix || !radii || !means2d || !depths || !conics || !colors || !clamped || !cov3d)
        return fail_arg("inria_preprocess_fwd: NULL required pointer");
    if (!cov3d_precomp && (!scales || !quats)) return fail_arg("inria_preprocess_fwd: need scales+quats or cov3d_precomp");
    if (!colors_precomp && (!shs || !campos)) return fail_arg("inria_preprocess_fwd: need shs+campos or colors_precomp");
    if (shs && !colors_precomp && (degree < 0 || degree > 4 || n_coeffs < (degree + 1) * (degree + 1)))
        return fail_arg("inria_preprocess_fwd: bad degree / n_coeffs");
    hipStream_t s = (hipStream_t)stream;
    const int grid = (N + 255) / 256;
    if (phases & GSPL_INRIA_GEOMETRY) {
        int rc = inria_geometry_launch(N, means, scales, quats, cov3d_precomp, viewmatrix, projmatrix, width, height, tile_size, tanfovx, tanfovy,
                                       scale_modifier, radii, means2d, depths, conics, cov3d, nullptr, nullptr, s);
        if (rc != GSPL_OK) return rc;
    }
    if (!(phases & GSPL_INRIA_COLOURS)) return GSPL_OK;
    if (colors_precomp) {
        hipLaunchKernelGGL(masked_copy3_kernel, dim3(grid), dim3(256), 0, s, N, radii, colors_precomp, 3, colors, clamped);
        return check_launch("inria_preprocess_fwd(colors_precomp)");
    }
    // shs_rest == NULL: `shs` holds all n_coeffs rows of a Gaussian ([N, n_coeffs, 3]); otherwise `shs` is the DC row ([N, 1, 3]) and
    // `shs_rest` the others ([N, n_coeffs - 1, 3]) — the two parameters of the reference's model, read where they are
    const int stride = 3 * n_coeffs;
    if (shs_rest) return sh_fwd_launch(N, 1, degree, means, campos, shs, 3, shs_rest, stride - 3, nullptr, radii,
                                       GSPL_SH_ADD_HALF_CLAMP, colors, clamped, stream, sh_jac);
    return sh_fwd_launch(N, 1, degree, means, campos, shs, stride, shs + 3, stride, nullptr, radii,
                         GSPL_SH_ADD_HALF_CLAMP, colors, clamped, stream, sh_jac);
}

namespace gspl {
// opac_act != NULL: scales / quats are RAW parameters (see inria_preprocess_fwd_kernel<true>), opac_act the activated opacities
int inria_preprocess_bwd_impl(int N, int degree, int n_coeffs,
                                         const float* means, const float* scales, const float* quats,
                                         const float* cov3d, const float* shs, const float* shs_rest,
                                         const float* viewmatrix, const float* projmatrix, const float* campos,
                                         int width, int height, float tanfovx, float tanfovy, float scale_modifier,
                                         const int32_t* radii, const uint8_t* clamped,
                                         const float* v_means2d, const float* v_conics, const float* v_colors, int grad_stride,
                                         float* v_means, float* v_scales, float* v_quats,
                                         float* v_cov3d_precomp, float* v_shs, float* v_shs_rest, float* v_colors_precomp,
                                         float* v_means2d_ndc, const float* v_opacities_packed, float* v_opacities, const float* sh_jac,
                                         const float* opac_act, void* stream, const gspl_bwd_adam_plan* plan, BwdStats stats) {
    if (N < 0 || width <= 0 || height <= 0) return fail_arg("inria_preprocess_bwd: bad sizes");
    if ((stats.accum == nullptr) != (stats.denom == nullptr) || (stats.max_radii && !stats.accum))
        return fail_arg("inria_preprocess_bwd: the statistics' accum and denom go together (max_radii is optional beside them)");
    if (plan) {
        // Adam inside the backward: v_shs / v_shs_rest / v_scales / v_quats / v_opacities ARE the parameters (means, scales, quats too)
        if (!v_shs || !v_scales || !v_quats || !v_opacities || v_cov3d_precomp || v_colors_precomp || grad_stride <= 0)
            return fail_arg("inria_preprocess_bwd(adam): needs SH coefficients, scales + rotations and the packed gradient buffer");
        if (v_scales != scales || v_quats != quats || v_shs != shs || v_shs_rest != shs_rest)
            return fail_arg("inria_preprocess_bwd(adam): the update targets must be the parameters the backward reads");
        const gspl_bwd_adam_tensor* all[6] = {&plan->means, &plan->scales, &plan->rotations, &plan->opacities, &plan->shs, &plan->shs_rest};
        for (int k = 0; k < 6; ++k) {
            if (k == 5 && !shs_rest) continue;
            if (!all[k]->exp_avg || !all[k]->exp_avg_sq || !(all[k]->bias_correction1 > 0.f) || !(all[k]->bias_correction2_sqrt > 0.f))
                return fail_arg("inria_preprocess_bwd(adam): moments missing or bias corrections not positive (1 = none)");
        }
    }
    if (v_opacities && (!v_opacities_packed || grad_stride <= 0)) return fail_arg("inria_preprocess_bwd: v_opacities needs the packed gradient buffer");
    if (N == 0) return GSPL_OK;
    if (!means || !cov3d || !viewmatrix || !projmatrix || !radii || !v_means2d || !v_conics || !v_colors || !v_means || !v_means2d_ndc)
        return fail_arg("inria_preprocess_bwd: NULL required pointer");
    if ((v_scales == nullptr) != (v_quats == nullptr)) return fail_arg("inria_preprocess_bwd: v_scales and v_quats go together");
    if (v_scales && (!scales || !quats)) return fail_arg("inria_preprocess_bwd: scales/quats missing");
    hipStream_t s = (hipStream_t)stream;
    const int grid = (N + 255) / 256;
    // grad_stride == 0: three dense arrays ([N,2], [N,3], [N,3]); otherwise all three are columns of one packed
    // [N, grad_stride] buffer (gspl_composite_bwd_packed) and share its row stride
    const int gs2 = grad_stride > 0 ? grad_stride : 2, gs3 = grad_stride > 0 ? grad_stride : 3;
    bool accum = false;
    if (v_shs) {
        if (!shs || !campos || !clamped) return fail_arg("inria_preprocess_bwd: shs/campos/clamped missing");
        if (degree < 0 || degree > 4 || n_coeffs < (degree + 1) * (degree + 1)) return fail_arg("inria_preprocess_bwd: bad degree / n_coeffs");
        const int stride = 3 * n_coeffs;
        // dL/d(dir) lands in v_means; the geometry kernel accumulates on top
        if ((shs_rest == nullptr) != (v_shs_rest == nullptr)) return fail_arg("inria_preprocess_bwd: shs_rest and v_shs_rest go together");
        ShAdamHost sh_adam;
        if (plan) { sh_adam.dc = plan->shs; sh_adam.rest = shs_rest ? plan->shs_rest : plan->shs; }
        const ShAdamHost* sha = plan ? &sh_adam : nullptr;
        int rc = shs_rest
            ? sh_bwd_launch(N, 1, degree, n_coeffs, means, campos, shs, 3, shs_rest, stride - 3, nullptr, radii,
                            GSPL_SH_ADD_HALF_CLAMP, clamped, v_colors, gs3, v_shs, v_shs_rest, v_means, stream, sh_jac, sha)
            : sh_bwd_launch(N, 1, degree, n_coeffs, means, campos, shs, stride, shs + 3, stride, nullptr, radii,
                            GSPL_SH_ADD_HALF_CLAMP, clamped, v_colors, gs3, v_shs, v_shs + 3, v_means, stream, sh_jac, sha);
        if (rc != GSPL_OK) return rc;
        accum = true;
    }
    if (v_colors_precomp) {
        hipLaunchKernelGGL(masked_copy3_kernel, dim3(grid), dim3(256), 0, s, N, radii, v_colors, gs3, v_colors_precomp, (uint8_t*)nullptr);
        int rc = check_launch("inria_preprocess_bwd(colors_precomp)");
        if (rc != GSPL_OK) return rc;
    }
    if (opac_act && (!v_scales || !v_opacities || v_cov3d_precomp)) return fail_arg("inria_preprocess_bwd: raw parameters need v_scales, v_quats and v_opacities");
    PreAdam pre = {};
    if (plan) {
        auto target = [](float* p, const gspl_bwd_adam_tensor& t) {
            return AdamTarget{p, t.exp_avg, t.exp_avg_sq, AdamHyper{t.lr * (1.f / t.bias_correction1), t.beta1, t.beta2, 1.f / t.bias_correction2_sqrt, t.eps}};
        };
        pre.means = target(const_cast<float*>(means), plan->means);
        pre.scales = target(v_scales, plan->scales);
        pre.quats = target(v_quats, plan->rotations);
        pre.opac = target(v_opacities, plan->opacities);
    }
#define GSPL_LAUNCH_PRE_BWD(A, R, AD) hipLaunchKernelGGL((inria_preprocess_bwd_kernel<A, R, AD>), dim3(grid), dim3(256), 0, s, \
        N, means, scales, quats, cov3d, viewmatrix, projmatrix, width, height, tanfovx, tanfovy, scale_modifier, \
        radii, v_means2d, v_conics, gs2, gs3, v_means, v_scales, v_quats, v_cov3d_precomp, v_means2d_ndc, v_opacities_packed, v_opacities, opac_act, pre, stats)
    if (plan) { if (opac_act) GSPL_LAUNCH_PRE_BWD(true, true, true); else GSPL_LAUNCH_PRE_BWD(true, false, true); }
    else if (accum) { if (opac_act) GSPL_LAUNCH_PRE_BWD(true, true, false); else GSPL_LAUNCH_PRE_BWD(true, false, false); }
    else { if (opac_act) GSPL_LAUNCH_PRE_BWD(false, true, false); else GSPL_LAUNCH_PRE_BWD(false, false, false); }
#undef GSPL_LAUNCH_PRE_BWD
    return check_launch("inria_preprocess_bwd");
}
}  // namespace gspl

extern "C" int gspl_inria_preprocess_bwd(int N, int degree, int n_coeffs,
                                         const float* means, const float* scales, const float* quats,
                                         const float* cov3d, const float* shs, const float* shs_rest,
                                         const float* viewmatrix, const float* projmatrix, const float* campos,
                                         int width, int height, float tanfovx, float tanfovy, float scale_modifier,
                                         const int32_t* radii, const uint8_t* clamped,
                                         const float* v_means2d, const float* v_conics, const float* v_colors, int grad_stride,
                                         float* v_means, float* v_scales, float* v_quats,
                                         float* v_cov3d_precomp, float* v_shs, float* v_shs_rest, float* v_colors_precomp,
                                         float* v_means2d_ndc, const float* v_opacities_packed, float* v_opacities, const float* sh_jac, void* stream) {
    return gspl::inria_preprocess_bwd_impl(N, degree, n_coeffs, means, scales, quats, cov3d, shs, shs_rest, viewmatrix, projmatrix, campos, width, height,
                                           tanfovx, tanfovy, scale_modifier, radii, clamped, v_means2d, v_conics, v_colors, grad_stride, v_means, v_scales,
                                           v_quats, v_cov3d_precomp, v_shs, v_shs_rest, v_colors_precomp, v_means2d_ndc, v_opacities_packed, v_opacities,
                                           sh_jac, nullptr, stream, nullptr);
}
