// projection.hip — gsplat-convention EWA projection, forward and backward (gfx950).
//
// Replaces gsplat `fully_fused_projection` / v0 `project_gaussians` at the reference call sites
// internal/renderers/gsplat_v1_renderer.py:408-421, gsplat_renderer.py:64-79,
// gsplat_distributed_renderer.py:271-283.  The math is a restatement of the reference's own
// pure-PyTorch projection, internal/utils/gaussian_projection.py:6-138 (the only in-tree source),
// including its constants: z >= near (0.01), +eps2d low-pass with compensation sqrt(det0/det1),
// mean = K (p / (z + 1e-6)), radius = ceil(3 sqrt(mid + sqrt(max(0.1, mid^2 - det)))),
// tile rect [trunc((x-r)/T), trunc((x+r)/T)+1) clamped to the grid.
//
// Roofline: HBM-bound elementwise op; algorithmic bytes 40 B read + 36 B written per
// (camera, Gaussian).  One lane per (camera, Gaussian); AoS inputs are read with
// per-lane strided dword loads — a wave touches one contiguous 768 B / 1 KiB span per array, so
// every fetched line is fully used.
#include "gspl_device.h"
#include "gspl_host.h"

namespace gspl {

struct ProjCam {
    float W[9];
    float t[3];
    float fx, fy, cx, cy;
};

__device__ __forceinline__ void load_cam(const float* __restrict__ viewmats, const float* __restrict__ Ks, int cam, ProjCam& c) {
    const float* V = viewmats + cam * 16;
    const float* K = Ks + cam * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) c.W[i * 3 + j] = V[i * 4 + j];
        c.t[i] = V[i * 4 + 3];
    }
    c.fx = K[0]; c.cx = K[2]; c.fy = K[4]; c.cy = K[5];
}

template <int CAM>
__global__ __launch_bounds__(256) void project_fwd_kernel(
    int C, int N,
    const float* __restrict__ means, const float* __restrict__ scales, const float* __restrict__ quats,
    const float* __restrict__ viewmats, const float* __restrict__ Ks,
    int width, int height, int tile_size,
    float scale_modifier, float eps2d, float near_plane, float far_plane, float radius_clip,
    int32_t* __restrict__ radii, float* __restrict__ means2d, float* __restrict__ depths,
    float* __restrict__ conics, float* __restrict__ compensations, int32_t* __restrict__ tiles_hit, float* __restrict__ cov3d) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int cam = (int)(idx / N);
    const int g = (int)(idx - (int64_t)cam * N);

    ProjCam c;
    load_cam(viewmats, Ks, cam, c);

    const float p[3] = {means[g * 3 + 0], means[g * 3 + 1], means[g * 3 + 2]};
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = c.W[i * 3 + 0] * p[0] + c.W[i * 3 + 1] * p[1] + c.W[i * 3 + 2] * p[2] + c.t[i];

    bool ok = (pc[2] >= near_plane) && (pc[2] <= far_plane);

    float o_xy[2] = {0.f, 0.f}, o_depth = 0.f, o_conic[3] = {0.f, 0.f, 0.f}, o_comp = 0.f;
    float o_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int o_radius = 0, o_tiles = 0;

    if (ok) {
        const float s[3] = {scales[g * 3 + 0] * scale_modifier, scales[g * 3 + 1] * scale_modifier, scales[g * 3 + 2] * scale_modifier};
        const float q[4] = {quats[g * 4 + 0], quats[g * 4 + 1], quats[g * 4 + 2], quats[g * 4 + 3]};
        float R[9], S6[6];
        quat_to_rotmat(q, R);
        cov3d_from_scale_rot(s, R, S6);

        float a0, b0, c0, mx = 0.f, my = 0.f;
        if constexpr (CAM == GSPL_CAM_PINHOLE_) {
            const float limx = 1.3f * (0.5f * (float)width / c.fx);
            const float limy = 1.3f * (0.5f * (float)height / c.fy);
            EwaCtx ctx;
            ewa_fwd(pc, S6, c.W, c.fx, c.fy, limx, limy, a0, b0, c0, ctx);
        } else {
            float J[6], T0[3], T1[3];
            cam_project<CAM>(pc, c.fx, c.fy, J, mx, my);
            ewa_fwd_general(J, S6, c.W, a0, b0, c0, T0, T1);
        }

        const float det0 = a0 * c0 - b0 * b0;
        const float a = a0 + eps2d, cc = c0 + eps2d, b = b0;
        const float det = a * cc - b * b;
        if (det == 0.f) ok = false;   // the reference raises here (gaussian_projection.py:70-71); we cull
        if (ok) {
            const float comp = sqrtf(fmaxf(det0 / det, 0.f));
            const float inv_det = 1.f / det;
            float x2d, y2d;
            if constexpr (CAM == GSPL_CAM_PINHOLE_) {
                const float rz = 1.f / (pc[2] + 1e-6f);
                x2d = c.fx * (pc[0] * rz) + c.cx;
                y2d = c.fy * (pc[1] * rz) + c.cy;
            } else {
                x2d = mx + c.cx;
                y2d = my + c.cy;
            }
            const float mid = 0.5f * (a + cc);
            const float lambda = mid + sqrtf(fmaxf(mid * mid - det, 0.1f));
            const int radius = (int)ceilf(3.f * sqrtf(lambda));

            const int grid_x = (width + tile_size - 1) / tile_size;
            const int grid_y = (height + tile_size - 1) / tile_size;
            const float ts = (float)tile_size;
            const float rf = (float)radius;
            int minx = (int)((x2d - rf) / ts), miny = (int)((y2d - rf) / ts);
            int maxx = (int)((x2d + rf) / ts) + 1, maxy = (int)((y2d + rf) / ts) + 1;
            minx = min(max(minx, 0), grid_x); maxx = min(max(maxx, 0), grid_x);
            miny = min(max(miny, 0), grid_y); maxy = min(max(maxy, 0), grid_y);
            const int ntiles = (maxx - minx) * (maxy - miny);
            if (ntiles <= 0 || rf <= radius_clip) ok = false;
            if (ok) {
                o_xy[0] = x2d; o_xy[1] = y2d; o_depth = pc[2];
                o_conic[0] = cc * inv_det; o_conic[1] = -b * inv_det; o_conic[2] = a * inv_det;
                o_comp = comp; o_radius = radius; o_tiles = ntiles;
#pragma unroll
                for (int k = 0; k < 6; ++k) o_cov[k] = S6[k];
            }
        }
    }
    radii[idx] = o_radius;
    means2d[idx * 2 + 0] = o_xy[0]; means2d[idx * 2 + 1] = o_xy[1];
    depths[idx] = o_depth;
    conics[idx * 3 + 0] = o_conic[0]; conics[idx * 3 + 1] = o_conic[1]; conics[idx * 3 + 2] = o_conic[2];
    if (compensations) compensations[idx] = o_comp;
    if (tiles_hit) tiles_hit[idx] = o_tiles;
    if (cov3d) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov3d[idx * 6 + k] = o_cov[k];
    }
}

template <bool ATOMIC, int CAM>
__global__ __launch_bounds__(256) void project_bwd_kernel(
    int C, int N,
    const float* __restrict__ means, const float* __restrict__ scales, const float* __restrict__ quats,
    const float* __restrict__ viewmats, const float* __restrict__ Ks,
    int width, int height, float scale_modifier, float eps2d,
    const int32_t* __restrict__ radii,
    const float* __restrict__ v_means2d, int s2, const float* __restrict__ v_depths,
    const float* __restrict__ v_conics, int s3, const float* __restrict__ v_compensations,
    float* __restrict__ v_means, float* __restrict__ v_scales, float* __restrict__ v_quats) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int cam = (int)(idx / N);
    const int g = (int)(idx - (int64_t)cam * N);

    float vp[3] = {0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f};

    if (radii[idx] > 0) {
        ProjCam c;
        load_cam(viewmats, Ks, cam, c);
        const float p[3] = {means[g * 3 + 0], means[g * 3 + 1], means[g * 3 + 2]};
        float pc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pc[i] = c.W[i * 3 + 0] * p[0] + c.W[i * 3 + 1] * p[1] + c.W[i * 3 + 2] * p[2] + c.t[i];
        const float s[3] = {scales[g * 3 + 0] * scale_modifier, scales[g * 3 + 1] * scale_modifier, scales[g * 3 + 2] * scale_modifier};
        const float q[4] = {quats[g * 4 + 0], quats[g * 4 + 1], quats[g * 4 + 2], quats[g * 4 + 3]};
        float R[9], S6[6];
        quat_to_rotmat(q, R);
        cov3d_from_scale_rot(s, R, S6);
        float a0, b0, c0;
        EwaCtx ctx;
        float T0[3], T1[3];
        if constexpr (CAM == GSPL_CAM_PINHOLE_) {
            const float limx = 1.3f * (0.5f * (float)width / c.fx);
            const float limy = 1.3f * (0.5f * (float)height / c.fy);
            ewa_fwd(pc, S6, c.W, c.fx, c.fy, limx, limy, a0, b0, c0, ctx);
        } else {
            float J[6], mx, my;
            cam_project<CAM>(pc, c.fx, c.fy, J, mx, my);
            ewa_fwd_general(J, S6, c.W, a0, b0, c0, T0, T1);
        }
        const float a = a0 + eps2d, cc = c0 + eps2d, b = b0;
        const float det0 = a0 * c0 - b0 * b0;
        const float det = a * cc - b * b;

        // conic -> (a, b, c)
        float va, vb, vc;
        conic_bwd(a, b, cc, v_conics[idx * s3 + 0], v_conics[idx * s3 + 1], v_conics[idx * s3 + 2], va, vb, vc);
        // compensation = sqrt(max(det0/det, 0))
        if (v_compensations) {
            const float ratio = det0 / det;
            if (ratio > 0.f) {
                const float vr = v_compensations[idx] * 0.5f / sqrtf(ratio);
                const float rd2 = 1.f / (det * det);
                va += vr * (c0 * det - det0 * cc) * rd2;
                vc += vr * (a0 * det - det0 * a) * rd2;
                vb += vr * 2.f * b * (det0 - det) * rd2;
            }
        }
        float vpc[3] = {0.f, 0.f, 0.f};
        float G6[6];
        const float vx2 = v_means2d[idx * s2 + 0], vy2 = v_means2d[idx * s2 + 1];
        if constexpr (CAM == GSPL_CAM_PINHOLE_) {
            ewa_bwd<true>(pc, S6, c.W, c.fx, c.fy, ctx, va, vb, vc, vpc, G6);
            // 2D mean: x2d = fx * x / (z + 1e-6) + cx
            const float rz = 1.f / (pc[2] + 1e-6f);
            vpc[0] += vx2 * c.fx * rz;
            vpc[1] += vy2 * c.fy * rz;
            vpc[2] += -(vx2 * c.fx * pc[0] + vy2 * c.fy * pc[1]) * rz * rz;
        } else {
            float vJ[6];
            ewa_bwd_general(S6, c.W, T0, T1, va, vb, vc, vJ, G6);
            cam_project_bwd<CAM>(pc, c.fx, c.fy, vJ, vx2, vy2, vpc);
        }
        if (v_depths) vpc[2] += v_depths[idx];
        // p_c = W p + t
#pragma unroll
        for (int j = 0; j < 3; ++j) vp[j] = c.W[0 * 3 + j] * vpc[0] + c.W[1 * 3 + j] * vpc[1] + c.W[2 * 3 + j] * vpc[2];
        cov3d_bwd(s, q, G6, vs, vq);
#pragma unroll
        for (int j = 0; j < 3; ++j) vs[j] *= scale_modifier;
    }
    if (ATOMIC) {
        if (radii[idx] > 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { atomicAdd(&v_means[g * 3 + j], vp[j]); atomicAdd(&v_scales[g * 3 + j], vs[j]); }
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(&v_quats[g * 4 + j], vq[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) { v_means[g * 3 + j] = vp[j]; v_scales[g * 3 + j] = vs[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) v_quats[g * 4 + j] = vq[j];
    }
}

}  // namespace gspl

extern "C" int gspl_project_fwd(int C, int N,
                                const float* means, const float* scales, const float* quats,
                                const float* viewmats, const float* Ks,
                                int width, int height, int tile_size,
                                float scale_modifier, float eps2d, float near_plane, float far_plane,
                                float radius_clip, int camera_model,
                                int32_t* radii, float* means2d, float* depths, float* conics,
                                float* compensations, int32_t* tiles_hit, float* cov3d, void* stream) {
    if (C < 0 || N < 0 || width <= 0 || height <= 0 || tile_size <= 0) return gspl::fail_arg("project_fwd: bad sizes");
    if (camera_model < GSPL_CAMERA_PINHOLE || camera_model > GSPL_CAMERA_FISHEYE) return gspl::fail_arg("project_fwd: unknown camera model");
    if ((int64_t)C * N == 0) return GSPL_OK;
    if (!means || !scales || !quats || !viewmats || !Ks || !radii || !means2d || !depths || !conics)
        return gspl::fail_arg("project_fwd: NULL required pointer");
    const int64_t total = (int64_t)C * N;
    const int block = 256;
    const int64_t grid = (total + block - 1) / block;
#define GSPL_PROJECT_FWD(CAM)                                                                                              \
    hipLaunchKernelGGL(gspl::project_fwd_kernel<CAM>, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream, C, N, means,   \
                       scales, quats, viewmats, Ks, width, height, tile_size, scale_modifier, eps2d, near_plane, far_plane,   \
                       radius_clip, radii, means2d, depths, conics, compensations, tiles_hit, cov3d)
    if (camera_model == GSPL_CAMERA_ORTHO) GSPL_PROJECT_FWD(gspl::GSPL_CAM_ORTHO_);
    else if (camera_model == GSPL_CAMERA_FISHEYE) GSPL_PROJECT_FWD(gspl::GSPL_CAM_FISHEYE_);
    else GSPL_PROJECT_FWD(gspl::GSPL_CAM_PINHOLE_);
#undef GSPL_PROJECT_FWD
    return gspl::check_launch("project_fwd");
}

extern "C" int gspl_project_bwd(int C, int N,
                                const float* means, const float* scales, const float* quats,
                                const float* viewmats, const float* Ks,
                                int width, int height, float scale_modifier, float eps2d, int camera_model,
                                const int32_t* radii,
                                const float* v_means2d, int v_means2d_stride, const float* v_depths,
                                const float* v_conics, int v_conics_stride,
                                const float* v_compensations,
                                float* v_means, float* v_scales, float* v_quats, void* stream) {
    if (C < 0 || N < 0 || width <= 0 || height <= 0) return gspl::fail_arg("project_bwd: bad sizes");
    if (camera_model < GSPL_CAMERA_PINHOLE || camera_model > GSPL_CAMERA_FISHEYE) return gspl::fail_arg("project_bwd: unknown camera model");
    if ((int64_t)C * N == 0) return GSPL_OK;
    if (!means || !scales || !quats || !viewmats || !Ks || !radii || !v_means2d || !v_conics ||
        !v_means || !v_scales || !v_quats)
        return gspl::fail_arg("project_bwd: NULL required pointer");
    const int64_t total = (int64_t)C * N;
    const int block = 256;
    const int64_t grid = (total + block - 1) / block;
    const int s2 = v_means2d_stride > 0 ? v_means2d_stride : 2, s3 = v_conics_stride > 0 ? v_conics_stride : 3;
    if ((s2 != 2 || s3 != 3) && C != 1) return gspl::fail_arg("project_bwd: strided gradients need C == 1");
#define GSPL_PROJECT_BWD(ATOMIC, CAM)                                                                                      \
    hipLaunchKernelGGL((gspl::project_bwd_kernel<ATOMIC, CAM>), dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream, C, N, \
                       means, scales, quats, viewmats, Ks, width, height, scale_modifier, eps2d, radii, v_means2d, s2,        \
                       v_depths, v_conics, s3, v_compensations, v_means, v_scales, v_quats)
#define GSPL_PROJECT_BWD_CAM(CAM) do { if (C == 1) GSPL_PROJECT_BWD(false, CAM); else GSPL_PROJECT_BWD(true, CAM); } while (0)
    if (camera_model == GSPL_CAMERA_ORTHO) GSPL_PROJECT_BWD_CAM(gspl::GSPL_CAM_ORTHO_);
    else if (camera_model == GSPL_CAMERA_FISHEYE) GSPL_PROJECT_BWD_CAM(gspl::GSPL_CAM_FISHEYE_);
    else GSPL_PROJECT_BWD_CAM(gspl::GSPL_CAM_PINHOLE_);
#undef GSPL_PROJECT_BWD_CAM
#undef GSPL_PROJECT_BWD
    return gspl::check_launch("project_bwd");
}
