// gspl_device.h — device-side math shared by the projection / preprocess kernels (gfx950).
//
// Everything here is per-Gaussian scalar fp32 math executed by one lane; the kernels that use it
// are HBM-bound, so the code favours clarity over instruction count.
//
// Reference restated (not copied): internal/utils/gaussian_projection.py:211-287
// (build_rotation_matrix, compute_cov_3d, compute_cov_2d).  The backward formulas are derived by
// hand from those forward definitions and are checked against fp64 autograd of
// the oracle restatement in tests/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gspl_hip.h"   // GSPL_MODE_* / GSPL_LAYOUT_* enums

namespace gspl {

// ---- per-API constants -------------------------------------------------
template <int MODE> struct ModeTraits;
template <> struct ModeTraits<GSPL_MODE_GSPLAT> {
    static constexpr float kAlphaMax = 0.999f;
    static constexpr float kPixelCentre = 0.5f;
    static constexpr bool kStopInclusive = true;    // stop when next_T <= 1e-4
    static constexpr bool kClampKillsGrad = true;   // d min(amax, x)/dx = 0 when clamped
};
template <> struct ModeTraits<GSPL_MODE_INRIA> {
    static constexpr float kAlphaMax = 0.99f;
    static constexpr float kPixelCentre = 0.0f;
    static constexpr bool kStopInclusive = false;   // stop when test_T < 1e-4
    static constexpr bool kClampKillsGrad = false;  // Inria backward ignores the clamp
};
static constexpr float kAlphaMin = 1.0f / 255.0f;
static constexpr float kTStop = 1e-4f;

// ---- rotation from quaternion (w,x,y,z), used as given (gaussian_projection.py:211-232) --------
__device__ __forceinline__ void quat_to_rotmat(const float q[4], float R[9]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z);       R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y);       R[7] = 2.f * (y * z + w * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = (R S)(R S)^T, S = diag(s).  Returns the 6 unique entries (xx, xy, xz, yy, yz, zz)
// (same order as the reference's cov3D_precomp, internal/utils/general_utils.py:126-139).
__device__ __forceinline__ void cov3d_from_scale_rot(const float s[3], const float R[9], float S6[6]) {
    float M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = R[i * 3 + j] * s[j];
    S6[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    S6[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    S6[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    S6[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    S6[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    S6[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

// G6 = dL/dSigma as a full symmetric matrix (each off-diagonal *position* holds G_ij, so the
// derivative w.r.t. the single parameter sigma_ij, i != j, is 2*G_ij).
// v_M = 2 G M ; M = R S  ->  v_s[j] = sum_i v_M[i][j] R[i][j],  v_R[i][j] = v_M[i][j] s[j].
__device__ __forceinline__ void cov3d_bwd(const float s[3], const float q[4], const float G6[6],
                                          float v_s[3], float v_q[4]) {
    float R[9];
    quat_to_rotmat(q, R);
    const float G[9] = {G6[0], G6[1], G6[2], G6[1], G6[3], G6[4], G6[2], G6[4], G6[5]};
    float vR[9];
    v_s[0] = v_s[1] = v_s[2] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // v_M[i][j] = 2 * sum_k G[i][k] * M[k][j],  M[k][j] = R[k][j]*s[j]
            const float vm = 2.f * s[j] * (G[i * 3 + 0] * R[0 * 3 + j] + G[i * 3 + 1] * R[1 * 3 + j] + G[i * 3 + 2] * R[2 * 3 + j]);
            v_s[j] += vm * R[i * 3 + j];
            vR[i * 3 + j] = vm * s[j];
        }
    }
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    v_q[0] = 2.f * (-z * vR[1] + y * vR[2] + z * vR[3] - x * vR[5] - y * vR[6] + x * vR[7]);
    v_q[1] = 2.f * (y * vR[1] + z * vR[2] + y * vR[3] - 2.f * x * vR[4] - w * vR[5] + z * vR[6] + w * vR[7] - 2.f * x * vR[8]);
    v_q[2] = 2.f * (-2.f * y * vR[0] + x * vR[1] + w * vR[2] + x * vR[3] + z * vR[5] - w * vR[6] + z * vR[7] - 2.f * y * vR[8]);
    v_q[3] = 2.f * (-2.f * z * vR[0] - w * vR[1] + x * vR[2] + w * vR[3] - 2.f * z * vR[4] + y * vR[5] + x * vR[6] + y * vR[7]);
}

// ---- EWA: cov2d = T Sigma T^T with T = J W (gaussian_projection.py:257-287) ---------------------
// pc: camera-space mean; W: 3x3 world->camera rotation (row-major, standard orientation);
// limx/limy = 1.3 * tan(fov/2).  Outputs the *un-blurred* 2x2 entries (a0, b0, c0).
struct EwaCtx {
    float T0[3], T1[3];   // rows of T
    float tx, ty;         // clamped x, y used in J
    bool  in_x, in_y;     // clamp inactive (gradient passes to x / y)
    float ux, uy;         // clamped x/z, y/z
};

__device__ __forceinline__ void ewa_fwd(const float pc[3], const float S6[6], const float W[9],
                                        float fx, float fy, float limx, float limy,
                                        float& a0, float& b0, float& c0, EwaCtx& ctx) {
    const float z = pc[2];
    const float rz = 1.f / z;
    const float txtz = pc[0] * rz, tytz = pc[1] * rz;
    ctx.in_x = (txtz >= -limx) && (txtz <= limx);
    ctx.in_y = (tytz >= -limy) && (tytz <= limy);
    ctx.ux = fminf(limx, fmaxf(-limx, txtz));
    ctx.uy = fminf(limy, fmaxf(-limy, tytz));
    ctx.tx = ctx.ux * z;
    ctx.ty = ctx.uy * z;
    const float J00 = fx * rz, J02 = -fx * ctx.tx * rz * rz;
    const float J11 = fy * rz, J12 = -fy * ctx.ty * rz * rz;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ctx.T0[j] = J00 * W[0 * 3 + j] + J02 * W[2 * 3 + j];
        ctx.T1[j] = J11 * W[1 * 3 + j] + J12 * W[2 * 3 + j];
    }
    // Sigma * T0^T, Sigma * T1^T
    const float s0x = S6[0] * ctx.T0[0] + S6[1] * ctx.T0[1] + S6[2] * ctx.T0[2];
    const float s0y = S6[1] * ctx.T0[0] + S6[3] * ctx.T0[1] + S6[4] * ctx.T0[2];
    const float s0z = S6[2] * ctx.T0[0] + S6[4] * ctx.T0[1] + S6[5] * ctx.T0[2];
    const float s1x = S6[0] * ctx.T1[0] + S6[1] * ctx.T1[1] + S6[2] * ctx.T1[2];
    const float s1y = S6[1] * ctx.T1[0] + S6[3] * ctx.T1[1] + S6[4] * ctx.T1[2];
    const float s1z = S6[2] * ctx.T1[0] + S6[4] * ctx.T1[1] + S6[5] * ctx.T1[2];
    a0 = ctx.T0[0] * s0x + ctx.T0[1] * s0y + ctx.T0[2] * s0z;
    b0 = ctx.T0[0] * s1x + ctx.T0[1] * s1y + ctx.T0[2] * s1z;
    c0 = ctx.T1[0] * s1x + ctx.T1[1] * s1y + ctx.T1[2] * s1z;
}

// Backward of ewa_fwd.  (va, vb, vc) = dL/d(a0, b0, c0) with vb the derivative w.r.t. the single
// off-diagonal parameter.  Accumulates into v_pc; writes G6 (see cov3d_bwd).
// EXACT_CLAMP: true  -> exact derivative of clamp(x/z)*z (autograd of the reference Python);
//              false -> Inria behaviour: clamped coordinate contributes no gradient at all.
template <bool EXACT_CLAMP>
__device__ __forceinline__ void ewa_bwd(const float pc[3], const float S6[6], const float W[9],
                                        float fx, float fy, const EwaCtx& ctx,
                                        float va, float vb, float vc,
                                        float v_pc[3], float G6[6]) {
    const float* T0 = ctx.T0;
    const float* T1 = ctx.T1;
    const float hb = 0.5f * vb;
    G6[0] = va * T0[0] * T0[0] + vb * T0[0] * T1[0] + vc * T1[0] * T1[0];
    G6[3] = va * T0[1] * T0[1] + vb * T0[1] * T1[1] + vc * T1[1] * T1[1];
    G6[5] = va * T0[2] * T0[2] + vb * T0[2] * T1[2] + vc * T1[2] * T1[2];
    G6[1] = va * T0[0] * T0[1] + hb * (T0[0] * T1[1] + T1[0] * T0[1]) + vc * T1[0] * T1[1];
    G6[2] = va * T0[0] * T0[2] + hb * (T0[0] * T1[2] + T1[0] * T0[2]) + vc * T1[0] * T1[2];
    G6[4] = va * T0[1] * T0[2] + hb * (T0[1] * T1[2] + T1[1] * T0[2]) + vc * T1[1] * T1[2];

    // Sigma*T0, Sigma*T1
    const float s0[3] = {S6[0] * T0[0] + S6[1] * T0[1] + S6[2] * T0[2],
                         S6[1] * T0[0] + S6[3] * T0[1] + S6[4] * T0[2],
                         S6[2] * T0[0] + S6[4] * T0[1] + S6[5] * T0[2]};
    const float s1[3] = {S6[0] * T1[0] + S6[1] * T1[1] + S6[2] * T1[2],
                         S6[1] * T1[0] + S6[3] * T1[1] + S6[4] * T1[2],
                         S6[2] * T1[0] + S6[4] * T1[1] + S6[5] * T1[2]};
    float vT0[3], vT1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        vT0[j] = 2.f * va * s0[j] + vb * s1[j];
        vT1[j] = 2.f * vc * s1[j] + vb * s0[j];
    }
    const float vJ00 = vT0[0] * W[0] + vT0[1] * W[1] + vT0[2] * W[2];
    const float vJ02 = vT0[0] * W[6] + vT0[1] * W[7] + vT0[2] * W[8];
    const float vJ11 = vT1[0] * W[3] + vT1[1] * W[4] + vT1[2] * W[5];
    const float vJ12 = vT1[0] * W[6] + vT1[1] * W[7] + vT1[2] * W[8];

    const float z = pc[2];
    const float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
    float vz = -fx * rz2 * vJ00 - fy * rz2 * vJ11 + 2.f * fx * ctx.tx * rz3 * vJ02 + 2.f * fy * ctx.ty * rz3 * vJ12;
    const float vtx = -fx * rz2 * vJ02;
    const float vty = -fy * rz2 * vJ12;
    if (EXACT_CLAMP) {
        // tx = clamp(x/z) * z
        if (ctx.in_x) v_pc[0] += vtx; else vz += vtx * ctx.ux;
        if (ctx.in_y) v_pc[1] += vty; else vz += vty * ctx.uy;
    } else {
        if (ctx.in_x) v_pc[0] += vtx;
        if (ctx.in_y) v_pc[1] += vty;
    }
    v_pc[2] += vz;
}

// ---- camera models other than the pinhole (gsplat v1 `camera_model`: "ortho", "fisheye"; reference option
//      internal/renderers/gsplat_v1_renderer.py:50,154) ------------------------------------------------------------------
// The fork's kernels are un-vendored; the models are restated from the published gsplat definitions:
//   ortho    mean2d = (fx x + cx, fy y + cy),  J = [[fx, 0, 0], [0, fy, 0]]
//   fisheye  (equidistant)  r = |(x, y)| + eps, theta = atan2(r, z + eps), mean2d = (fx x theta / r + cx, fy y theta / r + cy),
//            J = d(mean2d)/d(x, y, z) in the closed form below (eps = 1e-7 exactly where the published code has it).
// No 1.3 tan(fov) clamp in these models.  EWA with a general 2x3 Jacobian: T = J W, cov2d = T Sigma T^T.
enum { GSPL_CAM_PINHOLE_ = 0, GSPL_CAM_ORTHO_ = 1, GSPL_CAM_FISHEYE_ = 2 };
static constexpr float FISHEYE_EPS = 1e-7f;

struct FisheyeTerms { float x, y, z, r, x2, y2, xy, s, q, theta, a, b; };

__device__ __forceinline__ void fisheye_terms(const float pc[3], FisheyeTerms& t) {
    t.x = pc[0]; t.y = pc[1]; t.z = pc[2];
    t.r = sqrtf(t.x * t.x + t.y * t.y) + FISHEYE_EPS;
    t.x2 = t.x * t.x + FISHEYE_EPS;
    t.y2 = t.y * t.y;
    t.xy = t.x * t.y;
    t.s = t.x2 + t.y2;
    t.q = 1.f / (t.s + t.z * t.z);
    t.theta = atan2f(t.r, t.z);
    t.b = t.theta / t.r / t.s;
    t.a = t.z * t.q / t.s;
}

// J (row-major 2x3) and the projected mean WITHOUT the principal point.
template <int CAM>
__device__ __forceinline__ void cam_project(const float pc[3], float fx, float fy, float J[6], float& mx, float& my) {
    if (CAM == GSPL_CAM_ORTHO_) {
        J[0] = fx; J[1] = 0.f; J[2] = 0.f; J[3] = 0.f; J[4] = fy; J[5] = 0.f;
        mx = fx * pc[0]; my = fy * pc[1];
    } else {
        FisheyeTerms t;
        fisheye_terms(pc, t);
        const float thm = atan2f(t.r, t.z + FISHEYE_EPS);
        mx = t.x * fx * thm / t.r;
        my = t.y * fy * thm / t.r;
        J[0] = fx * (t.x2 * t.a + t.y2 * t.b); J[1] = fx * t.xy * (t.a - t.b); J[2] = -fx * t.x * t.q;
        J[3] = fy * t.xy * (t.a - t.b); J[4] = fy * (t.y2 * t.a + t.x2 * t.b); J[5] = -fy * t.y * t.q;
    }
}

// v_pc += (d mean / d pc)^T (vmx, vmy) + (d J / d pc)^T vJ
template <int CAM>
__device__ __forceinline__ void cam_project_bwd(const float pc[3], float fx, float fy, const float vJ[6], float vmx, float vmy, float v_pc[3]) {
    if (CAM == GSPL_CAM_ORTHO_) {
        v_pc[0] += fx * vmx;
        v_pc[1] += fy * vmy;
        return;
    }
    FisheyeTerms t;
    fisheye_terms(pc, t);
    const float x = t.x, y = t.y, z = t.z, r = t.r;
    const float r0 = fmaxf(r - FISHEYE_EPS, 1e-30f);          // |(x, y)|: d r / d x = x / r0
    const float drdx = x / r0, drdy = y / r0;
    // mean: m = f * c * g(r, z) with g = atan2(r, z + eps) / r
    {
        const float ze = z + FISHEYE_EPS;
        const float den = 1.f / (r * r + ze * ze);
        const float thm = atan2f(r, ze);
        const float g = thm / r;
        const float dg_dr = (ze * den) / r - thm / (r * r);
        const float dg_dz = (-r * den) / r;
        const float wx = fx * vmx, wy = fy * vmy;
        const float common = (wx * x + wy * y) * dg_dr;
        v_pc[0] += wx * g + common * drdx;
        v_pc[1] += wy * g + common * drdy;
        v_pc[2] += (wx * x + wy * y) * dg_dz;
    }
    // Jacobian entries
    const float s = t.s, q = t.q, a = t.a, b = t.b, th = t.theta;
    const float den = 1.f / (r * r + z * z);
    const float dth_dr = z * den, dth_dz = -r * den;
    const float dq[3] = {-q * q * 2.f * x, -q * q * 2.f * y, -q * q * 2.f * z};
    const float ds[3] = {2.f * x, 2.f * y, 0.f};
    const float dr[3] = {drdx, drdy, 0.f};
    float da[3], db[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        da[k] = z * dq[k] / s - z * q / (s * s) * ds[k];
        db[k] = dth_dr * dr[k] / (r * s) - th / (r * r * s) * dr[k] - th / (r * s * s) * ds[k];
    }
    da[2] += q / s;
    db[2] += dth_dz / (r * s);
    const float amb = a - b;
    const float dx2[3] = {2.f * x, 0.f, 0.f}, dy2[3] = {0.f, 2.f * y, 0.f}, dxy[3] = {y, x, 0.f};
    const float dxk[3] = {1.f, 0.f, 0.f}, dyk[3] = {0.f, 1.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float dJ00 = fx * (dx2[k] * a + t.x2 * da[k] + dy2[k] * b + t.y2 * db[k]);
        const float dJ01 = fx * (dxy[k] * amb + t.xy * (da[k] - db[k]));
        const float dJ02 = -fx * (dxk[k] * q + x * dq[k]);
        const float dJ10 = fy * (dxy[k] * amb + t.xy * (da[k] - db[k]));
        const float dJ11 = fy * (dy2[k] * a + t.y2 * da[k] + dx2[k] * b + t.x2 * db[k]);
        const float dJ12 = -fy * (dyk[k] * q + y * dq[k]);
        v_pc[k] += vJ[0] * dJ00 + vJ[1] * dJ01 + vJ[2] * dJ02 + vJ[3] * dJ10 + vJ[4] * dJ11 + vJ[5] * dJ12;
    }
}

// cov2d (a0, b0, c0) = (J W) Sigma (J W)^T for a general J; T0 / T1 = the rows of J W.
__device__ __forceinline__ void ewa_fwd_general(const float J[6], const float S6[6], const float W[9],
                                                float& a0, float& b0, float& c0, float T0[3], float T1[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        T0[j] = J[0] * W[0 * 3 + j] + J[1] * W[1 * 3 + j] + J[2] * W[2 * 3 + j];
        T1[j] = J[3] * W[0 * 3 + j] + J[4] * W[1 * 3 + j] + J[5] * W[2 * 3 + j];
    }
    const float s0x = S6[0] * T0[0] + S6[1] * T0[1] + S6[2] * T0[2];
    const float s0y = S6[1] * T0[0] + S6[3] * T0[1] + S6[4] * T0[2];
    const float s0z = S6[2] * T0[0] + S6[4] * T0[1] + S6[5] * T0[2];
    const float s1x = S6[0] * T1[0] + S6[1] * T1[1] + S6[2] * T1[2];
    const float s1y = S6[1] * T1[0] + S6[3] * T1[1] + S6[4] * T1[2];
    const float s1z = S6[2] * T1[0] + S6[4] * T1[1] + S6[5] * T1[2];
    a0 = T0[0] * s0x + T0[1] * s0y + T0[2] * s0z;
    b0 = T0[0] * s1x + T0[1] * s1y + T0[2] * s1z;
    c0 = T1[0] * s1x + T1[1] * s1y + T1[2] * s1z;
}

// (va, vb, vc) as in ewa_bwd -> vJ (row-major 2x3) and G6.
__device__ __forceinline__ void ewa_bwd_general(const float S6[6], const float W[9], const float T0[3], const float T1[3],
                                                float va, float vb, float vc, float vJ[6], float G6[6]) {
    const float hb = 0.5f * vb;
    G6[0] = va * T0[0] * T0[0] + vb * T0[0] * T1[0] + vc * T1[0] * T1[0];
    G6[3] = va * T0[1] * T0[1] + vb * T0[1] * T1[1] + vc * T1[1] * T1[1];
    G6[5] = va * T0[2] * T0[2] + vb * T0[2] * T1[2] + vc * T1[2] * T1[2];
    G6[1] = va * T0[0] * T0[1] + hb * (T0[0] * T1[1] + T1[0] * T0[1]) + vc * T1[0] * T1[1];
    G6[2] = va * T0[0] * T0[2] + hb * (T0[0] * T1[2] + T1[0] * T0[2]) + vc * T1[0] * T1[2];
    G6[4] = va * T0[1] * T0[2] + hb * (T0[1] * T1[2] + T1[1] * T0[2]) + vc * T1[1] * T1[2];
    const float s0[3] = {S6[0] * T0[0] + S6[1] * T0[1] + S6[2] * T0[2],
                         S6[1] * T0[0] + S6[3] * T0[1] + S6[4] * T0[2],
                         S6[2] * T0[0] + S6[4] * T0[1] + S6[5] * T0[2]};
    const float s1[3] = {S6[0] * T1[0] + S6[1] * T1[1] + S6[2] * T1[2],
                         S6[1] * T1[0] + S6[3] * T1[1] + S6[4] * T1[2],
                         S6[2] * T1[0] + S6[4] * T1[1] + S6[5] * T1[2]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            v0 += (2.f * va * s0[j] + vb * s1[j]) * W[k * 3 + j];
            v1 += (2.f * vc * s1[j] + vb * s0[j]) * W[k * 3 + j];
        }
        vJ[k] = v0;
        vJ[3 + k] = v1;
    }
}

// conic (A,B,C) = inverse of [[a,b],[b,c]]; given dL/d(A,B,C) (vB w.r.t. the single parameter B)
// return dL/d(a,b,c).
__device__ __forceinline__ void conic_bwd(float a, float b, float c, float vA, float vB, float vC,
                                          float& va, float& vb, float& vc) {
    const float det = a * c - b * b;
    const float rd2 = 1.f / (det * det);
    va = rd2 * (-c * c * vA + b * c * vB - b * b * vC);
    vc = rd2 * (-b * b * vA + a * b * vB - a * a * vC);
    vb = rd2 * (2.f * b * c * vA - (a * c + b * b) * vB + 2.f * a * b * vC);
}

// ---- wave64 helpers -----------------------------------------------------------------------------
// v_writelane_b32: lane `lane` (wave-uniform) of `old` is replaced by the wave-uniform `value` (this clang has no builtin
// for it; the declaration binds the LLVM intrinsic by name).
__device__ int gspl_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}
// Sum over the 16 lanes of each DPP row; every lane ends up holding its row's sum.
__device__ __forceinline__ float row_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);   // row_mirror
    return v;
}
// In-place v += <dpp>(v) for N registers, written as v_add_f32_dpp by hand.  When the adds of a reduction's LAST step
// are only consumed under a lane condition the compiler sinks them into the conditional block and can no longer fuse
// them with their DPP moves (3 instructions per value instead of 1).  One s_nop 1 covers the VALU-write -> DPP-read
// hazard (2 wait states) for each group of up to four independent adds.
#define GSPL_DPP_ADD_INPLACE(NAME, CTRL)                                                                         \
    template <int N>                                                                                             \
    __device__ __forceinline__ void NAME(float (&v)[N]) {                                                        \
        int k = 0;                                                                                               \
        _Pragma("unroll") for (; k + 4 <= N; k += 4)                                                             \
            asm volatile("s_nop 1\n\t"                                                                           \
                         "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
                         "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
                         "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"         \
                         "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1"               \
                         : "+v"(v[k]), "+v"(v[k + 1]), "+v"(v[k + 2]), "+v"(v[k + 3]));                          \
        _Pragma("unroll") for (; k < N; ++k)                                                                     \
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1"  \
                         : "+v"(v[k]));                                                                          \
    }
GSPL_DPP_ADD_INPLACE(row_mirror_add, "row_mirror")
GSPL_DPP_ADD_INPLACE(row_ror8_add, "row_ror:8")
GSPL_DPP_ADD_INPLACE(quad_xor1_add, "quad_perm:[1,0,3,2]")
GSPL_DPP_ADD_INPLACE(quad_xor2_add, "quad_perm:[2,3,0,1]")
GSPL_DPP_ADD_INPLACE(half_mirror_add, "row_half_mirror")

// Sum over the 64 lanes of the wave; the total is valid in lane 63 (row 3).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);   // row_mirror          -> every lane holds its 16-row sum
    v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2,3
    return v;
}

// ZeroJob of gspl_host.h, run by every thread of the calling kernel (grid-stride; the tables are a few hundred KB)
__device__ __forceinline__ void zero_table(uint4* __restrict__ p, uint32_t n16) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// a c - b^2 without the cancellation of its two products (Kahan): w = fl(b b), e = w - b b exactly, f = fl(a c - w); det = f + e
__device__ __forceinline__ float conic_det(float a, float b, float c) {
    const float w = b * b;
    const float e = fmaf(-b, b, w);
    const float f = fmaf(a, c, -w);
    return f + e;
}

// One element of the Adam update (adam.hip; the per-Gaussian kernels of the backward in their Adam-applying form, sh.hip / inria.hip).
// Written with explicit fmaf so that every translation unit contracts it the same way: the two-kernel path (gradient written, read
// back by selective_adam_kernel) and the update inside the backward are bit-identical for identical gradients.
//     m = b1 m + (1 - b1) g;   v = b2 v + (1 - b2) g^2;   p -= step m / (sqrt(v) inv_bc2 + eps)
struct AdamHyper { float step, b1, b2, inv_bc2, eps; };      // step = lr * (1 / bias_correction1), inv_bc2 = 1 / bias_correction2_sqrt: as selective_adam_kernel forms them
struct AdamTarget { float* p; float* m; float* v; AdamHyper h; };
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float step, float b1, float b2, float inv_bc2, float eps) {
    m = fmaf(b1, m, (1.f - b1) * g);
    v = fmaf(b2, v, ((1.f - b2) * g) * g);
    p -= (step * m) / fmaf(sqrtf(v), inv_bc2, eps);
}
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamHyper& h) {
    adam_elem(p, g, m, v, h.step, h.b1, h.b2, h.inv_bc2, h.eps);
}

// XCD-aware tile order.  The dispatcher places workgroup b on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch"),
// and each XCD runs its workgroups in increasing b.  Units are dealt to the XCDs in RUNS of `run` consecutive units
// (run r -> XCD r % 8): neighbouring tiles of a run share splat records (L2 hits inside the XCD), while all eight XCDs
// sweep the image top to bottom together, so an image whose splats are concentrated in some rows (the usual case)
// loads every XCD alike.  (Giving each XCD one contiguous eighth of the image, the first version of this mapping,
// left the XCDs that own the dense middle rows running long after the others had finished.)  Speed only.
#ifndef GSPL_XCD_RUN
#define GSPL_XCD_RUN 32
#endif
__device__ __forceinline__ int xcd_remap(int b, int n, int run = GSPL_XCD_RUN) {
    const int group = 8 * run;
    const int body = (n / group) * group;
    if (b >= body) return b;         // tail handled in place
    const int x = b & 7, i = b >> 3; // XCD, position in that XCD's queue
    return ((i / run) * 8 + x) * run + (i % run);
}

}  // namespace gspl
