// sh.hip — spherical-harmonics colour, forward and backward (gfx950).
//
// Replaces gsplat `spherical_harmonics` / `spherical_harmonics_decomposed` at the reference call
// sites internal/renderers/gsplat_renderer.py:105, gsplat_v1_renderer.py:121-130,
// gsplat_distributed_renderer.py:416-421.  Basis, constants, signs and coefficient order are a
// restatement of internal/utils/sh_utils.py:26-171 (degrees 0..4).
//
// Roofline: HBM-bound.  Algorithmic bytes per visible Gaussian: (12K + 24) forward,
// 2*12K backward (K = (deg+1)^2 = 16 -> 216 B / 384 B).
// The coefficient tensors are AoS ([N,K,3]: 180-192 B per Gaussian), which a lane-per-Gaussian
// kernel would read as 48 strided dwords (64 cache lines per wave instruction).  Instead a
// workgroup streams its 256 rows as one flat, fully coalesced 16-B-per-lane copy into LDS
// (odd row stride -> conflict-free), and lanes then read their own row from LDS.  Backward writes
// the coefficient gradients the same way in reverse.
// Floating-point contraction as the LANGUAGE defines it (a * b + c inside one expression), not as the back end finds it: the
// gradient-writing and the Adam-applying instantiations of the backward kernels below must produce bit-identical gradient values
// (tests/test_fused_backward_adam.py), and -ffp-contract=fast (hipcc's default) lets the DAG combiner fuse across statements
// differently in each instantiation (measured: the means' gradient differed in its last bit from the third step on).
#pragma clang fp contract(on)
#include "gspl_device.h"
#include "gspl_host.h"

namespace gspl {

#ifndef GSPL_SH_U
#define GSPL_SH_U 6
#endif
#ifndef GSPL_SH_BLOCK
#define GSPL_SH_BLOCK 256
#endif
#ifndef GSPL_SH_ADAM_U
#define GSPL_SH_ADAM_U 2
#endif
static constexpr int SH_BLOCK = GSPL_SH_BLOCK;      // rows (= threads) per workgroup
static constexpr int SH_MAX_K = 25;

__device__ __constant__ const float kC0 = 0.28209479177387814f;
__device__ __constant__ const float kC1 = 0.4886025119029199f;

// basis values b[0..K) for unit direction (x,y,z)
__device__ __forceinline__ void sh_basis(int degree, float x, float y, float z, float* b) {
    b[0] = 0.28209479177387814f;
    if (degree < 1) return;
    b[1] = -0.4886025119029199f * y;
    b[2] = 0.4886025119029199f * z;
    b[3] = -0.4886025119029199f * x;
    if (degree < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = 1.0925484305920792f * xy;
    b[5] = -1.0925484305920792f * yz;
    b[6] = 0.31539156525252005f * (2.f * zz - xx - yy);
    b[7] = -1.0925484305920792f * xz;
    b[8] = 0.5462742152960396f * (xx - yy);
    if (degree < 3) return;
    b[9] = -0.5900435899266435f * y * (3.f * xx - yy);
    b[10] = 2.890611442640554f * xy * z;
    b[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
    b[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
    b[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
    b[14] = 1.445305721320277f * z * (xx - yy);
    b[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
    if (degree < 4) return;
    b[16] = 2.5033429417967046f * xy * (xx - yy);
    b[17] = -1.7701307697799304f * yz * (3.f * xx - yy);
    b[18] = 0.9461746957575601f * xy * (7.f * zz - 1.f);
    b[19] = -0.6690465435572892f * yz * (7.f * zz - 3.f);
    b[20] = 0.10578554691520431f * (zz * (35.f * zz - 30.f) + 3.f);
    b[21] = -0.6690465435572892f * xz * (7.f * zz - 3.f);
    b[22] = 0.47308734787878004f * (xx - yy) * (7.f * zz - 1.f);
    b[23] = -1.7701307697799304f * xz * (xx - 3.f * yy);
    b[24] = 0.6258357354491761f * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy));
}

// gradient of sum_k w[k]*b[k] w.r.t. (x,y,z) treated as independent variables
__device__ __forceinline__ void sh_basis_grad(int degree, float x, float y, float z, const float* w,
                                              float& gx, float& gy, float& gz) {
    gx = gy = gz = 0.f;
    if (degree < 1) return;
    gy += -0.4886025119029199f * w[1];
    gz += 0.4886025119029199f * w[2];
    gx += -0.4886025119029199f * w[3];
    if (degree < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    { const float c = 1.0925484305920792f * w[4]; gx += c * y; gy += c * x; }
    { const float c = -1.0925484305920792f * w[5]; gy += c * z; gz += c * y; }
    { const float c = 0.31539156525252005f * w[6]; gx += -2.f * c * x; gy += -2.f * c * y; gz += 4.f * c * z; }
    { const float c = -1.0925484305920792f * w[7]; gx += c * z; gz += c * x; }
    { const float c = 0.5462742152960396f * w[8]; gx += 2.f * c * x; gy += -2.f * c * y; }
    if (degree < 3) return;
    { const float c = -0.5900435899266435f * w[9]; gx += c * 6.f * xy; gy += c * (3.f * xx - 3.f * yy); }
    { const float c = 2.890611442640554f * w[10]; gx += c * yz; gy += c * xz; gz += c * xy; }
    { const float c = -0.4570457994644658f * w[11]; gx += c * (-2.f * xy); gy += c * (4.f * zz - xx - 3.f * yy); gz += c * 8.f * yz; }
    { const float c = 0.3731763325901154f * w[12]; gx += c * (-6.f * xz); gy += c * (-6.f * yz); gz += c * (6.f * zz - 3.f * xx - 3.f * yy); }
    { const float c = -0.4570457994644658f * w[13]; gx += c * (4.f * zz - 3.f * xx - yy); gy += c * (-2.f * xy); gz += c * 8.f * xz; }
    { const float c = 1.445305721320277f * w[14]; gx += c * 2.f * xz; gy += c * (-2.f * yz); gz += c * (xx - yy); }
    { const float c = -0.5900435899266435f * w[15]; gx += c * (3.f * xx - 3.f * yy); gy += c * (-6.f * xy); }
    if (degree < 4) return;
    { const float c = 2.5033429417967046f * w[16]; gx += c * (3.f * xx * y - yy * y); gy += c * (xx * x - 3.f * x * yy); }
    { const float c = -1.7701307697799304f * w[17]; gx += c * 6.f * xy * z; gy += c * (3.f * xx * z - 3.f * yy * z); gz += c * (3.f * xx * y - yy * y); }
    { const float c = 0.9461746957575601f * w[18]; gx += c * (7.f * y * zz - y); gy += c * (7.f * x * zz - x); gz += c * 14.f * xy * z; }
    { const float c = -0.6690465435572892f * w[19]; gy += c * (7.f * zz * z - 3.f * z); gz += c * (21.f * y * zz - 3.f * y); }
    { const float c = 0.10578554691520431f * w[20]; gz += c * (140.f * zz * z - 60.f * z); }
    { const float c = -0.6690465435572892f * w[21]; gx += c * (7.f * zz * z - 3.f * z); gz += c * (21.f * x * zz - 3.f * x); }
    { const float c = 0.47308734787878004f * w[22]; gx += c * 2.f * x * (7.f * zz - 1.f); gy += c * (-2.f) * y * (7.f * zz - 1.f); gz += c * 14.f * z * (xx - yy); }
    { const float c = -1.7701307697799304f * w[23]; gx += c * (3.f * xx * z - 3.f * yy * z); gy += c * (-6.f * xy * z); gz += c * (xx * x - 3.f * x * yy); }
    { const float c = 0.6258357354491761f * w[24]; gx += c * (4.f * xx * x - 12.f * x * yy); gy += c * (-12.f * xx * y + 4.f * yy * y); }
}

// The coefficient rows and their gradients are streamed once per launch: nontemporal accesses (-DGSPL_SH_PLAIN: plain ones).
typedef float sh_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 sh_load16(const float4* p) {
#ifdef GSPL_SH_PLAIN
    return *p;
#else
    const sh_v4f t = __builtin_nontemporal_load(reinterpret_cast<const sh_v4f*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
#endif
}
__device__ __forceinline__ void sh_store16(float4* p, const float4& q) {
#ifdef GSPL_SH_PLAIN
    *p = q;
#else
    const sh_v4f t = {q.x, q.y, q.z, q.w};
    __builtin_nontemporal_store(t, reinterpret_cast<sh_v4f*>(p));
#endif
}

// Flat coalesced copy of `rows` rows x `rs` floats (contiguous in global) into LDS rows of stride ls.
__device__ __forceinline__ void tile_load(const float* __restrict__ g, int rows, int rs, int ls, float inv_rs,
                                          float* lds, bool vec_ok) {
    const int total = rows * rs;
    const int t = threadIdx.x;
    int done = 0;
    if (vec_ok) {
        const int n4 = total >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(g);
        // batches of U independent 16-B loads per lane before any LDS write: keeps U KiB per wave in flight
        // (a load -> scatter -> load chain leaves the memory pipe idle most of the time: SQ_WAIT_ANY was 77 %)
        constexpr int U = GSPL_SH_U;
        for (int b4 = t; b4 < n4; b4 += SH_BLOCK * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e4 = b4 + u * SH_BLOCK;
                v[u] = (e4 < n4) ? sh_load16(g4 + e4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e4 = b4 + u * SH_BLOCK;
                if (e4 < n4) {
                    const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int e = e4 * 4 + k;
                        const int row = (int)(((float)e + 0.5f) * inv_rs);
                        lds[row * ls + (e - row * rs)] = vv[k];
                    }
                }
            }
        }
        done = n4 << 2;
    }
    for (int e = done + t; e < total; e += SH_BLOCK) {
        const int row = (int)(((float)e + 0.5f) * inv_rs);
        lds[row * ls + (e - row * rs)] = g[e];
    }
}
__device__ __forceinline__ void tile_store(float* __restrict__ g, int rows, int rs, int ls, float inv_rs,
                                           const float* lds, bool vec_ok) {
    const int total = rows * rs;
    const int t = threadIdx.x;
    int done = 0;
    if (vec_ok) {
        const int n4 = total >> 2;
        float4* g4 = reinterpret_cast<float4*>(g);
        for (int e4 = t; e4 < n4; e4 += SH_BLOCK) {
            float vv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e4 * 4 + k;
                const int row = (int)(((float)e + 0.5f) * inv_rs);
                vv[k] = lds[row * ls + (e - row * rs)];
            }
            sh_store16(g4 + e4, make_float4(vv[0], vv[1], vv[2], vv[3]));
        }
        done = n4 << 2;
    }
    for (int e = done + t; e < total; e += SH_BLOCK) {
        const int row = (int)(((float)e + 0.5f) * inv_rs);
        g[e] = lds[row * ls + (e - row * rs)];
    }
}

// The Adam-applying counterpart of tile_store (backward kernels of the "update inside the backward" form): the rows of
// LDS hold the GRADIENT of `rows` rows x `rs` floats; parameter and moments are streamed through once, flat and coalesced, 16 bytes
// per lane (GSPL_SH_ADAM_U chunks = 3 U loads in flight per lane), updated and written back — the gradient never reaches HBM.
__device__ __forceinline__ void tile_adam(const AdamTarget& T, int64_t base, int rows, int rs, int ls, float inv_rs, const float* lds, bool vec_ok) {
    const int total = rows * rs;
    const int t = threadIdx.x;
    float* __restrict__ gp = T.p + base;
    float* __restrict__ gm = T.m + base;
    float* __restrict__ gv = T.v + base;
    int done = 0;
    if (vec_ok) {
        const int n4 = total >> 2;
        float4* p4 = reinterpret_cast<float4*>(gp);
        float4* m4 = reinterpret_cast<float4*>(gm);
        float4* v4 = reinterpret_cast<float4*>(gv);
        constexpr int U = GSPL_SH_ADAM_U;      // 16-byte chunks per lane and iteration: 3 U loads in flight
        for (int b4 = t; b4 < n4; b4 += U * SH_BLOCK) {
            float4 p[U], m[U], v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e4 = b4 + u * SH_BLOCK;
                if (e4 < n4) { p[u] = sh_load16(p4 + e4); m[u] = sh_load16(m4 + e4); v[u] = sh_load16(v4 + e4); }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e4 = b4 + u * SH_BLOCK;
                if (e4 >= n4) continue;
                float g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = e4 * 4 + k;
                    const int row = (int)(((float)e + 0.5f) * inv_rs);
                    g[k] = lds[row * ls + (e - row * rs)];
                }
                adam_elem(p[u].x, g[0], m[u].x, v[u].x, T.h);
                adam_elem(p[u].y, g[1], m[u].y, v[u].y, T.h);
                adam_elem(p[u].z, g[2], m[u].z, v[u].z, T.h);
                adam_elem(p[u].w, g[3], m[u].w, v[u].w, T.h);
                p4[e4] = p[u];                       // read again by the next frame's colour kernel: a plain store
                sh_store16(m4 + e4, m[u]);
                sh_store16(v4 + e4, v[u]);
            }
        }
        done = n4 << 2;
    }
    for (int e = done + t; e < total; e += SH_BLOCK) {
        const int row = (int)(((float)e + 0.5f) * inv_rs);
        float p = gp[e], m = gm[e], v = gv[e];
        adam_elem(p, lds[row * ls + (e - row * rs)], m, v, T.h);
        gp[e] = p; gm[e] = m; gv[e] = v;
    }
}

// Adam targets of sh_bwd_kernel<.., ADAM = true>: `tile` = the parameter the staged rows belong to (shs_rest, or the merged shs),
// `dc` = the separate DC parameter ([N, 1, 3]; unused when merged)
struct ShAdam { AdamTarget tile, dc; };

// tile geometry:
//   merged  : tile rows are the full [K,3] rows starting at dc (row stride rs = dc_stride), dc at col 0
//   separate: tile rows are the `rest` rows (row stride rs = rest_stride); dc is read per lane
struct ShTile {
    int rs;       // global row stride (floats)
    int ls;       // LDS row stride (odd)
    int merged;   // 1: dc inside the tile at col 0, rest at col 3
};

// C cameras per launch (C > 1: the Gaussian-sharded renderer evaluates its shard for every rank's camera,
// gsplat_distributed_renderer.py:252-311): the coefficient rows are staged ONCE and evaluated against C origins; origin,
// mask, mask32, colors and clamped are then arrays of C consecutive per-camera blocks.
template <int DEG>
__global__ __launch_bounds__(SH_BLOCK) void sh_fwd_kernel(
    int N, int C,
    const float* __restrict__ dirs, const float* __restrict__ origin,
    const float* __restrict__ dc, int dc_stride, const float* __restrict__ rest,
    const uint8_t* __restrict__ mask, const int32_t* __restrict__ mask32, int flags, ShTile tile, int vec_ok,
    float* __restrict__ colors, uint8_t* __restrict__ clamped, float* __restrict__ jac) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n0 = blockIdx.x * SH_BLOCK;
    const int rows = min(SH_BLOCK, N - n0);
    constexpr int degree = DEG;
    constexpr int K = (DEG + 1) * (DEG + 1);
    if (degree > 0) {
        const float* base = tile.merged ? dc + (int64_t)n0 * tile.rs : rest + (int64_t)n0 * tile.rs;
        tile_load(base, rows, tile.rs, tile.ls, 1.f / (float)tile.rs, lds, vec_ok != 0);
        __syncthreads();
    }
    const int r = threadIdx.x;
    if (r >= rows) return;
    const int n = n0 + r;
    const float px = dirs[n * 3 + 0], py = dirs[n * 3 + 1], pz = dirs[n * 3 + 2];
    for (int cam = 0; cam < C; ++cam) {
        const int64_t cn = (int64_t)cam * N + n;
        float out[3] = {0.f, 0.f, 0.f};
        uint8_t cl[3] = {0, 0, 0};
        const bool live = ((mask == nullptr) || (mask[cn] != 0)) && ((mask32 == nullptr) || (mask32[cn] > 0));
        if (live) {
            float dx = px, dy = py, dz = pz;
            if (origin) { dx -= origin[cam * 3 + 0]; dy -= origin[cam * 3 + 1]; dz -= origin[cam * 3 + 2]; }
            const float inv = rsqrtf(dx * dx + dy * dy + dz * dz);
            dx *= inv; dy *= inv; dz *= inv;
            float b[K];
            sh_basis(degree, dx, dy, dz, b);
            const float* d0 = (tile.merged && degree > 0) ? (lds + r * tile.ls) : (dc + (int64_t)n * dc_stride);
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] = b[0] * d0[c];
            const float* row = lds + r * tile.ls + (tile.merged ? 3 : 0);
#pragma unroll
            for (int k = 1; k < K; ++k) {
#pragma unroll
                for (int c = 0; c < 3; ++c) out[c] += b[k] * row[(k - 1) * 3 + c];
            }
            if (flags & GSPL_SH_ADD_HALF_CLAMP) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    out[c] += 0.5f;
                    if (out[c] < 0.f) { out[c] = 0.f; cl[c] = 1; }
                }
            }
            // d colour_c / d (unit direction), 9 floats per splat: with it the backward gets the direction gradient without
            // reading the 12 K bytes of coefficients again (single camera only)
            if (jac && degree > 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float w[K];
                    w[0] = 0.f;
#pragma unroll
                    for (int k = 1; k < K; ++k) w[k] = row[(k - 1) * 3 + c];
                    float gx, gy, gz;
                    sh_basis_grad(degree, dx, dy, dz, w, gx, gy, gz);
                    jac[(int64_t)n * 9 + c * 3 + 0] = gx; jac[(int64_t)n * 9 + c * 3 + 1] = gy; jac[(int64_t)n * 9 + c * 3 + 2] = gz;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) colors[cn * 3 + c] = out[c];
        if (clamped) {
#pragma unroll
            for (int c = 0; c < 3; ++c) clamped[cn * 3 + c] = cl[c];
        }
    }
}

// WITH_DIRS: also produce v_dirs (needs the coefficients -> stages them first).
// ADAM: the coefficient gradients are not written: the kernel applies the Adam update to the coefficient rows it has just produced the
// gradient of (tile_adam; v_dc / v_rest are then only consulted for the layout, `adam` carries parameter, moments and hyper-parameters).
template <int DEG, bool WITH_DIRS, bool ADAM = false>
__global__ __launch_bounds__(SH_BLOCK) void sh_bwd_kernel(
    int N, int C, int n_coeffs,
    const float* __restrict__ dirs, const float* __restrict__ origin,
    const float* __restrict__ dc, int dc_stride, const float* __restrict__ rest,
    const uint8_t* __restrict__ mask, const int32_t* __restrict__ mask32, int flags, const uint8_t* __restrict__ clamped,
    const float* __restrict__ v_colors, int vc_stride, ShTile tile, int vec_ok_in, int vec_ok_out,
    float* __restrict__ v_dc, float* __restrict__ v_rest, float* __restrict__ v_dirs, const float* __restrict__ jac, ShAdam adam) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n0 = blockIdx.x * SH_BLOCK;
    const int rows = min(SH_BLOCK, N - n0);
    constexpr int degree = DEG;
    constexpr int K = (DEG + 1) * (DEG + 1);
    const int r = threadIdx.x;
    const int n = n0 + r;
    const bool has_tile = n_coeffs > 1;   // there is a `rest` block to write

    float vc[3] = {0.f, 0.f, 0.f};
    float dx = 0.f, dy = 0.f, dz = 1.f, inv = 1.f;
    float b[K];
#pragma unroll
    for (int k = 0; k < K; ++k) b[k] = 0.f;
    bool live = false;
    // camera `cam`: masked / clamp-masked colour gradient and the basis of the view direction
    auto load_camera = [&](int cam) {
        const int64_t cn = (int64_t)cam * N + n;
        live = ((mask == nullptr) || (mask[cn] != 0)) && ((mask32 == nullptr) || (mask32[cn] > 0));
#pragma unroll
        for (int c = 0; c < 3; ++c) vc[c] = 0.f;
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                vc[c] = v_colors[cn * vc_stride + c];
                if ((flags & GSPL_SH_ADD_HALF_CLAMP) && clamped && clamped[cn * 3 + c]) vc[c] = 0.f;
            }
            dx = dirs[n * 3 + 0]; dy = dirs[n * 3 + 1]; dz = dirs[n * 3 + 2];
            if (origin) { dx -= origin[cam * 3 + 0]; dy -= origin[cam * 3 + 1]; dz -= origin[cam * 3 + 2]; }
            inv = rsqrtf(dx * dx + dy * dy + dz * dz);
            dx *= inv; dy *= inv; dz *= inv;
            sh_basis(degree, dx, dy, dz, b);
        }
    };
    if (r < rows) load_camera(0);

    if (WITH_DIRS && jac) {      // the forward left d colour / d direction: no coefficient read at all
        if (r < rows) {
            float g[3] = {0.f, 0.f, 0.f};
            if (live && degree > 0) {
                float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    gx += vc[c] * jac[(int64_t)n * 9 + c * 3 + 0];
                    gy += vc[c] * jac[(int64_t)n * 9 + c * 3 + 1];
                    gz += vc[c] * jac[(int64_t)n * 9 + c * 3 + 2];
                }
                const float dot = gx * dx + gy * dy + gz * dz;
                g[0] = (gx - dx * dot) * inv; g[1] = (gy - dy * dot) * inv; g[2] = (gz - dz * dot) * inv;
            }
            v_dirs[n * 3 + 0] = g[0]; v_dirs[n * 3 + 1] = g[1]; v_dirs[n * 3 + 2] = g[2];
        }
    } else if (WITH_DIRS) {      // one camera only (checked by the launcher)
        if (degree > 0) {
            const float* base = tile.merged ? dc + (int64_t)n0 * tile.rs : rest + (int64_t)n0 * tile.rs;
            tile_load(base, rows, tile.rs, tile.ls, 1.f / (float)tile.rs, lds, vec_ok_in != 0);
            __syncthreads();
        }
        if (r < rows) {
            float g[3] = {0.f, 0.f, 0.f};
            if (live && degree > 0) {
                float w[K];
                const float* row = lds + r * tile.ls + (tile.merged ? 3 : 0);
                w[0] = 0.f;
#pragma unroll
                for (int k = 1; k < K; ++k)
                    w[k] = row[(k - 1) * 3 + 0] * vc[0] + row[(k - 1) * 3 + 1] * vc[1] + row[(k - 1) * 3 + 2] * vc[2];
                float gx, gy, gz;
                sh_basis_grad(degree, dx, dy, dz, w, gx, gy, gz);
                // through d = v / |v| :  v_v = (g - d (d.g)) / |v|
                const float dot = gx * dx + gy * dy + gz * dz;
                g[0] = (gx - dx * dot) * inv; g[1] = (gy - dy * dot) * inv; g[2] = (gz - dz * dot) * inv;
            }
            v_dirs[n * 3 + 0] = g[0]; v_dirs[n * 3 + 1] = g[1]; v_dirs[n * 3 + 2] = g[2];
        }
        __syncthreads();   // everyone is done reading the staged coefficients
    }

    // coefficient gradients: write own row into LDS (summed over the cameras), then one flat coalesced store
    if (r < rows) {
        float d0[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) d0[c] = live ? b[0] * vc[c] : 0.f;
        float* row = lds + r * tile.ls;
        const int col0 = (has_tile && tile.merged) ? 3 : 0;
        if (has_tile) {
#pragma unroll
            for (int k = 1; k < K; ++k) {
                const float bk = live ? b[k] : 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) row[col0 + (k - 1) * 3 + c] = bk * vc[c];
            }
            for (int j = col0 + (K - 1) * 3; j < col0 + (n_coeffs - 1) * 3; ++j) row[j] = 0.f;   // above the active degree
            // zero any padding columns between the used row length and the global row stride
            for (int j = col0 + (n_coeffs - 1) * 3; j < tile.rs; ++j) row[j] = 0.f;
        }
        for (int cam = 1; cam < C; ++cam) {
            load_camera(cam);
            if (!live) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) d0[c] += b[0] * vc[c];
            if (has_tile) {
#pragma unroll
                for (int k = 1; k < K; ++k) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) row[col0 + (k - 1) * 3 + c] += b[k] * vc[c];
                }
            }
        }
        if (has_tile && tile.merged) { row[0] = d0[0]; row[1] = d0[1]; row[2] = d0[2]; }
        if (!tile.merged || !has_tile) {
            if constexpr (ADAM) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int64_t e = (int64_t)n * dc_stride + c;
                    float p = adam.dc.p[e], m = adam.dc.m[e], v = adam.dc.v[e];
                    adam_elem(p, d0[c], m, v, adam.dc.h);
                    adam.dc.p[e] = p; adam.dc.m[e] = m; adam.dc.v[e] = v;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) v_dc[(int64_t)n * dc_stride + c] = d0[c];
            }
        }
    }
    if (has_tile) {
        __syncthreads();
        if constexpr (ADAM) {
            tile_adam(adam.tile, (int64_t)n0 * tile.rs, rows, tile.rs, tile.ls, 1.f / (float)tile.rs, lds, vec_ok_out != 0);
        } else {
            float* base = tile.merged ? v_dc + (int64_t)n0 * tile.rs : v_rest + (int64_t)n0 * tile.rs;
            tile_store(base, rows, tile.rs, tile.ls, 1.f / (float)tile.rs, lds, vec_ok_out != 0);
        }
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace gspl

namespace gspl {
int sh_fwd_launch(int N, int C, int degree,
                  const float* dirs, const float* origin,
                  const float* dc, int dc_stride, const float* rest, int rest_stride,
                  const uint8_t* mask, const int32_t* mask32, int flags,
                  float* colors, uint8_t* clamped, void* stream, float* jac) {
    if (N < 0 || C < 1 || degree < 0 || degree > 4) return fail_arg("sh_fwd: bad N/C/degree");
    if (jac && C != 1) return fail_arg("sh_fwd: the direction Jacobian is kept for one camera only");
    if (C > 1 && !origin) return fail_arg("sh_fwd: several cameras need their origins");
    if (N == 0) return GSPL_OK;
    if (!dirs || !dc || !colors || (degree > 0 && !rest)) return fail_arg("sh_fwd: NULL required pointer");
    ShTile tile;
    tile.merged = (degree > 0 && rest == dc + 3 && dc_stride == rest_stride) ? 1 : 0;
    tile.rs = tile.merged ? dc_stride : rest_stride;
    if (degree > 0 && tile.rs < 3 * ((degree + 1) * (degree + 1) - 1) + (tile.merged ? 3 : 0)) return fail_arg("sh_fwd: row stride too small for degree");
    tile.ls = tile.rs | 1;
    const float* base = tile.merged ? dc : rest;
    // every block starts at n0*rs floats; 256*rs*4 bytes is a multiple of 16, so only the base matters
    const int vec_ok = (degree > 0 && aligned16(base)) ? 1 : 0;
    const size_t lds_bytes = degree > 0 ? (size_t)SH_BLOCK * tile.ls * sizeof(float) : 0;
    if (lds_bytes > 160 * 1024) return fail_arg("sh_fwd: coefficient row too long for LDS staging");
    const int grid = (N + SH_BLOCK - 1) / SH_BLOCK;
#define GSPL_SH_FWD(DEG)                                                                                              \
    case DEG: {                                                                                                       \
        if (lds_bytes > 64 * 1024) {                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sh_fwd_kernel<DEG>),                     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);           \
            if (e != hipSuccess) return check_hip(e, "sh_fwd: hipFuncSetAttribute");                                  \
        }                                                                                                             \
        hipLaunchKernelGGL(sh_fwd_kernel<DEG>, dim3(grid), dim3(SH_BLOCK), lds_bytes, (hipStream_t)stream, N, C, dirs, \
                           origin, dc, dc_stride, rest, mask, mask32, flags, tile, vec_ok, colors, clamped, jac);     \
    } break;
    switch (degree) { GSPL_SH_FWD(0) GSPL_SH_FWD(1) GSPL_SH_FWD(2) GSPL_SH_FWD(3) GSPL_SH_FWD(4) }
#undef GSPL_SH_FWD
    return check_launch("sh_fwd");
}
}  // namespace gspl

extern "C" int gspl_sh_fwd(int N, int degree,
                           const float* dirs, const float* origin,
                           const float* dc, int dc_stride, const float* rest, int rest_stride,
                           const uint8_t* mask, int flags,
                           float* colors, uint8_t* clamped, void* stream) {
    return gspl::sh_fwd_launch(N, 1, degree, dirs, origin, dc, dc_stride, rest, rest_stride, mask, nullptr, flags, colors, clamped, stream, nullptr);
}

// C cameras in one launch: origins [C,3], radii [C,N] (radius > 0 = evaluate; may be NULL), colors [C,N,3], clamped [C,N,3].
// The coefficient rows are read once for all cameras.
extern "C" int gspl_sh_fwd_batched(int C, int N, int degree,
                                   const float* means, const float* origins,
                                   const float* dc, int dc_stride, const float* rest, int rest_stride,
                                   const int32_t* radii, int flags,
                                   float* colors, uint8_t* clamped, void* stream) {
    return gspl::sh_fwd_launch(N, C, degree, means, origins, dc, dc_stride, rest, rest_stride, nullptr, radii, flags, colors, clamped, stream, nullptr);
}

namespace gspl {
int sh_bwd_launch(int N, int C, int degree, int n_coeffs,
                  const float* dirs, const float* origin,
                  const float* dc, int dc_stride, const float* rest, int rest_stride,
                  const uint8_t* mask, const int32_t* mask32, int flags, const uint8_t* clamped,
                  const float* v_colors, int vc_stride,
                  float* v_dc, float* v_rest, float* v_dirs, void* stream, const float* jac, const ShAdamHost* adam_host) {
    if (N < 0 || C < 1 || degree < 0 || degree > 4 || n_coeffs < (degree + 1) * (degree + 1)) return fail_arg("sh_bwd: bad N/C/degree/n_coeffs");
    // adam_host: apply the Adam update instead of writing the coefficient gradients; v_dc / v_rest are then the PARAMETERS (the
    // layout is decided on them as it would be on the gradient arrays, which have the parameters' strides)
    ShAdam adam = {};
    if (adam_host) {
        if (C != 1 || !v_dirs) return fail_arg("sh_bwd: the Adam-applying form is the one-camera backward with the direction gradient");
        if (!v_dc || !adam_host->dc.exp_avg || !adam_host->dc.exp_avg_sq || (n_coeffs > 1 && (!v_rest || !adam_host->rest.exp_avg || !adam_host->rest.exp_avg_sq)))
            return fail_arg("sh_bwd: Adam targets missing");
    }
    if (jac && !v_dirs) jac = nullptr;
    if (C > 1 && (v_dirs || !origin)) return fail_arg("sh_bwd: several cameras need their origins and give no direction gradient");
    if (N == 0) return GSPL_OK;
    if (!dirs || !v_colors || !v_dc || (n_coeffs > 1 && !v_rest)) return fail_arg("sh_bwd: NULL required pointer");
    if (v_dirs && !jac && (!dc || (degree > 0 && !rest))) return fail_arg("sh_bwd: v_dirs needs the coefficients (or the forward's Jacobian)");
    ShTile tile;
    // geometry is decided on the OUTPUT arrays (they have the same strides as the inputs)
    tile.merged = (n_coeffs > 1 && v_rest == v_dc + 3 && dc_stride == rest_stride) ? 1 : 0;
    tile.rs = tile.merged ? dc_stride : rest_stride;
    if (n_coeffs > 1 && tile.rs < 3 * (n_coeffs - 1) + (tile.merged ? 3 : 0)) return fail_arg("sh_bwd: row stride too small");
    if (n_coeffs <= 1) tile.rs = 1;
    tile.ls = tile.rs | 1;
    if (v_dirs && !jac && degree > 0) {
        const int in_merged = (rest == dc + 3) ? 1 : 0;
        if (in_merged != tile.merged) return fail_arg("sh_bwd: input/output coefficient layouts differ");
    }
    const float* in_base = tile.merged ? dc : rest;
    float* out_base = tile.merged ? v_dc : v_rest;
    const int vec_in = (in_base && aligned16(in_base)) ? 1 : 0;
    int vec_out = (out_base && aligned16(out_base)) ? 1 : 0;
    if (adam_host) {
        auto target = [](float* p, const ShAdamTargetHost& t) {
            return AdamTarget{p, t.exp_avg, t.exp_avg_sq, AdamHyper{t.lr * (1.f / t.bias_correction1), t.beta1, t.beta2, 1.f / t.bias_correction2_sqrt, t.eps}};
        };
        // merged: one parameter (dc + rest rows in one tensor) -> its moments are `dc`'s; separate: tile = rest, dc on its own
        adam.tile = tile.merged ? target(v_dc, adam_host->dc) : target(v_rest, adam_host->rest);
        adam.dc = target(v_dc, adam_host->dc);
        if (n_coeffs > 1 && !(aligned16(adam.tile.p) && aligned16(adam.tile.m) && aligned16(adam.tile.v))) vec_out = 0;
    }
    const size_t lds_bytes = n_coeffs > 1 ? (size_t)SH_BLOCK * tile.ls * sizeof(float) : 0;
    if (lds_bytes > 160 * 1024) return fail_arg("sh_bwd: coefficient row too long for LDS staging");
    const int grid = (N + SH_BLOCK - 1) / SH_BLOCK;
#define GSPL_SH_BWD(DEG, WD, AD)                                                                                      \
    {                                                                                                                 \
        if (lds_bytes > 64 * 1024) {                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sh_bwd_kernel<DEG, WD, AD>),             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);           \
            if (e != hipSuccess) return check_hip(e, "sh_bwd: hipFuncSetAttribute");                                  \
        }                                                                                                             \
        hipLaunchKernelGGL((sh_bwd_kernel<DEG, WD, AD>), dim3(grid), dim3(SH_BLOCK), lds_bytes, (hipStream_t)stream, N, C, \
                           n_coeffs, dirs, origin, dc, dc_stride, rest, mask, mask32, flags, clamped, v_colors, vc_stride, tile, vec_in, \
                           vec_out, v_dc, v_rest, v_dirs, jac, adam);                                                 \
    }
#define GSPL_SH_BWD_CASE(DEG) \
    case DEG:                 \
        if (adam_host) GSPL_SH_BWD(DEG, true, true) else if (v_dirs) GSPL_SH_BWD(DEG, true, false) else GSPL_SH_BWD(DEG, false, false) break;
    switch (degree) { GSPL_SH_BWD_CASE(0) GSPL_SH_BWD_CASE(1) GSPL_SH_BWD_CASE(2) GSPL_SH_BWD_CASE(3) GSPL_SH_BWD_CASE(4) }
#undef GSPL_SH_BWD_CASE
#undef GSPL_SH_BWD
    return check_launch("sh_bwd");
}
}  // namespace gspl

extern "C" int gspl_sh_bwd(int N, int degree, int n_coeffs,
                           const float* dirs, const float* origin,
                           const float* dc, int dc_stride, const float* rest, int rest_stride,
                           const uint8_t* mask, int flags, const uint8_t* clamped,
                           const float* v_colors, int v_colors_stride,
                           float* v_dc, float* v_rest, float* v_dirs, void* stream) {
    return gspl::sh_bwd_launch(N, 1, degree, n_coeffs, dirs, origin, dc, dc_stride, rest, rest_stride, mask, nullptr, flags, clamped, v_colors,
                               v_colors_stride > 0 ? v_colors_stride : 3, v_dc, v_rest, v_dirs, stream, nullptr, nullptr);
}

// Backward of gspl_sh_fwd_batched: v_colors [C,N,3] (dense) -> v_dc / v_rest summed over the cameras, written once.
extern "C" int gspl_sh_bwd_batched(int C, int N, int degree, int n_coeffs,
                                   const float* means, const float* origins,
                                   int dc_stride, int rest_stride,
                                   const int32_t* radii, int flags, const uint8_t* clamped,
                                   const float* v_colors, float* v_dc, float* v_rest, void* stream) {
    return gspl::sh_bwd_launch(N, C, degree, n_coeffs, means, origins, nullptr, dc_stride, nullptr, rest_stride, nullptr, radii, flags, clamped,
                               v_colors, 3, v_dc, v_rest, nullptr, stream, nullptr, nullptr);
}
