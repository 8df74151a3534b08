// loss.hip — fused photometric loss terms: mean |x - y| and mean SSIM of two image batches, forward and backward (gfx950).
//
// "Next" row of SURVEY.md §8f (rank 2): the step either side of the rasterizer.  Replaces, for the training loss
//     loss = (1 - lambda) * L1 + lambda * (1 - SSIM)            (internal/metrics/vanilla_metrics.py:57-70)
// the reference's `l1_loss` + `ssim` (internal/utils/ssim.py:17-63: five depth-wise 11x11 Gaussian convolutions and
// a dozen element-wise kernels over [3,H,W]) and its opt-in `fused_ssim` CUDA package (vanilla_metrics.py:35-39).
// SSIM restated from internal/utils/ssim.py:23-63: window = outer product of an 11-tap Gaussian (sigma 1.5), zero
// padding ("same"), C1 = 0.01^2, C2 = 0.03^2,
//     m = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)),   mean over all elements.
//
// One workgroup = one 32x16 output tile of one (batch, channel) plane.  The 42x26 input patch of both images goes to
// LDS once; the window is separable, so the five local moments cost 11 taps along x (into LDS) and 11 along y instead
// of 121 taps each.  The forward keeps three derivative maps (dm/dmu1, dm/ds1, dm/ds12 with the mu1 dependence of the
// variances folded in); the backward convolves them with the same window:
//     dL/dx(p) = g_ssim * [ conv(dm_dmu1) + 2 x(p) conv(dm_ds1) + y(p) conv(dm_ds12) ](p) + g_l1 * sign(x(p) - y(p)).
// Sums are reduced per workgroup and then in a fixed order by a second one-workgroup kernel: results are deterministic.
// HBM-bound: forward reads 2 and writes 3 planes, backward reads 5 and writes 1 (4 B per element each).
#include "gspl_device.h"
#include "gspl_host.h"
#include <cmath>

namespace gspl {

static constexpr int LTX = 32, LTY = 16; // output tile: 32 columns (128-byte rows in global memory) x 16 rows
static constexpr int LH = 5;             // window half width
static constexpr int LPX = LTX + 2 * LH; // input patch 42 x 26
static constexpr int LPY = LTY + 2 * LH;

struct SsimWindow { float w[11]; };

static SsimWindow make_window() {
    SsimWindow k;
    float sum = 0.f;
    for (int i = 0; i < 11; ++i) { k.w[i] = (float)std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += k.w[i]; }
    for (int i = 0; i < 11; ++i) k.w[i] /= sum;
    return k;
}

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
    const int t = threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((t & 63) == 0) s_red[t >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// Both passes use a register sliding window — a thread produces FOUR adjacent outputs from 14 inputs fetched with
// 16-byte LDS reads (a tap-per-ds_read version was bound by LDS instruction issue) — and PACKED fp32 math: the two
// images are interleaved in LDS as (x, y) pairs, so (mu1, mu2) and (E[x^2], E[y^2]) are each one chain of
// v_pk_fma_f32 and only E[xy] is a scalar chain: 33 instead of 55 instructions per output and pass.  Horizontal results
// are stored transposed ([column][row]) so that the vertical pass reads its 14 rows the same way.
typedef float lv2 __attribute__((ext_vector_type(2)));
// LDS strides, chosen with a bank-conflict model of the access patterns below (64 banks for 8/16-byte accesses, the b128 lane
// groups of MI355X_MICROARCH.md; horizontal threads mapped row-fastest): LDS cycles per workgroup 1040 -> 524 (forward),
// 956 -> 512 (backward) against 384 / 352 conflict-free.  The previous strides (44 / 28, segment-fastest mapping) made 70 % of
// the LDS-active cycles conflicts (rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
static constexpr int LPS = 46;           // padded patch row of (x, y) pairs: segments of 4 outputs read 16 pixels (<= column 43)
static constexpr int LPSC = 44;          // padded patch row of the scalar plane of the backward (16-byte aligned rows)
static constexpr int LRS = 30;           // padded column of the transposed horizontal pair results (26 rows used)
static constexpr int LRSC = 28;          // ... of the transposed scalar results
static constexpr int LHSEG = LTX / 4;    // horizontal pass: segments of 4 columns per patch row
static constexpr int LVSEG = LTY / 4;    // vertical pass: segments of 4 rows per column

// o[j] = sum_k w[k] * v[j + k], j = 0..3 (v: 14 used of 16)
__device__ __forceinline__ void window4(const float (&v)[16], const SsimWindow& win, float (&o)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) acc = fmaf(win.w[k], v[j + k], acc);
        o[j] = acc;
    }
}
__device__ __forceinline__ void window4(const lv2 (&v)[16], const SsimWindow& win, lv2 (&o)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lv2 acc = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; ++k) acc = __builtin_elementwise_fma((lv2){win.w[k], win.w[k]}, v[j + k], acc);
        o[j] = acc;
    }
}
__device__ __forceinline__ void load16(const float* p, float (&v)[16]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 t = q[i]; v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
}
__device__ __forceinline__ void load16(const lv2* p, lv2 (&v)[16]) {       // 14 pairs used: seven 16-byte reads, the last pair is padding
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float4 t = q[i]; v[2 * i] = (lv2){t.x, t.y}; v[2 * i + 1] = (lv2){t.z, t.w}; }
}

// Workgroup -> tile.  The dispatcher places workgroup b on XCD b % 8 (MI355X_MICROARCH.md): each XCD gets ONE contiguous
// eighth of the (plane, row, column) tile sequence and walks it in order, so that the tiles whose halos overlap (left / right
// neighbours: the same 128-byte lines; up / down: 10 of 26 patch rows) are served by the same L2 shortly after one another.
// With the plain 3-D grid neighbouring tiles sit on different XCDs and every XCD's L2 fetches the shared lines again.
// Returns false for the padding workgroups of the last eighth.
__device__ __forceinline__ bool loss_tile(int tiles_x, int tiles_y, int planes, int& tx, int& ty, int& plane, size_t& tile) {
    const int n = tiles_x * tiles_y * planes, per = (n + 7) / 8;
    const int b = blockIdx.x;
    const int t = (b & 7) * per + (b >> 3);
    if ((b >> 3) >= per || t >= n) return false;
    plane = t / (tiles_x * tiles_y);
    const int rem = t - plane * (tiles_x * tiles_y);
    ty = rem / tiles_x;
    tx = rem - ty * tiles_x;
    tile = (size_t)t;
    return true;
}

// partials[(plane * tiles + tile) * 2 + {0,1}] = tile sums of |x - y| and of the SSIM map
template <bool TRAIN>
__global__ __launch_bounds__(256) void loss_fwd_kernel(
    int planes, int H, int W, const float* __restrict__ img1, const float* __restrict__ img2, SsimWindow win,
    float* __restrict__ dm_dmu1, float* __restrict__ dm_ds1, float* __restrict__ dm_ds12, float* __restrict__ partials) {
    // LDS: the input patch, then (after every thread has its inputs in registers) the transposed horizontal results on
    // top of it — 18 KB per workgroup instead of 27, i.e. 8 instead of 5 workgroups per CU
    constexpr int PATCH_BYTES = LPY * LPS * (int)sizeof(lv2);
    constexpr int HRES_BYTES = LTX * (LRS * 2 * (int)sizeof(lv2) + LRSC * (int)sizeof(float));
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[PATCH_BYTES > HRES_BYTES ? PATCH_BYTES : HRES_BYTES];
    lv2 (*s_p)[LPS] = reinterpret_cast<lv2 (*)[LPS]>(s_raw);                                      // (x, y)
    lv2 (*s_hm)[LRS] = reinterpret_cast<lv2 (*)[LRS]>(s_raw);                                     // [column][row] (mu1, mu2)
    lv2 (*s_hq)[LRS] = reinterpret_cast<lv2 (*)[LRS]>(s_raw + LTX * LRS * sizeof(lv2));           // (E[x^2], E[y^2])
    float (*s_hc)[LRSC] = reinterpret_cast<float (*)[LRSC]>(s_raw + 2 * LTX * LRS * sizeof(lv2)); // E[xy]
    __shared__ float s_red[4];
    int tile_x, tile_y, plane;
    size_t tile_index;
    if (!loss_tile((W + LTX - 1) / LTX, (H + LTY - 1) / LTY, planes, tile_x, tile_y, plane, tile_index)) return;
    const int x0 = tile_x * LTX, y0 = tile_y * LTY;
    const int t = threadIdx.x;
    const float* p1 = img1 + (size_t)plane * H * W;
    const float* p2 = img2 + (size_t)plane * H * W;
    // patch load: all of a thread's global loads are issued before the first LDS store (the rolled loop was a chain of
    // five ~1.5 us round trips per workgroup, which at 5 workgroups per CU bounded the whole kernel)
    // (thread = one patch column and every fourth patch row: the column's bounds test and address are computed once, a row step is
    // one add — the linear-index version spent ~25 instructions per load on div/mod by the row stride and five range tests)
    constexpr int NLOAD = (LPY + 3) / 4;
    lv2 pv[NLOAD];
    const int pcol = t & 63, prow = t >> 6;
    {
        const int gx = x0 + pcol - LH;
        const bool col_in = pcol < LPX && gx >= 0 && gx < W;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int r = prow + 4 * k;
            const int gy = y0 + r - LH;
            const bool in = col_in && r < LPY && gy >= 0 && gy < H;
            const size_t o = (size_t)(in ? gy : 0) * W + (in ? gx : 0);
            pv[k] = in ? (lv2){p1[o], p2[o]} : (lv2){0.f, 0.f};
        }
    }
    if (pcol < LPS) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int r = prow + 4 * k;
            if (r < LPY) s_p[r][pcol] = pv[k];
        }
    }
    __syncthreads();
    // horizontal pass: 26 rows x 8 segments of 4 columns (208 threads, consecutive lanes = consecutive rows: the transposed
    // stores below then go to consecutive addresses); the L1 term of the segment's own pixels rides along
    float l1_acc = 0.f, ssim_acc = 0.f;
    const bool hthread = t < LPY * LHSEG;
    const int hr = t % LPY, hc0 = (t / LPY) * 4;
    lv2 hv[16];
    if (hthread) {
        load16(&s_p[hr][hc0], hv);
        if (hr >= LH && hr < LH + LTY) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gx = x0 + hc0 + j, gy = y0 + hr - LH;
                if (gx < W && gy < H) l1_acc += fabsf(hv[LH + j].x - hv[LH + j].y);
            }
        }
    }
    __syncthreads();                 // every thread holds its inputs: the patch may be overwritten
    if (hthread) {
        lv2 q[16], o[4];
        float xy[16], oc[4];
        window4(hv, win, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) s_hm[hc0 + j][hr] = o[j];
#pragma unroll
        for (int i = 0; i < 16; ++i) { q[i] = hv[i] * hv[i]; xy[i] = hv[i].x * hv[i].y; }
        window4(q, win, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) s_hq[hc0 + j][hr] = o[j];
        window4(xy, win, oc);
#pragma unroll
        for (int j = 0; j < 4; ++j) s_hc[hc0 + j][hr] = oc[j];
    }
    __syncthreads();
    // vertical pass: 32 columns x 4 segments of 4 rows (128 threads), each thread finishes 4 pixels
    if (t < LTX * LVSEG) {
        const int c = t % LTX, r0 = (t / LTX) * 4;
        lv2 v[16], mu[4], ee[4];
        float vc[16], e12[4];
        load16(&s_hm[c][r0], v); window4(v, win, mu);
        load16(&s_hq[c][r0], v); window4(v, win, ee);
        load16(&s_hc[c][r0], vc); window4(vc, win, e12);
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = x0 + c, gy = y0 + r0 + j;
            if (gx < W && gy < H) {
                const float mu1 = mu[j].x, mu2 = mu[j].y;
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float s1 = ee[j].x - mu1_sq, s2 = ee[j].y - mu2_sq, s12 = e12[j] - mu12;
                const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cd = mu1_sq + mu2_sq + C1, Dd = s1 + s2 + C2;
                // v_rcp_f32 (1 ulp) instead of four IEEE divisions (~10 instructions each) per pixel
                const float inv_c = __builtin_amdgcn_rcpf(Cd), inv_d = __builtin_amdgcn_rcpf(Dd);
                const float inv_cd = inv_c * inv_d;
                const float m = (A * B) * inv_cd;
                ssim_acc += m;
                if (TRAIN) {
                    // partial derivatives of m w.r.t. (mu1, s1, s12), then the mu1 dependence of s1 = E[x^2] - mu1^2 and
                    // s12 = E[xy] - mu1 mu2 folded into the first, so that the backward only needs dE-type convolutions
                    const float d_s1 = -m * inv_d;                      // dm/ds1
                    const float d_s12 = 2.f * A * inv_cd;               // dm/ds12
                    const float d_mu1 = 2.f * mu2 * B * inv_cd - 2.f * mu1 * m * inv_c - 2.f * mu1 * d_s1 - mu2 * d_s12;
                    const size_t o = (size_t)plane * H * W + (size_t)gy * W + gx;
                    dm_dmu1[o] = d_mu1; dm_ds1[o] = d_s1; dm_ds12[o] = d_s12;
                }
            }
        }
    }
    const float l1_sum = block_sum_256(l1_acc, s_red);
    const float ssim_sum = block_sum_256(ssim_acc, s_red);
    if (t == 0) {
        partials[tile_index * 2 + 0] = l1_sum;
        partials[tile_index * 2 + 1] = ssim_sum;
    }
}

// out[0] = mean |x - y|, out[1] = mean SSIM; one workgroup of 1024 threads, fixed summation order, fp64 accumulation.
// The per-thread loads are issued in batches of eight (a one-load-per-iteration loop is a chain of ~1.5 us global-memory
// round trips: it took 100 us for the 24 480 tiles of a 1080p image).
__global__ __launch_bounds__(1024) void loss_reduce_kernel(int n_tiles, const float2* __restrict__ partials, double inv_count, float* __restrict__ out,
                                                           float w_l1, float w_ssim, int write_loss) {
    __shared__ double s_a[16], s_b[16];
    double a = 0.0, b = 0.0;
    const int t = threadIdx.x;
    for (int base = 0; base < n_tiles; base += 16 * 1024) {
        float2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = base + k * 1024 + t;
            v[k] = (i < n_tiles) ? partials[i] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) { a += (double)v[k].x; b += (double)v[k].y; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    if ((t & 63) == 0) { s_a[t >> 6] = a; s_b[t >> 6] = b; }
    __syncthreads();
    if (t == 0) {
        double ta = 0.0, tb = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { ta += s_a[k]; tb += s_b[k]; }
        const float l1 = (float)(ta * inv_count), ssim = (float)(tb * inv_count);
        out[0] = l1; out[1] = ssim;
        if (write_loss) out[2] = w_l1 * l1 + w_ssim * (1.f - ssim);      // the training loss of vanilla_metrics.py:66-68
    }
}

// v_img1 = g_ssim * (conv(dm_dmu1) + 2 x conv(dm_ds1) + y conv(dm_ds12)) + g_l1 * sign(x - y); g_* read from device
// scalars (upstream gradients of the two means, already divided by the element count by the caller's scale factors)
__global__ __launch_bounds__(256) void loss_bwd_kernel(
    int planes, int H, int W, const float* __restrict__ img1, const float* __restrict__ img2, SsimWindow win,
    const float* __restrict__ dm_dmu1, const float* __restrict__ dm_ds1, const float* __restrict__ dm_ds12,
    const float* __restrict__ v_l1_mean, const float* __restrict__ v_ssim_mean, float scale_l1, float scale_ssim,
    float* __restrict__ v_img1) {
    constexpr int PATCH_BYTES = LPY * (LPS * (int)sizeof(lv2) + LPSC * (int)sizeof(float));
    constexpr int HRES_BYTES = LTX * (LRS * (int)sizeof(lv2) + LRSC * (int)sizeof(float));
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[PATCH_BYTES > HRES_BYTES ? PATCH_BYTES : HRES_BYTES];
    lv2 (*s_da)[LPS] = reinterpret_cast<lv2 (*)[LPS]>(s_raw);                                     // (dm_dmu1, dm_ds1)
    float (*s_dc)[LPSC] = reinterpret_cast<float (*)[LPSC]>(s_raw + LPY * LPS * sizeof(lv2));     // dm_ds12
    lv2 (*s_ha)[LRS] = reinterpret_cast<lv2 (*)[LRS]>(s_raw);                                     // [column][row], over the patch
    float (*s_hc)[LRSC] = reinterpret_cast<float (*)[LRSC]>(s_raw + LTX * LRS * sizeof(lv2));
    int tile_x, tile_y, plane;
    size_t tile_index;
    if (!loss_tile((W + LTX - 1) / LTX, (H + LTY - 1) / LTY, planes, tile_x, tile_y, plane, tile_index)) return;
    const int x0 = tile_x * LTX, y0 = tile_y * LTY;
    const int t = threadIdx.x;
    const size_t pbase = (size_t)plane * H * W;
    const bool want_ssim = dm_dmu1 != nullptr;
    const float g_l1 = (v_l1_mean ? v_l1_mean[0] : 1.f) * scale_l1;
    if (!want_ssim) {            // L1 only: element-wise
        for (int i = t; i < LTX * LTY; i += 256) {
            const int gx = x0 + i % LTX, gy = y0 + i / LTX;
            if (gx < W && gy < H) {
                const size_t o = pbase + (size_t)gy * W + gx;
                const float d = img1[o] - img2[o];
                v_img1[o] = g_l1 * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
            }
        }
        return;
    }
    const float g_ssim = (v_ssim_mean ? v_ssim_mean[0] : 1.f) * scale_ssim;
    constexpr int NLOAD = (LPY + 3) / 4;            // thread = one patch column, every fourth row (see the forward)
    lv2 pa[NLOAD];
    float pc[NLOAD];
    const int pcol = t & 63, prow = t >> 6;
    {
        const int gx = x0 + pcol - LH;
        const bool col_in = pcol < LPX && gx >= 0 && gx < W;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {       // all global loads in flight before the first LDS store
            const int r = prow + 4 * k;
            const int gy = y0 + r - LH;
            const bool in = col_in && r < LPY && gy >= 0 && gy < H;
            const size_t o = pbase + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
            pa[k] = in ? (lv2){dm_dmu1[o], dm_ds1[o]} : (lv2){0.f, 0.f};
            pc[k] = in ? dm_ds12[o] : 0.f;
        }
    }
    if (pcol < LPS) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int r = prow + 4 * k;
            if (r < LPY) {
                s_da[r][pcol] = pa[k];
                if (pcol < LPSC) s_dc[r][pcol] = pc[k];
            }
        }
    }
    __syncthreads();
    {
        const bool hthread = t < LPY * LHSEG;
        const int r = t % LPY, c0 = (t / LPY) * 4;      // row-fastest: see the forward
        lv2 v[16], o[4];
        float vc[16], oc[4];
        if (hthread) { load16(&s_da[r][c0], v); load16(&s_dc[r][c0], vc); }
        __syncthreads();             // every thread holds its inputs: the patch may be overwritten
        if (hthread) {
            window4(v, win, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) s_ha[c0 + j][r] = o[j];
            window4(vc, win, oc);
#pragma unroll
            for (int j = 0; j < 4; ++j) s_hc[c0 + j][r] = oc[j];
        }
    }
    __syncthreads();
    if (t < LTX * LVSEG) {
        const int c = t % LTX, r0 = (t / LTX) * 4;
        lv2 v[16], ca[4];
        float vc[16], cc[4];
        load16(&s_ha[c][r0], v); window4(v, win, ca);
        load16(&s_hc[c][r0], vc); window4(vc, win, cc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = x0 + c, gy = y0 + r0 + j;
            if (gx < W && gy < H) {
                const size_t o = pbase + (size_t)gy * W + gx;
                const float xv = img1[o], yv = img2[o];
                float g = g_ssim * (ca[j].x + 2.f * xv * ca[j].y + yv * cc[j]);
                if (scale_l1 != 0.f) {
                    const float d = xv - yv;
                    g += g_l1 * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));      // torch.abs backward: sign(0) = 0
                }
                v_img1[o] = g;
            }
        }
    }
}

}  // namespace gspl

extern "C" size_t gspl_loss_workspace_bytes(int planes, int H, int W) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    const size_t tiles = (size_t)planes * ((H + gspl::LTY - 1) / gspl::LTY) * ((W + gspl::LTX - 1) / gspl::LTX);
    return tiles * 2 * sizeof(float);
}

static int loss_fwd_common(int planes, int H, int W, const float* img1, const float* img2, float w_l1, float w_ssim, int write_loss,
                           float* out_means, float* dm_dmu1, float* dm_ds1, float* dm_ds12,
                           void* workspace, size_t workspace_bytes, void* stream, const char* who) {
    using namespace gspl;
    if (planes <= 0 || H <= 0 || W <= 0) return fail_arg(who);
    if (!img1 || !img2 || !out_means || !workspace) return fail_arg(who);
    const bool train = dm_dmu1 != nullptr;
    if (train && (!dm_ds1 || !dm_ds12)) return fail_arg(who);
    if (workspace_bytes < gspl_loss_workspace_bytes(planes, H, W)) return fail_ws(who);
    if (planes > 65535) return fail_arg(who);
    static const SsimWindow win = make_window();
    hipStream_t s = (hipStream_t)stream;
    const int n_wg = ((W + LTX - 1) / LTX) * ((H + LTY - 1) / LTY) * planes;
    const dim3 grid((unsigned)((n_wg + 7) / 8 * 8)), block(256);      // workgroup -> tile: loss_tile
    float* partials = (float*)workspace;
    if (train)
        hipLaunchKernelGGL(loss_fwd_kernel<true>, grid, block, 0, s, planes, H, W, img1, img2, win, dm_dmu1, dm_ds1, dm_ds12, partials);
    else
        hipLaunchKernelGGL(loss_fwd_kernel<false>, grid, block, 0, s, planes, H, W, img1, img2, win, dm_dmu1, dm_ds1, dm_ds12, partials);
    int rc = check_launch("loss_fwd");
    if (rc != GSPL_OK) return rc;
    const int n_tiles = n_wg;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, s, n_tiles, (const float2*)partials,
                       1.0 / ((double)planes * H * W), out_means, w_l1, w_ssim, write_loss);
    return check_launch("loss_reduce");
}

extern "C" int gspl_loss_l1_ssim_fwd(int planes, int H, int W, const float* img1, const float* img2,
                                     float* out_means, float* dm_dmu1, float* dm_ds1, float* dm_ds12,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return loss_fwd_common(planes, H, W, img1, img2, 0.f, 0.f, 0, out_means, dm_dmu1, dm_ds1, dm_ds12, workspace, workspace_bytes, stream,
                           "loss_l1_ssim_fwd: bad argument");
}

extern "C" int gspl_loss_photometric_fwd(int planes, int H, int W, const float* img1, const float* img2,
                                         float weight_l1, float weight_ssim, float* out_terms,
                                         float* dm_dmu1, float* dm_ds1, float* dm_ds12,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    return loss_fwd_common(planes, H, W, img1, img2, weight_l1, weight_ssim, 1, out_terms, dm_dmu1, dm_ds1, dm_ds12, workspace, workspace_bytes,
                           stream, "loss_photometric_fwd: bad argument");
}

extern "C" int gspl_loss_l1_ssim_bwd(int planes, int H, int W, const float* img1, const float* img2,
                                     const float* dm_dmu1, const float* dm_ds1, const float* dm_ds12,
                                     const float* v_l1_mean, const float* v_ssim_mean, float weight_l1, float weight_ssim,
                                     float* v_img1, void* stream) {
    using namespace gspl;
    if (planes <= 0 || H <= 0 || W <= 0) return fail_arg("loss_l1_ssim_bwd: bad sizes");
    if (!img1 || !img2 || !v_img1) return fail_arg("loss_l1_ssim_bwd: NULL required pointer");
    if ((dm_dmu1 == nullptr) != (dm_ds1 == nullptr) || (dm_dmu1 == nullptr) != (dm_ds12 == nullptr))
        return fail_arg("loss_l1_ssim_bwd: the three derivative maps go together");
    if (planes > 65535) return fail_arg("loss_l1_ssim_bwd: more than 65535 planes");
    static const SsimWindow win = make_window();
    hipStream_t s = (hipStream_t)stream;
    const int n_wg = ((W + LTX - 1) / LTX) * ((H + LTY - 1) / LTY) * planes;
    const dim3 grid((unsigned)((n_wg + 7) / 8 * 8)), block(256);      // workgroup -> tile: loss_tile
    const float inv_n = (float)(1.0 / ((double)planes * H * W));
    hipLaunchKernelGGL(loss_bwd_kernel, grid, block, 0, s, planes, H, W, img1, img2, win, dm_dmu1, dm_ds1, dm_ds12,
                       v_l1_mean, v_ssim_mean, weight_l1 * inv_n, (dm_dmu1 ? weight_ssim : 0.f) * inv_n, v_img1);
    return check_launch("loss_bwd");
}
