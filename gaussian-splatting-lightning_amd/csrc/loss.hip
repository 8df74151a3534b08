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
// One WAVE = one strip of 64 columns x 32 output rows of one (batch, channel) plane, lanes = columns; the wave streams its 42 input
// rows top to bottom (row-streaming formulation below).  The window is separable: 11 taps along x through a wave-private LDS line,
// 11 along y over a ring of the column's last 11 horizontal results held in registers.  The forward keeps three derivative maps
// (dm/dmu1, dm/ds1, dm/ds12 with the mu1 dependence of the variances folded in); the backward convolves them with the same window:
//     dL/dx(p) = g_ssim * [ conv(dm_dmu1) + 2 x(p) conv(dm_ds1) + y(p) conv(dm_ds12) ](p) + g_l1 * sign(x(p) - y(p)).
// Sums are reduced per wave and then in a fixed order by a second one-workgroup kernel: results are deterministic.
// HBM-bound: forward reads 2 and writes 3 planes, backward reads 5 and writes 1 (4 B per element each), plus the halo rows and columns
// of a strip (1.31 x 1.16), most of which the XCD's L2 serves.  Round 4: 41 + 36 us at 3 x 1080p against 63 + 41 us for the previous
// formulation (32 x 16 tiles staged whole in LDS, two passes with three workgroup barriers).
#include "gspl_device.h"
#include "gspl_host.h"
#include <cmath>

namespace gspl {

static constexpr int LH = 5;             // window half width

struct SsimWindow { float w[11]; };

static SsimWindow make_window() {
    SsimWindow k;
    float sum = 0.f;
    for (int i = 0; i < 11; ++i) { k.w[i] = (float)std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += k.w[i]; }
    for (int i = 0; i < 11; ++i) k.w[i] /= sum;
    return k;
}

typedef float lv2 __attribute__((ext_vector_type(2)));      // (x, y) of the two images: one chain of v_pk_fma_f32 per moment pair

// out[0] = mean |x - y|, out[1] = mean SSIM; one workgroup of 1024 threads, fixed summation order, fp64 accumulation.
// The per-thread loads are issued in batches of eight (a one-load-per-iteration loop is a chain of ~1.5 us global-memory
// round trips).
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

// L1 only (no SSIM term): v_img1 = g_l1 * sign(x - y), element-wise
__global__ __launch_bounds__(256) void loss_l1_bwd_kernel(size_t n, const float* __restrict__ img1, const float* __restrict__ img2,
                                                          const float* __restrict__ v_l1_mean, float scale_l1, float* __restrict__ v_img1) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g_l1 = (v_l1_mean ? v_l1_mean[0] : 1.f) * scale_l1;
    const float d = img1[i] - img2[i];
    v_img1[i] = g_l1 * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));      // torch.abs backward: sign(0) = 0
}

// ================================================================================================================================
// Row-streaming formulation.  One WAVE = one strip of 64 columns x SR output rows of one plane, lanes = columns; the
// wave walks its SR + 10 input rows top to bottom:
//   * a row of both images arrives with one coalesced 256-byte load per image (+ 10 halo columns), three rows ahead of its use;
//   * horizontal pass: the row goes through a wave-private LDS line (74 (x, y) pairs), every lane reads its 11 neighbours and forms
//     the five horizontal moments of ITS column;
//   * vertical pass: the last 11 horizontal results of the column live in REGISTERS (a ring of 11 x 5 values; the row loop is unrolled
//     by 11 so that the ring slots are static) — no second trip through LDS, no vertical halo recomputation inside the strip;
//   * the SSIM map, its three derivative maps and the L1 term of the row that just completed, coalesced stores.
// No workgroup barrier anywhere (the four waves of a workgroup are independent), every input row is loaded once per strip
// (1.31x with SR = 32 against 2.1x for the 32 x 16 tiles above), the horizontal pass runs on 1.31x instead of 1.63x the rows.
#ifndef GSPL_LOSS_SR
#define GSPL_LOSS_SR 32
#endif
#ifndef GSPL_LOSS_SPF
#define GSPL_LOSS_SPF 3
#endif
static constexpr int SW = 64;            // strip width = lanes of a wave
static constexpr int SR = GSPL_LOSS_SR;  // output rows per wave
static constexpr int SWAVES = 4;         // (independent) waves per workgroup
static constexpr int SPF = GSPL_LOSS_SPF; // rows in flight ahead of the one being processed
static constexpr int SLINE = SW + 2 * LH + 2;

__device__ __forceinline__ bool strip_unit(int planes, int strips, int chunks, int& plane, int& strip, int& chunk, int& unit) {
    // workgroup b runs on XCD b % 8 (observed placement): every XCD takes ONE contiguous eighth of the (plane, chunk, strip) sequence,
    // so that the strips and chunks whose halos overlap are served by the same L2
    const int n_wg = (planes * strips * chunks + SWAVES - 1) / SWAVES, per = (n_wg + 7) / 8;
    const int b = blockIdx.x;
    if ((b >> 3) >= per) return false;
    unit = ((b & 7) * per + (b >> 3)) * SWAVES + (threadIdx.x >> 6);
    if (unit >= planes * strips * chunks) return false;
    strip = unit % strips;
    chunk = (unit / strips) % chunks;
    plane = unit / (strips * chunks);
    return true;
}

template <bool TRAIN>
__global__ __launch_bounds__(256) void loss_fwd_rows_kernel(
    int planes, int H, int W, const float* __restrict__ img1, const float* __restrict__ img2, SsimWindow win,
    float* __restrict__ dm_dmu1, float* __restrict__ dm_ds1, float* __restrict__ dm_ds12, float* __restrict__ partials, int strips, int chunks) {
    __shared__ __attribute__((aligned(16))) lv2 s_line[SWAVES][SLINE];
    int plane, strip, chunk, unit;
    if (!strip_unit(planes, strips, chunks, plane, strip, chunk, unit)) return;      // (whole waves; nothing below synchronises workgroups)
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int c0 = strip * SW, y0 = chunk * SR;
    const int rows_out = min(SR, H - y0), n_in = rows_out + 2 * LH;
    const float* p1 = img1 + (size_t)plane * H * W;
    const float* p2 = img2 + (size_t)plane * H * W;
    const int gxa = c0 - LH + l, gxb = c0 + SW - LH + l;      // the lane's column of the line, and (lanes 0..9) its halo column
    const bool xa_in = gxa >= 0 && gxa < W, xb_in = l < 2 * LH && gxb < W;
    const int gxo = c0 + l;                                  // the lane's OUTPUT column
    const bool xo_in = gxo < W;
    // (loads are unconditional at clamped addresses and zeroed afterwards: no divergent branch around a load, so the compiler can
    // count the loads in flight instead of draining them all)
    const int cxa = min(max(gxa, 0), W - 1), cxb = min(gxb, W - 1);
    auto load_row = [&](int r, lv2& a, lv2& b) {
        const int gy = y0 - LH + r;
        const bool yin = r < n_in && gy >= 0 && gy < H;
        const size_t o = (size_t)min(max(gy, 0), H - 1) * W;
        a = (lv2){p1[o + cxa], p2[o + cxa]};      // raw: zeroed when the row is taken from the queue (`row_mask`), not here —
        b = (lv2){p1[o + cxb], p2[o + cxb]};      // a select at load time would make the wave wait for the load at once
        (void)yin;
    };
    auto row_in = [&](int r) { const int gy = y0 - LH + r; return r < n_in && gy >= 0 && gy < H; };
    // rows in flight: slot (r % 11) of a static ring (the row loop is unrolled by 11, so every index is a compile-time constant and
    // the queue needs no register moves); only SPF slots are live at a time
    lv2 qa[11], qb[11];
#pragma unroll
    for (int k = 0; k < SPF; ++k) load_row(k, qa[k], qb[k]);
    lv2 hm[11], hq[11];
    float hx[11];
    float l1_acc = 0.f, ssim_acc = 0.f;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    for (int base = 0; base < n_in; base += 11) {
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int r = base + j;                          // rows past n_in - 1 are zero rows whose results nobody stores
            // the row that is due, and the load of the one SPF rows further down
            const bool yin = row_in(r);
            const lv2 a = (yin && xa_in) ? qa[j] : (lv2){0.f, 0.f}, b = (yin && xb_in) ? qb[j] : (lv2){0.f, 0.f};
            load_row(r + SPF, qa[(j + SPF) % 11], qb[(j + SPF) % 11]);
            // through the wave's LDS line (LDS operations of one wave execute in program order).  (Squares and products are formed per
            // tap: a version that put them through LDS lines of their own — 33 instead of 55 instructions in the tap loop — held 55
            // more registers across it and ran at 48 instead of 41 us.)
            s_line[wv][l] = a;
            if (l < 2 * LH) s_line[wv][SW + l] = b;
            __builtin_amdgcn_wave_barrier();
            lv2 v[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) v[k] = s_line[wv][l + k];
            __builtin_amdgcn_wave_barrier();
            lv2 m = {0.f, 0.f}, q = {0.f, 0.f};
            float x = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const lv2 wk = {win.w[k], win.w[k]};
                m = __builtin_elementwise_fma(wk, v[k], m);
                q = __builtin_elementwise_fma(wk, v[k] * v[k], q);
                x = fmaf(win.w[k], v[k].x * v[k].y, x);
            }
            hm[j] = m; hq[j] = q; hx[j] = x;
            if (r >= LH && r < LH + rows_out && xo_in) l1_acc += fabsf(v[LH].x - v[LH].y);
            if (r >= 2 * LH && r < n_in) {
                // vertical pass over the ring: slot (j + 1 + k) % 11 holds input row r - 10 + k
                lv2 mu = {0.f, 0.f}, ee = {0.f, 0.f};
                float e12 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const int slot = (j + 1 + k) % 11;
                    const lv2 wk = {win.w[k], win.w[k]};
                    mu = __builtin_elementwise_fma(wk, hm[slot], mu);
                    ee = __builtin_elementwise_fma(wk, hq[slot], ee);
                    e12 = fmaf(win.w[k], hx[slot], e12);
                }
                const int gy = y0 + r - 2 * LH;
                if (xo_in) {
                    const float mu1 = mu.x, mu2 = mu.y;
                    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                    const float s1 = ee.x - mu1_sq, s2 = ee.y - mu2_sq, s12 = e12 - mu12;
                    const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cd = mu1_sq + mu2_sq + C1, Dd = s1 + s2 + C2;
                    const float inv_c = __builtin_amdgcn_rcpf(Cd), inv_d = __builtin_amdgcn_rcpf(Dd);
                    const float inv_cd = inv_c * inv_d;
                    const float mval = (A * B) * inv_cd;
                    ssim_acc += mval;
                    if (TRAIN) {
                        const float d_s1 = -mval * inv_d;
                        const float d_s12 = 2.f * A * inv_cd;
                        const float d_mu1 = 2.f * mu2 * B * inv_cd - 2.f * mu1 * mval * inv_c - 2.f * mu1 * d_s1 - mu2 * d_s12;
                        const size_t o = (size_t)plane * H * W + (size_t)gy * W + gxo;
                        dm_dmu1[o] = d_mu1; dm_ds1[o] = d_s1; dm_ds12[o] = d_s12;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { l1_acc += __shfl_xor(l1_acc, off); ssim_acc += __shfl_xor(ssim_acc, off); }
    if (l == 0) {
        partials[(size_t)unit * 2 + 0] = l1_acc;
        partials[(size_t)unit * 2 + 1] = ssim_acc;
    }
}

// v_img1 = g_ssim * (conv(dm_dmu1) + 2 x conv(dm_ds1) + y conv(dm_ds12)) + g_l1 * sign(x - y), row-streaming (see above)
__global__ __launch_bounds__(256) void loss_bwd_rows_kernel(
    int planes, int H, int W, const float* __restrict__ img1, const float* __restrict__ img2, SsimWindow win,
    const float* __restrict__ dm_dmu1, const float* __restrict__ dm_ds1, const float* __restrict__ dm_ds12,
    const float* __restrict__ v_l1_mean, const float* __restrict__ v_ssim_mean, float scale_l1, float scale_ssim,
    float* __restrict__ v_img1, int strips, int chunks) {
    __shared__ __attribute__((aligned(16))) lv2 s_la[SWAVES][SLINE];
    __shared__ float s_lc[SWAVES][SLINE];
    int plane, strip, chunk, unit;
    if (!strip_unit(planes, strips, chunks, plane, strip, chunk, unit)) return;
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int c0 = strip * SW, y0 = chunk * SR;
    const int rows_out = min(SR, H - y0), n_in = rows_out + 2 * LH;
    const size_t pbase = (size_t)plane * H * W;
    const int gxa = c0 - LH + l, gxb = c0 + SW - LH + l;
    const bool xa_in = gxa >= 0 && gxa < W, xb_in = l < 2 * LH && gxb < W;
    const int gxo = c0 + l;
    const bool xo_in = gxo < W;
    const float g_l1 = (v_l1_mean ? v_l1_mean[0] : 1.f) * scale_l1;
    const float g_ssim = (v_ssim_mean ? v_ssim_mean[0] : 1.f) * scale_ssim;
    struct Row { lv2 a, b; float ca, cb, xv, yv; };      // + the images' pixels of the OUTPUT row this input row completes
    const int cxa = min(max(gxa, 0), W - 1), cxb = min(gxb, W - 1);
    auto load_row = [&](int r, Row& q) {      // unconditional loads at clamped addresses, zeroed afterwards (see the forward)
        const int gy = y0 - LH + r;
        const bool yin = r < n_in && gy >= 0 && gy < H;
        const size_t o = pbase + (size_t)min(max(gy, 0), H - 1) * W;
        (void)yin;
        q.a = (lv2){dm_dmu1[o + cxa], dm_ds1[o + cxa]};      // raw: zeroed when the row is taken from the queue
        q.ca = dm_ds12[o + cxa];
        q.b = (lv2){dm_dmu1[o + cxb], dm_ds1[o + cxb]};
        q.cb = dm_ds12[o + cxb];
        const size_t oo = pbase + (size_t)min(max(gy - LH, 0), H - 1) * W + min(gxo, W - 1);
        q.xv = img1[oo]; q.yv = img2[oo];
    };
    auto row_in = [&](int r) { const int gy = y0 - LH + r; return r < n_in && gy >= 0 && gy < H; };
    Row q[11];                                               // static ring of rows in flight (see the forward)
#pragma unroll
    for (int k = 0; k < SPF; ++k) load_row(k, q[k]);
    lv2 ha[11];
    float hc[11];
    for (int base = 0; base < n_in; base += 11) {
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int r = base + j;
            Row cur = q[j];
            {
                const bool yin = row_in(r), ia = yin && xa_in, ib = yin && xb_in;
                cur.a = ia ? cur.a : (lv2){0.f, 0.f}; cur.ca = ia ? cur.ca : 0.f;
                cur.b = ib ? cur.b : (lv2){0.f, 0.f}; cur.cb = ib ? cur.cb : 0.f;
            }
            load_row(r + SPF, q[(j + SPF) % 11]);
            const int gy = y0 + r - 2 * LH;
            const float xv = cur.xv, yv = cur.yv;
            const bool out_row = r >= 2 * LH && r < n_in;
            s_la[wv][l] = cur.a; s_lc[wv][l] = cur.ca;
            if (l < 2 * LH) { s_la[wv][SW + l] = cur.b; s_lc[wv][SW + l] = cur.cb; }
            __builtin_amdgcn_wave_barrier();
            lv2 va[11];
            float vc[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) { va[k] = s_la[wv][l + k]; vc[k] = s_lc[wv][l + k]; }
            __builtin_amdgcn_wave_barrier();
            lv2 m = {0.f, 0.f};
            float x = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                m = __builtin_elementwise_fma((lv2){win.w[k], win.w[k]}, va[k], m);
                x = fmaf(win.w[k], vc[k], x);
            }
            ha[j] = m; hc[j] = x;
            if (out_row && xo_in) {
                lv2 ca = {0.f, 0.f};
                float cc = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const int slot = (j + 1 + k) % 11;
                    ca = __builtin_elementwise_fma((lv2){win.w[k], win.w[k]}, ha[slot], ca);
                    cc = fmaf(win.w[k], hc[slot], cc);
                }
                float g = g_ssim * (ca.x + 2.f * xv * ca.y + yv * cc);
                if (scale_l1 != 0.f) {
                    const float d = xv - yv;
                    g += g_l1 * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));      // torch.abs backward: sign(0) = 0
                }
                v_img1[pbase + (size_t)gy * W + gxo] = g;
            }
        }
    }
}

}  // namespace gspl

extern "C" size_t gspl_loss_workspace_bytes(int planes, int H, int W) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    // one pair of partial sums per wave of the forward
    const size_t units = (size_t)planes * ((H + gspl::SR - 1) / gspl::SR) * ((W + gspl::SW - 1) / gspl::SW);
    return units * 2 * sizeof(float);
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
    const int strips = (W + SW - 1) / SW, chunks = (H + SR - 1) / SR;
    const int n_units = strips * chunks * planes;
    const dim3 grid((unsigned)((((n_units + SWAVES - 1) / SWAVES + 7) / 8) * 8)), block(64 * SWAVES);      // workgroup -> units: strip_unit
    float* partials = (float*)workspace;
    if (train)
        hipLaunchKernelGGL(loss_fwd_rows_kernel<true>, grid, block, 0, s, planes, H, W, img1, img2, win, dm_dmu1, dm_ds1, dm_ds12, partials, strips, chunks);
    else
        hipLaunchKernelGGL(loss_fwd_rows_kernel<false>, grid, block, 0, s, planes, H, W, img1, img2, win, dm_dmu1, dm_ds1, dm_ds12, partials, strips, chunks);
    int rc = check_launch("loss_fwd");
    if (rc != GSPL_OK) return rc;
    const int n_tiles = n_units;
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
    const float inv_n = (float)(1.0 / ((double)planes * H * W));
    if (dm_dmu1) {
        const int strips = (W + SW - 1) / SW, chunks = (H + SR - 1) / SR;
        hipLaunchKernelGGL(loss_bwd_rows_kernel, dim3((unsigned)((((strips * chunks * planes + SWAVES - 1) / SWAVES + 7) / 8) * 8)), dim3(64 * SWAVES), 0, s, planes, H, W,
                           img1, img2, win, dm_dmu1, dm_ds1, dm_ds12, v_l1_mean, v_ssim_mean, weight_l1 * inv_n, weight_ssim * inv_n, v_img1, strips, chunks);
    } else {
        const size_t n = (size_t)planes * H * W;
        hipLaunchKernelGGL(loss_l1_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, img1, img2, v_l1_mean, weight_l1 * inv_n, v_img1);
    }
    return check_launch("loss_bwd");
}
