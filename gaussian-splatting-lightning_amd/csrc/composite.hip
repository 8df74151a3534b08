// composite.hip — 16x16-tile alpha compositing, forward and backward (gfx950, wave64).
//
// Replaces gsplat `rasterize_to_pixels` (internal/renderers/gsplat_v1_renderer.py:588-601), v0
// `rasterize_gaussians` (gsplat_renderer.py:86-99, pypreprocess_gsplat_renderer.py:45-58) and the
// render stage of the Inria `GaussianRasterizer` (vanilla_renderer.py:111-120), forward and the
// autograd backward the reference enters through `manual_backward` (gaussian_splatting.py:380).
// Neither CUDA package is vendored in the reference; the algorithm restated here is the published
// 3DGS compositing rule with the per-API constants of SURVEY.md Appendix B (ModeTraits).
//
// Design (MI355X-first)
//   * workgroup = one 16x16 tile = 4 wave64; each WAVE owns an 8x8 pixel quadrant, so that the
//     64 lanes that share an exec mask are spatially compact: a splat whose footprint misses the
//     quadrant is rejected with one wave-uniform ballot branch, and early termination is decided
//     per quadrant instead of per tile.
//   * the tile's depth-sorted splat list is gathered in chunks of 256 records into LDS
//     (one record per lane, 36 B: xy, 0.5*conic.a, conic.b, 0.5*conic.c, opacity, colour[D]) and
//     then read back with wave-uniform (broadcast) LDS reads.
//   * backward walks the list back-to-front from each pixel's last contributor, reduces the 9..11
//     per-splat gradient values over the wave with DPP row operations (no LDS traffic), combines
//     the four waves with one-lane LDS atomics, and issues ONE fp32 L2 atomic per value per
//     (tile, splat) — 36 B per intersection, the algorithmic minimum of SURVEY.md §8d.
//   * workgroup -> tile mapping is XCD-aware (xcd_remap): each XCD's L2 serves a contiguous band
//     of tiles, whose lists overlap heavily.
// Roofline: algorithmic bytes fwd 40*I + 20*P, bwd 76*I + 20*P (+8*I with absgrad); the kernels
// are VALU/exp-bound under that model — bench.py reports both fractions.
#include "gspl_device.h"
#include "gspl_host.h"

namespace gspl {

static constexpr int TILE = 16;
static constexpr int CHUNK = 256;

// sigma = 0.5 (a dx^2 + c dy^2) + b dx dy, written with explicit fma so that forward and backward
// evaluate bit-identical values (the skip / stop decisions of the two passes must agree).
__device__ __forceinline__ float eval_sigma(float half_a, float b, float half_c, float dx, float dy) {
    return fmaf(half_a * dx, dx, fmaf(half_c * dy, dy, (b * dx) * dy));
}

// Half-widths (hx, hy) of the axis-aligned box around the region where this splat can reach
// alpha >= 1/255:  o * exp(-sigma) >= 1/255  <=>  sigma <= tau = ln(255 o), and the ellipse
// { d : 1/2 d^T Q d <= tau } (Q = conic) has the bounding box |dx| <= sqrt(2 tau Q^-1_xx), Q^-1_xx = c / det.
// Inflated by a small margin so that fp32 rounding of exp/log can never cull a pair the exact test would
// keep; candidates still go through the exact per-pixel test.  (-1,-1): can never contribute.
__device__ __forceinline__ float2 splat_extent(float a, float b, float c, float opacity) {
    const float tau = __logf(255.f * opacity) * 1.0002f + 2e-4f;
    if (!(tau > 0.f)) return make_float2(-1.f, -1.f);
    const float det = a * c - b * b;
    if (!(det > 0.f) || !(a > 0.f) || !(c > 0.f)) return make_float2(INFINITY, INFINITY);   // not an ellipse: never cull
    const float k = 2.f * tau / det;
    return make_float2(sqrtf(k * c) * 1.0002f + 1e-3f, sqrtf(k * a) * 1.0002f + 1e-3f);
}

__device__ __forceinline__ void tile_range(int tile, int n_tiles, int64_t n_isects,
                                           const int32_t* __restrict__ offsets, int& start, int& end) {
    start = offsets[tile];
    end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
}

template <int D, int MODE, bool CHW>
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
    float* __restrict__ out_colors, float* __restrict__ out_alphas, float* __restrict__ final_Ts,
    int32_t* __restrict__ last_ids) {
    using TR = ModeTraits<MODE>;
    __shared__ float2 s_xy[CHUNK];
    __shared__ float4 s_co[CHUNK];      // 0.5a, b, 0.5c, opacity
    __shared__ float2 s_ext[CHUNK];     // conservative half-extent of the alpha >= 1/255 region
    __shared__ float s_col[CHUNK * D];
    __shared__ int s_wdone[4];

    const int tile = xcd_remap(blockIdx.x, n_tiles);
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int px = (tile % tile_w) * TILE + (w & 1) * 8 + (l & 7);
    const int py = (tile / tile_w) * TILE + (w >> 1) * 8 + (l >> 3);
    const bool inside = (px < width) && (py < height);
    const float pxf = (float)px + TR::kPixelCentre, pyf = (float)py + TR::kPixelCentre;
    // pixel-centre bounds of this wave's 8x8 quadrant (wave-uniform)
    const float qx0 = (float)((tile % tile_w) * TILE + (w & 1) * 8) + TR::kPixelCentre, qx1 = qx0 + 7.f;
    const float qy0 = (float)((tile / tile_w) * TILE + (w >> 1) * 8) + TR::kPixelCentre, qy1 = qy0 + 7.f;

    int start, end;
    tile_range(tile, n_tiles, n_isects, offsets, start, end);

    float T = 1.f;
    float acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    int last = start;            // one past the last contributing index
    bool done = !inside;
    bool wave_done = __all(done);

    for (int base = start; base < end; base += CHUNK) {
        if (l == 0) s_wdone[w] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_wdone[0] & s_wdone[1] & s_wdone[2] & s_wdone[3]) break;
        const int i = base + t;
        if (i < end) {
            const int g = flatten_ids[i];
            const float ca = conics[g * 3 + 0], cb = conics[g * 3 + 1], cc = conics[g * 3 + 2], op = opacities[g];
            s_xy[t] = make_float2(means2d[g * 2 + 0], means2d[g * 2 + 1]);
            s_co[t] = make_float4(0.5f * ca, cb, 0.5f * cc, op);
            s_ext[t] = splat_extent(ca, cb, cc, op);
#pragma unroll
            for (int c = 0; c < D; ++c) s_col[t * D + c] = colors[(int64_t)g * D + c];
        }
        __syncthreads();
        if (!wave_done) {
            const int cnt = min(CHUNK, end - base);
            // wave-level culling: each LANE tests one splat of the chunk against the quadrant's box, the
            // ballot is the candidate list; only candidates run the per-pixel loop (4 x 64 splats per chunk)
#pragma unroll 1
            for (int k = 0; k < CHUNK / 64 && !wave_done; ++k) {
                const int idx = k * 64 + l;
                const float2 cxy = s_xy[idx];
                const float2 ext = s_ext[idx];
                const bool cand = (idx < cnt) && (cxy.x + ext.x >= qx0) && (cxy.x - ext.x <= qx1) &&
                                  (cxy.y + ext.y >= qy0) && (cxy.y - ext.y <= qy1);
                unsigned long long mask = __ballot(cand);
                while (mask) {
                    const int j = k * 64 + (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const float2 xy = s_xy[j];
                    const float4 co = s_co[j];
                    const float dx = xy.x - pxf, dy = xy.y - pyf;
                    const float sigma = eval_sigma(co.x, co.y, co.z, dx, dy);
                    const float alpha = fminf(TR::kAlphaMax, co.w * __expf(-sigma));
                    bool valid = !done && (sigma >= 0.f) && (alpha >= kAlphaMin);
                    if (!__any(valid)) continue;
                    const float next_T = T * (1.f - alpha);
                    const bool stop = valid && (TR::kStopInclusive ? (next_T <= kTStop) : (next_T < kTStop));
                    done = done || stop;
                    valid = valid && !stop;
                    if (valid) {
                        const float wgt = alpha * T;
#pragma unroll
                        for (int c = 0; c < D; ++c) acc[c] += s_col[j * D + c] * wgt;
                        T = next_T;
                        last = base + j + 1;
                    }
                    if (__all(done)) { wave_done = true; break; }
                }
            }
            wave_done = __all(done);
        }
    }

    if (inside) {
        const int64_t pix = (int64_t)py * width + px;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const float bgc = backgrounds ? backgrounds[c] : 0.f;
            const float v = acc[c] + T * bgc;
            if (CHW) out_colors[(int64_t)c * width * height + pix] = v;
            else out_colors[pix * D + c] = v;
        }
        out_alphas[pix] = 1.f - T;
        final_Ts[pix] = T;     // kept exactly: 1 - (1 - T) would lose the small transmittances backward divides by
        last_ids[pix] = last;
    }
}

static_assert(true, "");
// number of per-splat gradient values reduced per (wave, splat): xy(2) conic(3) opacity(1) colour(D) [+abs xy(2)]
template <int D, bool ABS> struct BwdVals { static constexpr int N = 6 + D + (ABS ? 2 : 0); };

template <int D, int MODE, bool CHW, bool ABS>
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
    const float* __restrict__ final_Ts, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_out_colors, const float* __restrict__ v_out_alphas,
    float* __restrict__ v_means2d, float* __restrict__ v_means2d_abs,
    float* __restrict__ v_conics, float* __restrict__ v_colors, float* __restrict__ v_opacities) {
    using TR = ModeTraits<MODE>;
    constexpr int NV = BwdVals<D, ABS>::N;
    __shared__ int s_id[CHUNK];
    __shared__ float2 s_xy[CHUNK];
    __shared__ float4 s_co[CHUNK];       // a, b, c, opacity (unscaled: needed for the gradients)
    __shared__ float2 s_ext[CHUNK];
    __shared__ float s_col[CHUNK * D];
    __shared__ float s_acc[CHUNK * NV];
    __shared__ int s_last;

    const int tile = xcd_remap(blockIdx.x, n_tiles);
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int px = (tile % tile_w) * TILE + (w & 1) * 8 + (l & 7);
    const int py = (tile / tile_w) * TILE + (w >> 1) * 8 + (l >> 3);
    const bool inside = (px < width) && (py < height);
    const float pxf = (float)px + TR::kPixelCentre, pyf = (float)py + TR::kPixelCentre;
    const float qx0 = (float)((tile % tile_w) * TILE + (w & 1) * 8) + TR::kPixelCentre, qx1 = qx0 + 7.f;
    const float qy0 = (float)((tile / tile_w) * TILE + (w >> 1) * 8) + TR::kPixelCentre, qy1 = qy0 + 7.f;
    const int64_t pix = (int64_t)py * width + px;
    const int row_pos = l & 15;          // position inside the 16-lane DPP row

    int start, end;
    tile_range(tile, n_tiles, n_isects, offsets, start, end);

    const int last = inside ? last_ids[pix] : start;
    const float T_final = inside ? final_Ts[pix] : 1.f;
    float T = T_final;
    float v_out[D];
    float buffer[D];
    float bgdot = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        buffer[c] = 0.f;
        v_out[c] = 0.f;
        if (inside) v_out[c] = CHW ? v_out_colors[(int64_t)c * width * height + pix] : v_out_colors[pix * D + c];
        if (backgrounds) bgdot += backgrounds[c] * v_out[c];
    }
    const float v_out_a = (inside && v_out_alphas) ? v_out_alphas[pix] : 0.f;
    // d(out)/d(alpha_i) carries  T_final/(1-alpha_i) * (v_out_alpha - bg . v_out)
    const float tail = T_final * (v_out_a - bgdot);

    if (t == 0) s_last = start;
    for (int k = t; k < CHUNK * NV; k += 256) s_acc[k] = 0.f;
    __syncthreads();
    // wave-max of `last`, then one LDS atomic per wave
    int wl = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wl = max(wl, __shfl_xor(wl, off));
    if (l == 0) atomicMax(&s_last, wl);
    __syncthreads();
    const int block_last = s_last;
    const int wave_last = wl;

    for (int hi = block_last; hi > start; hi -= CHUNK) {
        const int lo = max(start, hi - CHUNK);
        const int cnt = hi - lo;
        // stage [lo, hi) in reverse: slot j holds index hi-1-j
        if (t < cnt) {
            const int g = flatten_ids[hi - 1 - t];
            s_id[t] = g;
            const float4 co = make_float4(conics[g * 3 + 0], conics[g * 3 + 1], conics[g * 3 + 2], opacities[g]);
            s_xy[t] = make_float2(means2d[g * 2 + 0], means2d[g * 2 + 1]);
            s_co[t] = co;
            s_ext[t] = splat_extent(co.x, co.y, co.z, co.w);
#pragma unroll
            for (int c = 0; c < D; ++c) s_col[t * D + c] = colors[(int64_t)g * D + c];
        }
        __syncthreads();
        if (wave_last > lo) {
#pragma unroll 1
            for (int kk = 0; kk < CHUNK / 64; ++kk) {
              const int slot = kk * 64 + l;
              const float2 cxy = s_xy[slot];
              const float2 ext = s_ext[slot];
              // candidate: staged, reached by some pixel of this quadrant, and its alpha >= 1/255 box touches the quadrant
              const bool cand = (slot < cnt) && (hi - 1 - slot < wave_last) && (cxy.x + ext.x >= qx0) && (cxy.x - ext.x <= qx1) &&
                                (cxy.y + ext.y >= qy0) && (cxy.y - ext.y <= qy1);
              unsigned long long mask = __ballot(cand);
              while (mask) {
                const int j = kk * 64 + (int)__builtin_ctzll(mask);
                mask &= mask - 1;
                const int idx = hi - 1 - j;
                const float2 xy = s_xy[j];
                const float4 co = s_co[j];
                const float dx = xy.x - pxf, dy = xy.y - pyf;
                const float sigma = eval_sigma(0.5f * co.x, co.y, 0.5f * co.z, dx, dy);
                const float vis = __expf(-sigma);
                const float alpha = fminf(TR::kAlphaMax, co.w * vis);
                const bool valid = (idx < last) && (sigma >= 0.f) && (alpha >= kAlphaMin);
                if (!__any(valid)) continue;

                float vals[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) vals[k] = 0.f;
                if (valid) {
                    const float ra = 1.f / (1.f - alpha);
                    T *= ra;                               // transmittance in front of this splat
                    const float fac = alpha * T;
                    float v_alpha = tail * ra;
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        const float col = s_col[j * D + c];
                        vals[6 + c] = fac * v_out[c];
                        v_alpha += (col * T - buffer[c] * ra) * v_out[c];
                        buffer[c] += col * fac;
                    }
                    if (!TR::kClampKillsGrad || (co.w * vis <= TR::kAlphaMax)) {
                        const float v_sigma = -co.w * vis * v_alpha;
                        const float gx = v_sigma * (co.x * dx + co.y * dy);
                        const float gy = v_sigma * (co.y * dx + co.z * dy);
                        vals[0] = gx;
                        vals[1] = gy;
                        vals[2] = 0.5f * v_sigma * dx * dx;
                        vals[3] = v_sigma * dx * dy;
                        vals[4] = 0.5f * v_sigma * dy * dy;
                        vals[5] = vis * v_alpha;
                        if constexpr (ABS) {
                            vals[6 + D] = fabsf(gx);
                            vals[7 + D] = fabsf(gy);
                        }
                    }
                }
                // reduce each value over the 16 lanes of its DPP row (4 fused v_add_f32_dpp), then let lane p of
                // every row add value p into the tile accumulator: ONE ds_add_f32 with 4*NV active lanes
                // (per-lane addresses, so the compiler's uniform-address atomic expansion does not kick in)
#pragma unroll
                for (int k = 0; k < NV; ++k) vals[k] = row_sum(vals[k]);
                float mine = vals[0];
#pragma unroll
                for (int k = 1; k < NV; ++k) mine = (row_pos == k) ? vals[k] : mine;
                if (row_pos < NV) atomicAdd(&s_acc[j * NV + row_pos], mine);
              }
            }
        }
        __syncthreads();
        // flush: one lane per splat of the chunk, one L2 atomic per non-zero value
        if (t < cnt) {
            const int g = s_id[t];
            float v[NV];
            bool any_nz = false;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                v[k] = s_acc[t * NV + k];
                s_acc[t * NV + k] = 0.f;
                any_nz = any_nz || (v[k] != 0.f);
            }
            if (any_nz) {
                atomicAdd(&v_means2d[g * 2 + 0], v[0]);
                atomicAdd(&v_means2d[g * 2 + 1], v[1]);
                atomicAdd(&v_conics[g * 3 + 0], v[2]);
                atomicAdd(&v_conics[g * 3 + 1], v[3]);
                atomicAdd(&v_conics[g * 3 + 2], v[4]);
                atomicAdd(&v_opacities[g], v[5]);
#pragma unroll
                for (int c = 0; c < D; ++c) atomicAdd(&v_colors[(int64_t)g * D + c], v[6 + c]);
                if constexpr (ABS) {
                    atomicAdd(&v_means2d_abs[g * 2 + 0], v[6 + D]);
                    atomicAdd(&v_means2d_abs[g * 2 + 1], v[7 + D]);
                }
            }
        }
        __syncthreads();
    }
}

template <int D, int MODE, bool CHW>
static int launch_fwd(int n_tiles, int tile_w, int width, int height, int64_t n_isects,
                      const float* means2d, const float* conics, const float* colors, const float* opacities,
                      const float* backgrounds, const int32_t* offsets, const int32_t* flatten_ids,
                      float* out_colors, float* out_alphas, float* final_Ts, int32_t* last_ids, hipStream_t s) {
    hipLaunchKernelGGL((composite_fwd_kernel<D, MODE, CHW>), dim3(n_tiles), dim3(256), 0, s,
                       n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                       offsets, flatten_ids, out_colors, out_alphas, final_Ts, last_ids);
    return check_launch("composite_fwd");
}

template <int D, int MODE, bool CHW>
static int launch_bwd(bool absgrad, int n_tiles, int tile_w, int width, int height, int64_t n_isects,
                      const float* means2d, const float* conics, const float* colors, const float* opacities,
                      const float* backgrounds, const int32_t* offsets, const int32_t* flatten_ids,
                      const float* final_Ts, const int32_t* last_ids,
                      const float* v_out_colors, const float* v_out_alphas,
                      float* v_means2d, float* v_means2d_abs, float* v_conics, float* v_colors, float* v_opacities,
                      hipStream_t s) {
    if (absgrad)
        hipLaunchKernelGGL((composite_bwd_kernel<D, MODE, CHW, true>), dim3(n_tiles), dim3(256), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities);
    else
        hipLaunchKernelGGL((composite_bwd_kernel<D, MODE, CHW, false>), dim3(n_tiles), dim3(256), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities);
    return check_launch("composite_bwd");
}

static int check_common(int N, int64_t n_isects, int D, int mode, int layout, int width, int height,
                        int tile_size, int tile_w, int tile_h, const char* who) {
    if (N < 0 || n_isects < 0 || width <= 0 || height <= 0) return fail_arg(who);
    if (tile_size != TILE) { set_error(who, "only tile_size 16 is built"); return GSPL_ERR_UNSUPPORTED; }
    if (tile_w != (width + TILE - 1) / TILE || tile_h != (height + TILE - 1) / TILE) return fail_arg(who);
    if (mode != GSPL_MODE_GSPLAT && mode != GSPL_MODE_INRIA) return fail_arg(who);
    if (layout != GSPL_LAYOUT_HWC && layout != GSPL_LAYOUT_CHW) return fail_arg(who);
    if (!(D == 1 || D == 2 || D == 3 || D == 4 || D == 8)) { set_error(who, "D must be 1,2,3,4 or 8"); return GSPL_ERR_UNSUPPORTED; }
    if (n_isects > 0x7fffffffll) return fail_arg(who);
    return GSPL_OK;
}

}  // namespace gspl

#define GSPL_DISPATCH_D(D_, MODE_, CHW_, CALL)                    \
    switch (D_) {                                                 \
        case 1: { constexpr int kD = 1; CALL(kD, MODE_, CHW_); } break; \
        case 2: { constexpr int kD = 2; CALL(kD, MODE_, CHW_); } break; \
        case 3: { constexpr int kD = 3; CALL(kD, MODE_, CHW_); } break; \
        case 4: { constexpr int kD = 4; CALL(kD, MODE_, CHW_); } break; \
        case 8: { constexpr int kD = 8; CALL(kD, MODE_, CHW_); } break; \
    }

extern "C" int gspl_composite_fwd(int N, int64_t n_isects, int D, int mode, int layout,
                                  const float* means2d, const float* conics, const float* colors,
                                  const float* opacities, const float* backgrounds,
                                  int width, int height, int tile_size, int tile_w, int tile_h,
                                  const int32_t* offsets, const int32_t* flatten_ids,
                                  float* out_colors, float* out_alphas, float* final_Ts, int32_t* last_ids, void* stream) {
    using namespace gspl;
    int rc = check_common(N, n_isects, D, mode, layout, width, height, tile_size, tile_w, tile_h, "composite_fwd: bad argument");
    if (rc != GSPL_OK) return rc;
    if (!offsets || !out_colors || !out_alphas || !final_Ts || !last_ids) return fail_arg("composite_fwd: NULL required pointer");
    if (n_isects > 0 && (!means2d || !conics || !colors || !opacities || !flatten_ids)) return fail_arg("composite_fwd: NULL required pointer");
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = (hipStream_t)stream;
    rc = GSPL_ERR_UNSUPPORTED;
#define CALL_FWD(kD, M, C) rc = launch_fwd<kD, M, C>(n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, out_colors, out_alphas, final_Ts, last_ids, s)
    if (mode == GSPL_MODE_GSPLAT) {
        if (layout == GSPL_LAYOUT_HWC) { GSPL_DISPATCH_D(D, GSPL_MODE_GSPLAT, false, CALL_FWD) }
        else { GSPL_DISPATCH_D(D, GSPL_MODE_GSPLAT, true, CALL_FWD) }
    } else {
        if (layout == GSPL_LAYOUT_HWC) { GSPL_DISPATCH_D(D, GSPL_MODE_INRIA, false, CALL_FWD) }
        else { GSPL_DISPATCH_D(D, GSPL_MODE_INRIA, true, CALL_FWD) }
    }
#undef CALL_FWD
    return rc;
}

extern "C" int gspl_composite_bwd(int N, int64_t n_isects, int D, int mode, int layout,
                                  const float* means2d, const float* conics, const float* colors,
                                  const float* opacities, const float* backgrounds,
                                  int width, int height, int tile_size, int tile_w, int tile_h,
                                  const int32_t* offsets, const int32_t* flatten_ids,
                                  const float* final_Ts, const int32_t* last_ids,
                                  const float* v_out_colors, const float* v_out_alphas,
                                  float* v_means2d, float* v_means2d_abs,
                                  float* v_conics, float* v_colors, float* v_opacities, void* stream) {
    using namespace gspl;
    int rc = check_common(N, n_isects, D, mode, layout, width, height, tile_size, tile_w, tile_h, "composite_bwd: bad argument");
    if (rc != GSPL_OK) return rc;
    if (n_isects == 0 || N == 0) return GSPL_OK;
    if (!means2d || !conics || !colors || !opacities || !offsets || !flatten_ids || !final_Ts || !last_ids ||
        !v_out_colors || !v_means2d || !v_conics || !v_colors || !v_opacities)
        return fail_arg("composite_bwd: NULL required pointer");
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = (hipStream_t)stream;
    const bool absgrad = v_means2d_abs != nullptr;
    rc = GSPL_ERR_UNSUPPORTED;
#define CALL_BWD(kD, M, C) rc = launch_bwd<kD, M, C>(absgrad, n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, s)
    if (mode == GSPL_MODE_GSPLAT) {
        if (layout == GSPL_LAYOUT_HWC) { GSPL_DISPATCH_D(D, GSPL_MODE_GSPLAT, false, CALL_BWD) }
        else { GSPL_DISPATCH_D(D, GSPL_MODE_GSPLAT, true, CALL_BWD) }
    } else {
        if (layout == GSPL_LAYOUT_HWC) { GSPL_DISPATCH_D(D, GSPL_MODE_INRIA, false, CALL_BWD) }
        else { GSPL_DISPATCH_D(D, GSPL_MODE_INRIA, true, CALL_BWD) }
    }
#undef CALL_BWD
    return rc;
}
