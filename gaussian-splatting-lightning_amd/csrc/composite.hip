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
//   * workgroup -> tile mapping is XCD-aware (xcd_remap): runs of consecutive tiles (whose lists overlap heavily)
//     share an XCD's L2, and the runs are dealt round-robin so that every XCD sees the same mix of dense and sparse rows.
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
    // hardware rcp / sqrt (1 ulp) instead of the IEEE expansions (~10 instructions each): the margins absorb it
    const float k = 2.f * tau * __builtin_amdgcn_rcpf(det);
    return make_float2(__builtin_amdgcn_sqrtf(k * c) * 1.0004f + 1e-3f, __builtin_amdgcn_sqrtf(k * a) * 1.0004f + 1e-3f);
}

// Two splats at once with packed fp32 math; each component is bit-identical to eval_sigma.
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_t eval_sigma2(v2f_t half_a, v2f_t b, v2f_t half_c, v2f_t dx, v2f_t dy) {
    return __builtin_elementwise_fma(half_a * dx, dx, __builtin_elementwise_fma(half_c * dy, dy, (b * dx) * dy));
}

// Exact test "can this splat reach alpha >= 1/255 at some pixel centre of the box [x0,x1] x [y0,y1]" (continuous box,
// conservative margins): the x-span of (ellipse 1/2 d^T Q d <= tau) intersected with the band dy in [y0-my, y1-my] is
// [left, right] with right = hx if the ellipse's rightmost point lies in the band, else the larger chord end at the band
// edges (see binning.hip, row_span); the box is reachable iff that span meets [x0, x1].  About 21 % of the candidates
// that pass the bounding-box test fail this one (measured), and with the predicated inner loop of the forward kernel
// every candidate costs the same whether or not a pixel is touched.
__device__ __forceinline__ bool box_reachable(float mx, float my, float a, float b, float c, float opacity,
                                              float x0, float x1, float y0, float y1) {
    const float tau = __logf(255.f * opacity) * 1.0002f + 2e-4f;
    if (!(tau > 0.f)) return false;
    const float det = a * c - b * b;
    if (!(det > 0.f) || !(a > 0.f) || !(c > 0.f)) return true;     // not an ellipse: never cull
    const float two_tau = 2.f * tau;
    const float rdet = __builtin_amdgcn_rcpf(det);
    const float hy = __builtin_amdgcn_sqrtf(two_tau * a * rdet) * 1.0004f + 1e-3f;
    float lo = y0 - my, hi = y1 - my;
    if (hi < -hy || lo > hy) return false;
    lo = fmaxf(lo, -hy); hi = fminf(hi, hy);
    const float tta = two_tau * a;
    const float rlo = __builtin_amdgcn_sqrtf(fmaxf(0.f, tta - det * lo * lo));
    const float rhi = __builtin_amdgcn_sqrtf(fmaxf(0.f, tta - det * hi * hi));
    const float hx = __builtin_amdgcn_sqrtf(two_tau * c * rdet);
    const float dys = b * __builtin_amdgcn_sqrtf(two_tau * rdet * __builtin_amdgcn_rcpf(c));
    const float inv_a = __builtin_amdgcn_rcpf(a);
    const float right = (-dys >= lo && -dys <= hi) ? hx : fmaxf((-b * lo + rlo) * inv_a, (-b * hi + rhi) * inv_a);
    const float left = (dys >= lo && dys <= hi) ? -hx : fminf((-b * lo - rlo) * inv_a, (-b * hi - rhi) * inv_a);
    const float eps = 2e-3f + 5e-4f * hx;
    return (mx + right + eps >= x0) && (mx + left - eps <= x1);
}

__device__ __forceinline__ void tile_range(int tile, int n_tiles, int64_t n_isects,
                                           const int32_t* __restrict__ offsets, int& start, int& end) {
    start = offsets[tile];
    // n_isects < 0: `offsets` has n_tiles + 1 entries, the last one is the list length (device-side count, gspl_bin_sort_device_count)
    end = (tile + 1 < n_tiles || n_isects < 0) ? offsets[tile + 1] : (int)n_isects;
}

// Forward: ONE WAVE PER WORKGROUP.  A workgroup is a single wave64 that owns one 8x8 quadrant of a tile and walks
// the tile's list on its own: no workgroup barrier anywhere, a quadrant that saturates (T <= 1e-4 everywhere) or has
// few candidates retires immediately and frees its slot, and up to 32 such waves per CU hide each other's LDS and
// gather latency.  The four quadrants of a tile each gather the tile's records (L1/L2 hits: the four workgroups are
// adjacent in dispatch order and XCD-remapped together).
//
// Per round of 64 splats:
//   1. lane l gathers splat base+l and tests ITS splat's alpha >= 1/255 box against the quadrant;
//   2. ballot + prefix count (v_mbcnt) COMPACT the candidates into a structure-of-arrays list in LDS
//      (x[], y[], ha[], b[], hc[], opacity[], colour[][D]);
//   3. the wave walks the list TWO candidates at a time: one ds_read_b64 per array yields the pair as a
//      64-bit register pair, so sigma / exp argument / alpha are evaluated with packed fp32 math
//      (v_pk_add/mul/fma_f32: half the instructions — and issue slots, scalar bookkeeping — for the same arithmetic),
//      and the short sequential transmittance update is branch-free (predicated) to keep scalar-unit work low:
//      the first version of this loop was bound by SALU mask bookkeeping (136 M scalar vs 118 M vector instructions).
typedef float v2f __attribute__((ext_vector_type(2)));
static constexpr int FCHUNK = 64;
static constexpr int FLIST = FCHUNK + 2;      // room for the odd-count padding entry

#ifdef GSPL_COUNT_PAIRS
// instrumentation build only (tools/micro/pair_stats.py).  Backward: [0] half-tile candidates, [1] valid pixel pairs, [2] candidates
// with any valid pixel, [3] ... touching both quadrants.  Forward: [4] staging rounds (64 list entries each, per quadrant wave),
// [5] candidates that passed the box test, [6] two-candidate iterations executed.
__device__ unsigned long long g_pair_stats[8];
#endif

// HITS: also report, per splat, whether any pixel composited it (hit_flags[g] = 1; the fork's `has_hit_any_pixels`, set by its
// rasterizer forward: gsplat_v1_renderer.py:287 reads it as `acc_vis`).  Per candidate one wave-wide "any lane contributed" bit
// is collected in a scalar mask; after the round the lane that gathered the candidate stores its flag.
template <int D, int MODE, bool CHW, bool HITS>
__global__ __launch_bounds__(64) void composite_fwd_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
    float* __restrict__ out_colors, float* __restrict__ out_alphas, float* __restrict__ final_Ts,
    int32_t* __restrict__ last_ids, uint8_t* __restrict__ hit_flags) {
    using TR = ModeTraits<MODE>;
    __shared__ __attribute__((aligned(16))) float s_x[FLIST];
    __shared__ __attribute__((aligned(16))) float s_y[FLIST];
    __shared__ __attribute__((aligned(16))) float s_ha[FLIST];
    __shared__ __attribute__((aligned(16))) float s_b[FLIST];
    __shared__ __attribute__((aligned(16))) float s_hc[FLIST];
    __shared__ __attribute__((aligned(16))) float s_op[FLIST];
    __shared__ __attribute__((aligned(16))) int s_pos[FLIST];      // list index (base + lane) of each candidate
    __shared__ __attribute__((aligned(16))) float s_col[FLIST * D];

    const int unit = xcd_remap(blockIdx.x, 4 * n_tiles, 4 * GSPL_XCD_RUN);     // (tile, quadrant): the four quadrants of a tile stay on one XCD
    const int tile = unit >> 2, w = unit & 3, l = threadIdx.x;
    const int px = (tile % tile_w) * TILE + (w & 1) * 8 + (l & 7);
    const int py = (tile / tile_w) * TILE + (w >> 1) * 8 + (l >> 3);
    const bool inside = (px < width) && (py < height);
    const float pxf = (float)px + TR::kPixelCentre, pyf = (float)py + TR::kPixelCentre;
    const v2f pxf2 = {pxf, pxf}, pyf2 = {pyf, pyf};
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

    if (!__all(done)) {
        int g_next = (start + l < end) ? flatten_ids[start + l] : 0;
        for (int base = start; base < end; base += FCHUNK) {
            const int i = base + l;
            const int g = g_next;
            if (base + FCHUNK + l < end) g_next = flatten_ids[base + FCHUNK + l];     // prefetch next round's id
            bool cand = false;
            float2 xy = make_float2(0.f, 0.f);
            float ca = 0.f, cb = 0.f, cc = 0.f, op = 0.f;
            if (i < end) {
                ca = conics[g * 3 + 0]; cb = conics[g * 3 + 1]; cc = conics[g * 3 + 2]; op = opacities[g];
                xy = make_float2(means2d[g * 2 + 0], means2d[g * 2 + 1]);
                cand = box_reachable(xy.x, xy.y, ca, cb, cc, op, qx0, qx1, qy0, qy1);
            }
            const unsigned long long mask = __ballot(cand);
            const int ncand = __builtin_popcountll(mask);
#ifdef GSPL_COUNT_PAIRS
            if (l == 0) { atomicAdd(&g_pair_stats[4], 1ull); atomicAdd(&g_pair_stats[5], (unsigned long long)ncand); }
#endif
            if (ncand == 0) continue;
            // compaction: candidate k of the round goes to slot k (k = number of candidates in lower lanes)
            const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (cand) {
                s_x[slot] = xy.x; s_y[slot] = xy.y;
                s_ha[slot] = 0.5f * ca; s_b[slot] = cb; s_hc[slot] = 0.5f * cc; s_op[slot] = op;
                s_pos[slot] = i + 1;
#pragma unroll
                for (int c = 0; c < D; ++c) s_col[slot * D + c] = colors[(int64_t)g * D + c];
            }
            if (l == 0) {        // padding entry for an odd count: opacity 0 -> alpha 0 -> never valid
                s_x[ncand] = 0.f; s_y[ncand] = 0.f; s_ha[ncand] = 0.f; s_b[ncand] = 0.f; s_hc[ncand] = 0.f; s_op[ncand] = 0.f;
                s_pos[ncand] = 0;
#pragma unroll
                for (int c = 0; c < D; ++c) s_col[ncand * D + c] = 0.f;
            }
            bool all_done = false;
            unsigned long long hitmask = 0ull;      // HITS: bit k = some pixel composited candidate k of this round
            for (int k = 0; k < ncand; k += 2) {
                const v2f x2 = *reinterpret_cast<const v2f*>(&s_x[k]);
                const v2f y2 = *reinterpret_cast<const v2f*>(&s_y[k]);
                const v2f ha2 = *reinterpret_cast<const v2f*>(&s_ha[k]);
                const v2f b2 = *reinterpret_cast<const v2f*>(&s_b[k]);
                const v2f hc2 = *reinterpret_cast<const v2f*>(&s_hc[k]);
                const v2f op2 = *reinterpret_cast<const v2f*>(&s_op[k]);
                const int2 pos2 = *reinterpret_cast<const int2*>(&s_pos[k]);
                const v2f dx2 = x2 - pxf2, dy2 = y2 - pyf2;
                const v2f sigma2 = eval_sigma2(ha2, b2, hc2, dx2, dy2);
                const v2f arg2 = sigma2 * (v2f){-1.4426950408889634f, -1.4426950408889634f};
                const v2f e2 = {__builtin_amdgcn_exp2f(arg2.x), __builtin_amdgcn_exp2f(arg2.y)};
                const v2f raw2 = op2 * e2;
                const float alpha[2] = {fminf(TR::kAlphaMax, raw2.x), fminf(TR::kAlphaMax, raw2.y)};
                const float sig[2] = {sigma2.x, sigma2.y};
                const int pos[2] = {pos2.x, pos2.y};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const bool valid = !done && (sig[e] >= 0.f) && (alpha[e] >= kAlphaMin);
                    const float next_T = T * (1.f - alpha[e]);
                    const bool stop = valid && (TR::kStopInclusive ? (next_T <= kTStop) : (next_T < kTStop));
                    const bool contrib = valid && !stop;
                    const float wgt = contrib ? alpha[e] * T : 0.f;
#pragma unroll
                    for (int c = 0; c < D; ++c) acc[c] = fmaf(s_col[(k + e) * D + c], wgt, acc[c]);
                    T = contrib ? next_T : T;
                    last = contrib ? pos[e] : last;
                    done = done || stop;
                    if constexpr (HITS) hitmask |= (__ballot(contrib) != 0ull ? 1ull : 0ull) << (k + e);
                }
#ifdef GSPL_COUNT_PAIRS
                if (l == 0) atomicAdd(&g_pair_stats[6], 1ull);
#endif
                if (__all(done)) { all_done = true; break; }
            }
            if constexpr (HITS) {
                if (cand && ((hitmask >> slot) & 1ull)) hit_flags[g] = 1;
            }
            if (all_done) break;
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

// number of per-splat gradient values accumulated per (tile, splat): xy(2) conic(3) opacity(1) colour(D) [+abs xy(2)]
template <int D, bool ABS> struct BwdVals { static constexpr int N = 6 + D + (ABS ? 2 : 0); };

static constexpr int BCHUNK = 128;     // splats staged per round in the backward kernel (LDS footprint -> occupancy)
#ifndef GSPL_BWD_BATCH
#define GSPL_BWD_BATCH 8
#endif
static constexpr int BATCH = GSPL_BWD_BATCH;   // splats per phase-2 batch
static constexpr int GROUP = 64 / BATCH;       // phase-2 lanes that share one splat (8 or 4)
static constexpr int COLS = 8 / GROUP;         // pixel columns each of those lanes walks (1 or 2)
static_assert(BATCH == 8 || BATCH == 16, "phase-2 batch must be 8 or 16");

// Staged record in LDS (floats): x y a/2 b | c/2 opacity quadrant-mask - | colour[D] (padded to a multiple of 4):
// one address register per candidate, the fields are fetched with immediate offsets (b128 + b64 [+ colour]).
template <int D> struct BwdRec { static constexpr int STRIDE = 8 + ((D + 3) & ~3); };

// Sum over each aligned group of GROUP lanes (fused v_add_f32_dpp); every lane of the group gets the sum.
__device__ __forceinline__ float group_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    if (GROUP == 8) v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
    return v;
}

// box_reachable for the four 8x8 quadrants of one tile at once (bit w = quadrant w = (x half) | (y half) << 1); the
// per-splat terms (tau, extents, the tangent offset) are shared, the chord ends are per y band.  (tx0, ty0) is the
// centre of the tile's first pixel.
__device__ __forceinline__ unsigned quadrant_mask(float mx, float my, float a, float b, float c, float opacity, float tx0, float ty0) {
    const float tau = __logf(255.f * opacity) * 1.0002f + 2e-4f;
    if (!(tau > 0.f)) return 0u;
    const float det = a * c - b * b;
    if (!(det > 0.f) || !(a > 0.f) || !(c > 0.f)) return 0xFu;      // not an ellipse: never cull
    const float two_tau = 2.f * tau;
    const float rdet = __builtin_amdgcn_rcpf(det);
    const float hy = __builtin_amdgcn_sqrtf(two_tau * a * rdet) * 1.0004f + 1e-3f;
    const float tta = two_tau * a;
    const float hx = __builtin_amdgcn_sqrtf(two_tau * c * rdet);
    const float dys = b * __builtin_amdgcn_sqrtf(two_tau * rdet * __builtin_amdgcn_rcpf(c));
    const float inv_a = __builtin_amdgcn_rcpf(a);
    const float eps = 2e-3f + 5e-4f * hx;
    unsigned m = 0u;
#pragma unroll
    for (int band = 0; band < 2; ++band) {
        const float y0 = ty0 + 8.f * (float)band;
        float lo = y0 - my, hi = (y0 + 7.f) - my;
        if (hi < -hy || lo > hy) continue;
        lo = fmaxf(lo, -hy); hi = fminf(hi, hy);
        const float rlo = __builtin_amdgcn_sqrtf(fmaxf(0.f, tta - det * lo * lo));
        const float rhi = __builtin_amdgcn_sqrtf(fmaxf(0.f, tta - det * hi * hi));
        const float right = (-dys >= lo && -dys <= hi) ? hx : fmaxf((-b * lo + rlo) * inv_a, (-b * hi + rhi) * inv_a);
        const float left = (dys >= lo && dys <= hi) ? -hx : fminf((-b * lo - rlo) * inv_a, (-b * hi - rhi) * inv_a);
        const float xr = mx + right + eps, xl = mx + left - eps;
        if (xr >= tx0 && xl <= tx0 + 7.f) m |= 1u << (2 * band);
        if (xr >= tx0 + 8.f && xl <= tx0 + 15.f) m |= 2u << (2 * band);
    }
    return m;
}

// Backward, two phases per wave (no extra workgroup barriers):
//   phase 1 (lane = pixel of the wave's 8x8 quadrant): walk the candidate splats back-to-front, rebuild alpha and T,
//            and emit just TWO numbers per (pixel, splat): fac = alpha*T (colour weight) and sp = dL/dsigma.  The
//            running "colour behind" enters dL/dalpha only through its dot product with dL/dout, so ONE scalar
//            R = T_final (v_alpha_out - bg.v_out) - sum_behind fac_k (colour_k . v_out) replaces D accumulators:
//            dL/dalpha = R/(1-alpha) + T (colour . v_out).   fac and sp go to a wave-private LDS slab, stored
//            column-major ([slot][column*8 + row]) so that phase 2 reads a pixel column as two b128 loads.
//   phase 2 (lane = (splat slot, pixel column[s]), after BATCH active splats): each lane walks the 8 pixels of its
//            column(s) for ITS splat with packed fp32 math (rows in pairs) and accumulates the moments
//            sum(sp), sum(sp dy), sum(sp dy^2) and the colour sums (dx is constant down a column); a 2- or 3-step
//            DPP reduction over the lanes of the group finishes the quadrant.
//   The per-(tile, splat) totals of the four waves meet in LDS (one ds_add_f32 per lane) and leave as ONE
//   fp32 L2 atomic per value per (tile, splat).
// Staging evaluates the exact alpha >= 1/255 reachability of each quadrant once per (tile, splat) (quadrant_mask), so
// a wave only visits splats that can touch its 64 pixels.
#ifndef GSPL_BWD_WAVES
#define GSPL_BWD_WAVES 5     // <= 96 VGPRs: 5 waves/SIMD (with the ~28 KB LDS footprint: 5 blocks/CU)
#endif
template <int D, int MODE, bool CHW, bool ABS, bool PACKED>
__global__ __launch_bounds__(256, GSPL_BWD_WAVES) void composite_bwd_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
    const float* __restrict__ final_Ts, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_out_colors, const float* __restrict__ v_out_alphas,
    float* __restrict__ v_means2d, float* __restrict__ v_means2d_abs,
    float* __restrict__ v_conics, float* __restrict__ v_colors, float* __restrict__ v_opacities, int packed_stride,
    uint8_t* __restrict__ hit_flags) {
    using TR = ModeTraits<MODE>;
    constexpr int NV = BwdVals<D, ABS>::N;
    constexpr int RS = BwdRec<D>::STRIDE;
    constexpr bool VO_REGS = COLS * 8 * D <= 24;      // dL/dout of the lane's phase-2 column(s) lives in registers
    constexpr int SLAB = 2 * BATCH * 64;              // floats per wave: fac plane, sp plane
    __shared__ int s_id[BCHUNK];
    __shared__ __attribute__((aligned(16))) float s_rec[BCHUNK * RS];
    __shared__ float s_acc[BCHUNK * NV];
    __shared__ __attribute__((aligned(16))) float s_slab[4 * SLAB];
    __shared__ __attribute__((aligned(16))) float s_vo_keep[VO_REGS ? 4 : 4 * 64 * D];
    static_assert(4 * 64 * D <= 4 * SLAB, "s_vo alias too small");
    float* s_vo = VO_REGS ? s_slab : s_vo_keep;       // [wave][column][channel][row]; aliased onto the slab when only read before the main loop
    __shared__ int s_last;

    const int tile = xcd_remap(blockIdx.x, n_tiles);
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int tx = (tile % tile_w) * TILE, ty = (tile / tile_w) * TILE;
    const int px = tx + (w & 1) * 8 + (l & 7);
    const int py = ty + (w >> 1) * 8 + (l >> 3);
    const bool inside = (px < width) && (py < height);
    const float pxf = (float)px + TR::kPixelCentre, pyf = (float)py + TR::kPixelCentre;
    const float qx0 = (float)(tx + (w & 1) * 8) + TR::kPixelCentre;
    const float qy0 = (float)(ty + (w >> 1) * 8) + TR::kPixelCentre;
    const int64_t pix = (int64_t)py * width + px;
    const int tl = (l & 7) * 8 + (l >> 3);             // this pixel's slot in the column-major slab
    const int ps = l / GROUP, pg = l % GROUP;          // phase-2 role: splat slot, lane within the splat's group
    float* slab = s_slab + w * SLAB;

    int start, end;
    tile_range(tile, n_tiles, n_isects, offsets, start, end);

    const int last = inside ? last_ids[pix] : start;
    const float T_final = inside ? final_Ts[pix] : 1.f;
    float T = T_final;
    float v_out[D];
    float bgdot = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        v_out[c] = 0.f;
        if (inside) v_out[c] = CHW ? v_out_colors[(int64_t)c * width * height + pix] : v_out_colors[pix * D + c];
        if (backgrounds) bgdot += backgrounds[c] * v_out[c];
        s_vo[((w * 8 + (l & 7)) * D + c) * 8 + (l >> 3)] = v_out[c];
    }
    const float v_out_a = (inside && v_out_alphas) ? v_out_alphas[pix] : 0.f;
    // R: the part of dL/dalpha_i * (1 - alpha_i) that does not depend on splat i's own colour; starts as
    // T_final (v_out_alpha - bg . v_out) and loses fac_k (colour_k . v_out) for every splat k walked (see above)
    float R = T_final * (v_out_a - bgdot);

    if (t == 0) s_last = start;
    for (int k = t; k < BCHUNK * NV; k += 256) s_acc[k] = 0.f;
    __syncthreads();
    // phase-2 view of dL/d(out): rows in pairs, for the lane's column(s)
    v2f vo2[VO_REGS ? COLS : 1][4][D];
    if constexpr (VO_REGS) {
#pragma unroll
        for (int cc = 0; cc < COLS; ++cc)
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const float4* vp = reinterpret_cast<const float4*>(s_vo + ((w * 8 + pg * COLS + cc) * D + c) * 8);
                const float4 v0 = vp[0], v1 = vp[1];
                vo2[cc][0][c] = (v2f){v0.x, v0.y}; vo2[cc][1][c] = (v2f){v0.z, v0.w};
                vo2[cc][2][c] = (v2f){v1.x, v1.y}; vo2[cc][3][c] = (v2f){v1.z, v1.w};
            }
    }
    // wave-max of `last`, then one LDS atomic per wave
    int wl = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wl = max(wl, __shfl_xor(wl, off));
    if (l == 0) atomicMax(&s_last, wl);
    __syncthreads();
    const int block_last = s_last;
    const int wave_last = wl;

    int nb = 0;                              // splats waiting in the phase-2 batch (wave-uniform)
    int batch_j = 0;                         // lane b holds the staged slot index of batch entry b

    auto phase2 = [&](int count) {
        __builtin_amdgcn_wave_barrier();
        float vals[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) vals[k] = 0.f;
        const int j = __builtin_amdgcn_ds_bpermute(ps << 2, batch_j);
        const bool live = ps < count;
        float ca = 0.f, cb = 0.f, cc_ = 0.f, co_ = 1.f;
        if (live) {
            const float* rec = s_rec + j * RS;
            const float4 r0 = *reinterpret_cast<const float4*>(rec);          // x y a/2 b
            const float2 r1 = *reinterpret_cast<const float2*>(rec + 4);      // c/2 opacity
            ca = 2.f * r0.z; cb = r0.w; cc_ = 2.f * r1.x; co_ = r1.y;
            const float dy0 = r0.y - qy0;
            const v2f dy0v = {dy0, dy0};
            float Sx = 0.f, Sxx = 0.f, Sxy = 0.f, S0 = 0.f, Sy = 0.f, ax = 0.f, ay = 0.f;
            v2f syy2 = {0.f, 0.f};
            v2f rgb2[D];
#pragma unroll
            for (int c = 0; c < D; ++c) rgb2[c] = (v2f){0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < COLS; ++cc) {
                const int colq = pg * COLS + cc;
                const float dx = r0.x - (qx0 + (float)colq);
                const float4* Fp = reinterpret_cast<const float4*>(slab + ps * 64 + colq * 8);
                const float4* Sp = reinterpret_cast<const float4*>(slab + BATCH * 64 + ps * 64 + colq * 8);
                const float4 f0 = Fp[0], f1 = Fp[1], q0 = Sp[0], q1 = Sp[1];
                const v2f F2[4] = {{f0.x, f0.y}, {f0.z, f0.w}, {f1.x, f1.y}, {f1.z, f1.w}};
                const v2f S2[4] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}};
                v2f s02 = {0.f, 0.f}, sy2 = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const v2f dy2 = dy0v - (v2f){(float)(2 * k), (float)(2 * k + 1)};
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        v2f vo;
                        if constexpr (VO_REGS) vo = vo2[cc][k][c];
                        else vo = *reinterpret_cast<const v2f*>(s_vo + ((w * 8 + colq) * D + c) * 8 + 2 * k);
                        rgb2[c] = __builtin_elementwise_fma(F2[k], vo, rgb2[c]);
                    }
                    s02 += S2[k];
                    const v2f tq = S2[k] * dy2;
                    sy2 += tq;
                    syy2 = __builtin_elementwise_fma(tq, dy2, syy2);
                    if constexpr (ABS) {
                        ax += fabsf(S2[k].x * (ca * dx + cb * dy2.x)) + fabsf(S2[k].y * (ca * dx + cb * dy2.y));
                        ay += fabsf(S2[k].x * (cb * dx + cc_ * dy2.x)) + fabsf(S2[k].y * (cb * dx + cc_ * dy2.y));
                    }
                }
                const float s0c = s02.x + s02.y, syc = sy2.x + sy2.y;
                const float sxc = s0c * dx;
                S0 += s0c; Sy += syc;
                Sx += sxc;
                Sxx = fmaf(sxc, dx, Sxx);
                Sxy = fmaf(syc, dx, Sxy);
            }
            vals[0] = Sx;                    // -> sum sp*dx
            vals[1] = Sy;                    // -> sum sp*dy
            vals[2] = Sxx;                   // -> sum sp*dx^2
            vals[3] = Sxy;                   // -> sum sp*dx*dy
            vals[4] = syy2.x + syy2.y;       // -> sum sp*dy^2
            vals[5] = S0;                    // -> sum sp
#pragma unroll
            for (int c = 0; c < D; ++c) vals[6 + c] = rgb2[c].x + rgb2[c].y;
            if constexpr (ABS) { vals[6 + D] = ax; vals[7 + D] = ay; }
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) vals[k] = group_sum(vals[k]);
        // moments -> gradients (every lane of the group holds the group totals)
        const float Sx = vals[0], Sy = vals[1];
        vals[0] = ca * Sx + cb * Sy;                         // dL/dx
        vals[1] = cb * Sx + cc_ * Sy;                        // dL/dy
        vals[2] = 0.5f * vals[2];                            // dL/da
        vals[4] = 0.5f * vals[4];                            // dL/dc      (vals[3] = dL/db as is)
        vals[5] = (co_ != 0.f) ? -vals[5] * __builtin_amdgcn_rcpf(co_) : 0.f;     // dL/dopacity = sum(vis * v_alpha) = -sum(sp) / o
        // lane g of the group adds values g, g + GROUP, ...: per-lane addresses, ceil(NV / GROUP) ds_add_f32 per batch
#pragma unroll
        for (int r = 0; r * GROUP < NV; ++r) {
            float mine = vals[r * GROUP];
#pragma unroll
            for (int k = 1; k < GROUP && r * GROUP + k < NV; ++k) mine = (pg == k) ? vals[r * GROUP + k] : mine;
            if (live && r * GROUP + pg < NV) atomicAdd(&s_acc[j * NV + r * GROUP + pg], mine);
        }
    };

    for (int hi = block_last; hi > start; hi -= BCHUNK) {
        const int lo = max(start, hi - BCHUNK);
        const int cnt = hi - lo;
        // stage [lo, hi) in reverse: slot j holds index hi-1-j
        if (t < cnt) {
            const int g = flatten_ids[hi - 1 - t];
            s_id[t] = g;
            const float ca = conics[g * 3 + 0], cb = conics[g * 3 + 1], cc = conics[g * 3 + 2], op = opacities[g];
            const float mx = means2d[g * 2 + 0], my = means2d[g * 2 + 1];
            const unsigned qm = quadrant_mask(mx, my, ca, cb, cc, op, (float)tx + TR::kPixelCentre, (float)ty + TR::kPixelCentre);
            float* rec = s_rec + t * RS;
            *reinterpret_cast<float4*>(rec) = make_float4(mx, my, 0.5f * ca, cb);
            *reinterpret_cast<float4*>(rec + 4) = make_float4(0.5f * cc, op, __uint_as_float(qm), 0.f);
#pragma unroll
            for (int c = 0; c < D; ++c) rec[8 + c] = colors[(int64_t)g * D + c];
        }
        __syncthreads();
        if (wave_last > lo) {
#pragma unroll 1
            for (int kk = 0; kk < BCHUNK / 64; ++kk) {
                const int slot = kk * 64 + l;
                const unsigned qm = __float_as_uint(s_rec[slot * RS + 6]);
                // candidate: staged, reached by some pixel of this quadrant, and able to reach alpha >= 1/255 inside it
                const bool cand = (slot < cnt) && (hi - 1 - slot < wave_last) && ((qm >> w) & 1u);
                unsigned long long mask = __ballot(cand);
                while (mask) {
                    const int j = kk * 64 + (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int idx = hi - 1 - j;
                    const float* rec = s_rec + j * RS;
                    const float4 r0 = *reinterpret_cast<const float4*>(rec);          // x y a/2 b
                    const float2 r1 = *reinterpret_cast<const float2*>(rec + 4);      // c/2 opacity
                    const float dx = r0.x - pxf, dy = r0.y - pyf;
                    const float sigma = eval_sigma(r0.z, r0.w, r1.x, dx, dy);
                    const float vis = __expf(-sigma);
                    const float raw = r1.y * vis;
                    const float alpha = fminf(TR::kAlphaMax, raw);
                    const bool valid = (idx < last) && (sigma >= 0.f) && (alpha >= kAlphaMin);
                    if (!__any(valid)) continue;
                    if (hit_flags && l == 0) hit_flags[s_id[j]] = 1;      // some pixel takes this splat (has_hit_any_pixels)
                    float fac = 0.f, sp = 0.f;
                    if (valid) {
                        // v_rcp_f32 (1 ulp): an IEEE division here expands to ~10 VALU instructions per (pixel, splat) pair
                        const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
                        T *= ra;                               // transmittance in front of this splat
                        fac = alpha * T;
                        float cdot = rec[8] * v_out[0];
#pragma unroll
                        for (int c = 1; c < D; ++c) cdot = fmaf(rec[8 + c], v_out[c], cdot);
                        const float v_alpha = fmaf(cdot, T, R * ra);
                        R = fmaf(-cdot, fac, R);
                        if (!TR::kClampKillsGrad || (raw <= TR::kAlphaMax)) sp = -raw * v_alpha;
                    }
                    slab[nb * 64 + tl] = fac;
                    slab[BATCH * 64 + nb * 64 + tl] = sp;
                    batch_j = gspl_writelane_i32(j, nb, batch_j);     // lane nb remembers the staged slot (one v_writelane)
                    if (++nb == BATCH) { phase2(BATCH); nb = 0; }
                }
            }
            if (nb) { phase2(nb); nb = 0; }
        }
        __syncthreads();
        // flush to global memory, one fp32 L2 atomic per non-zero value per (tile, splat)
        if constexpr (PACKED) {
            // packed rows [N][NV] (x, y, a, b, c, opacity, colour[D], abs x, abs y): thread e handles element e of the
            // round's [cnt][NV] block, so the 64 lanes of one atomic instruction cover ~64/NV splat rows with contiguous
            // components (a handful of cache lines per instruction instead of 64 scattered dwords)
            float* __restrict__ v_packed = v_means2d;
            for (int e = t; e < cnt * NV; e += 256) {
                const float v = s_acc[e];
                s_acc[e] = 0.f;
                const int row = e / NV;
                if (v != 0.f) atomicAdd(&v_packed[(int64_t)s_id[row] * packed_stride + (e - row * NV)], v);
            }
        } else if (t < cnt) {
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

// ---------------------------------------------------------------------------------------------------------------
// Backward, TWO PIXELS PER LANE.  A workgroup is 2 waves per tile; wave w owns the 16x8 half tile of rows [8w, 8w+8)
// and lane l carries pixel A = (column l&7, row l>>3) and pixel B = (column 8 + (l&7), same row).  Every per-pixel
// quantity is a 2-vector {A, B}, so the whole phase-1 chain runs on packed fp32 instructions (v_pk_add/mul/fma_f32:
// the splat's wave-uniform parameters are broadcast with op_sel), each LDS record read feeds 128 pixels instead of
// 64, and the scalar loop bookkeeping per (tile, splat) halves.  dy is shared by the two pixels (same row).
// Pixels that do not take a splat (alpha < 1/255, behind their last contributor, outside the image) run with
// alpha = 0, which leaves T, R and the emitted (fac, sp) exactly neutral (1/(1-0) = 1), so no exec masking is needed.
// Phase 2: lane = (slot s of P2_SLOTS, column c of the half tile's 16), 8 rows per lane, row_sum (16-lane DPP) finish.
#ifndef GSPL_BWD2_CHUNK
#define GSPL_BWD2_CHUNK 64
#endif
#ifndef GSPL_BWD2_WAVES
#define GSPL_BWD2_WAVES 5
#endif
static constexpr int B2CHUNK = GSPL_BWD2_CHUNK;   // splats staged per round
static constexpr int P2_SLOTS = 4;                // splats per phase-2 batch (16 lanes each)
#ifdef GSPL_BWD2_SOLO
// EXPERIMENT (A/B builds): one wave per WORKGROUP = one 16x8 half tile that stages, walks and flushes the tile's list on its own:
// no partner wave to wait for at the two barriers of a round (the halves of a tile saturate at different depths and pass
// different numbers of candidates), at the price of staging and flushing every list entry twice.  Needs GSPL_BWD2_CHUNK <= 64.
static constexpr int B2_NW = 1;
#else
static constexpr int B2_NW = 2;                   // waves per workgroup: the two half tiles of a tile share the staged records
#endif
static constexpr int B2_NT = 64 * B2_NW;
static_assert(B2CHUNK <= B2_NT || B2_NW == 2, "a round is staged by one pass of the workgroup's threads");

#ifdef GSPL_BWD_LPT
// EXPERIMENT (A/B builds): workgroups take the tiles longest list first, so that the heavy tiles of the image centre do not start
// in the middle of the launch and define its end.
__device__ int g_tile_order[1 << 16];
__global__ __launch_bounds__(1024) void tile_order_kernel(const int32_t* __restrict__ offsets, int n_tiles, int64_t n_isects) {
    __shared__ int s_max;
    __shared__ int s_hist[256], s_cur[256];
    const int t = threadIdx.x;
    if (t == 0) s_max = 0;
    if (t < 256) s_hist[t] = 0;
    __syncthreads();
    int mx = 0;
    for (int k = t; k < n_tiles; k += 1024) {
        const int len = (int)((k + 1 < n_tiles ? (int64_t)offsets[k + 1] : n_isects) - offsets[k]);
        mx = max(mx, len);
    }
    atomicMax(&s_max, mx);
    __syncthreads();
    const float scale = 256.f / (float)(s_max + 1);
    for (int k = t; k < n_tiles; k += 1024) {
        const int len = (int)((k + 1 < n_tiles ? (int64_t)offsets[k + 1] : n_isects) - offsets[k]);
        atomicAdd(&s_hist[255 - min(255, (int)((float)len * scale))], 1);
    }
    __syncthreads();
    if (t == 0) { int run = 0; for (int b = 0; b < 256; ++b) { s_cur[b] = run; run += s_hist[b]; } }
    __syncthreads();
    for (int k = t; k < n_tiles; k += 1024) {
        const int len = (int)((k + 1 < n_tiles ? (int64_t)offsets[k + 1] : n_isects) - offsets[k]);
        g_tile_order[atomicAdd(&s_cur[255 - min(255, (int)((float)len * scale))], 1)] = k;
    }
}
#endif

template <int D, int MODE, bool CHW, bool ABS, bool PACKED>
__global__ __launch_bounds__(B2_NT, GSPL_BWD2_WAVES) void composite_bwd2_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
    const float* __restrict__ final_Ts, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_out_colors, const float* __restrict__ v_out_alphas,
    float* __restrict__ v_means2d, float* __restrict__ v_means2d_abs,
    float* __restrict__ v_conics, float* __restrict__ v_colors, float* __restrict__ v_opacities, int packed_stride,
    uint8_t* __restrict__ hit_flags) {
    using TR = ModeTraits<MODE>;
    constexpr int NV = BwdVals<D, ABS>::N;
    constexpr int RS = BwdRec<D>::STRIDE;
    constexpr bool VO_REGS = D <= 4;                  // dL/dout of the lane's phase-2 column lives in registers
    constexpr int SLAB = 2 * P2_SLOTS * 128;          // floats per wave: fac plane, sp plane, [slot][column*8 + row]
    static_assert(NV <= 16, "one ds_add round per batch");
    __shared__ int s_id[B2CHUNK];
    __shared__ __attribute__((aligned(16))) float s_rec[B2CHUNK * RS];
    __shared__ float s_acc[B2CHUNK * NV];
    __shared__ __attribute__((aligned(16))) float s_slab[B2_NW * SLAB];
    __shared__ __attribute__((aligned(16))) float s_vo_keep[VO_REGS ? 4 : B2_NW * 128 * D];
    static_assert(!VO_REGS || B2_NW * 128 * D <= B2_NW * SLAB, "s_vo alias too small");
    float* s_vo = VO_REGS ? s_slab : s_vo_keep;       // [wave][column 0..15][channel][row]
    __shared__ int s_last;

#if defined(GSPL_BWD_LPT)
    const int tile = g_tile_order[blockIdx.x];
#elif defined(GSPL_BWD_BLOCKS)
    // EXPERIMENT (A/B builds): the runs of 32 tiles an XCD works through are 8 x 4 BLOCKS of tiles instead of 32 tiles of one
    // tile row, so that vertical neighbours (which share as many splats as horizontal ones) meet in the same L2 too.
    int tile;
    {
        const int tile_h = n_tiles / tile_w, bx = (tile_w + 7) / 8, by = (tile_h + 3) / 4;
        const int sidx = xcd_remap(blockIdx.x, bx * by * 32);
        const int blk = sidx >> 5, pos = sidx & 31;
        const int txx = (blk % bx) * 8 + (pos & 7), tyy = (blk / bx) * 4 + (pos >> 3);
        if (txx >= tile_w || tyy >= tile_h) return;
        tile = tyy * tile_w + txx;
    }
#elif defined(GSPL_BWD2_SOLO)
    const int unit = xcd_remap(blockIdx.x, 2 * n_tiles, 2 * GSPL_XCD_RUN);      // (tile, half): both halves of a tile on one XCD
    const int tile = unit >> 1;
#else
    const int tile = xcd_remap(blockIdx.x, n_tiles);
#endif
#ifdef GSPL_BWD2_SOLO
    const int t = threadIdx.x, l = t;
    const int w = unit & 1;           // half tile (rows [8w, 8w+8))
    constexpr int ws = 0;             // this wave's slab / dL/dout slot in LDS
#else
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int ws = w;
#endif
    const int tx = (tile % tile_w) * TILE, ty = (tile / tile_w) * TILE;
    const int pxA = tx + (l & 7), pxB = pxA + 8;
    const int py = ty + w * 8 + (l >> 3);
    const bool insideA = (pxA < width) && (py < height), insideB = (pxB < width) && (py < height);
    const v2f pxf2 = {(float)pxA + TR::kPixelCentre, (float)pxB + TR::kPixelCentre};
    const float pyf = (float)py + TR::kPixelCentre;
    const float hx0 = (float)tx + TR::kPixelCentre;                 // centre of the half tile's first column
    const float hy0 = (float)(ty + w * 8) + TR::kPixelCentre;       // ... and first row
    const int64_t pixA = (int64_t)py * width + pxA, pixB = pixA + 8;
#ifdef GSPL_BWD2_SWZ
    // slab of one slot: rows 0-3 of column c at floats [4c, 4c+4), rows 4-7 at [64 + 4c, ...): the 16 lanes of a phase-2 slot
    // read 16 consecutive 16-byte words per ds_read_b128 (conflict-free), phase 1 writes 64 distinct dwords per instruction
    const int tl = ((l >> 3) >> 2) * 64 + (l & 7) * 4 + ((l >> 3) & 3);      // pixel A (B: + 32)
    constexpr int TLB = 32;
#else
    const int tl = (l & 7) * 8 + (l >> 3);             // pixel A's place in the column-major slab (B: + 64)
    constexpr int TLB = 64;
#endif
    const int ps = l >> 4, pc = l & 15;                // phase-2 role: splat slot, column of the half tile
    float* slab = s_slab + ws * SLAB;

    int start, end;
    tile_range(tile, n_tiles, n_isects, offsets, start, end);

    const int lastA = insideA ? last_ids[pixA] : start, lastB = insideB ? last_ids[pixB] : start;
    v2f T2 = {insideA ? final_Ts[pixA] : 1.f, insideB ? final_Ts[pixB] : 1.f};
    v2f vo[D];
    v2f bgdot = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < D; ++c) {
        vo[c] = (v2f){0.f, 0.f};
        if (insideA) vo[c].x = CHW ? v_out_colors[(int64_t)c * width * height + pixA] : v_out_colors[pixA * D + c];
        if (insideB) vo[c].y = CHW ? v_out_colors[(int64_t)c * width * height + pixB] : v_out_colors[pixB * D + c];
        if (backgrounds) bgdot += backgrounds[c] * vo[c];
        s_vo[((ws * 16 + (l & 7)) * D + c) * 8 + (l >> 3)] = vo[c].x;
        s_vo[((ws * 16 + 8 + (l & 7)) * D + c) * 8 + (l >> 3)] = vo[c].y;
    }
    const v2f v_out_a = {(insideA && v_out_alphas) ? v_out_alphas[pixA] : 0.f, (insideB && v_out_alphas) ? v_out_alphas[pixB] : 0.f};
    // R: see composite_bwd_kernel
    v2f R2 = T2 * (v_out_a - bgdot);

    if (t == 0) s_last = start;
    for (int k = t; k < B2CHUNK * NV; k += B2_NT) s_acc[k] = 0.f;
    __syncthreads();
    v2f vo2[VO_REGS ? 4 : 1][VO_REGS ? D : 1];         // phase-2 view of dL/dout: column pc, rows in pairs
    if constexpr (VO_REGS) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const float4* vp = reinterpret_cast<const float4*>(s_vo + ((ws * 16 + pc) * D + c) * 8);
            const float4 v0 = vp[0], v1 = vp[1];
            vo2[0][c] = (v2f){v0.x, v0.y}; vo2[1][c] = (v2f){v0.z, v0.w};
            vo2[2][c] = (v2f){v1.x, v1.y}; vo2[3][c] = (v2f){v1.z, v1.w};
        }
    }
    int wl = max(lastA, lastB);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wl = max(wl, __shfl_xor(wl, off));
    if (l == 0) atomicMax(&s_last, wl);
    __syncthreads();
    const int block_last = s_last;
    const int wave_last = wl;

    int nb = 0;                              // splats waiting in the phase-2 batch (wave-uniform)
    int batch_j = 0;                         // lane b holds the staged slot index of batch entry b

    auto phase2 = [&](int count) {
        __builtin_amdgcn_wave_barrier();
        float vals[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) vals[k] = 0.f;
        const int j = __builtin_amdgcn_ds_bpermute(ps << 2, batch_j);
        const bool live = ps < count;
        float ca = 0.f, cb = 0.f, cc_ = 0.f, co_ = 1.f;
        if (live) {
            const float* rec = s_rec + j * RS;
            const float4 r0 = *reinterpret_cast<const float4*>(rec);          // x y a/2 c/2
            const float2 r1 = *reinterpret_cast<const float2*>(rec + 4);      // b opacity
            ca = 2.f * r0.z; cb = r1.x; cc_ = 2.f * r0.w; co_ = r1.y;
            const float dx = r0.x - (hx0 + (float)pc);
            const float dy0 = r0.y - hy0;
            const v2f dy0v = {dy0, dy0};
#ifdef GSPL_BWD2_SWZ
            const float4* Fp = reinterpret_cast<const float4*>(slab + ps * 128 + pc * 4);
            const float4* Sp = reinterpret_cast<const float4*>(slab + P2_SLOTS * 128 + ps * 128 + pc * 4);
            const float4 f0 = Fp[0], f1 = Fp[16], q0 = Sp[0], q1 = Sp[16];
#else
            const float4* Fp = reinterpret_cast<const float4*>(slab + ps * 128 + pc * 8);
            const float4* Sp = reinterpret_cast<const float4*>(slab + P2_SLOTS * 128 + ps * 128 + pc * 8);
            const float4 f0 = Fp[0], f1 = Fp[1], q0 = Sp[0], q1 = Sp[1];
#endif
            const v2f F2[4] = {{f0.x, f0.y}, {f0.z, f0.w}, {f1.x, f1.y}, {f1.z, f1.w}};
            const v2f S2[4] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}};
            v2f s02 = {0.f, 0.f}, sy2 = {0.f, 0.f}, syy2 = {0.f, 0.f};
            v2f rgb2[D];
            float ax = 0.f, ay = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) rgb2[c] = (v2f){0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v2f dy2 = dy0v - (v2f){(float)(2 * k), (float)(2 * k + 1)};
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    v2f vv;
                    if constexpr (VO_REGS) vv = vo2[k][c];
                    else vv = *reinterpret_cast<const v2f*>(s_vo + ((ws * 16 + pc) * D + c) * 8 + 2 * k);
                    rgb2[c] = __builtin_elementwise_fma(F2[k], vv, rgb2[c]);
                }
                s02 += S2[k];
                const v2f tq = S2[k] * dy2;
                sy2 += tq;
                syy2 = __builtin_elementwise_fma(tq, dy2, syy2);
                if constexpr (ABS) {
                    ax += fabsf(S2[k].x * (ca * dx + cb * dy2.x)) + fabsf(S2[k].y * (ca * dx + cb * dy2.y));
                    ay += fabsf(S2[k].x * (cb * dx + cc_ * dy2.x)) + fabsf(S2[k].y * (cb * dx + cc_ * dy2.y));
                }
            }
            const float S0 = s02.x + s02.y, Sy = sy2.x + sy2.y;
            const float Sx = S0 * dx;
            // this lane's share of the gradients (linear in the moments, so the conversion commutes with the reduction)
            vals[0] = ca * Sx + cb * Sy;                                    // dL/dx
            vals[1] = cb * Sx + cc_ * Sy;                                   // dL/dy
            vals[2] = 0.5f * (Sx * dx);                                     // dL/da
            vals[3] = Sy * dx;                                              // dL/db
            vals[4] = 0.5f * (syy2.x + syy2.y);                             // dL/dc
            vals[5] = (co_ != 0.f) ? -S0 * __builtin_amdgcn_rcpf(co_) : 0.f;      // dL/dopacity = sum(vis * v_alpha) = -sum(sp) / o
#pragma unroll
            for (int c = 0; c < D; ++c) vals[6 + c] = rgb2[c].x + rgb2[c].y;
            if constexpr (ABS) { vals[6 + D] = ax; vals[7 + D] = ay; }
        }
        // Reduction over the 16 lanes of the slot in two halves: first inside each quad, for all NV values; then lane q of
        // every quad keeps only the values k = q (mod 4) and those are summed across the four quads (row_ror 4, 8 keep
        // q), so that the lanes of quad 0 end up owning values q, q+4, q+8, ... and add them to the tile's LDS totals.
#ifndef GSPL_ABL_NOREDUCE     // ablation builds only (timing without the first two reduction levels; results are wrong)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            vals[k] = dpp_add<0xB1, 0xF>(vals[k]);    // quad_perm [1,0,3,2]
            vals[k] = dpp_add<0x4E, 0xF>(vals[k]);    // quad_perm [2,3,0,1]
        }
#endif
        constexpr int NK = (NV + 3) / 4;
        const int pq = pc & 3;
        float kept[NK];
#pragma unroll
        for (int m = 0; m < NK; ++m) {
            float v = vals[4 * m];
#pragma unroll
            for (int q = 1; q < 4; ++q) v = (pq == q) ? ((4 * m + q < NV) ? vals[(4 * m + q < NV) ? 4 * m + q : 0] : 0.f) : v;
            kept[m] = dpp_add<0x124, 0xF>(v);         // row_ror:4
        }
        row_ror8_add<NK>(kept);
#pragma unroll
        for (int m = 0; m < NK; ++m)
            if (live && pc < 4 && 4 * m + pc < NV) atomicAdd(&s_acc[j * NV + 4 * m + pc], kept[m]);
    };

    // the staged Gaussian ids are fetched one round ahead, so that a round's gather does not wait for them
    int g_next = (block_last - 1 - t >= start && t < B2CHUNK) ? flatten_ids[block_last - 1 - t] : 0;
    for (int hi = block_last; hi > start; hi -= B2CHUNK) {
        const int lo = max(start, hi - B2CHUNK);
        const int cnt = hi - lo;
        const int g = g_next;
        {
            const int i_next = hi - B2CHUNK - 1 - t;
            if (i_next >= start && t < B2CHUNK) g_next = flatten_ids[i_next];
        }
        if (t < cnt) {
            s_id[t] = g;
            const float ca = conics[g * 3 + 0], cb = conics[g * 3 + 1], cc = conics[g * 3 + 2], op = opacities[g];
            const float mx = means2d[g * 2 + 0], my = means2d[g * 2 + 1];
            const unsigned qm = quadrant_mask(mx, my, ca, cb, cc, op, (float)tx + TR::kPixelCentre, (float)ty + TR::kPixelCentre);
            float* rec = s_rec + t * RS;
            *reinterpret_cast<float4*>(rec) = make_float4(mx, my, 0.5f * ca, 0.5f * cc);
            *reinterpret_cast<float4*>(rec + 4) = make_float4(cb, op, __uint_as_float(qm), 0.f);
#pragma unroll
            for (int c = 0; c < D; ++c) rec[8 + c] = colors[(int64_t)g * D + c];
        }
        __syncthreads();
        if (wave_last > lo) {
#pragma unroll 1
            for (int kk = 0; kk < (B2CHUNK + 63) / 64; ++kk) {
                const int slot = kk * 64 + l;
                const unsigned qm = (slot < B2CHUNK) ? __float_as_uint(s_rec[slot * RS + 6]) : 0u;
                // candidate: staged, in front of some pixel's last contributor, and able to reach alpha >= 1/255 in this half tile
                const bool cand = (slot < cnt) && (hi - 1 - slot < wave_last) && ((qm >> (2 * w)) & 3u);
                unsigned long long mask = __ballot(cand);
#ifdef GSPL_BWD2_PREFETCH
                // EXPERIMENT (A/B builds): the record of the NEXT candidate is read while the current one is processed
                float4 n_r0 = make_float4(0.f, 0.f, 0.f, 0.f);
                float2 n_r1 = make_float2(0.f, 0.f);
                float n_col[D];
                int n_j = 0;
                if (mask) {
                    n_j = kk * 64 + (int)__builtin_ctzll(mask);
                    const float* nrec = s_rec + n_j * RS;
                    n_r0 = *reinterpret_cast<const float4*>(nrec);
                    n_r1 = *reinterpret_cast<const float2*>(nrec + 4);
#if GSPL_BWD2_PREFETCH >= 2
#pragma unroll
                    for (int c = 0; c < D; ++c) n_col[c] = nrec[8 + c];
#endif
                }
#endif
                while (mask) {
#ifdef GSPL_BWD2_PREFETCH
                    const int j = n_j;
                    const float4 r0 = n_r0;
                    const float2 r1 = n_r1;
                    float col[D];
#if GSPL_BWD2_PREFETCH >= 2
#pragma unroll
                    for (int c = 0; c < D; ++c) col[c] = n_col[c];
#else
#pragma unroll
                    for (int c = 0; c < D; ++c) col[c] = s_rec[j * RS + 8 + c];
#endif
                    mask &= mask - 1;
                    if (mask) {
                        n_j = kk * 64 + (int)__builtin_ctzll(mask);
                        const float* nrec = s_rec + n_j * RS;
                        n_r0 = *reinterpret_cast<const float4*>(nrec);
                        n_r1 = *reinterpret_cast<const float2*>(nrec + 4);
#if GSPL_BWD2_PREFETCH >= 2
#pragma unroll
                        for (int c = 0; c < D; ++c) n_col[c] = nrec[8 + c];
#endif
                    }
                    const int idx = hi - 1 - j;
#else
                    const int j = kk * 64 + (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int idx = hi - 1 - j;
                    const float* rec = s_rec + j * RS;
                    const float4 r0 = *reinterpret_cast<const float4*>(rec);          // x y a/2 c/2
                    const float2 r1 = *reinterpret_cast<const float2*>(rec + 4);      // b opacity
                    float col[D];                                                    // fetched with the record: one LDS round trip per candidate
#pragma unroll
                    for (int c = 0; c < D; ++c) col[c] = rec[8 + c];
#endif
                    // sigma, bit-identical per element to eval_sigma: fma(ha dx, dx, fma(hc dy, dy, (b dx) dy))
                    const v2f dx2 = (v2f){r0.x, r0.x} - pxf2;
                    const float dy = r0.y - pyf;
                    const float hcdy = r0.w * dy;
                    const v2f dy2 = {dy, dy};
                    const v2f inner = __builtin_elementwise_fma((v2f){hcdy, hcdy}, dy2, ((v2f){r1.x, r1.x} * dx2) * dy2);
                    const v2f sigma2 = __builtin_elementwise_fma((v2f){r0.z, r0.z} * dx2, dx2, inner);
                    const v2f arg2 = sigma2 * (v2f){-1.4426950408889634f, -1.4426950408889634f};
                    const v2f vis2 = {__builtin_amdgcn_exp2f(arg2.x), __builtin_amdgcn_exp2f(arg2.y)};
                    const v2f raw2 = (v2f){r1.y, r1.y} * vis2;
                    // alpha = min(kAlphaMax, raw) >= 1/255  <=>  raw >= 1/255
                    const bool validA = (idx < lastA) && (sigma2.x >= 0.f) && (raw2.x >= kAlphaMin);
                    const bool validB = (idx < lastB) && (sigma2.y >= 0.f) && (raw2.y >= kAlphaMin);
#ifdef GSPL_COUNT_PAIRS
                    {
                        const unsigned long long ba = __ballot(validA), bb = __ballot(validB);
                        if (l == 0) {
                            atomicAdd(&g_pair_stats[0], 1ull);
                            atomicAdd(&g_pair_stats[1], (unsigned long long)(__builtin_popcountll(ba) + __builtin_popcountll(bb)));
                            if (ba | bb) atomicAdd(&g_pair_stats[2], 1ull);
                            if (ba && bb) atomicAdd(&g_pair_stats[3], 1ull);
                        }
                    }
#endif
                    if (!__any(validA || validB)) continue;
                    // some pixel takes this splat (has_hit_any_pixels): tagged in LDS with a fire-and-forget ds_or (a read-modify-write
                    // would put an LDS round trip into every candidate's critical path), reported at the flush
                    if (hit_flags && l == 0) atomicOr(&s_id[j], (int)0x80000000);
                    const v2f rv2 = {validA ? raw2.x : 0.f, validB ? raw2.y : 0.f};
#ifdef GSPL_BWD2_FMED3
                    // rv >= 0: the median of (rv, 0, alpha_max) is min(alpha_max, rv) in ONE instruction (fminf costs a
                    // canonicalising v_max in front of the v_min because the select above hides that rv is already quiet)
                    const v2f a2 = {__builtin_amdgcn_fmed3f(rv2.x, 0.f, TR::kAlphaMax), __builtin_amdgcn_fmed3f(rv2.y, 0.f, TR::kAlphaMax)};
#else
                    const v2f a2 = {fminf(TR::kAlphaMax, rv2.x), fminf(TR::kAlphaMax, rv2.y)};
#endif
                    v2f rw2 = rv2;     // o * vis where the pixel takes a gradient through alpha, else 0
                    if (TR::kClampKillsGrad) rw2 = (v2f){(rv2.x <= TR::kAlphaMax) ? rv2.x : 0.f, (rv2.y <= TR::kAlphaMax) ? rv2.y : 0.f};
                    const v2f om2 = (v2f){1.f, 1.f} - a2;
                    const v2f ra2 = {__builtin_amdgcn_rcpf(om2.x), __builtin_amdgcn_rcpf(om2.y)};
                    T2 *= ra2;                                 // transmittance in front of this splat
                    const v2f fac2 = a2 * T2;
                    v2f cdot2 = (v2f){col[0], col[0]} * vo[0];
#pragma unroll
                    for (int c = 1; c < D; ++c) cdot2 = __builtin_elementwise_fma((v2f){col[c], col[c]}, vo[c], cdot2);
                    const v2f v_alpha2 = __builtin_elementwise_fma(cdot2, T2, R2 * ra2);
                    R2 = __builtin_elementwise_fma(-cdot2, fac2, R2);
                    const v2f sp2 = -rw2 * v_alpha2;
                    float* F = slab + nb * 128 + tl;
                    F[0] = fac2.x; F[TLB] = fac2.y;
                    F[P2_SLOTS * 128] = sp2.x; F[P2_SLOTS * 128 + TLB] = sp2.y;
                    batch_j = gspl_writelane_i32(j, nb, batch_j);
                    if (++nb == P2_SLOTS) { phase2(P2_SLOTS); nb = 0; }
                }
            }
            if (nb) { phase2(nb); nb = 0; }
        }
        __syncthreads();
        if constexpr (PACKED) {
            float* __restrict__ v_packed = v_means2d;
            for (int e = t; e < cnt * NV; e += B2_NT) {
                const float v = s_acc[e];
                s_acc[e] = 0.f;
                const int row = e / NV;
                if (v != 0.f) atomicAdd(&v_packed[(int64_t)(s_id[row] & 0x7fffffff) * packed_stride + (e - row * NV)], v);
            }
        } else if (t < cnt) {
            const int g = s_id[t] & 0x7fffffff;
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
        if (hit_flags && t < cnt && s_id[t] < 0) hit_flags[s_id[t] & 0x7fffffff] = 1;      // one store per (tile, splat) that was composited
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, third structure: phase 1 of composite_bwd2_kernel (two pixels per lane, packed fp32), a phase 2 that works on
// 8x8 QUADRANT entries and raw moments.
//   * A candidate of the 16x8 half tile touches its left quadrant (the lanes' pixels A), its right one (pixels B) or both
//     (64 % on the metric workload).  Phase 1 emits one slab ENTRY per touched quadrant — decided by "some pixel of the
//     quadrant really takes the splat", not by the coarser reachability mask — instead of one 128-pixel slot per candidate whose
//     other half is zeros a third of the time.
//   * Phase 2 lanes are (quadrant q = l >> 5, entry e of 4, column c of 8): lanes 0-31 serve left-quadrant entries, lanes
//     32-63 right-quadrant ones, so that each lane's dL/dout column stays in registers.  A batch closes when either side has
//     4 entries: on average 7.3 entries = 4.45 candidates per batch (4 before), the cross-lane reduction spans 8 lanes
//     (three DPP levels) instead of 16.
//   * The reduction carries RAW moments about the splat centre (sum sp, sp dx, sp dy, sp dx^2, sp dx dy, sp dy^2) and colour
//     sums; the conversion to gradients (linear in them) happens once per (tile, splat) at the flush instead of once per
//     (wave, candidate, lane).  No `live` branch, no zero-initialised accumulators: lanes of unused entries compute on stale
//     slab contents and simply do not add (the reduction never crosses an entry).
#ifndef GSPL_BWD3_WAVES
#define GSPL_BWD3_WAVES 5
#endif
template <int D, int MODE, bool CHW, bool ABS, bool PACKED>
__global__ __launch_bounds__(128, GSPL_BWD3_WAVES) void composite_bwd3_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
    const float* __restrict__ final_Ts, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_out_colors, const float* __restrict__ v_out_alphas,
    float* __restrict__ v_means2d, float* __restrict__ v_means2d_abs,
    float* __restrict__ v_conics, float* __restrict__ v_colors, float* __restrict__ v_opacities, int packed_stride,
    uint8_t* __restrict__ hit_flags) {
    using TR = ModeTraits<MODE>;
    constexpr int NV = BwdVals<D, ABS>::N;            // s_acc row: Sx Sy Sxx Sxy Syy S0 colour[D] (abs x, abs y)
    constexpr int RS = BwdRec<D>::STRIDE;
    constexpr bool VO_REGS = D <= 4;
    constexpr int QE = 4;                             // entries per quadrant side and batch
    constexpr int PLANE = 2 * QE * 64;                // floats of one plane (fac or sp) of a wave's slab
    constexpr int SLAB = 2 * PLANE;
    static_assert(NV <= 16, "kept[] covers 16 values");
    __shared__ int s_id[B2CHUNK];
    __shared__ __attribute__((aligned(16))) float s_rec[B2CHUNK * RS];
    __shared__ float s_acc[B2CHUNK * NV];
    __shared__ __attribute__((aligned(16))) float s_slab[2 * SLAB];
    __shared__ __attribute__((aligned(16))) float s_vo_keep[VO_REGS ? 4 : 2 * 128 * D];
    static_assert(!VO_REGS || 2 * 128 * D <= 2 * SLAB, "s_vo alias too small");
    float* s_vo = VO_REGS ? s_slab : s_vo_keep;       // [wave][column 0..15][channel][row]
    __shared__ int s_last;

    const int tile = xcd_remap(blockIdx.x, n_tiles);
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int tx = (tile % tile_w) * TILE, ty = (tile / tile_w) * TILE;
    const int pxA = tx + (l & 7), pxB = pxA + 8;
    const int py = ty + w * 8 + (l >> 3);
    const bool insideA = (pxA < width) && (py < height), insideB = (pxB < width) && (py < height);
    const v2f pxf2 = {(float)pxA + TR::kPixelCentre, (float)pxB + TR::kPixelCentre};
    const float pyf = (float)py + TR::kPixelCentre;
    const float hx0 = (float)tx + TR::kPixelCentre;
    const float hy0 = (float)(ty + w * 8) + TR::kPixelCentre;
    const int64_t pixA = (int64_t)py * width + pxA, pixB = pixA + 8;
    const int tl = (l & 7) * 8 + (l >> 3);            // the pixel's place inside an entry (column-major 8x8)
    const int pq = l >> 5, pe = (l >> 3) & 3, pc = l & 7;      // phase-2 role: quadrant side, entry, column
    float* slab = s_slab + w * SLAB;

    int start, end;
    tile_range(tile, n_tiles, n_isects, offsets, start, end);

    const int lastA = insideA ? last_ids[pixA] : start, lastB = insideB ? last_ids[pixB] : start;
    v2f T2 = {insideA ? final_Ts[pixA] : 1.f, insideB ? final_Ts[pixB] : 1.f};
    v2f vo[D];
    v2f bgdot = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < D; ++c) {
        vo[c] = (v2f){0.f, 0.f};
        if (insideA) vo[c].x = CHW ? v_out_colors[(int64_t)c * width * height + pixA] : v_out_colors[pixA * D + c];
        if (insideB) vo[c].y = CHW ? v_out_colors[(int64_t)c * width * height + pixB] : v_out_colors[pixB * D + c];
        if (backgrounds) bgdot += backgrounds[c] * vo[c];
        s_vo[((w * 16 + (l & 7)) * D + c) * 8 + (l >> 3)] = vo[c].x;
        s_vo[((w * 16 + 8 + (l & 7)) * D + c) * 8 + (l >> 3)] = vo[c].y;
    }
    const v2f v_out_a = {(insideA && v_out_alphas) ? v_out_alphas[pixA] : 0.f, (insideB && v_out_alphas) ? v_out_alphas[pixB] : 0.f};
    v2f R2 = T2 * (v_out_a - bgdot);

    if (t == 0) s_last = start;
    for (int q = t; q < B2CHUNK * NV; q += 128) s_acc[q] = 0.f;
    __syncthreads();
    v2f vo2[VO_REGS ? 4 : 1][VO_REGS ? D : 1];        // phase-2 view of dL/dout: column pq * 8 + pc, rows in pairs
    if constexpr (VO_REGS) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const float4* vp = reinterpret_cast<const float4*>(s_vo + ((w * 16 + pq * 8 + pc) * D + c) * 8);
            const float4 v0 = vp[0], v1 = vp[1];
            vo2[0][c] = (v2f){v0.x, v0.y}; vo2[1][c] = (v2f){v0.z, v0.w};
            vo2[2][c] = (v2f){v1.x, v1.y}; vo2[3][c] = (v2f){v1.z, v1.w};
        }
    }
    int wl = max(lastA, lastB);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wl = max(wl, __shfl_xor(wl, off));
    if (l == 0) atomicMax(&s_last, wl);
    __syncthreads();
    const int block_last = s_last;
    const int wave_last = wl;

    int nbA = 0, nbB = 0;                    // entries waiting on the left / right side (wave-uniform)
    int batch_j = 0;                         // lane q * 4 + e holds the staged slot index of entry (q, e)

    auto phase2 = [&](int countA, int countB) {
        __builtin_amdgcn_wave_barrier();
        const int j = __builtin_amdgcn_ds_bpermute((pq * QE + pe) << 2, batch_j);
        const bool live = pe < (pq ? countB : countA);
        const float* rec = s_rec + j * RS;
        const float2 xy = *reinterpret_cast<const float2*>(rec);
        const float dx = xy.x - (hx0 + (float)(pq * 8 + pc));
        const float dy0 = xy.y - hy0;
        const v2f dy0v = {dy0, dy0};
        const float4* Fp = reinterpret_cast<const float4*>(slab + (pq * QE + pe) * 64 + pc * 8);
        const float4* Sp = reinterpret_cast<const float4*>(slab + PLANE + (pq * QE + pe) * 64 + pc * 8);
        const float4 f0 = Fp[0], f1 = Fp[1], q0 = Sp[0], q1 = Sp[1];
        const v2f F2[4] = {{f0.x, f0.y}, {f0.z, f0.w}, {f1.x, f1.y}, {f1.z, f1.w}};
        const v2f S2[4] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}};
        v2f s02 = {0.f, 0.f}, sy2 = {0.f, 0.f}, syy2 = {0.f, 0.f};
        v2f rgb2[D];
        float ax = 0.f, ay = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) rgb2[c] = (v2f){0.f, 0.f};
        float ca = 0.f, cb = 0.f, cc_ = 0.f;
        if constexpr (ABS) { ca = 2.f * rec[2]; cc_ = 2.f * rec[3]; cb = rec[4]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const v2f dy2 = dy0v - (v2f){(float)(2 * k), (float)(2 * k + 1)};
#pragma unroll
            for (int c = 0; c < D; ++c) {
                v2f vv;
                if constexpr (VO_REGS) vv = vo2[k][c];
                else vv = *reinterpret_cast<const v2f*>(s_vo + ((w * 16 + pq * 8 + pc) * D + c) * 8 + 2 * k);
                rgb2[c] = __builtin_elementwise_fma(F2[k], vv, rgb2[c]);
            }
            s02 += S2[k];
            const v2f tq = S2[k] * dy2;
            sy2 += tq;
            syy2 = __builtin_elementwise_fma(tq, dy2, syy2);
            if constexpr (ABS) {
                ax += fabsf(S2[k].x * (ca * dx + cb * dy2.x)) + fabsf(S2[k].y * (ca * dx + cb * dy2.y));
                ay += fabsf(S2[k].x * (cb * dx + cc_ * dy2.x)) + fabsf(S2[k].y * (cb * dx + cc_ * dy2.y));
            }
        }
        float vals[NV];
        const float S0 = s02.x + s02.y, Sy = sy2.x + sy2.y;
        const float Sx = S0 * dx;
        vals[0] = Sx;                        // sum sp dx
        vals[1] = Sy;                        // sum sp dy
        vals[2] = Sx * dx;                   // sum sp dx^2
        vals[3] = Sy * dx;                   // sum sp dx dy
        vals[4] = syy2.x + syy2.y;           // sum sp dy^2
        vals[5] = S0;                        // sum sp
#pragma unroll
        for (int c = 0; c < D; ++c) vals[6 + c] = rgb2[c].x + rgb2[c].y;
        if constexpr (ABS) { vals[6 + D] = ax; vals[7 + D] = ay; }
        // 8 lanes per entry: two quad levels on every value, then lane k of the first quad keeps the values k, k + 4, ... and
        // its mirror lane 7 - k of the second quad keeps the same ones; one row_half_mirror add finishes them
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            vals[k] = dpp_add<0xB1, 0xF>(vals[k]);    // quad_perm [1,0,3,2]
            vals[k] = dpp_add<0x4E, 0xF>(vals[k]);    // quad_perm [2,3,0,1]
        }
        constexpr int NK = (NV + 3) / 4;
        const int idx = (pc & 4) ? 3 - (pc & 3) : (pc & 3);
        float kept[NK];
#pragma unroll
        for (int m = 0; m < NK; ++m) {
            float v = vals[4 * m];
#pragma unroll
            for (int q = 1; q < 4; ++q) v = (idx == q) ? ((4 * m + q < NV) ? vals[(4 * m + q < NV) ? 4 * m + q : 0] : 0.f) : v;
            kept[m] = dpp_add<0x141, 0xF>(v);         // row_half_mirror
        }
#pragma unroll
        for (int m = 0; m < NK; ++m)
            if (live && pc < 4 && 4 * m + pc < NV) atomicAdd(&s_acc[j * NV + 4 * m + pc], kept[m]);
    };

    int g_next = (block_last - 1 - t >= start && t < B2CHUNK) ? flatten_ids[block_last - 1 - t] : 0;
    for (int hi = block_last; hi > start; hi -= B2CHUNK) {
        const int lo = max(start, hi - B2CHUNK);
        const int cnt = hi - lo;
        const int g = g_next;
        {
            const int i_next = hi - B2CHUNK - 1 - t;
            if (i_next >= start && t < B2CHUNK) g_next = flatten_ids[i_next];
        }
        if (t < cnt) {
            s_id[t] = g;
            const float ca = conics[g * 3 + 0], cb = conics[g * 3 + 1], cc = conics[g * 3 + 2], op = opacities[g];
            const float mx = means2d[g * 2 + 0], my = means2d[g * 2 + 1];
            const unsigned qm = quadrant_mask(mx, my, ca, cb, cc, op, (float)tx + TR::kPixelCentre, (float)ty + TR::kPixelCentre);
            float* rec = s_rec + t * RS;
            *reinterpret_cast<float4*>(rec) = make_float4(mx, my, 0.5f * ca, 0.5f * cc);
            *reinterpret_cast<float4*>(rec + 4) = make_float4(cb, op, __uint_as_float(qm), 0.f);
#pragma unroll
            for (int c = 0; c < D; ++c) rec[8 + c] = colors[(int64_t)g * D + c];
        }
        __syncthreads();
        if (wave_last > lo) {
#pragma unroll 1
            for (int kk = 0; kk < (B2CHUNK + 63) / 64; ++kk) {
                const int slot = kk * 64 + l;
                const unsigned qm = (slot < B2CHUNK) ? __float_as_uint(s_rec[slot * RS + 6]) : 0u;
                const bool cand = (slot < cnt) && (hi - 1 - slot < wave_last) && ((qm >> (2 * w)) & 3u);
                unsigned long long mask = __ballot(cand);
                while (mask) {
                    const int j = kk * 64 + (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int idx = hi - 1 - j;
                    const float* rec = s_rec + j * RS;
                    const float4 r0 = *reinterpret_cast<const float4*>(rec);          // x y a/2 c/2
                    const float2 r1 = *reinterpret_cast<const float2*>(rec + 4);      // b opacity
                    float col[D];
#pragma unroll
                    for (int c = 0; c < D; ++c) col[c] = rec[8 + c];
                    const v2f dx2 = (v2f){r0.x, r0.x} - pxf2;
                    const float dy = r0.y - pyf;
                    const float hcdy = r0.w * dy;
                    const v2f dy2 = {dy, dy};
                    const v2f inner = __builtin_elementwise_fma((v2f){hcdy, hcdy}, dy2, ((v2f){r1.x, r1.x} * dx2) * dy2);
                    const v2f sigma2 = __builtin_elementwise_fma((v2f){r0.z, r0.z} * dx2, dx2, inner);
                    const v2f arg2 = sigma2 * (v2f){-1.4426950408889634f, -1.4426950408889634f};
                    const v2f vis2 = {__builtin_amdgcn_exp2f(arg2.x), __builtin_amdgcn_exp2f(arg2.y)};
                    const v2f raw2 = (v2f){r1.y, r1.y} * vis2;
                    const bool validA = (idx < lastA) && (sigma2.x >= 0.f) && (raw2.x >= kAlphaMin);
                    const bool validB = (idx < lastB) && (sigma2.y >= 0.f) && (raw2.y >= kAlphaMin);
                    const bool anyA = __ballot(validA) != 0ull, anyB = __ballot(validB) != 0ull;      // wave-uniform
                    if (!(anyA || anyB)) continue;
                    if (hit_flags && l == 0) atomicOr(&s_id[j], (int)0x80000000);
                    // a side without room closes the batch first
                    if ((anyA && nbA == QE) || (anyB && nbB == QE)) { phase2(nbA, nbB); nbA = nbB = 0; }
                    const v2f rv2 = {validA ? raw2.x : 0.f, validB ? raw2.y : 0.f};
                    // rv >= 0: the median of (rv, 0, alpha_max) is min(alpha_max, rv) in one instruction
                    const v2f a2 = {__builtin_amdgcn_fmed3f(rv2.x, 0.f, TR::kAlphaMax), __builtin_amdgcn_fmed3f(rv2.y, 0.f, TR::kAlphaMax)};
                    v2f rw2 = rv2;
                    if (TR::kClampKillsGrad) rw2 = (v2f){(rv2.x <= TR::kAlphaMax) ? rv2.x : 0.f, (rv2.y <= TR::kAlphaMax) ? rv2.y : 0.f};
                    const v2f om2 = (v2f){1.f, 1.f} - a2;
                    const v2f ra2 = {__builtin_amdgcn_rcpf(om2.x), __builtin_amdgcn_rcpf(om2.y)};
                    T2 *= ra2;
                    const v2f fac2 = a2 * T2;
                    v2f cdot2 = (v2f){col[0], col[0]} * vo[0];
#pragma unroll
                    for (int c = 1; c < D; ++c) cdot2 = __builtin_elementwise_fma((v2f){col[c], col[c]}, vo[c], cdot2);
                    const v2f v_alpha2 = __builtin_elementwise_fma(cdot2, T2, R2 * ra2);
                    R2 = __builtin_elementwise_fma(-cdot2, fac2, R2);
                    const v2f sp2 = -rw2 * v_alpha2;
                    if (anyA) {
                        float* F = slab + nbA * 64 + tl;
                        F[0] = fac2.x; F[PLANE] = sp2.x;
                        batch_j = gspl_writelane_i32(j, nbA, batch_j);
                        nbA = __builtin_amdgcn_readfirstlane(nbA + 1);      // (keeps the counter in a scalar register)
                    }
                    if (anyB) {
                        float* F = slab + (QE + nbB) * 64 + tl;
                        F[0] = fac2.y; F[PLANE] = sp2.y;
                        batch_j = gspl_writelane_i32(j, QE + nbB, batch_j);
                        nbB = __builtin_amdgcn_readfirstlane(nbB + 1);
                    }
                }
            }
            if (nbA | nbB) { phase2(nbA, nbB); nbA = nbB = 0; }
        }
        __syncthreads();
        // raw moments -> gradients, once per staged splat and in place (thread = row: no divergence, rows are 9 floats apart:
        // no bank conflicts); then the flush of composite_bwd2_kernel: one fp32 L2 atomic per non-zero value
        if (t < cnt) {
            float* a = s_acc + t * NV;
            const float* rec = s_rec + t * RS;
            const float ca = 2.f * rec[2], cc = 2.f * rec[3], cb = rec[4], op = rec[5];
            const float Sx = a[0], Sy = a[1];
            a[0] = ca * Sx + cb * Sy;                                              // dL/dx
            a[1] = cb * Sx + cc * Sy;                                              // dL/dy
            a[2] *= 0.5f;                                                          // dL/da   (a[3] = dL/db as is)
            a[4] *= 0.5f;                                                          // dL/dc
            a[5] = (op != 0.f) ? -a[5] * __builtin_amdgcn_rcpf(op) : 0.f;          // dL/dopacity = -sum(sp) / o
        }
        __syncthreads();
        if constexpr (PACKED) {
            float* __restrict__ v_packed = v_means2d;
            for (int e = t; e < cnt * NV; e += 128) {
                const float v = s_acc[e];
                s_acc[e] = 0.f;
                const int row = e / NV;
                if (v != 0.f) atomicAdd(&v_packed[(int64_t)(s_id[row] & 0x7fffffff) * packed_stride + (e - row * NV)], v);
            }
        } else if (t < cnt) {
            const int g = s_id[t] & 0x7fffffff;
            float v[NV];
            bool any_nz = false;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                v[q] = s_acc[t * NV + q];
                s_acc[t * NV + q] = 0.f;
                any_nz = any_nz || (v[q] != 0.f);
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
        if (hit_flags && t < cnt && s_id[t] < 0) hit_flags[s_id[t] & 0x7fffffff] = 1;
        __syncthreads();
    }
}

template <int D, int MODE, bool CHW>
static int launch_fwd(int n_tiles, int tile_w, int width, int height, int64_t n_isects,
                      const float* means2d, const float* conics, const float* colors, const float* opacities,
                      const float* backgrounds, const int32_t* offsets, const int32_t* flatten_ids,
                      float* out_colors, float* out_alphas, float* final_Ts, int32_t* last_ids, uint8_t* hit_flags, hipStream_t s) {
    if (hit_flags)
        hipLaunchKernelGGL((composite_fwd_kernel<D, MODE, CHW, true>), dim3(4 * n_tiles), dim3(64), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, out_colors, out_alphas, final_Ts, last_ids, hit_flags);
    else
        hipLaunchKernelGGL((composite_fwd_kernel<D, MODE, CHW, false>), dim3(4 * n_tiles), dim3(64), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, out_colors, out_alphas, final_Ts, last_ids, hit_flags);
    return check_launch("composite_fwd");
}

template <int D, int MODE, bool CHW, bool PACKED = false>
static int launch_bwd(bool absgrad, int n_tiles, int tile_w, int width, int height, int64_t n_isects,
                      const float* means2d, const float* conics, const float* colors, const float* opacities,
                      const float* backgrounds, const int32_t* offsets, const int32_t* flatten_ids,
                      const float* final_Ts, const int32_t* last_ids,
                      const float* v_out_colors, const float* v_out_alphas,
                      float* v_means2d, float* v_means2d_abs, float* v_conics, float* v_colors, float* v_opacities,
                      hipStream_t s, int packed_stride = 0, uint8_t* hit_flags = nullptr) {
#ifdef GSPL_BWD_V3       // quadrant-entry phase 2, raw moments (A/B builds until it is the default)
    if (absgrad)
        hipLaunchKernelGGL((composite_bwd3_kernel<D, MODE, CHW, true, PACKED>), dim3(n_tiles), dim3(128), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, packed_stride, hit_flags);
    else
        hipLaunchKernelGGL((composite_bwd3_kernel<D, MODE, CHW, false, PACKED>), dim3(n_tiles), dim3(128), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, packed_stride, hit_flags);
    return check_launch("composite_bwd");
#endif
#ifndef GSPL_BWD_V2      // default: the two-pixels-per-lane kernel; -DGSPL_BWD_V2 selects the one-pixel-per-lane kernel (A/B builds)
#ifdef GSPL_BWD_LPT
    if (n_tiles > (1 << 16)) return fail_arg("composite_bwd (LPT experiment): more than 65536 tiles");
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, offsets, n_tiles, n_isects);
#endif
#ifdef GSPL_BWD_BLOCKS
    const int grid_bwd2 = ((tile_w + 7) / 8) * ((n_tiles / tile_w + 3) / 4) * 32;
#else
    const int grid_bwd2 = n_tiles * (2 / B2_NW);
#endif
    if (absgrad)
        hipLaunchKernelGGL((composite_bwd2_kernel<D, MODE, CHW, true, PACKED>), dim3(grid_bwd2), dim3(B2_NT), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, packed_stride, hit_flags);
    else
        hipLaunchKernelGGL((composite_bwd2_kernel<D, MODE, CHW, false, PACKED>), dim3(grid_bwd2), dim3(B2_NT), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, packed_stride, hit_flags);
    return check_launch("composite_bwd");
#else                    // one-pixel-per-lane kernel: instantiated in -DGSPL_BWD_V2 builds only
    if (absgrad)
        hipLaunchKernelGGL((composite_bwd_kernel<D, MODE, CHW, true, PACKED>), dim3(n_tiles), dim3(256), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, packed_stride, hit_flags);
    else
        hipLaunchKernelGGL((composite_bwd_kernel<D, MODE, CHW, false, PACKED>), dim3(n_tiles), dim3(256), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, packed_stride, hit_flags);
    return check_launch("composite_bwd");
#endif
}

// ---- per-splat statistics of a compositing pass (no image) -----------------------------------------------------------------------
// What LightGaussian-style pruning and Taming-3DGS-style densification scores read from a rasterization (reference call sites:
// internal/renderers/gsplat_hit_pixel_count_renderer.py:34-44 -> gsplat fork `hit_pixel_count`;
// internal/density_controllers/taming_3dgs_density_controller.py:429-439 -> gsplat fork `rasterize_to_weights`; both kernels are
// un-vendored, the sums are restated from the published methods).  Same traversal and the same discrete rules as
// composite_fwd_kernel (alpha >= 1/255, stop when the transmittance would fall below 1e-4); for every splat g, summed over the
// pixels p it contributes to:  count += 1, opacity += opacity[g], alpha += alpha, visibility += alpha T,
// weighted += w[p] alpha T, dist += |p - mean[g]|.  One wave per 8x8 quadrant, one candidate at a time, wave reduction, one
// atomic per (quadrant, splat, quantity).  A statistics pass, run once per pruning / scoring event: not tuned.
template <int MODE>
__global__ __launch_bounds__(64) void composite_scores_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ opacities,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids, const float* __restrict__ pixel_weights,
    int32_t* __restrict__ count, float* __restrict__ opacity_sum, float* __restrict__ alpha_sum, float* __restrict__ vis_sum,
    float* __restrict__ weighted_sum, float* __restrict__ dist_sum) {
    using TR = ModeTraits<MODE>;
    __shared__ float s_x[64], s_y[64], s_ha[64], s_b[64], s_hc[64], s_op[64];
    __shared__ int s_g[64];
    const int unit = blockIdx.x;
    const int tile = unit >> 2, w = unit & 3, l = threadIdx.x;
    const int px = (tile % tile_w) * TILE + (w & 1) * 8 + (l & 7);
    const int py = (tile / tile_w) * TILE + (w >> 1) * 8 + (l >> 3);
    const bool inside = (px < width) && (py < height);
    const float pxf = (float)px + TR::kPixelCentre, pyf = (float)py + TR::kPixelCentre;
    const float qx0 = (float)((tile % tile_w) * TILE + (w & 1) * 8) + TR::kPixelCentre, qx1 = qx0 + 7.f;
    const float qy0 = (float)((tile / tile_w) * TILE + (w >> 1) * 8) + TR::kPixelCentre, qy1 = qy0 + 7.f;
    const float wpx = (pixel_weights && inside) ? pixel_weights[(int64_t)py * width + px] : 0.f;
    int start, end;
    tile_range(tile, n_tiles, n_isects, offsets, start, end);
    float T = 1.f;
    bool done = !inside;
    for (int base = start; base < end && !__all(done); base += 64) {
        const int i = base + l;
        bool cand = false;
        float mx = 0.f, my = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, op = 0.f;
        int g = 0;
        if (i < end) {
            g = flatten_ids[i];
            ca = conics[g * 3 + 0]; cb = conics[g * 3 + 1]; cc = conics[g * 3 + 2]; op = opacities[g];
            mx = means2d[g * 2 + 0]; my = means2d[g * 2 + 1];
            cand = box_reachable(mx, my, ca, cb, cc, op, qx0, qx1, qy0, qy1);
        }
        const unsigned long long mask = __ballot(cand);
        const int ncand = __builtin_popcountll(mask);
        const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        __builtin_amdgcn_wave_barrier();
        if (cand) { s_x[slot] = mx; s_y[slot] = my; s_ha[slot] = 0.5f * ca; s_b[slot] = cb; s_hc[slot] = 0.5f * cc; s_op[slot] = op; s_g[slot] = g; }
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < ncand; ++k) {
            const float dx = s_x[k] - pxf, dy = s_y[k] - pyf, o = s_op[k];
            const float sigma = s_ha[k] * dx * dx + s_hc[k] * dy * dy + s_b[k] * dx * dy;
            const float alpha = fminf(TR::kAlphaMax, o * __expf(-sigma));
            const bool valid = !done && (sigma >= 0.f) && (alpha >= kAlphaMin);
            const float next_T = T * (1.f - alpha);
            const bool stop = valid && (TR::kStopInclusive ? (next_T <= kTStop) : (next_T < kTStop));
            const bool contrib = valid && !stop;
            const unsigned long long hit = __ballot(contrib);
            if (hit) {
                float a = contrib ? alpha : 0.f, v = contrib ? alpha * T : 0.f;
                float wv = v * wpx, ds = contrib ? sqrtf(dx * dx + dy * dy) : 0.f;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    a += __shfl_xor(a, d); v += __shfl_xor(v, d); wv += __shfl_xor(wv, d); ds += __shfl_xor(ds, d);
                }
                if (l == 0) {
                    const int gk = s_g[k];
                    const int c = __builtin_popcountll(hit);
                    if (count) atomicAdd(count + gk, c);
                    if (opacity_sum) atomicAdd(opacity_sum + gk, (float)c * o);
                    if (alpha_sum) atomicAdd(alpha_sum + gk, a);
                    if (vis_sum) atomicAdd(vis_sum + gk, v);
                    if (weighted_sum) atomicAdd(weighted_sum + gk, wv);
                    if (dist_sum) atomicAdd(dist_sum + gk, ds);
                }
            }
            T = contrib ? next_T : T;
            done = done || stop;
            if (__all(done)) break;
        }
    }
}

static int check_common(int N, int64_t n_isects, int D, int mode, int layout, int width, int height,
                        int tile_size, int tile_w, int tile_h, const char* who) {
    if (N < 0 || n_isects < -1 || width <= 0 || height <= 0) return fail_arg(who);
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
                                  float* out_colors, float* out_alphas, float* final_Ts, int32_t* last_ids,
                                  uint8_t* hit_flags, void* stream) {
    using namespace gspl;
    int rc = check_common(N, n_isects, D, mode, layout, width, height, tile_size, tile_w, tile_h, "composite_fwd: bad argument");
    if (rc != GSPL_OK) return rc;
    if (!offsets || !out_colors || !out_alphas || !final_Ts || !last_ids) return fail_arg("composite_fwd: NULL required pointer");
    if (n_isects != 0 && (!means2d || !conics || !colors || !opacities || !flatten_ids)) return fail_arg("composite_fwd: NULL required pointer");
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = (hipStream_t)stream;
    rc = GSPL_ERR_UNSUPPORTED;
#define CALL_FWD(kD, M, C) rc = launch_fwd<kD, M, C>(n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, out_colors, out_alphas, final_Ts, last_ids, hit_flags, s)
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
                                  float* v_conics, float* v_colors, float* v_opacities, uint8_t* hit_flags, void* stream) {
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
#define CALL_BWD(kD, M, C) rc = launch_bwd<kD, M, C>(absgrad, n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, s, 0, hit_flags)
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

// Same backward, gradients delivered as ONE packed row per splat: v_packed [N, packed_stride >= 6 + D (+2 with absgrad)] =
// (dL/dx, dL/dy, dL/da, dL/db, dL/dc, dL/dopacity, dL/dcolour[D], [sum|dL/dx|, sum|dL/dy|]); must be zero-initialised.
// The flush then issues atomics whose 64 lanes cover contiguous components of a few rows instead of 64 scattered
// dwords per instruction (see kernel).  Consumers read the columns with a row stride (gspl_inria_preprocess_bwd's
// grad_stride, or strided views on the host side).
extern "C" int gspl_composite_bwd_packed(int N, int64_t n_isects, int D, int mode, int layout,
                                         const float* means2d, const float* conics, const float* colors,
                                         const float* opacities, const float* backgrounds,
                                         int width, int height, int tile_size, int tile_w, int tile_h,
                                         const int32_t* offsets, const int32_t* flatten_ids,
                                         const float* final_Ts, const int32_t* last_ids,
                                         const float* v_out_colors, const float* v_out_alphas,
                                         float* v_packed, int packed_stride, int absgrad, uint8_t* hit_flags, void* stream) {
    using namespace gspl;
    int rc = check_common(N, n_isects, D, mode, layout, width, height, tile_size, tile_w, tile_h, "composite_bwd_packed: bad argument");
    if (rc != GSPL_OK) return rc;
    if (packed_stride < 6 + D + (absgrad ? 2 : 0)) return fail_arg("composite_bwd_packed: packed_stride smaller than the row");
    if (n_isects == 0 || N == 0) return GSPL_OK;
    if (!means2d || !conics || !colors || !opacities || !offsets || !flatten_ids || !final_Ts || !last_ids || !v_out_colors || !v_packed)
        return fail_arg("composite_bwd_packed: NULL required pointer");
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = (hipStream_t)stream;
    const bool ag = absgrad != 0;
    rc = GSPL_ERR_UNSUPPORTED;
#define CALL_BWDP(kD, M, C) rc = launch_bwd<kD, M, C, true>(ag, n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas, v_packed, nullptr, nullptr, nullptr, nullptr, s, packed_stride, hit_flags)
    if (mode == GSPL_MODE_GSPLAT) {
        if (layout == GSPL_LAYOUT_HWC) { GSPL_DISPATCH_D(D, GSPL_MODE_GSPLAT, false, CALL_BWDP) }
        else { GSPL_DISPATCH_D(D, GSPL_MODE_GSPLAT, true, CALL_BWDP) }
    } else {
        if (layout == GSPL_LAYOUT_HWC) { GSPL_DISPATCH_D(D, GSPL_MODE_INRIA, false, CALL_BWDP) }
        else { GSPL_DISPATCH_D(D, GSPL_MODE_INRIA, true, CALL_BWDP) }
    }
#undef CALL_BWDP
    return rc;
}

// Name of the kernel template gspl_composite_bwd / gspl_composite_bwd_packed launch in this build (for profile look-ups).
extern "C" const char* gspl_composite_bwd_kernel_name(void) {
#if defined(GSPL_BWD_V3)
    return "composite_bwd3_kernel";
#elif !defined(GSPL_BWD_V2)
    return "composite_bwd2_kernel";
#else
    return "composite_bwd_kernel";
#endif
}

#ifdef GSPL_COUNT_PAIRS
// instrumentation build only: copy (and optionally reset) the counters of the compositing kernels
extern "C" int gspl_debug_pair_stats(unsigned long long* out8, int reset) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(gspl::g_pair_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gspl::g_pair_stats), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

extern "C" int gspl_composite_scores(int N, int64_t n_isects, int mode,
                                     const float* means2d, const float* conics, const float* opacities,
                                     int width, int height, int tile_size, int tile_w, int tile_h,
                                     const int32_t* offsets, const int32_t* flatten_ids, const float* pixel_weights,
                                     int32_t* count, float* opacity_sum, float* alpha_sum, float* visibility_sum,
                                     float* weighted_sum, float* dist_sum, void* stream) {
    using namespace gspl;
    int rc = check_common(N, n_isects, 1, mode, GSPL_LAYOUT_HWC, width, height, tile_size, tile_w, tile_h, "composite_scores: bad argument");
    if (rc != GSPL_OK) return rc;
    if (N == 0 || n_isects == 0) return GSPL_OK;
    if (!means2d || !conics || !opacities || !offsets || !flatten_ids) return fail_arg("composite_scores: NULL required pointer");
    if (weighted_sum && !pixel_weights) return fail_arg("composite_scores: weighted_sum needs pixel_weights");
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = (hipStream_t)stream;
    if (mode == GSPL_MODE_GSPLAT)
        hipLaunchKernelGGL(composite_scores_kernel<GSPL_MODE_GSPLAT>, dim3(4 * n_tiles), dim3(64), 0, s, n_tiles, tile_w, width, height, n_isects, means2d, conics,
                           opacities, offsets, flatten_ids, pixel_weights, count, opacity_sum, alpha_sum, visibility_sum, weighted_sum, dist_sum);
    else
        hipLaunchKernelGGL(composite_scores_kernel<GSPL_MODE_INRIA>, dim3(4 * n_tiles), dim3(64), 0, s, n_tiles, tile_w, width, height, n_isects, means2d, conics,
                           opacities, offsets, flatten_ids, pixel_weights, count, opacity_sum, alpha_sum, visibility_sum, weighted_sum, dist_sum);
    return check_launch("composite_scores");
}
