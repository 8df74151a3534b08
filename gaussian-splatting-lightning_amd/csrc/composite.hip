// composite.hip — 16x16-tile alpha compositing, FORWARD, and the per-splat statistics pass (gfx950, wave64).
//
// Replaces gsplat `rasterize_to_pixels` (internal/renderers/gsplat_v1_renderer.py:588-601), v0
// `rasterize_gaussians` (gsplat_renderer.py:86-99, pypreprocess_gsplat_renderer.py:45-58) and the
// render stage of the Inria `GaussianRasterizer` (vanilla_renderer.py:111-120).  The backward is in composite_bwd.hip.
// Neither CUDA package is vendored in the reference; the algorithm restated here is the published
// 3DGS compositing rule with the per-API constants of SURVEY.md Appendix B (ModeTraits).
// Roofline: algorithmic bytes 40*I + 20*P; VALU/exp-bound under that model — bench.py reports both fractions.
#include "gspl_composite.h"

namespace gspl {

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
static constexpr int FCHUNK = 64;
static constexpr int FLIST = FCHUNK + 2;      // room for the odd-count padding entry


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
    int32_t* __restrict__ last_ids, uint8_t* __restrict__ hit_flags, ListTiles lt, SegState seg) {
    using TR = ModeTraits<MODE>;
    __shared__ __attribute__((aligned(16))) float s_x[FLIST];
    __shared__ __attribute__((aligned(16))) float s_y[FLIST];
    __shared__ __attribute__((aligned(16))) float s_ha[FLIST];
    __shared__ __attribute__((aligned(16))) float s_k[FLIST];       // sigma_coef: sigma = ha (dx + k dy)^2 + hd dy^2
    __shared__ __attribute__((aligned(16))) float s_hd[FLIST];
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

    int start, end;      // the list of the LIST tile this 8x8 block lies in (n_tiles / tile_w are the 16x16 compute grid)
    block_list_range(lt, (tile % tile_w) * 2 + (w & 1), (tile / tile_w) * 2 + (w >> 1), width, height, n_isects, offsets, start, end);

    zero_table(seg.zero_p, seg.zero_n16);      // the backward's packed rows (this kernel waits for VALU issue: the stores ride for free)
    if (seg.walk && blockIdx.x == 0) {
        // the walk statistics the LAST backward left (complete: it is behind us on the stream): does the frame have a tail?
        static_assert(SEG_WALK_SLOTS == 64, "one slot per lane");
        uint32_t sum = seg.walk[l * 4 + 0], mx = seg.walk[l * 4 + 1], cnt = seg.walk[l * 4 + 2];
        seg.walk[l * 4 + 0] = 0u; seg.walk[l * 4 + 1] = 0u; seg.walk[l * 4 + 2] = 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum += __shfl_xor(sum, off); cnt += __shfl_xor(cnt, off);
            const uint32_t o = __shfl_xor(mx, off); mx = o > mx ? o : mx;
        }
        if (l == 0 && seg.host_flag) {
            const bool tail = cnt > 0u && mx > (uint32_t)SEG_TRIGGER && 2ull * mx * cnt > (unsigned long long)SEG_TAIL_X2 * sum;
            // (cnt == 0: no backward ran since the last forward — no verdict, the slot keeps its old ticket)
            if (cnt > 0u) *(volatile uint32_t*)seg.host_flag = (seg.ticket << 1) | (tail ? 1u : 0u);
        }
    }
    float T = 1.f;
    float acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    int last = start;            // one past the last contributing index
    bool done = !inside;
    int n_ckpt = 0;              // checkpoints this wave has left (boundaries start + k SEG, k = 1 .. n_ckpt)

    if (!__all(done)) {
        int g_next = (start + l < end) ? flatten_ids[start + l] : 0;
        for (int base = start; base < end; base += FCHUNK) {
            if constexpr (D == 3) {
                // segmented backward: every pixel's state in front of list position `base`, each SEG entries (gspl_composite.h)
                if (seg.ckpt && base > start && (((base - start) & (SEG - 1)) == 0)) {
                    seg.ckpt[(size_t)((unsigned)base >> SEG_LOG2) * 256u + (unsigned)(w * 64 + l)] = make_float4(T, acc[0], acc[1], acc[2]);
                    acc[0] = acc[1] = acc[2] = 0.f;      // the colour restarts with every segment (see below)
                    ++n_ckpt;
                }
            }
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
            if (ncand == 0) continue;
            // compaction: candidate k of the round goes to slot k (k = number of candidates in lower lanes)
            const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (cand) {
                s_x[slot] = xy.x; s_y[slot] = xy.y;
                const SigmaCoef sc = sigma_coef(ca, cb, cc);
                s_ha[slot] = sc.ha; s_k[slot] = sc.k; s_hd[slot] = sc.hd; s_op[slot] = op;
                s_pos[slot] = i + 1;
#pragma unroll
                for (int c = 0; c < D; ++c) s_col[slot * D + c] = colors[(int64_t)g * D + c];
            }
            if (l == 0) {        // padding entry for an odd count: opacity 0 -> alpha 0 -> never valid
                s_x[ncand] = 0.f; s_y[ncand] = 0.f; s_ha[ncand] = 0.f; s_k[ncand] = 0.f; s_hd[ncand] = 0.f; s_op[ncand] = 0.f;
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
                const v2f k2 = *reinterpret_cast<const v2f*>(&s_k[k]);
                const v2f hd2 = *reinterpret_cast<const v2f*>(&s_hd[k]);
                const v2f op2 = *reinterpret_cast<const v2f*>(&s_op[k]);
                const int2 pos2 = *reinterpret_cast<const int2*>(&s_pos[k]);
                const v2f dx2 = x2 - pxf2, dy2 = y2 - pyf2;
                const v2f sigma2 = eval_sigma2(ha2, k2, hd2, dx2, dy2);
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
                if (__all(done)) { all_done = true; break; }
            }
            if constexpr (HITS) {
                if (cand && ((hitmask >> slot) & 1ull)) hit_flags[g] = 1;
            }
            if (all_done) break;
        }
    }

    if constexpr (D == 3) {
        if (seg.ckpt) {
            if (blockIdx.x == 0 && l < 2) seg.words[l] = 0u;      // the backward's item counter is zero when the backward starts
            // checkpoint k holds the colour of segment k - 1 and `acc` that of the last one: suffix sums from the back (every lane
            // rewrites its own pixel's entries), and the image = the sum of all segments
            for (int k = n_ckpt; k >= 1; --k) {
                float4* c = seg.ckpt + (size_t)((unsigned)(start + k * SEG) >> SEG_LOG2) * 256u + (unsigned)(w * 64 + l);
                const float4 v = *c;
                *c = make_float4(v.x, acc[0], acc[1], acc[2]);
                acc[0] += v.y; acc[1] += v.z; acc[2] += v.w;
            }
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

template <int D, int MODE, bool CHW>
static int launch_fwd(int n_tiles, int tile_w, int width, int height, int64_t n_isects,
                      const float* means2d, const float* conics, const float* colors, const float* opacities,
                      const float* backgrounds, const int32_t* offsets, const int32_t* flatten_ids,
                      float* out_colors, float* out_alphas, float* final_Ts, int32_t* last_ids, uint8_t* hit_flags, hipStream_t s, ListTiles lt,
                      const SegState* seg_in) {
    SegState seg = {};
    if (seg_in && D == 3 && lt.log2 == 4) seg = *seg_in;      // checkpoints: 16-pixel list tiles, three channels
    else if (seg_in) { seg.zero_p = seg_in->zero_p; seg.zero_n16 = seg_in->zero_n16; }
    if (hit_flags)
        hipLaunchKernelGGL((composite_fwd_kernel<D, MODE, CHW, true>), dim3(4 * n_tiles), dim3(64), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, out_colors, out_alphas, final_Ts, last_ids, hit_flags, lt, seg);
    else
        hipLaunchKernelGGL((composite_fwd_kernel<D, MODE, CHW, false>), dim3(4 * n_tiles), dim3(64), 0, s,
                           n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds,
                           offsets, flatten_ids, out_colors, out_alphas, final_Ts, last_ids, hit_flags, lt, seg);
    return check_launch("composite_fwd");
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
    __shared__ float s_x[64], s_y[64], s_ha[64], s_k[64], s_hd[64], s_op[64];
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
        if (cand) {
            const SigmaCoef sc = sigma_coef(ca, cb, cc);
            s_x[slot] = mx; s_y[slot] = my; s_ha[slot] = sc.ha; s_k[slot] = sc.k; s_hd[slot] = sc.hd; s_op[slot] = op; s_g[slot] = g;
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < ncand; ++k) {
            const float dx = s_x[k] - pxf, dy = s_y[k] - pyf, o = s_op[k];
            // the compositing kernels' own expression tree (eval_sigma, exp2 of the scaled argument): the same pairs are counted as blended
            const float sigma = eval_sigma(s_ha[k], s_k[k], s_hd[k], dx, dy);
            const float alpha = fminf(TR::kAlphaMax, o * __builtin_amdgcn_exp2f(sigma * -1.4426950408889634f));
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

int check_composite_args(int N, int64_t n_isects, int D, int mode, int layout, int width, int height,
                        int tile_size, int tile_w, int tile_h, const char* who) {
    if (N < 0 || n_isects < -1 || width <= 0 || height <= 0) return fail_arg(who);
    if (tile_size != 8 && tile_size != 16 && tile_size != 32) { set_error(who, "tile_size must be 8, 16 or 32"); return GSPL_ERR_UNSUPPORTED; }
    if (tile_w != (width + tile_size - 1) / tile_size || tile_h != (height + tile_size - 1) / tile_size) return fail_arg(who);
    if (mode != GSPL_MODE_GSPLAT && mode != GSPL_MODE_INRIA) return fail_arg(who);
    if (layout != GSPL_LAYOUT_HWC && layout != GSPL_LAYOUT_CHW) return fail_arg(who);
    if (!(D == 1 || D == 2 || D == 3 || D == 4 || D == 8)) { set_error(who, "D must be 1,2,3,4 or 8"); return GSPL_ERR_UNSUPPORTED; }
    if (n_isects > 0x7fffffffll) return fail_arg(who);
    return GSPL_OK;
}

}  // namespace gspl

extern "C" int gspl_composite_fwd(int N, int64_t n_isects, int D, int mode, int layout,
                                  const float* means2d, const float* conics, const float* colors,
                                  const float* opacities, const float* backgrounds,
                                  int width, int height, int tile_size, int tile_w, int tile_h,
                                  const int32_t* offsets, const int32_t* flatten_ids,
                                  float* out_colors, float* out_alphas, float* final_Ts, int32_t* last_ids,
                                  uint8_t* hit_flags, void* stream) {
    return gspl::composite_fwd_impl(N, n_isects, D, mode, layout, means2d, conics, colors, opacities, backgrounds, width, height, tile_size, tile_w, tile_h,
                                    offsets, flatten_ids, out_colors, out_alphas, final_Ts, last_ids, hit_flags, stream, nullptr);
}

// (seg: the fused Inria call's checkpoint state for the segmented backward, gspl_composite.h; NULL = off)
int gspl::composite_fwd_impl(int N, int64_t n_isects, int D, int mode, int layout,
                                  const float* means2d, const float* conics, const float* colors,
                                  const float* opacities, const float* backgrounds,
                                  int width, int height, int tile_size, int tile_w, int tile_h,
                                  const int32_t* offsets, const int32_t* flatten_ids,
                                  float* out_colors, float* out_alphas, float* final_Ts, int32_t* last_ids,
                                  uint8_t* hit_flags, void* stream, const SegState* seg) {
    using namespace gspl;
    int rc = check_composite_args(N, n_isects, D, mode, layout, width, height, tile_size, tile_w, tile_h, "composite_fwd: bad argument");
    if (rc != GSPL_OK) return rc;
    if (!offsets || !out_colors || !out_alphas || !final_Ts || !last_ids) return fail_arg("composite_fwd: NULL required pointer");
    if (n_isects != 0 && (!means2d || !conics || !colors || !opacities || !flatten_ids)) return fail_arg("composite_fwd: NULL required pointer");
    // the kernel walks 8x8 blocks grouped into 16x16 compute tiles; the lists are those of the caller's tile_size (8, 16 or 32)
    const ListTiles lt = list_tiles(tile_size, tile_w, tile_h);
    const int ctw = (width + TILE - 1) / TILE, n_tiles = ctw * ((height + TILE - 1) / TILE);
    hipStream_t s = (hipStream_t)stream;
    rc = GSPL_ERR_UNSUPPORTED;
#define CALL_FWD(kD, M, C) rc = launch_fwd<kD, M, C>(n_tiles, ctw, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, out_colors, out_alphas, final_Ts, last_ids, hit_flags, s, lt, seg)
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

extern "C" int gspl_composite_scores(int N, int64_t n_isects, int mode,
                                     const float* means2d, const float* conics, const float* opacities,
                                     int width, int height, int tile_size, int tile_w, int tile_h,
                                     const int32_t* offsets, const int32_t* flatten_ids, const float* pixel_weights,
                                     int32_t* count, float* opacity_sum, float* alpha_sum, float* visibility_sum,
                                     float* weighted_sum, float* dist_sum, void* stream) {
    using namespace gspl;
    int rc = check_composite_args(N, n_isects, 1, mode, GSPL_LAYOUT_HWC, width, height, tile_size, tile_w, tile_h, "composite_scores: bad argument");
    if (rc != GSPL_OK) return rc;
    if (tile_size != TILE) { set_error("composite_scores", "the statistics pass is built for tile_size 16"); return GSPL_ERR_UNSUPPORTED; }
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
