// composite_bwd.hip — 16x16-tile alpha compositing, BACKWARD (gfx950, wave64).
//
// Autograd backward of gsplat `rasterize_to_pixels` (internal/renderers/gsplat_v1_renderer.py:588-601), v0 `rasterize_gaussians`
// (gsplat_renderer.py:86-99) and the render stage of the Inria `GaussianRasterizer` (vanilla_renderer.py:111-120), which the
// reference enters through `manual_backward` (gaussian_splatting.py:380).  Neither CUDA package is vendored in the reference; the
// algorithm restated here is the published 3DGS compositing rule with the per-API constants of SURVEY.md Appendix B (ModeTraits).
//
// Two kernels:
//   composite_bwd4_kernel  default for D <= 4: ONE wave per tile, FOUR pixels per lane, EIGHT 8x4-pixel units that each walk their
//                          own candidate queue; the per-splat reduction is in-lane + 8 lanes of DPP + LDS adds (no transposition).
//   composite_bwd2_kernel  D = 8 (and GSPL_BWD_KERNEL=2): two waves per tile, two pixels per lane, slab-transposed phase 2.
// Both deliver ONE fp32 L2 atomic per value per (tile, splat): 36 B per intersection, the algorithmic minimum of SURVEY.md §8d.
// Roofline: algorithmic bytes 76*I + 20*P (+8*I with absgrad); the kernels are VALU-bound under that model.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "gspl_composite.h"

namespace gspl {

// Staged record of composite_bwd2_kernel in LDS (floats): x y a/2 c/2 | b opacity quadrant-mask - | colour[D] (padded to a
// multiple of 4): one address register per candidate, the fields are fetched with immediate offsets (b128 + b64 [+ colour]).
template <int D> struct BwdRec { static constexpr int STRIDE = 8 + ((D + 3) & ~3); };

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
static constexpr int B2_NW = 2;                   // waves per workgroup: the two half tiles of a tile share the staged records
static constexpr int B2_NT = 64 * B2_NW;
static_assert(B2CHUNK <= B2_NT || B2_NW == 2, "a round is staged by one pass of the workgroup's threads");


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

    const int tile = xcd_remap(blockIdx.x, n_tiles);
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int ws = w;
    const int tx = (tile % tile_w) * TILE, ty = (tile / tile_w) * TILE;
    const int pxA = tx + (l & 7), pxB = pxA + 8;
    const int py = ty + w * 8 + (l >> 3);
    const bool insideA = (pxA < width) && (py < height), insideB = (pxB < width) && (py < height);
    const v2f pxf2 = {(float)pxA + TR::kPixelCentre, (float)pxB + TR::kPixelCentre};
    const float pyf = (float)py + TR::kPixelCentre;
    const float hx0 = (float)tx + TR::kPixelCentre;                 // centre of the half tile's first column
    const float hy0 = (float)(ty + w * 8) + TR::kPixelCentre;       // ... and first row
    const int64_t pixA = (int64_t)py * width + pxA, pixB = pixA + 8;
    const int tl = (l & 7) * 8 + (l >> 3);             // pixel A's place in the column-major slab (B: + 64)
    constexpr int TLB = 64;
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
            const float4* Fp = reinterpret_cast<const float4*>(slab + ps * 128 + pc * 8);
            const float4* Sp = reinterpret_cast<const float4*>(slab + P2_SLOTS * 128 + ps * 128 + pc * 8);
            const float4 f0 = Fp[0], f1 = Fp[1], q0 = Sp[0], q1 = Sp[1];
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
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            vals[k] = dpp_add<0xB1, 0xF>(vals[k]);    // quad_perm [1,0,3,2]
            vals[k] = dpp_add<0x4E, 0xF>(vals[k]);    // quad_perm [2,3,0,1]
        }
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
            const unsigned qm = band_half_mask<2>(mx, my, ca, cb, cc, op, (float)tx + TR::kPixelCentre, (float)ty + TR::kPixelCentre);
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
                while (mask) {
                    const int j = kk * 64 + (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int idx = hi - 1 - j;
                    const float* rec = s_rec + j * RS;
                    const float4 r0 = *reinterpret_cast<const float4*>(rec);          // x y a/2 c/2
                    const float2 r1 = *reinterpret_cast<const float2*>(rec + 4);      // b opacity
                    float col[D];                                                    // fetched with the record: one LDS round trip per candidate
#pragma unroll
                    for (int c = 0; c < D; ++c) col[c] = rec[8 + c];
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
                    if (!__any(validA || validB)) continue;
                    // some pixel takes this splat (has_hit_any_pixels): tagged in LDS with a fire-and-forget ds_or (a read-modify-write
                    // would put an LDS round trip into every candidate's critical path), reported at the flush
                    if (hit_flags && l == 0) atomicOr(&s_id[j], (int)0x80000000);
                    const v2f rv2 = {validA ? raw2.x : 0.f, validB ? raw2.y : 0.f};
                    const v2f a2 = {fminf(TR::kAlphaMax, rv2.x), fminf(TR::kAlphaMax, rv2.y)};
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
// Backward, ONE WAVE PER TILE, FOUR PIXELS PER LANE, EIGHT UNIT QUEUES (default for D <= 4).
//
// Why (tools/analysis/pair_stats.py, S-1080p-1M): a wave that walks ONE candidate list for its 128 or 256 pixels evaluates two
// pixel slots for every valid (pixel, splat) pair — the splats are a few pixels wide and the pixels of a tile saturate at
// different depths.  Here the tile is cut into eight UNITS of 8 x 4 pixels (unit u: columns [8 (u & 1), +8), rows
// [4 (u >> 1), +4)); unit u is owned by lanes [8u, 8u + 8), lane 8u + i carries the four pixels of column i of the unit, and every
// unit walks ITS OWN queue of candidates: the staged splats that can reach alpha >= 1/255 inside the unit's box (exact
// ellipse-vs-box test, band_half_mask<4>) and lie in front of the deepest last contributor of the unit's 32 pixels.  One
// iteration of the wave therefore processes eight different (unit, splat) pairs; it runs max_u(queue length) iterations per
// round: 0.76 M iterations of 256 pixel slots on the metric workload (60 % of them valid) where composite_bwd2_kernel runs 1.86 M
// of 128 (49 %).
// Because the lanes of a unit all hold the SAME splat, the per-splat sums never have to be transposed: each lane adds up its own
// four pixels (moments sum sp, sum sp dy, sum sp dy^2 and the colour sums; dx is constant down its column), converts them to
// gradient shares (linear, so before the reduction), two DPP quad steps finish each half of the unit and lanes 0 and 4 of the unit
// add the NV values to the splat's row of the round's LDS totals.  No slab, no second phase, no workgroup barrier in the loop,
// no exec masking (a pixel that does not take a splat, and a unit whose queue has run out, run with alpha = 0: exactly neutral).
// Per round of 64 staged splats: gather + mask (lane = staged splat), eight ballots compact the queues into LDS (one byte per
// entry), walk, flush = one fp32 L2 atomic per value per (tile, splat) into the packed gradient rows.
// All per-pixel arithmetic is the same expression tree as composite_bwd2_kernel / the forward (eval_sigma order; exp2; rcp).
#ifndef GSPL_BWD4_WAVES
#define GSPL_BWD4_WAVES 5
#endif
#ifndef GSPL_BWD4_CHUNK
#define GSPL_BWD4_CHUNK 64
#endif
#ifndef GSPL_BWD4_PK
#define GSPL_BWD4_PK 0
#endif
#ifndef GSPL_BWD4_PRIO
#define GSPL_BWD4_PRIO 0
#endif
static constexpr int B4CHUNK = GSPL_BWD4_CHUNK;     // splats staged per round (at most one per lane)
static_assert(B4CHUNK <= 64 && B4CHUNK % 8 == 0, "one staged splat per lane");
static constexpr int B4RS = 12;        // floats per staged record: x y a/2 c/2 | b opacity -1/opacity a | colour[<= 3] c (or colour[4])
static constexpr int B4DUMMY = B4CHUNK;   // record slot of an exhausted queue: opacity 0
// Transposition buffer of the walk: row v holds value v of all 64 lanes ([unit][lane of the unit]); a lane WRITES its share of value k
// to row k (64 consecutive floats per instruction: conflict-free) and READS the eight shares of its unit's value li as four 8-byte
// reads.  The rows start at 64 v + 2 (v & 3) + 32 (v >> 2) floats: with that skew the 32 lanes of an 8-byte read group (4 units x 8
// values) start on 32 different even banks — two plain 16-byte reads of unskewed rows cost 32 LDS cycles per iteration instead of 8.
__host__ __device__ constexpr int b4_t_row(int v) { return 64 * v + 2 * (v & 3) + 32 * (v >> 2); }
static constexpr int B4T_FLOATS = b4_t_row(7) + 64;

template <int CTRL>
__device__ __forceinline__ int dpp_max_i32(int v) {
    const int moved = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
    return max(v, moved);
}

// Pixel pair {row k, row k + 1} of a lane's column as a 2-vector: with GSPL_BWD4_PK the arithmetic of the walk is written on pairs
// (v_pk_mul/add/fma_f32: measured 5.2 / 4.9 / 6.5 cycles per wave-instruction against 4.0 / 3.9 / 3.3 for the plain forms at four
// waves per SIMD, tools/micro/valu_rates.hip — the multiplies and adds are a third cheaper per element, the fmas equal).
template <int D, int MODE, bool CHW, bool ABS, bool PACKED>
__global__ __launch_bounds__(64, GSPL_BWD4_WAVES) void composite_bwd4_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
    const float* __restrict__ final_Ts, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_out_colors, const float* __restrict__ v_out_alphas,
    float* __restrict__ v_means2d, float* __restrict__ v_means2d_abs,
    float* __restrict__ v_conics, float* __restrict__ v_colors, float* __restrict__ v_opacities, int packed_stride,
    uint8_t* __restrict__ hit_flags, const int32_t* __restrict__ tile_order) {
    static_assert(D <= 4, "the record holds four colour channels");
    using TR = ModeTraits<MODE>;
    constexpr int NV = BwdVals<D, ABS>::N;
    constexpr int NHI = NV > 8 ? NV - 8 : 0;          // values beyond the eight that travel through the LDS transposition
    constexpr int RS = B4RS;
    __shared__ int s_id[B4CHUNK];
    __shared__ __attribute__((aligned(16))) float s_rec[(B4CHUNK + 1) * RS];
    __shared__ float s_acc[B4CHUNK * NV];
    __shared__ __attribute__((aligned(16))) float s_t[B4T_FLOATS];     // [value 0..7][unit][lane of the unit], rows skewed (b4_t_row)
    __shared__ uint8_t s_q[8 * B4CHUNK];
    __shared__ uint8_t s_tag[B4CHUNK];

    const int tile = tile_order ? tile_order[blockIdx.x] : xcd_remap(blockIdx.x, n_tiles);
    const int l = threadIdx.x;
    const int u = l >> 3, li = l & 7;                       // unit of this lane, lane within the unit
    const int tx = (tile % tile_w) * TILE, ty = (tile / tile_w) * TILE;
    const int px = tx + (u & 1) * 8 + li;
    const int py0 = ty + (u >> 1) * 4;
    const float pxf = (float)px + TR::kPixelCentre;
    float pyf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pyf[k] = (float)(py0 + k) + TR::kPixelCentre;

    int start, end;
    tile_range(tile, n_tiles, n_isects, offsets, start, end);

    // per-pixel state: transmittance T (starts at the final one and is divided back), R (see composite_bwd2_kernel), dL/dout
    int last[4];
    float T[4], R[4], vo[D][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool inside = (px < width) && (py0 + k < height);
        const int64_t pix = (int64_t)(py0 + k) * width + px;
        last[k] = inside ? last_ids[pix] : start;
        T[k] = inside ? final_Ts[pix] : 1.f;
        float bgdot = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            vo[c][k] = inside ? (CHW ? v_out_colors[(int64_t)c * width * height + pix] : v_out_colors[pix * D + c]) : 0.f;
            if (backgrounds) bgdot += backgrounds[c] * vo[c][k];
        }
        const float voa = (inside && v_out_alphas) ? v_out_alphas[pix] : 0.f;
        R[k] = T[k] * (voa - bgdot);
    }

    // deepest last contributor of each unit (all eight lanes of the unit hold it), and of the tile
    int ul = max(max(last[0], last[1]), max(last[2], last[3]));
    ul = dpp_max_i32<0xB1>(ul);      // quad_perm [1,0,3,2]
    ul = dpp_max_i32<0x4E>(ul);      // quad_perm [2,3,0,1]
    ul = dpp_max_i32<0x141>(ul);     // row_half_mirror
    int ulast[8];
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) ulast[uu] = __builtin_amdgcn_readlane(ul, 8 * uu);
    int tile_last = ulast[0];
#pragma unroll
    for (int uu = 1; uu < 8; ++uu) tile_last = max(tile_last, ulast[uu]);

    for (int k = l; k < B4CHUNK * NV; k += 64) s_acc[k] = 0.f;
    if (l < RS) s_rec[B4DUMMY * RS + l] = 0.f;
    const uint8_t* q = s_q + u * B4CHUNK;
    const float* t_row = s_t + b4_t_row(li) + 8 * u;      // the eight shares of the unit's value li (read as four b64)
    float* t_col = s_t + 8 * u + li;                      // this lane's own share of value k goes to t_col[b4_t_row(k)]

    // One (unit, splat) share on its way into the round's LDS totals.  A unit's eight lanes hold the SAME splat, so its NV sums are
    // transposed through LDS (lane i writes its share of values 0..7, reads back the eight shares of value i: two b128 reads and seven
    // adds instead of three DPP levels on every value; values 8.. take the DPP route) and added to the splat's row of s_acc with PLAIN
    // read-modify-write instructions: an LDS float atomic costs ~3 cycles PER LANE on this part (tools/micro/lds_atomic_bench.hip:
    // 193 cycles for 64 lanes, integer atomics 4.5), 216 LDS cycles per iteration and CU for the 72 values.  Two units that hold the
    // same splat in the same iteration must not write together: every unit stamps its number on the splat's tag byte and reads it
    // back; the unit whose stamp survived adds, the others stamp again (a second pass only when units coincide).
    // The whole sequence is one iteration BEHIND the walk (shares written at the end of iteration i are summed and added at the end of
    // iteration i + 1, their LDS reads issued at its top), so that no LDS round trip sits in an iteration's dependency chain.
    auto lds_ld = [](const auto* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_st = [](auto* p, auto v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto settle = [&](int pslot, float m0, float mhi, int tag, float a0, float ahi) {
        uint8_t* tagp = s_tag + min(pslot, B4CHUNK - 1);
        float* accp = s_acc + min(pslot, B4CHUNK - 1) * NV + li;
        bool pending = pslot != B4DUMMY;
        const bool win = pending && tag == u;
        if (win) {
            if (li < NV) lds_st(accp, a0 + m0);
            if (li < NHI) lds_st(accp + 8, ahi + mhi);
        }
        pending = pending && !win;
        if (__builtin_expect(__any(pending), 0)) {
            do {                          // units that coincided on a splat: one more of them gets through per pass
                if (pending && li == 0) lds_st(tagp, (uint8_t)u);
                const int tg = lds_ld(tagp);
                const bool w = pending && tg == u;
                if (w) {
                    if (li < NV) lds_st(accp, lds_ld(accp) + m0);
                    if (li < NHI) lds_st(accp + 8, lds_ld(accp + 8) + mhi);
                }
                pending = pending && !w;
            } while (__any(pending));
        }
    };

    // the staged Gaussian ids are fetched one round ahead, so that a round's gather does not wait for them
    int g_next = (tile_last - 1 - l >= start && l < B4CHUNK) ? flatten_ids[tile_last - 1 - l] : 0;
    for (int hi = tile_last; hi > start; hi -= B4CHUNK) {
        const int lo = max(start, hi - B4CHUNK);
        const int cnt = hi - lo;
        const int g = g_next;
#if GSPL_BWD4_PRIO
        // The launch ends with its longest tile: a wave with much of its list still ahead takes precedence in the SIMD's instruction
        // arbitration (priority, then age) over waves that are nearly done.
        {
            const int rem = hi - start;
            if (rem > 3 * GSPL_BWD4_PRIO) __builtin_amdgcn_s_setprio(3);
            else if (rem > 2 * GSPL_BWD4_PRIO) __builtin_amdgcn_s_setprio(2);
            else if (rem > GSPL_BWD4_PRIO) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        {
            const int i_next = hi - B4CHUNK - 1 - l;
            if (i_next >= start && l < B4CHUNK) g_next = flatten_ids[i_next];
        }
        // ---- stage: lane = staged splat (slot l <-> list index hi - 1 - l: slot 0 is the deepest)
        unsigned m8 = 0u;
        if (l < cnt) {
            s_id[l] = g;
            const float ca = conics[g * 3 + 0], cb = conics[g * 3 + 1], cc = conics[g * 3 + 2], op = opacities[g];
            const float mx = means2d[g * 2 + 0], my = means2d[g * 2 + 1];
            m8 = band_half_mask<4>(mx, my, ca, cb, cc, op, (float)tx + TR::kPixelCentre, (float)ty + TR::kPixelCentre);
            float* rec = s_rec + l * RS;
            *reinterpret_cast<float4*>(rec) = make_float4(mx, my, 0.5f * ca, 0.5f * cc);
            *reinterpret_cast<float4*>(rec + 4) = make_float4(cb, op, (op != 0.f) ? -__builtin_amdgcn_rcpf(op) : 0.f, ca);
            float4 cv = make_float4(0.f, 0.f, 0.f, cc);          // (D = 4: the fourth channel takes the place of c; the walk doubles c/2 then)
            cv.x = colors[(int64_t)g * D + 0];
            if (D > 1) cv.y = colors[(int64_t)g * D + 1];
            if (D > 2) cv.z = colors[(int64_t)g * D + 2];
            if (D > 3) cv.w = colors[(int64_t)g * D + 3];
            *reinterpret_cast<float4*>(rec + 8) = cv;
        }
        // ---- queues: unit uu takes the staged splats that reach its box and lie in front of its deepest last contributor
        const int my_idx = hi - 1 - l;
        int my_qlen = 0, max_q = 0;
#pragma unroll
        for (int uu = 0; uu < 8; ++uu) {
            const bool c = ((m8 >> uu) & 1u) && (my_idx < ulast[uu]);
            const unsigned long long bal = __ballot(c);
            const int n = __builtin_popcountll(bal);
            if (c) s_q[uu * B4CHUNK + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (uint8_t)l;
            my_qlen = (u == uu) ? n : my_qlen;
            max_q = max(max_q, n);
        }
        __syncthreads();

        // ---- walk: eight (unit, splat) pairs per iteration; the record is fetched one iteration ahead, the queue entry two
        int slot = (my_qlen > 0) ? (int)q[0] : B4DUMMY;
        int slot_next = (my_qlen > 1) ? (int)q[1] : B4DUMMY;
        float4 r0 = *reinterpret_cast<const float4*>(s_rec + slot * RS);              // x y a/2 c/2
        float4 r1 = *reinterpret_cast<const float4*>(s_rec + slot * RS + 4);          // b opacity -1/opacity a
        float4 cv = *reinterpret_cast<const float4*>(s_rec + slot * RS + 8);          // colour, c
        int pslot = B4DUMMY;             // the share still on its way into s_acc (previous iteration)
        float phi = 0.f;
#pragma unroll 1
        for (int it = 0; it < max_q; ++it) {
            // the NEXT iteration's record and the queue entry after it
            const float* nrec = s_rec + slot_next * RS;
            const float4 n0 = *reinterpret_cast<const float4*>(nrec);
            const float4 n1 = *reinterpret_cast<const float4*>(nrec + 4);
            const float4 ncv = *reinterpret_cast<const float4*>(nrec + 8);
            const int qn = q[min(it + 2, B4CHUNK - 1)];
            const int slot_after = (it + 2 < my_qlen) ? qn : B4DUMMY;
            // the previous iteration's share: its LDS reads are issued here and consumed at the end of this iteration
            const int pc = min(pslot, B4CHUNK - 1);
            const float2 t0 = *reinterpret_cast<const float2*>(t_row), t1 = *reinterpret_cast<const float2*>(t_row + 2);
            const float2 t2 = *reinterpret_cast<const float2*>(t_row + 4), t3 = *reinterpret_cast<const float2*>(t_row + 6);
            const int ptag = lds_ld(s_tag + pc);
            const float pa0 = lds_ld(s_acc + pc * NV + min(li, NV - 1));
            const float pahi = NHI ? lds_ld(s_acc + pc * NV + 8 + min(li, max(NHI, 1) - 1)) : 0.f;

            const float col[4] = {cv.x, cv.y, cv.z, cv.w};
            const int idx = hi - 1 - slot;
            const float dx = r0.x - pxf;
            const float hadx = r0.z * dx, bdx = r1.x * dx;
            float dy[4], sp[4], fac[4];
            bool any_valid = false;
#if GSPL_BWD4_PK
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k0 = 2 * h, k1 = 2 * h + 1;
                const v2f dy2 = (v2f){r0.y, r0.y} - (v2f){pyf[k0], pyf[k1]};
                // sigma, per element bit-identical to eval_sigma: fma(ha dx, dx, fma(hc dy, dy, (b dx) dy))
                const v2f hcdy2 = (v2f){r0.w, r0.w} * dy2;
                const v2f sigma2 = __builtin_elementwise_fma((v2f){hadx, hadx}, (v2f){dx, dx},
                                                             __builtin_elementwise_fma(hcdy2, dy2, (v2f){bdx, bdx} * dy2));
                const v2f arg2 = sigma2 * (v2f){-1.4426950408889634f, -1.4426950408889634f};
                const v2f vis2 = {__builtin_amdgcn_exp2f(arg2.x), __builtin_amdgcn_exp2f(arg2.y)};
                const v2f raw2 = (v2f){r1.y, r1.y} * vis2;
                // alpha = min(kAlphaMax, raw) >= 1/255  <=>  raw >= 1/255
                const bool v0 = (idx < last[k0]) && (sigma2.x >= 0.f) && (raw2.x >= kAlphaMin);
                const bool v1 = (idx < last[k1]) && (sigma2.y >= 0.f) && (raw2.y >= kAlphaMin);
                any_valid = any_valid || v0 || v1;
                const v2f rv2 = {v0 ? raw2.x : 0.f, v1 ? raw2.y : 0.f};
                // rv >= 0: the median of (rv, 0, alpha_max) is min(alpha_max, rv) in ONE instruction
                const v2f a2 = {__builtin_amdgcn_fmed3f(rv2.x, 0.f, TR::kAlphaMax), __builtin_amdgcn_fmed3f(rv2.y, 0.f, TR::kAlphaMax)};
                v2f rw2 = rv2;                 // o * vis where the pixel takes a gradient through alpha, else 0
                if (TR::kClampKillsGrad) rw2 = (v2f){(rv2.x <= TR::kAlphaMax) ? rv2.x : 0.f, (rv2.y <= TR::kAlphaMax) ? rv2.y : 0.f};
                const v2f om2 = (v2f){1.f, 1.f} - a2;
                const v2f ra2 = {__builtin_amdgcn_rcpf(om2.x), __builtin_amdgcn_rcpf(om2.y)};
                v2f T2 = (v2f){T[k0], T[k1]} * ra2;              // transmittance in front of this splat
                const v2f fac2 = a2 * T2;
                v2f cdot2 = (v2f){col[0], col[0]} * (v2f){vo[0][k0], vo[0][k1]};
#pragma unroll
                for (int c = 1; c < D; ++c) cdot2 = __builtin_elementwise_fma((v2f){col[c], col[c]}, (v2f){vo[c][k0], vo[c][k1]}, cdot2);
                v2f R2 = {R[k0], R[k1]};
                const v2f v_alpha2 = __builtin_elementwise_fma(cdot2, T2, R2 * ra2);
                R2 = __builtin_elementwise_fma(-cdot2, fac2, R2);
                const v2f sp2 = -rw2 * v_alpha2;
                T[k0] = T2.x; T[k1] = T2.y; R[k0] = R2.x; R[k1] = R2.y;
                dy[k0] = dy2.x; dy[k1] = dy2.y; sp[k0] = sp2.x; sp[k1] = sp2.y; fac[k0] = fac2.x; fac[k1] = fac2.y;
            }
#else
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // sigma, bit-identical to eval_sigma: fma(ha dx, dx, fma(hc dy, dy, (b dx) dy))
                dy[k] = r0.y - pyf[k];
                const float hcdy = r0.w * dy[k];
                const float sigma = fmaf(hadx, dx, fmaf(hcdy, dy[k], bdx * dy[k]));
                const float vis = __builtin_amdgcn_exp2f(sigma * -1.4426950408889634f);
                const float raw = r1.y * vis;
                // alpha = min(kAlphaMax, raw) >= 1/255  <=>  raw >= 1/255
                const bool valid = (idx < last[k]) && (sigma >= 0.f) && (raw >= kAlphaMin);
                any_valid = any_valid || valid;
                const float rv = valid ? raw : 0.f;
                // rv >= 0: the median of (rv, 0, alpha_max) is min(alpha_max, rv) in ONE instruction (fminf costs a canonicalising
                // v_max in front of the v_min because the select above hides that rv is already quiet)
                const float a = __builtin_amdgcn_fmed3f(rv, 0.f, TR::kAlphaMax);
                float rw = rv;                 // o * vis where the pixel takes a gradient through alpha, else 0
                if (TR::kClampKillsGrad) rw = (rv <= TR::kAlphaMax) ? rv : 0.f;
                const float ra = __builtin_amdgcn_rcpf(1.f - a);
                T[k] *= ra;                    // transmittance in front of this splat
                fac[k] = a * T[k];
                float cdot = col[0] * vo[0][k];
#pragma unroll
                for (int c = 1; c < D; ++c) cdot = fmaf(col[c], vo[c][k], cdot);
                const float v_alpha = fmaf(cdot, T[k], R[k] * ra);
                R[k] = fmaf(-cdot, fac[k], R[k]);
                sp[k] = -rw * v_alpha;
            }
#endif
            // some pixel takes this splat (has_hit_any_pixels): tagged in LDS with a fire-and-forget ds_or, reported at the flush
            if (hit_flags && any_valid) atomicOr(&s_id[slot], (int)0x80000000);

            // ---- this lane's share of the splat's gradients (column of four pixels; dx constant, dy per row)
            float vals[NV];
            const float tq0 = sp[0] * dy[0], tq1 = sp[1] * dy[1], tq2 = sp[2] * dy[2], tq3 = sp[3] * dy[3];
            const float S0 = (sp[0] + sp[1]) + (sp[2] + sp[3]);
            const float Sy = (tq0 + tq1) + (tq2 + tq3);
            const float Syy = fmaf(tq0, dy[0], tq1 * dy[1]) + fmaf(tq2, dy[2], tq3 * dy[3]);
            const float Sx = S0 * dx;
            const float ca = r1.w, cb = r1.x, cc = (D > 3) ? 2.f * r0.w : cv.w;
            vals[0] = fmaf(ca, Sx, cb * Sy);                                // dL/dx
            vals[1] = fmaf(cb, Sx, cc * Sy);                                // dL/dy
            vals[2] = 0.5f * (Sx * dx);                                     // dL/da
            vals[3] = Sy * dx;                                              // dL/db
            vals[4] = 0.5f * Syy;                                           // dL/dc
            vals[5] = S0 * r1.z;                                            // dL/dopacity = sum(vis * v_alpha) = -sum(sp) / o
#pragma unroll
            for (int c = 0; c < D; ++c)
                vals[6 + c] = fmaf(fac[0], vo[c][0], fac[1] * vo[c][1]) + fmaf(fac[2], vo[c][2], fac[3] * vo[c][3]);
            if constexpr (ABS) {
                float ax = 0.f, ay = 0.f;
                const float cax = ca * dx, cbx = cb * dx;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ax += fabsf(sp[k] * fmaf(cb, dy[k], cax));
                    ay += fabsf(sp[k] * fmaf(cc, dy[k], cbx));
                }
                vals[6 + D] = ax; vals[7 + D] = ay;
            }

            // ---- the previous iteration's share goes into the round's totals ...
            settle(pslot, ((t0.x + t0.y) + (t1.x + t1.y)) + ((t2.x + t2.y) + (t3.x + t3.y)), phi, ptag, pa0, pahi);
            // ---- ... and this iteration's share sets out: values 0..7 through the transposition buffer, the rest over three DPP levels
#pragma unroll
            for (int k = 0; k < 8 && k < NV; ++k) t_col[b4_t_row(k)] = vals[k];
            if constexpr (NHI > 0) {
                float hi_vals[NHI];
#pragma unroll
                for (int k = 0; k < NHI; ++k) hi_vals[k] = vals[8 + k];
                quad_xor1_add<NHI>(hi_vals);
                quad_xor2_add<NHI>(hi_vals);
                half_mirror_add<NHI>(hi_vals);
                phi = hi_vals[0];
#pragma unroll
                for (int k = 1; k < NHI; ++k) phi = (li == k) ? hi_vals[k] : phi;
            }
            if (li == 0 && slot != B4DUMMY) lds_st(s_tag + slot, (uint8_t)u);
            pslot = slot;
            slot = slot_next; slot_next = slot_after;
            r0 = n0; r1 = n1; cv = ncv;
        }
        if (max_q > 0) {        // the last iteration's share
            const int pc = min(pslot, B4CHUNK - 1);
            const float2 t0 = *reinterpret_cast<const float2*>(t_row), t1 = *reinterpret_cast<const float2*>(t_row + 2);
            const float2 t2 = *reinterpret_cast<const float2*>(t_row + 4), t3 = *reinterpret_cast<const float2*>(t_row + 6);
            const int ptag = lds_ld(s_tag + pc);
            const float pa0 = lds_ld(s_acc + pc * NV + min(li, NV - 1));
            const float pahi = NHI ? lds_ld(s_acc + pc * NV + 8 + min(li, max(NHI, 1) - 1)) : 0.f;
            settle(pslot, ((t0.x + t0.y) + (t1.x + t1.y)) + ((t2.x + t2.y) + (t3.x + t3.y)), phi, ptag, pa0, pahi);
        }
        __syncthreads();

        // ---- flush: one fp32 L2 atomic per value per (tile, splat)
        if constexpr (PACKED) {
            float* __restrict__ v_packed = v_means2d;
            for (int e = l; e < cnt * NV; e += 64) {
                const float v = s_acc[e];
                s_acc[e] = 0.f;
                const int row = e / NV;
                if (v != 0.f) atomicAdd(&v_packed[(int64_t)(s_id[row] & 0x7fffffff) * packed_stride + (e - row * NV)], v);
            }
        } else if (l < cnt) {
            const int gid = s_id[l] & 0x7fffffff;
            float v[NV];
            bool any_nz = false;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                v[k] = s_acc[l * NV + k];
                s_acc[l * NV + k] = 0.f;
                any_nz = any_nz || (v[k] != 0.f);
            }
            if (any_nz) {
                atomicAdd(&v_means2d[gid * 2 + 0], v[0]);
                atomicAdd(&v_means2d[gid * 2 + 1], v[1]);
                atomicAdd(&v_conics[gid * 3 + 0], v[2]);
                atomicAdd(&v_conics[gid * 3 + 1], v[3]);
                atomicAdd(&v_conics[gid * 3 + 2], v[4]);
                atomicAdd(&v_opacities[gid], v[5]);
#pragma unroll
                for (int c = 0; c < D; ++c) atomicAdd(&v_colors[(int64_t)gid * D + c], v[6 + c]);
                if constexpr (ABS) {
                    atomicAdd(&v_means2d_abs[gid * 2 + 0], v[6 + D]);
                    atomicAdd(&v_means2d_abs[gid * 2 + 1], v[7 + D]);
                }
            }
        }
        if (hit_flags && l < cnt && s_id[l] < 0) hit_flags[s_id[l] & 0x7fffffff] = 1;      // one store per (tile, splat) that was composited
        __syncthreads();
    }
}

// Longest lists first.  A tile is ONE wave's serial walk, so the launch cannot end before its longest tile does — and a long tile that
// starts in the second wave of workgroups ends long after everything else (measured: 3.0 of 5 waves resident on average with the
// tiles in image order).  Workgroup b takes tile order[b], the tiles bucket-sorted by list length, longest first: the saturated
// tiles of the dense image regions start together at t = 0 and the short lists of the sparse regions fill the tail.
// Scratch: a rotating set of per-process device arrays (the C-ABI has no workspace argument for the backward): at most
// B4_ORDER_SLOTS backward launches of this kernel may be in flight at once in one process; more than 65536 tiles keep image order.
static constexpr int B4_ORDER_SLOTS = 8;
static constexpr int B4_ORDER_MAX_TILES = 1 << 16;
__device__ int32_t g_b4_order[B4_ORDER_SLOTS][B4_ORDER_MAX_TILES];

__global__ __launch_bounds__(1024) void composite_bwd4_order_kernel(const int32_t* __restrict__ offsets, int n_tiles, int64_t n_isects,
                                                                     int32_t* __restrict__ order) {
    __shared__ int s_max;
    __shared__ int s_hist[256], s_cur[256];
    const int t = threadIdx.x;
    if (t == 0) s_max = 0;
    if (t < 256) s_hist[t] = 0;
    __syncthreads();
    auto length = [&](int k) {
        int a, b;
        tile_range(k, n_tiles, n_isects, offsets, a, b);
        return b - a;
    };
    int mx = 0;
    for (int k = t; k < n_tiles; k += 1024) mx = max(mx, length(k));
    atomicMax(&s_max, mx);
    __syncthreads();
    const float scale = 256.f / (float)(s_max + 1);
    for (int k = t; k < n_tiles; k += 1024) atomicAdd(&s_hist[255 - min(255, (int)((float)length(k) * scale))], 1);
    __syncthreads();
    if (t == 0) { int run = 0; for (int b = 0; b < 256; ++b) { s_cur[b] = run; run += s_hist[b]; } }
    __syncthreads();
    for (int k = t; k < n_tiles; k += 1024) order[atomicAdd(&s_cur[255 - min(255, (int)((float)length(k) * scale))], 1)] = k;
}

// Which backward kernel serves D <= 4: 4 (default) or 2 (GSPL_BWD_KERNEL=2 in the environment: A/B runs and bisecting).
static int bwd_kernel_choice() {
    static const int choice = [] {
        const char* e = std::getenv("GSPL_BWD_KERNEL");
        return (e && std::strcmp(e, "2") == 0) ? 2 : 4;
    }();
    return choice;
}

template <int D, int MODE, bool CHW, bool PACKED = false>
static int launch_bwd(bool absgrad, int n_tiles, int tile_w, int width, int height, int64_t n_isects,
                      const float* means2d, const float* conics, const float* colors, const float* opacities,
                      const float* backgrounds, const int32_t* offsets, const int32_t* flatten_ids,
                      const float* final_Ts, const int32_t* last_ids,
                      const float* v_out_colors, const float* v_out_alphas,
                      float* v_means2d, float* v_means2d_abs, float* v_conics, float* v_colors, float* v_opacities,
                      hipStream_t s, int packed_stride = 0, uint8_t* hit_flags = nullptr) {
#define GSPL_BWD_ARGS n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, \
                      final_Ts, last_ids, v_out_colors, v_out_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities,   \
                      packed_stride, hit_flags
    if constexpr (D <= 4) {
        if (bwd_kernel_choice() == 4) {
            int32_t* order = nullptr;
            if (n_tiles <= B4_ORDER_MAX_TILES && n_tiles > 1024) {
                static std::atomic<unsigned> next_slot{0};
                static int32_t* base = [] { void* p = nullptr; return hipGetSymbolAddress(&p, HIP_SYMBOL(g_b4_order)) == hipSuccess ? (int32_t*)p : nullptr; }();
                if (base) {
                    order = base + (size_t)(next_slot.fetch_add(1) % B4_ORDER_SLOTS) * B4_ORDER_MAX_TILES;
                    hipLaunchKernelGGL(composite_bwd4_order_kernel, dim3(1), dim3(1024), 0, s, offsets, n_tiles, n_isects, order);
                }
            }
            if (absgrad) hipLaunchKernelGGL((composite_bwd4_kernel<D, MODE, CHW, true, PACKED>), dim3(n_tiles), dim3(64), 0, s, GSPL_BWD_ARGS, (const int32_t*)order);
            else hipLaunchKernelGGL((composite_bwd4_kernel<D, MODE, CHW, false, PACKED>), dim3(n_tiles), dim3(64), 0, s, GSPL_BWD_ARGS, (const int32_t*)order);
            return check_launch("composite_bwd");
        }
    }
    if (absgrad) hipLaunchKernelGGL((composite_bwd2_kernel<D, MODE, CHW, true, PACKED>), dim3(n_tiles), dim3(B2_NT), 0, s, GSPL_BWD_ARGS);
    else hipLaunchKernelGGL((composite_bwd2_kernel<D, MODE, CHW, false, PACKED>), dim3(n_tiles), dim3(B2_NT), 0, s, GSPL_BWD_ARGS);
#undef GSPL_BWD_ARGS
    return check_launch("composite_bwd");
}

}  // namespace gspl

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
    int rc = check_composite_args(N, n_isects, D, mode, layout, width, height, tile_size, tile_w, tile_h, "composite_bwd: bad argument");
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
    int rc = check_composite_args(N, n_isects, D, mode, layout, width, height, tile_size, tile_w, tile_h, "composite_bwd_packed: bad argument");
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

// Name of the kernel template gspl_composite_bwd / gspl_composite_bwd_packed launch for D <= 4 in this process (profile look-ups).
extern "C" const char* gspl_composite_bwd_kernel_name(void) {
    return gspl::bwd_kernel_choice() == 4 ? "composite_bwd4_kernel" : "composite_bwd2_kernel";
}
