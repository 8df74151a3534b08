// composite_bwd.hip — 16x16-tile alpha compositing, BACKWARD (gfx950, wave64).
//
// Autograd backward of gsplat `rasterize_to_pixels` (internal/renderers/gsplat_v1_renderer.py:588-601), v0 `rasterize_gaussians`
// (gsplat_renderer.py:86-99) and the render stage of the Inria `GaussianRasterizer` (vanilla_renderer.py:111-120), which the
// reference enters through `manual_backward` (gaussian_splatting.py:380).  Neither CUDA package is vendored in the reference; the
// algorithm restated here is the published 3DGS compositing rule with the per-API constants of SURVEY.md Appendix B (ModeTraits).
//
// One kernel, composite_bwd2_kernel: two waves per tile, two pixels per lane, slab-transposed per-splat reduction; ONE fp32 L2 atomic
// per value per (tile, splat): 36 B per intersection, the algorithmic minimum of SURVEY.md §8d.  (Round 3 built and measured a
// one-wave-per-tile variant with per-unit candidate queues: tools/experiments/composite_bwd4.inc)
// Roofline: algorithmic bytes 76*I + 20*P (+8*I with absgrad); the kernels are VALU-bound under that model.
#include "gspl_composite.h"

namespace gspl {

// Staged record of composite_bwd2_kernel in LDS (floats): x y a/2 k | hd opacity quadrant-mask b | colour[D] c/2 (padded to a
// multiple of 4): one address register per candidate, the fields are fetched with immediate offsets (b128 + b64 [+ colour]).
// (a/2, k, hd) = sigma_coef: what the walk evaluates sigma with; b and c/2 are what phase 2 turns the moments into gradients with.
template <int D> struct BwdRec { static constexpr int STRIDE = 8 + ((D + 1 + 3) & ~3); static constexpr int HC = 8 + D; };

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


// SEGM (gspl_composite.h, SegState) — 0: the plain walk, one workgroup per tile, whole list (it leaves its walk lengths for the next
// forward, which decides whether the following frames run segmented).  1: the workgroup walks segment 0 of its tile — the whole walk when it
// is at most SEG entries — and publishes the other segments of a longer walk as work items.  2: the launch behind it: one workgroup
// per item slot, the published items are served, the other workgroups leave at once.
template <int D, int MODE, bool CHW, bool ABS, bool PACKED, int SEGM = 0>
__global__ __launch_bounds__(B2_NT, GSPL_BWD2_WAVES) void composite_bwd2_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
    const float* __restrict__ final_Ts, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_out_colors, const float* __restrict__ v_out_alphas,
    float* __restrict__ v_means2d, float* __restrict__ v_means2d_abs,
    float* __restrict__ v_conics, float* __restrict__ v_colors, float* __restrict__ v_opacities, int packed_stride,
    uint8_t* __restrict__ hit_flags, SegState seg) {
    using TR = ModeTraits<MODE>;
    constexpr int NV = BwdVals<D, ABS>::N;
    constexpr int RS = BwdRec<D>::STRIDE;
    constexpr bool VO_REGS = D <= 4;                  // dL/dout of the lane's phase-2 column lives in registers
    constexpr int SLAB = 2 * P2_SLOTS * 128;          // floats per wave: fac plane, sp plane, [slot][column*8 + row]
    static_assert(NV <= 16, "one ds_add round per batch");
    __shared__ int s_id[B2CHUNK];
    __shared__ __attribute__((aligned(16))) float s_rec[B2CHUNK * RS];
    // per-(tile, splat) totals of the round, shared by the two waves (ds_add_f32).  Measured alternative (round 3): one copy per wave
    // with plain read-modify-write — an LDS float atomic costs ~3 LDS cycles PER LANE on this part (tools/micro/lds_atomic_bench.hip)
    // — is 1 % slower here: 48 lane-atomics per four candidates are not what binds this kernel, the second copy costs occupancy.
    __shared__ float s_acc[B2CHUNK * NV];
    __shared__ __attribute__((aligned(16))) float s_slab[B2_NW * SLAB];
    __shared__ __attribute__((aligned(16))) float s_vo_keep[VO_REGS ? 4 : B2_NW * 128 * D];
    static_assert(!VO_REGS || B2_NW * 128 * D <= B2_NW * SLAB, "s_vo alias too small");
    float* s_vo = VO_REGS ? s_slab : s_vo_keep;       // [wave][column 0..15][channel][row]
    __shared__ int s_last;

    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int ws = w;
    // (SEGM == 2: one workgroup per work-item SLOT — the host knows how many there can be, not how many there are; the surplus
    // leaves at once.  A loop over items instead keeps every kernel argument alive across the walk: 106 scalar registers, spills.)
    int tile = 0, segidx = 0;
    if constexpr (SEGM == 2) {
        if (blockIdx.x >= min(*seg.count(), seg.slots)) return;
        const uint32_t item = seg.work()[blockIdx.x];
        tile = (int)(item >> 8); segidx = (int)(item & 255u);
    } else {
        tile = xcd_remap(blockIdx.x, n_tiles);
    }
    {
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
    // the list range [seg_lo, seg_hi) this workgroup walks: the whole walk [start, block_last), or one segment of a long one
    int seg_lo = start, seg_hi = block_last;
    if constexpr (SEGM != 2) {
        // this tile's walk, for the next forward's "does the frame have a tail" (gspl_composite.h, ADAPTIVE): fire-and-forget atomics
        // on one of 64 rows
        if (seg.walk && t == 0 && block_last > start) {
            uint32_t* row = seg.walk + (size_t)((unsigned)tile & (SEG_WALK_SLOTS - 1)) * 4u;
            atomicAdd(row + 0, (uint32_t)(block_last - start));
            atomicMax(row + 1, (uint32_t)(block_last - start));
            atomicAdd(row + 2, 1u);
        }
    }
    if constexpr (SEGM != 0) {
        const int walked = block_last - start;
        const int nseg = walked > 0 ? min((walked + SEG - 1) >> SEG_LOG2, SEG_MAX) : 0;
        if constexpr (SEGM == 1) {
            if (nseg > 1 && t == 0) {
                // segments 1.. of this walk: work items of the launch that follows (plain stores: read after this kernel has ended)
                const uint32_t pos = atomicAdd(seg.count(), (uint32_t)(nseg - 1));
                for (int sgm = 1; sgm < nseg; ++sgm)
                    if (pos + (uint32_t)(sgm - 1) < seg.slots) seg.work()[pos + (uint32_t)(sgm - 1)] = ((uint32_t)tile << 8) | (uint32_t)sgm;
            }
        }
        if (nseg > 1) {
            seg_lo = start + segidx * SEG;
            seg_hi = (segidx + 1 < nseg) ? start + (segidx + 1) * SEG : block_last;
            if (seg_hi < block_last) {
                // a pixel that goes on beyond the far end of the segment starts from the forward's checkpoint there: T in front of
                // entry seg_hi, and R = T_final (v_alpha - bg.vo) - vo . (colour accumulated behind that entry)
                const size_t slot = (size_t)((unsigned)seg_hi >> SEG_LOG2) * 256u;
                const int lasts[2] = {lastA, lastB};
                float Tn[2] = {T2.x, T2.y}, Rn[2] = {R2.x, R2.y};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (lasts[e] > seg_hi) {
                        const float4 ck = seg.ckpt[slot + (unsigned)((2 * w + e) * 64 + l)];
                        const float ckc[3] = {ck.y, ck.z, ck.w};
                        float behind = 0.f;
#pragma unroll
                        for (int c = 0; c < D; ++c) behind = fmaf(e == 0 ? vo[c].x : vo[c].y, ckc[c < 3 ? c : 0], behind);
                        Tn[e] = ck.x;
                        Rn[e] = Rn[e] - behind;
                    }
                }
                T2 = (v2f){Tn[0], Tn[1]}; R2 = (v2f){Rn[0], Rn[1]};
            }
        } else if (segidx > 0) {
            seg_lo = seg_hi = start;      // (a work item for a walk that is not long: nothing to do — cannot happen, the forward lists none)
        }
    }

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
            const float4 r0 = *reinterpret_cast<const float4*>(rec);          // x y a/2 k
            const float4 r1 = *reinterpret_cast<const float4*>(rec + 4);      // hd opacity mask b
            ca = 2.f * r0.z; cb = r1.w; cc_ = 2.f * rec[BwdRec<D>::HC]; co_ = r1.y;
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
    int g_next = (seg_hi - 1 - t >= seg_lo && t < B2CHUNK) ? flatten_ids[seg_hi - 1 - t] : 0;
    for (int hi = seg_hi; hi > seg_lo; hi -= B2CHUNK) {
        const int lo = max(seg_lo, hi - B2CHUNK);
        const int cnt = hi - lo;
        const int g = g_next;
        {
            const int i_next = hi - B2CHUNK - 1 - t;
            if (i_next >= seg_lo && t < B2CHUNK) g_next = flatten_ids[i_next];
        }
        if (t < cnt) {
            s_id[t] = g;
            const float ca = conics[g * 3 + 0], cb = conics[g * 3 + 1], cc = conics[g * 3 + 2], op = opacities[g];
            const float mx = means2d[g * 2 + 0], my = means2d[g * 2 + 1];
            const unsigned qm = band_half_mask<2>(mx, my, ca, cb, cc, op, (float)tx + TR::kPixelCentre, (float)ty + TR::kPixelCentre);
            float* rec = s_rec + t * RS;
            const SigmaCoef sc = sigma_coef(ca, cb, cc);
            *reinterpret_cast<float4*>(rec) = make_float4(mx, my, sc.ha, sc.k);
            *reinterpret_cast<float4*>(rec + 4) = make_float4(sc.hd, op, __uint_as_float(qm), cb);
#pragma unroll
            for (int c = 0; c < D; ++c) rec[8 + c] = colors[(int64_t)g * D + c];
            rec[BwdRec<D>::HC] = 0.5f * cc;
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
                    const float4 r0 = *reinterpret_cast<const float4*>(rec);          // x y a/2 k
                    const float2 r1 = *reinterpret_cast<const float2*>(rec + 4);      // hd opacity
                    float col[D];                                                    // fetched with the record: one LDS round trip per candidate
#pragma unroll
                    for (int c = 0; c < D; ++c) col[c] = rec[8 + c];
                    // sigma, bit-identical per element to eval_sigma: u = fma(k, dy, dx); fma(ha u, u, (hd dy) dy) — dy, and with it the
                    // second term, is shared by the lane's two pixels (same row)
                    const v2f dx2 = (v2f){r0.x, r0.x} - pxf2;
                    const float dy = r0.y - pyf;
                    const float e = (r1.x * dy) * dy;
                    const v2f u2 = __builtin_elementwise_fma((v2f){r0.w, r0.w}, (v2f){dy, dy}, dx2);
                    const v2f sigma2 = __builtin_elementwise_fma((v2f){r0.z, r0.z} * u2, u2, (v2f){e, e});
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
            // packed_stride < 0 (gspl_set_deterministic): one row per LIST ENTRY instead of one per splat — the staged slot `row` is
            // list position hi - 1 - row, a single (tile, splat) pair, so every row has one writer and a later pass adds a splat's
            // rows in list order (ordered_reduce_kernel below)
            const bool by_entry = packed_stride < 0;
            const int ps = by_entry ? -packed_stride : packed_stride;
            for (int e = t; e < cnt * NV; e += B2_NT) {
                const float v = s_acc[e];
                s_acc[e] = 0.f;
                const int row = e / NV;
                const int64_t dst_row = by_entry ? (int64_t)(hi - 1 - row) : (int64_t)(s_id[row] & 0x7fffffff);
                if (v != 0.f) atomicAdd(&v_packed[dst_row * ps + (e - row * NV)], v);
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
}

// ---------------------------------------------------------------------------------------------------------------
// Backward for tile lists cut on 8- or 32-pixel tiles (`tile_size` 8 / 32 of the callers, gsplat_v1_renderer.py:23-41): ONE WAVE
// PER 8x8 BLOCK, one pixel per lane, the block walks the list of the list tile that holds it (gspl_composite.h, ListTiles) back
// to front.  Per round of 64 list entries the lanes cull their entry against the block (box_reachable) and compact the candidates
// with ballot + mbcnt (ascending lane = descending list index); per candidate every lane evaluates its pixel (the same expression
// tree as composite_bwd2_kernel), the NV values are summed over the wave with DPP and lane 63 issues the L2 atomics.  The same
// arithmetic and the same discrete decisions as the 16-pixel path; a compatibility path, correct before fast (the default tile size
// of every configuration of the reference is 16).
template <int D, int MODE, bool CHW, bool ABS, bool PACKED>
__global__ __launch_bounds__(64) void composite_bwd_block_kernel(
    int n_tiles, int tile_w, int width, int height, int64_t n_isects,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids,
    const float* __restrict__ final_Ts, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_out_colors, const float* __restrict__ v_out_alphas,
    float* __restrict__ v_means2d, float* __restrict__ v_means2d_abs,
    float* __restrict__ v_conics, float* __restrict__ v_colors, float* __restrict__ v_opacities, int packed_stride,
    uint8_t* __restrict__ hit_flags, ListTiles lt) {
    using TR = ModeTraits<MODE>;
    constexpr int NV = BwdVals<D, ABS>::N;
    __shared__ float s_x[64], s_y[64], s_ha[64], s_b[64], s_hc[64], s_op[64], s_k[64], s_hd[64];
    __shared__ float s_col[64 * D];
    __shared__ int s_g[64], s_idx[64];

    const int unit = xcd_remap(blockIdx.x, 4 * n_tiles, 4 * GSPL_XCD_RUN);
    const int tile = unit >> 2, w = unit & 3, l = threadIdx.x;
    const int bx = (tile % tile_w) * 2 + (w & 1), by = (tile / tile_w) * 2 + (w >> 1);
    const int px = bx * 8 + (l & 7), py = by * 8 + (l >> 3);
    const bool inside = (px < width) && (py < height);
    const float pxf = (float)px + TR::kPixelCentre, pyf = (float)py + TR::kPixelCentre;
    const float qx0 = (float)(bx * 8) + TR::kPixelCentre, qy0 = (float)(by * 8) + TR::kPixelCentre;
    int start, end;
    block_list_range(lt, bx, by, width, height, n_isects, offsets, start, end);

    const int64_t pix = (int64_t)py * width + px;
    const int last = inside ? last_ids[pix] : start;
    float T = inside ? final_Ts[pix] : 1.f;
    float vo[D];
    float bgdot = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        vo[c] = inside ? (CHW ? v_out_colors[(int64_t)c * width * height + pix] : v_out_colors[pix * D + c]) : 0.f;
        if (backgrounds) bgdot += backgrounds[c] * vo[c];
    }
    float R = T * (((inside && v_out_alphas) ? v_out_alphas[pix] : 0.f) - bgdot);
    int block_last = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) block_last = max(block_last, __shfl_xor(block_last, off));

    for (int hi = block_last; hi > start; hi -= 64) {
        const int idx = hi - 1 - l;                 // lane 0 = the deepest entry of the round
        bool cand = false;
        int g = 0;
        float mx = 0.f, my = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, op = 0.f;
        if (idx >= start) {
            g = flatten_ids[idx];
            ca = conics[g * 3 + 0]; cb = conics[g * 3 + 1]; cc = conics[g * 3 + 2]; op = opacities[g];
            mx = means2d[g * 2 + 0]; my = means2d[g * 2 + 1];
            cand = box_reachable(mx, my, ca, cb, cc, op, qx0, qx0 + 7.f, qy0, qy0 + 7.f);
        }
        const unsigned long long mask = __ballot(cand);
        const int ncand = __builtin_popcountll(mask);
        const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        __syncthreads();
        if (cand) {
            const SigmaCoef sc = sigma_coef(ca, cb, cc);
            s_x[slot] = mx; s_y[slot] = my; s_ha[slot] = sc.ha; s_b[slot] = cb; s_hc[slot] = 0.5f * cc; s_op[slot] = op;
            s_k[slot] = sc.k; s_hd[slot] = sc.hd;
            s_g[slot] = g; s_idx[slot] = idx;
#pragma unroll
            for (int c = 0; c < D; ++c) s_col[slot * D + c] = colors[(int64_t)g * D + c];
        }
        __syncthreads();
        for (int k = 0; k < ncand; ++k) {
            const int kidx = s_idx[k];
            const float ha = s_ha[k], b = s_b[k], hc = s_hc[k], o = s_op[k];
            const float dx = s_x[k] - pxf, dy = s_y[k] - pyf;
            const float sigma = eval_sigma(ha, s_k[k], s_hd[k], dx, dy);
            const float vis = __builtin_amdgcn_exp2f(sigma * -1.4426950408889634f);
            const float raw = o * vis;
            const bool valid = (kidx < last) && (sigma >= 0.f) && (raw >= kAlphaMin);
            if (!__any(valid)) continue;
            const float rv = valid ? raw : 0.f;
            const float a = fminf(TR::kAlphaMax, rv);
            float rw = rv;
            if (TR::kClampKillsGrad) rw = (rv <= TR::kAlphaMax) ? rv : 0.f;
            const float ra = __builtin_amdgcn_rcpf(1.f - a);
            T *= ra;
            const float fac = a * T;
            float cdot = s_col[k * D] * vo[0];
#pragma unroll
            for (int c = 1; c < D; ++c) cdot = fmaf(s_col[k * D + c], vo[c], cdot);
            const float v_alpha = fmaf(cdot, T, R * ra);
            R = fmaf(-cdot, fac, R);
            const float sp = -rw * v_alpha;
            const float fa = 2.f * ha, fc = 2.f * hc;
            float vals[NV];
            vals[0] = sp * fmaf(fa, dx, b * dy);
            vals[1] = sp * fmaf(b, dx, fc * dy);
            vals[2] = 0.5f * sp * dx * dx;
            vals[3] = sp * dx * dy;
            vals[4] = 0.5f * sp * dy * dy;
            vals[5] = (o != 0.f) ? -sp * __builtin_amdgcn_rcpf(o) : 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) vals[6 + c] = fac * vo[c];
            if constexpr (ABS) { vals[6 + D] = fabsf(vals[0]); vals[7 + D] = fabsf(vals[1]); }
#pragma unroll
            for (int v = 0; v < NV; ++v) vals[v] = wave_sum_to_lane63(vals[v]);
            if (l == 63) {
                const int gk = s_g[k];
                if constexpr (PACKED) {
                    float* row = v_means2d + (int64_t)gk * packed_stride;
#pragma unroll
                    for (int v = 0; v < NV; ++v) if (vals[v] != 0.f) atomicAdd(&row[v], vals[v]);
                } else {
                    atomicAdd(&v_means2d[gk * 2 + 0], vals[0]);
                    atomicAdd(&v_means2d[gk * 2 + 1], vals[1]);
                    atomicAdd(&v_conics[gk * 3 + 0], vals[2]);
                    atomicAdd(&v_conics[gk * 3 + 1], vals[3]);
                    atomicAdd(&v_conics[gk * 3 + 2], vals[4]);
                    atomicAdd(&v_opacities[gk], vals[5]);
#pragma unroll
                    for (int c = 0; c < D; ++c) atomicAdd(&v_colors[(int64_t)gk * D + c], vals[6 + c]);
                    if constexpr (ABS) {
                        atomicAdd(&v_means2d_abs[gk * 2 + 0], vals[6 + D]);
                        atomicAdd(&v_means2d_abs[gk * 2 + 1], vals[7 + D]);
                    }
                }
                if (hit_flags) hit_flags[gk] = 1;
            }
        }
    }
}

template <int D, int MODE, bool CHW, bool PACKED = false>
static int launch_bwd(bool absgrad, int n_tiles, int tile_w, int width, int height, int64_t n_isects,
                      const float* means2d, const float* conics, const float* colors, const float* opacities,
                      const float* backgrounds, const int32_t* offsets, const int32_t* flatten_ids,
                      const float* final_Ts, const int32_t* last_ids,
                      const float* v_out_colors, const float* v_out_alphas,
                      float* v_means2d, float* v_means2d_abs, float* v_conics, float* v_colors, float* v_opacities,
                      hipStream_t s, int packed_stride, uint8_t* hit_flags, ListTiles lt, const SegState* seg_in = nullptr) {
#define GSPL_BWD_ARGS n_tiles, tile_w, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, \
                      final_Ts, last_ids, v_out_colors, v_out_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities,   \
                      packed_stride, hit_flags
    if (lt.log2 != 4) {       // lists on 8- or 32-pixel tiles: the block-wise compatibility kernel
        if (absgrad) hipLaunchKernelGGL((composite_bwd_block_kernel<D, MODE, CHW, true, PACKED>), dim3(4 * n_tiles), dim3(64), 0, s, GSPL_BWD_ARGS, lt);
        else hipLaunchKernelGGL((composite_bwd_block_kernel<D, MODE, CHW, false, PACKED>), dim3(4 * n_tiles), dim3(64), 0, s, GSPL_BWD_ARGS, lt);
        return check_launch("composite_bwd");
    }
    SegState plain = {};
    if (seg_in) { plain.host_flag = seg_in->host_flag; plain.walk = seg_in->walk; }
    if constexpr (D == 3 && CHW && PACKED) {
        // the segmented form (the fused Inria call with checkpoints from its forward): regular gradients only — the deterministic mode
        // and absgrad keep the one-workgroup-per-tile walk
        if (seg_in && seg_in->ckpt && !absgrad && packed_stride > 0) {
            hipLaunchKernelGGL((composite_bwd2_kernel<D, MODE, CHW, false, PACKED, 1>), dim3(n_tiles), dim3(B2_NT), 0, s, GSPL_BWD_ARGS, *seg_in);
            hipLaunchKernelGGL((composite_bwd2_kernel<D, MODE, CHW, false, PACKED, 2>), dim3(seg_in->slots), dim3(B2_NT), 0, s, GSPL_BWD_ARGS, *seg_in);
            return check_launch("composite_bwd(segmented)");
        }
    }
    if (absgrad) hipLaunchKernelGGL((composite_bwd2_kernel<D, MODE, CHW, true, PACKED>), dim3(n_tiles), dim3(B2_NT), 0, s, GSPL_BWD_ARGS, plain);
    else hipLaunchKernelGGL((composite_bwd2_kernel<D, MODE, CHW, false, PACKED>), dim3(n_tiles), dim3(B2_NT), 0, s, GSPL_BWD_ARGS, plain);
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
    const ListTiles lt = list_tiles(tile_size, tile_w, tile_h);
    const int ctw = (width + TILE - 1) / TILE, n_tiles = ctw * ((height + TILE - 1) / TILE);      // 16x16 compute tiles
    hipStream_t s = (hipStream_t)stream;
    const bool absgrad = v_means2d_abs != nullptr;
    rc = GSPL_ERR_UNSUPPORTED;
#define CALL_BWD(kD, M, C) rc = launch_bwd<kD, M, C>(absgrad, n_tiles, ctw, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, s, 0, hit_flags, lt)
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

// ---- deterministic (debug) mode ----------------------------------------------------------------------------------------------
// gspl_set_deterministic(1): gspl_composite_bwd_packed delivers bit-reproducible gradients.  The regular launch adds every
// (tile, splat) total to the splat's row with an fp32 L2 atomic, so the ORDER of a splat's additions follows the dispatch order of
// its tiles: run-to-run differences of 1e-7 relative, which tests of HIP against HIP could only bound (1e-4 after the projection
// chain, profiles/r04_flaky_v_means.txt).  In this mode the kernel writes one row per list entry (single writer), the entries
// are sorted by splat id (stable: ties in list order) and one thread per splat adds its rows in that order.  Three extra passes over
// the entries and scratch from hipMallocAsync: a mode for tests and debugging, not for the timed path.
#include "gspl_sort.h"
namespace gspl {
static int g_deterministic = 0;

__global__ __launch_bounds__(256) void iota_ids_kernel(int64_t n, const int32_t* __restrict__ ids, uint32_t* __restrict__ keys, uint32_t* __restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { keys[i] = (uint32_t)ids[i]; pos[i] = (uint32_t)i; }
}
// thread i = the first entry of a run of equal ids (sorted, stable): adds the run's rows in list order into the splat's row
__global__ __launch_bounds__(256) void ordered_reduce_kernel(int64_t n, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ pos,
                                                             const float* __restrict__ entries, int nv, int entry_stride,
                                                             float* __restrict__ v_packed, int packed_stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = keys[i];
    if (i > 0 && keys[i - 1] == g) return;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int64_t j = i; j < n && keys[j] == g; ++j) {
        const float* row = entries + (int64_t)pos[j] * entry_stride;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < nv) acc[k] += row[k];
    }
    float* out = v_packed + (int64_t)g * packed_stride;
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < nv) out[k] += acc[k];      // (the row was zero, or holds what the caller put there: one writer)
}
}  // namespace gspl

extern "C" int gspl_set_deterministic(int on) { const int was = gspl::g_deterministic; gspl::g_deterministic = on ? 1 : 0; return was; }
extern "C" int gspl_get_deterministic(void) { return gspl::g_deterministic; }

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
    return gspl::composite_bwd_packed_impl(N, n_isects, D, mode, layout, means2d, conics, colors, opacities, backgrounds, width, height, tile_size, tile_w,
                                           tile_h, offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas, v_packed, packed_stride, absgrad,
                                           hit_flags, stream, nullptr);
}

int gspl::composite_bwd_packed_impl(int N, int64_t n_isects, int D, int mode, int layout,
                                         const float* means2d, const float* conics, const float* colors,
                                         const float* opacities, const float* backgrounds,
                                         int width, int height, int tile_size, int tile_w, int tile_h,
                                         const int32_t* offsets, const int32_t* flatten_ids,
                                         const float* final_Ts, const int32_t* last_ids,
                                         const float* v_out_colors, const float* v_out_alphas,
                                         float* v_packed, int packed_stride, int absgrad, uint8_t* hit_flags, void* stream, const SegState* seg) {
    using namespace gspl;
    int rc = check_composite_args(N, n_isects, D, mode, layout, width, height, tile_size, tile_w, tile_h, "composite_bwd_packed: bad argument");
    if (rc != GSPL_OK) return rc;
    if (packed_stride < 6 + D + (absgrad ? 2 : 0)) return fail_arg("composite_bwd_packed: packed_stride smaller than the row");
    if (n_isects == 0 || N == 0) return GSPL_OK;
    if (!means2d || !conics || !colors || !opacities || !offsets || !flatten_ids || !final_Ts || !last_ids || !v_out_colors || !v_packed)
        return fail_arg("composite_bwd_packed: NULL required pointer");
    const ListTiles lt = list_tiles(tile_size, tile_w, tile_h);
    const int ctw = (width + TILE - 1) / TILE, n_tiles = ctw * ((height + TILE - 1) / TILE);      // 16x16 compute tiles
    hipStream_t s = (hipStream_t)stream;
    const bool ag = absgrad != 0;
    // deterministic mode (see gspl_set_deterministic below): rows per list entry, then an ordered reduction per splat
    const int nv = 6 + D + (ag ? 2 : 0);
    const bool ordered = gspl_get_deterministic() != 0 && lt.log2 == 4 && n_isects > 0;
    float* entries = nullptr;
    uint32_t *k0 = nullptr, *k1 = nullptr, *p0 = nullptr, *p1 = nullptr;
    void* sort_ws = nullptr;
    size_t sort_ws_bytes = 0;
    int id_bits = 1;
    float* const v_packed_out = v_packed;
    const int packed_stride_out = packed_stride;
    if (ordered) {
        if (n_isects < 0) return fail_arg("composite_bwd_packed: the deterministic mode needs the list length on the host (n_isects >= 0)");
        while (id_bits < 32 && (1ll << id_bits) < (long long)N) ++id_bits;
        sort_ws_bytes = gspl_radix_sort_workspace_bytes(n_isects, 4, 0, id_bits);
        hipError_t e = hipMallocAsync((void**)&entries, (size_t)n_isects * nv * sizeof(float), s);
        if (e == hipSuccess) e = hipMallocAsync((void**)&k0, (size_t)n_isects * 4 * sizeof(uint32_t), s);
        if (e == hipSuccess) e = hipMallocAsync(&sort_ws, sort_ws_bytes ? sort_ws_bytes : 16, s);
        if (e == hipSuccess) e = hipMemsetAsync(entries, 0, (size_t)n_isects * nv * sizeof(float), s);
        if (e != hipSuccess) return check_hip(e, "composite_bwd_packed(deterministic): scratch");
        k1 = k0 + n_isects; p0 = k1 + n_isects; p1 = p0 + n_isects;
        v_packed = entries;
        packed_stride = -nv;
    }
    rc = GSPL_ERR_UNSUPPORTED;
    if (ordered) seg = nullptr;      // (rows per list entry, one writer each: the plain walk)
#define CALL_BWDP(kD, M, C) rc = launch_bwd<kD, M, C, true>(ag, n_tiles, ctw, width, height, n_isects, means2d, conics, colors, opacities, backgrounds, offsets, flatten_ids, final_Ts, last_ids, v_out_colors, v_out_alphas, v_packed, nullptr, nullptr, nullptr, nullptr, s, packed_stride, hit_flags, lt, seg)
    if (mode == GSPL_MODE_GSPLAT) {
        if (layout == GSPL_LAYOUT_HWC) { GSPL_DISPATCH_D(D, GSPL_MODE_GSPLAT, false, CALL_BWDP) }
        else { GSPL_DISPATCH_D(D, GSPL_MODE_GSPLAT, true, CALL_BWDP) }
    } else {
        if (layout == GSPL_LAYOUT_HWC) { GSPL_DISPATCH_D(D, GSPL_MODE_INRIA, false, CALL_BWDP) }
        else { GSPL_DISPATCH_D(D, GSPL_MODE_INRIA, true, CALL_BWDP) }
    }
#undef CALL_BWDP
    if (ordered) {
        if (rc == GSPL_OK) {
            hipLaunchKernelGGL(iota_ids_kernel, dim3((unsigned)((n_isects + 255) / 256)), dim3(256), 0, s, n_isects, flatten_ids, k0, p0);
            int which = 0;
            rc = gspl_radix_sort_pairs_u32(n_isects, k0, k1, p0, p1, 0, id_bits, &which, sort_ws, sort_ws_bytes, s);
            if (rc == GSPL_OK) {
                hipLaunchKernelGGL(ordered_reduce_kernel, dim3((unsigned)((n_isects + 255) / 256)), dim3(256), 0, s, n_isects, which ? k1 : k0, which ? p1 : p0,
                                   entries, nv, nv, v_packed_out, packed_stride_out);
                rc = check_launch("composite_bwd_packed(ordered reduce)");
            }
        }
        (void)hipFreeAsync(entries, s); (void)hipFreeAsync(k0, s); (void)hipFreeAsync(sort_ws, s);
    }
    return rc;
}

// Name of the kernel template gspl_composite_bwd / gspl_composite_bwd_packed launch (profile look-ups).
extern "C" const char* gspl_composite_bwd_kernel_name(void) { return "composite_bwd2_kernel"; }
