// gspl_composite.h — device helpers shared by the compositing kernels (composite.hip: forward + per-splat statistics;
// composite_bwd.hip: backward).  gfx950, wave64.
//
// The compositing rule restated here is the published 3DGS rule with the per-API constants of SURVEY.md Appendix B (ModeTraits
// in gspl_device.h); call sites replaced: internal/renderers/gsplat_v1_renderer.py:588-601 (`rasterize_to_pixels`),
// gsplat_renderer.py:86-99 (`rasterize_gaussians`), vanilla_renderer.py:111-120 (Inria `GaussianRasterizer`).
#pragma once
#include "gspl_device.h"
#include "gspl_host.h"

namespace gspl {

static constexpr int TILE = 16;

typedef float v2f __attribute__((ext_vector_type(2)));

// sigma = 0.5 (a dx^2 + c dy^2) + b dx dy, evaluated in the FACTORED form
//     sigma = ha u^2 + hd dy^2,   u = dx + k dy,   ha = a / 2,  k = b / a,  hd = (c - b^2 / a) / 2 = det / (2 a).
// Why (round 5): the three-term form cancels.  A long anisotropic splat (a trained scene's "needles": footprints of hundreds of
// pixels, a c / det ~ 1e4) has a dx^2, c dy^2 and 2 b dx dy ~ 1e5-1e6 each for a sum of ~5 on its ridge, and fp32 then carries an
// absolute error of 1e-3 ... 4e-2 in sigma — 0.1 ... 4 % in alpha, far outside the 2e-5 band in which the oracle calls a 1/255
// decision fragile (tests/test_locked_parity.py on `synthetic.scene_surfaces`: robust pixels off by one flipped splat, 2.8e-3).
// Both terms of the factored form are products of the same sign; what is left is the rounding of u (1e-7 x the distance to the
// centre): measured worst |d sigma| over the 3000 largest splats of that scene 1.5e-3 -> 1e-5.  Same instruction count per pixel
// pair.  The forward and the backward evaluate bit-identical values (their skip / stop decisions must agree): explicit fma, one
// definition, and the per-splat factors come from `sigma_coef` in both.
struct SigmaCoef { float ha, k, hd; };
// (conic_det: gspl_device.h — a c - b^2 without the cancellation of its two products)
__device__ __forceinline__ SigmaCoef sigma_coef(float a, float b, float c) {
    SigmaCoef s;
    s.ha = 0.5f * a;
    if (a != 0.f) {
        // k: a correctly rounded division — its rounding, times the distance to the centre, is what is left of sigma's error.  hd only
        // scales the (positive, <= tau) second term: the 1-ulp hardware reciprocal is plenty
        s.k = b / a;
        s.hd = (0.5f * conic_det(a, b, c)) * __builtin_amdgcn_rcpf(a);
    } else {
        // a == 0: sigma = b dx dy + c dy^2 / 2.  With b == 0 that IS the factored form (k = 0, hd = c / 2); with b != 0 the matrix is
        // indefinite (det = -b^2 < 0) — not a Gaussian: such a splat is never composited (sigma = NaN fails every `sigma >= 0`)
        s.k = 0.f;
        s.hd = (b == 0.f) ? 0.5f * c : __builtin_nanf("");
    }
    return s;
}
__device__ __forceinline__ float eval_sigma(float ha, float k, float hd, float dx, float dy) {
    const float u = fmaf(k, dy, dx);
    return fmaf(ha * u, u, (hd * dy) * dy);
}

// Two splats (or two pixels) at once with packed fp32 math; each component is bit-identical to eval_sigma.
__device__ __forceinline__ v2f eval_sigma2(v2f ha, v2f k, v2f hd, v2f dx, v2f dy) {
    const v2f u = __builtin_elementwise_fma(k, dy, dx);
    return __builtin_elementwise_fma(ha * u, u, (hd * dy) * dy);
}

// Exact test "can this splat reach alpha >= 1/255 at some pixel centre of the box [x0,x1] x [y0,y1]" (continuous box,
// conservative margins).  alpha >= 1/255  <=>  sigma <= tau = ln(255 o); the x-span of (ellipse 1/2 d^T Q d <= tau) intersected
// with the band dy in [y0-my, y1-my] is [left, right] with right = hx if the ellipse's rightmost point lies in the band, else
// the larger chord end at the band edges (see binning.hip, row_span); the box is reachable iff that span meets [x0, x1].
// The margins absorb the 1-ulp hardware rcp / sqrt and the fp32 rounding of exp / log: a pair the exact per-pixel test would keep
// is never culled; candidates still go through the exact per-pixel test.
__device__ __forceinline__ bool box_reachable(float mx, float my, float a, float b, float c, float opacity,
                                              float x0, float x1, float y0, float y1) {
    const float tau = __logf(255.f * opacity) * 1.0002f + 2e-4f;
    if (!(tau > 0.f)) return false;
    const float det = conic_det(a, b, c);
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

// box_reachable for a GRID of boxes of one tile at once: BANDS horizontal bands of 16 / BANDS pixel rows, each cut into a left and
// a right half of 8 pixel columns.  Bit (2 * band + half) of the result = that 8 x (16 / BANDS) box is reachable.  The per-splat
// terms (tau, extents, the tangent offset) are shared, the chord ends are per band.  (tx0, ty0) is the centre of the tile's first
// pixel.  BANDS = 2: the four 8x8 quadrants; BANDS = 4: eight 8x4 units.
template <int BANDS>
__device__ __forceinline__ unsigned band_half_mask(float mx, float my, float a, float b, float c, float opacity, float tx0, float ty0) {
    constexpr float BH = (float)(TILE / BANDS);
    const float tau = __logf(255.f * opacity) * 1.0002f + 2e-4f;
    if (!(tau > 0.f)) return 0u;
    const float det = conic_det(a, b, c);
    if (!(det > 0.f) || !(a > 0.f) || !(c > 0.f)) return (1u << (2 * BANDS)) - 1u;      // not an ellipse: never cull
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
    for (int band = 0; band < BANDS; ++band) {
        const float y0 = ty0 + BH * (float)band;
        float lo = y0 - my, hi = (y0 + (BH - 1.f)) - my;
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

// ---- segmented backward (round 5) -------------------------------------------------------------------------------------------------
// The backward walks a tile's list back to front with one workgroup: in a scene whose lists are heavy-tailed (a trained model: a few
// tiles walk thousands of entries where the median walks a few hundred) the launch lasts as long as its longest tile
// (`synthetic.scene_surfaces`: 0.57 ms against ~0.40 for the same work spread evenly).  The forward therefore leaves a CHECKPOINT of
// every pixel's running state each SEG list entries, and the backward cuts a walk longer than SEG into segments that ANY workgroup
// may process: a pixel that continues beyond a segment's far end starts from the checkpoint there (its transmittance in front of
// that entry, and the colour accumulated from that entry on) instead of from its final state.
//   forward   accumulates the colour PER SEGMENT (the sum restarts at every boundary; the image is the sum of the segment sums) and,
//             when it is done, turns the stored segment sums into suffix sums from the back — small deep contributions are added to
//             each other first, as the backward's own running sum does (a difference "final - prefix" would cancel).
//             ckpt[(p >> SEG_LOG2) * 256 + quadrant * 64 + lane] = (T in front of entry p, colour accumulated from entry p on),
//             p = tile start + k SEG, k >= 1: boundaries of different tiles are more than SEG apart (each lies at least SEG behind
//             its tile's start), so p >> SEG_LOG2 is a collision-free slot index.  It also clears the frame's item counter.
//   backward  two launches: the first is the plain kernel cut to segment 0 of every tile (a walk of at most SEG entries: all of it);
//             a tile with a longer walk PUBLISHES its segments 1.. as work items, and the second launch — one workgroup per item
//             slot (list capacity / SEG: what the host knows) — serves them.  No workgroup waits for another; nothing depends on
//             dispatch order.
//   ADAPTIVE  a frame is segmented only when ITS VIEW's walks have a tail.  Every backward workgroup adds its tile's walk length to a small
//             table of device words (sum, maximum, non-empty tiles; 64 rows by tile index, fire-and-forget atomics); the NEXT frame's
//             forward kernel — behind that backward on the stream, so the table is complete — reduces it, stores a verdict into the slot of
//             a small ring of pinned host words it was given (ticket << 1 | tail; tail = the longest walk exceeds SEG_TRIGGER entries AND
//             4.5 x the mean walk of the non-empty tiles, SEG_TAIL_X2 / 2) and clears the table.  The host — no synchronisation — reads the
//             slots that have landed before a later forward and remembers the verdict PER VIEW (the address of the view matrix the
//             previous frame was rendered with: a data set's cameras are persistent tensors); a frame is segmented when its view's last
//             verdict was "tail", a view without a verdict follows the stream (segmented for 64 frames after any "tail").
//             Measured per view (profiles/r23_walk_tail_probe.txt): with a longest walk above 768 the plain kernel wins by 44-61 us up
//             to a ratio of 3.3, the segmented form by 43-960 us from 5.6 on — every view of a trained-scene-shaped frame, and ONE far
//             view of the metric workload's heterogeneous set (-131 us), which a per-stream switch could not take without dragging the
//             fifteen views that lose 14-61 us along.  (Round 5's trigger was the length alone: close-up views of the uniform cloud walk
//             800-1000 entries in EVERY tile — no tail, nothing to spread — and paid 38 us for checkpoints and a second launch.)
// D == 3 only (a checkpoint is one float4); off (ckpt == nullptr) everywhere else.
#ifndef GSPL_SEG_LOG2
#define GSPL_SEG_LOG2 8      // 256 entries per segment (measured on scene_surfaces, per step: 64 -> 1.63 ms, 128 -> 1.45, 256 -> 1.47, 512 -> 1.56, 1024 -> 1.72;
                             // 128 costs the forward 4 % more for its checkpoints and twice the checkpoint memory: profiles/r07v_segment_sizes.txt)
#endif
static constexpr int SEG_LOG2 = GSPL_SEG_LOG2;
static constexpr int SEG = 1 << SEG_LOG2;
static constexpr int SEG_TRIGGER = 3 * SEG;    // a walk longer than this (and SEG_TAIL x the mean walk) switches the segmented form on (a tile of two segments is not worth a second launch)
static constexpr int SEG_TAIL_X2 = 9;          // ... the launch lasts as long as its longest tile only when that tile is far longer than the bulk: 2 x longest > 9 x mean
static constexpr int SEG_WALK_SLOTS = 64;      // rows of (sum, max, non-empty tiles, -) in SegState::walk
static constexpr int SEG_MAX = 255;            // segments per tile (8 bits in a work item); the last one takes whatever is left
struct SegState {
    float4* ckpt;
    uint32_t* words;      // [0]: published work items (zero before the backward); [2 .. 2 + slots): the items, (tile << 8) | segment
    uint32_t* host_flag;  // pinned host word (a slot of the verdict ring): receives (ticket << 1) | "the last backward's walks had a tail" (nullable)
    uint32_t ticket;      // ... of this forward
    uint32_t* walk;       // device, [SEG_WALK_SLOTS][4]: the backward's walk statistics, reduced and cleared by the next forward (nullable)
    uint4* zero_p;        // forward only: a block this launch clears on behalf of the backward (its packed rows: GSPL_BUF_PACKED), 16-byte units
    uint32_t zero_n16;
    uint32_t slots;
    __host__ __device__ uint32_t* count() const { return words; }
    __host__ __device__ uint32_t* work() const { return words + 2; }
};

__device__ __forceinline__ void tile_range(int tile, int n_tiles, int64_t n_isects,
                                           const int32_t* __restrict__ offsets, int& start, int& end) {
    start = offsets[tile];
    // n_isects < 0: `offsets` has n_tiles + 1 entries, the last one is the list length (device-side count, gspl_bin_sort_device_count)
    end = (tile + 1 < n_tiles || n_isects < 0) ? offsets[tile + 1] : (int)n_isects;
}

// The tile lists may be cut on 8-, 16- or 32-pixel tiles (`tile_size` of the callers: gsplat_v1_renderer.py:23-41); the kernels
// always work on 8x8 pixel blocks grouped into 16x16 compute tiles.  The list of a block is the list of the LIST tile that holds it.
struct ListTiles {
    int log2;      // list tile side = 1 << log2 (3, 4, 5)
    int w;         // list tiles per row
    int n;         // number of list tiles
};
static inline ListTiles list_tiles(int tile_size, int tile_w, int tile_h) {
    ListTiles lt;
    lt.log2 = tile_size == 8 ? 3 : (tile_size == 32 ? 5 : 4);
    lt.w = tile_w;
    lt.n = tile_w * tile_h;
    return lt;
}
__device__ __forceinline__ void block_list_range(const ListTiles& lt, int bx, int by, int width, int height, int64_t n_isects,
                                                 const int32_t* __restrict__ offsets, int& start, int& end) {
    if (bx * 8 >= width || by * 8 >= height) { start = end = 0; return; }      // (a block of the compute grid outside the image)
    const int t = (by >> (lt.log2 - 3)) * lt.w + (bx >> (lt.log2 - 3));
    start = offsets[t];
    end = (t + 1 < lt.n || n_isects < 0) ? offsets[t + 1] : (int)n_isects;
}

// number of per-splat gradient values accumulated per (tile, splat): xy(2) conic(3) opacity(1) colour(D) [+abs xy(2)]
template <int D, bool ABS> struct BwdVals { static constexpr int N = 6 + D + (ABS ? 2 : 0); };

int check_composite_args(int N, int64_t n_isects, int D, int mode, int layout, int width, int height,
                         int tile_size, int tile_w, int tile_h, const char* who);

// gspl_composite_fwd / gspl_composite_bwd_packed with the segmentation state of the fused Inria call (fused.hip); seg == NULL: off
int composite_fwd_impl(int N, int64_t n_isects, int D, int mode, int layout, const float* means2d, const float* conics, const float* colors,
                       const float* opacities, const float* backgrounds, int width, int height, int tile_size, int tile_w, int tile_h,
                       const int32_t* offsets, const int32_t* flatten_ids, float* out_colors, float* out_alphas, float* final_Ts, int32_t* last_ids,
                       uint8_t* hit_flags, void* stream, const SegState* seg);
int composite_bwd_packed_impl(int N, int64_t n_isects, int D, int mode, int layout, const float* means2d, const float* conics, const float* colors,
                              const float* opacities, const float* backgrounds, int width, int height, int tile_size, int tile_w, int tile_h,
                              const int32_t* offsets, const int32_t* flatten_ids, const float* final_Ts, const int32_t* last_ids,
                              const float* v_out_colors, const float* v_out_alphas, float* v_packed, int packed_stride, int absgrad, uint8_t* hit_flags,
                              void* stream, const SegState* seg);

}  // namespace gspl

#define GSPL_DISPATCH_D(D_, MODE_, CHW_, CALL)                    \
    switch (D_) {                                                 \
        case 1: { constexpr int kD = 1; CALL(kD, MODE_, CHW_); } break; \
        case 2: { constexpr int kD = 2; CALL(kD, MODE_, CHW_); } break; \
        case 3: { constexpr int kD = 3; CALL(kD, MODE_, CHW_); } break; \
        case 4: { constexpr int kD = 4; CALL(kD, MODE_, CHW_); } break; \
        case 8: { constexpr int kD = 8; CALL(kD, MODE_, CHW_); } break; \
    }
