// binning.hip — tile binning: per-Gaussian tile counts, prefix sum, (tile|depth) key emission,
// stable LSD radix sort, per-tile ranges (gfx950).
//
// Replaces gsplat `isect_tiles` + `isect_offset_encode` (internal/renderers/gsplat_v1_renderer.py:446-458)
// and the binning half of v0 `rasterize_gaussians` / Inria `GaussianRasterizer`
// (gsplat_renderer.py:86-99, vanilla_renderer.py:111-120).  Key layout is the one the reference
// pins in Python: internal/utils/gaussian_projection.py:159-208
//     key = (tile_id << 32) | bits(depth as f32),  tile_id = y * tile_w + x,  value = Gaussian id,
// emitted per Gaussian in row-major tile order, so that after a *stable* sort equal-depth ties
// stay in Gaussian-id order.
//
// Roofline: HBM-bound.  Emit: 16 B read per visible Gaussian + 12 B written per intersection.
// Sort: 12 B x I x 2 x ceil(bits/8) with bits = 32 + ceil(log2(tiles)).
// Scans and sorts are the in-tree kernels of sort.hip (gspl_sort.h: count -> scatter passes, block-sum scans; no workgroup
// waits for another) on every path and at every size; only the significant key bits are sorted.
#include <cstring>
#include <cstdlib>
#include "gspl_device.h"
#include "gspl_host.h"
#include "gspl_sort.h"
#include "gspl_sort_device.h"

namespace gspl {

// tile rectangle of one splat, per API convention
template <int MODE>
__device__ __forceinline__ void tile_rect(float x, float y, int radius, int tile_size, int tile_w, int tile_h,
                                          int& minx, int& miny, int& maxx, int& maxy) {
    const float ts = (float)tile_size;
    const float r = (float)radius;
    if (MODE == GSPL_MODE_GSPLAT) {
        // gaussian_projection.py:117-125 : trunc((p - r)/T), trunc((p + r)/T) + 1
        minx = (int)((x - r) / ts); miny = (int)((y - r) / ts);
        maxx = (int)((x + r) / ts) + 1; maxy = (int)((y + r) / ts) + 1;
    } else {
        // Inria getRect: (int)((p - r)/T), (int)((p + r + T - 1)/T)
        minx = (int)((x - r) / ts); miny = (int)((y - r) / ts);
        maxx = (int)((x + r + ts - 1.f) / ts); maxy = (int)((y + r + ts - 1.f) / ts);
    }
    minx = min(max(minx, 0), tile_w); maxx = min(max(maxx, 0), tile_w);
    miny = min(max(miny, 0), tile_h); maxy = min(max(maxy, 0), tile_h);
}

template <int MODE>
__global__ __launch_bounds__(256) void isect_count_kernel(
    int N, const float* __restrict__ means2d, const int32_t* __restrict__ radii,
    int tile_size, int tile_w, int tile_h, int32_t* __restrict__ tiles_per_gauss, int32_t* __restrict__ counts32) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    int n = 0;
    const int radius = radii[g];
    if (radius > 0) {
        int minx, miny, maxx, maxy;
        tile_rect<MODE>(means2d[g * 2 + 0], means2d[g * 2 + 1], radius, tile_size, tile_w, tile_h, minx, miny, maxx, maxy);
        n = max(maxx - minx, 0) * max(maxy - miny, 0);
    }
    tiles_per_gauss[g] = n;
    counts32[g] = n;
}

template <int MODE>
__global__ __launch_bounds__(256) void isect_emit_kernel(
    int N, const float* __restrict__ means2d, const int32_t* __restrict__ radii, const float* __restrict__ depths,
    const int64_t* __restrict__ cum_tiles, int tile_size, int tile_w, int tile_h,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const int radius = radii[g];
    if (radius <= 0) return;
    int minx, miny, maxx, maxy;
    tile_rect<MODE>(means2d[g * 2 + 0], means2d[g * 2 + 1], radius, tile_size, tile_w, tile_h, minx, miny, maxx, maxy);
    int64_t off = (g == 0) ? 0 : cum_tiles[g - 1];
    const uint64_t depth_bits = (uint64_t)__float_as_uint(depths[g]);
    for (int ty = miny; ty < maxy; ++ty) {
        for (int tx = minx; tx < maxx; ++tx) {
            const uint64_t tile = (uint64_t)(ty * tile_w + tx);
            keys[off] = (tile << 32) | depth_bits;
            vals[off] = (uint32_t)g;
            ++off;
        }
    }
}

// offsets[t] = first index i with tile(keys[i]) >= t ; offsets has tile_w*tile_h entries.
__global__ __launch_bounds__(256) void isect_offsets_kernel(
    int64_t n_isects, const int64_t* __restrict__ keys, int n_tiles, int32_t* __restrict__ offsets) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_isects) return;
    const int cur = (int)((uint64_t)keys[i] >> 32);
    if (i == 0) {
        for (int t = 0; t <= cur && t < n_tiles; ++t) offsets[t] = 0;
    } else {
        const int prev = (int)((uint64_t)keys[i - 1] >> 32);
        for (int t = prev + 1; t <= cur && t < n_tiles; ++t) offsets[t] = (int32_t)i;
    }
    if (i == n_isects - 1) {
        for (int t = cur + 1; t < n_tiles; ++t) offsets[t] = (int32_t)n_isects;
    }
}

__global__ void fill_i32_kernel(int n, int32_t v, int32_t* __restrict__ p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline int key_bits(int n_tiles) {
    int b = 0;
    while ((1ll << b) < (long long)n_tiles) ++b;
    return 32 + b;
}

// 64-bit keyed path (isect_tiles API): counts (i32) | scan states | keys + values scratch | two sort plans (the key has up to
// 32 + log2(tiles) significant bits: sorted as [0, 24) then [24, bits), three passes each at most)
static constexpr int ISECT_MAX_KEY_BITS = 56;      // 32 depth bits + up to 2^24 tiles
struct IsectWorkspace {
    size_t counts_off, scan_off, keys_off, vals_off, sort_off, sort_bytes;
    size_t total_count, total;
    RadixPlan lo, hi;
    bool two_sorts;
};

static int plan_workspace(int N, int64_t n_isects, int key_bits_total, IsectWorkspace& w) {
    const size_t n = (size_t)(N > 0 ? N : 1), ni = (size_t)(n_isects > 0 ? n_isects : 1);
    if (ni > RADIX_MAX_ITEMS || n > RADIX_MAX_ITEMS) { set_error("isect", "more than 2^30-1 items"); return GSPL_ERR_UNSUPPORTED; }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.counts_off = take(4 * n);
    w.scan_off = take(scan_workspace_bytes(n));
    w.total_count = off;
    w.keys_off = take(8 * ni);
    w.vals_off = take(4 * ni);
    const int split = key_bits_total > 32 ? 24 : 0;
    w.two_sorts = key_bits_total > 32;
    bool ok = true;
    if (w.two_sorts) ok = radix_plan(ni, 0, split, 8, RADIX_TILE_U64, w.lo) && radix_plan(ni, split, key_bits_total, 8, RADIX_TILE_U64, w.hi);
    else ok = radix_plan(ni, 0, key_bits_total, 8, RADIX_TILE_U64, w.lo);
    if (!ok) { set_error("isect", "key bits not representable"); return GSPL_ERR_UNSUPPORTED; }
    w.sort_bytes = w.lo.total_bytes > (w.two_sorts ? w.hi.total_bytes : 0) ? w.lo.total_bytes : w.hi.total_bytes;
    w.sort_off = take(w.sort_bytes);
    w.total = off;
    return GSPL_OK;
}


// =================================================================================================
// Two-level binning ("depth first"): sort the N splats by depth ONCE (32-bit keys), emit their tile
// hits in that order, then a stable radix sort on the tile id alone (ceil(log2 tiles) bits, 2 passes)
// puts every tile's list in depth order.  Ordering is identical to the single 64-bit (tile|depth)
// sort — both are stable, ties end in Gaussian-id order — but the traffic per intersection drops
// from 12 B x 2 x 6 passes to 8 B x 2 x 2 passes (+ 8 B x 2 x 4 passes per *splat*).
// =================================================================================================
// Exact "which tiles of this tile ROW can the splat reach with alpha >= 1/255" test.
// The region alpha >= 1/255 is the ellipse  1/2 d^T Q d <= tau  (d = p - mu, Q = conic, tau = ln(255 opacity)).
// For a horizontal band dy in [lo, hi] (the pixel centres of one tile row) the ellipse's chord at height dy is
//     dx in [(-b dy - sqrt(D))/a, (-b dy + sqrt(D))/a],  D = 2 tau a - det dy^2,
// the right end is concave in dy with its maximum hx = sqrt(2 tau c / det) at dy = -dys, the left end convex with its
// minimum -hx at dy = +dys  (dys = b sqrt(2 tau / (det c))).  The ellipse-band intersection is convex, so a tile of the
// row is reachable iff its pixel-centre span meets the x-projection [xl, xr] of that intersection: the test is exact
// and costs two square roots per tile ROW instead of a test per tile.  tau and the span are inflated by small margins so
// that fp32 rounding can never drop a pair the compositing kernels' per-pixel test would keep: images and gradients
// are unchanged by the culling.
struct SplatCull {
    float mx, my, a, b, det, two_tau_a, inv_a, hx, hy, dys;
    int kind;    // 0: never contributes, 1: ellipse test, 2: never cull (degenerate conic)
};
__device__ __forceinline__ SplatCull make_cull(float mx, float my, float a, float b, float c, float opacity) {
    SplatCull s;
    s.mx = mx; s.my = my; s.a = a; s.b = b;
    const float tau = __logf(255.f * opacity);
    s.det = conic_det(a, b, c);      // (without the cancellation of a c - b b: the spans of long anisotropic splats hang on it)
    if (!(tau > 0.f)) { s.kind = 0; return s; }
    if (!(s.det > 0.f) || !(a > 0.f) || !(c > 0.f)) { s.kind = 2; return s; }
    s.kind = 1;
    const float two_tau = 2.f * (tau * 1.001f + 1e-3f);
    s.two_tau_a = two_tau * a;
    // hardware rcp / sqrt (1 ulp) instead of the ~10-instruction IEEE expansions: the margins absorb the difference,
    // and the count and emit kernels call the very same code, so their per-splat tile counts agree bit for bit
    const float rdet = __builtin_amdgcn_rcpf(s.det);
    s.inv_a = __builtin_amdgcn_rcpf(a);
    s.hx = __builtin_amdgcn_sqrtf(two_tau * c * rdet) * 1.0002f;
    s.hy = __builtin_amdgcn_sqrtf(two_tau * a * rdet) * 1.0002f;
    s.dys = b * __builtin_amdgcn_sqrtf(two_tau * rdet * __builtin_amdgcn_rcpf(c));
    return s;
}
// x-span (absolute pixel coordinates) reachable inside the band y in [y0, y1]; returns false when empty.
__device__ __forceinline__ bool row_span(const SplatCull& s, float y0, float y1, float& xl, float& xr) {
    float lo = y0 - s.my, hi = y1 - s.my;
    if (hi < -s.hy || lo > s.hy) return false;
    lo = fmaxf(lo, -s.hy); hi = fminf(hi, s.hy);
    const float rlo = __builtin_amdgcn_sqrtf(fmaxf(0.f, s.two_tau_a - s.det * lo * lo));
    const float rhi = __builtin_amdgcn_sqrtf(fmaxf(0.f, s.two_tau_a - s.det * hi * hi));
    const float right = (-s.dys >= lo && -s.dys <= hi) ? s.hx : fmaxf((-s.b * lo + rlo) * s.inv_a, (-s.b * hi + rhi) * s.inv_a);
    const float left = (s.dys >= lo && s.dys <= hi) ? -s.hx : fminf((-s.b * lo - rlo) * s.inv_a, (-s.b * hi - rhi) * s.inv_a);
    const float eps = 2e-3f + 3e-4f * s.hx;
    xl = s.mx + left - eps;
    xr = s.mx + right + eps;
    return true;
}
// reachable tile columns [c0, c1) of tile row ty inside the rect columns [minx, maxx)
template <int MODE>
__device__ __forceinline__ void row_columns(const SplatCull& s, int ty, int tile_size, int minx, int maxx, int& c0, int& c1) {
    if (s.kind == 2) { c0 = minx; c1 = maxx; return; }
    c0 = c1 = minx;
    if (s.kind == 0) return;
    const float off = MODE == GSPL_MODE_GSPLAT ? 0.5f : 0.f;
    const float ts = (float)tile_size, span = (float)(tile_size - 1);
    const float y0 = (float)(ty * tile_size) + off;
    float xl, xr;
    if (!row_span(s, y0, y0 + span, xl, xr)) return;
    // tile tx covers pixel centres [tx*ts + off, tx*ts + off + span]
    const float rts = 1.f / ts;      // tile sizes are powers of two: exact
    c0 = max(minx, (int)ceilf((xl - off - span) * rts));
    c1 = min(maxx, (int)floorf((xr - off) * rts) + 1);
    if (c1 < c0) c1 = c0;
}

// Per-splat span record written by bin_keys_kernel (a coalesced pass in memory order) and gathered by the emit kernel
// (which walks the splats in DEPTH order, i.e. at random addresses): one 32-byte line per splat instead of four
// scattered reads (means2d, radii, conics, opacities) plus the span arithmetic all over again.
struct __attribute__((aligned(16))) SpanRecord {
    uint16_t miny;              // first tile row
    uint16_t rows;              // number of tile rows; SPAN_BIG: not representable (more than EMIT_ROWS rows or a row wider than 255)
    uint16_t c0[8];             // first reachable tile column of each row
    uint8_t n[8];               // reachable tiles of each row
    uint32_t pad;
};
static_assert(sizeof(SpanRecord) == 32, "span record is one 32-byte line");
// Rows 8..15 of splats taller than 8 tile rows live in a second 32-byte record (words: c0[8..15] as 4 words, n[8..15] as 2
// words), in a second array after the N primary records: written and read only for those splats, so the common case
// stays at one line per splat while close-up splats up to 16 tile rows (radius ~128 px) keep the load-balanced path.
static constexpr int EMIT_ROWS = 16;
static constexpr uint16_t SPAN_BIG = 0xFFFFu;

// HEADER: the kernel also prepares the depth sort (gspl_sort_device.h): the first pass's digit counts of the keys it writes.
template <int MODE, bool HEADER>
__global__ __launch_bounds__(256) void bin_keys_kernel(
    int N, const float* __restrict__ means2d, const int32_t* __restrict__ radii, const float* __restrict__ depths,
    const float* __restrict__ conics, const float* __restrict__ opacities,
    int tile_size, int tile_w, int tile_h, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids, int32_t* __restrict__ counts,
    SpanRecord* __restrict__ spans, RadixProducer hdr) {
    __shared__ uint32_t s_hist[HEADER ? RADIX_BINS : 1];
    if (HEADER) {
        radix_producer_clear(s_hist);
        __syncthreads();
    }
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = g < N;
    uint32_t key = 0xFFFFFFFFu;
    if (valid) {
        int n = 0;
        const int radius = radii[g];
        // the records are assembled in 32-bit words (static indices only, so that they stay in registers)
        uint32_t w0 = 0u, wc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, wn[4] = {0u, 0u, 0u, 0u};
        int rows = 0;
        bool big = false;
        if (radius > 0) {
            int minx, miny, maxx, maxy;
            const float mx = means2d[g * 2 + 0], my = means2d[g * 2 + 1];
            tile_rect<MODE>(mx, my, radius, tile_size, tile_w, tile_h, minx, miny, maxx, maxy);
            SplatCull sc;
            sc.kind = 2;
            if (conics) sc = make_cull(mx, my, conics[g * 3 + 0], conics[g * 3 + 1], conics[g * 3 + 2], opacities[g]);
            rows = max(maxy - miny, 0);
            big = rows > EMIT_ROWS;
#pragma unroll
            for (int r = 0; r < EMIT_ROWS; ++r) {
                int c0 = 0, c1 = 0;
                if (r < rows) row_columns<MODE>(sc, miny + r, tile_size, minx, maxx, c0, c1);
                const int w = c1 - c0;
                n += w;
                big = big || w > 255;
                wc[r >> 1] |= (uint32_t)(c0 & 0xFFFF) << (16 * (r & 1));
                wn[r >> 2] |= (uint32_t)(w & 0xFF) << (8 * (r & 3));
            }
            for (int r = EMIT_ROWS; r < rows; ++r) {
                int c0, c1;
                row_columns<MODE>(sc, miny + r, tile_size, minx, maxx, c0, c1);
                n += c1 - c0;
            }
            w0 = (uint32_t)(miny & 0xFFFF) | ((big ? (uint32_t)SPAN_BIG : (uint32_t)rows) << 16);
        }
        // bit 31 tags the splats the emission leaves to bin_emit_big_kernel (the scan of the counts ranks them on its way)
        counts[g] = n | ((big && n > 0) ? (int)0x80000000 : 0);
        ids[g] = (uint32_t)g;
        key = n > 0 ? __float_as_uint(depths[g]) : 0xFFFFFFFFu;      // splats without tiles sort to the end
        keys[g] = key;
        uint4* dst = reinterpret_cast<uint4*>(spans + g);      // layout of SpanRecord (little endian)
        dst[0] = make_uint4(w0, wc[0], wc[1], wc[2]);
        dst[1] = make_uint4(wc[3], wn[0], wn[1], 0u);
        if (rows > 8 && !big) {
            uint4* ext = reinterpret_cast<uint4*>(spans + N + g);
            ext[0] = make_uint4(wc[4], wc[5], wc[6], wc[7]);
            ext[1] = make_uint4(wn[2], wn[3], 0u, 0u);
        }
    }
    if (HEADER) {
        radix_producer_add<uint32_t>(s_hist, hdr, key, valid);
        __syncthreads();
        radix_producer_flush(s_hist, hdr, (size_t)blockIdx.x * blockDim.x);
    }
}

// Load-balanced emission.  A lane-per-splat loop would let every lane write its own run of records (8 B stores 30-60 B
// apart: partial-line writes, and one huge splat serialises a whole wave).  Here a wave owns 64 consecutive splats of
// the depth order = one CONTIGUOUS output range; lanes walk that range in stride (coalesced 8-B stores) and find the
// owner of each output slot by binary search over the wave's prefix counts in LDS, then the tile row by a short scan
// of the owner's per-row prefix (splats with more than EMIT_ROWS rows are emitted cooperatively, lanes over rows).
__device__ __forceinline__ int wave_excl_scan(int v, int lane) {
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    return incl - v;
}

template <int MODE>
__global__ __launch_bounds__(256) void bin_emit_lb_kernel(
    int N, const float* __restrict__ means2d, const int32_t* __restrict__ radii, const uint32_t* __restrict__ order,
    const float* __restrict__ conics, const float* __restrict__ opacities,
    const int64_t* __restrict__ cum_sorted, const SpanRecord* __restrict__ spans, const int32_t* __restrict__ big_list,
    int tile_size, int tile_w, int tile_h, uint64_t* __restrict__ tile_keys, int64_t capacity, RadixProducer hdr) {
    // The kernel also prepares the tile sort (gspl_sort_device.h): the first pass's digit counts per sort workgroup.  A sort
    // workgroup owns hdr.span_items consecutive records; the records a workgroup emits in one phase start in one span (`home`) and
    // almost always end in it or the next: those are counted in LDS (two rows), the stragglers straight in memory.
    __shared__ uint32_t s_hist[2 * RADIX_BINS];
    auto hist_clear = [&]() { for (int j = threadIdx.x; j < 2 * RADIX_BINS; j += 256) s_hist[j] = 0u; };
    auto hist_flush = [&](uint32_t home) {
        for (int j = threadIdx.x; j < 2 * RADIX_BINS; j += 256) {
            const uint32_t c = s_hist[j];
            if (c) {
                const size_t g = (size_t)home + (uint32_t)(j >> 8);
                atomicAdd(hdr.counts0 + g * RADIX_BINS + (j & 255), c);
                atomicAdd(hdr.groups0 + (g / RADIX_GROUP) * RADIX_BINS + (j & 255), c);
            }
        }
    };
    // consecutive output slots carry different tile ids: one LDS atomic per record (no wave-uniform shortcut)
    auto count_tile = [&](uint32_t tile_id, uint32_t out, uint32_t home_first) {
        const uint32_t d = (tile_id >> (hdr.shift - 32)) & hdr.mask;
        const uint32_t rel = out - home_first;
        if (rel < 2u * hdr.span_items) {
            atomicAdd(&s_hist[(rel >= hdr.span_items ? RADIX_BINS : 0) + d], 1u);
        } else {
            const size_t g = out / hdr.span_items;
            atomicAdd(hdr.counts0 + g * RADIX_BINS + d, 1u);
            atomicAdd(hdr.groups0 + (g / RADIX_GROUP) * RADIX_BINS + d, 1u);
        }
    };
    hist_clear();
    __syncthreads();
    // the workgroup's first output slot (phase A); slots at or past `capacity` are never written
    const int64_t wg_base = blockIdx.x == 0 ? 0 : cum_sorted[(size_t)blockIdx.x * 256 - 1];
    const uint32_t home_a = (uint32_t)((wg_base < capacity ? wg_base : 0) / hdr.span_items);
    const uint32_t home_a_first = home_a * hdr.span_items;
    __shared__ int s_start[4][65];
    __shared__ int s_out[4][64];
    __shared__ uint32_t s_gid[4][64];
    __shared__ int s_row0[4][64];
    __shared__ uint16_t s_c0[4][64][EMIT_ROWS];
    __shared__ uint16_t s_pre[4][64][EMIT_ROWS + 1];
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int i = blockIdx.x * 256 + t;
    const int wave_first = blockIdx.x * 256 + w * 64;
    // ---- phase A: the wave's 64 splats, except the big ones ----------------------------------------------------------------
    if (wave_first < N) {
        const int64_t wave_base = (wave_first == 0) ? 0 : cum_sorted[wave_first - 1];
        int g = 0, cnt = 0, start = 0;
        uint4 ra = make_uint4(0u, 0u, 0u, 0u), rb = make_uint4(0u, 0u, 0u, 0u);      // the splat's SpanRecord as eight words
        if (i < N) {
            const int64_t off = (i == 0) ? 0 : cum_sorted[i - 1];
            start = (int)(off - wave_base);
            cnt = (int)(cum_sorted[i] - off);
            g = (int)order[i];
            if (cnt > 0) {
                const uint4* src = reinterpret_cast<const uint4*>(spans + g);
                ra = src[0];
                rb = src[1];
            }
        }
        const uint32_t rec_rows = ra.x >> 16, rec_miny = ra.x & 0xFFFFu;
        // splats the records cannot describe (more than EMIT_ROWS tile rows, or a very wide row) are emitted in phase B:
        // their slots are taken out of the range the lanes walk (a screen-filling splat has thousands)
        const bool big = cnt > 0 && rec_rows == (uint32_t)SPAN_BIG;
        uint4 ea = make_uint4(0u, 0u, 0u, 0u), eb = make_uint4(0u, 0u, 0u, 0u);      // rows 8..15 (second record), only when present
        if (cnt > 0 && !big && rec_rows > 8u) {
            const uint4* src = reinterpret_cast<const uint4*>(spans + N + g);
            ea = src[0];
            eb = src[1];
        }
        const uint32_t wc[8] = {ra.y, ra.z, ra.w, rb.x, ea.x, ea.y, ea.z, ea.w}, wn[4] = {rb.y, rb.z, eb.x, eb.y};
        const int cnt_small = big ? 0 : cnt;
        const int start_small = wave_excl_scan(cnt_small, l);
        const int total = __shfl(start_small + cnt_small, 63);
        s_start[w][l] = start_small;
        if (l == 0) s_start[w][64] = total;
        s_out[w][l] = start;
        s_gid[w][l] = (uint32_t)g;
        s_row0[w][l] = (int)rec_miny;
        if (cnt_small > 0) {
            int acc = 0;
#pragma unroll
            for (int r = 0; r < EMIT_ROWS; ++r) {
                s_c0[w][l][r] = (uint16_t)(wc[r >> 1] >> (16 * (r & 1)));
                s_pre[w][l][r] = (uint16_t)acc;
                acc += (int)((wn[r >> 2] >> (8 * (r & 3))) & 0xFFu);
            }
            s_pre[w][l][EMIT_ROWS] = (uint16_t)acc;
        }
        __builtin_amdgcn_wave_barrier();
        // every output slot of the wave's (small) range, owners found by binary search
        for (int k = l; k < total; k += 64) {
            int lo = 0, hi = 63;                       // largest o with s_start[o] <= k and a non-empty segment after it
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_start[w][mid] <= k) lo = mid; else hi = mid - 1;
            }
            // lo may point at an empty segment sharing the start: the owner is the LAST lane with start <= k, which the
            // search returns because empty segments before the owner have start == owner's start and come earlier
            const int o = lo;
            const int kk = k - s_start[w][o];
            int r = 0;                                  // last row whose prefix <= kk (empty rows share their successor's prefix)
#pragma unroll
            for (int step = EMIT_ROWS / 2; step > 0; step >>= 1) r += (kk >= (int)s_pre[w][o][r + step]) ? step : 0;
            const int tx = (int)s_c0[w][o][r] + kk - (int)s_pre[w][o][r];
            const int ty = s_row0[w][o] + r;
            const int64_t out = wave_base + s_out[w][o] + kk;
            if (out < capacity) {    // a speculative launch may have guessed the list length too low (the host redoes it)
                const uint32_t tile_id = (uint32_t)(ty * tile_w + tx);
                tile_keys[out] = ((uint64_t)tile_id << 32) | s_gid[w][o];
                count_tile(tile_id, (uint32_t)out, home_a_first);
            }
        }
    }
    // ---- phase B: the big splats of the whole frame, dealt out to the workgroups -------------------------------------------
    // (close-up splats are consecutive in depth order: left to their own waves, a few waves would emit up to 64
    // screen-filling splats one after the other.)  `big_list` / cum_sorted[N]: their depth-order indices and number, ranked
    // by the scan of the counts.  All 256 threads work on one splat: rows in chunks of 256 (one per thread: exact column
    // span, scan), then every output slot of the chunk by one thread (row by binary search in the chunk's prefix).
    __syncthreads();
    hist_flush(home_a);
    const int n_big = (int)cum_sorted[N];
    __shared__ int s_bpre[257];
    __shared__ int s_bc0[256];
    __shared__ int s_bwave[4];
    for (int b = blockIdx.x; b < n_big; b += gridDim.x) {      // (uniform per workgroup)
        const int bi = big_list[b];
        const int g = (int)order[bi];
        int64_t out = (bi == 0) ? 0 : cum_sorted[bi - 1];
        const uint32_t home_b = (uint32_t)((out < capacity ? out : 0) / hdr.span_items), home_b_first = home_b * hdr.span_items;
        __syncthreads();                               // the flush before is done with the table
        hist_clear();                                  // (the chunk loop below synchronises before its first record)
        int minx, miny, maxx, maxy;
        const float mx = means2d[g * 2 + 0], my = means2d[g * 2 + 1];
        tile_rect<MODE>(mx, my, radii[g], tile_size, tile_w, tile_h, minx, miny, maxx, maxy);
        SplatCull sc;
        sc.kind = 2;
        if (conics) sc = make_cull(mx, my, conics[g * 3 + 0], conics[g * 3 + 1], conics[g * 3 + 2], opacities[g]);
        for (int rbase = miny; rbase < maxy; rbase += 256) {
            const int ty = rbase + t;
            int c0 = 0, c1 = 0;
            if (ty < maxy) row_columns<MODE>(sc, ty, tile_size, minx, maxx, c0, c1);
            const int n = c1 - c0;
            const int pre = wave_excl_scan(n, l);
            __syncthreads();                           // the previous chunk (or phase A) is done with the shared arrays
            if (l == 63) s_bwave[w] = pre + n;
            __syncthreads();
            int woff = 0, total = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int c = s_bwave[k]; if (k < w) woff += c; total += c; }
            s_bpre[t] = woff + pre;
            s_bc0[t] = c0;
            if (t == 0) s_bpre[256] = total;
            __syncthreads();
            for (int k = t; k < total; k += 256) {
                int r = 0;
#pragma unroll
                for (int step = 128; step > 0; step >>= 1) r += (k >= s_bpre[r + step]) ? step : 0;
                const int tx = s_bc0[r] + k - s_bpre[r];
                if (out + k < capacity) {
                    const uint32_t tile_id = (uint32_t)((rbase + r) * tile_w + tx);
                    tile_keys[out + k] = ((uint64_t)tile_id << 32) | (uint32_t)g;
                    count_tile(tile_id, (uint32_t)(out + k), home_b_first);
                }
            }
            out += total;
        }
        __syncthreads();
        hist_flush(home_b);
    }
}

struct BinWorkspace {
    size_t keys_off, ids_off, keys2_off, counts_off, sort1_off, sort1_bytes, scan_states_off;
    size_t tkeys_off, tkeys2_off, sort2_off, sort2_bytes;
    size_t total_count, total;
    RadixPlan depth, tile;
};

// Workspace of the list-only binning.  Count half: depth keys / ids (x2 for the ping-pong), counts, the depth sort's tables
// followed by the scan's block sums.  Emit/sort half (sized by the list length or a guess of it): the 8-byte records (x2) and
// the tile sort's tables.
static int plan_bin(int N, int64_t n_isects, int n_tiles, BinWorkspace& w) {
    const size_t n = (size_t)(N > 0 ? N : 1), ni = (size_t)(n_isects > 0 ? n_isects : 1);
    if (n > RADIX_MAX_ITEMS || ni > RADIX_MAX_ITEMS) { set_error("bin", "more than 2^30-1 splats or intersections"); return GSPL_ERR_UNSUPPORTED; }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.keys_off = take(4 * n); w.ids_off = take(4 * n); w.keys2_off = take(4 * n);
    w.counts_off = take(4 * n);
    if (!radix_plan(n, 0, 32, 8, RADIX_TILE_U32, w.depth)) return fail_arg("bin: depth sort plan");
    w.scan_states_off = w.depth.total_bytes;          // the scan's block sums follow the sort's tables
    w.sort1_bytes = w.depth.total_bytes + scan_workspace_bytes(n);
    w.sort1_off = take(w.sort1_bytes);
    w.total_count = off;
    w.tkeys_off = take(8 * ni); w.tkeys2_off = take(8 * ni);
    // n_tiles <= 0: size query (the tile grid is not known to gspl_bin_workspace_bytes) -> the widest plan, four passes
    int bits = n_tiles > 0 ? key_bits(n_tiles) - 32 : 32;
    if (bits < 2) bits = 2;                             // at least two passes: the pass before the last clears the tile counters
    if (!radix_plan(ni, 32, 32 + bits, bits > 8 ? 8 : (bits + 1) / 2, RADIX_TILE_U64, w.tile) || w.tile.passes < 2) return fail_arg("bin: tile sort plan");
    w.sort2_bytes = w.tile.total_bytes;
    w.sort2_off = take(w.sort2_bytes);
    w.total = off;
    return GSPL_OK;
}

}  // namespace gspl

extern "C" size_t gspl_bin_workspace_bytes(int N, int64_t n_isects) {
    gspl::BinWorkspace w;
    if (gspl::plan_bin(N, n_isects, 0, w) != GSPL_OK) return 0;
    return n_isects > 0 ? w.total : w.total_count;
}

namespace gspl {
// gspl_bin_count; ticket != 0: host_counts has a third word that receives the ticket AFTER the two numbers (the caller polls it)
int bin_count_ticket(int N, int mode, const float* means2d, const int32_t* radii, const float* depths,
                              const float* conics, const float* opacities,
                              int tile_size, int tile_w, int tile_h,
                              int32_t* order, int64_t* cum_tiles, int32_t* big_list, void* spans, int64_t* host_counts,
                              void* workspace, size_t workspace_bytes, void* stream, unsigned long long ticket,
                              bool depth_header_zeroed, ZeroJob then_zero) {
    if (N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) return fail_arg("bin_count: bad sizes");
    if (mode != GSPL_MODE_GSPLAT && mode != GSPL_MODE_INRIA) return fail_arg("bin_count: bad mode");
    if (N == 0) return GSPL_OK;
    if (!means2d || !radii || !depths || !order || !cum_tiles || !big_list || !spans || !workspace) return fail_arg("bin_count: NULL required pointer");
    if (tile_w > 65535 || tile_h > 65535) { set_error("bin_count", "more than 65535 tile rows or columns"); return GSPL_ERR_UNSUPPORTED; }
    if ((conics == nullptr) != (opacities == nullptr)) return fail_arg("bin_count: conics and opacities go together");
    BinWorkspace w;
    int rc = plan_bin(N, 0, tile_w * tile_h, w);
    if (rc != GSPL_OK) return rc;
    if (workspace_bytes < w.total_count) return fail_ws("bin_count");
    static_assert(GSPL_BIN_SPAN_BYTES == 2 * sizeof(SpanRecord), "spans = N primary + N extension records");
    char* ws = (char*)workspace;
    uint32_t* keys = (uint32_t*)(ws + w.keys_off);
    uint32_t* ids = (uint32_t*)(ws + w.ids_off);
    uint32_t* keys2 = (uint32_t*)(ws + w.keys2_off);
    int32_t* counts = (int32_t*)(ws + w.counts_off);
    hipStream_t s = (hipStream_t)stream;
    const int grid = (N + 255) / 256;
    // Depth sort (sort.hip), prepared by the key pass itself (the first pass's counts).  Four 8-bit passes: the sorted sequence
    // ends where it started, so the key pass writes the ids straight into `order`.
    const RadixPlan& dp = w.depth;
    RadixProducer hdr;
    radix_producer_args(dp, ws + w.sort1_off, hdr);
    if (!depth_header_zeroed) {
        rc = radix_zero(ws + w.sort1_off, dp.header_bytes, s);
        if (rc != GSPL_OK) return rc;
    }
    if (mode == GSPL_MODE_GSPLAT)
        hipLaunchKernelGGL((bin_keys_kernel<GSPL_MODE_GSPLAT, true>), dim3(grid), dim3(256), 0, s, N, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, keys, (uint32_t*)order, counts, (SpanRecord*)spans, hdr);
    else
        hipLaunchKernelGGL((bin_keys_kernel<GSPL_MODE_INRIA, true>), dim3(grid), dim3(256), 0, s, N, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, keys, (uint32_t*)order, counts, (SpanRecord*)spans, hdr);
    rc = check_launch("bin_keys");
    if (rc != GSPL_OK) return rc;
    uint32_t* const kbuf[2] = {keys, keys2};
    uint32_t* const vbuf[2] = {(uint32_t*)order, ids};
    // the last pass leaves the tile counts in depth order where the sorted keys would go
    rc = radix_sort_u32(dp, ws + w.sort1_off, kbuf, vbuf, true, s, (const uint32_t*)counts);
    if (rc != GSPL_OK) return rc;
    // the scan also ranks the tagged (big) splats: big_list[rank] = depth index, cum_tiles[N] = how many — one 16-byte read-back
    // gives the host both numbers
    return scan_gathered_counts(nullptr, (const int32_t*)kbuf[dp.passes & 1], cum_tiles, (size_t)N, ws + w.sort1_off + w.scan_states_off, big_list, s, host_counts,
                                host_counts ? ticket : 0ull, then_zero);
}

// The two tables a frame clears, as jobs for whichever earlier kernel of the stream has threads to spare
int bin_depth_header(int N, int n_tiles, void* count_workspace, ZeroJob& job) {
    BinWorkspace w;
    int rc = plan_bin(N, 0, n_tiles, w);
    if (rc != GSPL_OK) return rc;
    job.p = (uint4*)((char*)count_workspace + w.sort1_off);
    job.n16 = (uint32_t)(w.depth.header_bytes / 16);
    return (w.depth.header_bytes % 16 || ((uintptr_t)job.p & 15u)) ? fail_arg("bin_depth_header: not 16-byte aligned") : GSPL_OK;
}
int bin_tile_header(int N, int64_t capacity, int n_tiles, void* workspace, ZeroJob& job) {
    BinWorkspace w;
    int rc = plan_bin(N, capacity, n_tiles, w);
    if (rc != GSPL_OK) return rc;
    job.p = (uint4*)((char*)workspace + w.sort2_off);
    job.n16 = (uint32_t)(w.tile.header_bytes / 16);
    return (w.tile.header_bytes % 16 || ((uintptr_t)job.p & 15u)) ? fail_arg("bin_tile_header: not 16-byte aligned") : GSPL_OK;
}
}  // namespace gspl

extern "C" int gspl_bin_count(int N, int mode, const float* means2d, const int32_t* radii, const float* depths,
                              const float* conics, const float* opacities,
                              int tile_size, int tile_w, int tile_h,
                              int32_t* order, int64_t* cum_tiles, int32_t* big_list, void* spans, int64_t* host_counts,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return gspl::bin_count_ticket(N, mode, means2d, radii, depths, conics, opacities, tile_size, tile_w, tile_h, order, cum_tiles, big_list, spans, host_counts,
                                  workspace, workspace_bytes, stream, 0ull);
}

// Emission half of gspl_bin_emit_sort.  `capacity` = records the workspace (gspl_bin_workspace_bytes(N, capacity)) has
// room for: it may be a GUESS of the list length, launched before the host knows the real one — records past it are
// dropped, and the caller repeats the call with the real length when the guess was too low.
// The kernel also prepares the tile sort: the first pass's digit counts of the records it writes.
extern "C" int gspl_bin_emit(int N, int mode, const float* means2d, const int32_t* radii,
                             const float* conics, const float* opacities,
                             const int32_t* order, const int64_t* cum_tiles, const int32_t* big_list, const void* spans,
                             int tile_size, int tile_w, int tile_h, int64_t capacity,
                             void* workspace, size_t workspace_bytes, void* stream) {
    return gspl::bin_emit_impl(N, mode, means2d, radii, conics, opacities, order, cum_tiles, big_list, spans, tile_size, tile_w, tile_h, capacity,
                               workspace, workspace_bytes, stream, false);
}

namespace gspl {
int bin_emit_impl(int N, int mode, const float* means2d, const int32_t* radii, const float* conics, const float* opacities,
                  const int32_t* order, const int64_t* cum_tiles, const int32_t* big_list, const void* spans,
                  int tile_size, int tile_w, int tile_h, int64_t capacity, void* workspace, size_t workspace_bytes, void* stream,
                  bool tile_header_zeroed) {
    if (N < 0 || capacity < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) return fail_arg("bin_emit: bad sizes");
    if (mode != GSPL_MODE_GSPLAT && mode != GSPL_MODE_INRIA) return fail_arg("bin_emit: bad mode");
    if (N == 0 || capacity == 0) return GSPL_OK;
    if (capacity > 0x7fffffffll) return fail_arg("bin_emit: more than 2^31-1 intersections");
    if (!means2d || !radii || !order || !cum_tiles || !big_list || !spans || !workspace) return fail_arg("bin_emit: NULL required pointer");
    BinWorkspace w;
    int rc = plan_bin(N, capacity, tile_w * tile_h, w);
    if (rc != GSPL_OK) return rc;
    if (workspace_bytes < w.total) return fail_ws("bin_emit");
    char* ws = (char*)workspace;
    uint64_t* tkeys = (uint64_t*)(ws + w.tkeys_off);
    hipStream_t s = (hipStream_t)stream;
    const int grid = (N + 255) / 256;
    // the tile sort's first-pass counts: exactly the records written, by the spans of the plan for `capacity` records (the real
    // list is not longer, else the emission is repeated; gspl_bin_sort keeps the spans)
    RadixProducer hdr;
    radix_producer_args(w.tile, ws + w.sort2_off, hdr);
    if (!tile_header_zeroed) {
        rc = radix_zero(ws + w.sort2_off, w.tile.header_bytes, s);
        if (rc != GSPL_OK) return rc;
    }
    if (mode == GSPL_MODE_GSPLAT)
        hipLaunchKernelGGL(bin_emit_lb_kernel<GSPL_MODE_GSPLAT>, dim3(grid), dim3(256), 0, s, N, means2d, radii, (const uint32_t*)order, conics, opacities, cum_tiles, (const SpanRecord*)spans, big_list, tile_size, tile_w, tile_h, tkeys, capacity, hdr);
    else
        hipLaunchKernelGGL(bin_emit_lb_kernel<GSPL_MODE_INRIA>, dim3(grid), dim3(256), 0, s, N, means2d, radii, (const uint32_t*)order, conics, opacities, cum_tiles, (const SpanRecord*)spans, big_list, tile_size, tile_w, tile_h, tkeys, capacity, hdr);
    return check_launch("bin_emit");
}
}  // namespace gspl

// Sort half: the first n_isects (<= capacity) records of the workspace gspl_bin_emit filled -> flatten_ids, offsets.
// Two (for more than 65536 tiles: three) passes on the tile id; the last one writes the splat ids alone and counts
// the records per tile, a one-workgroup scan turns the counts into `offsets`.
extern "C" int gspl_bin_sort(int N, int tile_w, int tile_h, int64_t n_isects, int64_t capacity,
                             int32_t* flatten_ids, int32_t* offsets, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gspl;
    if (N < 0 || n_isects < 0 || capacity < n_isects || tile_w <= 0 || tile_h <= 0) return fail_arg("bin_sort: bad sizes");
    if (!offsets) return fail_arg("bin_sort: NULL offsets");
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0 || n_isects == 0) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, s, n_tiles, 0, offsets);
        return check_launch("bin_sort(fill)");
    }
    if (capacity > 0x7fffffffll) return fail_arg("bin_sort: more than 2^31-1 intersections");
    if (!flatten_ids || !workspace) return fail_arg("bin_sort: NULL required pointer");
    BinWorkspace w;
    int rc = plan_bin(N, capacity, n_tiles, w);       // the layout gspl_bin_emit used
    if (rc != GSPL_OK) return rc;
    if (workspace_bytes < w.total) return fail_ws("bin_sort");
    char* ws = (char*)workspace;
    uint64_t* const tk[2] = {(uint64_t*)(ws + w.tkeys_off), (uint64_t*)(ws + w.tkeys2_off)};
    // same bit split and spans as the emission's counts, tile count of the REAL list length
    RadixPlan tp = w.tile;
    radix_replan_items(tp, (size_t)n_isects);
    rc = radix_sort_tiles(tp, ws + w.sort2_off, tk, true, (uint32_t*)flatten_ids, (uint32_t*)offsets, (uint32_t)n_tiles, s);
    if (rc != GSPL_OK) return rc;
    return tile_offsets_from_counts((uint32_t*)offsets, (uint32_t)n_tiles, s);
}

// gspl_bin_sort for a host that has NOT read the list length back yet: the records were emitted into room for `capacity` of them,
// their real number sits in *n_isects_dev (the scan's cum_tiles[N - 1]).  flatten_ids has room for `capacity` ids; offsets has
// tile_w * tile_h + 1 entries, the last one receives the list length (compositing calls take n_isects = -1 with such an array).
// The caller checks *n_isects_dev <= capacity afterwards (it has the number in host memory by then) and repeats the frame's
// emission and sort when the guess was too low.
extern "C" int gspl_bin_sort_device_count(int N, int tile_w, int tile_h, const int64_t* n_isects_dev, int64_t capacity,
                                          int32_t* flatten_ids, int32_t* offsets, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gspl;
    if (N <= 0 || capacity <= 0 || tile_w <= 0 || tile_h <= 0) return fail_arg("bin_sort_device_count: bad sizes");
    if (capacity > 0x7fffffffll) return fail_arg("bin_sort_device_count: more than 2^31-1 intersections");
    if (!n_isects_dev || !flatten_ids || !offsets || !workspace) return fail_arg("bin_sort_device_count: NULL required pointer");
    const int n_tiles = tile_w * tile_h;
    BinWorkspace w;
    int rc = plan_bin(N, capacity, n_tiles, w);       // the layout and the spans gspl_bin_emit used
    if (rc != GSPL_OK) return rc;
    if (workspace_bytes < w.total) return fail_ws("bin_sort_device_count");
    char* ws = (char*)workspace;
    hipStream_t s = (hipStream_t)stream;
    uint64_t* const tk[2] = {(uint64_t*)(ws + w.tkeys_off), (uint64_t*)(ws + w.tkeys2_off)};
    rc = radix_sort_tiles(w.tile, ws + w.sort2_off, tk, true, (uint32_t*)flatten_ids, (uint32_t*)offsets, (uint32_t)n_tiles + 1u, s, n_isects_dev);
    if (rc != GSPL_OK) return rc;
    return tile_offsets_from_counts((uint32_t*)offsets, (uint32_t)n_tiles + 1u, s);
}

extern "C" int gspl_bin_emit_sort(int N, int mode, const float* means2d, const int32_t* radii,
                                  const float* conics, const float* opacities,
                                  const int32_t* order, const int64_t* cum_tiles, const int32_t* big_list, const void* spans,
                                  int tile_size, int tile_w, int tile_h, int64_t n_isects,
                                  int32_t* flatten_ids, int32_t* offsets, void* workspace, size_t workspace_bytes, void* stream) {
    if (n_isects < 0) return gspl::fail_arg("bin_emit_sort: bad sizes");
    int rc = gspl_bin_emit(N, mode, means2d, radii, conics, opacities, order, cum_tiles, big_list, spans, tile_size, tile_w, tile_h, n_isects,
                           workspace, workspace_bytes, stream);
    if (rc != GSPL_OK) return rc;
    return gspl_bin_sort(N, tile_w, tile_h, n_isects, n_isects, flatten_ids, offsets, workspace, workspace_bytes, stream);
}

extern "C" size_t gspl_isect_workspace_bytes(int N, int64_t n_isects) {
    gspl::IsectWorkspace w;
    if (gspl::plan_workspace(N, n_isects, gspl::ISECT_MAX_KEY_BITS, w) != GSPL_OK) return 0;      // sized for the widest key
    return w.total;
}

extern "C" int gspl_isect_count(int N, int mode, const float* means2d, const int32_t* radii,
                                int tile_size, int tile_w, int tile_h,
                                int32_t* tiles_per_gauss, int64_t* cum_tiles,
                                void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gspl;
    if (N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) return fail_arg("isect_count: bad sizes");
    if (mode != GSPL_MODE_GSPLAT && mode != GSPL_MODE_INRIA) return fail_arg("isect_count: bad mode");
    if (N == 0) return GSPL_OK;
    if (!means2d || !radii || !tiles_per_gauss || !cum_tiles || !workspace) return fail_arg("isect_count: NULL required pointer");
    IsectWorkspace w;
    int rc = plan_workspace(N, 0, ISECT_MAX_KEY_BITS, w);
    if (rc != GSPL_OK) return rc;
    if (workspace_bytes < w.total_count) return fail_ws("isect_count");
    char* ws = (char*)workspace;
    int32_t* counts = (int32_t*)(ws + w.counts_off);
    const int grid = (N + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (mode == GSPL_MODE_GSPLAT)
        hipLaunchKernelGGL(isect_count_kernel<GSPL_MODE_GSPLAT>, dim3(grid), dim3(256), 0, s, N, means2d, radii, tile_size, tile_w, tile_h, tiles_per_gauss, counts);
    else
        hipLaunchKernelGGL(isect_count_kernel<GSPL_MODE_INRIA>, dim3(grid), dim3(256), 0, s, N, means2d, radii, tile_size, tile_w, tile_h, tiles_per_gauss, counts);
    rc = check_launch("isect_count");
    if (rc != GSPL_OK) return rc;
    // inclusive scan of the counts in memory order: the scan of sort.hip with the identity gather
    return scan_gathered_counts(nullptr, counts, cum_tiles, (size_t)N, ws + w.scan_off, nullptr, s);
}

extern "C" int gspl_isect_emit_sort(int N, int mode, const float* means2d, const int32_t* radii, const float* depths,
                                    const int64_t* cum_tiles, int tile_size, int tile_w, int tile_h, int64_t n_isects,
                                    int64_t* isect_ids, int32_t* flatten_ids,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gspl;
    if (N < 0 || n_isects < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) return fail_arg("isect_emit_sort: bad sizes");
    if (mode != GSPL_MODE_GSPLAT && mode != GSPL_MODE_INRIA) return fail_arg("isect_emit_sort: bad mode");
    if (N == 0 || n_isects == 0) return GSPL_OK;
    if (n_isects > 0x7fffffffll) return fail_arg("isect_emit_sort: more than 2^31-1 intersections");
    if (!means2d || !radii || !depths || !cum_tiles || !isect_ids || !flatten_ids || !workspace) return fail_arg("isect_emit_sort: NULL required pointer");
    IsectWorkspace w;
    const int bits = key_bits(tile_w * tile_h);
    int rc = plan_workspace(N, n_isects, bits, w);
    if (rc != GSPL_OK) return rc;
    IsectWorkspace wmax;
    rc = plan_workspace(N, n_isects, ISECT_MAX_KEY_BITS, wmax);       // the size gspl_isect_workspace_bytes promised
    if (rc != GSPL_OK) return rc;
    if (workspace_bytes < wmax.total) return fail_ws("isect_emit_sort");
    char* ws = (char*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const int grid = (N + 255) / 256;
    // Stable LSD sort of (u64 key, u32 value) pairs on the significant key bits: [0, 24) then [24, bits), at most three 8-bit
    // passes each.  The buffers ping-pong between the caller's arrays and the workspace; the emission starts in the one that
    // makes the LAST pass land in the caller's arrays.
    uint64_t* ukeys = (uint64_t*)isect_ids;
    uint32_t* uvals = (uint32_t*)flatten_ids;
    uint64_t* wkeys = (uint64_t*)(ws + wmax.keys_off);
    uint32_t* wvals = (uint32_t*)(ws + wmax.vals_off);
    const int total_passes = w.lo.passes + (w.two_sorts ? w.hi.passes : 0);
    const bool start_in_user = (total_passes % 2) == 0;
    uint64_t* k0 = start_in_user ? ukeys : wkeys;
    uint32_t* v0 = start_in_user ? uvals : wvals;
    uint64_t* k1 = start_in_user ? wkeys : ukeys;
    uint32_t* v1 = start_in_user ? wvals : uvals;
    if (mode == GSPL_MODE_GSPLAT)
        hipLaunchKernelGGL(isect_emit_kernel<GSPL_MODE_GSPLAT>, dim3(grid), dim3(256), 0, s, N, means2d, radii, depths, cum_tiles, tile_size, tile_w, tile_h, k0, v0);
    else
        hipLaunchKernelGGL(isect_emit_kernel<GSPL_MODE_INRIA>, dim3(grid), dim3(256), 0, s, N, means2d, radii, depths, cum_tiles, tile_size, tile_w, tile_h, k0, v0);
    rc = check_launch("isect_emit");
    if (rc != GSPL_OK) return rc;
    uint64_t* kb[2] = {k0, k1};
    uint32_t* vb[2] = {v0, v1};
    rc = radix_sort_u64(w.lo, ws + wmax.sort_off, kb, vb, false, s);
    if (rc != GSPL_OK) return rc;
    if (w.two_sorts) {
        if (w.lo.passes & 1) { kb[0] = k1; kb[1] = k0; vb[0] = v1; vb[1] = v0; }
        rc = radix_sort_u64(w.hi, ws + wmax.sort_off, kb, vb, false, s);
    }
    return rc;
}

extern "C" int gspl_isect_offsets(int64_t n_isects, const int64_t* isect_ids, int tile_w, int tile_h,
                                  int32_t* offsets, void* stream) {
    using namespace gspl;
    if (n_isects < 0 || tile_w <= 0 || tile_h <= 0) return fail_arg("isect_offsets: bad sizes");
    if (!offsets) return fail_arg("isect_offsets: NULL offsets");
    const int n_tiles = tile_w * tile_h;
    hipStream_t s = (hipStream_t)stream;
    if (n_isects == 0) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, s, n_tiles, 0, offsets);
        return check_launch("isect_offsets(fill)");
    }
    if (!isect_ids) return fail_arg("isect_offsets: NULL isect_ids");
    const int64_t grid = (n_isects + 255) / 256;
    hipLaunchKernelGGL(isect_offsets_kernel, dim3((unsigned)grid), dim3(256), 0, s, n_isects, isect_ids, n_tiles, offsets);
    return check_launch("isect_offsets");
}
