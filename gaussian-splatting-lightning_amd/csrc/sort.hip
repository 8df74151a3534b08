// sort.hip — stable LSD radix sort and prefix scans of the binning stage (gfx950, wave64).  Interface and rationale: gspl_sort.h.
//
// Replaces (inside gspl_bin_count / gspl_bin_sort / gspl_isect_*) the device radix sorts the reference's native rasterizers call
// between projection and compositing: gsplat `isect_tiles` -> cub::DeviceRadixSort::SortPairs and the Inria rasterizer's
// `cub::DeviceRadixSort::SortPairs(point_list_keys...)` (call sites: reference gsplat_v1_renderer.py:524-556,
// vanilla_renderer.py:111).  Ordering contract: stable, ascending on the selected key bits — identical to those.
//
// One pass = two launches WITHOUT any waiting between running workgroups:
//   count    workgroup g owns a CONTIGUOUS range of tiles and histograms the pass's digit over it -> counts[g][digit], and adds
//            the row to its GROUP's row (groups[g / 32][digit], atomics on zeroed memory).  The pass-0 rows can come from the
//            kernel that produced the keys instead (`prepared`, gspl_sort_device.h).
//   scatter  workgroup g first sums, per digit, the group rows (all of them: the digit totals, whose exclusive scan is the digit's
//            base; those below its own group: items of lower workgroups) and the rows of the lower workgroups of its own group —
//            at most 64 + 31 coalesced 1 KB reads — then walks its range tile by tile: in-wave ranks by digit matching, tile
//            permuted through LDS into digit order, written out as runs.
// The first version was a one-sweep sort (single read of the keys per pass, decoupled look-back between tiles).  Its look-back
// walks as many predecessor states as there are tiles in flight — ~770 on this part — so above one resident wave of tiles a pass of
// 6 M pairs ran at 0.7 TB/s (135 us; rocPRIM's one-sweep ~100 us), it needed tile counters or residency assumptions for forward
// progress, and a time-out path.  Reading the keys twice costs less than that: no polling, no ordering assumptions, nothing to
// dead-lock under contention from side streams, other processes or RCCL kernels, bit-reproducible by construction.
#include "gspl_device.h"
#include "gspl_host.h"
#include "gspl_sort.h"
#include "gspl_sort_device.h"
#include <cstdio>
#include <cstdlib>

namespace gspl {

static constexpr int RS_WAVES = 8;
static constexpr int RS_THREADS = RS_WAVES * 64;

// the keys and values of a pass are read once: streaming loads
template <typename T>
__device__ __forceinline__ T load_once(const T* p) {
#ifdef GSPL_RS_PLAIN_LOAD
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}

// Exclusive scan of 256 LDS words (src -> dst) by the first wave, four words per lane.
__device__ __forceinline__ void scan256_excl(const uint32_t* src, uint32_t* dst) {
    const int t = threadIdx.x;
    if (t < 64) {
        uint32_t v[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = src[t * 4 + k]; s += v[k]; }
        uint32_t incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if (t >= d) incl += up; }
        uint32_t run = incl - s;
#pragma unroll
        for (int k = 0; k < 4; ++k) { dst[t * 4 + k] = run; run += v[k]; }
    }
}

// ---- count: counts[g][digit] = items of workgroup g's tile range whose digit of this pass is `digit`; groups[g / 32][digit] += ----
template <typename KeyT, int IPT>
__global__ __launch_bounds__(RS_THREADS) void radix_count_kernel(const KeyT* __restrict__ keys, uint32_t n, uint32_t ntiles, uint32_t tiles_per_wg,
                                                                 int shift, int nbits, uint32_t* __restrict__ counts, uint32_t* __restrict__ groups,
                                                                 const int64_t* __restrict__ n_dev) {
    constexpr uint32_t TILE = RS_THREADS * IPT;
    if (n_dev) {      // the item count lives on the device (<= the n the grid was sized for): workgroups past it find empty ranges
        const int64_t m = *n_dev;
        n = (uint32_t)(m < 0 ? 0 : (m < (int64_t)n ? m : (int64_t)n));
        ntiles = (n + TILE - 1u) / TILE;
    }
    __shared__ uint32_t cnt[RS_WAVES][RADIX_BINS];      // per-wave counters: no inter-wave contention on hot digits
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const uint32_t mask = (1u << nbits) - 1u;
    for (int k = l; k < RADIX_BINS; k += 64) cnt[w][k] = 0u;
    __builtin_amdgcn_wave_barrier();
    const uint32_t t0 = blockIdx.x * tiles_per_wg, t1 = min(ntiles, t0 + tiles_per_wg);
    const size_t lo = (size_t)t0 * TILE, hi = min((size_t)n, (size_t)t1 * TILE);
    // the range is one contiguous span of keys: plain strided sweep, 4 keys in flight per lane
    for (size_t i = lo + (size_t)t; i < hi; i += (size_t)RS_THREADS * 4) {
        KeyT k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t j = i + (size_t)u * RS_THREADS; k[u] = j < hi ? load_once(keys + j) : (KeyT)0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t j = i + (size_t)u * RS_THREADS;
            const bool valid = j < hi;
            const uint32_t d = (uint32_t)(k[u] >> shift) & mask;
            // lanes that share the digit add once (the high digits of depth keys and tile ids are nearly wave-uniform)
            const unsigned long long act = __ballot(valid);
            if (act) {
                const int first = (int)__builtin_ctzll(act);
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
                if (__ballot(valid && d != d0) == 0ull) {
                    if (l == first) cnt[w][d0] += (uint32_t)__builtin_popcountll(act);
                } else if (valid) {
                    atomicAdd(&cnt[w][d], 1u);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    if (t < (1 << nbits)) {
        uint32_t c = 0u;
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) c += cnt[k][t];
        counts[(size_t)blockIdx.x * RADIX_BINS + t] = c;
        if (c) atomicAdd(groups + (size_t)(blockIdx.x / RADIX_GROUP) * RADIX_BINS + t, c);
    }
}

template <typename KeyT, bool VALUES, int IPT>
struct RadixShared {
    uint32_t wcnt[RS_WAVES][RADIX_BINS];      // per-wave digit counters, then exclusive prefix over the waves
    uint32_t next[RADIX_BINS];                // global position of the workgroup's next item of each digit
    uint32_t tilecnt[RADIX_BINS];             // the tile's digit histogram
    uint32_t dstart[RADIX_BINS];              // first in-tile slot of each digit
    uint32_t gbase[RADIX_BINS];               // global position of in-tile slot 0 as seen by each digit (modular)
    // the permutation buffer; during the ranking (when it is free) its head holds the match masks: match[wave][digit] = the lanes
    // of the wave whose current item has that digit
    union {
        struct {
            KeyT xkey[RS_THREADS * IPT];
            uint32_t xval[VALUES ? RS_THREADS * IPT : 1];
        };
        unsigned long long match[RS_WAVES][RADIX_BINS];
    };
};

// ---- scatter ------------------------------------------------------------------------------------------------------------------
// A tile = 512 x IPT items held in (wave, round, lane) = memory order.  Per tile:
//   1. in-wave ranks by digit matching (lanes OR their bit into the digit's mask in LDS and read the mask back: ~10 VALU
//      instructions per 64 items where a ballot per digit bit costs ~70), per-wave digit counters in LDS
//   2. counters -> tile histogram; exclusive scan -> first in-tile slot per digit
//   3. permute the tile through LDS into digit order
//   4. write out: consecutive lanes hold consecutive items of a digit run -> runs of consecutive addresses
// FINAL = FINAL_TILES (u64 keys, no values; the LAST pass of the tile sort of the binning): the sorted records leave as their low
//   words only (vals_out = the per-tile lists of splat ids), and the number of records per tile id (the key's high word) is counted
//   into aux[tile id].  The pass before left the records ordered by the low digit, so after the in-tile permutation equal tile ids
//   are contiguous runs in LDS: one atomic pair per run (subtract its first slot, add one past its last).
// FINAL = FINAL_GATHER (u32 keys with values; the LAST pass of the depth sort of the binning): nobody reads the sorted keys, so
//   keys_out receives gather[value] instead — the per-splat tile counts in depth order, which the scan then reads contiguously.
// aux_zero > 0: workgroup 0 clears aux[0, aux_zero) (the pass BEFORE the final one prepares the counters).
enum { FINAL_NONE = 0, FINAL_TILES = 1, FINAL_GATHER = 2 };
template <typename KeyT, bool VALUES, int IPT, int FINAL>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                   KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                                   uint32_t ntiles, uint32_t tiles_per_wg, int shift, int nbits,
                                                                   const uint32_t* __restrict__ groups, const uint32_t* __restrict__ counts,
                                                                   uint32_t* __restrict__ aux, uint32_t aux_zero,
                                                                   const uint32_t* __restrict__ gather, const int64_t* __restrict__ n_dev) {
    constexpr uint32_t TILE = RS_THREADS * IPT;
    if (n_dev) {      // see radix_count_kernel
        const int64_t m = *n_dev;
        n = (uint32_t)(m < 0 ? 0 : (m < (int64_t)n ? m : (int64_t)n));
        ntiles = (n + TILE - 1u) / TILE;
    }
    __shared__ RadixShared<KeyT, VALUES, IPT> sh;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const uint32_t mask = (1u << nbits) - 1u;
    const int nd = 1 << nbits;
    const uint32_t t0 = blockIdx.x * tiles_per_wg, t1 = min(ntiles, t0 + tiles_per_wg);
    KeyT key[IPT], key_next[IPT];
    uint32_t val[IPT], val_next[IPT], rk[IPT];
    auto load_tile = [&](uint32_t tl, KeyT (&k)[IPT], uint32_t (&v)[IPT]) {
        const uint32_t b0 = tl * TILE;
        const uint32_t tn = tl < t1 ? min(TILE, n - b0) : 0u;
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t slot = (uint32_t)(w * (64 * IPT) + r * 64 + l);
            const bool valid = slot < tn;
            k[r] = valid ? load_once(keys_in + b0 + slot) : (KeyT)0;
            v[r] = 0u;
            if (VALUES) v[r] = valid ? load_once(vals_in + b0 + slot) : 0u;
        }
    };
    if (aux_zero != 0u && blockIdx.x == 0)
        for (uint32_t j = (uint32_t)t; j < aux_zero; j += RS_THREADS) aux[j] = 0u;
    if (t0 >= t1) return;
    load_tile(t0, key, val);
    // digit totals (all group rows) -> digit bases; + the items of lower workgroups (lower groups' rows, then the lower rows of
    // the own group).  Thread = digit: every read is a coalesced 1 KB row.
    // All row reads of a batch are issued back to back (fixed trip counts, clamped row index): one memory latency per batch, not
    // one per row — this prologue is on the critical path of every workgroup.
    uint32_t below = 0u;
    if (t < RADIX_BINS) {
        uint32_t total = 0u;
        if (t < nd) {
            const uint32_t ngroups = (gridDim.x + RADIX_GROUP - 1) / RADIX_GROUP, gq = blockIdx.x / RADIX_GROUP;
            for (uint32_t j0 = 0; j0 < ngroups; j0 += 32u) {
                uint32_t v[32];
#pragma unroll
                for (uint32_t k = 0; k < 32u; ++k) v[k] = groups[(size_t)min(j0 + k, ngroups - 1u) * RADIX_BINS + t];
#pragma unroll
                for (uint32_t k = 0; k < 32u; ++k) {
                    const uint32_t j = j0 + k;
                    total += j < ngroups ? v[k] : 0u;
                    below += j < gq ? v[k] : 0u;
                }
            }
            {
                const uint32_t first = gq * RADIX_GROUP;      // rows [first, blockIdx.x) of the own group; RADIX_GROUP - 1 at most
                uint32_t v[RADIX_GROUP - 1];
#pragma unroll
                for (uint32_t k = 0; k < RADIX_GROUP - 1; ++k) v[k] = counts[(size_t)min(first + k, (uint32_t)blockIdx.x) * RADIX_BINS + t];
#pragma unroll
                for (uint32_t k = 0; k < RADIX_GROUP - 1; ++k) below += first + k < blockIdx.x ? v[k] : 0u;
            }
        }
        sh.tilecnt[t] = total;
    }
    __syncthreads();
    scan256_excl(sh.tilecnt, sh.next);
    __syncthreads();
    if (t < nd) sh.next[t] += below;

    for (uint32_t tile = t0; tile < t1; ++tile) {
        const uint32_t base = tile * TILE;
        const uint32_t tile_n = min(TILE, n - base);
        // ---- 1: rank (wave w owns slots [w*64*IPT, (w+1)*64*IPT) of the tile, 64 per round) ------------------------------
#pragma unroll
        for (int k = l; k < RADIX_BINS; k += 64) {
            sh.wcnt[w][k] = 0u;
#ifndef GSPL_RS_MATCH_BALLOT
            sh.match[w][k] = 0ull;
#endif
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t slot = (uint32_t)(w * (64 * IPT) + r * 64 + l);
            const bool valid = slot < tile_n;
            const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
#ifdef GSPL_RS_MATCH_BALLOT
            unsigned long long peers = __ballot(valid);
            for (int b = 0; b < nbits; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(valid && bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t cnt = (uint32_t)__builtin_popcountll(peers);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            uint32_t old = 0u;
            if (valid && below == 0u) { old = sh.wcnt[w][d]; sh.wcnt[w][d] = old + cnt; }
            const int leader = valid ? (int)__builtin_ctzll(peers) : l;
            old = __shfl(old, leader);
            rk[r] = old + below;
#else
            // A wave whose items all carry ONE digit (the top byte of depth keys: a handful of values for a whole frame) needs no
            // matching: the rank is the lane's position among the valid lanes.  64 lanes OR-ing into one LDS word serialise — the
            // last depth pass ran 40 % longer than the others (16.4 against 11.5 us at 1 M keys, 115 against 72 at 6 M).
            const unsigned long long act = __ballot(valid);
            const int first = act ? (int)__builtin_ctzll(act) : 0;
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
            if (act != 0ull && __ballot(valid && d != d0) == 0ull) {
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
                const uint32_t old = sh.wcnt[w][d0];
                __builtin_amdgcn_wave_barrier();
                if (l == first) sh.wcnt[w][d0] = old + (uint32_t)__builtin_popcountll(act);
                rk[r] = old + below;
            } else {
            // LDS operations of one wave execute in program order: every lane's OR lands before the reads below
            unsigned long long peers = 0ull;
            uint32_t old = 0u;
            if (valid) {
                atomicOr(&sh.match[w][d], 1ull << l);
                __builtin_amdgcn_wave_barrier();
                peers = sh.match[w][d];
                old = sh.wcnt[w][d];
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
            if (valid && below == 0u) {                 // the digit's first lane: advance the counter, clear the mask for the next round
                sh.wcnt[w][d] = old + (uint32_t)__builtin_popcountll(peers);
                sh.match[w][d] = 0ull;
            }
            rk[r] = old + below;
            }
#endif
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // ---- 2: tile histogram, in-tile digit starts ---------------------------------------------------------------------
        uint32_t my_count = 0u;
        if (t < nd) {
            uint32_t run = 0u;
#pragma unroll
            for (int k = 0; k < RS_WAVES; ++k) { const uint32_t c = sh.wcnt[k][t]; sh.wcnt[k][t] = run; run += c; }
            my_count = run;
        }
        if (t < RADIX_BINS) sh.tilecnt[t] = my_count;
        __syncthreads();
        scan256_excl(sh.tilecnt, sh.dstart);
        __syncthreads();
        // ---- 3: permute through LDS; the next tile's loads go out before the write-out ---------------------------------------
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t slot = (uint32_t)(w * (64 * IPT) + r * 64 + l);
            if (slot < tile_n) {
                const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
                const uint32_t pos = sh.dstart[d] + sh.wcnt[w][d] + rk[r];
                sh.xkey[pos] = key[r];
                if (VALUES) sh.xval[pos] = val[r];
            }
        }
        if (t < nd) {
            const uint32_t nx = sh.next[t];
            sh.gbase[t] = nx - sh.dstart[t];
            sh.next[t] = nx + my_count;
        }
        load_tile(tile + 1, key_next, val_next);
        __syncthreads();
        // ---- 4: write out ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t i = (uint32_t)(r * RS_THREADS + t);
            if (i < tile_n) {
                const KeyT k = sh.xkey[i];
                const uint32_t d = (uint32_t)(k >> shift) & mask;
                const uint32_t gpos = sh.gbase[d] + i;
                if constexpr (FINAL == FINAL_GATHER) {
                    const uint32_t v = sh.xval[i];
                    vals_out[gpos] = v;
                    keys_out[gpos] = (KeyT)gather[v];
                } else if constexpr (FINAL == FINAL_TILES) {
                    vals_out[gpos] = (uint32_t)k;
                    const uint32_t id = (uint32_t)((unsigned long long)k >> 32);
                    const bool run_first = (i == 0u) || ((uint32_t)((unsigned long long)sh.xkey[i - 1] >> 32) != id);
                    const bool run_last = (i + 1u == tile_n) || ((uint32_t)((unsigned long long)sh.xkey[i + 1] >> 32) != id);
                    if (run_first && run_last) atomicAdd(aux + id, 1u);
                    else if (run_first) atomicSub(aux + id, i);
                    else if (run_last) atomicAdd(aux + id, i + 1u);
                } else {
                    keys_out[gpos] = k;
                    if (VALUES) vals_out[gpos] = sh.xval[i];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < IPT; ++r) { key[r] = key_next[r]; val[r] = val_next[r]; }
        __syncthreads();
    }
}

// Clears the tables of a sort (16-byte multiples): an ordinary kernel, which follows the previous kernel of the stream without
// the ~6 us hand-over a memset command costs on either side (profiles/r02e_sequence.txt: 5 us fill + 6-8 us gaps).
__global__ __launch_bounds__(256) void radix_zero_kernel(uint4* __restrict__ p, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int radix_zero(void* p, size_t bytes, void* stream) {
    if (bytes == 0) return GSPL_OK;
    if (bytes % 16 != 0 || ((uintptr_t)p & 15u) != 0) return fail_arg("radix_zero: not 16-byte aligned");
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(radix_zero_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint4*)p, n16);
    return check_launch("radix_zero");
}

// Workgroups a pass runs with: GSPL_RS_WG_PER_CU per CU, never more than there are tiles or than RADIX_MAX_WG.
static unsigned pass_workgroups(uint32_t ntiles) {
    static unsigned cus = 0;
    if (cus == 0) {
        int dev = 0, c = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) c = 256;
        (void)hipGetLastError();
        cus = (unsigned)c;
    }
    unsigned g = cus * (unsigned)GSPL_RS_WG_PER_CU;
    if (const char* e = getenv("GSPL_SORT_WORKGROUPS")) { const int v = atoi(e); if (v >= 1) g = (unsigned)v; }
    if (g > (unsigned)RADIX_MAX_WG) g = (unsigned)RADIX_MAX_WG;
    if (g > ntiles) g = ntiles;
    return g ? g : 1u;
}

static void plan_items(RadixPlan& plan, size_t n) {
    plan.n = (uint32_t)n;
    plan.ntiles = (uint32_t)((n + plan.tile_items - 1) / plan.tile_items);
    plan.nwg = pass_workgroups(plan.ntiles);
    plan.tiles_per_wg = (plan.ntiles + plan.nwg - 1) / plan.nwg;
    if (plan.tiles_per_wg == 0) plan.tiles_per_wg = 1;
    plan.nwg = plan.ntiles ? (plan.ntiles + plan.tiles_per_wg - 1) / plan.tiles_per_wg : 1u;      // no empty workgroups at the end
}

bool radix_plan(size_t n, int begin_bit, int end_bit, int digit_bits, int tile_items, RadixPlan& plan) {
    if (n > RADIX_MAX_ITEMS || begin_bit < 0 || end_bit <= begin_bit || digit_bits < 1 || digit_bits > 8 || tile_items <= 0) return false;
    const int bits = end_bit - begin_bit;
    const int passes = (bits + digit_bits - 1) / digit_bits;
    if (passes > RADIX_MAX_PASSES) return false;
    plan.passes = passes;
    int at = begin_bit;
    for (int p = 0; p < passes; ++p) {
        const int left = end_bit - at, todo = passes - p;
        const int b = (left + todo - 1) / todo;
        plan.shift[p] = at;
        plan.bits[p] = b;
        at += b;
    }
    for (int p = passes; p < RADIX_MAX_PASSES; ++p) { plan.shift[p] = 0; plan.bits[p] = 0; }
    plan.tile_items = (uint32_t)tile_items;
    plan_items(plan, n);
    // tables sized for THIS item count (replanning for fewer items keeps them): group rows of every pass, the pass-0 rows a
    // preparing producer accumulates (both zero before use), the rows of the other passes
    plan.wg_cap = plan.nwg;
    const size_t rows = (size_t)RADIX_BINS * sizeof(uint32_t);
    plan.groups_off = 0;
    plan.groups_bytes = (size_t)RADIX_MAX_PASSES * ((plan.wg_cap + RADIX_GROUP - 1) / RADIX_GROUP) * rows;
    plan.counts0_off = plan.groups_bytes;
    plan.header_bytes = plan.counts0_off + plan.wg_cap * rows;
    plan.counts_off = plan.header_bytes;
    plan.total_bytes = plan.counts_off + plan.wg_cap * rows;
    return true;
}

// Same bit split, workspace and SPANS (keys per workgroup: a producer may already have counted by them), fewer items.
void radix_replan_items(RadixPlan& plan, size_t n) {
    plan.n = (uint32_t)n;
    plan.ntiles = (uint32_t)((n + plan.tile_items - 1) / plan.tile_items);
    plan.nwg = plan.ntiles ? (plan.ntiles + plan.tiles_per_wg - 1) / plan.tiles_per_wg : 1u;
}

void radix_producer_args(const RadixPlan& plan, void* workspace, RadixProducer& rp) {
    char* ws = (char*)workspace;
    rp.counts0 = (uint32_t*)(ws + plan.counts0_off);
    rp.groups0 = (uint32_t*)(ws + plan.groups_off);
    rp.span_items = plan.tile_items * plan.tiles_per_wg;
    rp.shift = plan.shift[0];
    rp.mask = (1u << plan.bits[0]) - 1u;
}

template <typename KeyT, int IPT>
static int radix_sort_impl(const RadixPlan& plan, void* workspace, KeyT* const keys[2], uint32_t* const vals[2], bool prepared, void* stream,
                           uint32_t* final_ids = nullptr, uint32_t* tile_counts = nullptr, uint32_t n_tile_counts = 0,
                           const uint32_t* gather = nullptr, const int64_t* n_dev = nullptr) {
    if (plan.n == 0) return GSPL_OK;
    if (plan.tile_items != (uint32_t)(RS_THREADS * IPT)) return fail_arg("radix_sort: plan made for another tile size");
    if (plan.nwg > plan.wg_cap) return fail_arg("radix_sort: plan replanned for more items than its tables hold");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const size_t group_rows = (plan.wg_cap + RADIX_GROUP - 1) / RADIX_GROUP;
    if (!prepared) {
        int rc = radix_zero(ws + plan.groups_off, plan.groups_bytes, s);
        if (rc != GSPL_OK) return rc;
    }
    for (int p = 0; p < plan.passes; ++p) {
        const KeyT* kin = keys[p & 1];
        const uint32_t* vin = vals ? vals[p & 1] : nullptr;
        KeyT* kout = keys[(p + 1) & 1];
        uint32_t* groups = (uint32_t*)(ws + plan.groups_off) + (size_t)p * group_rows * RADIX_BINS;
        uint32_t* counts = (uint32_t*)(ws + ((p == 0 && prepared) ? plan.counts0_off : plan.counts_off));
        if (!(p == 0 && prepared))
            hipLaunchKernelGGL((radix_count_kernel<KeyT, IPT>), dim3(plan.nwg), dim3(RS_THREADS), 0, s, kin, plan.n, plan.ntiles, plan.tiles_per_wg,
                               plan.shift[p], plan.bits[p], counts, groups, n_dev);
        const bool last = p == plan.passes - 1;
        uint32_t* aux = final_ids ? tile_counts : nullptr;
        const uint32_t aux_zero = (final_ids && p == plan.passes - 2) ? n_tile_counts : 0u;
#define GSPL_SCATTER(VALUES, FINAL, VIN, VOUT)                                                                                              \
        hipLaunchKernelGGL((radix_scatter_kernel<KeyT, VALUES, IPT, FINAL>), dim3(plan.nwg), dim3(RS_THREADS), 0, s, kin, VIN, kout, VOUT, plan.n, \
                           plan.ntiles, plan.tiles_per_wg, plan.shift[p], plan.bits[p], groups, counts, aux, aux_zero, gather, n_dev)
        if (final_ids && last) {
            if constexpr (sizeof(KeyT) == 8) GSPL_SCATTER(false, FINAL_TILES, nullptr, final_ids);
        } else if (gather && last) {
            if constexpr (sizeof(KeyT) == 4) GSPL_SCATTER(true, FINAL_GATHER, vin, vals[(p + 1) & 1]);
        } else if (vals) {
            GSPL_SCATTER(true, FINAL_NONE, vin, vals[(p + 1) & 1]);
        } else {
            GSPL_SCATTER(false, FINAL_NONE, nullptr, nullptr);
        }
#undef GSPL_SCATTER
        int rc = check_launch("radix_sort(pass)");
        if (rc != GSPL_OK) return rc;
    }
    return GSPL_OK;
}

int radix_sort_u32(const RadixPlan& plan, void* workspace, uint32_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream,
                   const uint32_t* gather) {
    if (gather && !vals) return fail_arg("radix_sort_u32: gather needs values");
    return radix_sort_impl<uint32_t, RADIX_TILE_U32 / RS_THREADS>(plan, workspace, keys, vals, prepared, stream, nullptr, nullptr, 0, gather);
}
int radix_sort_u64(const RadixPlan& plan, void* workspace, uint64_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream) {
    return radix_sort_impl<uint64_t, RADIX_TILE_U64 / RS_THREADS>(plan, workspace, keys, vals, prepared, stream);
}
// The tile sort of the binning: u64 records (tile id << 32 | splat id) sorted on the tile-id bits; the sorted low words go straight
// to `ids_out`, `tile_counts[0, n_tile_counts)` receives the number of records of every tile id.  plan.passes >= 2 (the pass
// before the last clears the counters).
int radix_sort_tiles(const RadixPlan& plan, void* workspace, uint64_t* const keys[2], bool prepared, uint32_t* ids_out,
                     uint32_t* tile_counts, uint32_t n_tile_counts, void* stream, const int64_t* n_dev) {
    if (plan.passes < 2) return fail_arg("radix_sort_tiles: at least two passes");
    return radix_sort_impl<uint64_t, RADIX_TILE_U64 / RS_THREADS>(plan, workspace, keys, nullptr, prepared, stream, ids_out, tile_counts, n_tile_counts,
                                                                  nullptr, n_dev);
}

// counts[0, n) (u32) -> exclusive prefix in place.  One workgroup: n is the number of image tiles (8160 at 1080p) or of scan blocks.
__global__ __launch_bounds__(1024) void tile_offsets_kernel(uint32_t* __restrict__ counts, uint32_t n) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    if (t == 0) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 4096u) {
        const uint32_t i0 = base + (uint32_t)t * 4u;
        uint32_t v[4], mine = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = (i0 + k < n) ? counts[i0 + k] : 0u; mine += v[k]; }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if (l >= d) incl += up; }
        if (l == 63) s_wave[w] = incl;
        __syncthreads();
        uint32_t woff = 0u, total = 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint32_t c = s_wave[k]; if (k < w) woff += c; total += c; }
        uint32_t run = s_carry + woff + (incl - mine);
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (i0 + k < n) counts[i0 + k] = run; run += v[k]; }
        __syncthreads();
        if (t == 0) s_carry += total;
        __syncthreads();
    }
}

int tile_offsets_from_counts(uint32_t* counts, uint32_t n, void* stream) {
    if (n == 0) return GSPL_OK;
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, n);
    return check_launch("tile_offsets");
}

// ---- plain exclusive scan of u32 (the cell histogram of knn.hip; a one-off initialisation op) ---------------------------------
// blocks of 4096 items: block sums -> exclusive scan of the sums by one workgroup -> per-block scan + base.
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t s_wave[16];
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const uint32_t i0 = blockIdx.x * 4096u + (uint32_t)t * 4u;
    uint32_t mine = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) mine += (i0 + k < n) ? in[i0 + k] : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
    if (l == 0) s_wave[w] = mine;
    __syncthreads();
    if (t == 0) { uint32_t tot = 0u; for (int k = 0; k < 16; ++k) tot += s_wave[k]; sums[blockIdx.x] = tot; }
}
__global__ __launch_bounds__(1024) void scan_blocks_kernel(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ bases,
                                                           uint32_t* __restrict__ out) {
    __shared__ uint32_t s_wave[16];
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const uint32_t i0 = blockIdx.x * 4096u + (uint32_t)t * 4u;
    uint32_t v[4], mine = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (i0 + k < n) ? in[i0 + k] : 0u; mine += v[k]; }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if (l >= d) incl += up; }
    if (l == 63) s_wave[w] = incl;
    __syncthreads();
    uint32_t woff = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < w) woff += s_wave[k];
    uint32_t run = bases[blockIdx.x] + woff + (incl - mine);
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
}
size_t exclusive_scan_u32_workspace_bytes(size_t n) { return ((n + 4095) / 4096 + 1) * sizeof(uint32_t); }
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, void* workspace, void* stream) {
    if (n == 0) return GSPL_OK;
    const unsigned blocks = (unsigned)((n + 4095) / 4096);
    uint32_t* sums = (uint32_t*)workspace;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(blocks), dim3(1024), 0, s, in, (uint32_t)n, sums);
    hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, s, sums, blocks);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(blocks), dim3(1024), 0, s, in, (uint32_t)n, (const uint32_t*)sums, out);
    return check_launch("exclusive_scan_u32");
}

// ---- scan of the tile counts in depth order -----------------------------------------------------------------------------------
// Inclusive scan of counts[order[i]] (int32 -> int64): block sums, [one-workgroup scan of the sums,] per-block scan (gspl_sort.h).
// Bit 31 of a count tags the item: the tagged items are ranked along the way (their number rides in bits 40.. of the scanned
// value; the counts themselves add up to < 2^31).
static constexpr int SC_IPT = SCAN_TILE / RS_THREADS;

__device__ __forceinline__ unsigned long long scan_item(const uint32_t* __restrict__ order, const int32_t* __restrict__ counts, uint32_t i, uint32_t n) {
    const uint32_t c = i < n ? (uint32_t)counts[order ? order[i] : i] : 0u;
    return (unsigned long long)(c & 0x7fffffffu) | ((unsigned long long)(c >> 31) << 40);
}

__global__ __launch_bounds__(RS_THREADS) void scan_gather_sums_kernel(const uint32_t* __restrict__ order, const int32_t* __restrict__ counts,
                                                                      uint32_t n, unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long s_wave[RS_WAVES];
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const uint32_t first = blockIdx.x * (uint32_t)SCAN_TILE + (uint32_t)t * SC_IPT;
    unsigned long long mine = 0ull;
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) mine += scan_item(order, counts, first + k, n);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
    if (l == 0) s_wave[w] = mine;
    __syncthreads();
    if (t == 0) { unsigned long long tot = 0ull; for (int k = 0; k < RS_WAVES; ++k) tot += s_wave[k]; sums[blockIdx.x] = tot; }
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(unsigned long long* __restrict__ sums, uint32_t n) {      // exclusive, in place
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    if (t == 0) s_carry = 0ull;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024u) {
        const uint32_t i = base + (uint32_t)t;
        const unsigned long long v = i < n ? sums[i] : 0ull;
        unsigned long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(incl, d); if (l >= d) incl += up; }
        if (l == 63) s_wave[w] = incl;
        __syncthreads();
        unsigned long long woff = 0ull, total = 0ull;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const unsigned long long c = s_wave[k]; if (k < w) woff += c; total += c; }
        if (i < n) sums[i] = s_carry + woff + (incl - v);
        __syncthreads();
        if (t == 0) s_carry += total;
        __syncthreads();
    }
}

// RAW_SUMS: `bases` holds the block sums as scan_gather_sums_kernel left them, and every workgroup adds up the ones in front of
// it itself (integer sums: the same numbers whichever way they are added) — for a few hundred workgroups that is cheaper than a
// one-workgroup scan launch in between.
template <bool RAW_SUMS>
__global__ __launch_bounds__(RS_THREADS) void scan_gather_kernel(const uint32_t* __restrict__ order, const int32_t* __restrict__ counts,
                                                                 int64_t* __restrict__ cum, uint32_t n, const unsigned long long* __restrict__ bases,
                                                                 int32_t* __restrict__ tagged_list, int64_t* __restrict__ host_words,
                                                                 unsigned long long ticket, uint4* __restrict__ zero_p, uint32_t zero_n16) {
    __shared__ unsigned long long s_wave[RS_WAVES];
    __shared__ unsigned long long s_front[RS_WAVES];
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    zero_table(zero_p, zero_n16);          // ZeroJob: the tables of the sort whose keys the NEXT kernel of the stream produces
    unsigned long long base;
    if (RAW_SUMS) {
        unsigned long long part = 0ull;
        for (uint32_t j = (uint32_t)t; j < blockIdx.x; j += (uint32_t)RS_THREADS) part += bases[j];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
        if (l == 0) s_front[w] = part;
    }
    const uint32_t first = blockIdx.x * (uint32_t)SCAN_TILE + (uint32_t)t * SC_IPT;       // SC_IPT consecutive items per thread
    unsigned long long v[SC_IPT], mine = 0ull;
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) { v[k] = scan_item(order, counts, first + k, n); mine += v[k]; }
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(incl, d); if (l >= d) incl += up; }
    if (l == 63) s_wave[w] = incl;
    __syncthreads();
    unsigned long long wave_off = 0ull;
#pragma unroll
    for (int k = 0; k < RS_WAVES; ++k) if (k < w) wave_off += s_wave[k];
    if (RAW_SUMS) {
        base = 0ull;
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) base += s_front[k];      // written before the __syncthreads above
    } else {
        base = bases[blockIdx.x];
    }
    unsigned long long run = base + wave_off + (incl - mine);
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) {
        run += v[k];
        const uint32_t i = first + k;
        if (i < n) {
            cum[i] = (int64_t)(run & ((1ull << 40) - 1ull));
            if (tagged_list) {
                if (v[k] >> 40) tagged_list[(run >> 40) - 1ull] = (int32_t)i;
                if (i == n - 1u) cum[n] = (int64_t)(run >> 40);      // how many are tagged
            }
            if (host_words && i == n - 1u) {                         // total and tagged count straight into host memory (no copy launch)
                host_words[0] = (int64_t)(run & ((1ull << 40) - 1ull));
                host_words[1] = (int64_t)(run >> 40);
                __threadfence_system();
                // ticket != 0: the host POLLS host_words[2] for it instead of waiting for an event (an event record between two
                // kernels of one stream is a ~6 us bubble on this part: profiles/r05d_sequence.txt, the gaps in front of radix_zero)
                if (ticket) __hip_atomic_store((unsigned long long*)host_words + 2, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

size_t scan_workspace_bytes(size_t n) {
    const size_t blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    return ((blocks ? blocks : 1) * sizeof(unsigned long long) + 15) / 16 * 16;
}

int scan_gathered_counts(const uint32_t* order, const int32_t* counts, int64_t* cum, size_t n, void* workspace, int32_t* tagged_list, void* stream,
                         int64_t* host_words, unsigned long long ticket, ZeroJob then_zero) {
    if (n == 0) return then_zero.n16 ? radix_zero(then_zero.p, (size_t)then_zero.n16 * 16, stream) : GSPL_OK;
    const unsigned blocks = (unsigned)((n + SCAN_TILE - 1) / SCAN_TILE);
    unsigned long long* sums = (unsigned long long*)workspace;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(scan_gather_sums_kernel, dim3(blocks), dim3(RS_THREADS), 0, s, order, counts, (uint32_t)n, sums);
    if (blocks <= SCAN_RAW_SUMS_BLOCKS) {
        // two launches: every workgroup reads the (at most 1024) sums in front of it from L2
        hipLaunchKernelGGL(scan_gather_kernel<true>, dim3(blocks), dim3(RS_THREADS), 0, s, order, counts, cum, (uint32_t)n, (const unsigned long long*)sums,
                           tagged_list, host_words, ticket, then_zero.p, then_zero.n16);
    } else {
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, sums, blocks);
        hipLaunchKernelGGL(scan_gather_kernel<false>, dim3(blocks), dim3(RS_THREADS), 0, s, order, counts, cum, (uint32_t)n, (const unsigned long long*)sums,
                           tagged_list, host_words, ticket, then_zero.p, then_zero.n16);
    }
    return check_launch("scan_gathered_counts");
}

}  // namespace gspl

// ---- C-ABI (declared in include/gspl_hip.h) ---------------------------------------------------------------------------
extern "C" size_t gspl_radix_sort_workspace_bytes(int64_t n, int key_bytes, int begin_bit, int end_bit) {
    gspl::RadixPlan plan;
    if (n < 0 || (key_bytes != 4 && key_bytes != 8)) return 0;
    if (!gspl::radix_plan((size_t)n, begin_bit, end_bit, 8, key_bytes == 4 ? gspl::RADIX_TILE_U32 : gspl::RADIX_TILE_U64, plan)) return 0;
    return plan.total_bytes;
}

static int sort_args(int64_t n, int key_bits, int begin_bit, int end_bit, const void* k0, const void* k1, const void* ws, size_t ws_bytes,
                     int* result_buffer, gspl::RadixPlan& plan, const char* who) {
    using namespace gspl;
    if (n < 0 || begin_bit < 0 || end_bit > key_bits || end_bit <= begin_bit || !result_buffer) return fail_arg(who);
    if ((size_t)n > RADIX_MAX_ITEMS) { set_error(who, "more than 2^30-1 items"); return GSPL_ERR_UNSUPPORTED; }
    if (!radix_plan((size_t)n, begin_bit, end_bit, 8, key_bits == 32 ? RADIX_TILE_U32 : RADIX_TILE_U64, plan)) {
        set_error(who, "more than 32 key bits selected");
        return GSPL_ERR_UNSUPPORTED;
    }
    *result_buffer = plan.passes & 1;
    if (n == 0) return GSPL_OK;
    if (!k0 || !k1 || !ws) return fail_arg(who);
    if (ws_bytes < plan.total_bytes) return fail_ws(who);
    return GSPL_OK;
}

extern "C" int gspl_radix_sort_pairs_u32(int64_t n, uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1,
                                         int begin_bit, int end_bit, int* result_buffer,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    gspl::RadixPlan plan;
    int rc = sort_args(n, 32, begin_bit, end_bit, keys0, keys1, workspace, workspace_bytes, result_buffer, plan, "radix_sort_pairs_u32");
    if (rc != GSPL_OK || n == 0) return rc;
    if (!vals0 || !vals1) return gspl::fail_arg("radix_sort_pairs_u32: NULL values");
    uint32_t* const keys[2] = {keys0, keys1};
    uint32_t* const vals[2] = {vals0, vals1};
    return gspl::radix_sort_u32(plan, workspace, keys, vals, false, stream);
}

extern "C" int gspl_radix_sort_keys_u64(int64_t n, uint64_t* keys0, uint64_t* keys1, int begin_bit, int end_bit, int* result_buffer,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    gspl::RadixPlan plan;
    int rc = sort_args(n, 64, begin_bit, end_bit, keys0, keys1, workspace, workspace_bytes, result_buffer, plan, "radix_sort_keys_u64");
    if (rc != GSPL_OK || n == 0) return rc;
    uint64_t* const keys[2] = {keys0, keys1};
    return gspl::radix_sort_u64(plan, workspace, keys, nullptr, false, stream);
}
