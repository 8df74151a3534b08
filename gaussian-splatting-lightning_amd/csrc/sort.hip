// sort.hip — stable one-sweep LSD radix sort for the binning stage (gfx950, wave64).  Interface and rationale: gspl_sort.h.
//
// Replaces (inside gspl_bin_count / gspl_bin_emit_sort) the device radix sorts the reference's native rasterizers call
// between projection and compositing: gsplat `isect_tiles` -> cub::DeviceRadixSort::SortPairs and the Inria rasterizer's
// `cub::DeviceRadixSort::SortPairs(point_list_keys...)` (call sites: reference gsplat_v1_renderer.py:524-556,
// vanilla_renderer.py:111).  Ordering contract: stable, ascending on the selected key bits — identical to those.
//
// Pass kernel (8 waves; a tile = 512 x IPT items, held in (wave, round, lane) = memory order); one tile per workgroup when the
// grid is certainly co-resident, tiles drawn from a counter otherwise (see radix_pass_kernel).  Per tile:
//   1. load; in-wave ranks by digit matching (one ballot per digit bit), per-wave digit counters in LDS
//   2. counters -> tile histogram -> publish LOCAL|count per digit; exclusive scan -> first in-tile slot per digit
//   3. permute the tile through LDS into digit order
//   4. look-back per digit over the preceding tiles' state words (LOCAL: add and go on, GLOBAL: add and stop);
//      publish GLOBAL|inclusive
//   5. write out: consecutive lanes hold consecutive items of a digit run -> runs of consecutive addresses
#include "gspl_device.h"
#include "gspl_host.h"
#include "gspl_sort.h"
#include "gspl_sort_device.h"
#include <cstdio>
#include <cstdlib>

namespace gspl {

static constexpr int RS_WAVES = 8;
static constexpr int RS_THREADS = RS_WAVES * 64;

#ifndef GSPL_RS_WINDOW
#define GSPL_RS_WINDOW 8
#endif
static constexpr int RS_WINDOW = GSPL_RS_WINDOW;         // look-back state loads kept in flight per step
#ifndef GSPL_RS_GROUP
#define GSPL_RS_GROUP 16
#endif
static constexpr uint32_t RS_GROUP = GSPL_RS_GROUP;           // tiles per look-back group
static constexpr uint32_t RS_FLAG_LOCAL = 1u << 30;
static constexpr uint32_t RS_FLAG_GLOBAL = 2u << 30;
static constexpr uint32_t RS_COUNT_MASK = (1u << 30) - 1u;

__device__ __forceinline__ uint32_t state_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void state_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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

// Header kernel, for sorts whose keys were not produced by one of our kernels: gspl_sort_device.h applied to an array.
template <typename KeyT>
__global__ __launch_bounds__(RS_THREADS) void radix_header_kernel(const KeyT* __restrict__ keys, uint32_t n, RadixHeader hdr) {
    __shared__ uint32_t h[RADIX_MAX_PASSES * RADIX_BINS];
    const int t = threadIdx.x;
    radix_hist_clear(h);
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * RS_THREADS;
    const size_t rounds = ((size_t)n + stride - 1) / stride;
    for (size_t r = 0; r < rounds; ++r) {
        const size_t i = r * stride + (size_t)blockIdx.x * RS_THREADS + t;
        const bool valid = i < n;
        radix_hist_add<KeyT>(h, hdr, valid ? keys[i] : (KeyT)0, valid);
    }
    __syncthreads();
    radix_hist_flush(h, hdr);
    radix_states_clear(hdr, (size_t)blockIdx.x * RS_THREADS + t, stride);
}

// Walk the state rows `p`, p-1, ..., lo of one digit column: add LOCAL counts and go on, add a GLOBAL count and finish;
// an unpublished row is polled.  RS_WINDOW rows are fetched per step (a state load is a round trip past the XCD L2).
// Returns the first row not consumed (lo - 1 when the range is exhausted).
// A predecessor that never publishes (only possible in the counter-free mode when the grid is neither co-resident nor
// dispatched in index order) does not hang or fault the device: after 2^22 polls the walk gives up, raises the sort's error
// word and carries on with what it has — the output is then garbage, every workgroup still terminates, and the host turns the
// word into a status / re-runs the sort with tiles drawn from a counter (gspl_bin_count reports it through cum_tiles[N + 1]).
__device__ uint32_t g_sort_error_sink;
__device__ __forceinline__ int walk_states(const uint32_t* __restrict__ rows, int t, int p, int lo, uint32_t& excl, bool& finished,
                                           uint32_t* __restrict__ err) {
    uint32_t spins = 0u;
    while (!finished && p >= lo) {
        uint32_t s[RS_WINDOW];
#pragma unroll
        for (int q = 0; q < RS_WINDOW; ++q) s[q] = (p - q >= lo) ? state_load(rows + (size_t)(p - q) * RADIX_BINS + t) : 0u;
        int adv = 0;
        bool stop = false;
#pragma unroll
        for (int q = 0; q < RS_WINDOW; ++q) {
            const uint32_t flag = s[q] >> 30;
            if (!stop) {
                if (flag == 0u) stop = true;                      // not published yet (or past the range): re-read from here
                else { excl += s[q] & RS_COUNT_MASK; ++adv; if (flag == 2u) { stop = true; finished = true; } }
            }
        }
        p -= adv;
        if (adv == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { atomicOr(err, 1u); finished = true; }      // a predecessor never published: flag it, do not hang
        }
    }
    return p;
}

template <typename KeyT, bool VALUES, int IPT>
struct RadixShared {
    uint32_t wcnt[RS_WAVES][RADIX_BINS];      // per-wave digit counters, then exclusive prefix over the waves
    uint32_t histo[RADIX_BINS];               // the pass's global histogram
    uint32_t dbase[RADIX_BINS];               // global position of the first item of each digit
    uint32_t tilecnt[RADIX_BINS];             // the tile's digit histogram
    uint32_t dstart[RADIX_BINS];              // first in-tile slot of each digit
    uint32_t gbase[RADIX_BINS];               // global position of in-tile slot 0 as seen by each digit (modular)
    KeyT xkey[RS_THREADS * IPT];
    uint32_t xval[VALUES ? RS_THREADS * IPT : 1];
};

// TICKET = false: one tile per workgroup (tile = blockIdx.x).  A tile waits for lower tiles only, so the launch makes progress
//   whenever the lowest unfinished tile is resident: always when the whole grid is co-resident (the launcher checks that it
//   fits a device that is otherwise idle), and under contention as long as workgroups start in index order (observed;
//   not promised — the look-back traps after 2^24 polls instead of hanging).  No counter, no queueing: the fast path of
//   the 1 M-splat depth sort.
// TICKET = true: workgroups draw tiles from a counter until it runs out, so every tile below a drawn one belongs to a
//   RUNNING workgroup whatever the dispatch order, the residency or the other kernels on the device (a resident grid
//   looping over tiles b, b + grid, ... was measured to dead-lock when three processes shared the GPU: resident
//   workgroups waited for tiles of workgroups that could not start).  The draw for the next tile is issued a tile ahead; a
//   single-address atomic is served at ~60 M/s, which bounds this mode at ~16 ns per tile.
// FINAL (u64 keys, no values; the LAST pass of the tile sort of the binning): the sorted records leave as their low words only
//   (vals_out = the per-tile lists of splat ids), and the number of records per tile id (the key's high word) is counted into
//   aux[tile id].  The pass before left the records ordered by the low digit, so after the in-tile permutation equal tile ids are
//   contiguous runs in LDS: one atomic pair per run (subtract its first slot, add one past its last).
// aux_zero > 0: workgroup 0 clears aux[0, aux_zero) (the pass BEFORE the final one prepares the counters).
template <typename KeyT, bool VALUES, int IPT, bool TICKET, bool FINAL>
__global__ __launch_bounds__(RS_THREADS) void radix_pass_kernel(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                                uint32_t ntiles, int shift, int nbits, const uint32_t* __restrict__ hist,
                                                                uint32_t* __restrict__ states, uint32_t* __restrict__ gstates,
                                                                uint32_t* __restrict__ ticket, uint32_t* __restrict__ err,
                                                                uint32_t* __restrict__ aux, uint32_t aux_zero) {
    constexpr uint32_t TILE = RS_THREADS * IPT;
    __shared__ RadixShared<KeyT, VALUES, IPT> sh;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const uint32_t mask = (1u << nbits) - 1u;
    const int nd = 1 << nbits;
    // The first tile's loads go out before anything else; the pass's histogram (sum of the copies) follows them and its
    // exclusive scan (= the digit bases) rides on the first tile's barriers.  Later tiles are prefetched one ahead.
    KeyT key[IPT], key_next[IPT];
    uint32_t val[IPT], val_next[IPT], rk[IPT];
    auto load_tile = [&](uint32_t tl, KeyT (&k)[IPT], uint32_t (&v)[IPT]) {
        const uint32_t b0 = tl * TILE;
        const uint32_t tn = tl < ntiles ? min(TILE, n - b0) : 0u;
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t slot = (uint32_t)(w * (64 * IPT) + r * 64 + l);
            const bool valid = slot < tn;
            k[r] = valid ? keys_in[b0 + slot] : (KeyT)0;
            v[r] = 0u;
            if (VALUES) v[r] = valid ? vals_in[b0 + slot] : 0u;
        }
    };
    __shared__ uint32_t s_tile;
    uint32_t tile = blockIdx.x;
    if (TICKET) {
        if (t == 0) s_tile = atomicAdd(ticket, 1u);
        __syncthreads();
        tile = s_tile;
    }
    if (aux_zero != 0u && blockIdx.x == 0)
        for (uint32_t j = (uint32_t)t; j < aux_zero; j += RS_THREADS) aux[j] = 0u;
    load_tile(tile, key, val);
    if (t < RADIX_BINS) {
        uint32_t c = 0u;
        if (t < nd) {
#pragma unroll
            for (int k = 0; k < RADIX_HIST_COPIES; ++k) c += hist[k * (RADIX_MAX_PASSES * RADIX_BINS) + t];
        }
        sh.histo[t] = c;
    }
    bool first = true;

    while (tile < ntiles) {
        const uint32_t base = tile * TILE;
        uint32_t drawn = 0u;
        if (TICKET && t == 0) drawn = atomicAdd(ticket, 1u);      // next tile; the value is needed after the third barrier
        const uint32_t tile_n = min(TILE, n - base);
        // ---- 1: rank (wave w owns slots [w*64*IPT, (w+1)*64*IPT) of the tile, 64 per round) ------------------------------
#pragma unroll
        for (int k = l; k < RADIX_BINS; k += 64) sh.wcnt[w][k] = 0u;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t slot = (uint32_t)(w * (64 * IPT) + r * 64 + l);
            const bool valid = slot < tile_n;
            const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
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
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // ---- 2: tile histogram, publish LOCAL, in-tile digit starts ------------------------------------------------------
        uint32_t my_count = 0u;
        if (t < nd) {
            uint32_t run = 0u;
#pragma unroll
            for (int k = 0; k < RS_WAVES; ++k) { const uint32_t c = sh.wcnt[k][t]; sh.wcnt[k][t] = run; run += c; }
            my_count = run;
            state_store(states + (size_t)tile * RADIX_BINS + t, (tile == 0u ? RS_FLAG_GLOBAL : RS_FLAG_LOCAL) | run);
        }
        if (t < RADIX_BINS) sh.tilecnt[t] = my_count;
        __syncthreads();
        scan256_excl(sh.tilecnt, sh.dstart);
        if (first) scan256_excl(sh.histo, sh.dbase);
        first = false;
        if (TICKET && t == 0) s_tile = drawn;
        __syncthreads();
        // ---- 3: permute through LDS ------------------------------------------------------------------------------------
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
        const uint32_t next_tile = TICKET ? s_tile : ntiles;        // (no ticket: one tile per workgroup)
        load_tile(next_tile, key_next, val_next);                   // in flight during the look-back and the write-out
        // ---- 4: look-back ------------------------------------------------------------------------------------------------
        // All tiles of a co-resident grid start together, so a tile's predecessors are mostly LOCAL and a tile-by-tile walk
        // would cover ~tile/2 of them.  The last tile of every RS_GROUP tiles (the "closer") also publishes the group's
        // aggregate; a walk crosses its own group tile by tile and everything older group by group.
        if (t < nd) {
            uint32_t excl = 0u;
            if (tile > 0u) {
                bool finished = false;
                int p = (int)tile - 1;
                const bool closer = ((tile + 1u) % RS_GROUP) == 0u;
                if (((uint32_t)(p + 1) % RS_GROUP) != 0u) p = walk_states(states, t, p, (p / RS_GROUP) * RS_GROUP, excl, finished, err);
                if (closer) state_store(gstates + (size_t)(tile / RS_GROUP) * RADIX_BINS + t, (finished ? RS_FLAG_GLOBAL : RS_FLAG_LOCAL) | (excl + my_count));
                if (!finished) walk_states(gstates, t, (p + 1) / RS_GROUP - 1, 0, excl, finished, err);
                state_store(states + (size_t)tile * RADIX_BINS + t, RS_FLAG_GLOBAL | (excl + my_count));
                if (closer) state_store(gstates + (size_t)(tile / RS_GROUP) * RADIX_BINS + t, RS_FLAG_GLOBAL | (excl + my_count));
            }
            sh.gbase[t] = sh.dbase[t] + excl - sh.dstart[t];
        }
        __syncthreads();
        // ---- 5: write out ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t i = (uint32_t)(r * RS_THREADS + t);
            if (i < tile_n) {
                const KeyT k = sh.xkey[i];
                const uint32_t d = (uint32_t)(k >> shift) & mask;
                const uint32_t gpos = sh.gbase[d] + i;
                if constexpr (FINAL) {
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
        tile = next_tile;
        __syncthreads();
    }
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
    plan.n = (uint32_t)n;
    plan.tile_items = (uint32_t)tile_items;
    plan.ntiles = (uint32_t)((n + tile_items - 1) / tile_items);
    plan.hist_off = 0;
    plan.ticket_off = (size_t)RADIX_HIST_COPIES * RADIX_MAX_PASSES * RADIX_BINS * sizeof(uint32_t);
    plan.header_bytes = plan.ticket_off + 64;               // one tile counter per pass (+ one for a scan that shares the header)
    plan.states_off = plan.header_bytes;
    plan.ngroups = plan.ntiles / RS_GROUP;                 // only complete groups are ever walked over
    plan.total_bytes = plan.states_off + (size_t)passes * ((size_t)plan.ntiles + plan.ngroups + 1) * RADIX_BINS * sizeof(uint32_t);
    return true;
}

// Same bit split and workspace, fewer items (n <= the n the plan was made for): the rows of every pass start earlier but stay
// inside the range the plan's header covers.
void radix_replan_items(RadixPlan& plan, size_t n) {
    plan.n = (uint32_t)n;
    plan.ntiles = (uint32_t)((n + plan.tile_items - 1) / plan.tile_items);
    plan.ngroups = plan.ntiles / RS_GROUP;
}

// Workgroups of a kernel the device is SURE to keep resident at once (certain = true; the bound of the one-tile-per-workgroup
// launches) or is expected to (certain = false; the grid of the ticketed launches, where an overestimate is harmless).  The occupancy API
// is exact for most shapes (tools/micro/residency_probe.hip) but a kernel sitting on a register-file boundary (64 VGPRs =
// "8 waves per SIMD") was observed to get one wave per SIMD less than promised, and a grid that is not co-resident
// dead-locks the look-back: one workgroup per CU is taken off the promise (never below one per CU, which always fits).
template <typename Kernel>
static unsigned resident_blocks(Kernel kernel, bool certain = true) {
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RS_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1;
    (void)hipGetLastError();
    if (certain && per_cu > 1) --per_cu;
    if (const char* cap = getenv("GSPL_SORT_MAX_PER_CU")) { const int c = atoi(cap); if (c >= 1 && c < per_cu) per_cu = c; }
    if (getenv("GSPL_SORT_DEBUG")) fprintf(stderr, "[gspl sort] resident blocks per CU %d, CUs %d\n", per_cu, cus);
    return (unsigned)per_cu * (unsigned)cus;
}

// force_ticket: process-wide switch to the counter mode (set by the host after a look-back time-out, or by GSPL_SORT_FORCE_TICKET)
static bool g_force_ticket = false;
void radix_force_ticket(bool on) { g_force_ticket = on; }
static bool ticket_forced() { static const bool env = getenv("GSPL_SORT_FORCE_TICKET") != nullptr; return env || g_force_ticket; }

struct PassAux { uint32_t* aux; uint32_t aux_zero; bool final_ids; };

template <typename KeyT, bool VALUES, int IPT>
static int launch_pass(const RadixPlan& plan, int p, const KeyT* kin, const uint32_t* vin, KeyT* kout, uint32_t* vout, const uint32_t* hist,
                       uint32_t* states, uint32_t* gstates, uint32_t* ticket, uint32_t* err, PassAux ax, hipStream_t s) {
    static unsigned safe = 0, full = 0;       // same values on every device of a node; a benign race at worst
    if (safe == 0) {
        full = resident_blocks(radix_pass_kernel<KeyT, VALUES, IPT, true, false>, false);
        safe = resident_blocks(radix_pass_kernel<KeyT, VALUES, IPT, false, false>, true);
    }
    const bool single = plan.ntiles <= safe && !ticket_forced();
    const dim3 grid(single ? plan.ntiles : (plan.ntiles < full ? plan.ntiles : full));
#define GSPL_PASS(TK, FN) hipLaunchKernelGGL((radix_pass_kernel<KeyT, VALUES, IPT, TK, FN>), grid, dim3(RS_THREADS), 0, s, kin, vin, kout, vout, plan.n, \
                                             plan.ntiles, plan.shift[p], plan.bits[p], hist, states, gstates, ticket, err, ax.aux, ax.aux_zero)
    if constexpr (sizeof(KeyT) == 8 && !VALUES) {
        if (ax.final_ids) { if (single) GSPL_PASS(false, true); else GSPL_PASS(true, true); return check_launch("radix_sort(final pass)"); }
    }
    if (single) GSPL_PASS(false, false); else GSPL_PASS(true, false);
#undef GSPL_PASS
    return check_launch("radix_sort(pass)");
}

void radix_header_args(const RadixPlan& plan, void* workspace, RadixHeader& hdr) {
    char* ws = (char*)workspace;
    hdr.hist = (uint32_t*)(ws + plan.hist_off);
    hdr.states = (uint4*)(ws + plan.states_off);
    hdr.state_vec4 = (uint32_t)((size_t)plan.passes * ((size_t)plan.ntiles + plan.ngroups) * RADIX_BINS / 4);
    hdr.passes = plan.passes;
    for (int p = 0; p < RADIX_MAX_PASSES; ++p) { hdr.shift[p] = plan.shift[p]; hdr.mask[p] = plan.bits[p] ? ((1u << plan.bits[p]) - 1u) : 0u; }
}

template <typename KeyT, int IPT>
static int radix_sort_impl(const RadixPlan& plan, void* workspace, KeyT* const keys[2], uint32_t* const vals[2], bool prepared, void* stream,
                           uint32_t* final_ids = nullptr, uint32_t* tile_counts = nullptr, uint32_t n_tile_counts = 0) {
    if (plan.n == 0) return GSPL_OK;
    if (plan.tile_items != (uint32_t)(RS_THREADS * IPT)) return fail_arg("radix_sort: plan made for another tile size");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    uint32_t* hist = (uint32_t*)(ws + plan.hist_off);
    uint32_t* states = (uint32_t*)(ws + plan.states_off);
    const size_t pass_words = ((size_t)plan.ntiles + plan.ngroups) * RADIX_BINS;      // tile rows, then group rows
    if (!prepared) {
        hipError_t e = hipMemsetAsync(ws + plan.hist_off, 0, plan.header_bytes, s);
        if (e != hipSuccess) return check_hip(e, "radix_sort: header clear");
        RadixHeader hdr;
        radix_header_args(plan, workspace, hdr);
        const size_t want = ((size_t)plan.n + RS_THREADS * 4 - 1) / (RS_THREADS * 4);
        const unsigned grid = (unsigned)(want < 1024 ? want : 1024);
        hipLaunchKernelGGL(radix_header_kernel<KeyT>, dim3(grid ? grid : 1), dim3(RS_THREADS), 0, s, (const KeyT*)keys[0], plan.n, hdr);
        int rc = check_launch("radix_sort(header)");
        if (rc != GSPL_OK) return rc;
    }
    uint32_t* err = (uint32_t*)(ws + plan.ticket_off) + RADIX_ERR_WORD;
    for (int p = 0; p < plan.passes; ++p) {
        int rc;
        PassAux ax = {nullptr, 0u, false};
        if (final_ids) {      // tile sort: the last pass writes ids + per-tile counts, the one before clears the counters
            const bool last = p == plan.passes - 1;
            ax.aux = tile_counts;
            ax.final_ids = last;
            if (p == plan.passes - 2) ax.aux_zero = n_tile_counts;
        }
        uint32_t* vout = (final_ids && p == plan.passes - 1) ? final_ids : (vals ? vals[(p + 1) & 1] : nullptr);
        if (vals) rc = launch_pass<KeyT, true, IPT>(plan, p, keys[p & 1], vals[p & 1], keys[(p + 1) & 1], vout, hist + p * RADIX_BINS, states + p * pass_words, states + p * pass_words + (size_t)plan.ntiles * RADIX_BINS, (uint32_t*)(ws + plan.ticket_off) + p, err, ax, s);
        else rc = launch_pass<KeyT, false, IPT>(plan, p, keys[p & 1], nullptr, keys[(p + 1) & 1], vout, hist + p * RADIX_BINS, states + p * pass_words, states + p * pass_words + (size_t)plan.ntiles * RADIX_BINS, (uint32_t*)(ws + plan.ticket_off) + p, err, ax, s);
        if (rc != GSPL_OK) return rc;
    }
    return GSPL_OK;
}

bool radix_sort_u32_is_single_wave_of_tiles(size_t n) {
    static unsigned safe = 0;
    if (safe == 0) safe = resident_blocks(radix_pass_kernel<uint32_t, true, RADIX_TILE_U32 / RS_THREADS, false, false>, true);
    return (n + RADIX_TILE_U32 - 1) / RADIX_TILE_U32 <= safe && !ticket_forced();
}

// The tile sort of the binning: u64 records (tile id << 32 | splat id) sorted on the tile-id bits [32, 32 + tile_bits);
// the sorted low words go straight to `ids_out`, `tile_counts[0, n_tile_counts)` receives the number of records of every tile id.
// The pass count must be >= 2 (the pass before the last clears the counters): callers with a single pass clear them themselves.
int radix_sort_tiles(const RadixPlan& plan, void* workspace, uint64_t* const keys[2], bool prepared, uint32_t* ids_out,
                     uint32_t* tile_counts, uint32_t n_tile_counts, void* stream) {
    return radix_sort_impl<uint64_t, RADIX_TILE_U64 / RS_THREADS>(plan, workspace, keys, nullptr, prepared, stream, ids_out, tile_counts, n_tile_counts);
}

int radix_sort_u32(const RadixPlan& plan, void* workspace, uint32_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream) {
    return radix_sort_impl<uint32_t, RADIX_TILE_U32 / RS_THREADS>(plan, workspace, keys, vals, prepared, stream);
}
int radix_sort_u64(const RadixPlan& plan, void* workspace, uint64_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream) {
    return radix_sort_impl<uint64_t, RADIX_TILE_U64 / RS_THREADS>(plan, workspace, keys, vals, prepared, stream);
}


// ---- scan of the tile counts in depth order -------------------------------------------------------------------------------
static constexpr unsigned long long SC_FLAG_LOCAL = 1ull << 62;
static constexpr unsigned long long SC_FLAG_GLOBAL = 2ull << 62;
static constexpr unsigned long long SC_VALUE_MASK = (1ull << 62) - 1ull;
static constexpr int SC_IPT = SCAN_TILE / RS_THREADS;

template <bool TICKET>      // as radix_pass_kernel: one tile per workgroup, or tiles drawn from a counter
__global__ __launch_bounds__(RS_THREADS) void scan_gather_kernel(const uint32_t* __restrict__ order, const int32_t* __restrict__ counts,
                                                                 int64_t* __restrict__ cum, uint32_t n, uint32_t ntiles,
                                                                 unsigned long long* __restrict__ states, uint32_t* __restrict__ ticket,
                                                                 int32_t* __restrict__ tagged_list, uint32_t* __restrict__ err) {
    __shared__ unsigned long long s_wave[RS_WAVES];
    __shared__ unsigned long long s_excl;
    __shared__ uint32_t s_tile;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    while (true) {
        uint32_t tile = blockIdx.x;
        if (TICKET) {
            if (t == 0) s_tile = atomicAdd(ticket, 1u);
            __syncthreads();
            tile = s_tile;
        }
        if (tile >= ntiles) break;
        const uint32_t first = tile * (uint32_t)SCAN_TILE + (uint32_t)t * SC_IPT;       // SC_IPT consecutive items per thread
        unsigned long long v[SC_IPT], mine = 0ull;
#pragma unroll
        for (int k = 0; k < SC_IPT; ++k) {
            const uint32_t i = first + k;
            // bit 31 of a count tags the item: the tagged items are ranked along the way (their number rides in bits 40.. of
            // the scanned value; the counts themselves add up to < 2^31)
            const uint32_t c = i < n ? (uint32_t)counts[order ? order[i] : i] : 0u;
            v[k] = (unsigned long long)(c & 0x7fffffffu) | ((unsigned long long)(c >> 31) << 40);
            mine += v[k];
        }
        unsigned long long incl = mine;                                                   // scan of the thread totals in the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = __shfl_up(incl, d);
            if (l >= d) incl += up;
        }
        if (l == 63) s_wave[w] = incl;
        __syncthreads();
        unsigned long long wave_off = 0ull, total = 0ull;
#pragma unroll
        for (int k = 0; k < RS_WAVES; ++k) { const unsigned long long c = s_wave[k]; if (k < w) wave_off += c; total += c; }
        if (w == 0) {
            // publish the tile's aggregate, then look back: lane j inspects tile p - j
            if (l == 0) __hip_atomic_store(states + tile, (tile == 0u ? SC_FLAG_GLOBAL : SC_FLAG_LOCAL) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long excl = 0ull;
            if (tile > 0u) {
                int p = (int)tile - 1;
                uint32_t spins = 0u;
                while (true) {
                    const int idx = p - l;
                    const unsigned long long sv = idx >= 0 ? __hip_atomic_load(states + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SC_FLAG_GLOBAL;
                    const unsigned flag = (unsigned)(sv >> 62);
                    const unsigned long long glob = __ballot(flag == 2u), empty = __ballot(flag == 0u);
                    const int fg = glob ? (int)__builtin_ctzll(glob) : 64, fe = empty ? (int)__builtin_ctzll(empty) : 64;
                    const int take = fe < fg ? fe : (fg < 64 ? fg + 1 : 64);              // lanes [0, take) are consumed
                    unsigned long long part = l < take ? (sv & SC_VALUE_MASK) : 0ull;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
                    excl += part;
                    if (fg < fe) break;
                    p -= take;
                    if (take == 0) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) { if (l == 0) atomicOr(err, 1u); break; }       // a predecessor never published: flag it, do not hang
                    }
                }
                if (l == 0) __hip_atomic_store(states + tile, SC_FLAG_GLOBAL | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (l == 0) s_excl = excl;
        }
        __syncthreads();
        unsigned long long run = s_excl + wave_off + (incl - mine);
#pragma unroll
        for (int k = 0; k < SC_IPT; ++k) {
            run += v[k];
            const uint32_t i = first + k;
            if (i < n) {
                cum[i] = (int64_t)(run & ((1ull << 40) - 1ull));
                if (tagged_list) {
                    if (v[k] >> 40) tagged_list[(run >> 40) - 1ull] = (int32_t)i;
                    if (i == n - 1u) {
                        cum[n] = (int64_t)(run >> 40);      // how many are tagged
                        // ... and the error word of the sorts that ran before this scan on the same header (the depth sort), plus
                        // this scan's own: the host reads it with the list length (0 = fine)
                        cum[n + 1] = (int64_t)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        __syncthreads();
        if (!TICKET) break;
    }
}

// counts[0, n) (u32, left by the FINAL pass of the tile sort) -> exclusive prefix in place, as int32 offsets.  One workgroup:
// n is the number of image tiles (8160 at 1080p; a few hundred thousand for very large images).
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

size_t scan_state_bytes(size_t n) {
    const size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    return ((tiles ? tiles : 1) * sizeof(unsigned long long) + 15) / 16 * 16;
}

int scan_gathered_counts(const uint32_t* order, const int32_t* counts, int64_t* cum, size_t n, void* states, uint32_t* ticket,
                         int32_t* tagged_list, uint32_t* err, void* stream) {
    if (n == 0) return GSPL_OK;
    if (!err) {
        if (hipGetSymbolAddress((void**)&err, HIP_SYMBOL(g_sort_error_sink)) != hipSuccess) return fail_arg("scan: no error word");
    }
    static unsigned safe = 0, full = 0;
    if (safe == 0) {
        full = resident_blocks(scan_gather_kernel<true>, false);
        safe = resident_blocks(scan_gather_kernel<false>, true);
    }
    const unsigned ntiles = (unsigned)((n + SCAN_TILE - 1) / SCAN_TILE);
    if (ntiles <= safe && !ticket_forced())
        hipLaunchKernelGGL(scan_gather_kernel<false>, dim3(ntiles), dim3(RS_THREADS), 0, (hipStream_t)stream, order, counts, cum, (uint32_t)n, ntiles,
                           (unsigned long long*)states, ticket, tagged_list, err);
    else
        hipLaunchKernelGGL(scan_gather_kernel<true>, dim3(ntiles < full ? ntiles : full), dim3(RS_THREADS), 0, (hipStream_t)stream, order, counts, cum,
                           (uint32_t)n, ntiles, (unsigned long long*)states, ticket, tagged_list, err);
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
