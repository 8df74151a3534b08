// sort.hip — stable one-sweep LSD radix sort for the binning stage (gfx950, wave64).  Interface and rationale: gspl_sort.h.
//
// Replaces (inside gspl_bin_count / gspl_bin_emit_sort) the device radix sorts the reference's native rasterizers call
// between projection and compositing: gsplat `isect_tiles` -> cub::DeviceRadixSort::SortPairs and the Inria rasterizer's
// `cub::DeviceRadixSort::SortPairs(point_list_keys...)` (call sites: reference gsplat_v1_renderer.py:524-556,
// vanilla_renderer.py:111).  Ordering contract: stable, ascending on the selected key bits — identical to those.
//
// Pass kernel, per tile of 2048 items (8 waves x 4 rounds x 64 lanes, in memory order):
//   1. ticket -> tile index; digit bases of the pass (exclusive scan of its 256-bin histogram)
//   2. load; in-wave ranks by digit matching (one ballot per digit bit), per-wave digit counters in LDS
//   3. counters -> tile histogram -> publish LOCAL|count per digit; exclusive scan -> first in-tile slot per digit
//   4. permute the tile through LDS into digit order
//   5. look-back per digit over the preceding tiles' state words (LOCAL: add and go on, GLOBAL: add and stop);
//      publish GLOBAL|inclusive
//   6. write out: consecutive lanes hold consecutive items of a digit run -> runs of consecutive addresses
#include "gspl_device.h"
#include "gspl_host.h"
#include "gspl_sort.h"

namespace gspl {

static constexpr int RS_WAVES = 8;
static constexpr int RS_THREADS = RS_WAVES * 64;
static constexpr int RS_IPT = RADIX_TILE / RS_THREADS;
static_assert(RS_IPT * RS_THREADS == RADIX_TILE, "tile = threads x items per thread");

static constexpr uint32_t RS_FLAG_LOCAL = 1u << 30;
static constexpr uint32_t RS_FLAG_GLOBAL = 2u << 30;
static constexpr uint32_t RS_COUNT_MASK = (1u << 30) - 1u;

__device__ __forceinline__ uint32_t state_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void state_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct RadixPassBits {
    int passes;
    int shift[RADIX_MAX_PASSES];
    uint32_t mask[RADIX_MAX_PASSES];
};

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

// Header kernel: digit histograms of every pass (LDS bins, one global atomic per non-empty bin and workgroup) and the
// look-back states of every pass cleared.  `hist` must be zero on entry.
template <typename KeyT>
__global__ __launch_bounds__(RS_THREADS) void radix_header_kernel(const KeyT* __restrict__ keys, uint32_t n, RadixPassBits pb,
                                                                  uint32_t* __restrict__ hist, uint4* __restrict__ states, size_t state_vec4) {
    __shared__ uint32_t h[RADIX_MAX_PASSES][RADIX_BINS];
    const int t = threadIdx.x;
    for (int j = t; j < RADIX_MAX_PASSES * RADIX_BINS; j += RS_THREADS) (&h[0][0])[j] = 0u;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * RS_THREADS;
    for (size_t i = (size_t)blockIdx.x * RS_THREADS + t; i < n; i += stride) {
        const KeyT k = keys[i];
#pragma unroll
        for (int p = 0; p < RADIX_MAX_PASSES; ++p)
            if (p < pb.passes) atomicAdd(&h[p][(uint32_t)(k >> pb.shift[p]) & pb.mask[p]], 1u);
    }
    __syncthreads();
    for (int j = t; j < pb.passes * RADIX_BINS; j += RS_THREADS) {
        const uint32_t c = (&h[0][0])[j];
        if (c) atomicAdd(hist + j, c);
    }
    for (size_t j = (size_t)blockIdx.x * RS_THREADS + t; j < state_vec4; j += stride) states[j] = make_uint4(0u, 0u, 0u, 0u);
}

template <typename KeyT, bool VALUES>
struct RadixShared {
    uint32_t wcnt[RS_WAVES][RADIX_BINS];      // per-wave digit counters, then exclusive prefix over the waves
    uint32_t histo[RADIX_BINS];               // the pass's global histogram
    uint32_t dbase[RADIX_BINS];               // global position of the first item of each digit
    uint32_t tilecnt[RADIX_BINS];             // the tile's digit histogram
    uint32_t dstart[RADIX_BINS];              // first in-tile slot of each digit
    uint32_t gbase[RADIX_BINS];               // global position of in-tile slot 0 as seen by each digit (modular)
    KeyT xkey[RADIX_TILE];
    uint32_t xval[VALUES ? RADIX_TILE : 1];
    uint32_t tile;
};

template <typename KeyT, bool VALUES>
__global__ __launch_bounds__(RS_THREADS) void radix_pass_kernel(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                                int shift, int nbits, const uint32_t* __restrict__ hist,
                                                                uint32_t* __restrict__ ticket, uint32_t* __restrict__ states) {
    __shared__ RadixShared<KeyT, VALUES> sh;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const uint32_t mask = (1u << nbits) - 1u;
    const int nd = 1 << nbits;
    // ---- 1: ticket, digit bases --------------------------------------------------------------------------------
    if (t == 0) sh.tile = atomicAdd(ticket, 1u);
    if (t < RADIX_BINS) sh.histo[t] = (t < nd) ? hist[t] : 0u;
#pragma unroll
    for (int k = l; k < RADIX_BINS; k += 64) sh.wcnt[w][k] = 0u;
    __syncthreads();
    const uint32_t tile = sh.tile;
    const uint32_t base = tile * (uint32_t)RADIX_TILE;
    const uint32_t tile_n = min((uint32_t)RADIX_TILE, n - base);
    scan256_excl(sh.histo, sh.dbase);
    // ---- 2: load and rank (wave w owns items [w*256, w*256+256) of the tile, 64 per round) ------------------------
    KeyT key[RS_IPT];
    uint32_t val[RS_IPT], rk[RS_IPT];
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const uint32_t slot = (uint32_t)(w * (64 * RS_IPT) + r * 64 + l);
        const bool valid = slot < tile_n;
        key[r] = valid ? keys_in[base + slot] : (KeyT)0;
        val[r] = 0u;
        if (VALUES) val[r] = valid ? vals_in[base + slot] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const uint32_t slot = (uint32_t)(w * (64 * RS_IPT) + r * 64 + l);
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
    // ---- 3: tile histogram, publish LOCAL, in-tile digit starts ------------------------------------------------------
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
    __syncthreads();
    // ---- 4: permute through LDS ------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const uint32_t slot = (uint32_t)(w * (64 * RS_IPT) + r * 64 + l);
        if (slot < tile_n) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
            const uint32_t pos = sh.dstart[d] + sh.wcnt[w][d] + rk[r];
            sh.xkey[pos] = key[r];
            if (VALUES) sh.xval[pos] = val[r];
        }
    }
    // ---- 5: look-back --------------------------------------------------------------------------------------------------
    if (t < nd) {
        uint32_t excl = 0u;
        if (tile > 0u) {
            uint32_t p = tile - 1u;
            uint32_t spins = 0u;
            while (true) {
                const uint32_t s = state_load(states + (size_t)p * RADIX_BINS + t);
                const uint32_t flag = s >> 30;
                if (flag == 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 24)) __builtin_trap();      // a predecessor never published: fail loudly, do not hang
                    continue;
                }
                excl += s & RS_COUNT_MASK;
                if (flag == 2u) break;
                --p;                                                    // tile 0 always publishes GLOBAL: p never passes it
            }
            state_store(states + (size_t)tile * RADIX_BINS + t, RS_FLAG_GLOBAL | (excl + my_count));
        }
        sh.gbase[t] = sh.dbase[t] + excl - sh.dstart[t];
    }
    __syncthreads();
    // ---- 6: write out ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const uint32_t i = (uint32_t)(r * RS_THREADS + t);
        if (i < tile_n) {
            const KeyT k = sh.xkey[i];
            const uint32_t d = (uint32_t)(k >> shift) & mask;
            const uint32_t gpos = sh.gbase[d] + i;
            keys_out[gpos] = k;
            if (VALUES) vals_out[gpos] = sh.xval[i];
        }
    }
}

bool radix_plan(size_t n, int begin_bit, int end_bit, int digit_bits, RadixPlan& plan) {
    if (n > RADIX_MAX_ITEMS || begin_bit < 0 || end_bit <= begin_bit || digit_bits < 1 || digit_bits > 8) return false;
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
    plan.ntiles = (uint32_t)((n + RADIX_TILE - 1) / RADIX_TILE);
    plan.hist_off = 0;
    plan.ticket_off = (size_t)RADIX_MAX_PASSES * RADIX_BINS * sizeof(uint32_t);
    plan.header_bytes = plan.ticket_off + 64;
    plan.states_off = plan.header_bytes;
    plan.total_bytes = plan.states_off + (size_t)passes * (plan.ntiles > 0 ? plan.ntiles : 1) * RADIX_BINS * sizeof(uint32_t);
    return true;
}

template <typename KeyT>
static int radix_sort_impl(const RadixPlan& plan, void* workspace, KeyT* const keys[2], uint32_t* const vals[2], bool prepared, void* stream) {
    if (plan.n == 0) return GSPL_OK;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    uint32_t* hist = (uint32_t*)(ws + plan.hist_off);
    uint32_t* tickets = (uint32_t*)(ws + plan.ticket_off);
    uint32_t* states = (uint32_t*)(ws + plan.states_off);
    const size_t pass_words = (size_t)plan.ntiles * RADIX_BINS;
    if (!prepared) {
        hipError_t e = hipMemsetAsync(ws + plan.hist_off, 0, plan.header_bytes, s);
        if (e != hipSuccess) return check_hip(e, "radix_sort: header clear");
        RadixPassBits pb;
        pb.passes = plan.passes;
        for (int p = 0; p < RADIX_MAX_PASSES; ++p) { pb.shift[p] = plan.shift[p]; pb.mask[p] = plan.bits[p] ? ((1u << plan.bits[p]) - 1u) : 0u; }
        const size_t want = ((size_t)plan.n + RS_THREADS * 4 - 1) / (RS_THREADS * 4);
        const unsigned grid = (unsigned)(want < 1024 ? want : 1024);
        hipLaunchKernelGGL(radix_header_kernel<KeyT>, dim3(grid ? grid : 1), dim3(RS_THREADS), 0, s, (const KeyT*)keys[0], plan.n, pb, hist,
                           (uint4*)states, (size_t)plan.passes * pass_words / 4);
        int rc = check_launch("radix_sort(header)");
        if (rc != GSPL_OK) return rc;
    }
    for (int p = 0; p < plan.passes; ++p) {
        const KeyT* kin = keys[p & 1];
        KeyT* kout = keys[(p + 1) & 1];
        if (vals) {
            hipLaunchKernelGGL((radix_pass_kernel<KeyT, true>), dim3(plan.ntiles), dim3(RS_THREADS), 0, s, kin, (const uint32_t*)vals[p & 1], kout,
                               vals[(p + 1) & 1], plan.n, plan.shift[p], plan.bits[p], hist + p * RADIX_BINS, tickets + p, states + p * pass_words);
        } else {
            hipLaunchKernelGGL((radix_pass_kernel<KeyT, false>), dim3(plan.ntiles), dim3(RS_THREADS), 0, s, kin, (const uint32_t*)nullptr, kout,
                               (uint32_t*)nullptr, plan.n, plan.shift[p], plan.bits[p], hist + p * RADIX_BINS, tickets + p, states + p * pass_words);
        }
        int rc = check_launch("radix_sort(pass)");
        if (rc != GSPL_OK) return rc;
    }
    return GSPL_OK;
}

int radix_sort_u32(const RadixPlan& plan, void* workspace, uint32_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream) {
    return radix_sort_impl<uint32_t>(plan, workspace, keys, vals, prepared, stream);
}
int radix_sort_u64(const RadixPlan& plan, void* workspace, uint64_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream) {
    return radix_sort_impl<uint64_t>(plan, workspace, keys, vals, prepared, stream);
}

}  // namespace gspl

// ---- C-ABI (declared in include/gspl_hip.h) ---------------------------------------------------------------------------
extern "C" size_t gspl_radix_sort_workspace_bytes(int64_t n, int begin_bit, int end_bit) {
    gspl::RadixPlan plan;
    if (n < 0 || !gspl::radix_plan((size_t)n, begin_bit, end_bit, 8, plan)) return 0;
    return plan.total_bytes;
}

static int sort_args(int64_t n, int key_bits, int begin_bit, int end_bit, const void* k0, const void* k1, const void* ws, size_t ws_bytes,
                     int* result_buffer, gspl::RadixPlan& plan, const char* who) {
    using namespace gspl;
    if (n < 0 || begin_bit < 0 || end_bit > key_bits || end_bit <= begin_bit || !result_buffer) return fail_arg(who);
    if ((size_t)n > RADIX_MAX_ITEMS) { set_error(who, "more than 2^30-1 items"); return GSPL_ERR_UNSUPPORTED; }
    if (!radix_plan((size_t)n, begin_bit, end_bit, 8, plan)) return fail_arg(who);
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
