// gspl_sort.h — stable LSD radix sort of the binning stage (host interface; kernels in sort.hip).
//
// The two sorts of a frame (1 M depth keys with splat ids; ~14 M (tile | rank) records) are small: a library sort spends
// as long in its per-pass bookkeeping launches (two buffer fills per pass for the look-back state, a histogram kernel and
// a scan kernel per sort: 18 extra launches, ~0.1 ms of a 1.3 ms training step) as in moving keys.  This sort keeps the
// one-sweep structure (a single read and a single write of the data per pass, chained-scan look-back between tiles) and
// strips the rest:
//   * ONE header kernel per sort computes the digit histograms of every pass and clears the look-back states of every
//     pass (or no kernel at all when the producer of the keys did both, see `RadixSort::prepared`);
//   * one kernel per pass.  When all tiles fit the device at once (the 1 M-splat depth sort: 489 tiles) every workgroup takes
//     the tile of its index — a tile only waits for lower tiles, so a co-resident grid needs no ordering at all and no
//     counter (a single-address atomic hands out ~60 M tickets/s on this part: 8 us of queueing per pass for 489 tiles).
//     Larger sorts draw their tiles from a counter, one per draw: every tile below a drawn one then belongs to a running
//     workgroup, whatever the dispatch order, the residency or the other kernels on the device.  (Measured and dropped:
//     a resident grid looping over tiles b, b + grid, ... dead-locks as soon as several processes share the GPU;
//     several consecutive tiles per draw serialise the look-back — a batch's first tile waits for the previous batch's LAST.)
//   * the per-tile state is one 32-bit word (2 flag bits | 30 count bits) moved with agent-scope relaxed atomics —
//     coherent across the eight XCD L2s without cache write-backs, and self-describing, so no fences.
// Sizes above 2^30 - 1 items do not fit the state word: callers report GSPL_ERR_UNSUPPORTED there.
#pragma once
#include <cstddef>
#include <cstdint>

namespace gspl {

static constexpr int RADIX_MAX_PASSES = 4;
static constexpr int RADIX_BINS = 256;                 // row width of the histogram / look-back tables
#ifndef GSPL_RS_TILE_U32
#define GSPL_RS_TILE_U32 2048
#endif
#ifndef GSPL_RS_TILE_U64
#define GSPL_RS_TILE_U64 4096
#endif
static constexpr int RADIX_TILE_U32 = GSPL_RS_TILE_U32;   // items per tile of the (u32 key, u32 value) sort: 512 threads x 4
static constexpr int RADIX_TILE_U64 = GSPL_RS_TILE_U64;   // items per tile of the u64 keys-only sort: 512 threads x 8
static constexpr int RADIX_HIST_COPIES = 8;             // copies of the global histogram (gspl_sort_device.h)
static constexpr size_t RADIX_MAX_ITEMS = (1u << 30) - 1u;

struct RadixPlan {
    int passes;
    int shift[RADIX_MAX_PASSES];
    int bits[RADIX_MAX_PASSES];
    uint32_t n;
    uint32_t tile_items;
    uint32_t ntiles;
    uint32_t ngroups;                                  // look-back groups (complete ones)
    // workspace layout (byte offsets): [hist: 8 copies x 4 passes x 256 u32][tile counters: 16 u32][states: passes x (ntiles + ngroups) x 256 u32]
    size_t hist_off, ticket_off, states_off, header_bytes, total_bytes;
};

// Plan a sort of key bits [begin_bit, end_bit).  digit_bits = widest digit (<= 8); passes = ceil(bits / digit_bits),
// the bits are spread evenly over the passes.  Returns false if the request is not representable.
bool radix_plan(size_t n, int begin_bit, int end_bit, int digit_bits, int tile_items, RadixPlan& plan);
void radix_replan_items(RadixPlan& plan, size_t n);      // the same plan for n <= plan.n items

// Sort.  keys[0]/vals[0] hold the input; buffer 1 is scratch of the same size.  The sorted sequence ends in buffer
// (plan.passes & 1).  vals may be nullptr (keys only).  `prepared`: the caller already zeroed the header
// (plan.header_bytes at workspace + plan.hist_off), accumulated the histograms and cleared the states.
// What a key-producing kernel needs to prepare a sort (struct RadixHeader of gspl_sort_device.h, filled on the host).
struct RadixHeader;
void radix_header_args(const RadixPlan& plan, void* workspace, RadixHeader& hdr);

// True when a (u32 key, u32 value) sort of n items runs one tile per workgroup (the fast path: no tile counter).  Callers with
// a library alternative use it to stay on that path only: with tiles drawn from a counter the pass kernel is bounded by the
// counter (~16 ns per tile) and queueing on it costs ~16 ns per tile.
bool radix_sort_u32_is_single_wave_of_tiles(size_t n);

static constexpr int RADIX_ERR_WORD = 15;              // word of the 16-word counter block that collects look-back time-outs

int radix_sort_u32(const RadixPlan& plan, void* workspace, uint32_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream);
// Tile sort of the binning: records (tile id << 32 | splat id) sorted on plan's bits (inside the high word); the sorted low
// words land in ids_out, tile_counts[0, n_tile_counts) receives the number of records per tile id (plan.passes >= 2).
int radix_sort_tiles(const RadixPlan& plan, void* workspace, uint64_t* const keys[2], bool prepared, uint32_t* ids_out,
                     uint32_t* tile_counts, uint32_t n_tile_counts, void* stream);
int tile_offsets_from_counts(uint32_t* counts, uint32_t n, void* stream);      // exclusive prefix, in place
// Exclusive scan of n u32 (n < 2^32; three launches; workspace: exclusive_scan_u32_workspace_bytes(n), 4-byte aligned).
size_t exclusive_scan_u32_workspace_bytes(size_t n);
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, void* workspace, void* stream);
// Process-wide: draw tiles from a counter in every sort and scan from now on (after a look-back time-out was reported).
void radix_force_ticket(bool on);
int radix_sort_u64(const RadixPlan& plan, void* workspace, uint64_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream);

// Inclusive scan of counts[order[i]] (int32 -> int64) in ONE launch: the same chained look-back as the sort passes (one
// 64-bit state word per tile of SCAN_TILE items; a wave inspects 64 predecessors per step).  `states`: scan_state_bytes(n)
// bytes, zero on entry (16-byte aligned; the key pass clears them together with the sort's rows).
static constexpr int SCAN_TILE = 2048;
size_t scan_state_bytes(size_t n);
// Counts with bit 31 set are TAGGED: the bit is not part of the count, and with `tagged_list` (nullable) the scan also writes
// the positions i of the tagged items in order to tagged_list[0..) and their number to cum[n] (cum then has n + 1 entries).
// order may be nullptr (identity).  `err`: the sort header's error word (nullable); with tagged_list the scan also copies it to
// cum[n + 1] (cum then has n + 2 entries), so that one host read-back carries the list length and the health of the sorts.
int scan_gathered_counts(const uint32_t* order, const int32_t* counts, int64_t* cum, size_t n, void* states, uint32_t* ticket /* zero on entry */,
                         int32_t* tagged_list, uint32_t* err, void* stream);

}  // namespace gspl
