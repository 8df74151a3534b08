// gspl_sort.h — stable LSD radix sort and scans of the binning stage (host interface; kernels in sort.hip).
//
// The two sorts of a frame (the splats' depth keys with their ids; the (tile | splat) records, ~7 records per splat) are small
// next to what a general-purpose device sort is tuned for, and a library sort spends as long in per-pass bookkeeping launches as
// in moving keys.  This sort:
//   * runs every pass as count -> scatter over contiguous tile ranges per workgroup (sort.hip): no workgroup waits for another,
//     hence no forward-progress assumptions, no polling and bit-reproducible output;
//   * can take the first pass's counts from the kernel that PRODUCES the keys (`prepared`; gspl_sort_device.h);
//   * can finish the depth sort with a pass that writes gather[value] in place of the keys nobody reads, and the tile sort with
//     a pass that writes the splat ids alone and counts the records per tile id.
// Up to 2^30 - 1 items (callers report GSPL_ERR_UNSUPPORTED above).
#pragma once
#include <cstddef>
#include <cstdint>
#include "gspl_host.h"

namespace gspl {

static constexpr int RADIX_MAX_PASSES = 4;
static constexpr int RADIX_BINS = 256;                 // row width of the histogram tables
#ifndef GSPL_RS_TILE_U32
#define GSPL_RS_TILE_U32 2048
#endif
#ifndef GSPL_RS_TILE_U64
#define GSPL_RS_TILE_U64 4096
#endif
// Workgroups of a pass per CU (capped by RADIX_MAX_WG and by the number of tiles).  Round 6: 3 -> 8 (= 2048 on 256 CUs).  The spans
// (keys per workgroup) are fixed by the plan's item count, and the tile sort is planned for the list CAPACITY of a speculative
// frame: with 768 spans over a capacity 2-3 x the real length only a third of the workgroups had keys (binning 0.269 -> 0.348 ms at
// 3 x over-capacity); with 2048 the same frame costs 0.272 ms, a tight capacity 0.264 instead of 0.269, and the 6 M scene's depth
// and tile sorts gain as well (binning 1.38 -> 1.21 ms); 4096 is no better (gpurun_out r21: profiles/r21_sort_workgroups.txt).
#ifndef GSPL_RS_WG_PER_CU
#define GSPL_RS_WG_PER_CU 8
#endif
static constexpr int RADIX_TILE_U32 = GSPL_RS_TILE_U32;   // items per tile of the (u32 key, u32 value) sort: 512 threads x 4
static constexpr int RADIX_TILE_U64 = GSPL_RS_TILE_U64;   // items per tile of the u64 sorts: 512 threads x 8
static constexpr int RADIX_MAX_WG = 2048;               // workgroups per pass at most
static constexpr int RADIX_GROUP = 32;                  // workgroups per group row (two-level prefix over the workgroups)
static constexpr size_t RADIX_MAX_ITEMS = (1u << 30) - 1u;

struct RadixPlan {
    int passes;
    int shift[RADIX_MAX_PASSES];
    int bits[RADIX_MAX_PASSES];
    uint32_t n;
    uint32_t tile_items;
    uint32_t ntiles;
    uint32_t nwg;                                      // workgroups of a pass
    uint32_t tiles_per_wg;                             // consecutive tiles each of them owns
    uint32_t wg_cap;                                   // workgroups the tables below were sized for (replanning keeps them)
    // workspace layout (byte offsets; rows of 256 u32):
    //   [groups: 4 passes x ceil(wg_cap / 32) rows][counts0: wg_cap rows]   <- header_bytes: zero before a PREPARED sort
    //   [counts: wg_cap rows]                                                 (an unprepared sort clears its group rows itself)
    size_t groups_off, groups_bytes, counts0_off, header_bytes, counts_off, total_bytes;
};

// Plan a sort of key bits [begin_bit, end_bit).  digit_bits = widest digit (<= 8); passes = ceil(bits / digit_bits),
// the bits are spread evenly over the passes.  Returns false if the request is not representable.
bool radix_plan(size_t n, int begin_bit, int end_bit, int digit_bits, int tile_items, RadixPlan& plan);
void radix_replan_items(RadixPlan& plan, size_t n);      // the same plan (bit split, workspace, keys per workgroup) for FEWER items

// Sort.  keys[0]/vals[0] hold the input; buffer 1 is scratch of the same size.  The sorted sequence ends in buffer
// (plan.passes & 1).  vals may be nullptr (keys only).  `prepared`: the caller zeroed plan.header_bytes at workspace and its
// key-producing kernel accumulated the pass-0 rows (struct RadixProducer of gspl_sort_device.h, filled on the host by
// radix_producer_args).
int radix_zero(void* p, size_t bytes, void* stream);      // clears sort tables (16-byte multiples) with a kernel, not a memset
struct RadixProducer;
void radix_producer_args(const RadixPlan& plan, void* workspace, RadixProducer& rp);

// gather (nullable; needs vals): the last pass writes gather[value] where the sorted keys would go.
int radix_sort_u32(const RadixPlan& plan, void* workspace, uint32_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream,
                   const uint32_t* gather = nullptr);
int radix_sort_u64(const RadixPlan& plan, void* workspace, uint64_t* const keys[2], uint32_t* const vals[2], bool prepared, void* stream);
// Tile sort of the binning: records (tile id << 32 | splat id) sorted on plan's bits (inside the high word); the sorted low
// words land in ids_out, tile_counts[0, n_tile_counts) receives the number of records per tile id (plan.passes >= 2).
// n_dev (nullable): the number of records lives on the device (<= plan.n, which sizes the grid): the host need not know it.
int radix_sort_tiles(const RadixPlan& plan, void* workspace, uint64_t* const keys[2], bool prepared, uint32_t* ids_out,
                     uint32_t* tile_counts, uint32_t n_tile_counts, void* stream, const int64_t* n_dev = nullptr);
int tile_offsets_from_counts(uint32_t* counts, uint32_t n, void* stream);      // exclusive prefix, in place, one workgroup
// Exclusive scan of n u32 (n < 2^32; three launches; workspace: exclusive_scan_u32_workspace_bytes(n), 4-byte aligned).
size_t exclusive_scan_u32_workspace_bytes(size_t n);
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, void* workspace, void* stream);

// Inclusive scan of counts[order[i]] (int32 -> int64; order may be nullptr = identity): block sums, then the per-block scan whose
// workgroups add up the sums in front of them (two launches), or with a one-workgroup scan of the sums in between (three).  workspace: scan_workspace_bytes(n), 8-byte aligned, no initialisation needed.
// Counts with bit 31 set are TAGGED: the bit is not part of the count, and with `tagged_list` (nullable) the scan also writes
// the positions i of the tagged items in order to tagged_list[0..) and their number to cum[n] (cum then has n + 1 entries).
// host_words (nullable): device-accessible HOST memory (pinned) for two int64 — the kernel itself stores the total and the tagged
// count there, so a host that waits for an event after the scan reads them without a copy launch.
// Up to SCAN_RAW_SUMS_BLOCKS block sums (2M items) the middle launch is folded into the last one: two launches.
// then_zero: a table the last kernel also clears (ZeroJob of gspl_host.h), for the kernel that follows on the stream.
static constexpr int SCAN_TILE = 2048;
static constexpr unsigned SCAN_RAW_SUMS_BLOCKS = 1024;
size_t scan_workspace_bytes(size_t n);
int scan_gathered_counts(const uint32_t* order, const int32_t* counts, int64_t* cum, size_t n, void* workspace, int32_t* tagged_list, void* stream,
                         int64_t* host_words = nullptr /* pinned: [0] total, [1] tagged count, and with a ticket [2] = ticket, stored last */,
                         unsigned long long ticket = 0, ZeroJob then_zero = ZeroJob());

}  // namespace gspl
