// gspl_sort_device.h — device-side half of the radix sort "header" (the global digit histograms of every pass), for kernels
// that PRODUCE the keys: they touch every key anyway, so the sort needs no header kernel of its own.
//   __shared__ uint32_t h[RADIX_MAX_PASSES * RADIX_BINS];
//   radix_hist_clear(h);  __syncthreads();
//   ... radix_hist_add(h, hdr, key, valid) for every key (called by whole waves) ...
//   __syncthreads();  radix_hist_flush(h, hdr);
// The global histogram has RADIX_HIST_COPIES copies (selected by workgroup index, summed by the pass kernels): thousands
// of workgroups flushing onto 1024 addresses would queue ~16 ns per atomic and address.
#pragma once
#include "gspl_sort.h"
#include <hip/hip_runtime.h>

namespace gspl {

struct RadixHeader {
    uint32_t* hist;            // [RADIX_HIST_COPIES][RADIX_MAX_PASSES][RADIX_BINS], zero on entry
    int passes;
    int shift[RADIX_MAX_PASSES];
    uint32_t mask[RADIX_MAX_PASSES];
};

__device__ __forceinline__ void radix_hist_clear(uint32_t* h) {
    for (int j = threadIdx.x; j < RADIX_MAX_PASSES * RADIX_BINS; j += blockDim.x) h[j] = 0u;
}

// One key per lane (valid: the lane holds a key); must be called by all lanes of the wave.  A wave whose keys share the
// digit — the usual case for the high digits — adds its count with one atomic instead of 64 on one address.
template <typename KeyT>
__device__ __forceinline__ void radix_hist_add(uint32_t* h, const RadixHeader& hdr, KeyT key, bool valid) {
    const unsigned long long act = __ballot(valid);
    if (act == 0ull) return;
    const int first = (int)__builtin_ctzll(act);
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
    for (int p = 0; p < RADIX_MAX_PASSES; ++p) {
        if (p < hdr.passes) {
            const uint32_t d = (uint32_t)(key >> hdr.shift[p]) & hdr.mask[p];
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
            if (__ballot(valid && d != d0) == 0ull) {
                if (lane == first) atomicAdd(&h[p * RADIX_BINS + d0], (uint32_t)__builtin_popcountll(act));
            } else if (valid) {
                atomicAdd(&h[p * RADIX_BINS + d], 1u);
            }
        }
    }
}

__device__ __forceinline__ void radix_hist_flush(const uint32_t* h, const RadixHeader& hdr) {
    uint32_t* dst = hdr.hist + (size_t)(blockIdx.x % RADIX_HIST_COPIES) * (RADIX_MAX_PASSES * RADIX_BINS);
    for (int j = threadIdx.x; j < hdr.passes * RADIX_BINS; j += blockDim.x) {
        const uint32_t c = h[j];
        if (c) atomicAdd(dst + j, c);
    }
}

}  // namespace gspl
