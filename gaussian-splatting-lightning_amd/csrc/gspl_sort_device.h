// gspl_sort_device.h — device-side half of a PREPARED radix sort (gspl_sort.h), for kernels that produce the keys: they touch
// every key anyway, so they count the first pass's digits and the sort's first launch is already a scatter.
//   __shared__ uint32_t h[RADIX_BINS];
//   radix_producer_clear(h);  __syncthreads();
//   ... radix_producer_add(h, rp, key, valid) for every key (called by whole waves) ...
//   __syncthreads();  radix_producer_flush(h, rp, index of the workgroup's first key);
// The keys of one producer workgroup must fall into ONE span of the sort (rp.span_items consecutive keys, a multiple of 2048):
// its row is added to that sort workgroup's pass-0 row and to the row of its group.  Both live in zeroed memory
// (plan.header_bytes, cleared by the caller before the producer runs).
#pragma once
#include "gspl_sort.h"
#include <hip/hip_runtime.h>

namespace gspl {

struct RadixProducer {
    uint32_t* counts0;         // [sort workgroup][RADIX_BINS]
    uint32_t* groups0;         // [sort workgroup / RADIX_GROUP][RADIX_BINS]
    uint32_t span_items;       // keys per sort workgroup
    int shift;                 // pass-0 digit = (key >> shift) & mask
    uint32_t mask;
};

__device__ __forceinline__ void radix_producer_clear(uint32_t* h) {
    for (int j = threadIdx.x; j < RADIX_BINS; j += blockDim.x) h[j] = 0u;
}

// One key per lane (valid: the lane holds a key); must be called by all lanes of the wave.
template <typename KeyT>
__device__ __forceinline__ void radix_producer_add(uint32_t* h, const RadixProducer& rp, KeyT key, bool valid) {
    if (valid) atomicAdd(&h[(uint32_t)(key >> rp.shift) & rp.mask], 1u);
}

__device__ __forceinline__ void radix_producer_flush(const uint32_t* h, const RadixProducer& rp, size_t first_key) {
    const size_t g = first_key / rp.span_items;
    for (int j = threadIdx.x; j < RADIX_BINS; j += blockDim.x) {
        const uint32_t c = h[j];
        if (c) {
            atomicAdd(rp.counts0 + g * RADIX_BINS + j, c);
            atomicAdd(rp.groups0 + (g / RADIX_GROUP) * RADIX_BINS + j, c);
        }
    }
}

}  // namespace gspl
