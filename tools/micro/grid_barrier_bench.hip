// Micro-benchmark: cost of a grid-wide barrier on MI355X for different grid sizes and barrier flavours.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/gb tools/micro/grid_barrier_bench.hip ; run: /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void bar_flat(unsigned* word, unsigned nblocks, unsigned& phase) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned target = (phase + 1u) * nblocks;
        __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    phase++;
    __syncthreads();
}
// two-level: 8 group counters (group = blockIdx % 8, i.e. the XCD under round-robin dispatch), one release flag
__device__ __forceinline__ void bar_tree(unsigned* words, unsigned nblocks, unsigned& phase) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned grp = blockIdx.x & 7u;
        const unsigned members = (nblocks - grp + 7u) / 8u;
        unsigned* gcnt = words + 64 * (1 + grp);       // separate 256-byte lines
        unsigned* top = words;
        unsigned* flag = words + 64 * 9;
        const unsigned old = __hip_atomic_fetch_add(gcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (phase + 1u) * members - 1u) {
            const unsigned groups = nblocks < 8u ? nblocks : 8u;
            const unsigned o2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 == (phase + 1u) * groups - 1u) __hip_atomic_store(flag, phase + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase + 1u) __builtin_amdgcn_s_sleep(1);
    }
    phase++;
    __syncthreads();
}

template <int MODE>
__global__ void k(unsigned* words, int iters, unsigned* out) {
    unsigned phase = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) bar_flat(words, gridDim.x, phase); else bar_tree(words, gridDim.x, phase);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = phase;
}

int main() {
    unsigned *words, *out;
    hipMalloc(&words, 4096 * 4); hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 200;
    for (int threads : {256, 512, 1024}) for (int G : {64, 128, 256, 512}) for (int mode : {0, 1}) {
        if (threads * G > 256 * 2048) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(words, 0, 4096 * 4);
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(threads), 0, 0, words, iters, out);
            else hipLaunchKernelGGL(k<1>, dim3(G), dim3(threads), 0, 0, words, iters, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("threads %4d grid %4d %s: %.2f us per barrier\n", threads, G, mode ? "tree" : "flat", best * 1000.f / iters);
    }
    return 0;
}
