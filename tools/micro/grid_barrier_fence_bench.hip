// Micro-benchmark: a grid-wide barrier WITH the memory semantics a multi-pass sort needs on MI355X — every workgroup writes a slice of a
// buffer (plain stores), the barrier makes it visible to every other workgroup (agent-scope release before the arrive, acquire after the
// wait: the eight XCDs have an L2 each), then every workgroup reads a slice another workgroup wrote.  Compared with the same two phases
// as two kernel launches.  build: hipcc --offload-arch=gfx950 -O3 -o /tmp/gbf tools/micro/grid_barrier_fence_bench.hip ; run: /tmp/gbf
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void bar(unsigned* words, unsigned nblocks, unsigned& phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);                       // agent scope by default in HIP device code: writes back this XCD's dirty lines
        const unsigned grp = blockIdx.x & 7u;
        const unsigned members = (nblocks - grp + 7u) / 8u;
        unsigned* gcnt = words + 64 * (1 + grp);
        unsigned* top = words;
        unsigned* flag = words + 64 * 9;
        const unsigned old = __hip_atomic_fetch_add(gcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (phase + 1u) * members - 1u) {
            const unsigned groups = nblocks < 8u ? nblocks : 8u;
            const unsigned o2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 == (phase + 1u) * groups - 1u) __hip_atomic_store(flag, phase + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase + 1u) __builtin_amdgcn_s_sleep(1);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);                       // invalidates this XCD's L2 / the CU's L1
    }
    phase++;
    __syncthreads();
}

// phase p: workgroup b writes slice (b + p) % G of buf[p & 1] and, after the barrier, checks slice (b + p + G / 2) % G written by another workgroup
__global__ __launch_bounds__(512) void coop(unsigned* words, unsigned* buf0, unsigned* buf1, int slice_words, int iters, unsigned* bad) {
    unsigned phase = 0;
    const int G = gridDim.x;
    for (int p = 0; p < iters; ++p) {
        unsigned* w = (p & 1) ? buf1 : buf0;
        const int mine = (blockIdx.x + p) % G;
        for (int i = threadIdx.x; i < slice_words; i += blockDim.x) w[(size_t)mine * slice_words + i] = (unsigned)(p * 131 + mine + i);
        bar(words, G, phase);
        const int other = (blockIdx.x + p + G / 2) % G;
        unsigned wrong = 0;
        for (int i = threadIdx.x; i < slice_words; i += blockDim.x) wrong += w[(size_t)other * slice_words + i] != (unsigned)(p * 131 + other + i);
        if (wrong) atomicAdd(bad, wrong);
    }
}
__global__ __launch_bounds__(512) void k_write(unsigned* w, int slice_words, int p) {
    const int G = gridDim.x, mine = (blockIdx.x + p) % G;
    for (int i = threadIdx.x; i < slice_words; i += blockDim.x) w[(size_t)mine * slice_words + i] = (unsigned)(p * 131 + mine + i);
}
__global__ __launch_bounds__(512) void k_check(const unsigned* w, int slice_words, int p, unsigned* bad) {
    const int G = gridDim.x, other = (blockIdx.x + p + G / 2) % G;
    unsigned wrong = 0;
    for (int i = threadIdx.x; i < slice_words; i += blockDim.x) wrong += w[(size_t)other * slice_words + i] != (unsigned)(p * 131 + other + i);
    if (wrong) atomicAdd(bad, wrong);
}

int main() {
    unsigned *words, *b0, *b1, *bad;
    const int iters = 100;
    (void)hipMalloc(&words, 4096 * 4); (void)hipMalloc(&bad, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int G : {64, 256}) for (int slice_kb : {4, 32, 128}) {
        const int sw = slice_kb * 256;
        (void)hipMalloc(&b0, (size_t)G * sw * 4); (void)hipMalloc(&b1, (size_t)G * sw * 4);
        float best = 1e9f, best2 = 1e9f;
        unsigned hbad = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(words, 0, 4096 * 4); (void)hipMemset(bad, 0, 4);
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(coop, dim3(G), dim3(512), 0, 0, words, b0, b1, sw, iters, bad);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
            (void)hipMemset(bad, 0, 4);
            (void)hipEventRecord(a);
            for (int p = 0; p < iters; ++p) {
                hipLaunchKernelGGL(k_write, dim3(G), dim3(512), 0, 0, (p & 1) ? b1 : b0, sw, p);
                hipLaunchKernelGGL(k_check, dim3(G), dim3(512), 0, 0, (p & 1) ? b1 : b0, sw, p, bad);
            }
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b); if (ms < best2) best2 = ms;
        }
        printf("grid %3d, %3d KB per workgroup and phase: cooperative %.2f us per (write, barrier, read) phase [wrong words: %u]; two launches per phase: %.2f us\n",
               G, slice_kb, best * 1000.f / iters, hbad, best2 * 1000.f / iters);
        (void)hipFree(b0); (void)hipFree(b1);
    }
    return 0;
}
