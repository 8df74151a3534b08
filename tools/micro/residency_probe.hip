// residency_probe.hip — how many workgroups of a given shape does an MI355X really keep resident at once?
// Every workgroup of a launch checks in on a counter and then waits (bounded by a clock) for the whole grid to check in:
// the launch is co-resident iff every workgroup saw the full count.  Compared with what the occupancy API promises.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/residency_probe.hip -o /tmp/residency_probe && /tmp/residency_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned* counter, unsigned* seen_all, unsigned grid, long long budget) {
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1u;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        bool ok = false;
        while (wall_clock64() - t0 < budget) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= grid) { ok = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (ok) __hip_atomic_fetch_add(seen_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned* d;
    hipMalloc(&d, 8);
    const int threads_list[] = {256, 512, 1024};
    const int lds_list[] = {1024, 16 * 1024, 29 * 1024, 45 * 1024, 64 * 1024};
    for (int threads : threads_list)
        for (int lds : lds_list) {
            hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            int api = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, probe, threads, lds);
            printf("threads %4d lds %6d: api %2d/CU; co-resident at per-CU =", threads, lds, api);
            for (int per = 1; per <= api + 1; ++per) {
                const unsigned grid = (unsigned)(per * cus);
                hipMemset(d, 0, 8);
                hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), lds, 0, d, d + 1, grid, 100000000ll / 10);   // 100 MHz clock: 0.1 s
                hipDeviceSynchronize();
                unsigned h[2];
                hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
                printf(" %d:%s", per, h[1] == grid ? "yes" : "NO");
            }
            printf("\n");
        }
    return 0;
}
