// hbm_bandwidth.hip — what does this MI355X stream?  read-only, write-only and copy kernels over 2 GiB, float4 per lane, grid-strided,
// plain and non-temporal accesses.   build: hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_bandwidth.hip -o tools/micro/hbm_bandwidth.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_read(const v4f* __restrict__ a, size_t n, float* out) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write(v4f* __restrict__ a, size_t n) {
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { if (NT) __builtin_nontemporal_store(v, a + i); else a[i] = v; }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const v4f v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
template <typename F> static float timed(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
    return best;
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    v4f *a, *b; float* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    for (int grid : {2048, 8192, 32768, 131072}) {
        const float r0 = timed([&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, a, n, out); });
        const float r1 = timed([&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, a, n, out); });
        const float w0 = timed([&] { hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, a, n); });
        const float w1 = timed([&] { hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, a, n); });
        const float c0 = timed([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, a, b, n); });
        const float c1 = timed([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, a, b, n); });
        const double g = bytes / 1e9;
        printf("grid %6d: read %.2f / nt %.2f TB/s   write %.2f / nt %.2f TB/s   copy (r+w) %.2f / nt %.2f TB/s\n", grid, g / r0, g / r1, g / w0, g / w1,
               2 * g / c0, 2 * g / c1);
    }
    const float m = timed([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpyAsync D2D: %.2f TB/s (r+w)\n", 2 * (bytes / 1e9) / m);
    return 0;
}
