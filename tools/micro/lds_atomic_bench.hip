// What does an LDS float atomic cost on gfx950?  (Question behind composite_bwd4_kernel's accumulation)
//   build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_atomic_bench.hip -o tools/micro/lds_atomic_bench.bin
// One workgroup per CU, W waves per workgroup; every wave issues ITERS x 8 LDS operations of one PATTERN; reported: LDS-pipe cycles
// per wave-instruction = kernel cycles / (W * ITERS * 8) (at 2.4 GHz), for W = 4 (one wave per SIMD) and W = 16.
//   0 ds_add_f32, 64 lanes, 64 distinct addresses          4 ds_add_f32, 64 lanes, 8 groups of 8 on one address
//   1 ds_add_f32, 16 lanes active, distinct                5 ds_add_u32, 64 lanes, distinct
//   2 ds_add_f32, 16 lanes active, pairs on one address    6 ds_write_b32, 64 lanes, distinct
//   3 ds_add_f32,  8 lanes active, distinct                7 ds_read_b32 + add + ds_write_b32 (plain read-modify-write), 64 lanes
//   8 ds_add_f32, 64 lanes, random rows of 9 floats (the kernel's pattern: lane = (unit, value))
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PATTERN>
__global__ __launch_bounds__(1024) void bench(float* out, int iters) {
    __shared__ float s[4096];
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    for (int i = t; i < 4096; i += blockDim.x) s[i] = 0.f;
    __syncthreads();
    float* base = s + (w & 3) * 1024;          // each wave of a SIMD group works in its own KB: no cross-wave conflicts
    unsigned* ubase = reinterpret_cast<unsigned*>(base);
    float v = (float)l * 1e-3f + 1.f;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int off = r * 64;
            if (PATTERN == 0) atomicAdd(&base[off + l], v);
            if (PATTERN == 1) { if ((l & 3) == 0) atomicAdd(&base[off + l], v); }
            if (PATTERN == 2) { if ((l & 3) == 0) atomicAdd(&base[off + (l >> 3)], v); }
            if (PATTERN == 3) { if ((l & 7) == 0) atomicAdd(&base[off + l], v); }
            if (PATTERN == 4) atomicAdd(&base[off + (l >> 3)], v);
            if (PATTERN == 5) atomicAdd(&ubase[off + l], (unsigned)l);
            if (PATTERN == 6) base[off + l] = v;
            if (PATTERN == 7) { const float o = base[off + l]; base[off + l] = o + v; }
            if (PATTERN == 8) atomicAdd(&base[(((l >> 3) * 37 + it * 11 + r * 5) & 63) * 9 + (l & 7)], v);
        }
        v += 1e-6f;
    }
    __syncthreads();
    acc = s[t & 4095];
    if (acc == 123.456f) out[0] = acc;
}

template <int PATTERN>
static void run(float* d_out, const char* name) {
    const int iters = 4000;
    for (int waves : {4, 16}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            bench<PATTERN><<<256, waves * 64>>>(d_out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double cyc = best * 1e-3 * 2.4e9 / ((double)waves * iters * 8);
        printf("%-62s waves/CU %2d: %.3f ms -> %.1f CU-cycles per wave-instruction\n", name, waves, best, cyc);
    }
}

int main() {
    float* d_out; hipMalloc(&d_out, 4);
    run<0>(d_out, "ds_add_f32 64 lanes distinct");
    run<1>(d_out, "ds_add_f32 16 lanes distinct");
    run<2>(d_out, "ds_add_f32 16 lanes, pairs on one address");
    run<3>(d_out, "ds_add_f32 8 lanes distinct");
    run<4>(d_out, "ds_add_f32 64 lanes, 8-way same address");
    run<5>(d_out, "ds_add_u32 64 lanes distinct");
    run<6>(d_out, "ds_write_b32 64 lanes distinct");
    run<7>(d_out, "ds_read + add + ds_write 64 lanes");
    run<8>(d_out, "ds_add_f32 64 lanes = 8 rows x 8 values (kernel pattern)");
    return 0;
}
