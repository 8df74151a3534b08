// Does fp32 MFMA (v_mfma_f32_16x16x4_f32) issued by one wave of a SIMD slow down packed-fp32 VALU work (v_pk_fma_f32) of the
// other waves of that SIMD?  (Question behind DESIGN.md §4.2: phase 2 of the compositing backward on the idle MFMA pipe.)
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_pk_overlap.hip -o /tmp/mfma_pk_overlap
//   run  : /tmp/mfma_pk_overlap
// A workgroup is 4*(P+M) waves: wave w runs on SIMD w%4; the first 4P waves loop over v_pk_fma_f32 (PLAIN=1: v_fma_f32),
// the last 4M waves over MFMAs.  One workgroup per CU.  Reported: kernel time for (P,0), (0,M) and (P,M).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool PLAIN>
__global__ void mix_kernel(float* out, int pk_waves, int iters_pk, int iters_mfma) {
    const int w = threadIdx.x >> 6;
    float r = 0.f;
    if (w < pk_waves) {
        if (PLAIN) {
            float a[16];
            for (int i = 0; i < 16; ++i) a[i] = (float)(threadIdx.x + i);
            const float m = 1.0001f, c = 0.5f;
            for (int it = 0; it < iters_pk; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], m, c);
            }
            for (int i = 0; i < 16; ++i) r += a[i];
        } else {
            v2f a[16];
            for (int i = 0; i < 16; ++i) a[i] = (v2f){(float)(threadIdx.x + i), 1.f};
            const v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
            for (int it = 0; it < iters_pk; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
            }
            for (int i = 0; i < 16; ++i) r += a[i].x + a[i].y;
        }
    } else {
        v4f acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        float a = (float)threadIdx.x * 1e-3f, b = 1.f + (float)(threadIdx.x & 15) * 1e-3f;
        for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
            }
        }
        r = acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w;
    }
    if (r == 123.456f) out[0] = r;
}

template <bool PLAIN>
static float run(int P, int M, int iters_pk, int iters_mfma, float* d_out) {
    const int waves = 4 * (P + M);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        mix_kernel<PLAIN><<<256, waves * 64>>>(d_out, 4 * P, iters_pk, iters_mfma);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float* d_out; hipMalloc(&d_out, 4);
    const int IP = 20000;
    for (int plain = 0; plain < 2; ++plain) {
        printf("%s VALU waves\n", plain ? "v_fma_f32 (plain)" : "v_pk_fma_f32 (packed)");
        for (int P : {1, 2, 3}) {        // 4 * (P + 1) waves <= 16 (a 1024-thread workgroup)
            const float a = plain ? run<true>(P, 0, IP, 0, d_out) : run<false>(P, 0, IP, 0, d_out);
            for (int IM : {IP / 4, IP / 2, IP}) {         // IM iterations x 8 MFMAs x 32 cycles vs IP x 16 VALU x 4 cycles x P waves
                const float b = plain ? run<true>(0, 1, 0, IM, d_out) : run<false>(0, 1, 0, IM, d_out);
                const float c = plain ? run<true>(P, 1, IP, IM, d_out) : run<false>(P, 1, IP, IM, d_out);
                printf("  P=%d VALU waves/SIMD alone %.3f ms | 1 MFMA wave/SIMD (%d MFMAs) alone %.3f ms | together %.3f ms  (max %.3f, sum %.3f)\n",
                       P, a, IM * 8, b, c, a > b ? a : b, a + b);
            }
        }
    }
    return 0;
}
