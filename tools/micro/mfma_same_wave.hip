// Can the per-splat reduction of the compositing backward ("phase 2") move to the fp32 MFMA pipe for free?
// The case tools/micro/mfma_pk_overlap.hip does not cover: MFMAs and packed-fp32 VALU issued BY THE SAME WAVE with
// independent operands, at the occupancy of composite_bwd2_kernel (2-wave workgroups, 5 waves per SIMD).
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_same_wave.hip -o /tmp/mfma_same_wave
// One loop iteration = one half-tile candidate of the real kernel, in instruction mix (SQ counters, profiles/r06end_pmc_SQ.csv: 67 VALU
// per candidate, ~22 of them phase 2):
//   P1 : the phase-1 chain: 45 VALU = 16 v_pk_fma_f32 + 6 v_pk_mul_f32 + 2 v_exp_f32 + 2 v_rcp_f32 + 19 plain (fma / cmp / cndmask)
//   V2 : P1 + 22 more VALU (11 v_pk_fma_f32 + 11 v_fma_f32)            — phase 2 as it is today
//   M4 : P1 + 4 v_mfma_f32_16x16x4_f32 (operands independent of P1)     — phase 2 as an exact-f32 contraction: 128 pixels x 16 columns
//        = 32 MFMAs per 8 candidates = 4 per candidate
//   M4+5: M4 + 5 VALU (the moments -> gradients conversion, amortised)
//   M  : the 4 MFMAs alone
// Reported: ns per iteration per wave-slot and the ratio to P1.  "M4 ~ P1" = the MFMA pipe is free; "M4 ~ P1 + M" = it adds.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(128, 5) void probe(float* out, int iters, float seed) {
    const int t = threadIdx.x;
    v2f a[8], m = {1.0001f + seed, 0.9999f}, c = {0.5f, 0.25f};
    float p[8];
    for (int i = 0; i < 8; ++i) { a[i] = (v2f){(float)(t + i) * 1e-3f, 1.f + seed}; p[i] = (float)(t * 3 + i) * 1e-3f; }
    v4f acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float ma = (float)t * 1e-3f + seed, mb = 1.f + (float)(t & 15) * 1e-3f;
    v2f e[6];
    float q[6];
    for (int i = 0; i < 6; ++i) { e[i] = (v2f){seed + i, 1.f}; q[i] = seed * i; }
    for (int it = 0; it < iters; ++it) {
        if (MODE != 4) {
            // ---- phase-1-like: 16 pk_fma, 6 pk_mul, 2 exp, 2 rcp, 19 plain
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[(i + 1) & 7], c, a[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) a[i] = a[i] * m;
            p[0] = __builtin_amdgcn_exp2f(p[0] * -0.001f); p[1] = __builtin_amdgcn_exp2f(p[1] * -0.001f);
            p[2] = __builtin_amdgcn_rcpf(1.5f + p[2] * 0.001f); p[3] = __builtin_amdgcn_rcpf(1.5f + p[3] * 0.001f);
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_fmaf(p[i], 0.999f, p[(i + 3) & 7] * 0.001f);   // 8 mul + 8 fma
            p[4] = (p[5] < p[6]) ? p[4] : p[7];                                                         // cmp + cndmask
        }
        if (MODE == 1 || MODE == 3 || MODE == 4) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, acc1, 0, 0, 0);
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 6; ++i) e[i] = __builtin_elementwise_fma(e[i], m, c);
#pragma unroll
            for (int i = 0; i < 5; ++i) e[i] = __builtin_elementwise_fma(e[i + 1], c, e[i]);
#pragma unroll
            for (int i = 0; i < 6; ++i) q[i] = __builtin_fmaf(q[i], 0.999f, 0.5f);
#pragma unroll
            for (int i = 0; i < 5; ++i) q[i] = __builtin_fmaf(q[i + 1], 0.001f, q[i]);
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 5; ++i) q[i] = __builtin_fmaf(q[i + 1], 0.001f, q[i]);
        }
    }
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + p[i];
    for (int i = 0; i < 6; ++i) r += e[i].x + e[i].y + q[i];
    r += acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w;
    if (r == 123.456f) out[0] = r;
}

template <int MODE>
static float run(int blocks, int iters, float* d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        probe<MODE><<<blocks, 128>>>(d_out, iters, 0.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    float* d_out; hipMalloc(&d_out, 4);
    const int iters = 4000;
    for (int per_cu : {2, 10}) {               // waves per SIMD: 1 (2-wave workgroups x 2 per CU), 5 (the real kernel's residency)
        const int blocks = 256 * per_cu;
        const float p1 = run<0>(blocks, iters, d_out), m4 = run<1>(blocks, iters, d_out), v2 = run<2>(blocks, iters, d_out),
                    m45 = run<3>(blocks, iters, d_out), mo = run<4>(blocks, iters, d_out);
        const double per = 1e6 / iters;        // ms -> ns per iteration
        printf("%d workgroups of 2 waves per CU (%.1f waves per SIMD), %d iterations\n", per_cu, per_cu * 2 / 4.0, iters);
        printf("  P1   (45 VALU)                 %8.3f ms  %7.1f ns/iter  1.000\n", p1, p1 * per);
        printf("  V2   (P1 + 22 VALU)            %8.3f ms  %7.1f ns/iter  %.3f\n", v2, v2 * per, v2 / p1);
        printf("  M4   (P1 + 4 f32 MFMA)         %8.3f ms  %7.1f ns/iter  %.3f\n", m4, m4 * per, m4 / p1);
        printf("  M4+5 (P1 + 4 f32 MFMA + 5 VALU)%8.3f ms  %7.1f ns/iter  %.3f\n", m45, m45 * per, m45 / p1);
        printf("  M    (4 f32 MFMA alone)        %8.3f ms  %7.1f ns/iter  %.3f   (P1 + M = %.3f)\n", mo, mo * per, mo / p1, (p1 + mo) / p1);
    }
    return 0;
}
