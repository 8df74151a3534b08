// Throughput of the VALU instructions the compositing kernels are made of, on gfx950 (DESIGN.md §4.2 pipe model).
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o tools/micro/valu_rates.bin
// Every wave executes ITERS x 16 independent instructions of one kind (inline asm, 16 accumulators); W waves per SIMD on every
// SIMD of every CU.  Reported: SIMD cycles per wave-instruction at W = 1, 2, 4, 8 (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ void rates(float* out, int iters) {
    float a[16];
    float2 p[16];
    for (int i = 0; i < 16; ++i) { a[i] = (float)(threadIdx.x + i) * 1e-3f + 0.5f; p[i] = make_float2(a[i], a[i] + 1.f); }
    const float m = 1.0001f, c = 0.25f;
    const float2 m2 = make_float2(1.0001f, 0.9999f), c2 = make_float2(0.25f, 0.125f);
    unsigned long long mask = (threadIdx.x & 1) ? 0x5555555555555555ull : 0xaaaaaaaaaaaaaaaaull;
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define CND(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
#define CMP(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(a[i]), "v"(c));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
#define PKF(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
#define PKM(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
#define PKA(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#define MED(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(m));
#define SUB(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
        if (KIND == 0) { REP16(FMA) }
        if (KIND == 1) { REP16(MUL) }
        if (KIND == 2) { REP16(ADD) }
        if (KIND == 3) { REP16(CND) }
        if (KIND == 4) { REP16(CMP) }
        if (KIND == 5) { REP16(EXP) }
        if (KIND == 6) { REP16(RCP) }
        if (KIND == 7) { REP16(DPP) }
        if (KIND == 8) { REP16(PKF) }
        if (KIND == 9) { REP16(PKM) }
        if (KIND == 10) { REP16(PKA) }
        if (KIND == 11) { REP16(MED) }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y;
    if (r == 123.456f) out[0] = r;
}

template <int KIND>
static void run(float* d_out, const char* name) {
    const int iters = 4000;
    printf("%-22s", name);
    for (int w : {1, 2, 4, 8}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            rates<KIND><<<256, 4 * w * 64>>>(d_out, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("  W=%d: %5.2f cyc", w, best * 1e-3 * 2.4e9 / ((double)w * iters * 16));
    }
    printf("\n");
}

int main() {
    float* d_out; (void)hipMalloc(&d_out, 4);
    run<0>(d_out, "v_fma_f32");
    run<1>(d_out, "v_mul_f32");
    run<2>(d_out, "v_add_f32");
    run<3>(d_out, "v_cndmask_b32 (sgpr)");
    run<4>(d_out, "v_cmp_lt_f32 -> vcc");
    run<5>(d_out, "v_exp_f32");
    run<6>(d_out, "v_rcp_f32");
    run<7>(d_out, "v_add_f32_dpp quad");
    run<8>(d_out, "v_pk_fma_f32");
    run<9>(d_out, "v_pk_mul_f32");
    run<10>(d_out, "v_pk_add_f32");
    run<11>(d_out, "v_med3_f32");
    return 0;
}
