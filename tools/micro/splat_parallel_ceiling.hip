// splat_parallel_ceiling.hip — what would the SPLAT-PARALLEL compositing backward cost?  (the one formulation of the
// graded kernel that had only been dismissed on paper.)  A CEILING experiment, not a kernel of the product: it executes the
// instruction stream such a backward cannot do without — and nothing else — on the real tile lists of a frame:
//   lane = list entry (64 consecutive entries of a tile's list per block, back to front); per block every lane walks the tile's 256
//   pixels; per pixel: alpha of the lane's splat, the transmittance in front of it = T checkpoint of the block x a product SCAN over
//   the lanes in front, the colour behind it = a sum SCAN over the lanes behind + what the deeper blocks left per pixel, then the
//   nine per-splat sums accumulate in the lane's registers (no cross-lane reduction, no LDS atomics) and leave as nine atomics per lane
//   and block.  The forward's T checkpoints (one per pixel and 64 entries) are read as if they existed (their values are stand-ins:
//   the result is NOT the gradient; the time is a lower bound of a kernel that computes it).
// Per-pixel data of the tile (dL/dout, last contributor, checkpoint) are staged in LDS once per block and read as broadcasts.
//   build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -shared -fPIC tools/micro/splat_parallel_ceiling.hip -o tools/micro/libsplatpar.so
//   run:   python tools/micro/splat_parallel_ceiling.py        (S-1080p-1M through the product, then this kernel on its lists)
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
// inclusive scans over the 64 lanes (lane 0 first): row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast 15 / 31
__device__ __forceinline__ float scan_mul(float v) {
    v *= dpp_f<0x111, 0xF, 0xF>(1.f, v);
    v *= dpp_f<0x112, 0xF, 0xF>(1.f, v);
    v *= dpp_f<0x114, 0xF, 0xF>(1.f, v);
    v *= dpp_f<0x118, 0xF, 0xF>(1.f, v);
    v *= dpp_f<0x142, 0xA, 0xF>(1.f, v);
    v *= dpp_f<0x143, 0xC, 0xF>(1.f, v);
    return v;
}
__device__ __forceinline__ float scan_add(float v) {
    v += dpp_f<0x111, 0xF, 0xF>(0.f, v);
    v += dpp_f<0x112, 0xF, 0xF>(0.f, v);
    v += dpp_f<0x114, 0xF, 0xF>(0.f, v);
    v += dpp_f<0x118, 0xF, 0xF>(0.f, v);
    v += dpp_f<0x142, 0xA, 0xF>(0.f, v);
    v += dpp_f<0x143, 0xC, 0xF>(0.f, v);
    return v;
}

// Inria constants: integer pixel centres, alpha <= 0.99, alpha >= 1/255
__global__ __launch_bounds__(64) void splat_parallel_ceiling_kernel(
    int n_tiles, int tile_w, int width, int height,
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors, const float* __restrict__ opacities,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ flatten_ids, int64_t n_isects,
    const float* __restrict__ final_Ts, const int32_t* __restrict__ last_ids, const float* __restrict__ v_out /* [3,H,W] */,
    float* __restrict__ v_packed /* [N, 9] */) {
    __shared__ __attribute__((aligned(16))) float s_px[256 * 4];      // per pixel: dL/dout r g b, checkpoint T
    __shared__ int s_last[256];
    __shared__ float s_R[256];                                        // colour behind, per pixel, carried from block to block
    const int tile = blockIdx.x, l = threadIdx.x;
    const int tx = (tile % tile_w) * 16, ty = (tile / tile_w) * 16;
    const int start = offsets[tile];
    const int end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
    int deepest = start;
    for (int p = l; p < 256; p += 64) {
        const int x = tx + (p & 15), y = ty + (p >> 4);
        const bool in = x < width && y < height;
        const int64_t pix = (int64_t)y * width + x;
        const int last = in ? last_ids[pix] : start;
        s_last[p] = last;
        s_R[p] = 0.f;
        s_px[p * 4 + 0] = in ? v_out[pix] : 0.f;
        s_px[p * 4 + 1] = in ? v_out[(int64_t)width * height + pix] : 0.f;
        s_px[p * 4 + 2] = in ? v_out[2 * (int64_t)width * height + pix] : 0.f;
        s_px[p * 4 + 3] = in ? final_Ts[pix] : 1.f;                    // (stand-in for the block's checkpoint)
        deepest = max(deepest, last);
    }
    for (int off = 32; off > 0; off >>= 1) deepest = max(deepest, __shfl_xor(deepest, off));
    (void)end;
    __syncthreads();
    for (int hi = deepest; hi > start; hi -= 64) {
        const int idx = hi - 1 - l;                                    // lane 0 = the deepest entry of the block
        const bool have = idx >= start;
        const int g = have ? flatten_ids[idx] : 0;
        const float mx = means2d[g * 2 + 0], my = means2d[g * 2 + 1];
        const float ha = 0.5f * conics[g * 3 + 0], cb = conics[g * 3 + 1], hc = 0.5f * conics[g * 3 + 2];
        const float op = have ? opacities[g] : 0.f;
        const float c0 = colors[g * 3 + 0], c1 = colors[g * 3 + 1], c2 = colors[g * 3 + 2];
        float S0 = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll 2
        for (int p = 0; p < 256; ++p) {
            const float4 px = *reinterpret_cast<const float4*>(s_px + p * 4);     // broadcast reads
            const int last = s_last[p];
            const float Rin = s_R[p];
            const float dx = mx - (float)(tx + (p & 15)), dy = my - (float)(ty + (p >> 4));
            const float sigma = fmaf(ha * dx, dx, fmaf(hc * dy, dy, (cb * dx) * dy));
            const float raw = op * __builtin_amdgcn_exp2f(sigma * -1.4426950408889634f);
            const bool valid = (idx < last) && (sigma >= 0.f) && (raw >= 1.f / 255.f);
            const float a = valid ? fminf(0.99f, raw) : 0.f;
            const float om = 1.f - a;
            // transmittance in front of this entry: the lanes ABOVE are in front (lane 0 is the deepest): suffix product = total / prefix
            const float incl = scan_mul(om);                               // product over lanes 0..l
            const float total = __shfl(incl, 63);
            const float T_here = px.w * total * __builtin_amdgcn_rcpf(incl);       // checkpoint x product over the lanes above
            const float w = a * T_here;
            const float cdot = fmaf(c2, px.z, fmaf(c1, px.y, c0 * px.x));
            const float contrib = w * cdot;
            const float behind_incl = scan_add(contrib);                   // lanes 0..l: the deeper entries of the block and this one
            const float behind = Rin + behind_incl - contrib;
            const float v_alpha = fmaf(cdot, T_here, -behind * __builtin_amdgcn_rcpf(om));
            const float sp = -raw * v_alpha * (valid ? 1.f : 0.f);
            S0 += sp; Sx = fmaf(sp, dx, Sx); Sy = fmaf(sp, dy, Sy);
            Sxx = fmaf(sp * dx, dx, Sxx); Sxy = fmaf(sp * dx, dy, Sxy); Syy = fmaf(sp * dy, dy, Syy);
            r0 = fmaf(w, px.x, r0); r1 = fmaf(w, px.y, r1); r2 = fmaf(w, px.z, r2);
            if (l == 63) s_R[p] = Rin + behind_incl;                      // what the next (shallower) block finds behind it
            if (l == 0) s_px[p * 4 + 3] = px.w * total;                   // (the next block would read ITS checkpoint here)
        }
        if (have) {
            float* row = v_packed + (int64_t)g * 9;
            const float ca = 2.f * ha, cc = 2.f * hc;
            atomicAdd(row + 0, ca * Sx + cb * Sy); atomicAdd(row + 1, cb * Sx + cc * Sy);
            atomicAdd(row + 2, 0.5f * Sxx); atomicAdd(row + 3, Sxy); atomicAdd(row + 4, 0.5f * Syy);
            atomicAdd(row + 5, op != 0.f ? -S0 / op : 0.f);
            atomicAdd(row + 6, r0); atomicAdd(row + 7, r1); atomicAdd(row + 8, r2);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" int splat_parallel_ceiling(int N, int64_t n_isects, const float* means2d, const float* conics, const float* colors, const float* opacities,
                                      int width, int height, const int32_t* offsets, const int32_t* flatten_ids,
                                      const float* final_Ts, const int32_t* last_ids, const float* v_out, float* v_packed, void* stream) {
    const int tile_w = (width + 15) / 16, n_tiles = tile_w * ((height + 15) / 16);
    hipLaunchKernelGGL(splat_parallel_ceiling_kernel, dim3(n_tiles), dim3(64), 0, (hipStream_t)stream, n_tiles, tile_w, width, height,
                       means2d, conics, colors, opacities, offsets, flatten_ids, n_isects, final_Ts, last_ids, v_out, v_packed);
    return (int)hipGetLastError();
}
