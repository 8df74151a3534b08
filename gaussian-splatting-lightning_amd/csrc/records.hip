// records.hip — the 48-byte visible-splat records of the Gaussian-sharded multi-GPU renderer: pack / unpack, forward and backward.
//
// Replaces, in HipGSplatDistributedRenderer, the tensor plumbing of the reference's exchange step
// (internal/renderers/gsplat_distributed_renderer.py:313-414: per camera `torch.concat` of the projected quantities, boolean-mask
// selection of the visible rows, `all_to_all`, `torch.split` back into per-quantity tensors) — a dozen elementwise / index /
// concat launches per direction and, in the backward, a sort-based index_put.  Record = 12 fp32:
//     [x, y, depth, conic a, b, c, compensation, opacity, r, g, b, radius (int32 bits)]
// pack:   every (camera, local splat) with radius > 0 -> one record; records grouped by camera (= destination rank), rows in
//         splat order inside a camera (flag -> exclusive scan -> scatter, no atomics, deterministic order).
// unpack: records a rank received -> the per-quantity tensors the rasterization stages take (opacity x compensation folded in).
// HBM-bound elementwise kernels: 48 B written per visible record + the inputs read once.
#include "gspl_device.h"
#include "gspl_host.h"
#include "gspl_sort.h"

namespace gspl {

static constexpr int REC = 12;
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__global__ __launch_bounds__(256) void records_flag_kernel(int64_t total, const int32_t* __restrict__ radii, uint32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) flags[i] = radii[i] > 0 ? 1u : 0u;
}

__global__ __launch_bounds__(256) void records_pack_kernel(
    int C, int N, const int32_t* __restrict__ radii, const float* __restrict__ means2d, const float* __restrict__ depths,
    const float* __restrict__ conics, const float* __restrict__ comps, const float* __restrict__ opacities, const float* __restrict__ colors,
    const uint32_t* __restrict__ scan, float* __restrict__ records, int32_t* __restrict__ slots, int64_t* __restrict__ ends,
    int64_t* __restrict__ host_ends) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int cam = (int)(idx / N);
    const int g = (int)(idx - (int64_t)cam * N);
    const int r = radii[idx];
    const uint32_t pos = scan[idx];
    const bool vis = r > 0;
    slots[idx] = vis ? (int32_t)pos : -1;
    if (g == N - 1) {                                 // one past the camera's last record
        const int64_t e = (int64_t)pos + (vis ? 1 : 0);
        ends[cam] = e;
        if (host_ends) { host_ends[cam] = e; __threadfence_system(); }
    }
    if (!vis) return;
    float4* dst = reinterpret_cast<float4*>(records + (size_t)pos * REC);
    dst[0] = make_float4(means2d[idx * 2 + 0], means2d[idx * 2 + 1], depths[idx], conics[idx * 3 + 0]);
    dst[1] = make_float4(conics[idx * 3 + 1], conics[idx * 3 + 2], comps ? comps[idx] : 1.f, opacities[g]);
    dst[2] = make_float4(colors[idx * 3 + 0], colors[idx * 3 + 1], colors[idx * 3 + 2], __int_as_float(r));
}

// Two-phase form of the pack: the COUNT phase needs the radii only (flag -> scan -> slots and per-camera ends, the ends also into
// pinned host memory), so the sizes of the exchange are on their way to the host before the colour kernel has run; the SCATTER phase
// writes the records to the slots.  Same slots, ends and records as records_pack_kernel.
__global__ __launch_bounds__(256) void records_slots_kernel(
    int C, int N, const int32_t* __restrict__ radii, const uint32_t* __restrict__ scan, int32_t* __restrict__ slots,
    int64_t* __restrict__ ends, int64_t* __restrict__ host_ends) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int cam = (int)(idx / N);
    const int g = (int)(idx - (int64_t)cam * N);
    const uint32_t pos = scan[idx];
    const bool vis = radii[idx] > 0;
    slots[idx] = vis ? (int32_t)pos : -1;
    if (g == N - 1) {                                 // one past the camera's last record
        const int64_t e = (int64_t)pos + (vis ? 1 : 0);
        ends[cam] = e;
        if (host_ends) { host_ends[cam] = e; __threadfence_system(); }
    }
}

__global__ __launch_bounds__(256) void records_scatter_kernel(
    int C, int N, const int32_t* __restrict__ radii, const int32_t* __restrict__ slots, const float* __restrict__ means2d,
    const float* __restrict__ depths, const float* __restrict__ conics, const float* __restrict__ comps, const float* __restrict__ opacities,
    const float* __restrict__ colors, float* __restrict__ records) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int32_t pos = slots[idx];
    if (pos < 0) return;
    const int cam = (int)(idx / N);
    const int g = (int)(idx - (int64_t)cam * N);
    float4* dst = reinterpret_cast<float4*>(records + (size_t)pos * REC);
    dst[0] = make_float4(means2d[idx * 2 + 0], means2d[idx * 2 + 1], depths[idx], conics[idx * 3 + 0]);
    dst[1] = make_float4(conics[idx * 3 + 1], conics[idx * 3 + 2], comps ? comps[idx] : 1.f, opacities[g]);
    dst[2] = make_float4(colors[idx * 3 + 0], colors[idx * 3 + 1], colors[idx * 3 + 2], __int_as_float(radii[idx]));
}

// The FIXED-SIZE exchange format: one record per (camera, local splat), camera-major, rows of invisible splats zeroed (radius 0
// keeps them out of the receiver's lists); slots = the row's own index for the visible ones, -1 otherwise (what
// records_pack_bwd_kernel takes).  Nothing about it depends on a number the host would have to wait for.
__global__ __launch_bounds__(256) void records_pad_kernel(
    int C, int N, const int32_t* __restrict__ radii, const float* __restrict__ means2d, const float* __restrict__ depths,
    const float* __restrict__ conics, const float* __restrict__ comps, const float* __restrict__ opacities, const float* __restrict__ colors,
    float* __restrict__ records, int32_t* __restrict__ slots) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int cam = (int)(idx / N);
    const int g = (int)(idx - (int64_t)cam * N);
    const int r = radii[idx];
    const bool vis = r > 0;
    slots[idx] = vis ? (int32_t)idx : -1;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
    if (vis) {
        a = make_float4(means2d[idx * 2 + 0], means2d[idx * 2 + 1], depths[idx], conics[idx * 3 + 0]);
        b = make_float4(conics[idx * 3 + 1], conics[idx * 3 + 2], comps ? comps[idx] : 1.f, opacities[g]);
        c = make_float4(colors[idx * 3 + 0], colors[idx * 3 + 1], colors[idx * 3 + 2], __int_as_float(r));
    }
    float4* dst = reinterpret_cast<float4*>(records + (size_t)idx * REC);
    dst[0] = a; dst[1] = b; dst[2] = c;
}

// Every (camera, splat) row of every gradient tensor is written (zeros for the invisible ones): no memset, and the torch ops
// between the projection and the pack (compensation product, activations) never see uninitialised values.
template <bool ATOMIC_OPACITY>
__global__ __launch_bounds__(256) void records_pack_bwd_kernel(
    int C, int N, const int32_t* __restrict__ slots, const float* __restrict__ v_records,
    float* __restrict__ v_means2d, float* __restrict__ v_depths, float* __restrict__ v_conics, float* __restrict__ v_comps,
    float* __restrict__ v_opacities, float* __restrict__ v_colors) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int cam = (int)(idx / N);
    const int g = (int)(idx - (int64_t)cam * N);
    const int32_t slot = slots[idx];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
    if (slot >= 0) {
        const float4* src = reinterpret_cast<const float4*>(v_records + (size_t)slot * REC);
        a = src[0]; b = src[1]; c = src[2];
    }
    v_means2d[idx * 2 + 0] = a.x; v_means2d[idx * 2 + 1] = a.y;
    v_depths[idx] = a.z;
    v_conics[idx * 3 + 0] = a.w; v_conics[idx * 3 + 1] = b.x; v_conics[idx * 3 + 2] = b.y;
    if (v_comps) v_comps[idx] = b.z;
    if (ATOMIC_OPACITY) { if (slot >= 0) atomicAdd(v_opacities + g, b.w); }
    else v_opacities[g] = b.w;
    v_colors[idx * 3 + 0] = c.x; v_colors[idx * 3 + 1] = c.y; v_colors[idx * 3 + 2] = c.z;
}

__global__ __launch_bounds__(256) void records_unpack_kernel(
    int64_t M, int fold_compensation, const float* __restrict__ records, int32_t* __restrict__ radii, float* __restrict__ means2d,
    float* __restrict__ depths, float* __restrict__ conics, float* __restrict__ opacities, float* __restrict__ colors) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float4* src = reinterpret_cast<const float4*>(records + (size_t)i * REC);
    const float4 a = src[0], b = src[1], c = src[2];
    means2d[i * 2 + 0] = a.x; means2d[i * 2 + 1] = a.y;
    depths[i] = a.z;
    conics[i * 3 + 0] = a.w; conics[i * 3 + 1] = b.x; conics[i * 3 + 2] = b.y;
    opacities[i] = fold_compensation ? b.w * b.z : b.w;
    colors[i * 3 + 0] = c.x; colors[i * 3 + 1] = c.y; colors[i * 3 + 2] = c.z;
    radii[i] = __float_as_int(c.w);
}

__global__ __launch_bounds__(256) void records_unpack_bwd_kernel(
    int64_t M, int fold_compensation, const float* __restrict__ records,
    const float* __restrict__ v_means2d, int s2, const float* __restrict__ v_depths, const float* __restrict__ v_conics, int s3,
    const float* __restrict__ v_opacities, int s1, const float* __restrict__ v_colors, int sc, float* __restrict__ v_records) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float4 b = reinterpret_cast<const float4*>(records + (size_t)i * REC)[1];      // compensation (z), opacity (w)
    const float vo = v_opacities ? v_opacities[i * s1] : 0.f;
    float4* dst = reinterpret_cast<float4*>(v_records + (size_t)i * REC);
    const float vx = v_means2d ? v_means2d[i * s2 + 0] : 0.f, vy = v_means2d ? v_means2d[i * s2 + 1] : 0.f;
    const float va = v_conics ? v_conics[i * s3 + 0] : 0.f, vb = v_conics ? v_conics[i * s3 + 1] : 0.f, vc = v_conics ? v_conics[i * s3 + 2] : 0.f;
    dst[0] = make_float4(vx, vy, v_depths ? v_depths[i] : 0.f, va);
    dst[1] = make_float4(vb, vc, fold_compensation ? vo * b.w : 0.f, fold_compensation ? vo * b.z : vo);
    dst[2] = make_float4(v_colors ? v_colors[i * sc + 0] : 0.f, v_colors ? v_colors[i * sc + 1] : 0.f, v_colors ? v_colors[i * sc + 2] : 0.f, 0.f);
}

struct RecordsWorkspace { size_t flags_off, scan_off, tmp_off, total; };
static void plan_records(int64_t total, RecordsWorkspace& w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t n = (size_t)(total > 0 ? total : 1);
    w.flags_off = take(4 * n); w.scan_off = take(4 * n); w.tmp_off = take(exclusive_scan_u32_workspace_bytes(n));
    w.total = off;
}

}  // namespace gspl

extern "C" size_t gspl_records_workspace_bytes(int C, int N) {
    if (C < 0 || N < 0) return 0;
    gspl::RecordsWorkspace w;
    gspl::plan_records((int64_t)C * N, w);
    return w.total;
}

extern "C" int gspl_records_pack_fwd(int C, int N, const int32_t* radii, const float* means2d, const float* depths, const float* conics,
                                     const float* compensations, const float* opacities, const float* colors,
                                     float* records, int32_t* slots, int64_t* ends, int64_t* host_ends,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gspl;
    if (C < 0 || N < 0) return fail_arg("records_pack_fwd: bad sizes");
    const int64_t total = (int64_t)C * N;
    if (total == 0) return GSPL_OK;
    if (total >= (1ll << 32)) { set_error("records_pack_fwd", "more than 2^32-1 (camera, splat) pairs"); return GSPL_ERR_UNSUPPORTED; }
    if (!radii || !means2d || !depths || !conics || !opacities || !colors || !records || !slots || !ends || !workspace)
        return fail_arg("records_pack_fwd: NULL required pointer");
    RecordsWorkspace w;
    plan_records(total, w);
    if (workspace_bytes < w.total) return fail_ws("records_pack_fwd");
    char* ws = (char*)workspace;
    uint32_t* flags = (uint32_t*)(ws + w.flags_off);
    uint32_t* scan = (uint32_t*)(ws + w.scan_off);
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(records_flag_kernel, dim3(grid), dim3(256), 0, s, total, radii, flags);
    int rc = exclusive_scan_u32(flags, scan, (size_t)total, ws + w.tmp_off, s);
    if (rc != GSPL_OK) return rc;
    hipLaunchKernelGGL(records_pack_kernel, dim3(grid), dim3(256), 0, s, C, N, radii, means2d, depths, conics, compensations, opacities, colors,
                       (const uint32_t*)scan, records, slots, ends, host_ends);
    return check_launch("records_pack_fwd");
}

extern "C" int gspl_records_count_fwd(int C, int N, const int32_t* radii, int32_t* slots, int64_t* ends, int64_t* host_ends,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gspl;
    if (C < 0 || N < 0) return fail_arg("records_count_fwd: bad sizes");
    const int64_t total = (int64_t)C * N;
    if (total == 0) return GSPL_OK;
    if (total >= (1ll << 32)) { set_error("records_count_fwd", "more than 2^32-1 (camera, splat) pairs"); return GSPL_ERR_UNSUPPORTED; }
    if (!radii || !slots || !ends || !workspace) return fail_arg("records_count_fwd: NULL required pointer");
    RecordsWorkspace w;
    plan_records(total, w);
    if (workspace_bytes < w.total) return fail_ws("records_count_fwd");
    char* ws = (char*)workspace;
    uint32_t* flags = (uint32_t*)(ws + w.flags_off);
    uint32_t* scan = (uint32_t*)(ws + w.scan_off);
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(records_flag_kernel, dim3(grid), dim3(256), 0, s, total, radii, flags);
    int rc = exclusive_scan_u32(flags, scan, (size_t)total, ws + w.tmp_off, s);
    if (rc != GSPL_OK) return rc;
    hipLaunchKernelGGL(records_slots_kernel, dim3(grid), dim3(256), 0, s, C, N, radii, (const uint32_t*)scan, slots, ends, host_ends);
    return check_launch("records_count_fwd");
}

extern "C" int gspl_records_scatter_fwd(int C, int N, const int32_t* radii, const int32_t* slots, const float* means2d, const float* depths,
                                        const float* conics, const float* compensations, const float* opacities, const float* colors,
                                        float* records, void* stream) {
    using namespace gspl;
    if (C < 0 || N < 0) return fail_arg("records_scatter_fwd: bad sizes");
    const int64_t total = (int64_t)C * N;
    if (total == 0) return GSPL_OK;
    if (!radii || !slots || !means2d || !depths || !conics || !opacities || !colors || !records)
        return fail_arg("records_scatter_fwd: NULL required pointer");
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(records_scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, N, radii, slots, means2d, depths, conics,
                       compensations, opacities, colors, records);
    return check_launch("records_scatter_fwd");
}

extern "C" int gspl_records_pad_fwd(int C, int N, const int32_t* radii, const float* means2d, const float* depths, const float* conics,
                                    const float* compensations, const float* opacities, const float* colors,
                                    float* records, int32_t* slots, void* stream) {
    using namespace gspl;
    if (C < 0 || N < 0) return fail_arg("records_pad_fwd: bad sizes");
    const int64_t total = (int64_t)C * N;
    if (total == 0) return GSPL_OK;
    if (total >= (1ll << 31)) { set_error("records_pad_fwd", "more than 2^31-1 (camera, splat) pairs"); return GSPL_ERR_UNSUPPORTED; }
    if (!radii || !means2d || !depths || !conics || !opacities || !colors || !records || !slots)
        return fail_arg("records_pad_fwd: NULL required pointer");
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(records_pad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, N, radii, means2d, depths, conics, compensations,
                       opacities, colors, records, slots);
    return check_launch("records_pad_fwd");
}

extern "C" int gspl_records_pack_bwd(int C, int N, const int32_t* slots, const float* v_records,
                                     float* v_means2d, float* v_depths, float* v_conics, float* v_compensations, float* v_opacities,
                                     float* v_colors, void* stream) {
    using namespace gspl;
    if (C < 0 || N < 0) return fail_arg("records_pack_bwd: bad sizes");
    const int64_t total = (int64_t)C * N;
    if (total == 0) return GSPL_OK;
    if (!slots || !v_means2d || !v_depths || !v_conics || !v_opacities || !v_colors) return fail_arg("records_pack_bwd: NULL required pointer");
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (C == 1) {
        hipLaunchKernelGGL(records_pack_bwd_kernel<false>, dim3(grid), dim3(256), 0, s, C, N, slots, v_records, v_means2d, v_depths, v_conics,
                           v_compensations, v_opacities, v_colors);
    } else {
        hipError_t e = hipMemsetAsync(v_opacities, 0, sizeof(float) * (size_t)N, s);
        if (e != hipSuccess) return check_hip(e, "records_pack_bwd: clear");
        hipLaunchKernelGGL(records_pack_bwd_kernel<true>, dim3(grid), dim3(256), 0, s, C, N, slots, v_records, v_means2d, v_depths, v_conics,
                           v_compensations, v_opacities, v_colors);
    }
    return check_launch("records_pack_bwd");
}

extern "C" int gspl_records_unpack_fwd(int64_t M, int fold_compensation, const float* records, int32_t* radii, float* means2d, float* depths,
                                       float* conics, float* opacities, float* colors, void* stream) {
    using namespace gspl;
    if (M < 0) return fail_arg("records_unpack_fwd: bad size");
    if (M == 0) return GSPL_OK;
    if (!records || !radii || !means2d || !depths || !conics || !opacities || !colors) return fail_arg("records_unpack_fwd: NULL required pointer");
    hipLaunchKernelGGL(records_unpack_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, fold_compensation, records,
                       radii, means2d, depths, conics, opacities, colors);
    return check_launch("records_unpack_fwd");
}

extern "C" int gspl_records_unpack_bwd(int64_t M, int fold_compensation, const float* records,
                                       const float* v_means2d, int v_means2d_stride, const float* v_depths,
                                       const float* v_conics, int v_conics_stride, const float* v_opacities, int v_opacities_stride,
                                       const float* v_colors, int v_colors_stride, float* v_records, void* stream) {
    using namespace gspl;
    if (M < 0) return fail_arg("records_unpack_bwd: bad size");
    if (M == 0) return GSPL_OK;
    if (!records || !v_records) return fail_arg("records_unpack_bwd: NULL required pointer");
    const int s2 = v_means2d_stride > 0 ? v_means2d_stride : 2, s3 = v_conics_stride > 0 ? v_conics_stride : 3;
    const int s1 = v_opacities_stride > 0 ? v_opacities_stride : 1, sc = v_colors_stride > 0 ? v_colors_stride : 3;
    hipLaunchKernelGGL(records_unpack_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, fold_compensation, records,
                       v_means2d, s2, v_depths, v_conics, s3, v_opacities, s1, v_colors, sc, v_records);
    return check_launch("records_unpack_bwd");
}
