// knn.hip — mean squared distance to the three nearest neighbours of every point (gfx950).
//
// Replaces `simple_knn._C.distCUDA2` (yzslab/simple-knn@44f76429, un-vendored), whose one call site sets the initial
// Gaussian scales from the SfM / random point cloud: internal/models/vanilla_gaussian.py:122-125
//     dist2 = clamp_min(distCUDA2(points), 1e-7);  scales = log(sqrt(dist2))
// Definition restated (the published simple-knn behaviour): for each point, the mean of the squared Euclidean
// distances (fp32) to its 3 nearest OTHER points (a coincident point is a neighbour at distance 0).
//
// Design: exact search on a uniform grid instead of the Morton-chunk search of the CUDA package.
//   1. bounds: min/max of the cloud (float atomics on order-preserving integer images)
//   2. cells:  cell edge h = cbrt(volume * TARGET / N), grid clamped to 256^3; per-point cell id, histogram
//   3. scan of the histogram (exclusive_scan_u32 of sort.hip), counting-sort scatter of the points into cell order
//   4. search: one lane per point walks the cube shells around its cell (Chebyshev radius r = 1, 2, ...), keeping the
//      three smallest d^2; after shell r every unvisited point is farther than r*h, so the search stops as soon as
//      the third-smallest d^2 <= (r*h)^2 — exact, and ~27 cells x TARGET points for a uniform cloud.
// One-shot at model initialisation (N = 1e5 .. 1e7), so the roofline is uninteresting; it takes ~1 ms per million points.
#include <cstring>
#include <cstdlib>
#include "gspl_device.h"
#include "gspl_host.h"
#include "gspl_sort.h"

namespace gspl {

static constexpr int KNN_MAX_DIM = 256;
static constexpr float KNN_TARGET = 6.f;       // points per cell aimed at

struct KnnGrid {
    float lo[3];
    float inv_h, h;
    int dim[3];
};

// order-preserving map float -> uint (so that unsigned atomicMin/Max order floats)
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__global__ __launch_bounds__(256) void knn_bounds_init_kernel(unsigned* bounds) {
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0xFFFFFFFFu;            // min images
    else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;                // max images
}

__global__ __launch_bounds__(256) void knn_bounds_kernel(int N, const float* __restrict__ pts, unsigned* __restrict__ bounds) {
    __shared__ unsigned s_min[3], s_max[3];
    if (threadIdx.x < 3) { s_min[threadIdx.x] = 0xFFFFFFFFu; s_max[threadIdx.x] = 0u; }
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[i * 3 + a];
            if (v == v && fabsf(v) <= 3.0e38f) {       // NaN / inf coordinates do not stretch the grid
                const unsigned o = f2ord(v);
                atomicMin(&s_min[a], o);
                atomicMax(&s_max[a], o);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&bounds[threadIdx.x], s_min[threadIdx.x]); atomicMax(&bounds[3 + threadIdx.x], s_max[threadIdx.x]); }
}

// one thread: derive the grid from the bounds
__global__ void knn_grid_kernel(int N, const unsigned* __restrict__ bounds, KnnGrid* __restrict__ grid, unsigned long long capacity) {
    float lo[3], ext[3];
    for (int a = 0; a < 3; ++a) {
        const float mn = ord2f(bounds[a]), mx = ord2f(bounds[3 + a]);
        lo[a] = (mn <= mx) ? mn : 0.f;
        ext[a] = (mn <= mx) ? fmaxf(mx - mn, 0.f) : 0.f;
    }
    const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
    float h;
    if (!(emax > 0.f)) {
        h = 1.f;                                   // all points coincide (or N <= 1)
    } else {
        // degenerate axes (a planar or linear cloud) count with a small thickness so that the cell edge stays sensible
        const float floor_e = emax * 1e-3f;
        const float vol = fmaxf(ext[0], floor_e) * fmaxf(ext[1], floor_e) * fmaxf(ext[2], floor_e);
        h = cbrtf(vol * KNN_TARGET / (float)(N > 0 ? N : 1));
        h = fmaxf(h, emax / (float)KNN_MAX_DIM * 1.0001f);
    }
    int dim[3];
    for (int guard = 0; guard < 64; ++guard) {       // the cell arrays hold `capacity` cells: coarsen until the grid fits
        for (int a = 0; a < 3; ++a) dim[a] = min(KNN_MAX_DIM, max(1, (int)floorf(ext[a] / h) + 1));
        if ((unsigned long long)dim[0] * dim[1] * dim[2] <= capacity) break;
        h *= 1.26f;
    }
    grid->h = h;
    grid->inv_h = 1.f / h;
    for (int a = 0; a < 3; ++a) {
        grid->lo[a] = lo[a];
        grid->dim[a] = dim[a];
    }
}

__device__ __forceinline__ void cell_of(const KnnGrid& g, float x, float y, float z, int& cx, int& cy, int& cz) {
    // non-finite coordinates land in cell 0 (their distances are NaN/inf and never enter anybody's best three)
    const float fx = (x - g.lo[0]) * g.inv_h, fy = (y - g.lo[1]) * g.inv_h, fz = (z - g.lo[2]) * g.inv_h;
    cx = (fx == fx) ? min(g.dim[0] - 1, max(0, (int)fminf(fx, 1e6f))) : 0;
    cy = (fy == fy) ? min(g.dim[1] - 1, max(0, (int)fminf(fy, 1e6f))) : 0;
    cz = (fz == fz) ? min(g.dim[2] - 1, max(0, (int)fminf(fz, 1e6f))) : 0;
}

__global__ __launch_bounds__(256) void knn_count_kernel(int N, const float* __restrict__ pts, const KnnGrid* __restrict__ grid,
                                                        int32_t* __restrict__ cell_id, uint32_t* __restrict__ cell_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const KnnGrid g = *grid;
    int cx, cy, cz;
    cell_of(g, pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2], cx, cy, cz);
    const int c = (cz * g.dim[1] + cy) * g.dim[0] + cx;
    cell_id[i] = c;
    atomicAdd(&cell_count[c], 1u);
}

// counting-sort scatter: cell_start is the exclusive scan of the counts; cursor (zeroed) hands out the slots of a cell
__global__ __launch_bounds__(256) void knn_scatter_kernel(int N, const float* __restrict__ pts, const int32_t* __restrict__ cell_id,
                                                          const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ cursor,
                                                          float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int c = cell_id[i];
    const uint32_t slot = cell_start[c] + atomicAdd(&cursor[c], 1u);
    sorted[slot] = make_float4(pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2], __int_as_float(i));
}

__device__ __forceinline__ void push3(float d, float& b0, float& b1, float& b2) {
    if (d < b2) {
        if (d < b1) {
            b2 = b1;
            if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
        } else {
            b2 = d;
        }
    }
}

__global__ __launch_bounds__(256) void knn_search_kernel(int N, const KnnGrid* __restrict__ grid, const uint32_t* __restrict__ cell_start,
                                                         const float4* __restrict__ sorted, float* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;      // walk the points in cell order: neighbouring lanes, neighbouring cells
    if (s >= N) return;
    const KnnGrid g = *grid;
    const float4 p = sorted[s];
    const int self = __float_as_int(p.w);
    int cx, cy, cz;
    cell_of(g, p.x, p.y, p.z, cx, cy, cz);
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    const int rmax = max(max(max(cx, g.dim[0] - 1 - cx), max(cy, g.dim[1] - 1 - cy)), max(cz, g.dim[2] - 1 - cz));
    auto visit = [&](int x, int y, int z) {
        const int c = (z * g.dim[1] + y) * g.dim[0] + x;
        const uint32_t lo = cell_start[c], hi = cell_start[c + 1];
        for (uint32_t k = lo; k < hi; ++k) {
            const float4 q = sorted[k];
            if (__float_as_int(q.w) == self) continue;
            const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
            push3(dx * dx + dy * dy + dz * dz, b0, b1, b2);
        }
    };
    for (int r = 0; r <= rmax; ++r) {
        // shell of Chebyshev radius r around (cx, cy, cz), clipped to the grid
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.dim[2] - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, g.dim[1] - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, g.dim[0] - 1);
        for (int z = z0; z <= z1; ++z) {
            const bool zface = (z == cz - r) || (z == cz + r);
            for (int y = y0; y <= y1; ++y) {
                const bool yface = (y == cy - r) || (y == cy + r);
                if (zface || yface) {
                    for (int x = x0; x <= x1; ++x) visit(x, y, z);
                } else {
                    if (cx - r >= 0) visit(cx - r, y, z);
                    if (r > 0 && cx + r <= g.dim[0] - 1) visit(cx + r, y, z);
                }
            }
        }
        // everything not visited yet is farther than r*h (the visited block extends at least r cells beyond the point's own)
        const float reach = (float)r * g.h * 0.99999f;       // margin for the rounding of the cell index
        if (b2 <= reach * reach) break;
    }
    // fewer than three other points (N <= 3): average what exists; a lone point gets 0
    float sum = 0.f;
    int n = 0;
    if (b0 < INFINITY) { sum += b0; ++n; }
    if (b1 < INFINITY) { sum += b1; ++n; }
    if (b2 < INFINITY) { sum += b2; ++n; }
    out[self] = (n == 3) ? sum / 3.f : (n > 0 ? sum / (float)n : 0.f);
}

struct KnnWorkspace {
    size_t bounds_off, grid_off, cell_id_off, count_off, start_off, cursor_off, sorted_off, scan_tmp_off, scan_tmp_bytes, total;
};
static constexpr size_t KNN_MAX_CELLS = (size_t)KNN_MAX_DIM * KNN_MAX_DIM * KNN_MAX_DIM;

static inline size_t knn_align(size_t x) { return (x + 255) / 256 * 256; }

// cells actually allocated: the grid never has more than ~N / TARGET * 8 cells unless an axis hits the 256 clamp
static size_t knn_cell_capacity(int N) {
    const size_t want = (size_t)N * 4 + 4096;
    return want < KNN_MAX_CELLS ? want : KNN_MAX_CELLS;
}

static int plan_knn(int N, KnnWorkspace& w) {
    const size_t n = (size_t)(N > 0 ? N : 1), cells = knn_cell_capacity(N);
    const size_t scan_tmp = exclusive_scan_u32_workspace_bytes(cells + 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = knn_align(off + bytes); return o; };
    w.bounds_off = take(8 * 4);
    w.grid_off = take(sizeof(KnnGrid));
    w.cell_id_off = take(4 * n);
    w.count_off = take(4 * (cells + 1));
    w.start_off = take(4 * (cells + 1));
    w.cursor_off = take(4 * (cells + 1));
    w.sorted_off = take(16 * n);
    w.scan_tmp_bytes = scan_tmp;
    w.scan_tmp_off = take(scan_tmp);
    w.total = off;
    return GSPL_OK;
}

}  // namespace gspl

extern "C" size_t gspl_knn_workspace_bytes(int N) {
    gspl::KnnWorkspace w;
    if (N < 0 || gspl::plan_knn(N, w) != GSPL_OK) return 0;
    return w.total;
}

extern "C" int gspl_knn3_mean_dist2(int N, const float* points, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace gspl;
    if (N < 0) return fail_arg("knn3_mean_dist2: bad size");
    if (N == 0) return GSPL_OK;
    if (!points || !out || !workspace) return fail_arg("knn3_mean_dist2: NULL required pointer");
    KnnWorkspace w;
    int rc = plan_knn(N, w);
    if (rc != GSPL_OK) return rc;
    if (workspace_bytes < w.total) return fail_ws("knn3_mean_dist2");
    char* ws = (char*)workspace;
    hipStream_t s = (hipStream_t)stream;
    unsigned* bounds = (unsigned*)(ws + w.bounds_off);
    KnnGrid* grid = (KnnGrid*)(ws + w.grid_off);
    int32_t* cell_id = (int32_t*)(ws + w.cell_id_off);
    uint32_t* count = (uint32_t*)(ws + w.count_off);
    uint32_t* start = (uint32_t*)(ws + w.start_off);
    uint32_t* cursor = (uint32_t*)(ws + w.cursor_off);
    float4* sorted = (float4*)(ws + w.sorted_off);
    const size_t cells = knn_cell_capacity(N);
    const int grid_n = (N + 255) / 256;

    hipLaunchKernelGGL(knn_bounds_init_kernel, dim3(1), dim3(256), 0, s, bounds);
    hipLaunchKernelGGL(knn_bounds_kernel, dim3(std::min(grid_n, 1024)), dim3(256), 0, s, N, points, bounds);
    hipLaunchKernelGGL(knn_grid_kernel, dim3(1), dim3(1), 0, s, N, (const unsigned*)bounds, grid, (unsigned long long)cells);
    rc = check_launch("knn_grid");
    if (rc != GSPL_OK) return rc;
    hipError_t e = hipMemsetAsync(count, 0, 4 * (cells + 1), s);
    if (e != hipSuccess) return check_hip(e, "knn: memset");
    e = hipMemsetAsync(cursor, 0, 4 * (cells + 1), s);
    if (e != hipSuccess) return check_hip(e, "knn: memset");
    hipLaunchKernelGGL(knn_count_kernel, dim3(grid_n), dim3(256), 0, s, N, points, (const KnnGrid*)grid, cell_id, count);
    rc = check_launch("knn_count");
    if (rc != GSPL_OK) return rc;
    rc = exclusive_scan_u32((const uint32_t*)count, start, cells + 1, ws + w.scan_tmp_off, s);
    if (rc != GSPL_OK) return rc;
    hipLaunchKernelGGL(knn_scatter_kernel, dim3(grid_n), dim3(256), 0, s, N, points, (const int32_t*)cell_id, (const uint32_t*)start, cursor, sorted);
    hipLaunchKernelGGL(knn_search_kernel, dim3(grid_n), dim3(256), 0, s, N, (const KnnGrid*)grid, (const uint32_t*)start, (const float4*)sorted, out);
    return check_launch("knn_search");
}
