// peer.hip — direct peer-to-peer exchange of the sharded renderer's splat records over xGMI: every rank writes its records
// STRAIGHT into the receive buffer of the rank that renders that camera (IPC-mapped device memory), then raises one flag per
// destination; the receiver's stream waits for its flags.  No collective, no host round trip, no count exchange.
//
// Replaces, for the per-step exchange of HipGSplatDistributedRenderer, `torch.distributed.nn.functional.all_to_all` of the
// reference (internal/renderers/gsplat_distributed_renderer.py:141-202: two collectives for the counts and the payload, each
// ~0.1 ms of launch + protocol latency through RCCL even with nothing on the wire — DESIGN.md §6) with three small launches:
//     put     one launch: every 48-byte row to its destination buffer (16-byte stores; xGMI peer writes, or local for the own slot)
//     signal  one launch: a system-scope release fence, then the step number into every destination's flag word of THIS source
//     wait    one launch on the receiving side: one wave polls the flag words of all sources (system-scope loads) until they carry
//             the step number; the kernels behind it on the stream then read the buffer
// The buffers are FINE-GRAINED device memory (hipExtMallocWithFlags): a remote GPU's stores are visible to the owner's kernels
// without an L2 invalidation (coarse-grained memory may sit stale in the owner's L2).  Ordering: put -> (kernel boundary + fence) ->
// flag store; flag load (acquire, system scope) -> kernel boundary -> reads.  The wait is BOUNDED: after `max_polls` polls it gives up,
// raises *error and lets the stream run on (the caller checks the error word; a lost peer must not hang the GPU).
#include <cstring>
#include "gspl_device.h"
#include "gspl_host.h"

namespace gspl {

static constexpr int PEER_MAX = 16;
struct PeerPut {
    float* dst[PEER_MAX];            // destination buffers (already offset to this source's slot)
    long long begin[PEER_MAX + 1];   // rows [begin[d], begin[d + 1]) of the send buffer go to destination d
};
struct PeerFlags { unsigned long long* flag[PEER_MAX]; };

// rows of `floats4` 16-byte quads each
__global__ __launch_bounds__(256) void peer_put_kernel(PeerPut p, int n_dst, const float4* __restrict__ rows, long long total_quads, int floats4) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total_quads) return;
    const long long row = q / floats4;
    int d = 0;
#pragma unroll
    for (int k = 1; k < PEER_MAX; ++k) d += (k < n_dst && row >= p.begin[k]) ? 1 : 0;
    const long long local = q - p.begin[d] * floats4;
    reinterpret_cast<float4*>(p.dst[d])[local] = rows[q];
}

__global__ void peer_signal_kernel(PeerFlags f, int n_dst, unsigned long long value) {
    const int d = threadIdx.x;
    if (d >= n_dst) return;
    __threadfence_system();          // everything this stream wrote before (the put kernel has completed) is visible system-wide
    __hip_atomic_store(f.flag[d], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void peer_wait_kernel(const unsigned long long* __restrict__ flags, int n_src, unsigned long long value, unsigned long long max_polls,
                                 int* __restrict__ error) {
    const int s = threadIdx.x;
    bool ok = true;
    if (s < n_src) {
        unsigned long long polls = 0;
        while (__hip_atomic_load(flags + s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
            if (++polls > max_polls) { ok = false; break; }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    if (!ok) atomicExch(error, 1 + s);
    __threadfence_system();
}

}  // namespace gspl

extern "C" int gspl_peer_alloc(size_t bytes, void** ptr, void* handle_out /* 64 bytes */) {
    using namespace gspl;
    if (!ptr || !handle_out || bytes == 0) return fail_arg("peer_alloc: bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle = 64 bytes");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return check_hip(e, "peer_alloc: hipExtMallocWithFlags(fine-grained)");
    e = hipMemset(p, 0, bytes);
    // the fill of DEVICE memory may still be executing when hipMemset returns: it must be over before the handle leaves this process
    // (a late fill would wipe a peer's first flag store: a spurious wait time-out)
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); return check_hip(e, "peer_alloc: clear"); }
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); return check_hip(e, "peer_alloc: hipIpcGetMemHandle (HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)"); }
    memcpy(handle_out, &h, sizeof(h));
    *ptr = p;
    return GSPL_OK;
}

extern "C" int gspl_peer_open(const void* handle /* 64 bytes */, void** ptr) {
    using namespace gspl;
    if (!handle || !ptr) return fail_arg("peer_open: bad argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return check_hip(e, "peer_open: hipIpcOpenMemHandle");
    *ptr = p;
    return GSPL_OK;
}

extern "C" int gspl_peer_close(void* ptr) { return ptr ? gspl::check_hip(hipIpcCloseMemHandle(ptr), "peer_close") : GSPL_OK; }
extern "C" int gspl_peer_free(void* ptr) { return ptr ? gspl::check_hip(hipFree(ptr), "peer_free") : GSPL_OK; }

extern "C" int gspl_peer_put_rows(int n_dst, const float* rows, const int64_t* row_begin /* host, [n_dst + 1] */,
                                  void* const* dst /* host, [n_dst]: destination buffer + this source's slot */, int floats_per_row, void* stream) {
    using namespace gspl;
    if (n_dst < 1 || n_dst > PEER_MAX || !row_begin || !dst || floats_per_row <= 0 || (floats_per_row & 3)) return fail_arg("peer_put_rows: bad argument");
    PeerPut p;
    memset(&p, 0, sizeof(p));
    for (int d = 0; d < n_dst; ++d) {
        if (row_begin[d + 1] < row_begin[d] || (row_begin[d + 1] > row_begin[d] && !dst[d])) return fail_arg("peer_put_rows: bad ranges");
        p.dst[d] = (float*)dst[d];
        p.begin[d] = row_begin[d];
    }
    for (int d = n_dst; d <= PEER_MAX; ++d) p.begin[d] = row_begin[n_dst];
    if (row_begin[0] != 0) return fail_arg("peer_put_rows: row_begin[0] must be 0");
    const long long total = row_begin[n_dst] * (floats_per_row / 4);
    if (total == 0) return GSPL_OK;
    if (!rows) return fail_arg("peer_put_rows: NULL rows");
    hipLaunchKernelGGL(peer_put_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, n_dst, (const float4*)rows, total,
                       floats_per_row / 4);
    return check_launch("peer_put_rows");
}

extern "C" int gspl_peer_signal(int n_dst, void* const* flags /* host, [n_dst]: device pointers to the flag words to write */, uint64_t value, void* stream) {
    using namespace gspl;
    if (n_dst < 1 || n_dst > PEER_MAX || !flags) return fail_arg("peer_signal: bad argument");
    PeerFlags f;
    memset(&f, 0, sizeof(f));
    for (int d = 0; d < n_dst; ++d) {
        if (!flags[d]) return fail_arg("peer_signal: NULL flag");
        f.flag[d] = (unsigned long long*)flags[d];
    }
    hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, f, n_dst, (unsigned long long)value);
    return check_launch("peer_signal");
}

extern "C" int gspl_peer_wait(const uint64_t* flags /* device: [n_src] flag words of this rank */, int n_src, uint64_t value, uint64_t max_polls,
                              int32_t* error /* device word, 0 = fine */, void* stream) {
    using namespace gspl;
    if (n_src < 1 || n_src > PEER_MAX || !flags || !error) return fail_arg("peer_wait: bad argument");
    hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)flags, n_src, (unsigned long long)value,
                       (unsigned long long)max_polls, (int*)error);
    return check_launch("peer_wait");
}
