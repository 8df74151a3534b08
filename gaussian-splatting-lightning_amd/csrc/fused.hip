// fused.hip — ONE C-ABI call per direction for the Inria rasterizer (`gspl_rasterize_inria_fwd/bwd`).
//
// Replaces the call `diff_gaussian_rasterization.GaussianRasterizer.forward/backward` makes into its native library
// (reference call site: internal/renderers/vanilla_renderer.py:62-120; SURVEY.md §8b "fused gs_rasterize_vanilla_fwd/bwd,
// Inria argument list").  The stage entry points of this library (preprocess, binning, compositing) are orchestrated
// here, on the host side of the C boundary, instead of from Python: seven ctypes calls, two dozen torch allocations and
// their bookkeeping per forward become one call and three allocation call-backs — the Inria library's own pattern (its
// `resizeFunctional` lambdas grow three torch byte tensors: geometry, binning, image state).
//
// Nothing here launches kernels of its own; ownership stays with the caller: every buffer comes from `alloc(ctx, tag, bytes)`
// (torch's caching allocator on the Python side), nothing is hipMalloc'ed, the only persistent host resource is a small
// pinned word per thread for the one read-back of the frame (the list length that sizes the tile sort).
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>
#include "gspl_composite.h"
#include "gspl_sort.h"

namespace gspl {

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// per-splat intermediates of one frame, carved out of ONE allocation (tag GSPL_BUF_GEOMETRY)
struct GeomLayout {
    size_t radii, means2d, depths, conics, colors, clamped, cov3d, sh_jac, order, cum, big_list, spans, opac, total;
};
static GeomLayout geom_layout(size_t n) {
    GeomLayout g;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = up256(off + b); return o; };
    g.means2d = take(8 * n); g.depths = take(4 * n); g.conics = take(12 * n); g.colors = take(12 * n);
    g.clamped = take(3 * n); g.cov3d = take(24 * n); g.sh_jac = take(36 * n);
    g.order = take(4 * n); g.cum = take(8 * (n + 1)); g.big_list = take(4 * n); g.spans = take((size_t)GSPL_BIN_SPAN_BYTES * n);
    g.opac = take(4 * n);
    g.radii = 0;       // radii are an OUTPUT tensor of the call, not part of the block
    g.total = off;
    return g;
}
struct ImageLayout { size_t alphas, final_Ts, last_ids, offsets, total; };
static ImageLayout image_layout(size_t pixels, size_t tiles) {
    ImageLayout m;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = up256(off + b); return o; };
    m.alphas = take(4 * pixels); m.final_Ts = take(4 * pixels); m.last_ids = take(4 * pixels); m.offsets = take(4 * (tiles + 1));
    m.total = off;
    return m;
}

// one pinned 32-byte block per host thread for the read-back (cum[N-1] = the list length, cum[N] = n_big)
static int64_t* pinned_words() {
    static thread_local int64_t* p = nullptr;
    if (!p) {
        void* q = nullptr;
        if (hipHostMalloc(&q, 4 * sizeof(int64_t), hipHostMallocDefault) != hipSuccess) return nullptr;
        p = (int64_t*)q;
    }
    return p;
}

// The host's wait for the frame's one number (the list length): it POLLS the pinned word the scan kernel stores its ticket into.
// An event for it — hipEventRecord between the scan and the emission kernels — costs the stream a ~6 us bubble every frame.
// Bounded: every ~65 k polls the stream is queried; a stream that has drained (or failed) without the ticket arriving ends the wait.
static bool wait_for_ticket(const int64_t* host, unsigned long long ticket, hipStream_t s) {
    const volatile unsigned long long* flag = (const volatile unsigned long long*)(host + 2);
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n((const unsigned long long*)flag, __ATOMIC_ACQUIRE) == ticket) return true;
        __builtin_ia32_pause();
        if ((spins & 0xffffu) == 0u) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) return __atomic_load_n((const unsigned long long*)flag, __ATOMIC_ACQUIRE) == ticket;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); return false; }
        }
    }
}
// Segmented backward, adaptive (gspl_composite.h): one word of pinned host memory per device that the backward kernels raise when a
// tile's walk is longer than a segment (written from the device, read by the host WITHOUT a synchronisation before a later forward —
// whenever it lands), and the number of frames the segmented form stays on after the word was last seen raised.  Process-wide:
// autograd runs the backward on a thread of its own.
static std::mutex g_seg_mu;
// Per device: a ring of pinned host words the forward kernels store their verdicts into ((ticket << 1) | tail, about the PREVIOUS frame's
// backward), the view each slot's verdict is about, the verdicts per view, and the stream-level stickiness for views without one.
static constexpr unsigned SEG_RING = 16;
struct SegAdaptive {
    uint32_t* ring = nullptr;                 // pinned, SEG_RING words
    uint32_t slot_ticket[SEG_RING] = {};      // ticket a slot is waiting for (0: free)
    uintptr_t slot_view[SEG_RING] = {};       // ... and the view that verdict will be about
    uint32_t next = 1;
    uintptr_t prev_view = 0;                  // the view of the last forward (whose backward the next forward's kernel judges)
    int sticky = 0;
    std::unordered_map<uintptr_t, uint8_t> verdict;      // view -> 1 tail / 0 no tail
};
static SegAdaptive g_seg[64];
// Called by the forward of a frame that may be segmented: collect the verdicts that have landed, decide for `view`, and hand out the ring
// slot + ticket this forward's kernel reports under.
static bool seg_decide(bool force, uintptr_t view, uint32_t** slot_out, uint32_t* ticket_out) {
    *slot_out = nullptr; *ticket_out = 0u;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return force;
    std::lock_guard<std::mutex> lk(g_seg_mu);
    SegAdaptive& a = g_seg[dev];
    if (!a.ring) {
        void* q = nullptr;
        if (hipHostMalloc(&q, SEG_RING * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return force; }
        a.ring = (uint32_t*)q;
        for (unsigned i = 0; i < SEG_RING; ++i) a.ring[i] = 0u;
    }
    for (unsigned i = 0; i < SEG_RING; ++i) {
        if (!a.slot_ticket[i]) continue;
        const uint32_t v = __atomic_load_n(a.ring + i, __ATOMIC_RELAXED);
        if ((v >> 1) != (a.slot_ticket[i] & 0x7fffffffu)) continue;      // not landed yet
        const bool tail = (v & 1u) != 0u;
        if (a.slot_view[i]) {
            if (a.verdict.size() > 65536) a.verdict.clear();
            a.verdict[a.slot_view[i]] = tail ? 1 : 0;
        }
        if (tail) a.sticky = 64;
        a.slot_ticket[i] = 0u;
    }
    bool on = force;
    const auto it = view ? a.verdict.find(view) : a.verdict.end();
    if (it != a.verdict.end()) on = on || it->second != 0;
    else on = on || a.sticky > 0;
    if (a.sticky > 0) --a.sticky;
    // this forward's kernel judges the backward of the previous forward's view
    const unsigned slot = a.next % SEG_RING;
    const uint32_t ticket = a.next & 0x7fffffffu;
    ++a.next; if ((a.next & 0x7fffffffu) == 0u) a.next = 1;
    a.slot_ticket[slot] = ticket;              // (an older verdict still waited for in this slot is given up: it never landed in 16 frames)
    a.slot_view[slot] = a.prev_view;
    a.prev_view = view;
    *slot_out = a.ring + slot; *ticket_out = ticket;
    return on;
}
// the table the backward workgroups leave their walk lengths in (SEG_WALK_SLOTS rows of four words), one per device, zero at first
static uint32_t* g_seg_walk[64] = {};
static uint32_t* seg_walk_words() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_seg_mu);
    if (!g_seg_walk[dev]) {
        void* q = nullptr;
        const size_t bytes = (size_t)SEG_WALK_SLOTS * 4 * sizeof(uint32_t);
        if (hipMalloc(&q, bytes) != hipSuccess || hipMemset(q, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            if (q) (void)hipFree(q);
            return nullptr;
        }
        g_seg_walk[dev] = (uint32_t*)q;
    }
    return g_seg_walk[dev];
}

static unsigned long long next_ticket() {
    static thread_local unsigned long long t = 0ull;
    return ++t;
}

// three events per host thread (geometry done, colours done, count copied), created once
// (Measured, round 4: hipEventReleaseToDevice / hipEventDisableSystemFence on geo and col do not shorten the ~7 us the stream idles
// at an event record — it is the marker packet itself, not its fence.)
#ifndef GSPL_EVENT_FLAGS
#define GSPL_EVENT_FLAGS hipEventDisableTiming
#endif
struct FrameEvents { hipEvent_t geo = nullptr, col = nullptr, cnt = nullptr; bool ok = false; };
static FrameEvents& frame_events() {
    static thread_local FrameEvents ev;
    if (!ev.ok)
        ev.ok = hipEventCreateWithFlags(&ev.geo, GSPL_EVENT_FLAGS) == hipSuccess && hipEventCreateWithFlags(&ev.col, GSPL_EVENT_FLAGS) == hipSuccess &&
                hipEventCreateWithFlags(&ev.cnt, hipEventDisableTiming) == hipSuccess;
    return ev;
}

// A LOW-priority stream per device for the colour kernel (created once, never destroyed).  The colour kernel is a 70-90 us
// bandwidth stream that runs next to the latency-bound key pass and depth sort; at equal priority it takes half the machine
// from kernels that are on the critical path.  (torch.cuda.Stream cannot ask for a priority below the default.)
extern "C" void* gspl_low_priority_stream(void) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!streams[dev]) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&streams[dev], hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); streams[dev] = nullptr; }
    }
    return streams[dev];
}

// Optional timing of the two compositing launches INSIDE the fused calls (bench.py's roofline: the launches cannot be bracketed
// from Python any more).  Events are recorded on the launch stream; durations are read after a synchronisation.
struct ProfSlot { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };
static int g_prof_period[2] = {0, 0};   // per direction; 0: off; k: every k-th launch is timed (an event pair costs ~6 us of stream idle time per side)
static unsigned g_prof_seen[2] = {0u, 0u};
static std::mutex g_prof_mu;
static ProfSlot g_prof[2];      // 0: composite forward, 1: composite backward
struct ProfScope {
    int which; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(int w, hipStream_t st) : which(w), s(st) {
        if (g_prof_period[w] <= 0) return;
        if ((g_prof_seen[w]++ % (unsigned)g_prof_period[w]) != 0u) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, s);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof[which].ev.emplace_back(a, b);
    }
};

}  // namespace gspl

extern "C" int gspl_profile_enable2(int period_fwd, int period_bwd) {
    std::lock_guard<std::mutex> lk(gspl::g_prof_mu);
    gspl::g_prof_period[0] = period_fwd > 0 ? period_fwd : 0;
    gspl::g_prof_period[1] = period_bwd > 0 ? period_bwd : 0;
    gspl::g_prof_seen[0] = gspl::g_prof_seen[1] = 0u;
    for (auto& slot : gspl::g_prof) {
        for (auto& e : slot.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        slot.ev.clear();
    }
    return GSPL_OK;
}

extern "C" int gspl_profile_enable(int period) { return gspl_profile_enable2(period, period); }

// which: 0 = composite forward, 1 = composite backward launches of the fused calls since gspl_profile_enable(1).
// Synchronises with the recorded events; returns the number of launches and their total duration.
extern "C" int gspl_profile_read(int which, int* count, float* total_ms) {
    using namespace gspl;
    if (which < 0 || which > 1 || !count || !total_ms) return fail_arg("profile_read: bad argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    float tot = 0.f;
    for (auto& e : g_prof[which].ev) {
        float ms = 0.f;
        if (hipEventSynchronize(e.second) != hipSuccess || hipEventElapsedTime(&ms, e.first, e.second) != hipSuccess) return check_hip(hipGetLastError(), "profile_read");
        tot += ms;
    }
    *count = (int)g_prof[which].ev.size();
    *total_ms = tot;
    return GSPL_OK;
}

extern "C" size_t gspl_rasterize_inria_geometry_bytes(int N) { return gspl::geom_layout((size_t)(N > 0 ? N : 1)).total; }
extern "C" size_t gspl_rasterize_inria_image_bytes(int width, int height) {
    return gspl::image_layout((size_t)width * height, (size_t)((width + 15) / 16) * ((height + 15) / 16)).total;
}

extern "C" int gspl_rasterize_inria_fwd(
    int N, int degree, int n_coeffs,
    const float* means3D, const float* scales, const float* rotations, const float* cov3D_precomp,
    const float* shs, const float* shs_rest, const float* colors_precomp, const float* opacities,
    const float* viewmatrix, const float* projmatrix, const float* campos, const float* bg,
    int width, int height, float tanfovx, float tanfovy, float scale_modifier,
    gspl_alloc_fn alloc, void* alloc_ctx, int64_t capacity_hint,
    float* out_color, int32_t* radii, gspl_inria_state* st, void* stream, void* side_stream) {
    using namespace gspl;
    if (N < 0 || width <= 0 || height <= 0 || !alloc || !st || !out_color) return fail_arg("rasterize_inria_fwd: bad argument");
    if (N > 0 && (!means3D || !opacities || !radii || !viewmatrix || !projmatrix || !campos)) return fail_arg("rasterize_inria_fwd: NULL required pointer");
    const int tile = 16, tile_w = (width + 15) / 16, tile_h = (height + 15) / 16, n_tiles = tile_w * tile_h;
    hipStream_t s = (hipStream_t)stream, ss = (hipStream_t)side_stream;
    const int flags = st->flags;
    if (flags & ~(GSPL_INRIA_RAW_PARAMS | GSPL_INRIA_NO_SEGMENTS | GSPL_INRIA_FORCE_SEGMENTS | GSPL_INRIA_WILL_BACKWARD)) return fail_arg("rasterize_inria_fwd: unknown state->flags (zero the struct before the call)");
    const bool raw = (flags & GSPL_INRIA_RAW_PARAMS) != 0;
    if (raw && (cov3D_precomp || (N > 0 && (!scales || !rotations)))) return fail_arg("rasterize_inria_fwd: raw parameters need scales and rotations");
    memset(st, 0, sizeof(*st));
    st->flags = flags;
    st->N = N; st->width = width; st->height = height;
    const size_t n = (size_t)(N > 0 ? N : 1);
    const GeomLayout g = geom_layout(n);
    const ImageLayout im = image_layout((size_t)width * height, (size_t)n_tiles);
    char* geom = (char*)alloc(alloc_ctx, GSPL_BUF_GEOMETRY, g.total);
    char* img = (char*)alloc(alloc_ctx, GSPL_BUF_IMAGE, im.total);
    if (!geom || !img) return fail_arg("rasterize_inria_fwd: allocation call-back returned NULL");
    st->means2d = (float*)(geom + g.means2d); st->depths = (float*)(geom + g.depths); st->conics = (float*)(geom + g.conics);
    st->colors = (float*)(geom + g.colors); st->clamped = (uint8_t*)(geom + g.clamped); st->cov3d = (float*)(geom + g.cov3d);
    st->sh_jac = (float*)(geom + g.sh_jac);
    const float* raw_opacities = raw ? opacities : nullptr;
    if (raw) opacities = (float*)(geom + g.opac);          // from here on: what binning and compositing read
    st->opacities = const_cast<float*>(opacities);
    st->alphas = (float*)(img + im.alphas); st->final_Ts = (float*)(img + im.final_Ts); st->last_ids = (int32_t*)(img + im.last_ids);
    st->offsets = (int32_t*)(img + im.offsets);
    // Segmented backward (gspl_composite.h): checkpoints + the frame's counters, sized by the list CAPACITY (known before the lists
    // are: the guess, or the real length), from one allocation kept until the backward.  Off with GSPL_INRIA_NO_SEGMENTS, in the
    // deterministic mode (its backward writes one row per list entry) and when the caller's allocator says no.
    SegState seg = {};
    const bool adaptive = !(flags & GSPL_INRIA_NO_SEGMENTS) && gspl_get_deterministic() == 0;
    uint32_t* verdict_slot = nullptr;
    uint32_t verdict_ticket = 0u;
    const bool want_seg = adaptive && seg_decide((flags & GSPL_INRIA_FORCE_SEGMENTS) != 0, (uintptr_t)viewmatrix, &verdict_slot, &verdict_ticket);
    // (with or without checkpoints the forward kernel reduces the walk table the last backward left and reports its verdict)
    // the backward's packed rows, cleared by the forward's compositing kernel (GSPL_INRIA_WILL_BACKWARD, ABI 35)
    uint4* packed_zero = nullptr;
    uint32_t packed_n16 = 0u;
    if ((flags & GSPL_INRIA_WILL_BACKWARD) && N > 0) {
        const size_t bytes = ((size_t)N * 9 * sizeof(float) + 15) & ~(size_t)15;
        packed_zero = (uint4*)alloc(alloc_ctx, GSPL_BUF_PACKED, bytes);
        if (packed_zero && bytes / 16 <= 0xffffffffull) { packed_n16 = (uint32_t)(bytes / 16); st->flags |= GSPL_INRIA_PACKED_READY; }
        else packed_zero = nullptr;
    }
    auto make_seg = [&](int64_t cap) -> const SegState* {
        seg = SegState{};
        seg.zero_p = packed_zero; seg.zero_n16 = packed_n16;
        if (adaptive) { seg.walk = seg_walk_words(); seg.host_flag = verdict_slot; seg.ticket = verdict_ticket; }
        st->seg_ckpt = nullptr; st->seg_words = nullptr; st->seg_slots = 0u;
        if (!want_seg || cap <= SEG) return (adaptive || packed_zero) ? &seg : nullptr;
        const uint32_t slots = (uint32_t)(cap >> SEG_LOG2) + 2u;
        const size_t words = 2 + (size_t)slots;
        const size_t head = up256(words * sizeof(uint32_t));
        char* blk = (char*)alloc(alloc_ctx, GSPL_BUF_CHECKPOINTS, head + (size_t)slots * 256 * sizeof(float4));
        if (!blk) return (adaptive || packed_zero) ? &seg : nullptr;
        uint32_t* wds = (uint32_t*)blk;
        seg.words = wds; seg.slots = slots;      // (the item counter is cleared by the forward kernel itself)
        seg.ckpt = (float4*)(blk + head);
        st->seg_ckpt = seg.ckpt; st->seg_words = wds; st->seg_slots = slots;
        return &seg;
    };
    int32_t* order = (int32_t*)(geom + g.order);
    int64_t* cum = (int64_t*)(geom + g.cum);
    int32_t* big_list = (int32_t*)(geom + g.big_list);
    void* spans = geom + g.spans;
    int rc = GSPL_OK;
    int64_t n_isects = 0;
    if (N > 0) {
        // every block whose size is known up front is asked for BEFORE the first launch: an allocation call-back costs the host
        // 5-10 us (it runs the caller's allocator), and between two launches that is a 6 us bubble on the device
        const size_t ws1_bytes = gspl_bin_workspace_bytes(N, 0);
        char* ws1 = (char*)alloc(alloc_ctx, GSPL_BUF_BINNING, ws1_bytes);
        if (!ws1) return fail_arg("rasterize_inria_fwd: allocation call-back returned NULL");
        int64_t capacity = 0;
        char* ws2 = nullptr;
        size_t ws2_bytes = 0;
        if (capacity_hint > 0) {
            capacity = capacity_hint;
            ws2_bytes = gspl_bin_workspace_bytes(N, capacity);
            ws2 = (char*)alloc(alloc_ctx, GSPL_BUF_LISTS_WORK, ws2_bytes);
            if (!ws2) return fail_arg("rasterize_inria_fwd: allocation call-back returned NULL");
        }
        int64_t* host = pinned_words();
        if (!host) return fail_arg("rasterize_inria_fwd: no pinned host word");
        // geometry; then two independent chains: the colour (SH) kernel on the side stream, the count / depth-sort half of the
        // binning on the caller's stream; the host meanwhile waits for the one number that sizes the tile sort
        if (!viewmatrix || !projmatrix) return fail_arg("rasterize_inria_fwd: NULL required pointer");
        if (!cov3D_precomp && (!scales || !rotations)) return fail_arg("rasterize_inria_fwd: need scales+rotations or cov3D_precomp");
        // the geometry kernel clears the depth sort's tables, the scan kernel the tile sort's (when its workspace exists by then):
        // two launches fewer in front of the kernels that fill those tables
        ZeroJob zero_depth, zero_tile;
        rc = bin_depth_header(N, tile_w * tile_h, ws1, zero_depth);
        if (rc == GSPL_OK && ws2) rc = bin_tile_header(N, capacity, tile_w * tile_h, ws2, zero_tile);
        if (rc != GSPL_OK) return rc;
        rc = inria_geometry_launch(N, means3D, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, width, height, tile, tanfovx, tanfovy, scale_modifier,
                                   radii, st->means2d, st->depths, st->conics, st->cov3d, raw_opacities, st->opacities, s, zero_depth);
        if (rc != GSPL_OK) return rc;
        FrameEvents& fe = frame_events();
        if (!fe.ok) return check_hip(hipGetLastError(), "rasterize_inria_fwd: event");
        hipEvent_t ev_geo = nullptr, ev_col = nullptr;
        if (ss && ss != s) {
            ev_geo = fe.geo; ev_col = fe.col;
            (void)hipEventRecord(ev_geo, s);
            (void)hipStreamWaitEvent(ss, ev_geo, 0);
        }
        hipStream_t cs = (ss && ss != s) ? ss : s;
        rc = gspl_inria_preprocess_fwd(N, degree, n_coeffs, means3D, scales, rotations, cov3D_precomp, shs, shs_rest, colors_precomp, viewmatrix, projmatrix,
                                       campos, width, height, tile, tanfovx, tanfovy, scale_modifier, radii, st->means2d, st->depths, st->conics,
                                       st->colors, st->clamped, st->cov3d, st->sh_jac, GSPL_INRIA_COLOURS, cs);
        if (rc == GSPL_OK && ev_col) (void)hipEventRecord(ev_col, cs);
        if (rc != GSPL_OK) return rc;
        {
            const unsigned long long ticket = next_ticket();
            rc = bin_count_ticket(N, GSPL_MODE_INRIA, st->means2d, radii, st->depths, st->conics, opacities, tile, tile_w, tile_h, order, cum, big_list,
                                  spans, host, ws1, ws1_bytes, s, ticket, true, zero_tile);      // the scan kernel stores the two numbers, then the ticket, into `host`
            if (rc != GSPL_OK) return rc;
            auto wait_count = [&]() -> bool { return wait_for_ticket(host, ticket, s); };
            // speculative emission with the caller's guess of the list length, while the host waits for the real one
            if (ws2) {
                rc = bin_emit_impl(N, GSPL_MODE_INRIA, st->means2d, radii, st->conics, opacities, order, cum, big_list, spans, tile, tile_w, tile_h,
                                   capacity, ws2, ws2_bytes, s, true);
                if (rc != GSPL_OK) { (void)wait_count(); return rc; }
            }
            if (ws2) {
                // The host does not wait for the list length: the sort reads it on the device (the grid is sized by `capacity`), the
                // compositing kernel finds the end of the last list in offsets[n_tiles].  The number is looked at AFTER everything is
                // enqueued — by then the scan has long finished — and only a guess that turns out too low costs a second round.
                st->flatten_ids = (int32_t*)alloc(alloc_ctx, GSPL_BUF_LISTS, 4 * (size_t)capacity);
                if (!st->flatten_ids) { (void)wait_count(); return fail_arg("rasterize_inria_fwd: allocation call-back returned NULL"); }
                const SegState* sg = make_seg(capacity);
                rc = gspl_bin_sort_device_count(N, tile_w, tile_h, cum + (N - 1), capacity, st->flatten_ids, st->offsets, ws2, ws2_bytes, s);
                if (rc != GSPL_OK) { (void)wait_count(); return rc; }
                if (ev_col) (void)hipStreamWaitEvent(s, ev_col, 0);      // colours are ready before compositing reads them
                {
                    ProfScope prof(0, s);
                    rc = composite_fwd_impl(N, -1, 3, GSPL_MODE_INRIA, GSPL_LAYOUT_CHW, st->means2d, st->conics, st->colors, opacities, bg, width, height,
                                            tile, tile_w, tile_h, st->offsets, st->flatten_ids, out_color, st->alphas, st->final_Ts, st->last_ids, nullptr, s, sg);
                }
                const bool arrived = wait_count();
                n_isects = host[0];
                if (rc != GSPL_OK) return rc;
                if (!arrived) return check_hip(hipStreamSynchronize(s), "rasterize_inria_fwd: the list length never arrived") ? GSPL_ERR_LAUNCH : fail_arg("rasterize_inria_fwd: the list length never arrived");
                if (n_isects <= capacity) { st->n_isects = n_isects; return GSPL_OK; }
                ws2 = nullptr;                                  // too low a guess: the frame is redone below with the real length
            } else {
                if (!wait_count()) return check_hip(hipStreamSynchronize(s), "rasterize_inria_fwd: the list length never arrived") ? GSPL_ERR_LAUNCH : fail_arg("rasterize_inria_fwd: the list length never arrived");
                n_isects = host[0];
            }
            if (n_isects > (int64_t)((1u << 30) - 1u)) {
                set_error("rasterize_inria_fwd", "more than 2^30-1 (tile, Gaussian) intersections in one frame: the per-tile lists hold at most 1073741823 entries");
                return GSPL_ERR_UNSUPPORTED;
            }
            if (n_isects > 0 && (!ws2 || capacity < n_isects)) {
                capacity = n_isects;
                ws2_bytes = gspl_bin_workspace_bytes(N, capacity);
                ws2 = (char*)alloc(alloc_ctx, GSPL_BUF_LISTS_WORK, ws2_bytes);
                if (!ws2) return fail_arg("rasterize_inria_fwd: allocation call-back returned NULL");
                rc = gspl_bin_emit(N, GSPL_MODE_INRIA, st->means2d, radii, st->conics, opacities, order, cum, big_list, spans, tile, tile_w, tile_h,
                                   capacity, ws2, ws2_bytes, s);
                if (rc != GSPL_OK) return rc;
            }
            st->flatten_ids = n_isects > 0 ? (int32_t*)alloc(alloc_ctx, GSPL_BUF_LISTS, 4 * (size_t)n_isects) : nullptr;
            if (n_isects > 0 && !st->flatten_ids) return fail_arg("rasterize_inria_fwd: allocation call-back returned NULL");
            if (n_isects > 0) (void)make_seg(n_isects); else (void)make_seg(0);
            rc = gspl_bin_sort(N, tile_w, tile_h, n_isects, capacity > n_isects ? capacity : n_isects, st->flatten_ids, st->offsets, ws2, ws2_bytes, s);
            if (rc != GSPL_OK) return rc;
        }
        if (ev_col) (void)hipStreamWaitEvent(s, ev_col, 0);      // colours are ready before compositing reads them
    } else {
        rc = gspl_bin_sort(0, tile_w, tile_h, 0, 0, nullptr, st->offsets, nullptr, 0, s);
        if (rc != GSPL_OK) return rc;
    }
    st->n_isects = n_isects;
    ProfScope prof(0, s);
    return composite_fwd_impl(N, n_isects, 3, GSPL_MODE_INRIA, GSPL_LAYOUT_CHW, st->means2d, st->conics, st->colors, opacities, bg, width, height, tile,
                              tile_w, tile_h, st->offsets, st->flatten_ids, out_color, st->alphas, st->final_Ts, st->last_ids, nullptr, s,
                              (seg.ckpt || seg.walk || seg.zero_p) ? &seg : nullptr);
}

namespace gspl {
static int rasterize_inria_bwd_impl(
    int degree, int n_coeffs,
    const float* means3D, const float* scales, const float* rotations, const float* shs, const float* shs_rest, const float* opacities,
    const float* viewmatrix, const float* projmatrix, const float* campos, const float* bg,
    float tanfovx, float tanfovy, float scale_modifier,
    const int32_t* radii, const gspl_inria_state* st, const float* v_out_color,
    float* packed, uint8_t* hit_flags,
    float* v_means3D, float* v_means2D_ndc, float* v_shs, float* v_shs_rest, float* v_colors_precomp, float* v_opacities,
    float* v_scales, float* v_rotations, float* v_cov3D, void* stream, const gspl_bwd_adam_plan* plan);
}

extern "C" int gspl_rasterize_inria_bwd(
    int degree, int n_coeffs,
    const float* means3D, const float* scales, const float* rotations, const float* shs, const float* shs_rest, const float* opacities,
    const float* viewmatrix, const float* projmatrix, const float* campos, const float* bg,
    float tanfovx, float tanfovy, float scale_modifier,
    const int32_t* radii, const gspl_inria_state* st, const float* v_out_color,
    float* packed, uint8_t* hit_flags,
    float* v_means3D, float* v_means2D_ndc, float* v_shs, float* v_shs_rest, float* v_colors_precomp, float* v_opacities,
    float* v_scales, float* v_rotations, float* v_cov3D, void* stream) {
    return gspl::rasterize_inria_bwd_impl(degree, n_coeffs, means3D, scales, rotations, shs, shs_rest, opacities, viewmatrix, projmatrix, campos, bg,
                                          tanfovx, tanfovy, scale_modifier, radii, st, v_out_color, packed, hit_flags, v_means3D, v_means2D_ndc, v_shs,
                                          v_shs_rest, v_colors_precomp, v_opacities, v_scales, v_rotations, v_cov3D, stream, nullptr);
}

// The same backward with the optimizer INSIDE it: the per-Gaussian kernels at the end of the backward (SH backward,
// preprocess backward) apply the Adam update to the parameter rows they have just produced the gradient of, instead of writing
// 236 B of gradient per Gaussian for a separate optimizer launch to read back.  The parameters are updated IN PLACE; only the
// screen-space gradient (what the density controller reads, vanilla_density_controller.py:101-123) is written.
extern "C" int gspl_rasterize_inria_bwd_adam(
    int degree, int n_coeffs,
    float* means3D, float* scales, float* rotations, float* shs, float* shs_rest, float* opacities,
    const float* viewmatrix, const float* projmatrix, const float* campos, const float* bg,
    float tanfovx, float tanfovy, float scale_modifier,
    const int32_t* radii, const gspl_inria_state* st, const float* v_out_color,
    float* packed, uint8_t* hit_flags, float* scratch_means, float* v_means2D_ndc, const gspl_bwd_adam_plan* plan, void* stream) {
    if (!plan) return gspl::fail_arg("rasterize_inria_bwd_adam: NULL plan");
    if (!scales || !rotations || !shs || !opacities) return gspl::fail_arg("rasterize_inria_bwd_adam: needs scales, rotations, SH coefficients and opacities");
    return gspl::rasterize_inria_bwd_impl(degree, n_coeffs, means3D, scales, rotations, shs, shs_rest, opacities, viewmatrix, projmatrix, campos, bg,
                                          tanfovx, tanfovy, scale_modifier, radii, st, v_out_color, packed, hit_flags, scratch_means, v_means2D_ndc, shs,
                                          shs_rest, nullptr, opacities, scales, rotations, nullptr, stream, plan);
}

static int gspl::rasterize_inria_bwd_impl(
    int degree, int n_coeffs,
    const float* means3D, const float* scales, const float* rotations, const float* shs, const float* shs_rest, const float* opacities,
    const float* viewmatrix, const float* projmatrix, const float* campos, const float* bg,
    float tanfovx, float tanfovy, float scale_modifier,
    const int32_t* radii, const gspl_inria_state* st, const float* v_out_color,
    float* packed /* [N, 9] scratch */, uint8_t* hit_flags,
    float* v_means3D, float* v_means2D_ndc, float* v_shs, float* v_shs_rest, float* v_colors_precomp, float* v_opacities,
    float* v_scales, float* v_rotations, float* v_cov3D, void* stream, const gspl_bwd_adam_plan* plan) {
    using namespace gspl;
    if (!st || st->N < 0) return fail_arg("rasterize_inria_bwd: bad state");
    const int N = st->N, width = st->width, height = st->height;
    if (N == 0) return GSPL_OK;
    if (!packed || !v_out_color || !v_means3D || !v_means2D_ndc || !v_opacities) return fail_arg("rasterize_inria_bwd: NULL required pointer");
    const int tile = 16, tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (!(st->flags & GSPL_INRIA_PACKED_READY)) {      // (else `packed` is the forward's GSPL_BUF_PACKED block, cleared by its compositing kernel)
        e = hipMemsetAsync(packed, 0, (size_t)N * 9 * sizeof(float), s);      // x y | a b c | opacity | r g b
        if (e != hipSuccess) return check_hip(e, "rasterize_inria_bwd: clear");
    }
    if (hit_flags) {
        e = hipMemsetAsync(hit_flags, 0, (size_t)N, s);
        if (e != hipSuccess) return check_hip(e, "rasterize_inria_bwd: clear");
    }
    int rc = GSPL_OK;
    const bool raw = (st->flags & GSPL_INRIA_RAW_PARAMS) != 0;
    if (raw) {
        if (!st->opacities || !v_scales || !v_rotations) return fail_arg("rasterize_inria_bwd: raw parameters need the forward's state, v_scales and v_rotations");
        opacities = st->opacities;         // the activated values the forward composited with
    }
    if (st->n_isects > 0) {
        SegState seg = {};
        if (st->seg_ckpt && st->seg_words) {      // the forward left checkpoints: long walks are cut into segments (gspl_composite.h)
            seg.ckpt = (float4*)st->seg_ckpt;
            seg.words = st->seg_words; seg.slots = st->seg_slots;
        }
        if (!(st->flags & GSPL_INRIA_NO_SEGMENTS) && gspl_get_deterministic() == 0) seg.walk = seg_walk_words();      // plain or segmented: the walk lengths are left for the next forward
        ProfScope prof(1, s);
        rc = composite_bwd_packed_impl(N, st->n_isects, 3, GSPL_MODE_INRIA, GSPL_LAYOUT_CHW, st->means2d, st->conics, st->colors, opacities, bg, width,
                                       height, tile, tile_w, tile_h, st->offsets, st->flatten_ids, st->final_Ts, st->last_ids, v_out_color, nullptr,
                                       packed, 9, 0, hit_flags, s, &seg);
        if (rc != GSPL_OK) return rc;
    }
    return inria_preprocess_bwd_impl(N, degree, n_coeffs, means3D, scales, rotations, st->cov3d, shs, shs_rest, viewmatrix, projmatrix, campos, width, height,
                                     tanfovx, tanfovy, scale_modifier, radii, st->clamped, packed, packed + 2, packed + 6, 9, v_means3D, v_scales,
                                     v_rotations, v_cov3D, v_shs, v_shs_rest, v_colors_precomp, v_means2D_ndc, packed + 5, v_opacities, st->sh_jac,
                                     raw ? st->opacities : nullptr, s, plan, BwdStats{st->stats_accum, st->stats_denom, st->stats_max_radii});
}
