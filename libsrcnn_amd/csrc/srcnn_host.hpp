// srcnn_host.hpp -- host-side state shared by the C-ABI translation units (srcnn_capi.cpp: contexts, plumbing,
// the device-resident hot path; srcnn_pipeline.cpp: the host-pointer pipelines and the node-level calls).
// Internal; the public surface is include/srcnn_amd.h.
//
// One process may drive several CONTEXTS.  A context is one device binding with everything that lives in that
// device's memory: the uploaded weights, the contribution-table cache, per-stream workspaces, the host-stream slots
// and the ProcessSRCNN lanes.  srcnn_init(device) creates context 0 (the round-1/2 behaviour: one device per process);
// srcnn_init_devices(list) creates one context per list entry -- the same physical device may appear several times
// ("virtual contexts"), which is how a 1-GPU box exercises the node-level paths.  The reference's single ProcessSRCNN
// call saturates its whole machine through OpenMP (src/libsrcnn.cpp:665,791-798,817-824); with several contexts the
// large-image path of srcnn_process_u8 does the same with the node's GPUs.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstddef>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/srcnn_amd.h"
#include "../../include/srcnn_amd_debug.h"
#include "srcnn_kernels.h"
#include "srcnn_settings.hpp"

namespace srcnn {

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void set_last_error(const char* msg);
const char* last_error();

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return ::srcnn::fail(SRCNN_E_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DeviceTable {          // one uploaded AxisTable; freed only when the last reference goes
    int* first = nullptr;
    int* taps = nullptr;
    double* weight = nullptr;
    int stride = 0;
    int max_taps = 0;
    unsigned long long stamp = 0;      // LRU clock of the cache
    // host copies (tiny) so that band planners can ask "which source rows does destination range [a,b) read"
    std::vector<int> h_first, h_taps;
    bool monotone = false;             // first[] and first[]+taps[] both non-decreasing: a tile's span is given by its ends
    DeviceTable() = default;
    DeviceTable(const DeviceTable&) = delete;
    DeviceTable& operator=(const DeviceTable&) = delete;
    ~DeviceTable() { (void)hipFree(first); (void)hipFree(taps); (void)hipFree(weight); }
    DevAxisTable view() const { return DevAxisTable{first, taps, weight, stride, max_taps, monotone ? 1 : 0, h_first.data(), h_taps.data()}; }
    // source index range [lo, hi) read by destination indices [a, b)
    void source_span(unsigned a, unsigned b, unsigned& lo, unsigned& hi) const
    {
        int l = 0x7fffffff, h = 0;
        for (unsigned u = a; u < b; ++u) { l = h_first[u] < l ? h_first[u] : l; const int e = h_first[u] + h_taps[u]; h = e > h ? e : h; }
        lo = (unsigned)l; hi = (unsigned)h;
    }
};
using TableRef = std::shared_ptr<DeviceTable>;

struct Workspace {          // scratch of one stream / graph / lane; grow-only
    std::mutex mu;          // held while a call enqueues work that uses this scratch
    float* tmp = nullptr;   size_t tmp_n = 0;    // first resampler pass
    float* up = nullptr;    size_t up_n = 0;     // upscaled Y (band)
    float* c2 = nullptr;    size_t c2_n = 0;     // 32 layer-2 planes (band)
    float* planes = nullptr; size_t planes_n = 0; // colour shell: split planes / Y' / resized chroma planes
    unsigned char* bytes = nullptr; size_t bytes_n = 0;
    unsigned* queue = nullptr;                   // two words: the tile queue of k_conv12_mfma launches on this workspace's stream
    bool queue_dirty = false;                    // a call failed after launching: zero the counters before the next launch
    bool frozen = false;    // a captured graph has these pointers baked in: growing is an error
    size_t footprint() const { return sizeof(float) * (tmp_n + up_n + c2_n + planes_n) + bytes_n; }
    void release()
    {
        (void)hipFree(tmp); (void)hipFree(up); (void)hipFree(c2); (void)hipFree(planes); (void)hipFree(bytes); (void)hipFree(queue);
        tmp = up = c2 = planes = nullptr; bytes = nullptr; queue = nullptr;
        tmp_n = up_n = c2_n = planes_n = bytes_n = 0;
    }
};

struct Ctx;

// One invocation of the path: on which context and stream it runs, on which scratch, with which numerics.  The mode is
// read ONCE at the public entry point, so a concurrent srcnn_set_mode never changes a call half way through, and
// `timing` is how a graph capture tells the stage timers to stay out (event pairs cannot be timed inside a capture).
// `hold` keeps every contribution table the call launches with referenced: for an eager call until the call returns
// (the cache itself only frees after a device sync), for a graph until the graph is destroyed.
struct Call {
    Ctx* cx = nullptr;
    hipStream_t s = nullptr;
    Workspace* ws = nullptr;
    int mode = SRCNN_MODE_STRICT;       // SRCNN_MODE_* in the low byte; for SRCNN_MODE_RELAXED the SRCNN_RELAX_* mask in bits 8..11
    bool timing = true;
    std::vector<TableRef>* hold = nullptr;
    bool strict() const { return mode == SRCNN_MODE_STRICT; }
    int tier() const { return mode & 0xff; }
    // the roundings this call gives up, as RELAX_* bits for the layer launchers (0 = strict)
    int relax() const
    {
        switch (mode & 0xff) {
        case SRCNN_MODE_STRICT: return 0;
        case SRCNN_MODE_RELAXED: return (mode >> 8) & 0xf;
        default: return RELAX_FAST;
        }
    }
};

struct StageSpan { hipEvent_t a, b; int stage; };

struct StreamSlot {         // one lane of the host-stream path; lives until srcnn_shutdown
    hipStream_t st = nullptr;          // kernels (slot 0's stream carries the kernels of BOTH slots)
    hipStream_t cst = nullptr;         // this slot's copies, in both directions
    hipEvent_t e_in = nullptr, e_k = nullptr, e_out = nullptr;   // frame landed / kernels done / result copied out
    float* din = nullptr;  size_t din_n = 0;
    float* dout = nullptr; size_t dout_n = 0;
    Workspace ws;                      // private: the captured graph has its pointers baked in
    std::vector<TableRef> tables;      // references taken by eager runs (trimmed by the runs themselves)
    std::vector<TableRef> graph_tables; // references baked into `exec`: live exactly as long as the graph
    hipGraphExec_t exec = nullptr;     // captured kernel sequence for (gw, gh, gmode)
    unsigned gw = 0, gh = 0; int gmode = -1;
    unsigned uses = 0;                 // eager runs at the current shape (capture needs one first)
    int graph_verdict = 0;             // use_graph == 1 ("auto"): 0 = not measured yet, 1 = replay is cheap, 2 = replay burns host CPU: plain launches
};

// One lane of srcnn_process_u8 (the ProcessSRCNN surface).  The reference's ProcessSRCNN allocates everything per
// call and is therefore re-entrant (src/libsrcnn.cpp:628-923); here a call leases a lane -- its own compute and
// copy streams, scratch, page-locked staging and events -- on every context it uses, for its whole duration, so
// concurrent calls from several host threads never share a buffer.  Lanes are created on demand up to kMaxLanes per
// context; further callers wait for one.
struct ProcLane {
    bool busy = false;
    hipStream_t st = nullptr, copy_st = nullptr, in_st = nullptr;   // kernels / results out (D2H) / source rows in (H2D)
    Workspace ws;
    unsigned char* pin_in = nullptr;  size_t pin_in_n = 0;
    unsigned char* pin_out = nullptr; size_t pin_out_n = 0;
    std::vector<hipEvent_t> band_events;      // per band: kernels done, result landed, source rows in
    void release_buffers();
    void release();
};
constexpr size_t kDefaultMaxLanes = 4;   // per context; env SRCNN_MAX_LANES (1..64) overrides
constexpr unsigned kClockSlots = 8192;   // layer-1+2 launches the clock probe can hold before it wraps
constexpr size_t kMaxTables = 64;      // cache bound per context; only unreferenced tables are ever evicted

// Per-context buffers of the node-level tiled frame (srcnn_y_upscale2x_f32_node_dev): the slab of the source frame this
// context's band reads, and the band it produces before it is copied to the root device.
struct NodeLane {
    hipStream_t st = nullptr, copy_st = nullptr;
    Workspace ws;
    float* in = nullptr;   size_t in_n = 0;
    float* band = nullptr; size_t band_n = 0;
    std::vector<hipEvent_t> events;
    void release();
};

// Pageable host memory never goes to a HIP copy.  Above 128 KB the runtime pins the caller's pages IN PLACE for the transfer
// (a userptr mapping, 32 MB at a time); when the range lies in the malloc heap and the C library trims that heap -- any free()
// on any thread can -- the mapping is invalidated under the copy engine and the process dies with "Memory access fault by GPU
// ... on address <a heap address>" (round 6: the GPU suite, 4 runs of 4, until MALLOC_TRIM_THRESHOLD_ was raised).  So every
// copy between the device and memory the library did not page-lock itself bounces through these two page-locked slots:
// memcpy + DMA, double-buffered, at memcpy speed -- which is what the runtime's own staging path costs too.
struct HostBounce {
    std::mutex mu;                      // one bounced copy at a time per context
    unsigned char* pin = nullptr;       // 2 slots of `slot` bytes, allocated on first use
    size_t slot = 0;                    // grows with the largest copy seen, 1 MB ... kSlot: page-locking 32 MB costs 5-80 ms, and
                                        // the first copy of a process is the 50 KB weight image of srcnn_init
    hipEvent_t ev[2] = {nullptr, nullptr};
    static constexpr size_t kSlot = 16u << 20;
    void release();
};
// device buffers of the host-pointer convenience calls (srcnn_y_path_f32 and what is built on it): grow-only; a call holds
// `mu` from its H2D to its D2H, so such calls are serialised per context (they shared the NULL stream before as well)
struct HostCallBuffers {
    std::mutex mu;
    float* d_in = nullptr;  size_t d_in_n = 0;
    float* d_out = nullptr; size_t d_out_n = 0;
    void release() { (void)hipFree(d_in); (void)hipFree(d_out); d_in = d_out = nullptr; d_in_n = d_out_n = 0; }
};

struct Ctx {
    int index = 0;          // position in Global::ctxs
    int device = 0;         // physical HIP device
    int numa_node = -1;     // host NUMA node next to the device (-1: unknown)
    std::mutex mu;          // tables, ws map, spans, event pool
    std::vector<StageSpan> spans;          // recorded, not yet read
    std::vector<hipEvent_t> event_pool;    // recycled events
    double stage_ms[SRCNN_STAGE_COUNT] = {0, 0, 0};
    unsigned long long stage_n[SRCNN_STAGE_COUNT] = {0, 0, 0};
    int num_cus = 256;
    unsigned long long* clock_buf = nullptr;   // srcnn_debug_clock_probe: kClockSlots x (cycles, ticks), one slot per conv12 launch
    std::atomic<unsigned> clock_n{0};
    FusedF16Weights* fused_w = nullptr;   // device copy of the fused fp16 kernel's weight image
    std::map<std::tuple<int, unsigned, unsigned>, TableRef> tables;
    unsigned long long table_clock = 0;
    std::map<hipStream_t, std::unique_ptr<Workspace>> ws;
    StreamSlot slots[2];
    std::mutex stream_mu;               // the host-stream path is serialised per context
    std::mutex lane_mu;                 // ProcessSRCNN lanes
    std::condition_variable lane_cv;
    std::vector<std::unique_ptr<ProcLane>> lanes;
    std::mutex node_mu;                 // the node-level tiled frame is serialised per context
    NodeLane node;
    HostBounce bounce;
    HostCallBuffers host_call;
};

struct Global {
    std::mutex mu;                                   // guards ctxs (creation / shutdown) and stream_ctx
    std::vector<std::unique_ptr<Ctx>> ctxs;
    std::atomic<int> nctx{0};                        // == ctxs.size(), readable without the lock
    std::map<hipStream_t, int> stream_ctx;           // streams made by srcnn_stream_create -> owning context
    std::atomic<bool> profiling{false};
    std::atomic<bool> clock_probe{false};            // srcnn_debug_clock_probe
    std::atomic<int> mode{SRCNN_MODE_STRICT};        // as Call::mode (tier in the low byte, relaxation mask above it)
    std::atomic<unsigned> relax_mask{SRCNN_RELAX_L3_X64};   // what SRCNN_MODE_RELAXED relaxes (srcnn_set_relaxation)
    std::atomic<size_t> ws_budget;
    size_t max_lanes = kDefaultMaxLanes;   // concurrent ProcessSRCNN calls per context before callers queue
    Global();
};
extern Global& G;

// ---- tracing: named ranges for rocprofv3 --marker-trace (the reference only has a wall-clock ms counter) ----
// Off unless SRCNN_ROCTX=1: then librocprofiler-sdk-roctx is dlopen'ed once and every ProcessSRCNN call, band and stage of the
// path pushes / pops a range, so a marker trace shows the host-side pipeline next to the kernels.
class TraceRange {
public:
    explicit TraceRange(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
private:
    bool on_;
};

// ---- contexts ----
int ensure_init();                       // at least context 0 exists (lazy: device 0, or env SRCNN_DEVICES)
Ctx* cur_ctx();                          // the calling thread's current context, device bound; nullptr + error if init fails
Ctx* ctx_for_stream(void* stream);       // the context that owns `stream` (srcnn_stream_create), else the current one
int bind(Ctx& cx);                       // hipSetDevice(cx.device) for the calling thread
int context_count();
Ctx* context_at(int k);

// ---- scratch / tables ----
template <class T>
int grow(T*& p, size_t& have, size_t want)
{
    if (want <= have) return SRCNN_OK;
    if (p) {
        // kernels launched earlier (any stream) may still be using the old block: drain before freeing it
        (void)hipDeviceSynchronize();
        (void)hipFree(p);
        p = nullptr; have = 0;
    }
    void* q = nullptr;
    if (hipMalloc(&q, want * sizeof(T)) != hipSuccess)
        return fail(SRCNN_E_DEVMEM, "hipMalloc(%zu bytes) failed", want * sizeof(T));
    p = static_cast<T*>(q);
    have = want;
    return SRCNN_OK;
}

template <class T>
int grow_ws(Workspace& ws, T*& p, size_t& have, size_t want)
{
    if (want <= have) return SRCNN_OK;
    if (ws.frozen) return fail(SRCNN_E_ARG, "workspace of a captured graph cannot grow (%zu > %zu elements)", want, have);
    return grow(p, have, want);
}

int get_table(Call& c, int filter, unsigned dst_len, unsigned src_len, TableRef& out);
Workspace* workspace_for(Ctx& cx, hipStream_t s);
void* pinned_alloc(Ctx& cx, size_t bytes);      // page-locked, visible to every device, on the context's NUMA node
int grow_pinned(Ctx& cx, unsigned char*& p, size_t& have, size_t want);
bool pinned_by_library(const void* p, size_t n);   // [p, p+n) lies inside a block from srcnn_host_alloc_pinned
bool host_is_page_locked(const void* p);           // hipHostMalloc / hipHostRegister memory (asks the runtime)
// Blocking copies between device memory and ANY host memory (see HostBounce): page-locked host memory goes straight to the copy
// engine, everything else through the context's bounce slots.  `after` (may be NULL): work already queued on that stream
// is finished first, so the call is ordered on it like the hipMemcpyAsync it replaces.
int copy_h2d_any(Ctx& cx, void* d_dst, const void* h_src, size_t bytes, hipStream_t after);
int copy_d2h_any(Ctx& cx, void* h_dst, const void* d_src, size_t bytes, hipStream_t after);

// ---- the path (srcnn_capi.cpp) ----
int check_plane(const void* in, unsigned w, unsigned h, const void* out);
int check_y_path_args(const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter, const float* d_out);
int resample_rows_range(Call& c, const float* d_in, unsigned sw, unsigned sh, unsigned dw, unsigned dh, int filter,
                        unsigned r0, unsigned r1, float* d_dst);
int y_path_rows(Call& c, const YSource& src, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                unsigned r0, unsigned r1, float* d_out);
int y_path_range(Call& c, const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                 unsigned r0, unsigned r1, float* d_out);
int y_path_frame(Call& c, const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter, float* d_out);
// source rows [lo, hi) of a (w x h) plane that output rows [r0, r1) of the Y path (resample + 3 layers) depend on
int y_path_source_rows(Call& c, unsigned h, unsigned dh, int filter, unsigned r0, unsigned r1, unsigned& lo, unsigned& hi);
// rows of layer-2 scratch one band may hold under the workspace budget (>= 16), for a dw-wide output
unsigned budget_band_rows(unsigned dw);
// cut [R0,R1) into pieces of about frac[i] of the range, each moved to where it fills whole rounds of the layer-1+2 grid
std::vector<unsigned> plan_cuts(unsigned R0, unsigned R1, unsigned dw, unsigned dh, const double* frac, int nfrac, int grid, int tile_rows);
// pieces of one rank's band of a tiled 2x frame (srcnn_tiled_piece): cut points, first = band start, last = band end
std::vector<unsigned> tiled_cuts(unsigned out_w, unsigned out_h, int rank, int nranks, int npieces);

// memcpy split over a few host threads: the destination is usually a fresh new[] block whose pages fault in on first
// touch, which a single thread does at only a few GB/s.
void parallel_memcpy(void* dst, const void* src, size_t n);
void async_chain_reset();     // srcnn_shutdown: forget the chain links of asynchronous ProcessSRCNN jobs (they own events)

// Host waits.  On this runtime hipEventSynchronize / hipStreamSynchronize hold a core at 100 % for the whole wait, whatever the
// event's flags -- hipEventBlockingSync included; only the process-wide hipDeviceScheduleBlockingSync changes that, and that
// flag is the host application's to set (tools/ubench/wait_cost.hip, profiles/r03_wait_cost.txt: a 5 ms wait costs 5.0 ms of
// CPU either way, 0.03 ms when polled).  The library's threads wait for milliseconds at a time, several of them per call, so
// they poll: a burst of queries for waits that are (almost) over, then sleep-and-query in 20..100 us naps.
// SRCNN_SPIN_WAIT=1 restores the runtime's own waits (A/B runs).
// query_guard: held around every query.  A thread that captures a stream into a graph takes the same mutex for the span of
// the capture: an event query that lands inside another thread's capture of the stream the event belongs to ends that capture.
hipError_t wait_event(hipEvent_t e, std::mutex* query_guard = nullptr);
hipError_t wait_stream(hipStream_t s);

// An ordered hand-off between a producer and ONE consumer thread (replaces the round-2 yield() spin loops): the producer
// publishes "items [0, n) are ready", the consumer blocks in wait_for(i) until item i is ready or the hand-off is
// cancelled.  Nobody spins.
class Handoff {
public:
    void publish(unsigned n) { { std::lock_guard<std::mutex> lk(m_); ready_ = n; } cv_.notify_all(); }
    void cancel() { { std::lock_guard<std::mutex> lk(m_); cancelled_ = true; } cv_.notify_all(); }
    bool wait_for(unsigned i)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return ready_ > i || cancelled_; });
        return ready_ > i;
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    unsigned ready_ = 0;
    bool cancelled_ = false;
};

// Lease of one ProcessSRCNN lane of a context for the duration of a call.
struct LaneLease {
    Ctx* cx = nullptr;
    ProcLane* lane = nullptr;
    int rc = SRCNN_OK;
    explicit LaneLease(Ctx& cx);
    ~LaneLease();
    LaneLease(const LaneLease&) = delete;
    LaneLease& operator=(const LaneLease&) = delete;
};

}  // namespace srcnn
