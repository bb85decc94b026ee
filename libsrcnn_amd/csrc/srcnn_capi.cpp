// srcnn_capi.cpp -- the extern "C" boundary (include/srcnn_amd.h) over the gfx950 kernels.
//
// Host-side orchestration only: argument validation with the reference's return codes
// (src/libsrcnn.cpp:951-966), lazily built + cached contribution tables
// (src/frawscale.cpp:8-112 -> resample_table.hpp), grow-only per-stream device workspaces, and the
// launch sequence that stands in for the body of libsrcnn::doSRCNN (src/libsrcnn.cpp:628-923).
// There is deliberately no CPU compute path in this file: if HIP is unusable the calls fail.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/srcnn_amd.h"
#include "resample_table.hpp"
#include "srcnn_kernels.h"

namespace {

using namespace srcnn;

thread_local char g_err[512] = "";

}  // namespace

namespace srcnn {
// shared with srcnn_comm.cpp so that srcnn_last_error() also reports RCCL failures
void set_last_error(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
}  // namespace srcnn

namespace {

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(SRCNN_E_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

const uint32_t kWeightBits[kWeightCount] = {
#include "srcnn_weights.inc"
};

struct DeviceTable {          // one uploaded AxisTable; freed only when the last reference goes
    int* first = nullptr;
    int* taps = nullptr;
    double* weight = nullptr;
    int stride = 0;
    int max_taps = 0;
    unsigned long long stamp = 0;      // LRU clock of the cache
    DeviceTable() = default;
    DeviceTable(const DeviceTable&) = delete;
    DeviceTable& operator=(const DeviceTable&) = delete;
    ~DeviceTable() { (void)hipFree(first); (void)hipFree(taps); (void)hipFree(weight); }
    DevAxisTable view() const { return DevAxisTable{first, taps, weight, stride, max_taps}; }
};
using TableRef = std::shared_ptr<DeviceTable>;

struct Workspace {          // scratch of one stream / graph / ProcessSRCNN lane; grow-only
    std::mutex mu;          // held while a call enqueues work that uses this scratch
    float* tmp = nullptr;   size_t tmp_n = 0;    // first resampler pass
    float* up = nullptr;    size_t up_n = 0;     // upscaled Y (band)
    float* c2 = nullptr;    size_t c2_n = 0;     // 32 layer-2 planes (band)
    float* planes = nullptr; size_t planes_n = 0; // colour shell: split + resized chroma planes
    unsigned char* bytes = nullptr; size_t bytes_n = 0;
    bool frozen = false;    // a captured graph has these pointers baked in: growing is an error
    void release()
    {
        (void)hipFree(tmp); (void)hipFree(up); (void)hipFree(c2); (void)hipFree(planes); (void)hipFree(bytes);
        tmp = up = c2 = planes = nullptr; bytes = nullptr;
        tmp_n = up_n = c2_n = planes_n = bytes_n = 0;
    }
};

// One invocation of the path: where it runs, on which scratch, with which numerics.  The mode is read ONCE at the
// public entry point, so a concurrent srcnn_set_mode never changes a call half way through, and `timing` is how a
// graph capture tells the stage timers to stay out (event pairs cannot be timed inside a capture) without touching
// any process-global setting.  `hold` keeps every contribution table the call launches with referenced: for an
// eager call until the call returns (the cache itself only frees after a device sync), for a graph until the graph
// is destroyed.
struct Call {
    hipStream_t s = nullptr;
    Workspace* ws = nullptr;
    int mode = SRCNN_MODE_STRICT;
    bool timing = true;
    std::vector<TableRef>* hold = nullptr;
    bool strict() const { return mode == SRCNN_MODE_STRICT; }
};

struct StageSpan { hipEvent_t a, b; int stage; };

struct StreamSlot {         // one lane of the host-stream path; lives until srcnn_shutdown
    hipStream_t st = nullptr;          // kernels (slot 0's stream carries the kernels of BOTH slots, see the stream entry point)
    hipStream_t cst = nullptr;         // this slot's copies, in both directions
    hipEvent_t e_in = nullptr, e_k = nullptr, e_out = nullptr;   // frame landed / kernels done / result copied out
    float* din = nullptr;  size_t din_n = 0;
    float* dout = nullptr; size_t dout_n = 0;
    Workspace ws;                      // private: the captured graph has its pointers baked in
    std::vector<TableRef> tables;      // ... and these tables
    hipGraphExec_t exec = nullptr;     // captured kernel sequence for (gw, gh, gmode)
    unsigned gw = 0, gh = 0; int gmode = -1;
    unsigned uses = 0;                 // eager runs at the current shape (capture needs one first)
};

// One lane of srcnn_process_u8 (the ProcessSRCNN surface).  The reference's ProcessSRCNN allocates everything per
// call and is therefore re-entrant (src/libsrcnn.cpp:628-923); here a call leases a lane -- its own compute and
// copy streams, scratch, page-locked staging and events -- for its whole duration, so concurrent calls from several
// host threads never share a buffer.  Lanes are created on demand up to kMaxLanes; further callers wait for one.
struct ProcLane {
    bool busy = false;
    hipStream_t st = nullptr, copy_st = nullptr;
    Workspace ws;
    unsigned char* pin_in = nullptr;  size_t pin_in_n = 0;
    unsigned char* pin_out = nullptr; size_t pin_out_n = 0;
    std::vector<hipEvent_t> band_events;
    void release()
    {
        ws.release();
        if (pin_in) (void)hipHostFree(pin_in);
        if (pin_out) (void)hipHostFree(pin_out);
        pin_in = pin_out = nullptr; pin_in_n = pin_out_n = 0;
        for (auto e : band_events) (void)hipEventDestroy(e);
        band_events.clear();
        if (st) (void)hipStreamDestroy(st);
        if (copy_st) (void)hipStreamDestroy(copy_st);
        st = copy_st = nullptr;
    }
};
constexpr size_t kMaxLanes = 4;
constexpr size_t kMaxTables = 64;      // cache bound; only unreferenced tables are ever evicted

struct Context {
    std::mutex mu;
    std::atomic<bool> profiling{false};
    std::vector<StageSpan> spans;          // recorded, not yet read
    std::vector<hipEvent_t> event_pool;    // recycled events
    double stage_ms[SRCNN_STAGE_COUNT] = {0, 0, 0};
    unsigned long long stage_n[SRCNN_STAGE_COUNT] = {0, 0, 0};
    bool ready = false;
    int device = 0;
    std::atomic<int> mode{SRCNN_MODE_STRICT};
    std::atomic<size_t> ws_budget{[] {      // bytes of layer-2 scratch one band may take (srcnn_set_workspace_limit)
        const char* e = getenv("SRCNN_MAX_WORKSPACE_MB");
        const size_t mb = e ? (size_t)strtoull(e, nullptr, 10) : 16384;
        return std::max<size_t>(mb, 1) << 20;
    }()};
    int num_cus = 256;
    int conv12_variant = 1;     // SRCNN_CONV12_VARIANT: see launch_conv12_mfma
    bool conv12_valu = false;   // SRCNN_CONV12=valu selects the VALU-only layer-1+2 kernel (A/B testing)
    FusedF16Weights* fused_w = nullptr;   // device copy of the fused fp16 kernel's weight image
    bool resample_two_pass = false;   // SRCNN_RESAMPLE_2PASS=1: always the two separate resampler passes (A/B testing)
    bool f16_unfused = false;   // SRCNN_F16_UNFUSED=1: FAST_F16 as k_conv12_f16 + k_conv3_fast (A/B testing)
    std::map<std::tuple<int, unsigned, unsigned>, TableRef> tables;
    unsigned long long table_clock = 0;
    std::map<hipStream_t, std::unique_ptr<Workspace>> ws;
    StreamSlot slots[2];
    std::mutex stream_mu;               // srcnn_y_upscale2x_f32_stream is serialised
    std::mutex lane_mu;                 // ProcessSRCNN lanes
    std::condition_variable lane_cv;
    std::vector<std::unique_ptr<ProcLane>> lanes;
};

// Never destroyed: at process exit the HIP runtime may already be gone when static destructors run, and the
// tables' destructors call hipFree.  srcnn_shutdown() is the orderly way to release everything.
Context& g = *new Context;

void build_dev_weights(DevWeights& d)
{
    const float* w = reinterpret_cast<const float*>(kWeightBits);
    const float* b1 = w;
    const float* w1 = b1 + 64;          // [k][i][j]      (src/convdata.h:32-674)
    const float* b2 = w1 + 64 * 81;
    const float* w2 = b2 + 32;          // [m][f]         (src/convdata.h:686-976)
    const float* b3 = w2 + 32 * 64;
    const float* w3 = b3 + 1;           // [m][x][y], x = column offset (src/libsrcnn.cpp:512)
    memset(&d, 0, sizeof d);
    for (int k = 0; k < 64; ++k) {
        d.b1[k] = b1[k];
        for (int t = 0; t < 81; ++t) d.w1t[t][k] = w1[k * 81 + t];
    }
    for (int m = 0; m < 32; ++m) {
        d.b2[m] = b2[m];
        for (int f = 0; f < 64; ++f) d.w2[m][f] = w2[m * 64 + f];
        for (int dy = 0; dy < 5; ++dy)
            for (int dx = 0; dx < 5; ++dx) d.w3[m][dy * 5 + dx] = w3[m * 25 + dx * 5 + dy];
    }
    d.b3 = *b3;
}

// fp32 -> (hi, lo) fp16 bit patterns of v * 2^8, the same split the device uses for activations
void split_f16_bits(float v, unsigned short& hi, unsigned short& lo)
{
    const _Float16 h = (_Float16)(v * 256.f);
    const _Float16 l = (_Float16)(v * 256.f - (float)h);
    memcpy(&hi, &h, 2);
    memcpy(&lo, &l, 2);
}

void build_fused_f16_weights(const DevWeights& d, FusedF16Weights& f)
{
    memset(&f, 0, sizeof f);
    for (int s = 0; s < FU_NK; ++s)
        for (int blk = 0; blk < 2; ++blk)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int ch = 32 * blk + (l & 31), h = l >> 5;
                    int dy = -1, dx = -1;                      // tap carried by this slot (none: weight 0)
                    if (s < 4) { dy = 2 * s + h; dx = j; }
                    else { if (h == 0) { dy = 8; dx = j; } else { dy = j; dx = 8; } }
                    const float w = dy >= 0 ? d.w1t[dy * 9 + dx][ch] : 0.f;
                    split_f16_bits(w, f.w1[s][blk][0][l][j], f.w1[s][blk][1][l][j]);
                }
    for (int blk = 0; blk < 2; ++blk)
        for (int ks = 0; ks < 2; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int m = l & 31, h = l >> 5;
                    const int c = 32 * blk + 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
                    split_f16_bits(d.w2[m][c], f.w2[blk][ks][0][l][j], f.w2[blk][ks][1][l][j]);
                }
    for (int ks = 0; ks < 2; ++ks)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int t = l & 31, h = l >> 5;
                const int m = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
                split_f16_bits(t < 25 ? d.w3[m][t] : 0.f, f.w3[ks][0][l][j], f.w3[ks][1][l][j]);
            }
    for (int hf = 0; hf < 2; ++hf) {
        for (int r = 0; r < 32; ++r) f.b1[hf * 32 + r] = 256.f * d.b1[32 * (r >> 4) + 8 * ((r & 15) >> 2) + 4 * hf + (r & 3)];
        for (int r = 0; r < 16; ++r) f.b2[hf * 16 + r] = 256.f * d.b2[8 * (r >> 2) + 4 * hf + (r & 3)];
        for (int r = 0; r < 32; ++r) f.w88[hf * 32 + r] = 256.f * d.w1t[80][32 * (r >> 4) + 8 * ((r & 15) >> 2) + 4 * hf + (r & 3)];
    }
    f.b3 = d.b3;
}

int ensure_init_locked(int device)
{
    if (g.ready) {
        if (device >= 0 && device != g.device)
            return fail(SRCNN_E_ARG, "srcnn_init: already bound to device %d (asked for %d)", g.device, device);
        return SRCNN_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SRCNN_E_NODEVICE, "no HIP device visible (%s); this library has no CPU path",
                    e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device < 0) device = 0;
    if (device >= n) return fail(SRCNN_E_ARG, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SRCNN_E_NODEVICE, "device %d is %s; this build only carries gfx950 code", device, prop.gcnArchName);
    auto dw = std::make_unique<DevWeights>();
    build_dev_weights(*dw);
    HIP_TRY(upload_weights(*dw));
    HIP_TRY(conv12_mfma_prepare());
    HIP_TRY(conv12_f16_prepare());
    {
        auto fw = std::make_unique<FusedF16Weights>();
        build_fused_f16_weights(*dw, *fw);
        if (!g.fused_w) HIP_TRY(hipMalloc((void**)&g.fused_w, sizeof(FusedF16Weights)));
        HIP_TRY(hipMemcpy(g.fused_w, fw.get(), sizeof(FusedF16Weights), hipMemcpyHostToDevice));
        HIP_TRY(fused_f16_prepare());
        const char* uf = getenv("SRCNN_F16_UNFUSED");
        g.f16_unfused = uf && atoi(uf) != 0;
        const char* r2 = getenv("SRCNN_RESAMPLE_2PASS");
        g.resample_two_pass = r2 && atoi(r2) != 0;
    }
    g.num_cus = prop.multiProcessorCount;
    const char* sel = getenv("SRCNN_CONV12");
    g.conv12_valu = sel && strcmp(sel, "valu") == 0;
    const char* var = getenv("SRCNN_CONV12_VARIANT");
    g.conv12_variant = var ? atoi(var) : 1;
    g.device = device;
    g.ready = true;
    return SRCNN_OK;
}

int ensure_init()
{
    std::lock_guard<std::mutex> lk(g.mu);
    int rc = ensure_init_locked(-1);
    if (rc == SRCNN_OK) {
        // a thread other than the one that called srcnn_init still needs the device selected
        hipError_t e = hipSetDevice(g.device);
        if (e != hipSuccess) return fail(SRCNN_E_HIP, "hipSetDevice(%d) -> %s", g.device, hipGetErrorString(e));
    }
    return rc;
}

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

// Look up / build / upload the contribution table of one axis.  The cache holds one reference, the caller gets
// another (and parks it in c.hold), and a table is only ever freed when nobody but the cache references it: a
// lookup can therefore never invalidate a table handed out earlier -- not the first of the two tables of a
// resample, not one another thread is about to launch with, not one baked into a captured graph.
int get_table(Call& c, int filter, unsigned dst_len, unsigned src_len, TableRef& out)
{
    std::lock_guard<std::mutex> lk(g.mu);
    const auto key = std::make_tuple(filter, dst_len, src_len);
    auto it = g.tables.find(key);
    if (it == g.tables.end()) {
        const AxisTable t = build_axis_table(filter, dst_len, src_len);
        auto d = std::make_shared<DeviceTable>();
        d->stride = t.stride;
        d->max_taps = t.max_taps;
        HIP_TRY(hipMalloc((void**)&d->first, sizeof(int) * dst_len));
        HIP_TRY(hipMalloc((void**)&d->taps, sizeof(int) * dst_len));
        HIP_TRY(hipMalloc((void**)&d->weight, sizeof(double) * t.weight.size()));
        HIP_TRY(hipMemcpy(d->first, t.first.data(), sizeof(int) * dst_len, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->taps, t.taps.data(), sizeof(int) * dst_len, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d->weight, t.weight.data(), sizeof(double) * t.weight.size(), hipMemcpyHostToDevice));
        if (g.tables.size() >= kMaxTables) {
            // evict the least recently used tables that only the cache still references, down to half the bound.
            // Kernels launched by calls that already returned may still be reading them, hence the drain first.
            std::vector<std::pair<unsigned long long, std::tuple<int, unsigned, unsigned>>> idle;
            for (auto& kv : g.tables)
                if (kv.second.use_count() == 1) idle.emplace_back(kv.second->stamp, kv.first);
            std::sort(idle.begin(), idle.end());
            if (!idle.empty()) (void)hipDeviceSynchronize();
            for (auto& e : idle) {
                if (g.tables.size() < kMaxTables / 2) break;
                g.tables.erase(e.second);
            }
        }
        it = g.tables.emplace(key, std::move(d)).first;
    }
    it->second->stamp = ++g.table_clock;
    out = it->second;
    if (c.hold) c.hold->push_back(out);
    return SRCNN_OK;
}

Workspace* workspace_for(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g.mu);
    auto& p = g.ws[s];
    if (!p) p = std::make_unique<Workspace>();
    return p.get();
}

// RAII bracket: records an event pair around one stage on the launch stream when profiling is on.
struct StageTimer {
    hipStream_t s; int stage; hipEvent_t a = nullptr, b = nullptr; bool on;
    static hipEvent_t take()
    {
        if (!g.event_pool.empty()) { hipEvent_t e = g.event_pool.back(); g.event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    StageTimer(int stage_, const Call& c) : s(c.s), stage(stage_), on(c.timing && g.profiling.load(std::memory_order_relaxed))
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g.mu);
        a = take(); b = take();
        if (!a || !b) { on = false; return; }
        (void)hipEventRecord(a, s);
    }
    ~StageTimer()
    {
        if (!on) return;
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(g.mu);
        g.spans.push_back(StageSpan{a, b, stage});
    }
};

void drain_spans_locked()
{
    for (auto& sp : g.spans) {
        float ms = 0.f;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            g.stage_ms[sp.stage] += ms;
            g.stage_n[sp.stage] += 1;
        }
        g.event_pool.push_back(sp.a);
        g.event_pool.push_back(sp.b);
    }
    g.spans.clear();
}

// Y holds rows [y_row_base, y_row_base + y_rows) of the (W x H) upscaled plane; the kernels clamp their halo
// reads to that range as well as to the image (tile rows past the end of a band are computed but never stored).
void run_conv12(const Call& c, const float* Y, int W, int H, int y_row_base, int y_rows, float* C2, size_t plane, int row0,
                int rows)
{
    if (c.mode == SRCNN_MODE_FAST_F16) launch_conv12_f16(Y, W, H, y_row_base, y_rows, C2, plane, row0, rows, g.num_cus, c.s);
    else if (g.conv12_valu) launch_conv12(Y, W, H, y_row_base, y_rows, C2, plane, row0, rows, c.strict(), c.s);
    else launch_conv12_mfma(Y, W, H, y_row_base, y_rows, C2, plane, row0, rows, c.strict(), g.num_cus, g.conv12_variant, c.s);
}

int check_plane(const void* in, unsigned w, unsigned h, const void* out)
{
    if (!in || !out || w == 0 || h == 0) return fail(SRCNN_E_ARG, "NULL pointer or zero dimension");
    if ((unsigned long long)w * h > 0x7fffffffULL) return fail(SRCNN_E_UNSUPPORTED, "plane too large");
    return SRCNN_OK;
}

// FRAWResizeEngine::scale (src/frawscale.cpp:162-286) for destination rows [r0,r1) only.
// dst holds rows [r0,r1) (row r0 at offset 0).  tmp is scratch from the call's workspace.
int resample_rows_range(Call& c, const float* d_in, unsigned sw, unsigned sh, unsigned dw, unsigned dh, int filter,
                        unsigned r0, unsigned r1, float* d_dst)
{
    Workspace& ws = *c.ws;
    hipStream_t s = c.s;
    if (sw == dw && sh == dh) {
        // The reference's identity branch copies sizeof(unsigned short) bytes per pixel into an
        // uninitialised buffer (src/frawscale.cpp:185-193), i.e. half the plane is garbage.  We copy the
        // whole plane (the evident intent); documented in DESIGN.md as the one deliberate deviation.
        HIP_TRY(hipMemcpyAsync(d_dst, d_in + (size_t)r0 * sw, sizeof(float) * (size_t)(r1 - r0) * sw,
                               hipMemcpyDeviceToDevice, s));
        return SRCNN_OK;
    }
    TableRef tv, th;
    int rc;
    if (dw <= sw) {
        // horizontal first over all source rows, then vertical (src/frawscale.cpp:195-237)
        const float* mid = d_in;
        if (sw != dw) {
            if ((rc = get_table(c, filter, dw, sw, th))) return rc;
            if (sh != dh) {
                if ((rc = grow_ws(ws, ws.tmp, ws.tmp_n, (size_t)dw * sh))) return rc;
                launch_resample_rows(d_in, sw, ws.tmp, dw, sh, th->view(), s);
                mid = ws.tmp;
            } else {
                launch_resample_rows(d_in + (size_t)r0 * sw, sw, d_dst, dw, r1 - r0, th->view(), s);
                return SRCNN_OK;
            }
        }
        if ((rc = get_table(c, filter, dh, sh, tv))) return rc;
        launch_resample_cols(mid, dw, 0, d_dst, r0, r1 - r0, tv->view(), s);
    } else {
        // vertical first, then horizontal (src/frawscale.cpp:238-278)
        if ((rc = get_table(c, filter, dw, sw, th))) return rc;
        const float* mid = d_in + (size_t)r0 * sw;
        if (sh != dh) {
            if ((rc = get_table(c, filter, dh, sh, tv))) return rc;
            if (!g.resample_two_pass && launch_resample_2d(d_in, sw, sh, d_dst, dw, dh, r0, r1 - r0, tv->view(), th->view(), s)) return SRCNN_OK;
            if ((rc = grow_ws(ws, ws.tmp, ws.tmp_n, (size_t)sw * (r1 - r0)))) return rc;
            launch_resample_cols(d_in, sw, 0, ws.tmp, r0, r1 - r0, tv->view(), s);
            mid = ws.tmp;
        }
        launch_resample_rows(mid, sw, d_dst, dw, r1 - r0, th->view(), s);
    }
    return SRCNN_OK;
}

// resample + conv12 + conv3 for output rows [r0,r1) of the (dw x dh) result.
int y_path_rows(Call& c, const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                unsigned r0, unsigned r1, float* d_out)
{
    if (r1 > dh || r0 >= r1) return fail(SRCNN_E_ARG, "row range [%u,%u) outside 0..%u", r0, r1, dh);
    if (dh > (1u << 20) || h > (1u << 20) || dw > 0x7fffffu || (r1 - r0) > 65535u * 16u)
        return fail(SRCNN_E_UNSUPPORTED, "output %ux%u too large", dw, dh);
    Workspace& ws = *c.ws;
    // rows of layer-2 activations that conv3 touches (clamp-to-edge of the ACTIVATIONS at the true
    // border), and rows of upscaled Y that conv1 touches for those.
    const unsigned ca = r0 >= 2 ? r0 - 2 : 0, cb = std::min(dh, r1 + 2);
    const unsigned ua = ca >= 4 ? ca - 4 : 0, ub = std::min(dh, cb + 4);
    int rc;
    if ((rc = grow_ws(ws, ws.up, ws.up_n, (size_t)dw * (ub - ua)))) return rc;
    const bool fused = c.mode == SRCNN_MODE_FAST_F16 && !g.f16_unfused;
    if (fused) {
        // non-parity tier: one kernel for all three layers, no layer-2 planes at all
        {
            StageTimer t(SRCNN_STAGE_RESAMPLE, c);
            if ((rc = resample_rows_range(c, d_in, w, h, dw, dh, filter, ua, ub, ws.up))) return rc;
        }
        {
            StageTimer t(SRCNN_STAGE_CONV12, c);
            launch_fused_f16(ws.up, (int)dw, (int)dh, (int)ua, (int)(ub - ua), d_out, (int)r0, (int)(r1 - r0), g.fused_w,
                             g.num_cus, c.s);
        }
        HIP_TRY(hipGetLastError());
        return SRCNN_OK;
    }
    if ((rc = grow_ws(ws, ws.c2, ws.c2_n, (size_t)C2N * dw * (cb - ca)))) return rc;
    {
        StageTimer t(SRCNN_STAGE_RESAMPLE, c);
        if ((rc = resample_rows_range(c, d_in, w, h, dw, dh, filter, ua, ub, ws.up))) return rc;
    }
    const size_t plane = (size_t)dw * (cb - ca);
    {
        StageTimer t(SRCNN_STAGE_CONV12, c);
        run_conv12(c, ws.up, (int)dw, (int)dh, (int)ua, (int)(ub - ua), ws.c2, plane, (int)ca, (int)(cb - ca));
    }
    {
        StageTimer t(SRCNN_STAGE_CONV3, c);
        launch_conv3(ws.c2, plane, (int)dw, (int)dh, (int)ca, (int)(cb - ca), d_out, (int)r0, (int)(r1 - r0),
                     c.strict(), c.s);
    }
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

// Output rows [r0,r1).  The 32 layer-2 planes are the big scratch (128 B per output pixel).  A range whose planes
// would exceed the workspace budget (default 16 GiB, SRCNN_MAX_WORKSPACE_MB) is produced in horizontal bands --
// bit-identical to the whole range -- so a 16K x 16K output needs the same scratch as an 8K one.
int y_path_range(Call& c, const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                 unsigned r0, unsigned r1, float* d_out)
{
    const size_t budget = g.ws_budget.load();
    if (r1 > dh || r0 >= r1) return fail(SRCNN_E_ARG, "row range [%u,%u) outside 0..%u", r0, r1, dh);
    const size_t row_bytes = (size_t)C2N * dw * sizeof(float);
    const bool no_planes = c.mode == SRCNN_MODE_FAST_F16 && !g.f16_unfused;      // the fused kernel has no layer-2 planes
    if (no_planes || row_bytes * ((size_t)(r1 - r0) + 4) <= budget) return y_path_rows(c, d_in, w, h, dw, dh, filter, r0, r1, d_out);
    const size_t fit = budget / row_bytes;
    const unsigned band = (unsigned)std::max<size_t>(16, fit > 4 ? fit - 4 : 1);
    for (unsigned a = r0; a < r1; a += band) {
        const unsigned b = std::min(r1, a + band);
        int rc = y_path_rows(c, d_in, w, h, dw, dh, filter, a, b, d_out + (size_t)(a - r0) * dw);
        if (rc) return rc;
    }
    return SRCNN_OK;
}

int y_path_frame(Call& c, const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter, float* d_out)
{
    return y_path_range(c, d_in, w, h, dw, dh, filter, 0, dh, d_out);
}

int check_y_path_args(const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter, const float* d_out)
{
    int rc = check_plane(d_in, w, h, d_out);
    if (rc) return rc;
    if (dw == 0 || dh == 0) return fail(SRCNN_E_SCALE, "scaled size %ux%u", dw, dh);
    if (filter < 0 || filter > 4) return fail(SRCNN_E_ARG, "bad filter %d", filter);
    return SRCNN_OK;
}

// An eager call on a caller-visible stream: the stream's own scratch, locked while this call enqueues.
struct StreamCall {
    std::vector<TableRef> tables;
    Call c;
    std::unique_lock<std::mutex> lk;
    explicit StreamCall(void* stream)
    {
        c.s = (hipStream_t)stream;
        c.ws = workspace_for(c.s);
        c.mode = g.mode.load();
        c.hold = &tables;
        lk = std::unique_lock<std::mutex>(c.ws->mu);
    }
};

int grow_pinned(unsigned char*& p, size_t& have, size_t want)
{
    if (want <= have) return SRCNN_OK;
    if (p) { (void)hipDeviceSynchronize(); (void)hipHostFree(p); p = nullptr; have = 0; }
    void* q = nullptr;
    if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) return fail(SRCNN_E_DEVMEM, "hipHostMalloc(%zu) failed", want);
    p = static_cast<unsigned char*>(q);
    have = want;
    return SRCNN_OK;
}

// memcpy split over a few host threads: the destination is usually a fresh new[] block whose pages fault in on
// first touch, which a single thread does at only a few GB/s.
void parallel_memcpy(void* dst, const void* src, size_t n)
{
    const size_t kChunk = 4u << 20;
    unsigned nt = (unsigned)std::min<size_t>(8, n / kChunk);
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = ((n / nt) + 4095) & ~size_t(4095);
    for (unsigned t = 0; t < nt; ++t) {
        const size_t off = (size_t)t * per;
        if (off >= n) break;
        const size_t len = std::min(per, n - off);
        th.emplace_back([=] { memcpy((char*)dst + off, (const char*)src + off, len); });
    }
    for (auto& t : th) t.join();
}

// Lease of one ProcessSRCNN lane for the duration of a call.
struct LaneLease {
    ProcLane* lane = nullptr;
    int rc = SRCNN_OK;
    LaneLease()
    {
        std::unique_lock<std::mutex> lk(g.lane_mu);
        for (;;) {
            for (auto& l : g.lanes)
                if (!l->busy) { lane = l.get(); break; }
            if (lane) break;
            if (g.lanes.size() < kMaxLanes) {
                auto l = std::make_unique<ProcLane>();
                if (hipStreamCreateWithFlags(&l->st, hipStreamNonBlocking) != hipSuccess ||
                    hipStreamCreateWithFlags(&l->copy_st, hipStreamNonBlocking) != hipSuccess) {
                    l->release();
                    rc = fail(SRCNN_E_HIP, "could not create the streams of a ProcessSRCNN lane");
                    return;
                }
                g.lanes.push_back(std::move(l));
                lane = g.lanes.back().get();
                break;
            }
            g.lane_cv.wait(lk);
        }
        lane->busy = true;
    }
    ~LaneLease()
    {
        if (!lane) return;
        // nothing of this call may still be running on the lane when the next caller takes it
        (void)hipStreamSynchronize(lane->st);
        (void)hipStreamSynchronize(lane->copy_st);
        { std::lock_guard<std::mutex> lk(g.lane_mu); lane->busy = false; }
        g.lane_cv.notify_one();
    }
};

}  // namespace

// ================================================================================================
extern "C" {

int srcnn_abi_version(void) { return SRCNN_AMD_ABI_VERSION; }

const char* srcnn_last_error(void) { return g_err; }

int srcnn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int srcnn_init(int device)
{
    std::lock_guard<std::mutex> lk(g.mu);
    return ensure_init_locked(device);
}

void srcnn_shutdown(void)
{
    {
        std::lock_guard<std::mutex> lk(g.mu);
        if (!g.ready) return;
    }
    (void)hipDeviceSynchronize();
    {
        std::lock_guard<std::mutex> lk(g.lane_mu);
        for (auto& l : g.lanes) l->release();
        g.lanes.clear();
    }
    std::lock_guard<std::mutex> lk(g.mu);
    g.tables.clear();                        // graphs still alive keep their own references
    for (auto& kv : g.ws) kv.second->release();
    g.ws.clear();
    for (auto& sl : g.slots) {
        if (sl.exec) (void)hipGraphExecDestroy(sl.exec);
        if (sl.st) (void)hipStreamDestroy(sl.st);
        if (sl.cst) (void)hipStreamDestroy(sl.cst);
        for (hipEvent_t* e : {&sl.e_in, &sl.e_k, &sl.e_out}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
        sl.cst = nullptr;
        (void)hipFree(sl.din); (void)hipFree(sl.dout);
        sl.ws.release();
        sl.tables.clear();
        sl.st = nullptr; sl.din = sl.dout = nullptr; sl.din_n = sl.dout_n = 0;
        sl.exec = nullptr; sl.gw = sl.gh = 0; sl.gmode = -1; sl.uses = 0;
    }
    g.ready = false;
}

int srcnn_set_mode(int mode)
{
    if (mode != SRCNN_MODE_STRICT && mode != SRCNN_MODE_FAST && mode != SRCNN_MODE_FAST_F16) return fail(SRCNN_E_ARG, "bad mode %d", mode);
    return g.mode.exchange(mode);
}

int srcnn_get_mode(void) { return g.mode.load(); }

size_t srcnn_set_workspace_limit(size_t bytes)
{
    return g.ws_budget.exchange(std::max<size_t>(bytes, 1u << 20));
}

int srcnn_device_name(char* buf, size_t cap)
{
    int rc = ensure_init();
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, g.device));
    snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return SRCNN_OK;
}

// ---- plumbing ----------------------------------------------------------------------------------
void* srcnn_dev_alloc(size_t bytes)
{
    if (ensure_init()) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail(SRCNN_E_DEVMEM, "hipMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void srcnn_dev_free(void* p) { if (p) (void)hipFree(p); }
void* srcnn_host_alloc_pinned(size_t bytes)
{
    if (ensure_init()) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail(SRCNN_E_DEVMEM, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void srcnn_host_free_pinned(void* p) { if (p) (void)hipHostFree(p); }

int srcnn_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if (stream) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    else HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return SRCNN_OK;
}
int srcnn_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if (stream) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    else HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return SRCNN_OK;
}
int srcnn_memset_dev(void* dst, int byte, size_t bytes, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(dst, byte, bytes, (hipStream_t)stream));
    return SRCNN_OK;
}
int srcnn_stream_create(void** stream)
{
    int rc = ensure_init(); if (rc) return rc;
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return SRCNN_OK;
}
int srcnn_stream_destroy(void* stream)
{
    if (!stream) return SRCNN_OK;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    std::unique_ptr<Workspace> ws;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        auto it = g.ws.find((hipStream_t)stream);
        if (it != g.ws.end()) { ws = std::move(it->second); g.ws.erase(it); }
    }
    if (ws) { std::lock_guard<std::mutex> wl(ws->mu); ws->release(); }
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return SRCNN_OK;
}
int srcnn_stream_sync(void* stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return SRCNN_OK; }
int srcnn_device_sync(void) { int rc = ensure_init(); if (rc) return rc; HIP_TRY(hipDeviceSynchronize()); return SRCNN_OK; }
int srcnn_event_create(void** ev)
{
    int rc = ensure_init(); if (rc) return rc;
    hipEvent_t e; HIP_TRY(hipEventCreate(&e)); *ev = e; return SRCNN_OK;
}
int srcnn_event_destroy(void* ev) { if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev)); return SRCNN_OK; }
int srcnn_event_record(void* ev, void* stream) { HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return SRCNN_OK; }
int srcnn_stream_wait_event(void* stream, void* ev) { HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0)); return SRCNN_OK; }
int srcnn_event_elapsed_ms(void* start, void* stop, float* ms)
{
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SRCNN_OK;
}

// ---- the hot path ------------------------------------------------------------------------------
int srcnn_y_path_f32_dev(const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                         float* d_out, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_y_path_args(d_in, w, h, dw, dh, filter, d_out))) return rc;
    StreamCall sc(stream);
    return y_path_frame(sc.c, d_in, w, h, dw, dh, filter, d_out);
}

int srcnn_y_upscale2x_f32_dev(const float* d_in, unsigned w, unsigned h, float* d_out, void* stream)
{
    return srcnn_y_path_f32_dev(d_in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, d_out, stream);
}

namespace {
int batch_frames(Call& c, const float* d_in, unsigned w, unsigned h, unsigned nframes, float* d_out)
{
    const size_t in_n = (size_t)w * h, out_n = in_n * 4;
    for (unsigned f = 0; f < nframes; ++f) {
        int rc = y_path_frame(c, d_in + f * in_n, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, d_out + f * out_n);
        if (rc) return rc;
    }
    return SRCNN_OK;
}
}  // namespace

int srcnn_y_upscale2x_f32_batch_dev(const float* d_in, unsigned w, unsigned h, unsigned nframes, float* d_out,
                                    void* stream)
{
    if (nframes == 0) return fail(SRCNN_E_ARG, "nframes == 0");
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_y_path_args(d_in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, d_out))) return rc;
    StreamCall sc(stream);
    return batch_frames(sc.c, d_in, w, h, nframes, d_out);
}

namespace {
// A captured batch owns everything its kernel nodes point at: its scratch and its contribution tables live exactly
// as long as the handle, whatever happens to the stream's own workspace or to the table cache in the meantime.
struct BatchGraph {
    hipGraphExec_t exec = nullptr;
    hipStream_t stream = nullptr;
    Workspace ws;
    std::vector<TableRef> tables;
};
}  // namespace

int srcnn_batch_graph_create(const float* d_in, unsigned w, unsigned h, unsigned nframes, float* d_out, void* stream,
                             void** graph)
{
    if (!graph) return fail(SRCNN_E_ARG, "graph == NULL");
    if (!stream) return fail(SRCNN_E_ARG, "graph capture needs a non-default stream");
    if (nframes == 0) return fail(SRCNN_E_ARG, "nframes == 0");
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_y_path_args(d_in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, d_out))) return rc;
    auto bg = std::make_unique<BatchGraph>();
    bg->stream = (hipStream_t)stream;
    Call c;
    c.s = bg->stream; c.ws = &bg->ws; c.mode = g.mode.load(); c.hold = &bg->tables;
    auto drop = [&](int code) { (void)hipStreamSynchronize(bg->stream); bg->ws.release(); return code; };
    // eager run first: builds tables and grows the private workspace, so nothing allocates inside the capture
    if ((rc = batch_frames(c, d_in, w, h, nframes, d_out))) return drop(rc);
    if (hipStreamSynchronize(bg->stream) != hipSuccess) return drop(fail(SRCNN_E_HIP, "stream sync before capture failed"));
    bg->ws.frozen = true;
    c.timing = false;                                      // event pairs cannot be timed inside a capture
    hipGraph_t gr = nullptr;
    hipError_t e = hipStreamBeginCapture(bg->stream, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) rc = batch_frames(c, d_in, w, h, nframes, d_out);
    hipError_t e2 = hipStreamEndCapture(bg->stream, &gr);
    if (e != hipSuccess || e2 != hipSuccess) {
        if (gr) (void)hipGraphDestroy(gr);
        return drop(fail(SRCNN_E_HIP, "stream capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2)));
    }
    if (rc) { if (gr) (void)hipGraphDestroy(gr); return drop(rc); }
    e = hipGraphInstantiate(&bg->exec, gr, nullptr, nullptr, 0);
    (void)hipGraphDestroy(gr);
    if (e != hipSuccess) return drop(fail(SRCNN_E_HIP, "hipGraphInstantiate -> %s", hipGetErrorString(e)));
    *graph = bg.release();
    return SRCNN_OK;
}

int srcnn_batch_graph_launch(void* graph)
{
    if (!graph) return fail(SRCNN_E_ARG, "graph == NULL");
    BatchGraph* b = static_cast<BatchGraph*>(graph);
    HIP_TRY(hipGraphLaunch(b->exec, b->stream));
    return SRCNN_OK;
}

int srcnn_batch_graph_destroy(void* graph)
{
    if (!graph) return SRCNN_OK;
    BatchGraph* b = static_cast<BatchGraph*>(graph);
    (void)hipStreamSynchronize(b->stream);
    (void)hipGraphExecDestroy(b->exec);
    b->ws.release();
    delete b;
    return SRCNN_OK;
}

int srcnn_y_upscale2x_f32_band_dev(const float* d_in, unsigned w, unsigned h, unsigned row0, unsigned rows,
                                   float* d_out_band, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_plane(d_in, w, h, d_out_band))) return rc;
    if (rows == 0) return fail(SRCNN_E_ARG, "rows == 0");
    if ((unsigned long long)row0 + rows > 2ull * h) return fail(SRCNN_E_ARG, "band [%u,+%u) outside the %u output rows", row0, rows, 2 * h);
    StreamCall sc(stream);
    return y_path_range(sc.c, d_in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, row0, row0 + rows, d_out_band);
}

// ---- per-kernel timing -------------------------------------------------------------------------
int srcnn_profile_enable(int on) { return g.profiling.exchange(on != 0) ? 1 : 0; }

int srcnn_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g.mu);
    drain_spans_locked();
    for (int i = 0; i < SRCNN_STAGE_COUNT; ++i) { g.stage_ms[i] = 0; g.stage_n[i] = 0; }
    return SRCNN_OK;
}

int srcnn_profile_read(int stage, double* total_ms, unsigned long long* launches)
{
    if (stage < 0 || stage >= SRCNN_STAGE_COUNT) return fail(SRCNN_E_ARG, "bad stage %d", stage);
    std::lock_guard<std::mutex> lk(g.mu);
    drain_spans_locked();
    if (total_ms) *total_ms = g.stage_ms[stage];
    if (launches) *launches = g.stage_n[stage];
    return SRCNN_OK;
}

// ---- stage-level -------------------------------------------------------------------------------
int srcnn_resample_f32_dev(const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                           float* d_out, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_y_path_args(d_in, w, h, dw, dh, filter, d_out))) return rc;
    if (dh > (1u << 20) || h > (1u << 20)) return fail(SRCNN_E_UNSUPPORTED, "too many rows");
    StreamCall sc(stream);
    rc = resample_rows_range(sc.c, d_in, w, h, dw, dh, filter, 0, dh, d_out);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

int srcnn_conv1_f32_dev(const float* d_y, unsigned w, unsigned h, float* d_c1, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_plane(d_y, w, h, d_c1))) return rc;
    if (h > 65535u * 4u) return fail(SRCNN_E_UNSUPPORTED, "too many rows");
    launch_conv1_planes(d_y, (int)w, (int)h, d_c1, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

int srcnn_conv2_f32_dev(const float* d_c1, unsigned w, unsigned h, float* d_c2, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_plane(d_c1, w, h, d_c2))) return rc;
    launch_conv2_planes(d_c1, (size_t)w * h, d_c2, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

int srcnn_conv3_f32_dev(const float* d_c2, unsigned w, unsigned h, float* d_out, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_plane(d_c2, w, h, d_out))) return rc;
    if (h > 65535u * 16u) return fail(SRCNN_E_UNSUPPORTED, "too many rows");
    launch_conv3(d_c2, (size_t)w * h, (int)w, (int)h, 0, (int)h, d_out, 0, (int)h, g.mode.load() == SRCNN_MODE_STRICT,
                 (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

int srcnn_conv12_f32_dev(const float* d_y, unsigned w, unsigned h, float* d_c2, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_plane(d_y, w, h, d_c2))) return rc;
    if (h > 65535u * 4u) return fail(SRCNN_E_UNSUPPORTED, "too many rows");
    Call c;
    c.s = (hipStream_t)stream; c.mode = g.mode.load();
    run_conv12(c, d_y, (int)w, (int)h, 0, (int)h, d_c2, (size_t)w * h, 0, (int)h);
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

// ---- host-pointer conveniences -----------------------------------------------------------------
int srcnn_y_path_f32(const float* in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter, float* out)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_plane(in, w, h, out))) return rc;
    if (dw == 0 || dh == 0) return fail(SRCNN_E_SCALE, "scaled size %ux%u", dw, dh);
    float *d_in = nullptr, *d_out = nullptr;
    const size_t in_b = sizeof(float) * (size_t)w * h, out_b = sizeof(float) * (size_t)dw * dh;
    if (hipMalloc((void**)&d_in, in_b) != hipSuccess || hipMalloc((void**)&d_out, out_b) != hipSuccess) {
        hipFree(d_in);
        return fail(SRCNN_E_DEVMEM, "device allocation of %zu+%zu bytes failed", in_b, out_b);
    }
    rc = SRCNN_OK;
    if (hipMemcpy(d_in, in, in_b, hipMemcpyHostToDevice) != hipSuccess) rc = fail(SRCNN_E_HIP, "H2D copy failed");
    if (!rc) rc = srcnn_y_path_f32_dev(d_in, w, h, dw, dh, filter, d_out, nullptr);
    if (!rc && hipMemcpy(out, d_out, out_b, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SRCNN_E_HIP, "D2H copy failed");
    hipFree(d_in); hipFree(d_out);
    return rc;
}

int srcnn_y_upscale2x_f32(const float* in, unsigned w, unsigned h, float* out)
{
    return srcnn_y_path_f32(in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, out);
}

int srcnn_y_upscale2x_f32_stream(const float* in, unsigned w, unsigned h, unsigned nframes, float* out, int use_graph)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_plane(in, w, h, out))) return rc;
    if (nframes == 0) return fail(SRCNN_E_ARG, "nframes == 0");
    const size_t in_n = (size_t)w * h, out_n = in_n * 4;
    const size_t in_b = in_n * sizeof(float), out_b = out_n * sizeof(float);
    const int mode = g.mode.load();

    // page-lock the caller's frames so the copies are truly asynchronous -- unless they already are (buffers from
    // srcnn_host_alloc_pinned / hipHostMalloc: registering a gigabyte again costs milliseconds per call); harmless if it fails
    auto pinned = [](const void* p) {
        hipPointerAttribute_t a;
        const bool yes = hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        return yes;
    };
    const bool reg_in = !pinned(in) && hipHostRegister(const_cast<float*>(in), in_b * nframes, hipHostRegisterDefault) == hipSuccess;
    const bool reg_out = !pinned(out) && hipHostRegister(out, out_b * nframes, hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();

    std::lock_guard<std::mutex> slk(g.stream_mu);
    const int nslots = nframes > 1 ? 2 : 1;
    for (int i = 0; i < nslots && !rc; ++i) {
        StreamSlot& sl = g.slots[i];
        if (!sl.st && hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking) != hipSuccess) rc = fail(SRCNN_E_HIP, "stream create");
        if (!rc && !sl.cst && hipStreamCreateWithFlags(&sl.cst, hipStreamNonBlocking) != hipSuccess) rc = fail(SRCNN_E_HIP, "stream create");
        for (hipEvent_t* e : {&sl.e_in, &sl.e_k, &sl.e_out})
            if (!rc && !*e && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) rc = fail(SRCNN_E_HIP, "event create");
        if (!rc && (sl.gw != w || sl.gh != h || sl.gmode != mode)) {     // shape or mode changed: drop the graph first,
            if (sl.exec) { (void)hipStreamSynchronize(g.slots[0].st); (void)hipGraphExecDestroy(sl.exec); sl.exec = nullptr; }
            sl.ws.frozen = false;                                          // then its buffers may move again
            sl.tables.clear();
            sl.gw = w; sl.gh = h; sl.gmode = mode; sl.uses = 0;
        }
        if (!rc) rc = grow(sl.din, sl.din_n, in_n);
        if (!rc) rc = grow(sl.dout, sl.dout_n, out_n);
    }
    // Pipeline.  Copies run on the slots' copy-only streams and every copy/kernel dependency that involves a copy is
    // resolved on the HOST: a copy that has to wait on another queue's event device-side does not overlap the other
    // slot's kernels on this runtime (measured with tools/hs_probe.py, 4K frames: 12.2-12.4 ms per frame with the D2H on
    // the kernel stream or behind hipStreamWaitEvent, 10.8 ms when the host waits for the kernels and then queues the
    // copy on an idle stream).  So a copier thread waits for frame f's kernels and then issues its D2H; the main thread
    // waits for the slot's previous D2H before it reuses the slot.
    const int dev = g.device;
    std::atomic<unsigned> launched{0};     // frames whose kernels have been queued (e_k recorded)
    std::atomic<unsigned> copied{0};       // frames whose D2H has been queued (e_out recorded)
    std::atomic<int> abort_copy{0}, copy_err{0};
    std::thread copier([&] {
        (void)hipSetDevice(dev);
        for (unsigned f = 0; f < nframes; ++f) {
            while (launched.load(std::memory_order_acquire) <= f) {
                if (abort_copy.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            StreamSlot& sl = g.slots[f % nslots];
            if (hipEventSynchronize(sl.e_k) != hipSuccess ||
                hipMemcpyAsync(out + f * out_n, sl.dout, out_b, hipMemcpyDeviceToHost, sl.cst) != hipSuccess ||
                hipEventRecord(sl.e_out, sl.cst) != hipSuccess) copy_err = 1;
            copied.store(f + 1, std::memory_order_release);
        }
    });
    hipStream_t ks = g.slots[0].st;        // ALL kernels go to one stream: frames back to back, never two frames' kernels
                                           // sharing the chip (that costs more than it overlaps: the persistent layer-1+2
                                           // kernel partitions its tiles over the workgroups it expects to be resident)
    for (unsigned f = 0; f < nframes && !rc; ++f) {
        StreamSlot& sl = g.slots[f % nslots];
        Call c;
        c.s = ks; c.ws = &sl.ws; c.mode = mode; c.hold = &sl.tables;
        if (f >= (unsigned)nslots) {
            // the slot's previous frame: its kernels are done (the copier saw e_k) once its D2H has been queued; wait for
            // that D2H to finish before din / dout are reused
            while (copied.load(std::memory_order_acquire) < f - nslots + 1) std::this_thread::yield();
            if (hipEventSynchronize(sl.e_out) != hipSuccess) { rc = fail(SRCNN_E_HIP, "D2H"); break; }
        }
        // frame in: also resolved on the host (the previous frame's kernels keep the device busy meanwhile)
        if (hipMemcpyAsync(sl.din, in + f * in_n, in_b, hipMemcpyHostToDevice, sl.cst) != hipSuccess ||
            hipEventRecord(sl.e_in, sl.cst) != hipSuccess || hipEventSynchronize(sl.e_in) != hipSuccess) {
            rc = fail(SRCNN_E_HIP, "H2D"); break;
        }
        if (use_graph && sl.uses >= 1 && !sl.exec) {
            // The slot has run this shape eagerly once: tables and workspaces exist, so the kernel sequence
            // can be captured without any allocation inside the capture.
            hipGraph_t graph = nullptr;
            sl.ws.frozen = true;
            c.timing = false;              // event pairs cannot be timed inside a capture
            if (hipStreamBeginCapture(ks, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = fail(SRCNN_E_HIP, "begin capture");
            if (!rc) rc = y_path_frame(c, sl.din, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, sl.dout);
            if (hipStreamEndCapture(ks, &graph) != hipSuccess && !rc) rc = fail(SRCNN_E_HIP, "end capture");
            c.timing = true;
            if (!rc && hipGraphInstantiate(&sl.exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(SRCNN_E_HIP, "graph instantiate");
            if (graph) (void)hipGraphDestroy(graph);
            if (rc) { sl.ws.frozen = false; break; }
        }
        if (use_graph && sl.exec) {
            if (hipGraphLaunch(sl.exec, ks) != hipSuccess) { rc = fail(SRCNN_E_HIP, "graph launch"); break; }
        } else {
            if (sl.tables.size() > 16) sl.tables.clear();   // eager runs re-take their references every frame
            rc = y_path_frame(c, sl.din, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, sl.dout);
            if (rc) break;
        }
        if (hipEventRecord(sl.e_k, ks) != hipSuccess) { rc = fail(SRCNN_E_HIP, "event record"); break; }
        launched.store(f + 1, std::memory_order_release);
        ++sl.uses;
    }
    if (rc) abort_copy.store(1, std::memory_order_release);     // the copier stops at the first frame that was never launched
    copier.join();
    for (int i = 0; i < nslots; ++i) {
        if (g.slots[i].st) (void)hipStreamSynchronize(g.slots[i].st);
        if (g.slots[i].cst) (void)hipStreamSynchronize(g.slots[i].cst);
    }
    if (!rc && copy_err) rc = fail(SRCNN_E_HIP, "a device-to-host copy of the frame stream failed");
    if (reg_in) (void)hipHostUnregister(const_cast<float*>(in));
    if (reg_out) (void)hipHostUnregister(out);
    return rc;
}

int srcnn_y_upscale2x_f32_batch(const float* in, unsigned w, unsigned h, unsigned nframes, float* out)
{
    return srcnn_y_upscale2x_f32_stream(in, w, h, nframes, out, 0);
}

int srcnn_process_u8(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float multiply, int filter,
                     unsigned char* out, unsigned char* conv_opt)
{
    if (!rgb || !out || w == 0 || h == 0 || d == 0) return fail(SRCNN_E_ARG, "NULL pointer or zero dimension");
    if (d != 3 && d != 4) return fail(SRCNN_E_UNSUPPORTED, "depth %u: the reference reads uninitialised planes for d<3 (src/libsrcnn.cpp:235-236)", d);
    if ((float)w * multiply <= 0.f || (float)h * multiply <= 0.f) return fail(SRCNN_E_SCALE, "non-positive scaled size");
    if (filter < 0 || filter > 4) return fail(SRCNN_E_ARG, "bad filter %d", filter);
    int rc = ensure_init(); if (rc) return rc;
    const unsigned dw = (unsigned)((float)w * multiply), dh = (unsigned)((float)h * multiply);   // src/libsrcnn.cpp:662-663
    if (dw == 0 || dh == 0) return fail(SRCNN_E_SCALE, "scaled size %ux%u", dw, dh);
    if ((unsigned long long)w * h > 0x7fffffffULL || (unsigned long long)dw * dh > 0x7fffffffULL)
        return fail(SRCNN_E_UNSUPPORTED, "plane too large");
    const size_t n = (size_t)w * h, dn = (size_t)dw * dh;

    // Everything below runs on a lane leased for this call only (see ProcLane): concurrent ProcessSRCNN calls from
    // several host threads are independent, like the reference's.
    LaneLease lease;
    if (lease.rc) return lease.rc;
    ProcLane& L = *lease.lane;
    Workspace& ws = L.ws;
    hipStream_t s = L.st;
    std::vector<TableRef> tables;
    Call c;
    c.s = s; c.ws = &ws; c.mode = g.mode.load(); c.hold = &tables;

    // planes: [Y Cb Cr A] at source size, then [Y' Cb' Cr' A'] at destination size
    if ((rc = grow_ws(ws, ws.planes, ws.planes_n, 4 * n + 4 * dn))) return rc;
    if ((rc = grow_ws(ws, ws.bytes, ws.bytes_n, n * d + dn * d + dn))) return rc;
    float* sp[4]; float* dp[4];
    for (int k = 0; k < 4; ++k) { sp[k] = ws.planes + k * n; dp[k] = ws.planes + 4 * n + k * dn; }
    unsigned char* d_rgb = ws.bytes; unsigned char* d_out = ws.bytes + n * d; unsigned char* d_conv = d_out + dn * d;
    const bool trace = getenv("SRCNN_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const int cfilter = (filter == SRCNN_FILTER_NEAREST) ? SRCNN_FILTER_NEAREST : SRCNN_FILTER_BILINEAR;   // src/libsrcnn.cpp:701-713
    const size_t out_bytes = dn * d;

    if (out_bytes < (8u << 20)) {
        // small image: one shot on the lane's stream
        HIP_TRY(hipMemcpyAsync(d_rgb, rgb, n * d, hipMemcpyHostToDevice, s));
        launch_rgb_split(d_rgb, n, (int)d, sp[0], sp[1], sp[2], sp[3], s);
        for (unsigned k = 1; k < d; ++k)
            if ((rc = resample_rows_range(c, sp[k], w, h, dw, dh, cfilter, 0, dh, dp[k]))) return rc;
        if ((rc = y_path_frame(c, sp[0], w, h, dw, dh, filter, dp[0]))) return rc;
        launch_ycc_merge(dp[0], dp[1], dp[2], dp[3], dn, (int)d, d_out, conv_opt ? d_conv : nullptr, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
        if (conv_opt) HIP_TRY(hipMemcpyAsync(conv_opt, d_conv, dn, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return SRCNN_OK;
    }

    // Large image: the reference's only benchmark is the wall time of this call (src/test.cpp:653-672), and
    // for a GPU that is dominated by moving ~4 B per output pixel to and from pageable host memory.  So:
    // page-locked staging on both sides, the output produced in horizontal bands (bit-identical to the whole
    // frame, tests/test_gpu_parity.py::test_bands_equal_whole_frame), each band's D2H on a copy stream while the
    // next band computes, and a helper thread that fans each landed band out to the caller's buffers.
    const auto t0 = now();
    if ((rc = grow_pinned(L.pin_in, L.pin_in_n, n * d))) return rc;
    if ((rc = grow_pinned(L.pin_out, L.pin_out_n, out_bytes + dn))) return rc;
    const unsigned nb = std::max(1u, std::min(8u, dh / 256u));
    while (L.band_events.size() < 2 * nb) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        L.band_events.push_back(e);
    }
    parallel_memcpy(L.pin_in, rgb, n * d);
    HIP_TRY(hipMemcpyAsync(d_rgb, L.pin_in, n * d, hipMemcpyHostToDevice, s));
    launch_rgb_split(d_rgb, n, (int)d, sp[0], sp[1], sp[2], sp[3], s);
    for (unsigned k = 1; k < d; ++k)
        if ((rc = resample_rows_range(c, sp[k], w, h, dw, dh, cfilter, 0, dh, dp[k]))) return rc;
    const auto t1 = now();

    unsigned char* pin_rgb = L.pin_out;
    unsigned char* pin_conv = L.pin_out + out_bytes;
    std::vector<unsigned> r0s(nb + 1);
    unsigned max_band = 0;
    for (unsigned b = 0; b <= nb; ++b) r0s[b] = (unsigned)((unsigned long long)dh * b / nb);
    for (unsigned b = 0; b < nb; ++b) max_band = std::max(max_band, r0s[b + 1] - r0s[b]);
    // size the band scratch once, for the largest band plus its halos, so no band re-allocates mid-pipeline
    const bool no_planes = c.mode == SRCNN_MODE_FAST_F16 && !g.f16_unfused;           // the fused kernel has no layer-2 planes
    if (!no_planes && (rc = grow_ws(ws, ws.c2, ws.c2_n, (size_t)C2N * dw * std::min(dh, max_band + 4)))) return rc;
    if ((rc = grow_ws(ws, ws.up, ws.up_n, (size_t)dw * std::min(dh, max_band + 12)))) return rc;
    if ((rc = grow_ws(ws, ws.tmp, ws.tmp_n, (size_t)std::max(w, dw) * std::max(h, std::min(dh, max_band + 12))))) return rc;
    const int dev = g.device;
    std::atomic<int> copy_err{0};
    std::atomic<unsigned> enqueued{0};      // bands whose kernels have been queued (their "computed" event recorded) in THIS call
    std::atomic<int> abort_bands{0};
    // The helper resolves the copy dependencies on the HOST: it waits for a band's kernels, then queues the band's D2H on
    // the idle copy stream (a copy that waits device-side on the kernel stream's event does not run beside the next band's
    // kernels on this runtime, profiles/r02_stream_overlap.txt), and fans the previous band out to the caller's buffers
    // while that copy is in flight.
    std::thread fanout([&] {
        (void)hipSetDevice(dev);
        auto fan = [&](unsigned b) {
            if (hipEventSynchronize(L.band_events[2 * b + 1]) != hipSuccess) { copy_err = 1; return; }
            const size_t p0 = (size_t)r0s[b] * dw, p1 = (size_t)r0s[b + 1] * dw;
            parallel_memcpy(out + p0 * d, pin_rgb + p0 * d, (p1 - p0) * d);
            if (conv_opt) parallel_memcpy(conv_opt + p0, pin_conv + p0, p1 - p0);
        };
        unsigned done = 0;
        for (unsigned b = 0; b < nb; ++b) {
            while (enqueued.load(std::memory_order_acquire) <= b) {
                if (abort_bands.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            const size_t p0 = (size_t)r0s[b] * dw, pn = (size_t)(r0s[b + 1] - r0s[b]) * dw;
            if (hipEventSynchronize(L.band_events[2 * b]) != hipSuccess ||
                hipMemcpyAsync(pin_rgb + p0 * d, d_out + p0 * d, pn * d, hipMemcpyDeviceToHost, L.copy_st) != hipSuccess ||
                (conv_opt && hipMemcpyAsync(pin_conv + p0, d_conv + p0, pn, hipMemcpyDeviceToHost, L.copy_st) != hipSuccess) ||
                hipEventRecord(L.band_events[2 * b + 1], L.copy_st) != hipSuccess) { copy_err = 1; return; }
            if (b > 0) fan(b - 1);
            done = b;
        }
        fan(done);
    });
    int launch_rc = SRCNN_OK;
    for (unsigned b = 0; b < nb; ++b) {
        const unsigned r0 = r0s[b], r1 = r0s[b + 1];
        const size_t p0 = (size_t)r0 * dw, pn = (size_t)(r1 - r0) * dw;
        launch_rc = y_path_rows(c, sp[0], w, h, dw, dh, filter, r0, r1, dp[0] + p0);
        if (!launch_rc) {
            launch_ycc_merge(dp[0] + p0, dp[1] + p0, dp[2] + p0, dp[3] + p0, pn, (int)d, d_out + p0 * d,
                             conv_opt ? d_conv + p0 : nullptr, s);
            if (hipEventRecord(L.band_events[2 * b], s) != hipSuccess) launch_rc = fail(SRCNN_E_HIP, "band %u event record failed", b);
        }
        if (launch_rc) { abort_bands.store(1, std::memory_order_release); break; }
        enqueued.store(b + 1, std::memory_order_release);
    }
    fanout.join();
    const auto t2 = now();
    hipError_t e1 = hipStreamSynchronize(s), e2 = hipStreamSynchronize(L.copy_st);
    if (launch_rc) return launch_rc;
    if (e1 != hipSuccess || e2 != hipSuccess || copy_err) return fail(SRCNN_E_HIP, "pipeline failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    HIP_TRY(hipGetLastError());
    if (trace)
        fprintf(stderr, "srcnn_process_u8 %ux%ux%u x%.2f: stage-in+chroma %.2f ms, %u bands (compute || D2H || fan-out) %.2f ms\n",
                w, h, d, (double)multiply, ms(t0, t1), nb, ms(t1, t2));
    return SRCNN_OK;
}

int srcnn_axis_table(int filter, unsigned dst_len, unsigned src_len, int* left, int* right, double* weights)
{
    if (dst_len == 0 || src_len == 0) return fail(SRCNN_E_ARG, "zero length");
    const srcnn::AxisTable t = srcnn::build_axis_table(filter, dst_len, src_len);
    if (left) memcpy(left, t.first.data(), sizeof(int) * dst_len);
    if (right) memcpy(right, t.last.data(), sizeof(int) * dst_len);
    if (weights) memcpy(weights, t.weight.data(), sizeof(double) * t.weight.size());
    return t.window;
}

// Diagnostic (tools/fused_timeline.py): the fused fp16 kernel on a device-resident UPSCALED plane with s_memtime stamps
// of workgroup 0 written to d_dbg (8 waves x 64 rows x 4 stamps: row start, layer 1 done, layers 2+3 done, row done).
int srcnn_fused_diag(const float* d_up, unsigned w, unsigned h, float* d_out, unsigned long long* d_dbg, void* stream)
{
    int rc = ensure_init(); if (rc) return rc;
    if ((rc = check_plane(d_up, w, h, d_out))) return rc;
    launch_fused_f16(d_up, (int)w, (int)h, 0, (int)h, d_out, 0, (int)h, g.fused_w, g.num_cus, (hipStream_t)stream, d_dbg);
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

// test hook: number of cached contribution tables / of ProcessSRCNN lanes created so far
int srcnn_debug_counts(int* tables, int* lanes)
{
    { std::lock_guard<std::mutex> lk(g.mu); if (tables) *tables = (int)g.tables.size(); }
    { std::lock_guard<std::mutex> lk(g.lane_mu); if (lanes) *lanes = (int)g.lanes.size(); }
    return SRCNN_OK;
}

}  // extern "C"
