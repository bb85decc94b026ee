// srcnn_capi.cpp -- the extern "C" boundary (include/srcnn_amd.h) over the gfx950 kernels: contexts, plumbing and the
// device-resident hot path.  The host-pointer pipelines (frame stream, ProcessSRCNN surface, node-level calls) are in
// srcnn_pipeline.cpp; the state both share is srcnn_host.hpp.
//
// Host-side orchestration only: argument validation with the reference's return codes
// (src/libsrcnn.cpp:951-966), lazily built + cached contribution tables
// (src/frawscale.cpp:8-112 -> resample_table.hpp), grow-only per-stream device workspaces, and the
// launch sequence that stands in for the body of libsrcnn::doSRCNN (src/libsrcnn.cpp:628-923).
// There is deliberately no CPU compute path in this file: if HIP is unusable the calls fail.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "resample_table.hpp"
#include <chrono>
#include <sys/auxv.h>
#include <unistd.h>

#include "srcnn_host.hpp"

namespace srcnn {

namespace {
thread_local char g_err[512] = "";
thread_local int t_ctx = 0;            // the calling thread's current context (srcnn_set_context)
}  // namespace

void set_last_error(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
const char* last_error() { return g_err; }

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        if (!settings().roctx) return;
        void* h = nullptr;
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr;
    }
};
const Roctx& roctx() { static const Roctx r; return r; }
}  // namespace

TraceRange::TraceRange(const char* fmt, ...) : on_(roctx().push != nullptr)
{
    if (!on_) return;
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    (void)roctx().push(buf);
}

TraceRange::~TraceRange() { if (on_) (void)roctx().pop(); }

// ---- settings: every SRCNN_* switch, read once (srcnn_settings.hpp) ----
Settings Settings::from_env()
{
    Settings st;
    // A switch that is set but not understood used to be dropped in silence (SRCNN_TRACE=yes read as 0; a retired switch of an
    // earlier round measured the production path and reported it as a variant): every such case is now said once, on stderr.
    auto warn = [](const char* fmt, const char* a, const char* b) { fprintf(stderr, fmt, a, b); };
    auto get = [](const char* name) -> const char* { const char* v = getenv(name); return (v && *v) ? v : nullptr; };
    auto parse_b = [&](const char* env, const char* v, bool& dst) {
        for (const char* t : {"1", "true", "yes", "on", "TRUE", "YES", "ON"}) if (!strcmp(v, t)) { dst = true; return; }
        for (const char* t : {"0", "false", "no", "off", "FALSE", "NO", "OFF"}) if (!strcmp(v, t)) { dst = false; return; }
        char* end = nullptr;
        const long n = strtol(v, &end, 10);
        if (end != v && *end == 0) { dst = n != 0; return; }
        warn("libsrcnn_amd: %s=%s is not a boolean (0/1, yes/no, on/off, true/false): the default stays\n", env, v);
    };
    auto parse_i = [&](const char* env, const char* v, long& dst) {
        char* end = nullptr;
        const long n = strtol(v, &end, 10);
        if (end != v && *end == 0) dst = n;
        else warn("libsrcnn_amd: %s=%s is not an integer: the default stays\n", env, v);
    };
    // SRCNN_RCCL_LIB names code to load into the process: like LD_PRELOAD it is not honoured under secure execution
    const bool secure = getauxval(AT_SECURE) != 0;
#define SRCNN_GET_B(m) parse_b(name, v, st.m);
#define SRCNN_GET_I(m) parse_i(name, v, st.m);
#define SRCNN_GET_S(m) if (secure && !strcmp(name, "SRCNN_RCCL_LIB")) warn("libsrcnn_amd: %s=%s ignored: secure-execution (set-uid / set-gid / capabilities) process\n", name, v); else st.m = v;
#define X(kind, member, env, def, values, effect) if (const char* v = get(env)) { const char* name = env; (void)name; SRCNN_GET_##kind(member) }
    SRCNN_SETTINGS(X)
#undef X
#undef SRCNN_GET_B
#undef SRCNN_GET_I
#undef SRCNN_GET_S
    // anything else that looks like one of ours: said once (names the tests, the bench and the build recipes use are not switches
    // of the library and pass)
    for (char** e = environ; e && *e; ++e) {
        if (strncmp(*e, "SRCNN_", 6) != 0) continue;
        const char* eq = strchr(*e, '=');
        const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
        bool known = false;
#define X(kind, member, env, def, values, effect) if (name == env) known = true;
        SRCNN_SETTINGS(X)
#undef X
        if (known || name == "SRCNN_TEST_SEED" || name == "SRCNN_AMD_LIB" || name == "SRCNN_FUSED_CFLAGS" || name.rfind("SRCNN_BENCH_", 0) == 0) continue;
        warn("libsrcnn_amd: %s is not a switch of this library (see srcnn_debug_settings / DESIGN.md 6)%s: ignored\n", name.c_str(), "");
    }
    st.max_workspace_mb = std::max(st.max_workspace_mb, 1L);
    st.max_lanes = std::min(64L, std::max(1L, st.max_lanes));
    st.comm_timeout_ms = std::max(0L, st.comm_timeout_ms);
    st.prefault_threads = std::max(1L, st.prefault_threads);
    st.rs_tpb = std::min(16L, std::max(0L, st.rs_tpb));
    st.async_chain = std::min(2L, std::max(0L, st.async_chain));
    return st;
}

std::string Settings::describe(bool markdown) const
{
    std::string out;
    char line[768];
    auto val_b = [](bool b) { return std::string(b ? "1" : "0"); };
    auto val_i = [](long i) { return std::to_string(i); };
    auto val_s = [](const std::string& t) { return t.empty() ? std::string("(unset)") : t; };
    const Settings defaults;
#define SRCNN_VAL_B(m) val_b(m)
#define SRCNN_VAL_I(m) val_i(m)
#define SRCNN_VAL_S(m) val_s(m)
#define X(kind, member, env, def, values, effect)                                                                                 \
    if (markdown) snprintf(line, sizeof line, "| `%s` | **%s**; %s | %s |\n", env, SRCNN_VAL_##kind(defaults.member).c_str(), values, effect); \
    else snprintf(line, sizeof line, "%s=%s (default %s)  -- %s\n", env, SRCNN_VAL_##kind(member).c_str(), SRCNN_VAL_##kind(defaults.member).c_str(), effect); \
    out += line;
    SRCNN_SETTINGS(X)
#undef X
#undef SRCNN_VAL_B
#undef SRCNN_VAL_I
#undef SRCNN_VAL_S
    return out;
}

const Settings& settings()
{
    static const Settings* st = new Settings(Settings::from_env());      // (never destroyed, like Global)
    return *st;
}

Global::Global()
{
    const Settings& st = settings();
    ws_budget.store((size_t)st.max_workspace_mb << 20);
    max_lanes = (size_t)st.max_lanes;
}

// Never destroyed: at process exit the HIP runtime may already be gone when static destructors run, and the
// tables' destructors call hipFree.  srcnn_shutdown() is the orderly way to release everything.
Global& G = *new Global;

namespace {

const uint32_t kWeightBits[kWeightCount] = {
#include "srcnn_weights.inc"
};

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
        for (int dy = 0; dy < 5; ++dy)
            for (int dx = 0; dx < 6; ++dx) d.w3p[m * 30 + dy * 6 + dx] = dx < 5 ? d.w3[m][dy * 5 + dx] : 0.f;
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

[[maybe_unused]] void build_fused_f16_weights(const DevWeights& d, FusedF16Weights& f)
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


// Host NUMA node next to a device: /sys/bus/pci/devices/<bdf>/numa_node (-1 when the platform does not say).
int device_numa_node(int device)
{
    char bdf[64] = "";
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* p = bdf; *p; ++p) *p = (char)tolower(*p);
    const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

// Create and initialise one context on `device` (caller holds G.mu).
int make_context_locked(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(SRCNN_E_NODEVICE, "no HIP device visible (%s); this library has no CPU path",
                    e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(SRCNN_E_ARG, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SRCNN_E_NODEVICE, "device %d is %s; this build only carries gfx950 code", device, prop.gcnArchName);
    auto cx = std::make_unique<Ctx>();
    cx->index = (int)G.ctxs.size();
    cx->device = device;
    // SRCNN_TRACE: where the time of creating a context goes (the first one also pays for loading the code object)
    const auto tr0 = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (settings().trace)
            fprintf(stderr, "srcnn_init ctx %d: %-28s at %7.2f ms\n", cx->index, what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count());
    };
    // __constant__ weights, the kernels' dynamic-LDS attributes and the fused tier's weight image are per DEVICE; a second
    // (virtual) context on the same device re-uploads identical bytes
    auto dw = std::make_unique<DevWeights>();
    build_dev_weights(*dw);
    stamp("weights built on the host");
    HIP_TRY(upload_weights(*dw));
    stamp("weights uploaded (module load)");
    HIP_TRY(conv12_mfma_prepare());
    stamp("layer-1+2 kernel attributes");
#ifndef SRCNN_STRICT_ONLY
    auto fw = std::make_unique<FusedF16Weights>();
    build_fused_f16_weights(*dw, *fw);
    HIP_TRY(hipMalloc((void**)&cx->fused_w, sizeof(FusedF16Weights)));
    struct FreeOnError {                       // the context is not published yet: a failure below must not leak its device block
        Ctx* cx;
        ~FreeOnError() { if (cx) { (void)hipFree(cx->fused_w); cx->fused_w = nullptr; } }
    } guard{cx.get()};
    stamp("fp16 tier weights built");
    if (int rc = copy_h2d_any(*cx, cx->fused_w, fw.get(), sizeof(FusedF16Weights), nullptr)) return rc;
    stamp("fp16 tier weights uploaded");
    HIP_TRY(fused_f16_prepare());
    stamp("fp16 tier kernel attributes");
    HIP_TRY(rs2d_prepare());
    stamp("resampler kernel attributes");
    guard.cx = nullptr;
#else
    HIP_TRY(rs2d_prepare());
    stamp("resampler kernel attributes");
#endif
    cx->num_cus = prop.multiProcessorCount;
    cx->numa_node = device_numa_node(device);
    // direct copies between the devices of a node (the node-level tiled frame moves its bands with hipMemcpyPeerAsync)
    for (auto& other : G.ctxs) {
        if (other->device == device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, other->device) == hipSuccess && can) {
            (void)hipSetDevice(device); (void)hipDeviceEnablePeerAccess(other->device, 0);
            (void)hipSetDevice(other->device); (void)hipDeviceEnablePeerAccess(device, 0);
            (void)hipGetLastError();               // "already enabled" is fine
            (void)hipSetDevice(device);
        }
    }
    G.ctxs.push_back(std::move(cx));
    G.nctx.store((int)G.ctxs.size());
    return SRCNN_OK;
}

// env SRCNN_DEVICES for a process that never calls srcnn_init*: "all", or a comma list of device ids (an id may repeat:
// virtual contexts on one device).  Unset: device 0.
int lazy_init_locked()
{
    if (!G.ctxs.empty()) return SRCNN_OK;
    const char* env = settings().devices.c_str();
    std::vector<int> devs;
    if (*env) {
        if (strcmp(env, "all") == 0) {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
            for (int d = 0; d < n; ++d) devs.push_back(d);
        } else {
            for (const char* p = env; *p;) {
                char* end = nullptr;
                const long v = strtol(p, &end, 10);
                if (end == p) return fail(SRCNN_E_ARG, "SRCNN_DEVICES=%s: expected \"all\" or a comma list of device ids", env);
                devs.push_back((int)v);
                p = (*end == ',') ? end + 1 : end;
            }
        }
    }
    if (devs.empty()) devs.push_back(0);
    for (int d : devs) {
        int rc = make_context_locked(d);
        if (rc) return rc;
    }
    return SRCNN_OK;
}

}  // namespace

int ensure_init()
{
    if (G.nctx.load(std::memory_order_acquire) > 0) return SRCNN_OK;
    std::lock_guard<std::mutex> lk(G.mu);
    return lazy_init_locked();
}

int context_count() { return G.nctx.load(std::memory_order_acquire); }

Ctx* context_at(int k)
{
    std::lock_guard<std::mutex> lk(G.mu);
    return (k >= 0 && k < (int)G.ctxs.size()) ? G.ctxs[k].get() : nullptr;
}

int bind(Ctx& cx)
{
    hipError_t e = hipSetDevice(cx.device);
    if (e != hipSuccess) return fail(SRCNN_E_HIP, "hipSetDevice(%d) -> %s", cx.device, hipGetErrorString(e));
    return SRCNN_OK;
}

Ctx* cur_ctx()
{
    if (ensure_init()) return nullptr;
    Ctx* cx = context_at(t_ctx);
    if (!cx) cx = context_at(0);
    if (!cx) { fail(SRCNN_E_NODEVICE, "no context"); return nullptr; }
    if (bind(*cx)) return nullptr;                 // a thread other than the one that called srcnn_init still needs the device selected
    return cx;
}

Ctx* ctx_for_stream(void* stream)
{
    if (ensure_init()) return nullptr;
    Ctx* cx = nullptr;
    if (stream) {
        std::lock_guard<std::mutex> lk(G.mu);
        auto it = G.stream_ctx.find((hipStream_t)stream);
        if (it != G.stream_ctx.end() && it->second < (int)G.ctxs.size()) cx = G.ctxs[it->second].get();
    }
    if (!cx) return cur_ctx();
    if (bind(*cx)) return nullptr;
    return cx;
}

// Look up / build / upload the contribution table of one axis.  The cache holds one reference, the caller gets
// another (and parks it in c.hold), and a table is only ever freed when nobody but the cache references it: a
// lookup can therefore never invalidate a table handed out earlier -- not the first of the two tables of a
// resample, not one another thread is about to launch with, not one baked into a captured graph.
int get_table(Call& c, int filter, unsigned dst_len, unsigned src_len, TableRef& out)
{
    Ctx& cx = *c.cx;
    std::lock_guard<std::mutex> lk(cx.mu);
    const auto key = std::make_tuple(filter, dst_len, src_len);
    auto it = cx.tables.find(key);
    if (it == cx.tables.end()) {
        const AxisTable t = build_axis_table(filter, dst_len, src_len);
        auto d = std::make_shared<DeviceTable>();
        d->stride = t.stride;
        d->max_taps = t.max_taps;
        d->h_first.assign(t.first.begin(), t.first.end());
        d->h_taps.assign(t.taps.begin(), t.taps.end());
        d->monotone = true;
        for (unsigned u = 1; u < dst_len; ++u)
            if (t.first[u] < t.first[u - 1] || t.first[u] + t.taps[u] < t.first[u - 1] + t.taps[u - 1]) { d->monotone = false; break; }
        HIP_TRY(hipMalloc((void**)&d->first, sizeof(int) * dst_len));
        HIP_TRY(hipMalloc((void**)&d->taps, sizeof(int) * dst_len));
        HIP_TRY(hipMalloc((void**)&d->weight, sizeof(double) * t.weight.size()));
        // (heap vectors: through the bounce slots like every other pageable source -- a tall frame's weight table is 400 KB)
        if (int rc = copy_h2d_any(cx, d->first, t.first.data(), sizeof(int) * dst_len, nullptr)) return rc;
        if (int rc = copy_h2d_any(cx, d->taps, t.taps.data(), sizeof(int) * dst_len, nullptr)) return rc;
        if (int rc = copy_h2d_any(cx, d->weight, t.weight.data(), sizeof(double) * t.weight.size(), nullptr)) return rc;
        if (cx.tables.size() >= kMaxTables) {
            // evict the least recently used tables that only the cache still references, down to half the bound.
            // Kernels launched by calls that already returned may still be reading them, hence the drain first.
            std::vector<std::pair<unsigned long long, std::tuple<int, unsigned, unsigned>>> idle;
            for (auto& kv : cx.tables)
                if (kv.second.use_count() == 1) idle.emplace_back(kv.second->stamp, kv.first);
            std::sort(idle.begin(), idle.end());
            if (!idle.empty()) (void)hipDeviceSynchronize();
            for (auto& e : idle) {
                if (cx.tables.size() < kMaxTables / 2) break;
                cx.tables.erase(e.second);
            }
        }
        it = cx.tables.emplace(key, std::move(d)).first;
    }
    it->second->stamp = ++cx.table_clock;
    out = it->second;
    if (c.hold) c.hold->push_back(out);
    return SRCNN_OK;
}

Workspace* workspace_for(Ctx& cx, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(cx.mu);
    auto& p = cx.ws[s];
    if (!p) p = std::make_unique<Workspace>();
    return p.get();
}

// Page-locked host memory for a context's staging: portable (every device of the node may DMA to it -- the node-level
// ProcessSRCNN shares one output staging), and placed on the NUMA node next to the device when the platform tells us
// which one that is (hipHostMallocNumaUser: the allocation follows the calling thread's memory policy, which is set to
// "prefer that node" around the call).  SRCNN_NUMA=0 disables the placement.
// The blocks handed out by srcnn_host_alloc_pinned: how srcnn_process_u8 recognises caller buffers it can copy from / into
// directly (asking the runtime about a pageable pointer works too, but logs an error line per question under AMD_LOG_LEVEL >= 1).
std::mutex g_pinned_mu;
std::map<uintptr_t, size_t>& pinned_blocks()
{
    static auto* m = new std::map<uintptr_t, size_t>;
    return *m;
}
bool pinned_by_library(const void* p, size_t n)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    auto& m = pinned_blocks();
    auto it = m.upper_bound(a);
    if (it == m.begin()) return false;
    --it;
    return a >= it->first && a + n <= it->first + it->second;
}

void* pinned_alloc(Ctx& cx, size_t bytes)
{
    void* p = nullptr;
    const bool want_place = settings().numa && cx.numa_node >= 0 && cx.numa_node < 1024;
    constexpr unsigned long kMaxNode = 1024;
    unsigned long mask[kMaxNode / (8 * sizeof(unsigned long))] = {0};
    // The policy is the CALLING thread's (an application thread inside ProcessSRCNN): whatever it was -- numactl --membind /
    // --interleave, or one the application set itself -- is read first and put back exactly; if it cannot be read, the
    // allocation is simply not placed.
    int saved_mode = 0;
    unsigned long saved_mask[kMaxNode / (8 * sizeof(unsigned long))] = {0};
    bool placed = false;
    if (want_place && syscall(SYS_get_mempolicy, &saved_mode, saved_mask, kMaxNode, nullptr, 0ul) == 0) {
        mask[cx.numa_node / (8 * sizeof(unsigned long))] |= 1ul << (cx.numa_node % (8 * sizeof(unsigned long)));
        placed = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, kMaxNode) == 0;
    }
    unsigned flags = hipHostMallocPortable | (placed ? hipHostMallocNumaUser : 0u);
    if (hipHostMalloc(&p, bytes ? bytes : 1, flags) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
        if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    }
    if (placed) {
        bool any = false;
        for (unsigned long m : saved_mask) any = any || m != 0;
        // MPOL_DEFAULT (and MPOL_LOCAL) take an empty node set; the others get the set they had
        if (syscall(SYS_set_mempolicy, saved_mode, any ? saved_mask : nullptr, any ? kMaxNode : 0ul) != 0)
            (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
    }
    if (!p) fail(SRCNN_E_DEVMEM, "hipHostMalloc(%zu) failed", bytes);
    return p;
}

bool host_is_page_locked(const void* p)
{
    hipPointerAttribute_t a;
    const bool yes = hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return yes;
}

void HostBounce::release()
{
    if (pin) (void)hipHostFree(pin);
    for (hipEvent_t& e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    pin = nullptr; slot = 0;
}

namespace {
int bounce_ready(Ctx& cx, HostBounce& b, size_t bytes)          // b.mu held; no copy of this context is in flight
{
    size_t want = 1u << 20;
    while (want < bytes && want < HostBounce::kSlot) want <<= 1;
    if (want > b.slot) {
        if (b.pin) (void)hipHostFree(b.pin);
        b.slot = 0;
        if (!(b.pin = static_cast<unsigned char*>(pinned_alloc(cx, 2 * want)))) return SRCNN_E_DEVMEM;
        b.slot = want;
    }
    for (hipEvent_t& e : b.ev)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return SRCNN_OK;
}
}  // namespace

// The bounced copies run on the stream the caller named (NULL: the default stream) -- NOT on a stream of their own: the
// runtime maps streams onto a handful of hardware queues, and one more stream per context made the frame stream's copy and
// kernel streams share a queue (host-stream rate 3.48 -> 3.05 GPix/s until it was taken out again, round 6).
int copy_h2d_any(Ctx& cx, void* d_dst, const void* h_src, size_t bytes, hipStream_t after)
{
    if (bytes == 0) return SRCNN_OK;
    if (pinned_by_library(h_src, bytes) || host_is_page_locked(h_src)) {
        if (after) HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, after));
        else HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
        return SRCNN_OK;
    }
    HostBounce& b = cx.bounce;
    std::lock_guard<std::mutex> lk(b.mu);
    if (int rc = bounce_ready(cx, b, bytes)) return rc;
    const size_t SLOT = b.slot;
    const unsigned char* src = static_cast<const unsigned char*>(h_src);
    unsigned char* dst = static_cast<unsigned char*>(d_dst);
    bool used[2] = {false, false};
    for (size_t off = 0, i = 0; off < bytes; off += SLOT, ++i) {
        const int k = (int)(i & 1);
        const size_t len = std::min(SLOT, bytes - off);
        if (used[k] && wait_event(b.ev[k]) != hipSuccess) return fail(SRCNN_E_HIP, "bounced H2D copy failed");
        parallel_memcpy(b.pin + k * SLOT, src + off, len);
        HIP_TRY(hipMemcpyAsync(dst + off, b.pin + k * SLOT, len, hipMemcpyHostToDevice, after));
        HIP_TRY(hipEventRecord(b.ev[k], after));
        used[k] = true;
    }
    for (int k = 0; k < 2; ++k)                      // the slots are free again (and the data is on the device) on return
        if (used[k] && wait_event(b.ev[k]) != hipSuccess) return fail(SRCNN_E_HIP, "bounced H2D copy failed");
    return SRCNN_OK;
}

int copy_d2h_any(Ctx& cx, void* h_dst, const void* d_src, size_t bytes, hipStream_t after)
{
    if (bytes == 0) return SRCNN_OK;
    if (pinned_by_library(h_dst, bytes) || host_is_page_locked(h_dst)) {
        if (after) HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, after));
        else HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
        return SRCNN_OK;
    }
    HostBounce& b = cx.bounce;
    std::lock_guard<std::mutex> lk(b.mu);
    if (int rc = bounce_ready(cx, b, bytes)) return rc;
    const size_t SLOT = b.slot;
    unsigned char* dst = static_cast<unsigned char*>(h_dst);
    const unsigned char* src = static_cast<const unsigned char*>(d_src);
    // chunk i+1 is on the copy engine while chunk i is copied out of its slot
    const size_t nchunks = (bytes + SLOT - 1) / SLOT;
    auto queue = [&](size_t i) -> int {
        const int k = (int)(i & 1);
        const size_t off = i * SLOT, len = std::min(SLOT, bytes - off);
        HIP_TRY(hipMemcpyAsync(b.pin + k * SLOT, src + off, len, hipMemcpyDeviceToHost, after));
        HIP_TRY(hipEventRecord(b.ev[k], after));
        return SRCNN_OK;
    };
    if (int rc = queue(0)) return rc;
    for (size_t i = 0; i < nchunks; ++i) {
        const int k = (int)(i & 1);
        const size_t off = i * SLOT, len = std::min(SLOT, bytes - off);
        if (wait_event(b.ev[k]) != hipSuccess) return fail(SRCNN_E_HIP, "bounced D2H copy failed");
        if (i + 1 < nchunks) { if (int rc = queue(i + 1)) return rc; }
        parallel_memcpy(dst + off, b.pin + k * SLOT, len);
    }
    return SRCNN_OK;
}

int grow_pinned(Ctx& cx, unsigned char*& p, size_t& have, size_t want)
{
    if (want <= have) return SRCNN_OK;
    if (p) { (void)hipDeviceSynchronize(); (void)hipHostFree(p); p = nullptr; have = 0; }
    p = static_cast<unsigned char*>(pinned_alloc(cx, want));
    if (!p) return SRCNN_E_DEVMEM;
    have = want;
    return SRCNN_OK;
}

namespace {
// The naps are 20 us (50 us after the first 2 ms).  The kernel rounds a sleep up by the thread's timer slack, 50 us by default,
// which would make every nap 70+ us; the slack is lowered for the duration of the wait only and put back afterwards.
template <class Q>
hipError_t poll_until_ready(Q&& query)
{
    hipError_t r;
    for (int i = 0; i < 64; ++i)
        if ((r = query()) != hipErrorNotReady) return r;
    const int slack = prctl(PR_GET_TIMERSLACK);
    if (slack > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000UL);
    for (int n = 0;; ++n) {
        std::this_thread::sleep_for(std::chrono::microseconds(n < 100 ? 20 : 50));
        if ((r = query()) != hipErrorNotReady) break;
    }
    if (slack > 1000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)slack);
    return r;
}
bool spin_waits() { return settings().spin_wait; }
}  // namespace

hipError_t wait_event(hipEvent_t e, std::mutex* query_guard)
{
    if (spin_waits()) return hipEventSynchronize(e);
    const hipError_t r = poll_until_ready([&] {
        if (!query_guard) return hipEventQuery(e);
        std::lock_guard<std::mutex> lk(*query_guard);
        return hipEventQuery(e);
    });
    (void)hipGetLastError();               // the "not ready" answers are not errors
    return r;
}

hipError_t wait_stream(hipStream_t s)
{
    if (spin_waits()) return hipStreamSynchronize(s);
    const hipError_t r = poll_until_ready([&] { return hipStreamQuery(s); });
    (void)hipGetLastError();
    return r;
}

void parallel_memcpy(void* dst, const void* src, size_t n)
{
    const size_t kChunk = 4u << 20;
    unsigned nt = (unsigned)std::min<size_t>(8, n / kChunk);
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = ((n / nt) + 4095) & ~size_t(4095);
    size_t done = 0;                    // bytes handed to threads that actually started
    try {
        for (unsigned t = 1; t < nt; ++t) {
            const size_t off = (size_t)t * per;
            if (off >= n) break;
            const size_t len = std::min(per, n - off);
            th.emplace_back([=] { memcpy((char*)dst + off, (const char*)src + off, len); });
            done = off + len;
        }
    } catch (...) {                     // thread creation failed: the rest is copied here
        done = th.empty() ? 0 : done;
    }
    // this thread takes the first chunk (and whatever could not be handed out)
    memcpy(dst, src, std::min(per, n));
    const size_t covered = th.empty() ? std::min(per, n) : done;
    if (covered < n) memcpy((char*)dst + covered, (const char*)src + covered, n - covered);
    for (auto& t : th) t.join();
}

void ProcLane::release_buffers()
{
    ws.release();
    if (pin_in) (void)hipHostFree(pin_in);
    if (pin_out) (void)hipHostFree(pin_out);
    pin_in = pin_out = nullptr; pin_in_n = pin_out_n = 0;
}

void ProcLane::release()
{
    release_buffers();
    for (auto e : band_events) (void)hipEventDestroy(e);
    band_events.clear();
    if (st) (void)hipStreamDestroy(st);
    if (copy_st) (void)hipStreamDestroy(copy_st);
    if (in_st) (void)hipStreamDestroy(in_st);
    st = copy_st = in_st = nullptr;
}

void NodeLane::release()
{
    ws.release();
    (void)hipFree(in); (void)hipFree(band);
    in = band = nullptr; in_n = band_n = 0;
    for (auto e : events) (void)hipEventDestroy(e);
    events.clear();
    if (st) (void)hipStreamDestroy(st);
    if (copy_st) (void)hipStreamDestroy(copy_st);
    st = copy_st = nullptr;
}

LaneLease::LaneLease(Ctx& c) : cx(&c)
{
    std::unique_lock<std::mutex> lk(c.lane_mu);
    for (;;) {
        for (auto& l : c.lanes)
            if (!l->busy) { lane = l.get(); break; }
        if (lane) break;
        if (c.lanes.size() < G.max_lanes) {
            auto l = std::make_unique<ProcLane>();
            if (hipStreamCreateWithFlags(&l->st, hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&l->copy_st, hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&l->in_st, hipStreamNonBlocking) != hipSuccess) {
                l->release();
                rc = fail(SRCNN_E_HIP, "could not create the streams of a ProcessSRCNN lane");
                return;
            }
            c.lanes.push_back(std::move(l));
            lane = c.lanes.back().get();
            break;
        }
        c.lane_cv.wait(lk);
    }
    lane->busy = true;
}

LaneLease::~LaneLease()
{
    if (!lane) return;
    // nothing of this call may still be running on the lane when the next caller takes it
    (void)hipSetDevice(cx->device);
    (void)hipStreamSynchronize(lane->st);
    (void)hipStreamSynchronize(lane->copy_st);
    (void)hipStreamSynchronize(lane->in_st);
    { std::lock_guard<std::mutex> lk(cx->lane_mu); lane->busy = false; }
    cx->lane_cv.notify_one();
}

namespace {

// RAII bracket: records an event pair around one stage on the launch stream when profiling is on.
struct StageTimer {
    Ctx& cx; hipStream_t s; int stage; hipEvent_t a = nullptr, b = nullptr; bool on;
    hipEvent_t take()
    {
        if (!cx.event_pool.empty()) { hipEvent_t e = cx.event_pool.back(); cx.event_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    StageTimer(int stage_, const Call& c) : cx(*c.cx), s(c.s), stage(stage_), on(c.timing && G.profiling.load(std::memory_order_relaxed))
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(cx.mu);
        a = take(); b = take();
        if (!a || !b) { on = false; return; }
        (void)hipEventRecord(a, s);
    }
    ~StageTimer()
    {
        if (!on) return;
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> lk(cx.mu);
        cx.spans.push_back(StageSpan{a, b, stage});
    }
};

void drain_spans_locked(Ctx& cx)
{
    for (auto& sp : cx.spans) {
        float ms = 0.f;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            cx.stage_ms[sp.stage] += ms;
            cx.stage_n[sp.stage] += 1;
        }
        cx.event_pool.push_back(sp.a);
        cx.event_pool.push_back(sp.b);
    }
    cx.spans.clear();
}

// Y holds rows [y_row_base, y_row_base + y_rows) of the (W x H) upscaled plane; the kernels clamp their halo
// reads to that range as well as to the image (tile rows past the end of a band are computed but never stored).
void run_conv12(const Call& c, const float* Y, int W, int H, int y_row_base, int y_rows, float* C2, size_t plane, int row0,
                int rows)
{
    unsigned long long* clk = nullptr;
    if (G.clock_probe.load(std::memory_order_relaxed) && c.cx->clock_buf)
        clk = c.cx->clock_buf + 2 * (size_t)(c.cx->clock_n.fetch_add(1) % kClockSlots);
    // the tile queue lives with the workspace: one per stream, so launches that share it are ordered
    unsigned* queue = nullptr;
    // A capture that is not one of the library's own (those freeze a PRIVATE workspace first): the caller may replay the graph
    // on any stream, beside eager calls that share this workspace's counters, and the probe slot would be baked in as well --
    // such launches deal their tiles with the static stride and stamp nothing.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool foreign_capture = c.s && c.ws && !c.ws->frozen && hipStreamIsCapturing(c.s, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
    if (foreign_capture) clk = nullptr;
    if (c.ws && settings().conv12_queue && !foreign_capture) {
        if (c.ws->queue && c.ws->queue_dirty) {
            // a call on this workspace failed after launching: a launch that never finished leaves its draws behind
            if (hipMemsetAsync(c.ws->queue, 0, 2 * sizeof(unsigned), c.s) == hipSuccess) c.ws->queue_dirty = false;
            else (void)hipGetLastError();
        }
        if (!c.ws->queue && !c.ws->frozen) {
            void* q = nullptr;
            // zeroed ON THE LAUNCH STREAM: lane streams are non-blocking, so a null-stream hipMemset is not ordered before
            // the first kernel that draws from the queue (seen as wrong tiles in the first call of a fresh lane)
            if (hipMalloc(&q, 2 * sizeof(unsigned)) == hipSuccess && hipMemsetAsync(q, 0, 2 * sizeof(unsigned), c.s) == hipSuccess) c.ws->queue = static_cast<unsigned*>(q);
            else { (void)hipFree(q); (void)hipGetLastError(); }
        }
        queue = c.ws->queue;
    }
    launch_conv12_mfma(Y, W, H, y_row_base, y_rows, C2, plane, row0, rows, c.relax(), c.cx->num_cus, c.s, clk, queue);
}

DevAxisTable view_of(const TableRef& t) { return t->view(); }

// FRAWResizeEngine::scale (src/frawscale.cpp:162-286) for destination rows [r0,r1) only, from either kind of source.
// dst holds rows [r0,r1) (row r0 at offset 0).  tmp is scratch from the call's workspace.
int resample_src_rows(Call& c, const YSource& src, unsigned sw, unsigned sh, unsigned dw, unsigned dh, int filter,
                      unsigned r0, unsigned r1, float* d_dst)
{
    Workspace& ws = *c.ws;
    hipStream_t s = c.s;
    const float* d_in = src.plane;
    if (sw == dw && sh == dh) {
        // The reference's identity branch copies sizeof(unsigned short) bytes per pixel into an
        // uninitialised buffer (src/frawscale.cpp:185-193), i.e. half the plane is garbage.  We copy the
        // whole plane (the evident intent); documented in DESIGN.md as the one deliberate deviation and pinned by
        // tests/test_gpu_parity.py::test_identity_size_deviation_is_pinned.
        if (!d_in) return fail(SRCNN_E_UNSUPPORTED, "identity-size resample needs a float plane");
        HIP_TRY(hipMemcpyAsync(d_dst, d_in + (size_t)r0 * sw, sizeof(float) * (size_t)(r1 - r0) * sw,
                               hipMemcpyDeviceToDevice, s));
        return SRCNN_OK;
    }
    TableRef tv, th;
    int rc;
    if (dw > sw && sh != dh) {
        // up-scale in both axes: vertical first, then horizontal (src/frawscale.cpp:238-278), both in one kernel
        if ((rc = get_table(c, filter, dw, sw, th))) return rc;
        if ((rc = get_table(c, filter, dh, sh, tv))) return rc;
        if (!settings().resample_2pass &&
            launch_rs2d(src, sw, sh, d_dst, dw, dh, r0, r1 - r0, view_of(tv), view_of(th), s)) return SRCNN_OK;
    }
    if (!d_in) return fail(SRCNN_E_UNSUPPORTED, "this resample shape needs a float source plane");
    if (dw <= sw) {
        // horizontal first over all source rows, then vertical (src/frawscale.cpp:195-237)
        const float* mid = d_in;
        int mid_row_base = 0;
        if (sw != dw) {
            if ((rc = get_table(c, filter, dw, sw, th))) return rc;
            if (sh != dh) {
                // only the source rows the vertical taps of rows [r0, r1) read (a banded call used to redo the pass over ALL
                // source rows for every band -- and over rows of a split plane that had not been staged yet)
                if ((rc = get_table(c, filter, dh, sh, tv))) return rc;
                unsigned lo = 0, hi = sh;
                tv->source_span(r0, r1, lo, hi);
                hi = std::min(hi, sh);
                if (hi <= lo) return fail(SRCNN_E_UNSUPPORTED, "empty source span for rows [%u,%u)", r0, r1);
                if ((rc = grow_ws(ws, ws.tmp, ws.tmp_n, (size_t)dw * (hi - lo)))) return rc;
                launch_resample_rows(d_in + (size_t)lo * sw, sw, ws.tmp, dw, hi - lo, view_of(th), s);
                mid = ws.tmp;
                mid_row_base = (int)lo;
            } else {
                launch_resample_rows(d_in + (size_t)r0 * sw, sw, d_dst, dw, r1 - r0, view_of(th), s);
                return SRCNN_OK;
            }
        }
        if (!tv && (rc = get_table(c, filter, dh, sh, tv))) return rc;
        launch_resample_cols(mid, dw, mid_row_base, d_dst, r0, r1 - r0, view_of(tv), s);
    } else {
        // vertical first, then horizontal (src/frawscale.cpp:238-278)
        if (!th && (rc = get_table(c, filter, dw, sw, th))) return rc;
        const float* mid = d_in + (size_t)r0 * sw;
        if (sh != dh) {
            if (!tv && (rc = get_table(c, filter, dh, sh, tv))) return rc;
            if ((rc = grow_ws(ws, ws.tmp, ws.tmp_n, (size_t)sw * (r1 - r0)))) return rc;
            launch_resample_cols(d_in, sw, 0, ws.tmp, r0, r1 - r0, view_of(tv), s);
            mid = ws.tmp;
        }
        launch_resample_rows(mid, sw, d_dst, dw, r1 - r0, view_of(th), s);
    }
    return SRCNN_OK;
}

}  // namespace

int resample_rows_range(Call& c, const float* d_in, unsigned sw, unsigned sh, unsigned dw, unsigned dh, int filter,
                        unsigned r0, unsigned r1, float* d_dst)
{
    return resample_src_rows(c, YSource::from_plane(d_in), sw, sh, dw, dh, filter, r0, r1, d_dst);
}

int check_plane(const void* in, unsigned w, unsigned h, const void* out)
{
    if (!in || !out || w == 0 || h == 0) return fail(SRCNN_E_ARG, "NULL pointer or zero dimension");
    if ((unsigned long long)w * h > 0x7fffffffULL) return fail(SRCNN_E_UNSUPPORTED, "plane too large");
    return SRCNN_OK;
}

// resample + conv12 + conv3 for output rows [r0,r1) of the (dw x dh) result.
int y_path_rows(Call& c, const YSource& src, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                unsigned r0, unsigned r1, float* d_out)
{
    if (r1 > dh || r0 >= r1) return fail(SRCNN_E_ARG, "row range [%u,%u) outside 0..%u", r0, r1, dh);
    if (dh > (1u << 20) || h > (1u << 20) || dw > 0x7fffffu || (r1 - r0) > 65535u * 16u)
        return fail(SRCNN_E_UNSUPPORTED, "output %ux%u too large", dw, dh);
    Workspace& ws = *c.ws;
    [[maybe_unused]] Ctx& cx = *c.cx;
    // rows of layer-2 activations that conv3 touches (clamp-to-edge of the ACTIVATIONS at the true
    // border), and rows of upscaled Y that conv1 touches for those.
    const unsigned ca = r0 >= 2 ? r0 - 2 : 0, cb = std::min(dh, r1 + 2);
    const unsigned ua = ca >= 4 ? ca - 4 : 0, ub = std::min(dh, cb + 4);
    int rc;
    if ((rc = grow_ws(ws, ws.up, ws.up_n, (size_t)dw * (ub - ua)))) return rc;
#ifndef SRCNN_STRICT_ONLY
    const bool fused = c.mode == SRCNN_MODE_FAST_F16;
    if (fused) {
        // non-parity tier: one kernel for all three layers, no layer-2 planes at all
        {
            StageTimer t(SRCNN_STAGE_RESAMPLE, c);
            if ((rc = resample_src_rows(c, src, w, h, dw, dh, filter, ua, ub, ws.up))) return rc;
        }
        {
            StageTimer t(SRCNN_STAGE_CONV12, c);
            launch_fused_f16(ws.up, (int)dw, (int)dh, (int)ua, (int)(ub - ua), d_out, (int)r0, (int)(r1 - r0), cx.fused_w,
                             cx.num_cus, c.s);
        }
        HIP_TRY(hipGetLastError());
        return SRCNN_OK;
    }
#endif
    if ((rc = grow_ws(ws, ws.c2, ws.c2_n, (size_t)C2N * dw * (cb - ca)))) return rc;
    TraceRange tr("srcnn y_path rows [%u,%u) of %ux%u", r0, r1, dw, dh);
    {
        StageTimer t(SRCNN_STAGE_RESAMPLE, c);
        if ((rc = resample_src_rows(c, src, w, h, dw, dh, filter, ua, ub, ws.up))) return rc;
    }
    const size_t plane = (size_t)dw * (cb - ca);
    {
        StageTimer t(SRCNN_STAGE_CONV12, c);
        run_conv12(c, ws.up, (int)dw, (int)dh, (int)ua, (int)(ub - ua), ws.c2, plane, (int)ca, (int)(cb - ca));
    }
    {
        StageTimer t(SRCNN_STAGE_CONV3, c);
        launch_conv3(ws.c2, plane, (int)dw, (int)dh, (int)ca, (int)(cb - ca), d_out, (int)r0, (int)(r1 - r0),
                     c.relax(), c.s);
    }
    if (const hipError_t e = hipGetLastError(); e != hipSuccess) {
        ws.queue_dirty = true;                         // re-zeroed on the launch stream before the next launch trusts it
        return fail(SRCNN_E_HIP, "y_path_rows: %s", hipGetErrorString(e));
    }
    return SRCNN_OK;
}

// Source rows [lo,hi) of the (w x h) input that output rows [r0,r1) of the Y path depend on: +-2 rows of layer-2
// activations, +-4 rows of upscaled Y for those, and the vertical taps of the resampler for that range -- read off the
// contribution table itself, so it is exact for every filter and ratio.
int y_path_source_rows(Call& c, unsigned h, unsigned dh, int filter, unsigned r0, unsigned r1, unsigned& lo, unsigned& hi)
{
    const unsigned ca = r0 >= 2 ? r0 - 2 : 0, cb = std::min(dh, r1 + 2);
    const unsigned ua = ca >= 4 ? ca - 4 : 0, ub = std::min(dh, cb + 4);
    if (h == dh) { lo = ua; hi = ub; return SRCNN_OK; }
    TableRef tv;
    int rc = get_table(c, filter, dh, h, tv);
    if (rc) return rc;
    tv->source_span(ua, ub, lo, hi);
    hi = std::min(hi, h);
    return SRCNN_OK;
}

unsigned budget_band_rows(unsigned dw)
{
    const size_t budget = G.ws_budget.load();
    const size_t row_bytes = (size_t)C2N * dw * sizeof(float);
    const size_t fit = budget / row_bytes;
    // 16-row floor: one tile row of the layer kernels (a smaller limit is exceeded rather than refused; srcnn_amd.h says so)
    return (unsigned)std::min<size_t>(std::max<size_t>(16, fit > 4 ? fit - 4 : 1), 1u << 20);
}

// Cut output rows [R0,R1) of a (dw x dh) frame into pieces of about frac[0], frac[1], ... of the range (the last piece takes the
// rest).  The layer-1+2 kernel is persistent: `grid` resident workgroups walk a piece's 64 x tile_rows tiles with a static
// stride, so a piece whose tile count is not a multiple of the grid wastes part of its last round (a plain percentage split
// of an 8K frame into 5 bands cost 68 rounds instead of 64, +6 % of the dominant kernel; four equal quarters of one rank's
// band of a 16K frame 36 instead of 32).  Each cut is therefore moved, within +-30 % of its target height, to where the
// layer-2 rows the piece computes (its rows + 2 halo rows per interior side) fill their rounds best.  Pure function of its
// arguments: every rank of a tiled frame derives the same table.
std::vector<unsigned> plan_cuts(unsigned R0, unsigned R1, unsigned dw, unsigned dh, const double* frac, int nfrac, int grid, int tile_rows)
{
    const unsigned rows = R1 - R0;
    std::vector<unsigned> cuts{R0};
    const unsigned tiles_x = (dw + 63) / 64;
    unsigned a = R0;
    for (int i = 0; i < nfrac; ++i) {
        const unsigned want = std::max(32u, (unsigned)(rows * frac[i]));
        unsigned b = a + (want & ~15u);
        if (grid > 0 && tile_rows > 0) {
            // layer-2 rows of piece [a, b) = [max(a-2,0), min(b+2,dh)): choose their tile-row count near the target
            const unsigned top = a >= 2 ? 2 : a;
            const unsigned t_want = (want + top + 2 + tile_rows - 1) / tile_rows;
            const unsigned t_lo = std::max(1u, (unsigned)(t_want * 0.7)), t_hi = std::max(t_lo, (unsigned)(t_want * 1.3));
            double best = -1.0;
            unsigned best_t = t_want;
            for (unsigned t = t_lo; t <= t_hi; ++t) {
                const unsigned long long tiles = (unsigned long long)t * tiles_x;
                const unsigned long long rounds = (tiles + grid - 1) / grid;
                const double fill = (double)tiles / (double)(rounds * grid);
                const double score = fill - 1e-4 * (t > t_want ? t - t_want : t_want - t);      // ties: closest to the target
                if (score > best) { best = score; best_t = t; }
            }
            if (best_t * tile_rows > top + 2) b = a + best_t * tile_rows - top - 2;
        }
        if (b <= a || b >= R1 || R1 - b < 32) break;
        cuts.push_back(b);
        a = b;
    }
    cuts.push_back(R1);
    return cuts;
}

// Output rows [r0,r1).  The 32 layer-2 planes are the big scratch (128 B per output pixel).  A range whose planes
// would exceed the workspace budget (default 4.5 GiB, SRCNN_MAX_WORKSPACE_MB) is produced in horizontal bands --
// bit-identical to the whole range -- so a 16K x 16K output needs the same scratch as an 8K one.
int y_path_range(Call& c, const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                 unsigned r0, unsigned r1, float* d_out)
{
    const size_t budget = G.ws_budget.load();
    if (r1 > dh || r0 >= r1) return fail(SRCNN_E_ARG, "row range [%u,%u) outside 0..%u", r0, r1, dh);
    const size_t row_bytes = (size_t)C2N * dw * sizeof(float);
    const bool no_planes = c.mode == SRCNN_MODE_FAST_F16;      // the fused kernel has no layer-2 planes
    const YSource src = YSource::from_plane(d_in);
    if (no_planes || row_bytes * ((size_t)(r1 - r0) + 4) <= budget) return y_path_rows(c, src, w, h, dw, dh, filter, r0, r1, d_out);
    const unsigned band = budget_band_rows(dw);
    for (unsigned a = r0; a < r1; a += band) {
        const unsigned b = std::min(r1, a + band);
        int rc = y_path_rows(c, src, w, h, dw, dh, filter, a, b, d_out + (size_t)(a - r0) * dw);
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

namespace {

// An eager call on a caller-visible stream: the stream's own scratch, locked while this call enqueues.
struct StreamCall {
    std::vector<TableRef> tables;
    Call c;
    std::unique_lock<std::mutex> lk;
    int rc = SRCNN_OK;
    explicit StreamCall(void* stream)
    {
        c.cx = ctx_for_stream(stream);
        if (!c.cx) { rc = SRCNN_E_NODEVICE; return; }
        c.s = (hipStream_t)stream;
        c.ws = workspace_for(*c.cx, c.s);
        c.mode = G.mode.load();
        c.hold = &tables;
        lk = std::unique_lock<std::mutex>(c.ws->mu);
    }
};

int batch_frames(Call& c, const float* d_in, unsigned w, unsigned h, unsigned nframes, float* d_out)
{
    const size_t in_n = (size_t)w * h, out_n = in_n * 4;
    for (unsigned f = 0; f < nframes; ++f) {
        int rc = y_path_frame(c, d_in + f * in_n, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, d_out + f * out_n);
        if (rc) return rc;
    }
    return SRCNN_OK;
}

// A captured batch owns everything its kernel nodes point at: its scratch and its contribution tables live exactly
// as long as the handle, whatever happens to the stream's own workspace or to the table cache in the meantime.
struct BatchGraph {
    Ctx* cx = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t stream = nullptr;
    Workspace ws;
    std::vector<TableRef> tables;
};

void release_context(Ctx& cx)
{
    (void)hipSetDevice(cx.device);
    (void)hipDeviceSynchronize();
    {
        std::lock_guard<std::mutex> lk(cx.lane_mu);
        for (auto& l : cx.lanes) l->release();
        cx.lanes.clear();
    }
    cx.node.release();
    std::lock_guard<std::mutex> lk(cx.mu);
    cx.tables.clear();                        // graphs still alive keep their own references
    for (auto& kv : cx.ws) kv.second->release();
    cx.ws.clear();
    for (auto& sl : cx.slots) {
        if (sl.exec) (void)hipGraphExecDestroy(sl.exec);
        if (sl.st) (void)hipStreamDestroy(sl.st);
        if (sl.cst) (void)hipStreamDestroy(sl.cst);
        for (hipEvent_t* e : {&sl.e_in, &sl.e_k, &sl.e_out}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
        (void)hipFree(sl.din); (void)hipFree(sl.dout);
        sl.ws.release();
        sl.tables.clear();
        sl.graph_tables.clear();
        sl.st = sl.cst = nullptr; sl.din = sl.dout = nullptr; sl.din_n = sl.dout_n = 0;
        sl.exec = nullptr; sl.gw = sl.gh = 0; sl.gmode = -1; sl.uses = 0;
    }
    drain_spans_locked(cx);
    for (auto e : cx.event_pool) (void)hipEventDestroy(e);
    cx.event_pool.clear();
    (void)hipFree(cx.fused_w);
    cx.fused_w = nullptr;
    (void)hipFree(cx.clock_buf);
    cx.clock_buf = nullptr;
    { std::lock_guard<std::mutex> bl(cx.bounce.mu); cx.bounce.release(); }
    { std::lock_guard<std::mutex> hl(cx.host_call.mu); cx.host_call.release(); }
}

}  // namespace
}  // namespace srcnn

using namespace srcnn;

// ================================================================================================
extern "C" {

int srcnn_abi_version(void) { return SRCNN_AMD_ABI_VERSION; }

const char* srcnn_last_error(void) { return srcnn::last_error(); }

int srcnn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int srcnn_init(int device)
{
    std::lock_guard<std::mutex> lk(G.mu);
    if (!G.ctxs.empty()) {
        if (device >= 0 && device != G.ctxs[0]->device)
            return fail(SRCNN_E_ARG, "srcnn_init: already bound to device %d (asked for %d)", G.ctxs[0]->device, device);
        return SRCNN_OK;
    }
    if (device < 0) return lazy_init_locked();
    return make_context_locked(device);
}

int srcnn_init_devices(const int* devices, int n)
{
    std::lock_guard<std::mutex> lk(G.mu);
    std::vector<int> want;
    if (!devices || n <= 0) {
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return fail(SRCNN_E_NODEVICE, "no HIP device visible; this library has no CPU path");
        for (int d = 0; d < have; ++d) want.push_back(d);
    } else {
        if (n > 64) return fail(SRCNN_E_ARG, "srcnn_init_devices: at most 64 contexts");
        want.assign(devices, devices + n);
    }
    if (!G.ctxs.empty()) {
        // idempotent for the same list; a prefix may be extended (srcnn_init(0) followed by srcnn_init_devices({0,1,..}))
        for (size_t k = 0; k < G.ctxs.size(); ++k)
            if (k >= want.size() || want[k] != G.ctxs[k]->device)
                return fail(SRCNN_E_ARG, "srcnn_init_devices: context %zu is already bound to device %d; call srcnn_shutdown first",
                            k, G.ctxs[k]->device);
    }
    for (size_t k = G.ctxs.size(); k < want.size(); ++k) {
        int rc = make_context_locked(want[k]);
        if (rc) return rc;
    }
    return SRCNN_OK;
}

int srcnn_context_count(void) { return context_count(); }

int srcnn_context_device(int k)
{
    Ctx* cx = context_at(k);
    return cx ? cx->device : fail(SRCNN_E_ARG, "no context %d", k);
}

int srcnn_set_context(int k)
{
    if (int rc = ensure_init()) return rc;
    if (k < 0 || k >= context_count()) return fail(SRCNN_E_ARG, "no context %d (have %d)", k, context_count());
    const int prev = t_ctx;
    t_ctx = k;
    Ctx* cx = context_at(k);
    if (cx) (void)bind(*cx);
    return prev;
}

int srcnn_get_context(void) { return t_ctx; }

void srcnn_shutdown(void)
{
    srcnn_comm_destroy();
    async_chain_reset();
    std::vector<std::unique_ptr<Ctx>> dying;
    {
        std::lock_guard<std::mutex> lk(G.mu);
        if (G.ctxs.empty()) return;
        G.nctx.store(0);
        dying.swap(G.ctxs);
        G.stream_ctx.clear();
    }
    for (auto& cx : dying) release_context(*cx);
    t_ctx = 0;
}

// Release what idle lanes and unreferenced cache entries hold (scratch, page-locked staging, contribution tables) without
// shutting the library down: a long-lived process that once handled a very large image gets its memory back.
int srcnn_trim(void)
{
    if (int rc = ensure_init()) return rc;
    for (int k = 0; k < context_count(); ++k) {
        Ctx* cx = context_at(k);
        if (!cx) continue;
        if (int rc = bind(*cx)) return rc;
        {
            std::lock_guard<std::mutex> lk(cx->lane_mu);
            for (auto& l : cx->lanes)
                if (!l->busy) { (void)hipStreamSynchronize(l->st); (void)hipStreamSynchronize(l->copy_st); l->release_buffers(); }
        }
        {
            std::lock_guard<std::mutex> bl(cx->bounce.mu);
            (void)hipDeviceSynchronize();
            cx->bounce.release();
        }
        {
            std::lock_guard<std::mutex> hl(cx->host_call.mu);      // (a convenience call in flight holds it: trim waits for it)
            (void)hipDeviceSynchronize();
            cx->host_call.release();
        }
        std::lock_guard<std::mutex> lk(cx->mu);
        bool any = false;
        for (auto it = cx->tables.begin(); it != cx->tables.end();) {
            if (it->second.use_count() == 1) {
                if (!any) { (void)hipDeviceSynchronize(); any = true; }
                it = cx->tables.erase(it);
            } else ++it;
        }
    }
    return SRCNN_OK;
}

int srcnn_set_mode(int mode)
{
    if (mode != SRCNN_MODE_STRICT && mode != SRCNN_MODE_FAST && mode != SRCNN_MODE_FAST_F16 && mode != SRCNN_MODE_RELAXED)
        return fail(SRCNN_E_ARG, "bad mode %d", mode);
#ifdef SRCNN_STRICT_ONLY
    if (mode != SRCNN_MODE_STRICT) return fail(SRCNN_E_UNSUPPORTED, "this is a strict-only build (make STRICT_ONLY=1): mode %d is not compiled in", mode);
#endif
    if (mode == SRCNN_MODE_RELAXED) mode |= (int)(G.relax_mask.load() & 0xfu) << 8;
    return G.mode.exchange(mode) & 0xff;
}

int srcnn_get_mode(void) { return G.mode.load() & 0xff; }

int srcnn_set_relaxation(unsigned mask)
{
    if (mask & ~0xfu) return fail(SRCNN_E_ARG, "bad relaxation mask 0x%x", mask);
    if ((mask & SRCNN_RELAX_L3_X64) && (mask & SRCNN_RELAX_L3_F32)) return fail(SRCNN_E_ARG, "layer 3 can be relaxed one way at a time");
    const unsigned prev = G.relax_mask.exchange(mask);
    int m = G.mode.load();
    while ((m & 0xff) == SRCNN_MODE_RELAXED && !G.mode.compare_exchange_weak(m, SRCNN_MODE_RELAXED | (int)(mask << 8))) {}
    return (int)prev;
}

size_t srcnn_set_workspace_limit(size_t bytes)
{
    return G.ws_budget.exchange(std::max<size_t>(bytes, 1u << 20));
}

int srcnn_device_name(char* buf, size_t cap)
{
    Ctx* cx = cur_ctx();
    if (!cx) return SRCNN_E_NODEVICE;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cx->device));
    snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return SRCNN_OK;
}

// ---- plumbing (all on the calling thread's current context) ------------------------------------
void* srcnn_dev_alloc(size_t bytes)
{
    if (!cur_ctx()) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail(SRCNN_E_DEVMEM, "hipMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
void srcnn_dev_free(void* p) { if (p) (void)hipFree(p); }
void* srcnn_host_alloc_pinned(size_t bytes)
{
    Ctx* cx = cur_ctx();
    if (!cx) return nullptr;
    void* p = pinned_alloc(*cx, bytes);
    if (p) { std::lock_guard<std::mutex> lk(g_pinned_mu); pinned_blocks()[reinterpret_cast<uintptr_t>(p)] = bytes ? bytes : 1; }
    return p;
}
void srcnn_host_free_pinned(void* p)
{
    if (!p) return;
    { std::lock_guard<std::mutex> lk(g_pinned_mu); pinned_blocks().erase(reinterpret_cast<uintptr_t>(p)); }
    (void)hipHostFree(p);
}

int srcnn_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
    Ctx* cx = ctx_for_stream(stream);
    if (!cx) return SRCNN_E_NODEVICE;
    if (int rc = bind(*cx)) return rc;
    return copy_h2d_any(*cx, dst, src, bytes, (hipStream_t)stream);
}
int srcnn_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
    Ctx* cx = ctx_for_stream(stream);
    if (!cx) return SRCNN_E_NODEVICE;
    if (int rc = bind(*cx)) return rc;
    return copy_d2h_any(*cx, dst, src, bytes, (hipStream_t)stream);
}
int srcnn_memset_dev(void* dst, int byte, size_t bytes, void* stream)
{
    if (!ctx_for_stream(stream)) return SRCNN_E_NODEVICE;
    HIP_TRY(hipMemsetAsync(dst, byte, bytes, (hipStream_t)stream));
    return SRCNN_OK;
}
int srcnn_stream_create(void** stream)
{
    Ctx* cx = cur_ctx();
    if (!cx) return SRCNN_E_NODEVICE;
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    { std::lock_guard<std::mutex> lk(G.mu); G.stream_ctx[s] = cx->index; }
    *stream = s;
    return SRCNN_OK;
}
int srcnn_stream_destroy(void* stream)
{
    if (!stream) return SRCNN_OK;
    Ctx* cx = ctx_for_stream(stream);
    if (!cx) return SRCNN_E_NODEVICE;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    std::unique_ptr<Workspace> ws;
    {
        std::lock_guard<std::mutex> lk(cx->mu);
        auto it = cx->ws.find((hipStream_t)stream);
        if (it != cx->ws.end()) { ws = std::move(it->second); cx->ws.erase(it); }
    }
    if (ws) { std::lock_guard<std::mutex> wl(ws->mu); ws->release(); }
    { std::lock_guard<std::mutex> lk(G.mu); G.stream_ctx.erase((hipStream_t)stream); }
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return SRCNN_OK;
}
int srcnn_stream_sync(void* stream)
{
    if (!ctx_for_stream(stream)) return SRCNN_E_NODEVICE;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return SRCNN_OK;
}
int srcnn_device_sync(void) { if (!cur_ctx()) return SRCNN_E_NODEVICE; HIP_TRY(hipDeviceSynchronize()); return SRCNN_OK; }
int srcnn_event_create(void** ev)
{
    if (!cur_ctx()) return SRCNN_E_NODEVICE;
    hipEvent_t e; HIP_TRY(hipEventCreate(&e)); *ev = e; return SRCNN_OK;
}
int srcnn_event_destroy(void* ev) { if (ev) HIP_TRY(hipEventDestroy((hipEvent_t)ev)); return SRCNN_OK; }
int srcnn_event_record(void* ev, void* stream)
{
    if (!ctx_for_stream(stream)) return SRCNN_E_NODEVICE;
    HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return SRCNN_OK;
}
int srcnn_stream_wait_event(void* stream, void* ev)
{
    if (!ctx_for_stream(stream)) return SRCNN_E_NODEVICE;
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return SRCNN_OK;
}
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
    int rc;
    if ((rc = check_y_path_args(d_in, w, h, dw, dh, filter, d_out))) return rc;
    StreamCall sc(stream);
    if (sc.rc) return sc.rc;
    return y_path_frame(sc.c, d_in, w, h, dw, dh, filter, d_out);
}

int srcnn_y_upscale2x_f32_dev(const float* d_in, unsigned w, unsigned h, float* d_out, void* stream)
{
    return srcnn_y_path_f32_dev(d_in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, d_out, stream);
}

int srcnn_y_upscale2x_f32_batch_dev(const float* d_in, unsigned w, unsigned h, unsigned nframes, float* d_out,
                                    void* stream)
{
    if (nframes == 0) return fail(SRCNN_E_ARG, "nframes == 0");
    int rc;
    if ((rc = check_y_path_args(d_in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, d_out))) return rc;
    StreamCall sc(stream);
    if (sc.rc) return sc.rc;
    return batch_frames(sc.c, d_in, w, h, nframes, d_out);
}

int srcnn_batch_graph_create(const float* d_in, unsigned w, unsigned h, unsigned nframes, float* d_out, void* stream,
                             void** graph)
{
    if (!graph) return fail(SRCNN_E_ARG, "graph == NULL");
    if (!stream) return fail(SRCNN_E_ARG, "graph capture needs a non-default stream");
    if (nframes == 0) return fail(SRCNN_E_ARG, "nframes == 0");
    int rc;
    if ((rc = check_y_path_args(d_in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, d_out))) return rc;
    Ctx* cx = ctx_for_stream(stream);
    if (!cx) return SRCNN_E_NODEVICE;
    auto bg = std::make_unique<BatchGraph>();
    bg->cx = cx;
    bg->stream = (hipStream_t)stream;
    Call c;
    c.cx = cx; c.s = bg->stream; c.ws = &bg->ws; c.mode = G.mode.load(); c.hold = &bg->tables;
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
    if (int rc = bind(*b->cx)) return rc;
    HIP_TRY(hipGraphLaunch(b->exec, b->stream));
    return SRCNN_OK;
}

int srcnn_batch_graph_destroy(void* graph)
{
    if (!graph) return SRCNN_OK;
    BatchGraph* b = static_cast<BatchGraph*>(graph);
    (void)hipSetDevice(b->cx->device);
    (void)hipStreamSynchronize(b->stream);
    (void)hipGraphExecDestroy(b->exec);
    b->ws.release();
    delete b;
    return SRCNN_OK;
}

int srcnn_y_upscale2x_f32_band_dev(const float* d_in, unsigned w, unsigned h, unsigned row0, unsigned rows,
                                   float* d_out_band, void* stream)
{
    int rc;
    if ((rc = check_plane(d_in, w, h, d_out_band))) return rc;
    if (rows == 0) return fail(SRCNN_E_ARG, "rows == 0");
    if ((unsigned long long)row0 + rows > 2ull * h) return fail(SRCNN_E_ARG, "band [%u,+%u) outside the %u output rows", row0, rows, 2 * h);
    StreamCall sc(stream);
    if (sc.rc) return sc.rc;
    return y_path_range(sc.c, d_in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, row0, row0 + rows, d_out_band);
}

// ---- per-kernel timing -------------------------------------------------------------------------
int srcnn_profile_enable(int on) { return G.profiling.exchange(on != 0) ? 1 : 0; }

int srcnn_profile_reset(void)
{
    for (int k = 0; k < context_count(); ++k) {
        Ctx* cx = context_at(k);
        if (!cx) continue;
        (void)hipSetDevice(cx->device);
        std::lock_guard<std::mutex> lk(cx->mu);
        drain_spans_locked(*cx);
        for (int i = 0; i < SRCNN_STAGE_COUNT; ++i) { cx->stage_ms[i] = 0; cx->stage_n[i] = 0; }
    }
    if (Ctx* cx = context_at(srcnn_get_context())) (void)hipSetDevice(cx->device);
    return SRCNN_OK;
}

int srcnn_profile_read(int stage, double* total_ms, unsigned long long* launches)
{
    if (stage < 0 || stage >= SRCNN_STAGE_COUNT) return fail(SRCNN_E_ARG, "bad stage %d", stage);
    double ms = 0; unsigned long long n = 0;
    for (int k = 0; k < context_count(); ++k) {            // summed over the contexts of the process
        Ctx* cx = context_at(k);
        if (!cx) continue;
        (void)hipSetDevice(cx->device);
        std::lock_guard<std::mutex> lk(cx->mu);
        drain_spans_locked(*cx);
        ms += cx->stage_ms[stage]; n += cx->stage_n[stage];
    }
    if (Ctx* cx = context_at(srcnn_get_context())) (void)hipSetDevice(cx->device);
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    return SRCNN_OK;
}

// ---- clock probe: at which shader clock did each layer-1+2 launch run? ----
// When on, every k_conv12_mfma launch stamps s_memtime (shader cycles) and s_memrealtime (constant 100 MHz) at the start and
// the end of its workgroup 0, which is resident for the whole launch: cycles / ticks x 100 = MHz, ticks / 100 = microseconds.
// This is how "the same kernels run slower inside a ProcessSRCNN call" is told apart into clock and everything else
// (tools/process_clock_probe.py, profiles/r04_process_clock.txt).
int srcnn_debug_clock_probe(int on)
{
    if (int rc = ensure_init()) return rc;
    for (int k = 0; k < context_count(); ++k) {
        Ctx* cx = context_at(k);
        if (!cx) continue;
        (void)hipSetDevice(cx->device);
        if (on && !cx->clock_buf) {
            void* p = nullptr;
            if (hipMalloc(&p, sizeof(unsigned long long) * 2 * kClockSlots) != hipSuccess) return fail(SRCNN_E_DEVMEM, "clock probe buffer");
            cx->clock_buf = static_cast<unsigned long long*>(p);
        }
        if (cx->clock_buf) { (void)hipMemset(cx->clock_buf, 0, sizeof(unsigned long long) * 2 * kClockSlots); (void)hipDeviceSynchronize(); }
        cx->clock_n = 0;
    }
    if (Ctx* cur = context_at(srcnn_get_context())) (void)hipSetDevice(cur->device);
    return G.clock_probe.exchange(on != 0) ? 1 : 0;
}

// launches recorded on `context` since the probe was switched on (in launch order); writes at most `cap` (cycles, ticks) pairs
int srcnn_debug_clock_read(int context, unsigned long long* cycles, unsigned long long* ticks, int cap)
{
    Ctx* cx = context_at(context);
    if (!cx || !cx->clock_buf) return fail(SRCNN_E_ARG, "no clock probe on context %d", context);
    (void)hipSetDevice(cx->device);
    HIP_TRY(hipDeviceSynchronize());
    const unsigned n = std::min(cx->clock_n.load(), kClockSlots);
    std::vector<unsigned long long> host(2 * (size_t)n);
    if (n) HIP_TRY(hipMemcpy(host.data(), cx->clock_buf, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost));
    for (unsigned i = 0; i < n && (int)i < cap; ++i) {
        if (cycles) cycles[i] = host[2 * i];
        if (ticks) ticks[i] = host[2 * i + 1];
    }
    if (Ctx* cur = context_at(srcnn_get_context())) (void)hipSetDevice(cur->device);
    return (int)n;
}

// the same for ONE context (which device is the straggler of a node-level call?)
int srcnn_profile_read_context(int context, int stage, double* total_ms, unsigned long long* launches)
{
    if (stage < 0 || stage >= SRCNN_STAGE_COUNT) return fail(SRCNN_E_ARG, "bad stage %d", stage);
    Ctx* cx = context_at(context);
    if (!cx) return fail(SRCNN_E_ARG, "no context %d", context);
    (void)hipSetDevice(cx->device);
    {
        std::lock_guard<std::mutex> lk(cx->mu);
        drain_spans_locked(*cx);
        if (total_ms) *total_ms = cx->stage_ms[stage];
        if (launches) *launches = cx->stage_n[stage];
    }
    if (Ctx* cur = context_at(srcnn_get_context())) (void)hipSetDevice(cur->device);
    return SRCNN_OK;
}

// ---- stage-level -------------------------------------------------------------------------------
int srcnn_resample_f32_dev(const float* d_in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                           float* d_out, void* stream)
{
    int rc;
    if ((rc = check_y_path_args(d_in, w, h, dw, dh, filter, d_out))) return rc;
    if (dh > (1u << 20) || h > (1u << 20)) return fail(SRCNN_E_UNSUPPORTED, "too many rows");
    StreamCall sc(stream);
    if (sc.rc) return sc.rc;
    rc = resample_rows_range(sc.c, d_in, w, h, dw, dh, filter, 0, dh, d_out);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

int srcnn_conv1_f32_dev(const float* d_y, unsigned w, unsigned h, float* d_c1, void* stream)
{
    int rc;
    if ((rc = check_plane(d_y, w, h, d_c1))) return rc;
    if (h > 65535u * 4u) return fail(SRCNN_E_UNSUPPORTED, "too many rows");
    if (!ctx_for_stream(stream)) return SRCNN_E_NODEVICE;
    launch_conv1_planes(d_y, (int)w, (int)h, d_c1, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

int srcnn_conv2_f32_dev(const float* d_c1, unsigned w, unsigned h, float* d_c2, void* stream)
{
    int rc;
    if ((rc = check_plane(d_c1, w, h, d_c2))) return rc;
    if (!ctx_for_stream(stream)) return SRCNN_E_NODEVICE;
    launch_conv2_planes(d_c1, (size_t)w * h, d_c2, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

int srcnn_conv3_f32_dev(const float* d_c2, unsigned w, unsigned h, float* d_out, void* stream)
{
    int rc;
    if ((rc = check_plane(d_c2, w, h, d_out))) return rc;
    if (h > 65535u * 16u) return fail(SRCNN_E_UNSUPPORTED, "too many rows");
    if (!ctx_for_stream(stream)) return SRCNN_E_NODEVICE;
    Call c;
    c.mode = G.mode.load();
    launch_conv3(d_c2, (size_t)w * h, (int)w, (int)h, 0, (int)h, d_out, 0, (int)h, c.relax(), (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
}

int srcnn_conv12_f32_dev(const float* d_y, unsigned w, unsigned h, float* d_c2, void* stream)
{
    int rc;
    if ((rc = check_plane(d_y, w, h, d_c2))) return rc;
    if (h > 65535u * 4u) return fail(SRCNN_E_UNSUPPORTED, "too many rows");
    Call c;
    c.cx = ctx_for_stream(stream);
    if (!c.cx) return SRCNN_E_NODEVICE;
    c.s = (hipStream_t)stream; c.mode = G.mode.load();
    run_conv12(c, d_y, (int)w, (int)h, 0, (int)h, d_c2, (size_t)w * h, 0, (int)h);
    HIP_TRY(hipGetLastError());
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
    int rc;
    if ((rc = check_plane(d_up, w, h, d_out))) return rc;
#ifdef SRCNN_STRICT_ONLY
    (void)d_dbg; (void)stream;
    return fail(SRCNN_E_UNSUPPORTED, "strict-only build: the fused fp16 kernel is not compiled in");
#else
    Ctx* cx = ctx_for_stream(stream);
    if (!cx) return SRCNN_E_NODEVICE;
    launch_fused_f16(d_up, (int)w, (int)h, 0, (int)h, d_out, 0, (int)h, cx->fused_w, cx->num_cus, (hipStream_t)stream, d_dbg);
    HIP_TRY(hipGetLastError());
    return SRCNN_OK;
#endif
}

// The switches this process runs with (srcnn_settings.hpp): text into buf (NUL-terminated, truncated to cap), returns the
// length the full text needs.  markdown != 0: the rows of DESIGN.md section 6 (defaults, not current values).  No device needed.
int srcnn_debug_settings(char* buf, size_t cap, int markdown)
{
    const std::string t = settings().describe(markdown != 0);
    if (buf && cap) snprintf(buf, cap, "%s", t.c_str());
    return (int)t.size();
}

// test hook: number of cached contribution tables / of ProcessSRCNN lanes created so far, summed over the contexts
int srcnn_debug_counts(int* tables, int* lanes)
{
    int nt = 0, nl = 0;
    for (int k = 0; k < context_count(); ++k) {
        Ctx* cx = context_at(k);
        if (!cx) continue;
        { std::lock_guard<std::mutex> lk(cx->mu); nt += (int)cx->tables.size(); }
        { std::lock_guard<std::mutex> lk(cx->lane_mu); nl += (int)cx->lanes.size(); }
    }
    if (tables) *tables = nt;
    if (lanes) *lanes = nl;
    return SRCNN_OK;
}

}  // extern "C"
