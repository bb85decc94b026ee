// srcnn_comm.cpp -- RCCL over xGMI for the ONE exchange step of the path: gathering the output
// bands of a frame that was tiled across the GPUs of a node (the reference has no
// counterpart -- it is single-process OpenMP).  One process per GPU.  librccl is opened lazily so
// single-GPU users never load it.  The gather is peer->root point-to-point (ncclSend/ncclRecv in
// one group): on MI355X's fully connected xGMI each band lands over its own link, instead of a
// ring that would be bound by one link.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/auxv.h>

#include "../../include/srcnn_amd.h"

#include "srcnn_host.hpp"
#include "srcnn_watchdog.hpp"

namespace {

static_assert(sizeof(ncclUniqueId) == SRCNN_COMM_ID_BYTES, "unique id size");

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;           // optional: what a missed deadline calls to free stuck ranks
};

Rccl R;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_nranks = 1;
int g_device = 0;                       // the device the communicator is bound to
float* g_token = nullptr;
hipStream_t g_comm_stream = nullptr;    // the tiled path's gathers run here, beside the compute stream
std::vector<hipEvent_t> g_events;       // sub-band "kernels queued" events + one "gathers done" event
std::mutex g_mu;
std::vector<unsigned long long> g_verified;   // checksums of the tables every rank was seen to agree on (bounded)
unsigned long long* g_check = nullptr;  // SRCNN_COMM_CHECK: two device words for the min/max all-reduce of a table's checksum
unsigned long long* g_check_host = nullptr;   // ... and their page-locked staging (lives as long as the communicator: a copy queued from it
                                              //     can never outlive its source, as one from a stack array could on an early return)
std::mutex g_check_mu;                        // one verification at a time (the words are shared)
std::atomic<bool> g_poisoned{false};    // a deadline was missed: the communicator was aborted, every later call fails at once

// ---- deadlines -------------------------------------------------------------------------------------------------------
// Nothing in this file may wait for a peer without a bound: a rank that died, or one that derived a different counts table,
// would otherwise leave every other rank of the job hanging inside RCCL (ncclGroupEnd blocks on the host while connections
// are set up; a send / recv kernel spins on the device until its peer shows up).  Two mechanisms:
//   * host side: Watchdog -- armed around every RCCL call that can block; when the deadline passes it calls ncclCommAbort
//     (the one RCCL call that is allowed from another thread and makes blocked calls return) and marks the communicator dead;
//   * device side: wait_stream_deadline() -- a polled wait for everything queued on a stream, used by srcnn_comm_barrier and
//     srcnn_comm_wait; on a miss it aborts the communicator the same way, which also ends the spinning kernels.
// SRCNN_COMM_TIMEOUT_MS / srcnn_comm_set_timeout_ms: default 60 s; 0 = wait for ever (the round-3 behaviour).
std::atomic<int> g_timeout_ms{(int)srcnn::settings().comm_timeout_ms};

// Communicator generation: bumped by every init and destroy.  An abort carries the generation of the communicator it was
// armed for and is ignored when that communicator is gone (a destroy + init on another thread must not get the new one
// aborted, and the freed one must not be touched again).  g_abort_mu is held while ncclCommAbort runs; srcnn_comm_destroy takes
// it too, so a destroy never runs ncclCommDestroy beside an abort of the same communicator.
std::atomic<unsigned> g_gen{0};
std::mutex g_abort_mu;
bool g_abort_called = false;            // under g_abort_mu: ncclCommAbort ran for the current generation

void abort_comm(ncclComm_t comm, unsigned gen)
{
    std::lock_guard<std::mutex> lk(g_abort_mu);
    if (gen != g_gen.load() || g_abort_called) return;
    g_poisoned = true;
    g_abort_called = true;
    (void)hipSetDevice(g_device);                          // (the watchdog's own thread has never bound a device)
    if (R.CommAbort && comm) (void)R.CommAbort(comm);
}

// (csrc/srcnn_watchdog.hpp: HIP-free, so the same class runs under ThreadSanitizer in tests/host/)
struct CommWatchdog {
    // (the mark is made under g_abort_mu like the abort itself: a destroy + init between the generation check and the store
    //  would otherwise leave the NEW communicator poisoned with g_abort_called still false -- ADVICE r5)
    srcnn::Watchdog w{[](unsigned gen) { std::lock_guard<std::mutex> lk(g_abort_mu); if (gen == g_gen.load()) g_poisoned = true; },
                      [](void* comm, unsigned gen) { abort_comm(static_cast<ncclComm_t>(comm), gen); }};
    bool arm(ncclComm_t comm, unsigned gen) { return w.arm(comm, gen, g_timeout_ms.load()); }
    bool disarm(bool armed) { return w.disarm(armed); }
};
CommWatchdog& watchdog() { static CommWatchdog* w = new CommWatchdog; return *w; }      // (never destroyed: its thread outlives main)

// everything queued on `s` has completed, or the deadline passed (then the communicator is aborted): hipSuccess / hipErrorNotReady
hipError_t wait_stream_deadline(hipStream_t s, ncclComm_t comm, unsigned gen)
{
    const int ms = g_timeout_ms.load();
    if (ms <= 0) return srcnn::wait_stream(s);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
    for (int n = 0;; ++n) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() >= deadline) { abort_comm(comm, gen); return hipErrorNotReady; }
        if (n > 64) std::this_thread::sleep_for(std::chrono::microseconds(n < 2000 ? 50 : 500));
    }
}

// A consistent view of the communicator for one call (taken under g_mu; the collective itself runs outside it so that a
// rank blocked in RCCL never blocks srcnn_comm_rank on another thread).
struct CommView { ncclComm_t comm; int rank, nranks, device; unsigned gen; };
bool view(CommView& v)
{
    std::lock_guard<std::mutex> lk(g_mu);
    v = CommView{g_comm, g_rank, g_nranks, g_device, g_gen.load()};
    return v.comm != nullptr;
}
// the streams communication was queued on since init (bounded; what srcnn_comm_destroy drains under the deadline)
std::vector<hipStream_t> g_streams;
// ... and, per stream, an event recorded behind the last communication queued on it: an event stays valid (and completes) after
// its stream has been destroyed, so srcnn_comm_destroy can wait -- under the deadline -- for communication on a RAW HIP
// stream of the caller's as well, which it must not query once the caller may have destroyed it
std::vector<hipEvent_t> g_stream_events;      // parallel to g_streams; nullptr until the first mark
void note_stream(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (std::find(g_streams.begin(), g_streams.end(), s) == g_streams.end() && g_streams.size() < 64) {
        g_streams.push_back(s);
        g_stream_events.push_back(nullptr);
    }
}
void mark_stream(hipStream_t s)               // after communication has been queued on s
{
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const auto it = std::find(g_streams.begin(), g_streams.end(), s);
        if (it == g_streams.end()) return;
        hipEvent_t& slot = g_stream_events[it - g_streams.begin()];
        if (!slot && hipEventCreateWithFlags(&slot, hipEventDisableTiming) != hipSuccess) { slot = nullptr; return; }
        ev = slot;
    }
    (void)hipEventRecord(ev, s);
}
thread_local char g_cerr[256];

int load()
{
    if (R.h) return 0;
    // RCCL must sit on the SAME HIP runtime this library is bound to.  A process that also imports PyTorch has two
    // ROCm stacks on disk (the wheel bundles its own libamdhip64 / librccl); which libamdhip64 this library got depends
    // on load order (the dynamic loader dedups by SONAME).  An RCCL from the other stack fails at ncclCommInitRank
    // ("unhandled cuda error", measured: profiles/r02_import_order.txt).  So: find the file hipFree comes from and
    // take the librccl that lives next to it, by absolute path; only then fall back to the loader's search.
    void* h = nullptr;
    Dl_info info;
    // SRCNN_RCCL_LIB: an explicit library, consulted first and alone (a site build of RCCL; the test-suite's stand-in that
    // moves data between processes sharing ONE device, tests/rccl_double/ -- the installed RCCL refuses two ranks per device)
    if (const char* path = srcnn::settings().rccl_lib.empty() ? nullptr : srcnn::settings().rccl_lib.c_str()) {
        // A TRUST BOUNDARY: this loads and runs whatever the environment names, with the process's privileges -- the same
        // power LD_PRELOAD has, and refused in the same situation: a process the kernel marked secure (set-uid / set-gid /
        // file capabilities: AT_SECURE) ignores the variable's intent and fails instead of loading it.
        if (getauxval(AT_SECURE)) {
            snprintf(g_cerr, sizeof g_cerr, "SRCNN_RCCL_LIB is refused in a secure-execution (set-uid / set-gid / capabilities) process");
            srcnn::set_last_error(g_cerr);
            return SRCNN_E_COMM;
        }
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            snprintf(g_cerr, sizeof g_cerr, "dlopen(SRCNN_RCCL_LIB=%s) failed: %s", path, dlerror());
            srcnn::set_last_error(g_cerr);
            return SRCNN_E_COMM;
        }
    } else
    if (dladdr(reinterpret_cast<void*>(&hipFree), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash);
            for (const char* leaf : {"/librccl.so.1", "/librccl.so"}) {
                h = dlopen((dir + leaf).c_str(), RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
        }
    }
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { if (h) break; h = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
    if (!h) {
        snprintf(g_cerr, sizeof g_cerr, "dlopen(librccl) failed: %s", dlerror());
        srcnn::set_last_error(g_cerr);
        return SRCNN_E_COMM;
    }
    Rccl r;
    const char* missing = nullptr;
#define SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(h, "nccl" #f)); if (!r.f && !missing) missing = "nccl" #f;
    SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(GroupStart) SYM(GroupEnd)
    SYM(Send) SYM(Recv) SYM(AllGather) SYM(AllReduce) SYM(GetErrorString)
#undef SYM
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));      // optional
    if (missing) {      // leave R untouched so that the next call reports the same failure instead of calling NULL
        snprintf(g_cerr, sizeof g_cerr, "librccl lacks %s", missing);
        srcnn::set_last_error(g_cerr);
        dlclose(h);
        return SRCNN_E_COMM;
    }
    r.h = h;
    R = r;
    return 0;
}

int comm_fail(const char* what)
{
    snprintf(g_cerr, sizeof g_cerr, "%s", what);
    srcnn::set_last_error(g_cerr);
    return SRCNN_E_COMM;
}

int poisoned_fail()
{
    return comm_fail("the communicator was aborted after a missed deadline (SRCNN_COMM_TIMEOUT_MS): destroy and re-create it");
}

// FNV-1a over the bytes of a table: what SRCNN_COMM_CHECK=1 compares across the ranks before a gather trusts it
unsigned long long fnv1a(const void* p, size_t n, unsigned long long h = 1469598103934665603ull)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

#define NCCL_TRY(expr)                                                                         \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            snprintf(g_cerr, sizeof g_cerr, "%s -> %s", #expr, R.GetErrorString(r_));          \
            srcnn::set_last_error(g_cerr);                                                     \
            return SRCNN_E_COMM;                                                               \
        }                                                                                      \
    } while (0)

}  // namespace

namespace srcnn {
constexpr int kTiledPlanGrid = 512, kTiledPlanTileRows = 16;
// ... written down as constants ON PURPOSE (ranks with different switches must derive the same table), but tied to the kernel's
// geometry at compile time: a change of the layer-1+2 tile or of its workgroups per CU has to be made here as well
static_assert(kTiledPlanTileRows == srcnn::kConv12TileRows && kTiledPlanGrid == srcnn::kConv12BlocksPerCU * 256,
              "the tiled band plan assumes the production layer-1+2 geometry on a 256-CU device");
std::vector<unsigned> tiled_cuts(unsigned out_w, unsigned out_h, int rank, int nranks, int npieces)
{
    unsigned b0 = 0, bn = 0;
    srcnn_band_rows(out_h, rank, nranks, &b0, &bn);
    if (bn == 0) return {b0, b0};
    if (npieces <= 1 || bn < 256) {                               // short band: equal pieces, nothing to gain from planning
        std::vector<unsigned> c((size_t)std::max(npieces, 1) + 1);
        for (int i = 0; i <= std::max(npieces, 1); ++i) c[i] = b0 + (unsigned)((unsigned long long)bn * (unsigned)i / (unsigned)std::max(npieces, 1));
        return c;
    }
    // decreasing targets n : n-1 : ... : 1 (40 / 30 / 20 / 10 % for 4 pieces): the piece whose gather stays exposed is the smallest.
    // The grid is that of an MI355X (256 CUs), NOT a device query: the table must not depend on which rank computes it.
    std::vector<double> frac((size_t)npieces - 1);
    const double total = 0.5 * npieces * (npieces + 1);
    for (int i = 0; i + 1 < npieces; ++i) frac[i] = (double)(npieces - i) / total;
    // ... nor on any switch of this process: the production layer-1+2 geometry (2 workgroups x 256 CUs, 16-row tiles) is
    // written down here, so ranks started with different SRCNN_* environments still derive the same gather table.
    constexpr int grid = kTiledPlanGrid, tile_rows = kTiledPlanTileRows;
    return plan_cuts(b0, b0 + bn, out_w, out_h, frac.data(), npieces - 1, grid, tile_rows);
}
}  // namespace srcnn

extern "C" {

int srcnn_comm_unique_id(unsigned char id[SRCNN_COMM_ID_BYTES])
{
    if (!srcnn::cur_ctx()) return SRCNN_E_NODEVICE;
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = load()) return rc;
    ncclUniqueId u;
    NCCL_TRY(R.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return SRCNN_OK;
}

int srcnn_comm_init(const unsigned char id[SRCNN_COMM_ID_BYTES], int rank, int nranks)
{
    // bind the calling thread to the current context's device first: a helper thread that never touched HIP would
    // otherwise create the communicator on device 0
    srcnn::Ctx* cx = srcnn::cur_ctx();
    if (!cx) return SRCNN_E_NODEVICE;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_comm) return comm_fail("srcnn_comm_init: communicator already initialised");
    if (!id || rank < 0 || nranks <= 0 || rank >= nranks) return comm_fail("srcnn_comm_init: bad rank / nranks / id");
    if (int rc = load()) return rc;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    NCCL_TRY(R.CommInitRank(&comm, nranks, u, rank));
    float* token = nullptr;
    if (hipMalloc((void**)&token, sizeof(float)) != hipSuccess || hipMemset(token, 0, sizeof(float)) != hipSuccess) {
        (void)hipFree(token);
        R.CommDestroy(comm);
        srcnn::set_last_error("srcnn_comm_init: device allocation failed");
        return SRCNN_E_DEVMEM;
    }
    if (hipMalloc((void**)&g_check, 2 * sizeof(unsigned long long)) != hipSuccess) g_check = nullptr;    // SRCNN_COMM_CHECK only
    if (hipHostMalloc((void**)&g_check_host, 2 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) g_check_host = nullptr;
    {
        std::lock_guard<std::mutex> ak(g_abort_mu);
        ++g_gen;
        g_abort_called = false;
        g_poisoned = false;
    }
    g_comm = comm; g_token = token;
    g_rank = rank; g_nranks = nranks; g_device = cx->device;
    g_verified.clear();
    for (hipEvent_t e : g_stream_events) if (e) (void)hipEventDestroy(e);
    g_stream_events.clear();
    g_streams.clear();
    return SRCNN_OK;
}

int srcnn_comm_destroy(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return SRCNN_OK;
    (void)hipSetDevice(g_device);
    {
        // an abort in flight (watchdog thread, or another thread's missed srcnn_comm_wait) finishes first; from here on aborts
        // armed for this communicator are stale and ignored
        std::lock_guard<std::mutex> ak(g_abort_mu);
        ++g_gen;
        bool drained = true;
        if (!g_poisoned.load()) {
            // A healthy communicator may still have sends / receives queued whose peer died after this rank's call returned: the
            // drain is bounded like every other wait on a peer (a bare hipDeviceSynchronize hung here for ever).  On a miss the
            // communicator is aborted -- which ends the spinning kernels -- instead of destroyed.
            const int ms = g_timeout_ms.load();
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
            // Streams are queried only when they are known to be alive: the NULL stream, the library's comm stream and streams
            // from srcnn_stream_create that have not been destroyed since (querying a destroyed handle crashes inside the
            // runtime).  Every stream communication was queued on -- a raw HIP stream of the caller's included -- also has an event
            // behind its last communication (mark_stream), and events outlive their streams: those are waited for below.
            std::vector<hipStream_t> streams;
            std::vector<hipEvent_t> marks;
            {
                std::lock_guard<std::mutex> gk(srcnn::G.mu);
                for (size_t i = 0; i < g_streams.size(); ++i) {
                    hipStream_t st = g_streams[i];
                    if (!st || srcnn::G.stream_ctx.count(st)) streams.push_back(st);
                    if (g_stream_events[i]) marks.push_back(g_stream_events[i]);
                }
            }
            for (hipEvent_t ev : marks) {
                for (int n = 0; drained; ++n) {
                    const hipError_t e = hipEventQuery(ev);
                    if (e != hipErrorNotReady) break;
                    if (ms > 0 && std::chrono::steady_clock::now() >= deadline) { drained = false; break; }
                    if (n > 64) std::this_thread::sleep_for(std::chrono::microseconds(n < 2000 ? 50 : 500));
                }
            }
            (void)hipGetLastError();
            if (g_comm_stream) streams.push_back(g_comm_stream);
            for (hipStream_t st : streams) {
                for (int n = 0; drained; ++n) {
                    const hipError_t e = hipStreamQuery(st);
                    if (e != hipErrorNotReady) break;                  // drained (or the stream is gone / in error: nothing to wait for)
                    if (ms > 0 && std::chrono::steady_clock::now() >= deadline) { drained = false; break; }
                    if (n > 64) std::this_thread::sleep_for(std::chrono::microseconds(n < 2000 ? 50 : 500));
                }
            }
        }
        if (g_abort_called) {
            // aborted after a missed deadline: ncclCommAbort has already ended its kernels and freed the communicator
        } else if (!drained || g_poisoned.load()) {
            // (a librccl without ncclCommAbort leaks the object rather than wait for peers that are gone)
            if (R.CommAbort) (void)R.CommAbort(g_comm);
        } else {
            R.CommDestroy(g_comm);
        }
        g_abort_called = false;
        g_poisoned = false;
    }
    g_comm = nullptr;
    hipFree(g_token); g_token = nullptr;
    hipFree(g_check); g_check = nullptr;
    if (g_check_host) { (void)hipHostFree(g_check_host); g_check_host = nullptr; }
    for (auto e : g_events) (void)hipEventDestroy(e);
    g_events.clear();
    if (g_comm_stream) (void)hipStreamDestroy(g_comm_stream);
    g_comm_stream = nullptr;
    for (hipEvent_t e : g_stream_events) if (e) (void)hipEventDestroy(e);
    g_stream_events.clear();
    g_streams.clear();
    g_rank = 0; g_nranks = 1;
    return SRCNN_OK;
}

int srcnn_comm_rank(int* rank, int* nranks)
{
    CommView v;
    if (!view(v)) return comm_fail("no communicator");
    if (rank) *rank = v.rank;
    if (nranks) *nranks = v.nranks;
    return SRCNN_OK;
}

namespace {
// Before a gather trusts a counts / offsets table, every rank contributes the table's checksum to a min/max all-reduce; ranks
// that derived different tables (which would otherwise pair a send with a receive of another size and hang or corrupt) all
// return SRCNN_E_COMM instead.  A table that was seen to agree is not checked again, so a steady stream of frames of one
// geometry pays 16 bytes of all-reduce per piece once.  On by default since round 5; SRCNN_COMM_CHECK=0 skips it.
int verify_table(const CommView& v, unsigned long long h, hipStream_t s)
{
    if (!srcnn::settings().comm_check) return SRCNN_OK;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (std::find(g_verified.begin(), g_verified.end(), h) != g_verified.end()) return SRCNN_OK;
        if (!g_check || !g_check_host) return comm_fail("SRCNN_COMM_CHECK: no words for the checksum");
    }
    std::lock_guard<std::mutex> one(g_check_mu);
    unsigned long long* words = g_check_host;
    words[0] = h; words[1] = ~h;                             // min(h) and min(~h) = ~max(h)
    if (hipMemcpyAsync(g_check, words, 2 * sizeof *words, hipMemcpyHostToDevice, s) != hipSuccess) return comm_fail("SRCNN_COMM_CHECK: upload failed");
    const bool armed = watchdog().arm(v.comm, v.gen);
    const ncclResult_t r = R.AllReduce(g_check, g_check, 2, ncclUint64, ncclMin, v.comm, s);
    const bool late = watchdog().disarm(armed);
    if (late || r != ncclSuccess) return comm_fail(late ? "SRCNN_COMM_CHECK: all-reduce missed its deadline" : "SRCNN_COMM_CHECK: all-reduce failed");
    if (wait_stream_deadline(s, v.comm, v.gen) != hipSuccess) return comm_fail("SRCNN_COMM_CHECK: a rank never arrived (deadline)");
    if (hipMemcpy(words, g_check, 2 * sizeof *words, hipMemcpyDeviceToHost) != hipSuccess) return comm_fail("SRCNN_COMM_CHECK: read-back failed");
    if (words[0] != ~words[1]) {
        snprintf(g_cerr, sizeof g_cerr, "SRCNN_COMM_CHECK: the ranks disagree about the gather table (this rank %016llx, min %016llx, max %016llx)",
                 h, words[0], ~words[1]);
        srcnn::set_last_error(g_cerr);
        return SRCNN_E_COMM;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_verified.size() >= 256) g_verified.clear();
    g_verified.push_back(h);
    return SRCNN_OK;
}

unsigned long long table_hash(const size_t* counts, const size_t* offsets, int nranks, int root)
{
    unsigned long long h = fnv1a(counts, sizeof(size_t) * (size_t)nranks);
    h = fnv1a(offsets, sizeof(size_t) * (size_t)nranks, h);
    const int tail[2] = {nranks, root};
    return fnv1a(tail, sizeof tail, h);
}
}  // namespace

// counts[r] floats from rank r land at d_recv + offsets[r] on the root.  counts / offsets must be identical on every rank: the
// callers in this library derive them from (width, height, nranks, pieces) alone; SRCNN_COMM_CHECK=1 verifies that across the
// ranks before the first gather with a given table.  A mismatch that slips through, or a rank that died, cannot hang the
// others for ever: the group is issued under the watchdog's deadline (see above) and the caller waits with srcnn_comm_wait.
int srcnn_comm_gatherv_at_f32(const float* d_send, const size_t* counts, const size_t* offsets, float* d_recv, int root,
                              void* stream)
{
    CommView v;
    if (!view(v)) return comm_fail("no communicator");
    if (g_poisoned.load()) return poisoned_fail();
    if (!counts || !offsets || root < 0 || root >= v.nranks) return comm_fail("srcnn_comm_gatherv_at_f32: bad counts / offsets / root");
    if (counts[v.rank] && !d_send) return comm_fail("srcnn_comm_gatherv_at_f32: d_send == NULL");
    if (v.rank == root && !d_recv) return comm_fail("srcnn_comm_gatherv_at_f32: d_recv == NULL on the root");
    if (hipSetDevice(v.device) != hipSuccess) return comm_fail("hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = verify_table(v, table_hash(counts, offsets, v.nranks, root), s)) return rc;
    // every Send/Recv of the group is attempted and GroupEnd always runs, so a failure never leaves the group open
    ncclResult_t first_bad = ncclSuccess;
    auto note = [&](ncclResult_t r) { if (r != ncclSuccess && first_bad == ncclSuccess) first_bad = r; };
    note_stream(s);
    const bool armed = watchdog().arm(v.comm, v.gen);
    note(R.GroupStart());
    if (v.rank == root) {
        for (int r = 0; r < v.nranks; ++r)
            if (r != root && counts[r]) note(R.Recv(d_recv + offsets[r], counts[r], ncclFloat, r, v.comm, s));
    } else if (counts[v.rank]) {
        note(R.Send(d_send, counts[v.rank], ncclFloat, root, v.comm, s));
    }
    note(R.GroupEnd());
    if (watchdog().disarm(armed)) return comm_fail("band gather: a peer did not arrive before the deadline (SRCNN_COMM_TIMEOUT_MS); communicator aborted");
    if (first_bad != ncclSuccess) {
        snprintf(g_cerr, sizeof g_cerr, "band gather failed: %s", R.GetErrorString(first_bad));
        srcnn::set_last_error(g_cerr);
        return SRCNN_E_COMM;
    }
    if (v.rank == root && counts[root] && d_recv + offsets[root] != d_send) {
        if (hipMemcpyAsync(d_recv + offsets[root], d_send, counts[root] * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
            srcnn::set_last_error("band gather: root's own copy failed");
            return SRCNN_E_HIP;
        }
    }
    mark_stream(s);
    return SRCNN_OK;
}

// counts[r] floats from rank r land at d_recv + sum(counts[0..r)) on the root: bands of unequal height (an
// output height that the rank count does not divide) assemble into one contiguous frame.
int srcnn_comm_gatherv_f32(const float* d_send, const size_t* counts, float* d_recv, int root, void* stream)
{
    CommView v;
    if (!view(v)) return comm_fail("no communicator");
    if (!counts) return comm_fail("srcnn_comm_gatherv_f32: bad counts / root");
    std::vector<size_t> offs((size_t)v.nranks);
    size_t pos = 0;
    for (int r = 0; r < v.nranks; ++r) { offs[r] = pos; pos += counts[r]; }
    return srcnn_comm_gatherv_at_f32(d_send, counts, offs.data(), d_recv, root, stream);
}

int srcnn_comm_gather_f32(const float* d_send, size_t count, float* d_recv, int root, void* stream)
{
    CommView v;
    if (!view(v)) return comm_fail("no communicator");
    std::vector<size_t> counts((size_t)v.nranks, count);
    return srcnn_comm_gatherv_f32(d_send, counts.data(), d_recv, root, stream);
}

int srcnn_band_rows(unsigned out_h, int rank, int nranks, unsigned* row0, unsigned* rows)
{
    if (nranks <= 0 || rank < 0 || rank >= nranks) return comm_fail("srcnn_band_rows: bad rank / nranks");
    const unsigned base = out_h / (unsigned)nranks, rem = out_h % (unsigned)nranks;
    if (rows) *rows = base + ((unsigned)rank < rem ? 1u : 0u);
    if (row0) *row0 = (unsigned)rank * base + std::min((unsigned)rank, rem);
    return SRCNN_OK;
}

// Piece `piece` of `npieces` of rank's band of an (out_w x out_h) tiled frame: rows [*row0, *row0 + *rows).  The pieces of a
// band partition it (large first, short last, cut where they fill whole rounds of the layer-1+2 grid: srcnn::tiled_cuts), the
// bands partition the frame; every rank derives the same table from (out_w, out_h, nranks, npieces) alone, which is what keeps
// the per-piece gathers of srcnn_comm_tiled_y_upscale2x_f32_dev consistent without any exchange.  A short band yields fewer
// pieces; the missing ones are empty.
int srcnn_tiled_piece(unsigned out_w, unsigned out_h, int rank, int nranks, int piece, int npieces, unsigned* row0, unsigned* rows)
{
    unsigned b0 = 0, bn = 0;
    if (int rc = srcnn_band_rows(out_h, rank, nranks, &b0, &bn)) return rc;
    if (npieces <= 0 || piece < 0 || piece >= npieces || out_w == 0) return comm_fail("srcnn_tiled_piece: bad piece / npieces / width");
    const std::vector<unsigned> cuts = srcnn::tiled_cuts(out_w, out_h, rank, nranks, npieces);
    const unsigned a = (size_t)piece < cuts.size() ? cuts[piece] : b0 + bn;
    const unsigned b = (size_t)piece + 1 < cuts.size() ? cuts[piece + 1] : b0 + bn;
    if (row0) *row0 = a;
    if (rows) *rows = b - a;
    return SRCNN_OK;
}

int srcnn_comm_tiled_y_upscale2x_f32_dev(const float* d_in, unsigned w, unsigned h, float* d_band, float* d_full, int root,
                                         int sub_bands, void* stream)
{
    CommView v;
    if (!view(v)) return comm_fail("no communicator");
    if (g_poisoned.load()) return poisoned_fail();
    if (!d_in || w == 0 || h == 0) return comm_fail("srcnn_comm_tiled: NULL pointer or zero dimension");
    if (root < 0 || root >= v.nranks) return comm_fail("srcnn_comm_tiled: bad root");
    if (v.rank == root && !d_full) return comm_fail("srcnn_comm_tiled: d_full == NULL on the root");
    if (hipSetDevice(v.device) != hipSuccess) return comm_fail("hipSetDevice failed");
    const unsigned dw = 2 * w, dh = 2 * h;
    const unsigned nsub = (unsigned)std::max(1, std::min(sub_bands <= 0 ? 4 : sub_bands, 16));
    hipStream_t s = (hipStream_t)stream;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_comm_stream && hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking) != hipSuccess) return comm_fail("comm stream");
        while (g_events.size() < nsub + 2) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return comm_fail("event create");
            g_events.push_back(e);
        }
    }
    // piece i of rank r: rows [row0_r + rows_r*i/nsub, row0_r + rows_r*(i+1)/nsub) -- every rank computes the same table
    unsigned my_row0 = 0, my_rows = 0;
    srcnn_band_rows(dh, v.rank, v.nranks, &my_row0, &my_rows);
    if (my_rows && !d_band) return comm_fail("srcnn_comm_tiled: d_band == NULL");
    // the comm stream must not start before what is already queued on the caller's stream (e.g. the upload of d_in on the
    // root, or a previous frame's use of d_full)
    if (hipEventRecord(g_events[nsub], s) != hipSuccess || hipStreamWaitEvent(g_comm_stream, g_events[nsub], 0) != hipSuccess)
        return comm_fail("srcnn_comm_tiled: stream hand-over failed");
    std::vector<size_t> counts((size_t)v.nranks), offs((size_t)v.nranks);
    for (unsigned i = 0; i < nsub; ++i) {
        unsigned a = 0, n = 0;
        srcnn_tiled_piece(dw, dh, v.rank, v.nranks, (int)i, (int)nsub, &a, &n);
        float* piece = d_band ? d_band + (size_t)(a - my_row0) * dw : nullptr;
        if (n) {
            int rc = srcnn_y_upscale2x_f32_band_dev(d_in, w, h, a, n, piece, stream);
            if (rc) return rc;
        }
        if (hipEventRecord(g_events[i], s) != hipSuccess || hipStreamWaitEvent(g_comm_stream, g_events[i], 0) != hipSuccess)
            return comm_fail("srcnn_comm_tiled: event hand-over failed");
        for (int r = 0; r < v.nranks; ++r) {
            unsigned ra = 0, rn = 0;
            srcnn_tiled_piece(dw, dh, r, v.nranks, (int)i, (int)nsub, &ra, &rn);
            counts[r] = (size_t)rn * dw;
            offs[r] = (size_t)ra * dw;
        }
        int rc = srcnn_comm_gatherv_at_f32(piece, counts.data(), offs.data(), d_full, root, g_comm_stream);
        if (rc) return rc;
    }
    if (hipEventRecord(g_events[nsub + 1], g_comm_stream) != hipSuccess || hipStreamWaitEvent(s, g_events[nsub + 1], 0) != hipSuccess)
        return comm_fail("srcnn_comm_tiled: final hand-over failed");
    return SRCNN_OK;
}

int srcnn_comm_allgather_f32(const float* d_send, size_t count, float* d_recv, void* stream)
{
    CommView v;
    if (!view(v)) return comm_fail("no communicator");
    if (g_poisoned.load()) return poisoned_fail();
    if (hipSetDevice(v.device) != hipSuccess) return comm_fail("hipSetDevice failed");
    note_stream((hipStream_t)stream);
    const bool armed = watchdog().arm(v.comm, v.gen);
    const ncclResult_t r = R.AllGather(d_send, d_recv, count, ncclFloat, v.comm, (hipStream_t)stream);
    if (watchdog().disarm(armed)) return comm_fail("all-gather: deadline missed while queueing; communicator aborted");
    NCCL_TRY(r);
    mark_stream((hipStream_t)stream);
    return SRCNN_OK;
}

int srcnn_comm_barrier(void* stream)
{
    CommView v;
    if (!view(v)) return comm_fail("no communicator");
    if (g_poisoned.load()) return poisoned_fail();
    if (hipSetDevice(v.device) != hipSuccess) return comm_fail("hipSetDevice failed");
    note_stream((hipStream_t)stream);
    const bool armed = watchdog().arm(v.comm, v.gen);
    const ncclResult_t r = R.AllReduce(g_token, g_token, 1, ncclFloat, ncclSum, v.comm, (hipStream_t)stream);
    if (watchdog().disarm(armed)) return comm_fail("barrier: deadline missed while queueing; communicator aborted");
    NCCL_TRY(r);
    mark_stream((hipStream_t)stream);
    return srcnn_comm_wait(stream);
}

// Host wait for everything queued on `stream` (NULL = the default stream) -- what a caller uses instead of srcnn_stream_sync
// after srcnn_comm_tiled_y_upscale2x_f32_dev / a gather: bounded by the deadline.  On a miss the communicator is aborted
// (which also ends send / recv kernels spinning for a peer that never came) and SRCNN_E_COMM is returned; every later
// srcnn_comm_* call then fails at once until the communicator is destroyed and re-created.
int srcnn_comm_wait(void* stream)
{
    CommView v;
    if (!view(v)) return comm_fail("no communicator");
    if (hipSetDevice(v.device) != hipSuccess) return comm_fail("hipSetDevice failed");
    const hipError_t e = wait_stream_deadline((hipStream_t)stream, v.comm, v.gen);
    if (e == hipErrorNotReady)
        return comm_fail("srcnn_comm_wait: the stream did not drain before the deadline (SRCNN_COMM_TIMEOUT_MS): a peer is missing or the ranks "
                         "disagree about a gather; communicator aborted");
    if (e != hipSuccess) { srcnn::set_last_error(hipGetErrorString(e)); return SRCNN_E_HIP; }
    if (g_poisoned.load()) return poisoned_fail();
    return SRCNN_OK;
}

// deadline for every wait on a peer, in milliseconds (0: none); returns the previous value
int srcnn_comm_set_timeout_ms(int ms)
{
    if (ms < 0) return comm_fail("srcnn_comm_set_timeout_ms: negative");
    return g_timeout_ms.exchange(ms);
}

}  // extern "C"
