// rccl_double.cpp -- TEST INFRASTRUCTURE.  A stand-in for librccl that lets N processes which share ONE physical GPU run the
// library's multi-rank code (csrc/srcnn_comm.cpp) for real: the installed RCCL refuses two ranks on one device ("Duplicate GPU
// detected"), and the development pool has one GPU per box.  Loaded by the product through its one hook, SRCNN_RCCL_LIB.
// Never shipped, never linked into the product; built into tests/rccl_double/_build/ by tests/rccl_double/__init__.py.
//
// It exports exactly the eleven symbols srcnn_comm.cpp resolves (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclAllGather, ncclAllReduce, ncclGetErrorString, ncclCommAbort) with
// RCCL's semantics as far as the product can observe them:
//   * calls are ASYNCHRONOUS with respect to the host: GroupEnd / a collective returns once the work is queued;
//   * the work is ordered on the caller's STREAM: it starts after what the stream held before (an event), and the stream does
//     not proceed past it until it is done -- a device-side wait (hipStreamWaitValue32 on a flag the helper thread sets), which
//     is how a send / receive kernel that spins for its peer looks from outside: hipStreamQuery says "not ready" until the
//     peer has shown up;
//   * sends and receives between a pair of ranks match in issue order; a count mismatch is an error on both sides;
//   * ncclCommAbort from another thread ends everything that is pending (flags are released so that the streams drain) and
//     frees the communicator;
//   * ncclCommInitRank blocks until all ranks have joined.
// Rendezvous: one POSIX shared-memory segment per communicator, named after the unique id, unlinked as soon as every rank has
// mapped it.  Data: the receiver maps the sender's buffer with hipIpcOpenMemHandle (the ranks share the device, so this is an
// ordinary device-to-device copy) and acknowledges; small reductions go through the segment.
// If the device cannot do hipStreamWaitValue32, RCCL_DOUBLE_BLOCKING=1 semantics are used: the call synchronises the stream and
// does the work on the calling thread (still correct, no overlap).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 16, kRing = 32, kFlags = 64;
constexpr size_t kCollBytes = 4096;
constexpr unsigned kMagic = 0x5CC1D0B1u;

struct Msg {                              // one send posted by `src` for `dst`
    std::atomic<unsigned> state;          // 0 free, 1 posted, 2 consumed, 3 refused (size mismatch)
    unsigned long long seq;
    hipIpcMemHandle_t handle;             // of the allocation the send buffer lies in
    unsigned long long offset, nbytes;
};
struct Shared {
    std::atomic<unsigned> magic, nranks, arrived, mapped;
    std::atomic<unsigned> bar_count, bar_gen;          // sense-reversing barrier of the small collectives
    Msg ring[kMaxRanks][kMaxRanks][kRing];             // [src][dst][seq % kRing]
    unsigned char coll[kMaxRanks][kCollBytes];
};

enum Kind { SEND, RECV, ALLREDUCE, COPY };
struct Op {
    Kind kind;
    const void* send; void* recv;
    size_t nbytes; int peer;
    ncclDataType_t dt; ncclRedOp_t op; size_t count;
};
struct Batch { std::vector<Op> ops; hipEvent_t ready = nullptr; volatile unsigned* flag = nullptr; };

const char* g_err = "";
thread_local int t_depth = 0;                          // ncclGroupStart nesting
thread_local std::vector<Op> t_ops;                    // ops collected between ncclGroupStart and ncclGroupEnd
thread_local struct ncclComm* t_group_comm = nullptr;
thread_local hipStream_t t_group_stream = nullptr;

size_t dt_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

}  // namespace

struct ncclComm {
    int rank = 0, nranks = 1, device = 0;
    Shared* sh = nullptr;
    bool blocking = false;
    std::atomic<bool> aborted{false};
    std::atomic<int> error{0};
    unsigned long long send_seq[kMaxRanks] = {}, recv_seq[kMaxRanks] = {};
    std::map<std::string, void*> mapped;               // peer allocations opened so far (key = the 64 handle bytes)
    hipStream_t copy = nullptr;
    unsigned* flags = nullptr;                          // kFlags words of host-coherent memory the streams wait on
    unsigned next_flag = 0;
    std::vector<hipEvent_t> events;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Batch> queue;
    bool stop = false;
    unsigned inflight = 0;

    // the stand-in's own safety net (RCCL_DOUBLE_TIMEOUT_S, default 150 s per batch): a test that has lost a rank must not
    // occupy the GPU box until the pool's limit kills it.  The product's deadlines are far shorter and fire first.
    std::chrono::steady_clock::time_point batch_deadline;
    bool nap(int& n)                                    // false: aborted (or the safety net fired)
    {
        if (aborted.load(std::memory_order_relaxed)) return false;
        if (++n > 200) {
            std::this_thread::sleep_for(std::chrono::microseconds(n < 5000 ? 20 : 200));
            if ((n & 1023) == 0 && std::chrono::steady_clock::now() > batch_deadline) {
                fprintf(stderr, "[rccl double] rank %d: a batch did not complete within its safety limit; giving up\n", rank);
                g_err = "stand-in safety limit"; error = 9;
                return false;
            }
        }
        return true;
    }
    bool barrier()                                      // all ranks (small collectives); false: aborted
    {
        const unsigned gen = sh->bar_gen.load();
        if (sh->bar_count.fetch_add(1) + 1 == (unsigned)nranks) { sh->bar_count.store(0); sh->bar_gen.fetch_add(1); return true; }
        int n = 0;
        while (sh->bar_gen.load() == gen) if (!nap(n)) return false;
        return true;
    }
    void* map_peer(const hipIpcMemHandle_t& h)
    {
        const std::string key(reinterpret_cast<const char*>(&h), sizeof h);
        auto it = mapped.find(key);
        if (it != mapped.end()) return it->second;
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        mapped[key] = p;
        return p;
    }
    bool run(Batch& b);
    void loop();
};

namespace {

template <class T>
void reduce_into(T* acc, const T* x, size_t n, ncclRedOp_t op)
{
    for (size_t i = 0; i < n; ++i) {
        switch (op) {
        case ncclSum: acc[i] = acc[i] + x[i]; break;
        case ncclProd: acc[i] = acc[i] * x[i]; break;
        case ncclMin: acc[i] = x[i] < acc[i] ? x[i] : acc[i]; break;
        case ncclMax: acc[i] = x[i] > acc[i] ? x[i] : acc[i]; break;
        default: break;
        }
    }
}

}  // namespace

// One batch = what one ncclGroupEnd (or one collective call) queued.  Runs on the helper thread (or, blocking mode, on the caller).
bool ncclComm::run(Batch& b)
{
    int n = 0;
    static const int limit_s = [] { const char* e = getenv("RCCL_DOUBLE_TIMEOUT_S"); return e ? atoi(e) : 150; }();
    batch_deadline = std::chrono::steady_clock::now() + std::chrono::seconds(limit_s);
    const bool verbose = getenv("RCCL_DOUBLE_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "[rccl double] rank %d: batch of %zu ops starts\n", rank, b.ops.size());
    if (b.ready) {                                      // the stream has reached this point: send buffers are final, receive buffers free
        for (;;) {
            const hipError_t e = hipEventQuery(b.ready);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) { (void)hipGetLastError(); error = 1; return false; }
            (void)hipGetLastError();
            if (!nap(n)) return false;
        }
    }
    // 1. post every send
    std::vector<Msg*> sent;
    for (const Op& o : b.ops) {
        if (o.kind != SEND) continue;
        Msg& m = sh->ring[rank][o.peer][send_seq[o.peer] % kRing];
        n = 0;
        while (m.state.load(std::memory_order_acquire) != 0) if (!nap(n)) return false;
        void* base = nullptr; size_t span = 0;
        if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &span, const_cast<void*>(o.send)) != hipSuccess ||
            hipIpcGetMemHandle(&m.handle, base) != hipSuccess) { (void)hipGetLastError(); error = 2; return false; }
        m.offset = (unsigned long long)((const char*)o.send - (const char*)base);
        m.nbytes = o.nbytes;
        m.seq = send_seq[o.peer]++;
        m.state.store(1, std::memory_order_release);
        sent.push_back(&m);
    }
    // 2. receives, in whatever order the peers show up
    std::vector<const Op*> recvs;
    std::vector<Msg*> slots;
    for (const Op& o : b.ops)
        if (o.kind == RECV) { recvs.push_back(&o); slots.push_back(&sh->ring[o.peer][rank][recv_seq[o.peer]++ % kRing]); }
    std::vector<bool> done(recvs.size(), false);
    size_t left = recvs.size();
    n = 0;
    while (left) {
        bool progress = false;
        for (size_t i = 0; i < recvs.size(); ++i) {
            if (done[i] || slots[i]->state.load(std::memory_order_acquire) != 1) continue;
            Msg& m = *slots[i];
            if (m.nbytes != recvs[i]->nbytes) { m.state.store(3, std::memory_order_release); error = 3; g_err = "send / receive size mismatch"; return false; }
            char* src = static_cast<char*>(map_peer(m.handle));
            if (!src || hipMemcpyAsync(recvs[i]->recv, src + m.offset, m.nbytes, hipMemcpyDeviceToDevice, copy) != hipSuccess) { (void)hipGetLastError(); error = 4; return false; }
            done[i] = true; --left; progress = true;
        }
        if (!progress && !nap(n)) return false;
    }
    // 3. local copies and small all-reduces
    for (const Op& o : b.ops) {
        if (o.kind == COPY) {
            if (o.recv != o.send && hipMemcpyAsync(o.recv, o.send, o.nbytes, hipMemcpyDeviceToDevice, copy) != hipSuccess) { error = 4; return false; }
        } else if (o.kind == ALLREDUCE) {
            if (o.nbytes > kCollBytes) { error = 5; g_err = "all-reduce larger than the stand-in supports"; return false; }
            // (never the synchronous hipMemcpy: it orders itself behind the NULL stream, which may be the very stream that waits
            //  for this batch's flag)
            if (hipMemcpyAsync(sh->coll[rank], o.send, o.nbytes, hipMemcpyDeviceToHost, copy) != hipSuccess ||
                hipStreamSynchronize(copy) != hipSuccess) { error = 4; return false; }
            if (!barrier()) return false;
            unsigned char acc[kCollBytes];
            memcpy(acc, sh->coll[0], o.nbytes);
            for (int r = 1; r < nranks; ++r) {
                switch (o.dt) {
                case ncclFloat32: reduce_into(reinterpret_cast<float*>(acc), reinterpret_cast<const float*>(sh->coll[r]), o.count, o.op); break;
                case ncclUint64: reduce_into(reinterpret_cast<unsigned long long*>(acc), reinterpret_cast<const unsigned long long*>(sh->coll[r]), o.count, o.op); break;
                case ncclInt32: reduce_into(reinterpret_cast<int*>(acc), reinterpret_cast<const int*>(sh->coll[r]), o.count, o.op); break;
                default: error = 5; g_err = "all-reduce type the stand-in does not support"; return false;
                }
            }
            if (!barrier()) return false;               // nobody overwrites its slot before everybody has read it
            if (hipMemcpyAsync(o.recv, acc, o.nbytes, hipMemcpyHostToDevice, copy) != hipSuccess ||
                hipStreamSynchronize(copy) != hipSuccess) { error = 4; return false; }
        }
    }
    if (hipStreamSynchronize(copy) != hipSuccess) { error = 4; return false; }
    for (Msg* m : slots) m->state.store(2, std::memory_order_release);          // the senders may reuse their buffers
    // 4. my sends have been consumed
    for (Msg* m : sent) {
        n = 0;
        for (;;) {
            const unsigned st = m->state.load(std::memory_order_acquire);
            if (st == 2) break;
            if (st == 3) { m->state.store(0); error = 3; g_err = "send / receive size mismatch"; return false; }
            if (!nap(n)) return false;
        }
        m->state.store(0, std::memory_order_release);
    }
    if (verbose) fprintf(stderr, "[rccl double] rank %d: batch done\n", rank);
    return true;
}

void ncclComm::loop()
{
    (void)hipSetDevice(device);
    for (;;) {
        Batch b;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop || !queue.empty(); });
            if (queue.empty()) return;
            b = std::move(queue.front());
            queue.pop_front();
        }
        if (!aborted.load()) (void)run(b);
        if (b.flag) *b.flag = 1u;                       // the stream proceeds (also after an abort: the device must drain)
        __sync_synchronize();
        { std::lock_guard<std::mutex> lk(mu); --inflight; }
        cv.notify_all();
    }
}

namespace {

ncclResult_t submit(ncclComm* c, std::vector<Op>&& ops, hipStream_t s)
{
    if (c->aborted.load()) return ncclInvalidUsage;
    if (c->error.load()) return ncclInternalError;
    if (ops.empty()) return ncclSuccess;
    Batch b;
    b.ops = std::move(ops);
    if (c->blocking) {
        if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
        return c->run(b) ? ncclSuccess : (c->aborted.load() ? ncclInvalidUsage : ncclInternalError);
    }
    {
        // one flag (and one event) per batch in flight; wait for the oldest if all are taken
        std::unique_lock<std::mutex> lk(c->mu);
        int n = 0;
        while (c->inflight >= kFlags) { lk.unlock(); if (!c->nap(n)) return ncclInvalidUsage; lk.lock(); }
        const unsigned k = c->next_flag++ % kFlags;
        b.flag = c->flags + k;
        b.ready = c->events[k];
        *b.flag = 0u;
        ++c->inflight;
    }
    __sync_synchronize();
    if (hipEventRecord(b.ready, s) != hipSuccess ||
        hipStreamWaitValue32(s, const_cast<unsigned*>(b.flag), 1u, hipStreamWaitValueEq, 0xffffffffu) != hipSuccess) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(c->mu);
        --c->inflight;
        return ncclUnhandledCudaError;
    }
    { std::lock_guard<std::mutex> lk(c->mu); c->queue.push_back(std::move(b)); }
    c->cv.notify_all();
    return ncclSuccess;
}

ncclResult_t add(ncclComm_t comm, Op o, hipStream_t s)
{
    if (!comm) return ncclInvalidArgument;
    if (t_depth > 0) {
        if (t_group_comm && (t_group_comm != comm || t_group_stream != s) && !t_ops.empty()) return ncclInvalidUsage;   // one comm, one stream per group here
        t_group_comm = comm; t_group_stream = s;
        t_ops.push_back(o);
        return ncclSuccess;
    }
    std::vector<Op> ops{o};
    return submit(comm, std::move(ops), s);
}

void teardown(ncclComm* c, bool aborting)
{
    if (aborting) c->aborted = true;
    {
        std::unique_lock<std::mutex> lk(c->mu);
        if (!aborting) c->cv.wait(lk, [&] { return c->inflight == 0; });     // destroy drains; abort lets the loop release what is queued
        c->stop = true;
    }
    c->cv.notify_all();
    if (c->worker.joinable()) c->worker.join();
    if (aborting) {
        // Whatever still waits on a flag passes (a call of the owner thread may have queued its wait without reaching the
        // queue).  Nothing is freed: that thread may still be inside this library with the pointer -- RCCL's own abort has
        // the same contract problem and the product never touches an aborted communicator again.
        if (c->flags) for (int k = 0; k < kFlags; ++k) c->flags[k] = 1u;
        __sync_synchronize();
        return;
    }
    (void)hipSetDevice(c->device);
    for (auto& kv : c->mapped) (void)hipIpcCloseMemHandle(kv.second);
    for (auto e : c->events) (void)hipEventDestroy(e);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    if (c->flags) (void)hipHostFree(c->flags);
    if (c->sh) munmap(c->sh, sizeof(Shared));
    delete c;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(id->internal, 1, 16, f) != 16) { if (f) fclose(f); return ncclSystemError; }
    fclose(f);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    char name[64] = "/srcnn_rccl_double_";
    for (int i = 0; i < 12; ++i) snprintf(name + strlen(name), 3, "%02x", (unsigned char)id.internal[i]);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { if (fd >= 0) close(fd); g_err = "shm_open failed"; return ncclSystemError; }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    auto* c = new ncclComm;
    c->sh = static_cast<Shared*>(p);
    c->rank = rank; c->nranks = nranks;
    (void)hipGetDevice(&c->device);
    unsigned expect = 0;
    c->sh->nranks.compare_exchange_strong(expect, (unsigned)nranks);
    c->sh->magic.store(kMagic);
    c->sh->arrived.fetch_add(1);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    while (c->sh->arrived.load() < (unsigned)nranks) {                            // RCCL's init is a rendezvous too
        if (std::chrono::steady_clock::now() > deadline) { shm_unlink(name); munmap(p, sizeof(Shared)); delete c; g_err = "rendezvous timed out"; return ncclSystemError; }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if (c->sh->nranks.load() != (unsigned)nranks) { munmap(p, sizeof(Shared)); delete c; g_err = "ranks disagree about nranks"; return ncclInvalidArgument; }
    if (c->sh->mapped.fetch_add(1) + 1 == (unsigned)nranks) shm_unlink(name);      // everybody has it mapped: the name can go
    int can = 0;
    const char* force = getenv("RCCL_DOUBLE_BLOCKING");
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device);
    c->blocking = (force && atoi(force) != 0) || !can;
    if (hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking) != hipSuccess) { teardown(c, false); return ncclUnhandledCudaError; }
    if (!c->blocking) {
        void* fl = nullptr;
        if (hipHostMalloc(&fl, sizeof(unsigned) * kFlags, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { teardown(c, false); return ncclUnhandledCudaError; }
        c->flags = static_cast<unsigned*>(fl);
        memset(c->flags, 0, sizeof(unsigned) * kFlags);
        for (int i = 0; i < kFlags; ++i) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { teardown(c, false); return ncclUnhandledCudaError; }
            c->events.push_back(e);
        }
        c->worker = std::thread([c] { c->loop(); });
    }
    if (getenv("RCCL_DOUBLE_VERBOSE")) fprintf(stderr, "[rccl double] rank %d/%d on device %d, %s\n", rank, nranks, c->device, c->blocking ? "blocking" : "stream-ordered");
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { if (comm) teardown(comm, false); return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t comm) { if (comm) teardown(comm, true); return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    ncclComm* c = t_group_comm;
    hipStream_t s = t_group_stream;
    std::vector<Op> ops;
    ops.swap(t_ops);
    t_group_comm = nullptr; t_group_stream = nullptr;
    if (!c) return ncclSuccess;
    return submit(c, std::move(ops), s);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
    if (!comm || peer < 0 || peer >= comm->nranks || peer == comm->rank || !dt_size(dt)) return ncclInvalidArgument;
    return add(comm, Op{SEND, buf, nullptr, count * dt_size(dt), peer, dt, ncclSum, count}, s);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
    if (!comm || peer < 0 || peer >= comm->nranks || peer == comm->rank || !dt_size(dt)) return ncclInvalidArgument;
    return add(comm, Op{RECV, nullptr, buf, count * dt_size(dt), peer, dt, ncclSum, count}, s);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t s)
{
    if (!comm || !dt_size(dt)) return ncclInvalidArgument;
    return add(comm, Op{ALLREDUCE, send, recv, count * dt_size(dt), -1, dt, op, count}, s);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t s)
{
    if (!comm || !dt_size(dt)) return ncclInvalidArgument;
    const size_t nb = count * dt_size(dt);
    std::vector<Op> ops;
    for (int r = 0; r < comm->nranks; ++r) {
        if (r == comm->rank) { ops.push_back(Op{COPY, send, static_cast<char*>(recv) + (size_t)r * nb, nb, r, dt, ncclSum, count}); continue; }
        ops.push_back(Op{SEND, send, nullptr, nb, r, dt, ncclSum, count});
        ops.push_back(Op{RECV, nullptr, static_cast<char*>(recv) + (size_t)r * nb, nb, r, dt, ncclSum, count});
    }
    if (t_depth > 0) { for (auto& o : ops) { ncclResult_t rc = add(comm, o, s); if (rc != ncclSuccess) return rc; } return ncclSuccess; }
    return submit(comm, std::move(ops), s);
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "[rccl double] unhandled HIP error";
    case ncclSystemError: return *g_err ? g_err : "[rccl double] system error";
    case ncclInternalError: return *g_err ? g_err : "[rccl double] internal error";
    case ncclInvalidArgument: return *g_err ? g_err : "[rccl double] invalid argument";
    case ncclInvalidUsage: return "[rccl double] invalid usage (aborted communicator?)";
    default: return "[rccl double] error";
    }
}

}  // extern "C"
