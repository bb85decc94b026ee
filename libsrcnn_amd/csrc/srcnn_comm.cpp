// srcnn_comm.cpp -- RCCL over xGMI for the ONE exchange step of the path: gathering the output
// bands of a frame that was tiled across the GPUs of a node (the reference has no
// counterpart -- it is single-process OpenMP).  One process per GPU.  librccl is opened lazily so
// single-GPU users never load it.  The gather is peer->root point-to-point (ncclSend/ncclRecv in
// one group): on MI355X's fully connected xGMI each band lands over its own link, instead of a
// ring that would be bound by one link.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/srcnn_amd.h"

namespace srcnn { void set_last_error(const char* msg); }

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
};

Rccl R;
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_nranks = 1;
float* g_token = nullptr;
std::mutex g_mu;
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
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { if (h) break; h = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
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

extern "C" {

int srcnn_comm_unique_id(unsigned char id[SRCNN_COMM_ID_BYTES])
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = load()) return rc;
    ncclUniqueId u;
    NCCL_TRY(R.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return SRCNN_OK;
}

int srcnn_comm_init(const unsigned char id[SRCNN_COMM_ID_BYTES], int rank, int nranks)
{
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
    g_comm = comm; g_token = token;
    g_rank = rank; g_nranks = nranks;
    return SRCNN_OK;
}

int srcnn_comm_destroy(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return SRCNN_OK;
    hipDeviceSynchronize();
    R.CommDestroy(g_comm);
    g_comm = nullptr;
    hipFree(g_token); g_token = nullptr;
    g_rank = 0; g_nranks = 1;
    return SRCNN_OK;
}

int srcnn_comm_rank(int* rank, int* nranks)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return comm_fail("no communicator");
    if (rank) *rank = g_rank;
    if (nranks) *nranks = g_nranks;
    return SRCNN_OK;
}

// counts[r] floats from rank r land at d_recv + sum(counts[0..r)) on the root: bands of unequal height (an
// output height that the rank count does not divide) assemble into one contiguous frame.
int srcnn_comm_gatherv_f32(const float* d_send, const size_t* counts, float* d_recv, int root, void* stream)
{
    if (!g_comm) return comm_fail("no communicator");
    if (!counts || root < 0 || root >= g_nranks) return comm_fail("srcnn_comm_gatherv_f32: bad counts / root");
    if (counts[g_rank] && !d_send) return comm_fail("srcnn_comm_gatherv_f32: d_send == NULL");
    if (g_rank == root && !d_recv) return comm_fail("srcnn_comm_gatherv_f32: d_recv == NULL on the root");
    hipStream_t s = (hipStream_t)stream;
    // every Send/Recv of the group is attempted and GroupEnd always runs, so a failure never leaves the group open
    ncclResult_t first_bad = ncclSuccess;
    auto note = [&](ncclResult_t r) { if (r != ncclSuccess && first_bad == ncclSuccess) first_bad = r; };
    note(R.GroupStart());
    size_t my_off = 0;
    if (g_rank == root) {
        size_t off = 0;
        for (int r = 0; r < g_nranks; ++r) {
            if (r == root) my_off = off;
            else if (counts[r]) note(R.Recv(d_recv + off, counts[r], ncclFloat, r, g_comm, s));
            off += counts[r];
        }
    } else if (counts[g_rank]) {
        note(R.Send(d_send, counts[g_rank], ncclFloat, root, g_comm, s));
    }
    note(R.GroupEnd());
    if (first_bad != ncclSuccess) {
        snprintf(g_cerr, sizeof g_cerr, "band gather failed: %s", R.GetErrorString(first_bad));
        srcnn::set_last_error(g_cerr);
        return SRCNN_E_COMM;
    }
    if (g_rank == root && counts[root] && d_recv + my_off != d_send) {
        if (hipMemcpyAsync(d_recv + my_off, d_send, counts[root] * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
            srcnn::set_last_error("band gather: root's own copy failed");
            return SRCNN_E_HIP;
        }
    }
    return SRCNN_OK;
}

int srcnn_comm_gather_f32(const float* d_send, size_t count, float* d_recv, int root, void* stream)
{
    if (!g_comm) return comm_fail("no communicator");
    size_t counts[1024];
    if (g_nranks > 1024) return comm_fail("too many ranks");
    for (int r = 0; r < g_nranks; ++r) counts[r] = count;
    return srcnn_comm_gatherv_f32(d_send, counts, d_recv, root, stream);
}

int srcnn_comm_allgather_f32(const float* d_send, size_t count, float* d_recv, void* stream)
{
    if (!g_comm) return comm_fail("no communicator");
    NCCL_TRY(R.AllGather(d_send, d_recv, count, ncclFloat, g_comm, (hipStream_t)stream));
    return SRCNN_OK;
}

int srcnn_comm_barrier(void* stream)
{
    if (!g_comm) return comm_fail("no communicator");
    NCCL_TRY(R.AllReduce(g_token, g_token, 1, ncclFloat, ncclSum, g_comm, (hipStream_t)stream));
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SRCNN_E_HIP;
    return SRCNN_OK;
}

}  // extern "C"
