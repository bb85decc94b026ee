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
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (R.h) break; }
    if (!R.h) {
        snprintf(g_cerr, sizeof g_cerr, "dlopen(librccl) failed: %s", dlerror());
        srcnn::set_last_error(g_cerr);
        return SRCNN_E_COMM;
    }
#define SYM(f) R.f = reinterpret_cast<decltype(R.f)>(dlsym(R.h, "nccl" #f)); if (!R.f) { snprintf(g_cerr, sizeof g_cerr, "missing nccl" #f); return SRCNN_E_COMM; }
    SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(GroupStart) SYM(GroupEnd)
    SYM(Send) SYM(Recv) SYM(AllGather) SYM(AllReduce) SYM(GetErrorString)
#undef SYM
    return 0;
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
    if (g_comm) return SRCNN_E_ARG;
    if (rank < 0 || nranks <= 0 || rank >= nranks) return SRCNN_E_ARG;
    if (int rc = load()) return rc;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    NCCL_TRY(R.CommInitRank(&g_comm, nranks, u, rank));
    g_rank = rank; g_nranks = nranks;
    if (hipMalloc((void**)&g_token, sizeof(float)) != hipSuccess) return SRCNN_E_DEVMEM;
    hipMemset(g_token, 0, sizeof(float));
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
    return SRCNN_OK;
}

int srcnn_comm_gather_f32(const float* d_send, size_t count, float* d_recv, int root, void* stream)
{
    if (!g_comm) return SRCNN_E_COMM;
    hipStream_t s = (hipStream_t)stream;
    NCCL_TRY(R.GroupStart());
    if (g_rank == root) {
        for (int r = 0; r < g_nranks; ++r) {
            if (r == root) continue;
            NCCL_TRY(R.Recv(d_recv + (size_t)r * count, count, ncclFloat, r, g_comm, s));
        }
    } else {
        NCCL_TRY(R.Send(d_send, count, ncclFloat, root, g_comm, s));
    }
    NCCL_TRY(R.GroupEnd());
    if (g_rank == root && d_recv + (size_t)root * count != d_send) {
        if (hipMemcpyAsync(d_recv + (size_t)root * count, d_send, count * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return SRCNN_E_HIP;
    }
    return SRCNN_OK;
}

int srcnn_comm_allgather_f32(const float* d_send, size_t count, float* d_recv, void* stream)
{
    if (!g_comm) return SRCNN_E_COMM;
    NCCL_TRY(R.AllGather(d_send, d_recv, count, ncclFloat, g_comm, (hipStream_t)stream));
    return SRCNN_OK;
}

int srcnn_comm_barrier(void* stream)
{
    if (!g_comm) return SRCNN_E_COMM;
    NCCL_TRY(R.AllReduce(g_token, g_token, 1, ncclFloat, ncclSum, g_comm, (hipStream_t)stream));
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SRCNN_E_HIP;
    return SRCNN_OK;
}

}  // extern "C"
