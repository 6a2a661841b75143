// comm.cpp -- RCCL communicator (1-D row partition: halo exchange of x vectors with
// the slab neighbours + scalar all-reduce of the Krylov dot products).  The reference
// has no distributed path at all; this is new.
#include "fasp_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>

#include "fasp_internal.h"

namespace fasp {
namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl       g_rccl;
ncclComm_t g_comm = nullptr;
int        g_rank = 0, g_size = 1;

int load_rccl()
{
    if (g_rccl.lib) return FASP_SUCCESS;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) {
        std::fprintf(stderr, "### ERROR: fasp_hip: cannot load RCCL: %s\n", dlerror());
        return ERROR_MISC;
    }
#define SYM(field, name)                                                              \
    *(void**)(&g_rccl.field) = dlsym(g_rccl.lib, name);                               \
    if (!g_rccl.field) {                                                              \
        std::fprintf(stderr, "### ERROR: fasp_hip: RCCL symbol %s missing\n", name);  \
        return ERROR_MISC;                                                            \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(AllGather, "ncclAllGather")
    SYM(Broadcast, "ncclBroadcast")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return FASP_SUCCESS;
}

#define NCK(expr)                                                                            \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            std::fprintf(stderr, "### ERROR: fasp_hip: %s failed: %s\n", #expr,              \
                         g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?");           \
            return ERROR_MISC;                                                               \
        }                                                                                    \
    } while (0)

}  // namespace

int comm_rank() { return g_rank; }
int comm_size() { return g_size; }

int comm_allreduce(double* dbuf, int n, unsigned maxmask, hipStream_t stream)
{
    if (g_size <= 1) return FASP_SUCCESS;
    // contiguous runs of equal reduction op
    int i = 0;
    NCK(g_rccl.GroupStart());
    while (i < n) {
        const bool mx = (maxmask >> i) & 1u;
        int j = i + 1;
        while (j < n && (((maxmask >> j) & 1u) != 0) == mx) ++j;
        NCK(g_rccl.AllReduce(dbuf + i, dbuf + i, (size_t)(j - i), ncclDouble, mx ? ncclMax : ncclSum, g_comm, stream));
        i = j;
    }
    NCK(g_rccl.GroupEnd());
    return FASP_SUCCESS;
}

int comm_exchange(const CommXfer* sends, int nsend, const CommXfer* recvs, int nrecv, hipStream_t stream)
{
    if (g_size <= 1) return FASP_SUCCESS;
    NCK(g_rccl.GroupStart());
    for (int i = 0; i < nrecv; ++i)
        if (recvs[i].count) NCK(g_rccl.Recv(recvs[i].buf, recvs[i].count, ncclDouble, recvs[i].peer, g_comm, stream));
    for (int i = 0; i < nsend; ++i)
        if (sends[i].count) NCK(g_rccl.Send(sends[i].buf, sends[i].count, ncclDouble, sends[i].peer, g_comm, stream));
    NCK(g_rccl.GroupEnd());
    return FASP_SUCCESS;
}

int comm_allgatherv(const double* sendbuf, int sendcount, double* recvbuf, const int* counts,
                    const int* displs, hipStream_t stream)
{
    if (g_size <= 1) {
        if (recvbuf + displs[0] != sendbuf)
            (void)hipMemcpyAsync(recvbuf + displs[0], sendbuf, sizeof(double) * sendcount,
                                 hipMemcpyDeviceToDevice, stream);
        return FASP_SUCCESS;
    }
    // allgatherv as a group of broadcasts (counts differ per rank)
    NCK(g_rccl.GroupStart());
    for (int r = 0; r < g_size; ++r)
        if (counts[r])
            NCK(g_rccl.Broadcast(r == g_rank ? sendbuf : recvbuf + displs[r], recvbuf + displs[r],
                                 (size_t)counts[r], ncclDouble, r, g_comm, stream));
    NCK(g_rccl.GroupEnd());
    return FASP_SUCCESS;
}

}  // namespace fasp

using namespace fasp;

extern "C" {

int fasp_hip_comm_unique_id(char* id_out)
{
    if (!id_out) return ERROR_INPUT_PAR;
    if (load_rccl() < 0) return ERROR_MISC;
    static_assert(sizeof(ncclUniqueId) == FASP_HIP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    NCK(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return FASP_SUCCESS;
}

int fasp_hip_comm_init(int rank, int nranks, const char* id_bytes)
{
    if (nranks <= 1) { g_rank = 0; g_size = 1; return FASP_SUCCESS; }
    if (!id_bytes || rank < 0 || rank >= nranks) return ERROR_INPUT_PAR;
    if (g_comm) return ERROR_INPUT_PAR;
    if (load_rccl() < 0) return ERROR_MISC;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    NCK(g_rccl.CommInitRank(&g_comm, nranks, id, rank));
    g_rank = rank;
    g_size = nranks;
    return FASP_SUCCESS;
}

int fasp_hip_comm_finalize(void)
{
    if (g_comm) {
        NCK(g_rccl.CommDestroy(g_comm));
        g_comm = nullptr;
    }
    g_rank = 0;
    g_size = 1;
    return FASP_SUCCESS;
}

int fasp_hip_comm_rank(void) { return g_rank; }
int fasp_hip_comm_size(void) { return g_size; }

}  // extern "C"
