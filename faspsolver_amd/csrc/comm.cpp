// comm.cpp -- communicator of the multi-GPU path (1-D row partition: halo exchange of
// vector entries with the slab neighbours, all-gather at the distributed -> replicated
// level boundary, scalar all-reduce of the Krylov dot products).  The reference has no
// distributed path at all; this is new.
//
// Three transports behind one interface:
//   RCCL  (production): one process per GPU, grouped ncclSend/ncclRecv + ncclAllReduce on
//         the compute stream, peer-to-peer over xGMI.  RCCL is dlopen'ed on first use so
//         the single-GPU path does not depend on it.
//   IPC   (round 4, opt-in: fasp_hip_comm_init_ipc / BENCH_COMM=ipc): peer windows -- every rank maps every peer's uncached
//         device window (hipIpcGetMemHandle / hipIpcOpenMemHandle; over xGMI between the GPUs of a node, and between
//         processes that share one GPU: how it is validated here), a halo exchange is ONE kernel that stores the boundary
//         entries straight into the neighbours' mailboxes, announces them by a sequence word and polls / copies its own
//         (comm_ipc.h, comm_ipc.hip); all-reduces of the Krylov scalars likewise, summed in rank order.  No RCCL call.
//   SHM   (validation): host-staged through a POSIX shared-memory segment.  It lets the
//         whole distributed solver (partition, halo plans, replicated levels, replicated
//         host control flow) be exercised by several processes that share ONE GPU, which
//         is all the development box has.  Reductions are summed in rank order, so every
//         rank obtains bit-identical scalars, exactly like an RCCL all-reduce.
#include "fasp_comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "comm_ipc.h"
#include "fasp_internal.h"

namespace fasp {
namespace {

enum Backend { NONE = 0, RCCL = 1, SHM = 2, IPC = 3 };
Backend g_backend = NONE;
int     g_rank = 0, g_size = 1;

// ------------------------------------------------------------------ RCCL
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl       g_rccl;
ncclComm_t g_comm = nullptr;

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

#define HCK(expr)                                                                            \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "### ERROR: fasp_hip: %s failed: %s\n", #expr, hipGetErrorString(e_)); \
            return ERROR_MISC;                                                               \
        }                                                                                    \
    } while (0)

// ------------------------------------------------------------------ SHM
constexpr size_t SHM_BOX_BYTES = 64u << 20;  // mailbox per rank
struct ShmHeader {
    std::atomic<int> arrive;
    std::atomic<int> generation;
    std::atomic<int> error;   // set by any rank that fails inside a collective; every rank sees it after the barrier
    int              nranks;
};
struct ShmBoxHeader {
    long long off[64];  // per destination: offset (in doubles) into this box's payload
    long long cnt[64];
};
std::string g_shm_name;
char*       g_shm_base = nullptr;
size_t      g_shm_bytes = 0;
double*     g_stage = nullptr;  // pinned host staging (SHM_BOX_BYTES)

ShmHeader*    shm_hdr() { return reinterpret_cast<ShmHeader*>(g_shm_base); }
char*         shm_box(int r) { return g_shm_base + 4096 + (size_t)r * SHM_BOX_BYTES; }
ShmBoxHeader* box_hdr(int r) { return reinterpret_cast<ShmBoxHeader*>(shm_box(r)); }
double*       box_data(int r) { return reinterpret_cast<double*>(shm_box(r) + sizeof(ShmBoxHeader)); }
constexpr size_t box_cap() { return (SHM_BOX_BYTES - sizeof(ShmBoxHeader)) / sizeof(double); }

// Every rank enters every barrier of every collective (a failing rank raises the error flag FIRST and still
// arrives), so no rank is ever left spinning on a peer that returned early; the wait itself is bounded.
bool g_comm_failed = false;   // sticky: a collective failed on this rank or on a peer
int  shm_timeout_s()
{
    static int t = -1;
    if (t < 0) { const char* e = std::getenv("FASP_HIP_SHM_TIMEOUT_S"); t = e ? std::atoi(e) : 300; if (t <= 0) t = 300; }
    return t;
}
// Returns 0, SHM_PEER_ERROR (a rank raised the error flag before this barrier released) or SHM_TIMEOUT (a peer never
// arrived: the caller must NOT enter another barrier of this collective -- it has already counted itself in and the
// generation it would wait for can never come).  The error word holds the generation (+ 1) in which it was raised, so
// a barrier only reports errors raised up to its own release: a fast peer that fails in the NEXT collective does not
// make a slow rank fail the previous one (and then skip the next).
constexpr int SHM_PEER_ERROR = -1, SHM_TIMEOUT = -2;
int shm_barrier()
{
    ShmHeader* h = shm_hdr();
    const int gen = h->generation.load(std::memory_order_acquire);
    if (h->arrive.fetch_add(1, std::memory_order_acq_rel) == h->nranks - 1) {
        h->arrive.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1, std::memory_order_release);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        long long polls = 0;
        while (h->generation.load(std::memory_order_acquire) == gen) {
            usleep(20);
            if ((++polls & 1023) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > (double)shm_timeout_s()) {
                std::fprintf(stderr, "### ERROR: fasp_hip: rank %d waited %d s at a shared-memory barrier: a peer is gone\n",
                             g_rank, shm_timeout_s());
                int none = 0;
                h->error.compare_exchange_strong(none, gen + 1, std::memory_order_acq_rel);
                g_comm_failed = true;
                return SHM_TIMEOUT;
            }
        }
    }
    const int err = h->error.load(std::memory_order_acquire);
    if (err != 0 && err <= gen + 1) { g_comm_failed = true; return SHM_PEER_ERROR; }
    return FASP_SUCCESS;
}
void shm_raise_error()   // (raised before the collective's first barrier releases: stamped with the current generation)
{
    ShmHeader* h = shm_hdr();
    int none = 0;
    h->error.compare_exchange_strong(none, h->generation.load(std::memory_order_acquire) + 1, std::memory_order_acq_rel);
    g_comm_failed = true;
}


// ------------------------------------------------------------------ IPC (peer windows)
struct IpcState {
    char*     win[IPC_MAX_RANKS] = {};       // win[me]: this rank's window; win[q]: peer q's, mapped
    size_t    cap = 0;                       // doubles per mailbox
    size_t    bytes = 0;
    unsigned* counters = nullptr;            // device: 4 pairs of arrival counters (one pair per stream seen)
    unsigned* err = nullptr;                 // pinned host word
    hipStream_t streams[4] = {};
    int       nstreams = 0;
    unsigned long long seq_to[IPC_MAX_RANKS] = {}, seq_from[IPC_MAX_RANKS] = {}, red_epoch = 0;
};
IpcState g_ipc;
size_t ipc_mbox_stride() { return g_ipc.cap * sizeof(double) + 256; }
size_t ipc_mbox_off(int from, int par) { return ((size_t)from * 2 + (size_t)par) * ipc_mbox_stride(); }
size_t ipc_ack_off(int from) { return (size_t)IPC_MAX_RANKS * 2 * ipc_mbox_stride() + (size_t)from * 256; }
size_t ipc_red_off(int from, int par) { return ipc_ack_off(IPC_MAX_RANKS) + ((size_t)from * 2 + (size_t)par) * 512; }
size_t ipc_window_bytes() { return ipc_red_off(IPC_MAX_RANKS, 0); }
unsigned* ipc_counters_for(hipStream_t s)
{
    for (int i = 0; i < g_ipc.nstreams; ++i) if (g_ipc.streams[i] == s) return g_ipc.counters + 2 * i;
    if (g_ipc.nstreams < 4) { g_ipc.streams[g_ipc.nstreams] = s; return g_ipc.counters + 2 * g_ipc.nstreams++; }
    return nullptr;   // (more streams than the solver has: an error -- sharing a pair of last-block counters between streams is a race)
}
// one exchange kernel: my messages into the peers' mailboxes, theirs out of mine (every message at most a mailbox long)
int ipc_exchange_once(const CommXfer* sends, int nsend, const CommXfer* recvs, int nrecv, hipStream_t stream)
{
    IpcXchgArgs a{};
    long long total = 0;
    for (int i = 0; i < nsend; ++i) {
        if (!sends[i].count) continue;
        const int q = sends[i].peer;
        if (sends[i].count > g_ipc.cap || a.ns >= IPC_MAX_RANKS) { std::fprintf(stderr, "### ERROR: fasp_hip: message of %zu doubles exceeds the peer-window mailbox (%zu: FASP_HIP_IPC_CAP)\n", sends[i].count, g_ipc.cap); return ERROR_MISC; }
        const unsigned long long seq = ++g_ipc.seq_to[q];
        IpcSend& S = a.s[a.ns++];
        char* box = g_ipc.win[q] + ipc_mbox_off(g_rank, (int)(seq & 1));
        S.remote_data = reinterpret_cast<double*>(box);
        S.remote_flag = reinterpret_cast<unsigned long long*>(box + g_ipc.cap * sizeof(double));
        S.ack_in = reinterpret_cast<const unsigned long long*>(g_ipc.win[g_rank] + ipc_ack_off(q));
        S.src = sends[i].buf; S.n = (long long)sends[i].count; S.seq = seq;
        total += S.n;
    }
    for (int i = 0; i < nrecv; ++i) {
        if (!recvs[i].count) continue;
        const int q = recvs[i].peer;
        if (recvs[i].count > g_ipc.cap || a.nr >= IPC_MAX_RANKS) return ERROR_MISC;
        const unsigned long long seq = ++g_ipc.seq_from[q];
        IpcRecv& R = a.r[a.nr++];
        const char* box = g_ipc.win[g_rank] + ipc_mbox_off(q, (int)(seq & 1));
        R.local_data = reinterpret_cast<const double*>(box);
        R.local_flag = reinterpret_cast<const unsigned long long*>(box + g_ipc.cap * sizeof(double));
        R.remote_ack = reinterpret_cast<unsigned long long*>(g_ipc.win[q] + ipc_ack_off(g_rank));
        R.dst = recvs[i].buf; R.n = (long long)recvs[i].count; R.seq = seq;
        total += R.n;
    }
    if (a.ns == 0 && a.nr == 0) return FASP_SUCCESS;
    a.counters = ipc_counters_for(stream);
    if (!a.counters) { std::fprintf(stderr, "### ERROR: fasp_hip: peer-window exchanges issued on more than four streams\n"); return ERROR_MISC; }
    a.err = g_ipc.err;
    return ipc_xchg_launch(a, total, stream) < 0 ? ERROR_MISC : FASP_SUCCESS;
}
// Messages longer than a mailbox (FASP_HIP_IPC_CAP doubles: a halo plane of more than 4 MB) go in pieces, as the all-gather's blocks
// do: sender and receiver of a pair know the same count, so both cut it the same way and number the pieces alike.
int ipc_exchange(const CommXfer* sends, int nsend, const CommXfer* recvs, int nrecv, hipStream_t stream)
{
    size_t longest = 0;
    for (int i = 0; i < nsend; ++i) longest = std::max(longest, sends[i].count);
    for (int i = 0; i < nrecv; ++i) longest = std::max(longest, recvs[i].count);
    if (longest <= g_ipc.cap) return ipc_exchange_once(sends, nsend, recvs, nrecv, stream);
    std::vector<CommXfer> s((size_t)nsend), r((size_t)nrecv);
    for (size_t off = 0; off < longest; off += g_ipc.cap) {
        for (int i = 0; i < nsend; ++i) { s[(size_t)i] = sends[i]; s[(size_t)i].buf = sends[i].buf + std::min(off, sends[i].count); s[(size_t)i].count = sends[i].count > off ? std::min(g_ipc.cap, sends[i].count - off) : 0; }
        for (int i = 0; i < nrecv; ++i) { r[(size_t)i] = recvs[i]; r[(size_t)i].buf = recvs[i].buf + std::min(off, recvs[i].count); r[(size_t)i].count = recvs[i].count > off ? std::min(g_ipc.cap, recvs[i].count - off) : 0; }
        const int st = ipc_exchange_once(s.data(), nsend, r.data(), nrecv, stream);
        if (st < 0) return st;
    }
    return FASP_SUCCESS;
}
}  // namespace

int  comm_rank() { return g_rank; }
int  comm_size() { return g_size; }
bool comm_failed() { return g_comm_failed || (g_backend == IPC && g_ipc.err && *g_ipc.err != 0u); }
static bool g_ipc_shared_device = false;
bool comm_is_peer_window() { return g_backend == IPC; }
bool comm_shares_devices() { return g_backend == SHM || (g_backend == IPC && g_ipc_shared_device); }
void comm_mark_failed() { g_comm_failed = true; }

static int comm_allreduce_impl(double* dbuf, int n, unsigned maxmask, hipStream_t stream)
{
    if (g_size <= 1) return FASP_SUCCESS;
    if (g_backend == SHM) {
        bool ok = n >= 0 && (size_t)n <= box_cap();
        if (ok) ok = hipMemcpyAsync(g_stage, dbuf, sizeof(double) * n, hipMemcpyDeviceToHost, stream) == hipSuccess &&
                     hipStreamSynchronize(stream) == hipSuccess;
        if (ok) std::memcpy(box_data(g_rank), g_stage, sizeof(double) * n);
        else shm_raise_error();
        { const int bs = shm_barrier(); if (bs < 0) { if (bs != SHM_TIMEOUT) (void)shm_barrier(); return ERROR_MISC; } }   // a peer's error: still arrive at the second barrier
        for (int i = 0; i < n; ++i) {
            const bool mx = (maxmask >> i) & 1u;
            double v = box_data(0)[i];  // seeded with rank 0's value: a true maximum also for negative entries, like ncclMax
            for (int r = 1; r < g_size; ++r) {  // rank order: identical result on every rank
                const double x = box_data(r)[i];
                v = mx ? (x > v ? x : v) : v + x;
            }
            g_stage[i] = v;
        }
        if (shm_barrier() < 0) return ERROR_MISC;
        HCK(hipMemcpyAsync(dbuf, g_stage, sizeof(double) * n, hipMemcpyHostToDevice, stream));
        HCK(hipStreamSynchronize(stream));
        return FASP_SUCCESS;
    }
    if (g_backend == IPC) {
        if (n < 0 || n > IPC_RED_MAX) return ERROR_INPUT_PAR;
        IpcRedArgs a{};
        const unsigned long long ep = ++g_ipc.red_epoch;
        const int par = (int)(ep & 1);
        for (int q = 0; q < g_size; ++q) {
            char* theirs = g_ipc.win[q] + ipc_red_off(g_rank, par);
            const char* mine = g_ipc.win[g_rank] + ipc_red_off(q, par);
            a.remote_val[q] = reinterpret_cast<double*>(theirs); a.remote_flag[q] = reinterpret_cast<unsigned long long*>(theirs + 256);
            a.local_val[q] = reinterpret_cast<const double*>(mine); a.local_flag[q] = reinterpret_cast<const unsigned long long*>(mine + 256);
        }
        a.dbuf = dbuf; a.n = n; a.me = g_rank; a.nranks = g_size; a.maxmask = maxmask; a.epoch = ep; a.err = g_ipc.err;
        return ipc_allreduce_launch(a, stream) < 0 ? ERROR_MISC : FASP_SUCCESS;
    }
    int i = 0;
    NCK(g_rccl.GroupStart());
    while (i < n) {  // contiguous runs of equal reduction op
        const bool mx = (maxmask >> i) & 1u;
        int j = i + 1;
        while (j < n && (((maxmask >> j) & 1u) != 0) == mx) ++j;
        NCK(g_rccl.AllReduce(dbuf + i, dbuf + i, (size_t)(j - i), ncclDouble, mx ? ncclMax : ncclSum, g_comm, stream));
        i = j;
    }
    NCK(g_rccl.GroupEnd());
    return FASP_SUCCESS;
}

static int comm_exchange_impl(const CommXfer* sends, int nsend, const CommXfer* recvs, int nrecv, hipStream_t stream)
{
    if (g_size <= 1) return FASP_SUCCESS;
    if (g_backend == SHM) {
        ShmBoxHeader* bh = box_hdr(g_rank);
        size_t off = 0;
        bool   ok = true;
        for (int q = 0; q < g_size; ++q) { bh->off[q] = 0; bh->cnt[q] = 0; }
        for (int i = 0; i < nsend && ok; ++i) {
            if (off + sends[i].count > box_cap()) {
                std::fprintf(stderr, "### ERROR: fasp_hip: shm mailbox too small\n");
                ok = false;
                break;
            }
            ok = hipMemcpyAsync(g_stage + off, sends[i].buf, sizeof(double) * sends[i].count, hipMemcpyDeviceToHost, stream) == hipSuccess;
            bh->off[sends[i].peer] = (long long)off;
            bh->cnt[sends[i].peer] = (long long)sends[i].count;
            off += sends[i].count;
        }
        if (ok) ok = hipStreamSynchronize(stream) == hipSuccess;
        if (ok) std::memcpy(box_data(g_rank), g_stage, sizeof(double) * off);
        else shm_raise_error();
        { const int bs = shm_barrier(); if (bs < 0) { if (bs != SHM_TIMEOUT) (void)shm_barrier(); return ERROR_MISC; } }   // a peer's error: still arrive at the second barrier
        size_t roff = 0;
        for (int i = 0; i < nrecv && ok; ++i) {
            const ShmBoxHeader* ph = box_hdr(recvs[i].peer);
            if ((size_t)ph->cnt[g_rank] != recvs[i].count) {
                std::fprintf(stderr, "### ERROR: fasp_hip: halo size mismatch: rank %d expects %zu from %d, offered %lld\n",
                             g_rank, recvs[i].count, recvs[i].peer, ph->cnt[g_rank]);
                ok = false;
                break;
            }
            std::memcpy(g_stage + roff, box_data(recvs[i].peer) + ph->off[g_rank], sizeof(double) * recvs[i].count);
            ok = hipMemcpyAsync(recvs[i].buf, g_stage + roff, sizeof(double) * recvs[i].count, hipMemcpyHostToDevice, stream) == hipSuccess;
            roff += recvs[i].count;
        }
        if (ok) ok = hipStreamSynchronize(stream) == hipSuccess;
        if (!ok) shm_raise_error();
        if (shm_barrier() < 0) return ERROR_MISC;
        return FASP_SUCCESS;
    }
    if (g_backend == IPC) return ipc_exchange(sends, nsend, recvs, nrecv, stream);
    NCK(g_rccl.GroupStart());
    for (int i = 0; i < nrecv; ++i)
        if (recvs[i].count) NCK(g_rccl.Recv(recvs[i].buf, recvs[i].count, ncclDouble, recvs[i].peer, g_comm, stream));
    for (int i = 0; i < nsend; ++i)
        if (sends[i].count) NCK(g_rccl.Send(sends[i].buf, sends[i].count, ncclDouble, sends[i].peer, g_comm, stream));
    NCK(g_rccl.GroupEnd());
    return FASP_SUCCESS;
}

static int comm_allgatherv_impl(const double* sendbuf, int sendcount, double* recvbuf, const int* counts,
                    const int* displs, hipStream_t stream)
{
    if (g_size <= 1) {
        if (recvbuf + displs[0] != sendbuf)
            HCK(hipMemcpyAsync(recvbuf + displs[0], sendbuf, sizeof(double) * sendcount, hipMemcpyDeviceToDevice, stream));
        return FASP_SUCCESS;
    }
    if (g_backend == SHM) {
        bool ok = (size_t)sendcount <= box_cap();
        if (ok) ok = hipMemcpyAsync(g_stage, sendbuf, sizeof(double) * sendcount, hipMemcpyDeviceToHost, stream) == hipSuccess &&
                     hipStreamSynchronize(stream) == hipSuccess;
        if (ok) std::memcpy(box_data(g_rank), g_stage, sizeof(double) * sendcount);
        else shm_raise_error();
        { const int bs = shm_barrier(); if (bs < 0) { if (bs != SHM_TIMEOUT) (void)shm_barrier(); return ERROR_MISC; } }
        for (int r = 0; r < g_size && ok; ++r) {
            if (r == g_rank || counts[r] == 0) continue;
            ok = hipMemcpy(recvbuf + displs[r], box_data(r), sizeof(double) * counts[r], hipMemcpyHostToDevice) == hipSuccess;
        }
        if (ok && recvbuf + displs[g_rank] != sendbuf)
            ok = hipMemcpyAsync(recvbuf + displs[g_rank], sendbuf, sizeof(double) * sendcount, hipMemcpyDeviceToDevice, stream) == hipSuccess;
        if (ok) ok = hipStreamSynchronize(stream) == hipSuccess;
        if (!ok) shm_raise_error();
        if (shm_barrier() < 0) return ERROR_MISC;
        return FASP_SUCCESS;
    }
    if (g_backend == IPC) {   // my block into every peer's mailbox, theirs out of mine (blocks longer than a mailbox: in pieces)
        long long done = 0, most = 0;
        for (int r = 0; r < g_size; ++r) most = std::max<long long>(most, counts[r]);
        if (recvbuf + displs[g_rank] != sendbuf)
            HCK(hipMemcpyAsync(recvbuf + displs[g_rank], sendbuf, sizeof(double) * sendcount, hipMemcpyDeviceToDevice, stream));
        while (done < most) {
            const long long piece = std::min<long long>((long long)g_ipc.cap, most - done);
            std::vector<CommXfer> sends, recvs;
            for (int r = 0; r < g_size; ++r) {
                if (r == g_rank) continue;
                const long long ns = std::max<long long>(0, std::min<long long>(piece, sendcount - done)), nr = std::max<long long>(0, std::min<long long>(piece, counts[r] - done));
                if (ns > 0) sends.push_back({r, const_cast<double*>(sendbuf) + done, (size_t)ns});
                if (nr > 0) recvs.push_back({r, recvbuf + displs[r] + done, (size_t)nr});
            }
            if (ipc_exchange(sends.data(), (int)sends.size(), recvs.data(), (int)recvs.size(), stream) < 0) return ERROR_MISC;
            done += piece;
        }
        return FASP_SUCCESS;
    }
    // all-gather with per-rank counts as a group of broadcasts
    NCK(g_rccl.GroupStart());
    for (int r = 0; r < g_size; ++r)
        if (counts[r])
            NCK(g_rccl.Broadcast(r == g_rank ? sendbuf : recvbuf + displs[r], recvbuf + displs[r],
                                 (size_t)counts[r], ncclDouble, r, g_comm, stream));
    NCK(g_rccl.GroupEnd());
    return FASP_SUCCESS;
}


// ---- counters and (diagnostic mode) times of the three primitives -------------------------------------------------------
// fasp_hip_comm_stats: [0] halo exchanges, [1] all-reduces, [2] all-gathers, [3] doubles sent in exchanges, [4] doubles
// contributed to all-gathers, [5..7] seconds in the three (diagnostic mode only: fasp_hip_comm_timing(1) drains the stream in
// front of and behind every call, so a call's time is its own -- and the solve is serialised: a breakdown, not a benchmark).
static double g_cstat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static bool   g_comm_timing = false;
struct CommTimer {
    hipStream_t s; int slot; double t0 = 0.0;
    CommTimer(hipStream_t s_, int slot_) : s(s_), slot(slot_) { if (g_comm_timing) { (void)hipStreamSynchronize(s); t0 = wall_seconds(); } }
    ~CommTimer() { if (g_comm_timing) { (void)hipStreamSynchronize(s); g_cstat[slot] += wall_seconds() - t0; } }
};
int comm_allreduce(double* dbuf, int n, unsigned maxmask, hipStream_t stream)
{
    if (g_size <= 1) return FASP_SUCCESS;
    g_cstat[1] += 1.0;
    CommTimer t(stream, 6);
    return comm_allreduce_impl(dbuf, n, maxmask, stream);
}
int comm_exchange(const CommXfer* sends, int nsend, const CommXfer* recvs, int nrecv, hipStream_t stream)
{
    if (g_size <= 1) return FASP_SUCCESS;
    g_cstat[0] += 1.0;
    for (int i = 0; i < nsend; ++i) g_cstat[3] += (double)sends[i].count;
    CommTimer t(stream, 5);
    return comm_exchange_impl(sends, nsend, recvs, nrecv, stream);
}
int comm_allgatherv(const double* sendbuf, int sendcount, double* recvbuf, const int* counts, const int* displs, hipStream_t stream)
{
    if (g_size > 1) { g_cstat[2] += 1.0; g_cstat[4] += (double)sendcount; }
    CommTimer t(stream, 7);
    return comm_allgatherv_impl(sendbuf, sendcount, recvbuf, counts, displs, stream);
}

}  // namespace fasp

using namespace fasp;

extern "C" {

int fasp_hip_comm_timing(int on) { g_comm_timing = on != 0; return FASP_SUCCESS; }
int fasp_hip_comm_stats(double* out, int reset)
{
    if (out) for (int i = 0; i < 8; ++i) out[i] = g_cstat[i];
    if (reset) for (double& v : g_cstat) v = 0.0;
    return FASP_SUCCESS;
}


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
    if (nranks <= 1) { g_rank = 0; g_size = 1; g_backend = NONE; return FASP_SUCCESS; }
    if (!id_bytes || rank < 0 || rank >= nranks) return ERROR_INPUT_PAR;
    if (g_backend != NONE) return ERROR_INPUT_PAR;
    if (load_rccl() < 0) return ERROR_MISC;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    NCK(g_rccl.CommInitRank(&g_comm, nranks, id, rank));
    g_rank = rank;
    g_size = nranks;
    g_backend = RCCL;
    g_comm_failed = false;
    return FASP_SUCCESS;
}

// Host-staged transport through the shared-memory segment /<name> (validation only).
int fasp_hip_comm_init_shm(int rank, int nranks, const char* name)
{
    if (nranks <= 1) { g_rank = 0; g_size = 1; g_backend = NONE; return FASP_SUCCESS; }
    if (!name || rank < 0 || rank >= nranks || nranks > 64 || g_backend != NONE) return ERROR_INPUT_PAR;
    g_shm_name = std::string("/") + name;
    g_shm_bytes = 4096 + (size_t)nranks * SHM_BOX_BYTES;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(g_shm_name.c_str());
        fd = shm_open(g_shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)g_shm_bytes) != 0) { std::perror("fasp_hip shm create"); return ERROR_MISC; }
    } else {
        for (int tries = 0; tries < 3000 && fd < 0; ++tries) {
            fd = shm_open(g_shm_name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < g_shm_bytes)) { close(fd); fd = -1; }
            if (fd < 0) usleep(10000);
        }
        if (fd < 0) { std::fprintf(stderr, "### ERROR: fasp_hip: cannot open shm segment %s\n", g_shm_name.c_str()); return ERROR_MISC; }
    }
    g_shm_base = (char*)mmap(nullptr, g_shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (g_shm_base == MAP_FAILED) { g_shm_base = nullptr; return ERROR_MISC; }
    if (rank == 0) {
        new (&shm_hdr()->arrive) std::atomic<int>(0);
        new (&shm_hdr()->generation) std::atomic<int>(0);
        new (&shm_hdr()->error) std::atomic<int>(0);
        shm_hdr()->nranks = nranks;
        std::atomic_thread_fence(std::memory_order_seq_cst);
        reinterpret_cast<std::atomic<int>*>(g_shm_base + 2048)->store(0x5A5A, std::memory_order_release);
    } else {
        while (reinterpret_cast<std::atomic<int>*>(g_shm_base + 2048)->load(std::memory_order_acquire) != 0x5A5A) usleep(1000);
    }
    if (hipHostMalloc((void**)&g_stage, SHM_BOX_BYTES, hipHostMallocDefault) != hipSuccess) return ERROR_ALLOC_MEM;
    g_rank = rank;
    g_size = nranks;
    g_backend = SHM;
    g_comm_failed = false;
    return shm_barrier();
}

// Peer windows (comm_ipc.h): the shared-memory segment /<name> carries the IPC handles and the barriers of start-up and
// shut-down only; everything during a solve goes from device to device.
int fasp_hip_comm_init_ipc(int rank, int nranks, const char* name)
{
    if (nranks <= 1) { g_rank = 0; g_size = 1; g_backend = NONE; return FASP_SUCCESS; }
    if (nranks > IPC_MAX_RANKS) return ERROR_INPUT_PAR;
    const int st = fasp_hip_comm_init_shm(rank, nranks, name);   // (segment, ready handshake, first barrier; the backend is switched below)
    if (st < 0) return st;
    size_t cap = 1u << 19;   // doubles per mailbox: 4 MB (a 512^2 halo plane is 2 MB)
    if (const char* e = std::getenv("FASP_HIP_IPC_CAP")) { const long long v = std::atoll(e); if (v >= 1024) cap = (size_t)v; }
    g_ipc = IpcState{};
    g_ipc.cap = cap;
    g_ipc.bytes = ipc_window_bytes();
    void* w = nullptr;
    bool ok = hipExtMallocWithFlags(&w, g_ipc.bytes, hipDeviceMallocUncached) == hipSuccess;
    if (ok) ok = hipMemset(w, 0, g_ipc.bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    hipIpcMemHandle_t hnd;
    if (ok) ok = hipIpcGetMemHandle(&hnd, w) == hipSuccess;
    int dev = -1;
    (void)hipGetDevice(&dev);
    char busid[64] = {0};
    (void)hipDeviceGetPCIBusId(busid, sizeof(busid), dev);
    struct Slot { hipIpcMemHandle_t h; char bus[64]; int ok; };
    static_assert(sizeof(Slot) <= 4096 - 2048 - 64, "handle slot");
    auto slot = [&](int r) { return reinterpret_cast<Slot*>(shm_box(r)); };   // (the mailbox area of the host-staged transport: unused here)
    if (ok) { slot(rank)->h = hnd; std::memcpy(slot(rank)->bus, busid, sizeof(busid)); }
    slot(rank)->ok = ok ? 1 : 0;
    if (!ok) shm_raise_error();
    // a failed start-up leaves nothing behind: the window is freed, the peers' mappings closed, the segment given back
    auto fail = [&]() {
        (void)hipDeviceSynchronize();
        for (int q = 0; q < nranks; ++q) if (q != rank && g_ipc.win[q]) (void)hipIpcCloseMemHandle(g_ipc.win[q]);
        if (w) (void)hipFree(w);
        if (g_ipc.counters) (void)hipFree(g_ipc.counters);
        if (g_ipc.err) (void)hipHostFree(g_ipc.err);
        g_ipc = IpcState{};
        (void)fasp_hip_comm_finalize();   // (backend SHM at this point: unmaps / unlinks the segment, back to NONE)
        return ERROR_MISC;
    };
    if (shm_barrier() < 0) { std::fprintf(stderr, "### ERROR: fasp_hip: peer-window start-up failed (window allocation or IPC export on a rank)\n"); return fail(); }
    g_ipc.win[rank] = static_cast<char*>(w);
    g_ipc_shared_device = false;
    for (int q = 0; q < nranks && ok; ++q) {
        if (q == rank) continue;
        if (!std::strcmp(slot(q)->bus, busid)) g_ipc_shared_device = true;   // validation: several ranks on one GPU
        void* p = nullptr;
        ok = hipIpcOpenMemHandle(&p, slot(q)->h, hipIpcMemLazyEnablePeerAccess) == hipSuccess;
        g_ipc.win[q] = static_cast<char*>(p);
    }
    if (ok) ok = hipMalloc((void**)&g_ipc.counters, sizeof(unsigned) * 8) == hipSuccess && hipMemset(g_ipc.counters, 0, sizeof(unsigned) * 8) == hipSuccess;
    if (ok) ok = hipHostMalloc((void**)&g_ipc.err, 64, hipHostMallocDefault) == hipSuccess;
    if (ok) { std::memset(g_ipc.err, 0, 64); ok = hipDeviceSynchronize() == hipSuccess; }
    if (!ok) { std::fprintf(stderr, "### ERROR: fasp_hip: rank %d cannot map its peers' windows (hipIpcOpenMemHandle)\n", rank); shm_raise_error(); }
    if (shm_barrier() < 0) return fail();
    g_backend = IPC;
    return FASP_SUCCESS;
}

int fasp_hip_comm_finalize(void)
{
    if (g_backend == IPC && g_shm_base) {
        (void)hipDeviceSynchronize();
        (void)shm_barrier();   // nobody unmaps a window a peer may still write
        for (int q = 0; q < g_size; ++q) if (q != g_rank && g_ipc.win[q]) (void)hipIpcCloseMemHandle(g_ipc.win[q]);
        (void)shm_barrier();
        if (g_ipc.win[g_rank]) (void)hipFree(g_ipc.win[g_rank]);
        if (g_ipc.counters) (void)hipFree(g_ipc.counters);
        if (g_ipc.err) (void)hipHostFree(g_ipc.err);
        g_ipc = IpcState{};
        g_backend = SHM;   // (the segment goes the way of the host-staged transport's)
    }
    if (g_backend == RCCL && g_comm) {
        NCK(g_rccl.CommDestroy(g_comm));
        g_comm = nullptr;
    }
    if (g_backend == SHM && g_shm_base) {
        (void)shm_barrier();
        munmap(g_shm_base, g_shm_bytes);
        g_shm_base = nullptr;
        if (g_rank == 0) shm_unlink(g_shm_name.c_str());
        if (g_stage) { (void)hipHostFree(g_stage); g_stage = nullptr; }
    }
    g_backend = NONE;
    g_rank = 0;
    g_size = 1;
    return FASP_SUCCESS;
}

// One-rank exercise of every RCCL entry point the transport uses (symbol loading, unique id,
// communicator, grouped all-reduce with mixed ops, broadcast, send/recv to self): what can be
// checked of the production transport on a box with a single GPU.
int fasp_hip_comm_selftest(void)
{
    if (load_rccl() < 0) return ERROR_MISC;
    ncclUniqueId id;
    NCK(g_rccl.GetUniqueId(&id));
    ncclComm_t c = nullptr;
    NCK(g_rccl.CommInitRank(&c, 1, id, 0));
    hipStream_t s = nullptr;
    HCK(hipStreamCreate(&s));
    double h[8] = {1.5, -2.0, 3.25, 7.0, 0.5, 6.0, 0.0, 0.0}, out[8] = {0};
    double* d = nullptr;
    HCK(hipMalloc((void**)&d, sizeof(double) * 16));
    HCK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
    NCK(g_rccl.GroupStart());
    NCK(g_rccl.AllReduce(d, d, 3, ncclDouble, ncclSum, c, s));
    NCK(g_rccl.AllReduce(d + 3, d + 3, 1, ncclDouble, ncclMax, c, s));
    NCK(g_rccl.GroupEnd());
    NCK(g_rccl.GroupStart());
    NCK(g_rccl.Broadcast(d + 4, d + 8, 2, ncclDouble, 0, c, s));
    NCK(g_rccl.GroupEnd());
    NCK(g_rccl.GroupStart());
    NCK(g_rccl.Recv(d + 10, 2, ncclDouble, 0, c, s));
    NCK(g_rccl.Send(d + 4, 2, ncclDouble, 0, c, s));
    NCK(g_rccl.GroupEnd());
    HCK(hipStreamSynchronize(s));
    double all[16];
    HCK(hipMemcpy(all, d, sizeof(all), hipMemcpyDeviceToHost));
    (void)out;
    int bad = 0;
    for (int i = 0; i < 4; ++i) bad += all[i] != h[i];
    bad += all[8] != 0.5 || all[9] != 6.0 || all[10] != 0.5 || all[11] != 6.0;
    NCK(g_rccl.CommDestroy(c));
    (void)hipStreamDestroy(s);
    (void)hipFree(d);
    return bad ? ERROR_MISC : FASP_SUCCESS;
}

int fasp_hip_comm_rank(void) { return g_rank; }
int fasp_hip_comm_size(void) { return g_size; }

}  // extern "C"
