// comm_ipc.hip -- kernels of the peer-window transport (comm_ipc.h).  Windows are uncached device memory: stores leave the chip,
// loads come from memory; what has to be ORDERED is the payload before its sequence word (system-scope release fence in every
// storing block, the word written by the block that arrives last) and the word before the payload reads (system-scope acquire).
#include "comm_ipc.h"

#include <cstdio>

namespace fasp {

typedef __attribute__((address_space(1))) unsigned long long ipc_gu64;
__device__ __forceinline__ unsigned long long ipc_ld(const unsigned long long* p)
{
    return __hip_atomic_load((ipc_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ipc_st(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store((ipc_gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// bounded poll (10 s at 100 MHz): until *p >= want (exact == false) or *p == want
__device__ __forceinline__ bool ipc_wait(const unsigned long long* p, unsigned long long want, bool exact, unsigned* err)
{
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long v = ipc_ld(p);
        if (exact ? v == want : v >= want) return true;
        if ((spins & 255u) == 255u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (!t0) t0 = now;
            else if (now - t0 > 1000000000ull) { *err = 1u; __threadfence_system(); return false; }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

constexpr int IPC_T = 256;
__global__ __launch_bounds__(IPC_T) void k_ipc_xchg(IpcXchgArgs a)
{
    const int tid = threadIdx.x;
    __shared__ int s_last;
    // flow control: a mailbox is rewritten two messages later; the receiver must have consumed the one that sits there
    if (tid < a.ns && a.s[tid].seq >= 2) (void)ipc_wait(a.s[tid].ack_in, a.s[tid].seq - 2, false, a.err);
    __syncthreads();
    // put: this rank's boundary entries straight into the peers' mailboxes
    for (int m = 0; m < a.ns; ++m) {
        const IpcSend& S = a.s[m];
        for (long long i = (long long)blockIdx.x * IPC_T + tid; i < S.n; i += (long long)gridDim.x * IPC_T) S.remote_data[i] = S.src[i];
    }
    __threadfence_system();   // (every storing block: its stores have left before it counts itself in)
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(a.counters, 1u) == gridDim.x - 1;
    __syncthreads();
    if (s_last) {
        __threadfence_system();   // (release: the arrival counts of all blocks -- and through them their payload stores -- before the word)
        if (tid < a.ns) ipc_st(a.s[tid].remote_flag, a.s[tid].seq);
        if (tid == 0) a.counters[0] = 0u;   // (the next launch on this stream finds it reset)
    }
    // get: wait for the peers' words, copy the payloads to the ghost entries.  The acquire fence sits BEHIND the barrier: every
    // wave of the block -- not only the polling threads -- orders its payload loads after the word has been seen.
    if (tid < a.nr) (void)ipc_wait(a.r[tid].local_flag, a.r[tid].seq, true, a.err);
    __syncthreads();
    __threadfence_system();
    for (int m = 0; m < a.nr; ++m) {
        const IpcRecv& R = a.r[m];
        for (long long i = (long long)blockIdx.x * IPC_T + tid; i < R.n; i += (long long)gridDim.x * IPC_T) R.dst[i] = R.local_data[i];
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(a.counters + 1, 1u) == gridDim.x - 1;
    __syncthreads();
    if (s_last) {   // every block has read its share: tell the senders
        __threadfence_system();
        if (tid < a.nr) ipc_st(a.r[tid].remote_ack, a.r[tid].seq);
        if (tid == 0) a.counters[1] = 0u;
    }
}

// all-reduce of n <= IPC_RED_MAX doubles: everybody writes its contribution into everybody's window, polls the P - 1 words of
// its own, and sums IN RANK ORDER (maximum for the masked entries, seeded with rank 0's value): the same bits on every rank
__global__ __launch_bounds__(64) void k_ipc_allreduce(IpcRedArgs a)
{
    __shared__ double s_v[IPC_MAX_RANKS][IPC_RED_MAX];
    const int q = threadIdx.x;
    if (q < a.nranks) {
        if (q == a.me) { for (int i = 0; i < a.n; ++i) s_v[q][i] = a.dbuf[i]; }
        else {
            for (int i = 0; i < a.n; ++i) a.remote_val[q][i] = a.dbuf[i];
            __threadfence_system();
            ipc_st(a.remote_flag[q], a.epoch);
            (void)ipc_wait(a.local_flag[q], a.epoch, true, a.err);
            __threadfence_system();   // (acquire in the polling thread itself: it is the one that reads the payload)
            for (int i = 0; i < a.n; ++i) s_v[q][i] = a.local_val[q][i];
        }
    }
    __syncthreads();
    if (q < a.n) {
        const bool mx = (a.maxmask >> q) & 1u;
        double v = s_v[0][q];
        for (int r = 1; r < a.nranks; ++r) { const double x = s_v[r][q]; v = mx ? (x > v ? x : v) : v + x; }
        a.dbuf[q] = v;
    }
}

int ipc_xchg_launch(const IpcXchgArgs& a, long long total_elems, hipStream_t stream)
{
    int grid = (int)((total_elems + 4095) / 4096);
    grid = grid < 1 ? 1 : grid > 64 ? 64 : grid;
    hipLaunchKernelGGL(k_ipc_xchg, dim3(grid), dim3(IPC_T), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int ipc_allreduce_launch(const IpcRedArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(k_ipc_allreduce, dim3(1), dim3(64), 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace fasp
