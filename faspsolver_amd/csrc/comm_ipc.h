// comm_ipc.h -- the peer-window transport of the multi-GPU path (comm.cpp builds the arguments, comm_ipc.hip holds the kernels).
// Every rank owns one UNCACHED device window (hipExtMallocWithFlags(hipDeviceMallocUncached), exported with hipIpcGetMemHandle,
// mapped by every peer): per peer and parity a mailbox {payload, sequence word}, per peer an acknowledgement word, per peer and
// parity a reduction slot.  A message is written by the SENDER's kernel straight into the receiver's mailbox (xGMI stores), made
// visible by a system-scope fence, announced by its sequence number; the receiver's part of the same kernel polls the word and
// copies the payload to where the vector's ghost entries live.  No RCCL call, no second stream hop, one launch per exchange.
#pragma once
#include <hip/hip_runtime.h>

namespace fasp {

constexpr int IPC_MAX_RANKS = 8;     // one node
constexpr int IPC_RED_MAX = 32;      // doubles per all-reduce

struct IpcSend { double* remote_data; unsigned long long* remote_flag; const unsigned long long* ack_in; const double* src; long long n; unsigned long long seq; };
struct IpcRecv { const double* local_data; const unsigned long long* local_flag; unsigned long long* remote_ack; double* dst; long long n; unsigned long long seq; };
struct IpcXchgArgs {
    IpcSend  s[IPC_MAX_RANKS];
    IpcRecv  r[IPC_MAX_RANKS];
    int      ns, nr;
    unsigned* counters;   // two words of this rank's own device memory (arrival counters of the two "last block" steps)
    unsigned* err;        // pinned host word: a poll that timed out
};
struct IpcRedArgs {
    double*             remote_val[IPC_MAX_RANKS];    // peer q's slot for MY contribution (this parity)
    unsigned long long* remote_flag[IPC_MAX_RANKS];
    const double*       local_val[IPC_MAX_RANKS];     // my slot for peer q's contribution
    const unsigned long long* local_flag[IPC_MAX_RANKS];
    double*  dbuf;
    int      n, me, nranks;
    unsigned maxmask;
    unsigned long long epoch;
    unsigned* err;
};
int ipc_xchg_launch(const IpcXchgArgs& a, long long total_elems, hipStream_t stream);
int ipc_allreduce_launch(const IpcRedArgs& a, hipStream_t stream);

}  // namespace fasp
