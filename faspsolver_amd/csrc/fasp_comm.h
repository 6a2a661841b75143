// fasp_comm.h -- multi-GPU communicator of libfasp_hip.so (one process per GPU,
// RCCL over xGMI).  RCCL is loaded with dlopen on first use so that single-GPU use
// does not depend on it.
#pragma once
#include <hip/hip_runtime.h>

namespace fasp {

int  comm_rank();
int  comm_size();
bool comm_failed();       // sticky: a collective failed on this rank or (SHM transport) on a peer
void comm_mark_failed();
bool comm_shares_devices();   // validation transport: several ranks may sit on one GPU (no kernel may assume it owns the chip)
bool comm_is_peer_window();   // the transport is hipIpc-mapped device windows (one small kernel per exchange / all-gather: comm_ipc.h)
// sum-reduce (and max-reduce the entries whose bit is set in maxmask) n doubles in
// place across all ranks, on `stream`.  No-op when comm_size() == 1.
int  comm_allreduce(double* dbuf, int n, unsigned maxmask, hipStream_t stream);
// point-to-point exchange of double buffers with an arbitrary set of peers; all the
// sends and receives are grouped into one RCCL group call.
struct CommXfer { int peer; double* buf; size_t count; };
int  comm_exchange(const CommXfer* sends, int nsend, const CommXfer* recvs, int nrecv, hipStream_t stream);
int  comm_allgatherv(const double* sendbuf, int sendcount, double* recvbuf, const int* counts,
                     const int* displs, hipStream_t stream);

}  // namespace fasp
