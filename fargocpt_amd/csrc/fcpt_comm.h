// The two communication points of the path over RCCL (xGMI between the GPUs of a node), internal interface:
//   * CommunicateBoundaries (commbound.cpp:130-158): MPI_Isend/Irecv of the packed overlap rings with the radial
//     neighbours CPU_Prev / CPU_Next  ->  one group of ncclSend/ncclRecv with rank - 1 / rank + 1
//   * condition_cfl's MPI_Allreduce(MPI_MIN) (cfl.cpp:379)  ->  ncclAllReduce(ncclMin) of one device double
// Both are enqueued on the caller's HIP stream: no host synchronisation, no event wait between the step's kernels
// and the transfers.  librccl is bound at run time (the copy already in the process when the host program is a
// PyTorch-ROCm one, /opt/rocm's otherwise); a single-GPU user of libfargocpt_hip.so never loads it.
#ifndef FCPT_COMM_H
#define FCPT_COMM_H

#include <hip/hip_runtime.h>

#include <cstddef>

namespace fcpt {

struct Comm;

// ncclGetUniqueId: 128 opaque bytes that rank 0 hands to every rank (the host's own bootstrap: MPI_Bcast, a file,
// a torch.distributed store) before comm_create
int comm_unique_id(void *id128);
// ncclCommInitRank on the current HIP device; collective over all `nranks` callers
int comm_create(const void *id128, int rank, int nranks, Comm **out);
// The same two operations for ranks that cannot use RCCL among themselves (several ranks on one GPU: tests, or more
// slabs than devices): ghost rings and the CFL value are staged through pinned host buffers and a shared-memory
// file `path` (created by rank 0; one file per run), `count` doubles per message.  Blocks the host in every
// operation -- a rehearsal transport, not a fast one.
int comm_create_host(const char *path, int rank, int nranks, size_t count, Comm **out);
bool comm_is_host_staged(const Comm *c);
void comm_destroy(Comm *c);
int comm_rank(const Comm *c);
int comm_size(const Comm *c);
// `count` doubles each way; peer < 0 skips that side.  Sends and receives of one call form one RCCL group.
int comm_neighbour_exchange(Comm *c, int peer_inner, const double *send_inner, double *recv_inner, int peer_outer,
                            const double *send_outer, double *recv_outer, size_t count, hipStream_t st);
int comm_allreduce_min(Comm *c, double *d_value, hipStream_t st);
// MPI_Barrier (+ completion of the work queued on st)
int comm_barrier(Comm *c, hipStream_t st);

} // namespace fcpt
#endif
