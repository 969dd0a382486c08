// piqp_amd/csrc/rccl_transport.hpp -- RCCL (xGMI) collectives issued by the library itself on a backend's stream.
// Used by the stage-partitioned sparse backend (pq_kkt_set_comm_rccl, include/piqp_amd.h): the three exchanges of a partitioned
// factor / solve become ncclAllReduce / ncclAllGather calls enqueued behind the producing kernels -- no stream drain, no host callback.
// librccl is loaded with dlopen on first use: libpiqp_amd.so keeps libamdhip64 as its only link-time dependency and loads on machines
// without RCCL (single-GPU use never touches this file).
#pragma once

#include <cstddef>

#include "common.hpp"

namespace pq {
namespace rccl {

constexpr int UNIQUE_ID_BYTES = 128;  // NCCL_UNIQUE_ID_BYTES (rccl.h:40)

struct Comm;  // owns one ncclComm_t

void unique_id(unsigned char out[UNIQUE_ID_BYTES]);                                     // ncclGetUniqueId (rank 0; ship the bytes to the other ranks)
Comm* comm_create(const unsigned char id[UNIQUE_ID_BYTES], int rank, int world, int device);  // ncclCommInitRank on `device`
void comm_destroy(Comm* c);
void comm_info(Comm* c, int out[3]);  // what the communicator itself reports: ncclCommCount, ncclCommUserRank, ncclCommCuDevice
void all_reduce_sum(Comm* c, double* buf, size_t count, hipStream_t s);                 // in place, fp64 sum
void all_reduce_max(Comm* c, double* buf, size_t count, hipStream_t s);                 // in place, fp64 max (the sharded refinement residual's norm)
void all_gather(Comm* c, double* buf, size_t count_per_rank, int rank, hipStream_t s);  // in place: this rank's chunk at rank * count_per_rank

}  // namespace rccl
}  // namespace pq
