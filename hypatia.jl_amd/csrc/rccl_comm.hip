// RCCL inside the library: the exchange step of the cone-sharded path (reference: the sum over cones of
// src/Solvers/systemsolvers/qrchol.jl:219-246 and the per-cone sweeps of src/Solvers/search.jl:118-134, which have no
// multi-device form in the reference).  One process per GPU; the communicator is created by the library from a unique id
// the caller distributes out of band (torch.distributed / MPI / a file), so the host framework never touches the data path:
// ncclAllReduce(sum | max | min, double) in place on the library's own stream over xGMI.
#include <rccl/rccl.h>
#include <cstring>
#include "../../include/hypatia_hip.h"
#include "hyp_internal.hpp"

namespace hyp {

#define HYP_NCCL(expr)                                                                                          \
  do {                                                                                                          \
    ncclResult_t r_ = (expr);                                                                                   \
    if (r_ != ncclSuccess)                                                                                      \
      throw hyp::HipError(-2000 - (int)r_, std::string(#expr) + ": " + ncclGetErrorString(r_) + " at " __FILE__ ":" + std::to_string(__LINE__)); \
  } while (0)

void rccl_allreduce_inplace(void* comm, double* d_buf, long count, int op, hipStream_t st) {
  const ncclRedOp_t o = (op == 0) ? ncclSum : (op == 1 ? ncclMax : ncclMin);
  HYP_NCCL(ncclAllReduce(d_buf, d_buf, (size_t)count, ncclDouble, o, (ncclComm_t)comm, st));
}

void rccl_unique_id(char* out128) {
  static_assert(NCCL_UNIQUE_ID_BYTES == 128, "ncclUniqueId size");
  ncclUniqueId id;
  HYP_NCCL(ncclGetUniqueId(&id));
  std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
}

void* rccl_init_rank(int device, int nranks, int rank, const char* id128) {
  HYP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks && id128 != nullptr, "hyp_comm_init_rank: arguments");
  HYP_CHECK(hipSetDevice(device));
  ncclUniqueId id;
  std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  HYP_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
  return (void*)comm;
}

void rccl_destroy(void* comm) {
  if (comm) (void)ncclCommDestroy((ncclComm_t)comm);
}

}  // namespace hyp
