// Inter-rank exchange of the adaptors over RCCL (xGMI) INSIDE the library: grouped ncclSend / ncclRecv per halo and ncclAllReduce for
// the dots and the coarse defect, all enqueued on the context's stream -- no callback into the host program and no event wait per
// exchange (what dune/ddm/hip/mpi_exchange.hh has to pay).  One rank = one GPU = one subdomain (examples/poisson.cc:128-131).
// Counterparts in the reference: the MPI calls behind comm->copyOwnerToAll / addOwnerCopyToOwnerCopy / addOwnerCopyToAll
// (schwarz.hh:125,138,142; nonoverlapping_operator.hh:38,48; galerkin_preconditioner.hh:162,190), comm->dot (MPI_Allreduce) and
// MPI_Gatherv / MPI_Scatterv of the coarse defect (galerkin_preconditioner.hh:171,183).
//
// RCCL needs one 128-byte id made on ONE rank and known to all before the communicator exists; how it travels is the host program's
// business (it already has MPI in a DUNE build):
//     ddm_hip::RcclId id;
//     if (rank == 0) id = ddm_hip::make_rccl_id();
//     MPI_Bcast(id.bytes, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
//     ddm_hip::install_rccl_exchange(id, rank, size, local_device);        // collective (ncclCommInitRank)
// or, with HAVE_MPI, the one-liner  ddm_hip::install_rccl_exchange(MPI_COMM_WORLD);  below.
// Call once, before constructing any operator.  self_test routes the rank's own halo segment and all reductions through RCCL too,
// which exercises the whole path on a communicator of size 1 (tests/cpp/rccl_exchange_check.cc).
#pragma once

#include <cstring>
#include <memory>

#include "backend.hh"

namespace ddm_hip {

struct RcclId {
  unsigned char bytes[128];
  RcclId() { std::memset(bytes, 0, sizeof bytes); }
};

inline RcclId make_rccl_id()
{
  RcclId id;
  if (ddm_rccl_unique_id(id.bytes) != DDM_OK) DUNE_THROW(Dune::Exception, "ddm_rccl_unique_id failed (librccl not found?)");
  return id;
}

inline std::shared_ptr<Context> install_rccl_exchange(const RcclId& id, int rank, int size, int device = -1, bool self_test = false)
{
  if (size < 1 || rank < 0 || rank >= size) DUNE_THROW(Dune::InvalidStateException, "install_rccl_exchange: rank " << rank << " of " << size);
  auto ctx = Context::get(device < 0 ? 0 : device);
  check(ctx->handle(), ddm_ctx_set_rccl(ctx->handle(), rank, size, id.bytes, self_test ? 1 : 0), "ddm_ctx_set_rccl");
  int seen = 0;
  check(ctx->handle(), ddm_ctx_rccl_size(ctx->handle(), &seen), "ddm_ctx_rccl_size");
  if (seen != size) DUNE_THROW(Dune::InvalidStateException, "RCCL reports " << seen << " ranks, the host program " << size);
  ctx->rank = rank;
  ctx->nranks = size;
  ctx->exchange_installed = true;
  return ctx;
}

}  // namespace ddm_hip

#if HAVE_MPI
#include <mpi.h>
namespace ddm_hip {
// id made on rank 0 and broadcast over the given communicator; device = -1: local rank modulo the visible devices
inline std::shared_ptr<Context> install_rccl_exchange(MPI_Comm comm, int device = -1)
{
  int rank = 0, size = 1;
  MPI_Comm_rank(comm, &rank);
  MPI_Comm_size(comm, &size);
  if (device < 0) {
    MPI_Comm local;
    MPI_Comm_split_type(comm, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &local);
    MPI_Comm_rank(local, &device);
    MPI_Comm_free(&local);
  }
  RcclId id;
  if (rank == 0) id = make_rccl_id();
  MPI_Bcast(id.bytes, (int)sizeof id.bytes, MPI_BYTE, 0, comm);
  return install_rccl_exchange(id, rank, size, device);
}
}  // namespace ddm_hip
#endif
