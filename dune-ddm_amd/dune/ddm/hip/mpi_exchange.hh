// Inter-rank exchange of the adaptors in a DUNE (MPI) build: installs the two callbacks of ddm_ctx_set_comm on the context.
// One MPI rank = one GPU = one subdomain (examples/poisson.cc:128-131).  The library hands over packed DEVICE buffers; with a
// GPU-aware MPI they go straight into MPI_Isend / MPI_Irecv, otherwise (DDM_HIP_STAGE_THROUGH_HOST) they are staged.
// Counterparts in the reference: the MPI calls behind comm->copyOwnerToAll / addOwnerCopyToOwnerCopy / addOwnerCopyToAll
// (schwarz.hh:125,138,142; nonoverlapping_operator.hh:38,48; galerkin_preconditioner.hh:162,190), comm->dot (MPI_Allreduce) and
// MPI_Gatherv / MPI_Scatterv of the coarse defect (galerkin_preconditioner.hh:171,183).
// MPI is driven from the host, so each callback has to wait until the data it sends exists: one event wait (ddm_ctx_fence) per
// exchange -- 3 halos + 4 reductions per CG iteration.  That cost is inherent to a host-driven transport; the exchange without
// any host involvement is ddm_ctx_set_rccl (everything enqueued on the stream).
// Call once, before constructing any operator:   ddm_hip::install_mpi_exchange(MPI_COMM_WORLD, device);
// (Alternative without MPI in the data path: ddm_ctx_set_rccl, include/ddm_hip.h -- RCCL over xGMI inside the library.)
#pragma once

#if HAVE_MPI
#include <mpi.h>

#include <vector>

#include "backend.hh"

namespace ddm_hip {

struct MpiExchange {
  MPI_Comm comm = MPI_COMM_NULL;
  std::shared_ptr<Context> ctx;
  std::vector<double> hsend, hrecv;   // host staging (only without GPU-aware MPI)
};

inline int mpi_alltoall_cb(void* user, int tag, const double* sendbuf, double* recvbuf)
{
  auto& x = *static_cast<MpiExchange*>(user);
  const auto it = x.ctx->halo_counts.find(tag);
  if (it == x.ctx->halo_counts.end()) return 1;
  const auto& c = it->second;
  if (ddm_ctx_fence(x.ctx->handle()) != DDM_OK) return 1;   // event wait: the pack kernel enqueued before this callback has finished
  const int P = (int)c.send_counts.size();
  int64_t ns = 0, nr = 0;
  for (int r = 0; r < P; ++r) { ns += c.send_counts[r]; nr += c.recv_counts[r]; }
  const double* s = sendbuf;
  double* r_ = recvbuf;
#ifdef DDM_HIP_STAGE_THROUGH_HOST
  x.hsend.resize(ns);
  x.hrecv.resize(nr);
  if (ns && ddm_memcpy_d2h(x.ctx->handle(), x.hsend.data(), sendbuf, ns * 8) != DDM_OK) return 1;
  s = x.hsend.data();
  r_ = x.hrecv.data();
#endif
  std::vector<MPI_Request> rq;
  int64_t so = 0, ro = 0;
  for (int p = 0; p < P; ++p) {
    if (p != x.ctx->rank) {
      if (c.recv_counts[p]) { rq.emplace_back(); MPI_Irecv(r_ + ro, (int)c.recv_counts[p], MPI_DOUBLE, p, tag, x.comm, &rq.back()); }
      if (c.send_counts[p]) { rq.emplace_back(); MPI_Isend(s + so, (int)c.send_counts[p], MPI_DOUBLE, p, tag, x.comm, &rq.back()); }
    }
    so += c.send_counts[p];
    ro += c.recv_counts[p];
  }
  const int rc = MPI_Waitall((int)rq.size(), rq.data(), MPI_STATUSES_IGNORE);
#ifdef DDM_HIP_STAGE_THROUGH_HOST
  if (nr && ddm_memcpy_h2d(x.ctx->handle(), recvbuf, x.hrecv.data(), nr * 8) != DDM_OK) return 1;
#endif
  return rc == MPI_SUCCESS ? 0 : 1;
}

inline int mpi_allreduce_cb(void* user, double* buf, int64_t n)
{
  auto& x = *static_cast<MpiExchange*>(user);
  if (ddm_ctx_fence(x.ctx->handle()) != DDM_OK) return 1;   // the partial sums are in `buf`
#ifdef DDM_HIP_STAGE_THROUGH_HOST
  x.hsend.resize(n);
  if (ddm_memcpy_d2h(x.ctx->handle(), x.hsend.data(), buf, n * 8) != DDM_OK) return 1;
  if (MPI_Allreduce(MPI_IN_PLACE, x.hsend.data(), (int)n, MPI_DOUBLE, MPI_SUM, x.comm) != MPI_SUCCESS) return 1;
  return ddm_memcpy_h2d(x.ctx->handle(), buf, x.hsend.data(), n * 8) == DDM_OK ? 0 : 1;
#else
  return MPI_Allreduce(MPI_IN_PLACE, buf, (int)n, MPI_DOUBLE, MPI_SUM, x.comm) == MPI_SUCCESS ? 0 : 1;
#endif
}

// device = -1: local rank modulo the visible devices
inline std::shared_ptr<Context> install_mpi_exchange(MPI_Comm comm, int device = -1)
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
  auto ctx = Context::get(device);
  static MpiExchange ex;   // lives as long as the program (the context is a process-wide singleton)
  ex.comm = comm;
  ex.ctx = ctx;
  check(ctx->handle(), ddm_ctx_set_comm(ctx->handle(), rank, size, mpi_alltoall_cb, mpi_allreduce_cb, &ex), "ddm_ctx_set_comm");
  ctx->rank = rank;
  ctx->nranks = size;
  ctx->exchange_installed = true;
  return ctx;
}

}  // namespace ddm_hip
#endif   // HAVE_MPI
