// Host-side glue between DUNE containers and the C ABI of libddm_hip.so (include/ddm_hip.h).
// Header-only; uses only the public dune-common / dune-istl API that the reference itself uses:
//   row iteration of BCRSMatrix            (cf. the flattening in dune/ddm/strumpack.hh:36-62)
//   BlockVector<FieldVector<double,1>>     contiguous doubles, &x[0][0]
//   Dune::Interface::interfaces()          per-neighbour send / receive index lists
// Errors of the C ABI become DUNE_THROW(Dune::Exception, ...) like the reference's own checks.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <dune/common/exceptions.hh>

#include "ddm_hip.h"

namespace ddm_hip {

inline void check(ddm_ctx* ctx, int rc, const char* what)
{
  if (rc == DDM_OK) return;
  const std::string msg = std::string(what) + ": " + (ctx ? ddm_last_error(ctx) : "no context");
  if (rc == DDM_ENOTIMPL) DUNE_THROW(Dune::NotImplemented, msg);
  if (rc == DDM_EINVAL) DUNE_THROW(Dune::InvalidStateException, msg);
  DUNE_THROW(Dune::Exception, msg);
}

// One context per process (= per MPI rank = per GPU), shared by all operators of the solve.
// Multi-rank runs: call ddm_hip::install_mpi_exchange(ctx, MPI_Comm) (mpi_exchange.hh) -- or ddm_ctx_set_rccl /
// ddm_ctx_set_comm yourself -- right after the first Context::get(); without an installed exchange every adaptor that
// meets a communicator of size > 1 throws Dune::NotImplemented instead of computing rank-local nonsense.
class Context {
public:
  static std::shared_ptr<Context> get(int device = -1)
  {
    static std::weak_ptr<Context> inst;
    auto p = inst.lock();
    if (!p) {
      p = std::shared_ptr<Context>(new Context(device < 0 ? 0 : device));
      inst = p;
    }
    return p;
  }
  ~Context() { ddm_ctx_destroy(h_); }
  ddm_ctx* handle() const { return h_; }
  int rank = 0, nranks = 1;
  bool exchange_installed = false;
  // called by every adaptor with its DUNE communicator: rank / size must agree with the installed exchange
  template <class Communicator>
  void require(const Communicator& cc)
  {
    const int sz = cc.size(), rk = cc.rank();
    if (sz == 1) return;
    if (!exchange_installed)
      DUNE_THROW(Dune::NotImplemented, "communicator of size " << sz << ": install the inter-rank exchange first (ddm_hip::install_mpi_exchange, dune/ddm/hip/mpi_exchange.hh)");
    if (sz != nranks || rk != rank)
      DUNE_THROW(Dune::InvalidStateException, "communicator rank/size (" << rk << "/" << sz << ") differ from the installed exchange (" << rank << "/" << nranks << ")");
  }
  // per-halo send/receive layout, kept for the MPI exchange (tag -> per-peer counts)
  struct HaloCounts {
    std::vector<int64_t> send_counts, recv_counts;
  };
  std::map<int, HaloCounts> halo_counts;

private:
  explicit Context(int device)
  {
    if (ddm_ctx_create(device, nullptr, &h_) != DDM_OK)
      DUNE_THROW(Dune::Exception, "ddm_ctx_create failed: no HIP device (the MI355X path has no CPU fallback)");
  }
  ddm_ctx* h_ = nullptr;
};

// RAII device vector of doubles
class DeviceVector {
public:
  DeviceVector(std::shared_ptr<Context> ctx, std::size_t n) : ctx_(std::move(ctx)), n_(n)
  {
    void* p = nullptr;
    check(ctx_->handle(), ddm_malloc(ctx_->handle(), (int64_t)(n * sizeof(double)), &p), "ddm_malloc");
    d_ = static_cast<double*>(p);
  }
  DeviceVector(const DeviceVector&) = delete;
  DeviceVector& operator=(const DeviceVector&) = delete;
  ~DeviceVector() { ddm_free(ctx_->handle(), d_); }
  double* data() const { return d_; }
  std::size_t size() const { return n_; }
  template <class Vec>
  void upload(const Vec& v)
  {
    check(ctx_->handle(), ddm_memcpy_h2d(ctx_->handle(), d_, &v[0][0], (int64_t)(n_ * sizeof(double))), "h2d");
  }
  template <class Vec>
  void download(Vec& v) const
  {
    check(ctx_->handle(), ddm_memcpy_d2h(ctx_->handle(), &v[0][0], d_, (int64_t)(n_ * sizeof(double))), "d2h");
  }

private:
  std::shared_ptr<Context> ctx_;
  std::size_t n_;
  double* d_ = nullptr;
};

// Flattened scalar BCRSMatrix on the device
class DeviceCsr {
public:
  template <class Mat>
  DeviceCsr(std::shared_ptr<Context> ctx, const Mat& A) : ctx_(std::move(ctx))
  {
    std::vector<int64_t> rp;
    std::vector<int32_t> ci;
    std::vector<double> va;
    rp.reserve(A.N() + 1);
    ci.reserve(A.nonzeroes());
    va.reserve(A.nonzeroes());
    rp.push_back(0);
    for (auto ri = A.begin(); ri != A.end(); ++ri) {
      for (auto cit = ri->begin(); cit != ri->end(); ++cit) {
        ci.push_back(static_cast<int32_t>(cit.index()));
        va.push_back((*cit)[0][0]);
      }
      rp.push_back(static_cast<int64_t>(ci.size()));
    }
    n_ = A.N();
    check(ctx_->handle(), ddm_csr_create(ctx_->handle(), (int64_t)A.N(), (int64_t)A.M(), rp.data(), ci.data(), va.data(), &h_), "ddm_csr_create");
  }
  DeviceCsr(const DeviceCsr&) = delete;
  ~DeviceCsr() { ddm_csr_destroy(h_); }
  ddm_csr* handle() const { return h_; }
  std::size_t N() const { return n_; }

private:
  std::shared_ptr<Context> ctx_;
  ddm_csr* h_ = nullptr;
  std::size_t n_ = 0;
};

// One DUNE interface (source attribute set -> destination attribute set) as a ddm_halo.
// `iface` is a built Dune::Interface: interfaces() maps a neighbour rank to its (send, receive)
// InterfaceInformation; both lists are ordered by global index on either side.
class Halo {
public:
  template <class Interface>
  Halo(std::shared_ptr<Context> ctx, int tag, int mode, const Interface& iface) : ctx_(std::move(ctx))
  {
    const int P = ctx_->nranks;
    std::vector<int64_t> send_idx, send_counts(P, 0), recv_counts(P, 0);
    std::vector<std::pair<int64_t, int64_t>> dst;   // (destination local index, position in recv buffer)
    int64_t pos = 0;
    for (const auto& [nbr, info] : iface.interfaces()) {   // std::map: ascending neighbour rank
      if (nbr < 0 || nbr >= P)
        DUNE_THROW(Dune::InvalidStateException, "interface names neighbour rank " << nbr << " but the context knows " << P << " rank(s): install the inter-rank exchange before building operators");
      for (std::size_t i = 0; i < info.first.size(); ++i) send_idx.push_back((int64_t)info.first[i]);
      send_counts[nbr] = (int64_t)info.first.size();
      for (std::size_t i = 0; i < info.second.size(); ++i) dst.emplace_back((int64_t)info.second[i], pos++);
      recv_counts[nbr] = (int64_t)info.second.size();
    }
    // group by destination entry, keeping the neighbour order (= the order DUNE scatters the messages)
    std::stable_sort(dst.begin(), dst.end(), [](auto& a, auto& b) { return a.first < b.first; });
    std::vector<int64_t> dst_idx, dst_ptr{0}, src_pos;
    for (std::size_t k = 0; k < dst.size(); ++k) {
      if (k == 0 || dst[k].first != dst[k - 1].first) {
        if (k) dst_ptr.push_back((int64_t)src_pos.size());
        dst_idx.push_back(dst[k].first);
      }
      src_pos.push_back(dst[k].second);
    }
    if (!dst.empty()) dst_ptr.push_back((int64_t)src_pos.size());
    check(ctx_->handle(),
          ddm_halo_create(ctx_->handle(), tag, mode, (int64_t)send_idx.size(), send_idx.data(), send_counts.data(), recv_counts.data(),
                          (int64_t)dst_idx.size(), dst_idx.data(), dst_ptr.data(), src_pos.data(), &h_),
          "ddm_halo_create");
    ctx_->halo_counts[tag] = Context::HaloCounts{send_counts, recv_counts};
  }
  Halo(const Halo&) = delete;
  ~Halo() { ddm_halo_destroy(h_); }
  ddm_halo* handle() const { return h_; }

private:
  std::shared_ptr<Context> ctx_;
  ddm_halo* h_ = nullptr;
};

// Non-template views of the device-backed adaptors, so that CombinedPreconditioner can fuse them
// into one ddm_combined object without knowing their matrix / communication types.
struct DeviceLevel {
  virtual ~DeviceLevel() = default;
  virtual ddm_schwarz* schwarz_handle(std::size_t /*n_novlp*/) { return nullptr; }
  virtual ddm_galerkin* galerkin_handle(std::size_t /*n_novlp*/) { return nullptr; }
};
struct DeviceOperator {
  virtual ~DeviceOperator() = default;
  virtual ddm_op* op_handle() const = 0;
};

}  // namespace ddm_hip
