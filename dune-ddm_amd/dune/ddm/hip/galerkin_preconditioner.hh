// MI355X drop-in for dune/ddm/galerkin_preconditioner.hh (GalerkinPreconditioner).
// apply():       R d on the device, replicated dense coarse solve (every rank holds A0^-1; replaces
//                MPI_Gatherv / rank-0 solve / MPI_Scatterv of :170-183), R^T x0, halo sum.
// build_solver(): A0 = R A R^T (:219-349): Y = A_dir r_j as device SpMV, <r_i, Y> as wavefront
//                reductions (ddm_galerkin_products); neighbours' vectors restricted to the shared
//                indices are fetched with one copy-halo per neighbour; the k x K slabs are summed
//                over the ranks (gatherMatrixFromRowsFlat, helpers.hh:204-339, replicated).
#pragma once

#include <cmath>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include <dune/common/parallel/interface.hh>
#include <dune/common/parametertree.hh>
#include <dune/istl/preconditioner.hh>
#include <dune/istl/solvercategory.hh>

#include "backend.hh"

namespace ddm_hip {
// The coarse solver behind `solver->apply` on rank 0 of the reference (galerkin_preconditioner.hh:49,178,338-346), chosen by the
// `type` key of the coarse sub-tree through the dune-istl factory.  Here every rank holds the replicated dense K x K matrix and
// applies its explicit inverse on the device (ddm_galerkin_apply: k_dense_mv), so the factory key selects the dense
// factorisation the inverse is formed from:
//   umfpack | superlu | spqr | hip_lu | lu          L U with partial pivoting (row interchanges), any non-singular A0
//   cholmod | ldl | hip_cholesky | cholesky         L L^T, A0 must be symmetric positive definite (Dune::Exception otherwise,
//                                                   as CHOLMOD reports "not positive definite")
// Iterative factory entries (cgsolver, ...) have no meaning for the replicated explicit inverse: Dune::NotImplemented.
inline std::vector<double> invert_lu(std::vector<double> a, int K)
{
  std::vector<int> piv(K);
  for (int c = 0; c < K; ++c) {
    int p = c;
    for (int r = c + 1; r < K; ++r)
      if (std::abs(a[(std::size_t)r * K + c]) > std::abs(a[(std::size_t)p * K + c])) p = r;
    if (a[(std::size_t)p * K + c] == 0.0) DUNE_THROW(Dune::Exception, "coarse matrix R A R^T is singular");
    piv[c] = p;
    if (p != c)
      for (int j = 0; j < K; ++j) std::swap(a[(std::size_t)p * K + j], a[(std::size_t)c * K + j]);
    const double d = 1.0 / a[(std::size_t)c * K + c];
    for (int r = c + 1; r < K; ++r) {
      const double f = a[(std::size_t)r * K + c] * d;
      a[(std::size_t)r * K + c] = f;
      if (f == 0.0) continue;
      for (int j = c + 1; j < K; ++j) a[(std::size_t)r * K + j] -= f * a[(std::size_t)c * K + j];
    }
  }
  // the inverse row by row: solve L U X = P I (the right-hand sides are the rows of the permuted identity)
  std::vector<double> inv((std::size_t)K * K, 0.0);
  for (int i = 0; i < K; ++i) inv[(std::size_t)i * K + i] = 1.0;
  for (int c = 0; c < K; ++c)
    if (piv[c] != c)
      for (int j = 0; j < K; ++j) std::swap(inv[(std::size_t)piv[c] * K + j], inv[(std::size_t)c * K + j]);
  for (int r = 1; r < K; ++r)
    for (int c = 0; c < r; ++c) {
      const double f = a[(std::size_t)r * K + c];
      if (f == 0.0) continue;
      for (int j = 0; j < K; ++j) inv[(std::size_t)r * K + j] -= f * inv[(std::size_t)c * K + j];
    }
  for (int r = K - 1; r >= 0; --r) {
    for (int c = r + 1; c < K; ++c) {
      const double f = a[(std::size_t)r * K + c];
      if (f == 0.0) continue;
      for (int j = 0; j < K; ++j) inv[(std::size_t)r * K + j] -= f * inv[(std::size_t)c * K + j];
    }
    const double d = 1.0 / a[(std::size_t)r * K + r];
    for (int j = 0; j < K; ++j) inv[(std::size_t)r * K + j] *= d;
  }
  return inv;
}
inline std::vector<double> invert_cholesky(std::vector<double> a, int K)
{
  for (int c = 0; c < K; ++c) {   // a = L L^T, L in the lower triangle
    double d = a[(std::size_t)c * K + c];
    for (int k = 0; k < c; ++k) d -= a[(std::size_t)c * K + k] * a[(std::size_t)c * K + k];
    if (!(d > 0.0)) DUNE_THROW(Dune::Exception, "coarse matrix R A R^T is not positive definite (this coarse solver type needs an SPD matrix; use umfpack)");
    const double l = std::sqrt(d);
    a[(std::size_t)c * K + c] = l;
    for (int r = c + 1; r < K; ++r) {
      double v = a[(std::size_t)r * K + c];
      for (int k = 0; k < c; ++k) v -= a[(std::size_t)r * K + k] * a[(std::size_t)c * K + k];
      a[(std::size_t)r * K + c] = v / l;
    }
  }
  std::vector<double> inv((std::size_t)K * K, 0.0);
  for (int i = 0; i < K; ++i) inv[(std::size_t)i * K + i] = 1.0;
  for (int r = 0; r < K; ++r) {   // L Y = I
    for (int c = 0; c < r; ++c) {
      const double f = a[(std::size_t)r * K + c];
      for (int j = 0; j <= c; ++j) inv[(std::size_t)r * K + j] -= f * inv[(std::size_t)c * K + j];
    }
    const double d = 1.0 / a[(std::size_t)r * K + r];
    for (int j = 0; j <= r; ++j) inv[(std::size_t)r * K + j] *= d;
  }
  for (int r = K - 1; r >= 0; --r) {   // L^T X = Y
    for (int c = r + 1; c < K; ++c) {
      const double f = a[(std::size_t)c * K + r];
      for (int j = 0; j < K; ++j) inv[(std::size_t)r * K + j] -= f * inv[(std::size_t)c * K + j];
    }
    const double d = 1.0 / a[(std::size_t)r * K + r];
    for (int j = 0; j < K; ++j) inv[(std::size_t)r * K + j] *= d;
  }
  return inv;
}
inline std::vector<double> coarse_inverse(const std::vector<double>& a0, int K, const std::string& type)
{
  if (type == "umfpack" || type == "superlu" || type == "spqr" || type == "hip_lu" || type == "lu") return invert_lu(a0, K);
  if (type == "cholmod" || type == "ldl" || type == "hip_cholesky" || type == "cholesky") return invert_cholesky(a0, K);
  DUNE_THROW(Dune::NotImplemented, "coarse solver type '" + type + "': the replicated device coarse solve takes the direct solvers of the factory (umfpack, superlu, spqr, cholmod, ldl)");
}
}  // namespace ddm_hip

template <class Vec, class Communication>
class GalerkinPreconditioner : public Dune::Preconditioner<Vec, Vec>, public ddm_hip::DeviceLevel {
public:
  // reference ctor: galerkin_preconditioner.hh:118-144
  template <class Mat>
  GalerkinPreconditioner(const Mat& A, const std::vector<Vec>& ts, std::shared_ptr<Communication> comm, const Dune::ParameterTree& ptree,
                         const std::string& subtree_name = "galerkin")
      : comm(std::move(comm)), n(A.N()), num_t((int)ts.size()), ctx(ddm_hip::Context::get())
  {
    // the coarse solver's factory key (:338-346), checked before any work like every other configuration error
    const auto& subtree = subtree_name.size() == 0 ? ptree : ptree.sub(subtree_name);
    if (not subtree.hasKey("type")) DUNE_THROW(Dune::Exception, "You must specify the solver in the subtree " << subtree_name << " using the key 'type'");   // :344-345
    coarse_solver_type = subtree.get("type", std::string(""));
    (void)ddm_hip::coarse_inverse(std::vector<double>{1.0}, 1, coarse_solver_type);   // unknown type: throws here
    ctx->require(this->comm->communicator());
    if (ts.size() == 0) DUNE_THROW(Dune::Exception, "Must at least pass one template vector");              // :129
    if (ts[0].N() != A.N()) DUNE_THROW(Dune::Exception, "Template vectors must match size of matrix");      // :131
    const auto& cc = this->comm->communicator();
    const int rank = cc.rank(), size = cc.size();
    num_t_per_rank.resize(size);
    cc.allgather(&num_t, 1, num_t_per_rank.data());                                                          // :248
    total_num_t = std::accumulate(num_t_per_rank.begin(), num_t_per_rank.end(), 0);
    offset_per_rank.assign(size, 0);
    std::exclusive_scan(num_t_per_rank.begin(), num_t_per_rank.end(), offset_per_rank.begin(), 0);           // :256
    const int kmax = *std::max_element(num_t_per_rank.begin(), num_t_per_rank.end());
    if (kmax > 256) DUNE_THROW(Dune::NotImplemented, "more than 256 coarse vectors per subdomain");   // COARSE_KMAX of the library

    // basis as kmax x n row-major (zero rows beyond num_t), copied like restr_vecs (:138-139)
    basis.assign((std::size_t)kmax * n, 0.0);
    for (int j = 0; j < num_t; ++j)
      for (std::size_t i = 0; i < n; ++i) basis[(std::size_t)j * n + i] = ts[j][i];

    dA = std::make_unique<ddm_hip::DeviceCsr>(ctx, A);
    typename Communication::OwnerSet owner;
    typename Communication::AllSet all;
    typename Communication::OwnerCopySet oc;
    Dune::Interface copy_if, add_if;
    copy_if.build(this->comm->remoteIndices(), owner, all);
    add_if.build(this->comm->remoteIndices(), oc, all);       // addOwnerCopyToAll (:190)
    h_copy = std::make_unique<ddm_hip::Halo>(ctx, 4, 0, copy_if);
    h_add = std::make_unique<ddm_hip::Halo>(ctx, 5, 1, add_if);
    build_solver(A, add_if, rank, size, kmax);
  }
  ~GalerkinPreconditioner() override { ddm_galerkin_destroy(G); }

  Dune::SolverCategory::Category category() const override { return Dune::SolverCategory::nonoverlapping; }
  void pre(Vec&, Vec&) override {}
  void post(Vec&) override {}

  void apply(Vec& x, const Vec& d) override   // :151-194
  {
    if (!G) create(d.N());
    dd->upload(d);
    ddm_hip::check(ctx->handle(), ddm_galerkin_apply(ctx->handle(), G, dx->data(), dd->data()), "ddm_galerkin_apply");
    dx->download(x);
  }
  ddm_galerkin* galerkin_handle(std::size_t n_novlp) override { return handle(n_novlp); }
  ddm_galerkin* handle(std::size_t n_novlp)
  {
    if (!G) create(n_novlp);
    return G;
  }
  const std::vector<double>& coarse_matrix() const { return A0; }   // K x K row-major (for inspection / tests)
  int coarse_size() const { return total_num_t; }

private:
  template <class Mat, class Interface>
  void build_solver(const Mat&, const Interface& all_if, int rank, int size, int kmax)   // :219-349
  {
    const int K = total_num_t;
    A0.assign((std::size_t)K * K, 0.0);
    ddm_hip::DeviceVector R(ctx, (std::size_t)kmax * n), V(ctx, (std::size_t)kmax * n);
    ddm_hip::check(ctx->handle(), ddm_memcpy_h2d(ctx->handle(), R.data(), basis.data(), (int64_t)(basis.size() * sizeof(double))), "h2d basis");
    std::vector<double> blk((std::size_t)kmax * kmax);
    // local x local (:292-295)
    ddm_hip::check(ctx->handle(), ddm_galerkin_products(ctx->handle(), dA->handle(), kmax, R.data(), kmax, R.data(), 0, (int64_t)n, blk.data()), "galerkin products");
    for (int i = 0; i < num_t; ++i)
      for (int j = 0; j < num_t; ++j) A0[(std::size_t)(offset_per_rank[rank] + i) * K + offset_per_rank[rank] + j] = blk[(std::size_t)j * kmax + i];
    // local x remote (:298-309, 321-327): one neighbour at a time; its vectors restricted to the shared indices
    // arrive through a copy-halo that carries only this neighbour's (all -> all) index lists.
    int tag = 100;
    for (const auto& [nbr, info] : all_if.interfaces()) {
      struct OneNeighbour {
        std::map<int, std::pair<decltype(info.first), decltype(info.second)>> m;
        const auto& interfaces() const { return m; }
      } one;
      one.m.emplace(nbr, std::make_pair(info.first, info.second));
      ddm_hip::Halo h(ctx, tag++, 0, one);
      // V_j = the neighbour's vector j on the shared indices, zero elsewhere (CopyGatherScatterWithRank, :66-103): pack from R,
      // unpack into the zeroed V -- all on the device (round 2 moved the whole basis over PCIe three times per neighbour for this)
      ddm_hip::check(ctx->handle(), ddm_memset_zero(ctx->handle(), V.data(), (int64_t)(basis.size() * sizeof(double))), "zero");
      for (int j = 0; j < kmax; ++j)
        ddm_hip::check(ctx->handle(), ddm_halo_exchange_to(ctx->handle(), h.handle(), R.data() + (std::size_t)j * n, V.data() + (std::size_t)j * n), "basis exchange");
      ddm_hip::check(ctx->handle(), ddm_galerkin_products(ctx->handle(), dA->handle(), kmax, R.data(), kmax, V.data(), 0, (int64_t)n, blk.data()), "galerkin products");
      for (int i = 0; i < num_t; ++i)
        for (int j = 0; j < num_t_per_rank[nbr]; ++j) A0[(std::size_t)(offset_per_rank[rank] + i) * K + offset_per_rank[nbr] + j] = blk[(std::size_t)j * kmax + i];
    }
    comm->communicator().sum(A0.data(), (int)A0.size());   // every rank gets the full K x K matrix (replaces the gather to rank 0, :331)
    (void)size;
    A0inv = ddm_hip::coarse_inverse(A0, K, coarse_solver_type);
  }

  void create(std::size_t n_novlp)
  {
    const int kmax = (int)(basis.size() / n);
    std::vector<int32_t> ext(n);
    for (std::size_t i = 0; i < n; ++i) ext[i] = i < n_novlp ? (int32_t)i : -1;
    const int64_t sub_ptr[2] = {0, (int64_t)n};
    std::vector<int64_t> cidx(kmax, -1);
    const int rank = comm->communicator().rank();
    for (int j = 0; j < num_t; ++j) cidx[j] = offset_per_rank[rank] + j;
    ddm_hip::check(ctx->handle(),
                   ddm_galerkin_create(ctx->handle(), (int64_t)n, (int64_t)n_novlp, ext.data(), 1, sub_ptr, kmax, basis.data(), cidx.data(), total_num_t,
                                       A0inv.data(), h_copy->handle(), h_add->handle(), &G),
                   "ddm_galerkin_create");
    dd = std::make_unique<ddm_hip::DeviceVector>(ctx, n_novlp);
    dx = std::make_unique<ddm_hip::DeviceVector>(ctx, n_novlp);
  }

  std::shared_ptr<Communication> comm;
  std::size_t n;
  int num_t;
  int total_num_t{};
  std::vector<int> num_t_per_rank, offset_per_rank;
  std::shared_ptr<ddm_hip::Context> ctx;
  std::vector<double> basis, A0, A0inv;
  std::string coarse_solver_type;
  std::unique_ptr<ddm_hip::DeviceCsr> dA;
  std::unique_ptr<ddm_hip::Halo> h_copy, h_add;
  std::unique_ptr<ddm_hip::DeviceVector> dd, dx;
  ddm_galerkin* G = nullptr;
};
