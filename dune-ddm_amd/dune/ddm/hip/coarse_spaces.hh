// MI355X drop-in for the coarse-space builders of dune/ddm/coarsespaces/coarse_spaces.hh that lie on the GenEO path:
//   CoarseSpaceBuilder<Vec>   (:219-256)  get_basis() / size() / get_setup_task()
//   GenEOCoarseSpace<Mat,Vec> (:286-331)  ctor (A, B, pou, ptree, taskflow, prefix = "geneo"); eigenproblem on the device
//   POUCoarseSpace<Vec>       (:1175-1231) pou / ||pou||_2, or POU-scaled template vectors
// and the remaining builders, all on ddm_geneo_basis / ddm_msgfem_basis / ddm_harmonic_*:
//   EnergyMinimalExtension<Mat,Vec>            (energy_minimal_extension.hh:36-229)
//   MsGFEMCoarseSpace<Mat,MaskVec1,MaskVec2,Vec>      (:663-831)    the default of examples/poisson.ini:36
//   ConstraintGenEOCoarseSpace<Mat,MaskVec,Vec>       (:394-490)    = GenEO: solve_gevp drops the callback (eigensolvers.hh:27-30)
//   GenEORingCoarseSpace<Mat,Vec>                     (:502-648)
//   MsGFEMRingCoarseSpace<Mat,MaskVec1,MaskVec2,Vec>  (:913-1163)
//   HarmonicExtensionCoarseSpace<Vec>                 (:1232-1266)
// Same names, constructor signatures, ParameterTree keys (`<prefix>.eigensolver.{nev, tolerance, shift, threshold, nev_max}`,
// dune/ddm/eigensolvers/eigensolver_params.hh:8-62) and exception texts.  The eigensolver is ddm_geneo_basis (C ABI): block
// method on the device instead of Spectra's single-vector Lanczos; `ncv`, `maxit`, `seed`, `blocksize` are parsed by the
// reference but have no meaning for it (the reference's driver hard-codes maxit = 100, spectra.hh:137).
// As in the reference the returned vectors are v <- D v / ||D v||_2 (finalize_eigenvectors, :52-61); the caller zeroes the
// Dirichlet entries (examples/poisson.cc:235-238).  Rows of A that the symmetric Dirichlet elimination turned into unit rows
// (examples/pdelab_helper.hh:33-46) are recognised from the matrix and their decoupled unit modes are not returned (csrc/geneo.hpp).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include <dune/common/exceptions.hh>
#include <dune/common/parametertree.hh>
#include <dune/istl/bvector.hh>

#include <dune/ddm/pou.hh>

#if DUNE_DDM_HAVE_TASKFLOW
#include <taskflow/taskflow.hpp>
#endif

#include "backend.hh"

template <class Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>>
class CoarseSpaceBuilder {
public:
  virtual ~CoarseSpaceBuilder() = default;
  CoarseSpaceBuilder(const CoarseSpaceBuilder&) = delete;
  CoarseSpaceBuilder& operator=(const CoarseSpaceBuilder&) = delete;
  CoarseSpaceBuilder(CoarseSpaceBuilder&&) = delete;
  CoarseSpaceBuilder& operator=(CoarseSpaceBuilder&&) = delete;

  virtual const std::vector<Vec>& get_basis() const { return basis_; }
  virtual std::size_t size() const { return basis_.size(); }
#if DUNE_DDM_HAVE_TASKFLOW
  virtual tf::Task& get_setup_task() { return setup_task; }
#endif

protected:
  CoarseSpaceBuilder() = default;
  std::vector<Vec> basis_;
#if DUNE_DDM_HAVE_TASKFLOW
  tf::Task setup_task;
#endif
};

namespace ddm_hip {
// the keys of `<prefix>.eigensolver` (dune/ddm/eigensolvers/eigensolver_params.hh:8-62) -> ddm_geneo_params
inline ddm_geneo_params eigensolver_params(const Dune::ParameterTree& eig_ptree)
{
  const auto type = eig_ptree.get("type", std::string("Spectra"));
  if (type != "Spectra") DUNE_THROW(Dune::NotImplemented, "Unknown eigensolver type '" + type + "'");   // eigensolver_params.hh:35
  ddm_geneo_params par;
  ddm_geneo_params_default(&par);
  par.nev = eig_ptree.get("nev", par.nev);
  par.nev_max = eig_ptree.hasKey("nev_max") ? par.nev : 2 * par.nev;   // sic: the key `nev_max` overwrites ncv in the reference (:23), nev_max stays 2 nev unless unset
  par.tolerance = eig_ptree.get("tolerance", par.tolerance);
  par.shift = eig_ptree.get("shift", par.shift);
  par.threshold = eig_ptree.get("threshold", par.threshold);
  par.verbose = eig_ptree.get("verbose", 0);
  return par;
}
// detail::finalize_eigenvectors (coarse_spaces.hh:52-61)
template <class Vec>
void finalize_eigenvectors(std::vector<Vec>& vecs, const PartitionOfUnity& pou)
{
  for (auto& v : vecs) {
    double nrm = 0;
    for (std::size_t i = 0; i < v.N(); ++i) {
      const double x = (double)v[i][0] * pou[i];
      v[i] = x;
      nrm += x * x;
    }
    const double s = 1.0 / std::sqrt(nrm);
    for (std::size_t i = 0; i < v.N(); ++i) v[i] = (double)v[i][0] * s;
  }
}
// the column indices of row i
template <class Mat, class F>
void for_each_neighbour(const Mat& A, std::size_t i, F&& f)
{
  for (auto c = A[i].begin(); c != A[i].end(); ++c) f(c.index());
}
}   // namespace ddm_hip

// EnergyMinimalExtension(A, interior_indices, boundary_indices) (energy_minimal_extension.hh:36-229): extend() returns the interior
// values u_i = -A_ii^-1 (A [0; u_b])_i; the interior block is factorised on construction (ddm_harmonic_create).
template <class Mat, class Vec>
class EnergyMinimalExtension {
public:
  EnergyMinimalExtension(const Mat& A, const std::vector<std::size_t>& interior_indices, const std::vector<std::size_t>& boundary_indices)
      : ctx(ddm_hip::Context::get()), n(A.N()), interior_indices(interior_indices), boundary_indices(boundary_indices), dA(ctx, A)
  {
    std::vector<int64_t> ii(interior_indices.begin(), interior_indices.end()), bb(boundary_indices.begin(), boundary_indices.end());
    const int64_t bp[2] = {0, (int64_t)n};
    ddm_hip::check(ctx->handle(), ddm_harmonic_create(ctx->handle(), dA.handle(), 1, bp, (int64_t)ii.size(), ii.data(), (int64_t)bb.size(), bb.data(), &H), "ddm_harmonic_create");
  }
  EnergyMinimalExtension(const EnergyMinimalExtension&) = delete;
  EnergyMinimalExtension& operator=(const EnergyMinimalExtension&) = delete;
  ~EnergyMinimalExtension() { ddm_harmonic_destroy(H); }

  Vec extend(const Vec& boundary_values)   // :104-131
  {
    std::vector<Vec> one(1, boundary_values);
    return extend(one)[0];
  }
  // all vectors at once (the reference's SIMD variant, :169-216, without its divisibility restriction)
  std::vector<Vec> extend(const std::vector<Vec>& boundary_vectors)
  {
    const std::size_t k = boundary_vectors.size();
    std::vector<Vec> out(k, Vec(interior_indices.size()));
    if (k == 0) return out;
    std::vector<double> X(n * k, 0.0);   // row-major n x k
    for (std::size_t j = 0; j < k; ++j) {
      if (boundary_vectors[j].N() != boundary_indices.size()) DUNE_THROW(Dune::Exception, "EnergyMinimalExtension: boundary vector has the wrong size");
      for (std::size_t i = 0; i < boundary_indices.size(); ++i) X[boundary_indices[i] * k + j] = boundary_vectors[j][i];
    }
    ddm_hip::DeviceVector dX(ctx, n * k);
    ddm_hip::check(ctx->handle(), ddm_memcpy_h2d(ctx->handle(), dX.data(), X.data(), (int64_t)(X.size() * sizeof(double))), "h2d");
    ddm_hip::check(ctx->handle(), ddm_harmonic_extend(ctx->handle(), H, (int)k, dX.data(), (int64_t)k), "ddm_harmonic_extend");
    ddm_hip::check(ctx->handle(), ddm_memcpy_d2h(ctx->handle(), X.data(), dX.data(), (int64_t)(X.size() * sizeof(double))), "d2h");
    for (std::size_t j = 0; j < k; ++j)
      for (std::size_t i = 0; i < interior_indices.size(); ++i) out[j][i] = X[interior_indices[i] * k + j];
    return out;
  }

private:
  std::shared_ptr<ddm_hip::Context> ctx;
  std::size_t n;
  const std::vector<std::size_t>& interior_indices;   // references, as in the reference (:225-226)
  const std::vector<std::size_t>& boundary_indices;
  ddm_hip::DeviceCsr dA;
  ddm_harmonic* H = nullptr;
};

template <class Mat, class Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>>
class GenEOCoarseSpace : public CoarseSpaceBuilder<Vec> {
public:
#if DUNE_DDM_HAVE_TASKFLOW
  // reference ctor: coarse_spaces.hh:286-293 (shared_ptrs by value: captured by the task)
  GenEOCoarseSpace(std::shared_ptr<const Mat> A, std::shared_ptr<const Mat> B, std::shared_ptr<const PartitionOfUnity> pou, const Dune::ParameterTree& ptree,
                   tf::Taskflow& taskflow, const std::string& ptree_prefix = "geneo")
  {
    const auto& subtree = ptree.sub(ptree_prefix);
    Dune::ParameterTree eig_ptree = subtree.sub("eigensolver");
    this->setup_task = taskflow.emplace([A, B, pou, eig_ptree, this] { setup_geneo_impl(A, B, pou, eig_ptree); }).name("GenEO coarse space setup");
  }
  template <class TaskflowOrSubflow>
  auto create_setup_task(TaskflowOrSubflow& tf_, std::shared_ptr<const Mat> A, std::shared_ptr<const Mat> B, std::shared_ptr<const PartitionOfUnity> pou,
                         const Dune::ParameterTree& eig_ptree) -> tf::Task
  {
    return tf_.emplace([A, B, pou, eig_ptree, this] { setup_geneo_impl(A, B, pou, eig_ptree); }).name("GenEO coarse space setup");
  }
#endif
  GenEOCoarseSpace() = default;

  // eigenvalues of the returned vectors (ascending) and what the device eigensolver reported
  const std::vector<double>& eigenvalues() const { return eigenvalues_; }
  const ddm_geneo_info& info() const { return info_; }

  // core of the setup (coarse_spaces.hh:319-331); callable from any task context
  void setup_geneo_impl(std::shared_ptr<const Mat> A, std::shared_ptr<const Mat> B, std::shared_ptr<const PartitionOfUnity> pou, const Dune::ParameterTree& eig_ptree)
  {
    if (pou->size() != A->N()) DUNE_THROW(Dune::Exception, "The matrix and the partition of unity must have the same size");   // :323
    auto ctx = ddm_hip::Context::get();
    ddm_geneo_params par = ddm_hip::eigensolver_params(eig_ptree);
    const std::size_t n = A->N();
    ddm_hip::DeviceCsr dA(ctx, *A);
    std::unique_ptr<ddm_hip::DeviceCsr> dBown;
    if (A.get() != B.get()) dBown = std::make_unique<ddm_hip::DeviceCsr>(ctx, *B);
    std::vector<double> w(n);
    for (std::size_t i = 0; i < n; ++i) w[i] = (*pou)[i];
    // rows turned into unit rows by the symmetric Dirichlet elimination: only the diagonal is non-zero, in A and in B
    std::vector<std::uint8_t> dir(n, 0);
    auto unit_row = [](const Mat& M, std::size_t i, bool allow_empty) {
      bool diag_one = false, any = false;
      for (auto c = M[i].begin(); c != M[i].end(); ++c) {
        any = true;
        const double v = (*c)[0][0];
        if (c.index() == i) diag_one = (v == 1.0);
        else if (v != 0.0) return false;
      }
      return any ? diag_one : allow_empty;
    };
    for (std::size_t i = 0; i < n; ++i) dir[i] = (unit_row(*A, i, false) && unit_row(*B, i, true)) ? 1 : 0;
    const int64_t sub_ptr[2] = {0, (int64_t)n};
    const int kmax = par.threshold > 0 ? std::max(par.nev, par.nev_max) : par.nev;
    std::vector<double> basis((std::size_t)kmax * n), eig((std::size_t)kmax);
    int32_t nconv = 0;
    ddm_hip::check(ctx->handle(),
                   ddm_geneo_basis(ctx->handle(), dA.handle(), dBown ? dBown->handle() : dA.handle(), 1, sub_ptr, w.data(), dir.data(), &par, kmax, basis.data(), &nconv,
                                   eig.data(), &info_),
                   "ddm_geneo_basis");
    if (!info_.converged)   // the reference aborts when Spectra fails (spectra.hh:149-210)
      DUNE_THROW(Dune::Exception, "GenEO eigensolver did not converge in " << info_.iterations << " block iterations (worst residual " << info_.worst_residual << ")");
    this->basis_.assign(nconv, Vec(n));
    eigenvalues_.assign(eig.begin(), eig.begin() + nconv);
    for (int j = 0; j < nconv; ++j)
      for (std::size_t i = 0; i < n; ++i) this->basis_[j][i] = basis[(std::size_t)j * n + i];
  }

private:
  std::vector<double> eigenvalues_;
  ddm_geneo_info info_{};
};

template <class Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>>
class POUCoarseSpace : public CoarseSpaceBuilder<Vec> {
public:
#if DUNE_DDM_HAVE_TASKFLOW
  // coarse_spaces.hh:1195-1209: the single vector pou / ||pou||_2
  POUCoarseSpace(std::shared_ptr<const PartitionOfUnity> pou, tf::Taskflow& taskflow)
  {
    this->setup_task = taskflow.emplace([pou, this] { build(std::vector<Vec>(), *pou); }).name("POU coarse space setup");
  }
#endif
  // :1211-1224: the same without a task
  explicit POUCoarseSpace(const PartitionOfUnity& pou) { build(std::vector<Vec>(), pou); }
  // :1226-1230: POU-scaled template vectors (TwoLevelSchwarzSolver uses 1, x, y, xy: twolevel_schwarz.hh:68-107)
  POUCoarseSpace(const std::vector<Vec>& template_vecs, const PartitionOfUnity& pou) { build(template_vecs, pou); }

private:
  void build(const std::vector<Vec>& ts, const PartitionOfUnity& pou)
  {
    const std::size_t n = pou.size();
    const std::size_t k = ts.empty() ? 1 : ts.size();
    this->basis_.assign(k, Vec(n));
    for (std::size_t t = 0; t < k; ++t) {
      if (!ts.empty() && ts[t].N() != n) DUNE_THROW(Dune::Exception, "Template vectors must match size of the partition of unity");
      double nrm = 0;
      for (std::size_t i = 0; i < n; ++i) {
        const double v = (ts.empty() ? 1.0 : (double)ts[t][i][0]) * pou[i];   // finalize_eigenvectors (:52-61)
        this->basis_[t][i] = v;
        nrm += v * v;
      }
      const double s = 1.0 / std::sqrt(nrm);
      for (std::size_t i = 0; i < n; ++i) this->basis_[t][i] = this->basis_[t][i][0] * s;
    }
  }
};

namespace ddm_hip {
// one call of ddm_geneo_basis (boundary == nullptr: pencil (A0, D A1 D)) or ddm_msgfem_basis (A0 = A_neu, A1 = A_dir) for one
// subdomain; returns the vectors as the library delivers them (finalised, or 2-normalised raw eigenvectors with par.raw = 1)
template <class Vec, class Mat>
std::vector<Vec> run_eigensolver(const char* what, const Mat& A0, const Mat& A1, const std::vector<double>& pou, const std::vector<std::uint8_t>& dirichlet,
                                 const std::vector<std::uint8_t>* boundary, ddm_geneo_params par, std::vector<double>& eigenvalues, ddm_geneo_info& info)
{
  auto ctx = Context::get();
  const std::size_t n = A0.N();
  DeviceCsr d0(ctx, A0);
  std::unique_ptr<DeviceCsr> d1;
  if (&A0 != &A1) d1 = std::make_unique<DeviceCsr>(ctx, A1);
  const int64_t sub_ptr[2] = {0, (int64_t)n};
  const int kmax = par.threshold > 0 ? std::max(par.nev, par.nev_max) : par.nev;
  std::vector<double> basis((std::size_t)kmax * n), eig((std::size_t)kmax);
  int32_t nconv = 0;
  ddm_csr* h1 = d1 ? d1->handle() : d0.handle();
  if (boundary)
    check(ctx->handle(), ddm_msgfem_basis(ctx->handle(), d0.handle(), h1, 1, sub_ptr, pou.data(), dirichlet.data(), boundary->data(), &par, kmax, basis.data(), &nconv, eig.data(), &info), what);
  else
    check(ctx->handle(), ddm_geneo_basis(ctx->handle(), d0.handle(), h1, 1, sub_ptr, pou.data(), dirichlet.data(), &par, kmax, basis.data(), &nconv, eig.data(), &info), what);
  if (!info.converged)   // the reference aborts when Spectra fails (spectra.hh:149-210)
    DUNE_THROW(Dune::Exception, what << ": eigensolver did not converge in " << info.iterations << " block iterations (worst residual " << info.worst_residual << ")");
  std::vector<Vec> out(nconv, Vec(n));
  eigenvalues.assign(eig.begin(), eig.begin() + nconv);
  for (int j = 0; j < nconv; ++j)
    for (std::size_t i = 0; i < n; ++i) out[j][i] = basis[(std::size_t)j * n + i];
  return out;
}
}   // namespace ddm_hip

// ConstraintGenEOCoarseSpace (coarse_spaces.hh:394-490).  In the snapshot the constraint callback never reaches the eigensolver
// (solve_gevp(A, B, callback, ptree) begins with "(void)callback", eigensolvers/eigensolvers.hh:27-30): the basis is GenEO's.
template <class Mat, class MaskVec, class Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>>
class ConstraintGenEOCoarseSpace : public CoarseSpaceBuilder<Vec> {
public:
#if DUNE_DDM_HAVE_TASKFLOW
  ConstraintGenEOCoarseSpace(std::shared_ptr<const Mat> /*A_dir*/, std::shared_ptr<const Mat> A, std::shared_ptr<const Mat> B, std::shared_ptr<const PartitionOfUnity> pou,
                             const MaskVec& /*subdomain_boundary*/, const Dune::ParameterTree& ptree, tf::Taskflow& taskflow, const std::string& ptree_prefix = "constraint_geneo")
  {
    Dune::ParameterTree eig_ptree = ptree.sub(ptree_prefix).sub("eigensolver");
    this->setup_task = taskflow
                           .emplace([A, B, pou, eig_ptree, this] {
                             GenEOCoarseSpace<Mat, Vec> g;
                             g.setup_geneo_impl(A, B, pou, eig_ptree);
                             this->basis_ = g.get_basis();
                           })
                           .name("GenEO coarse space setup");
  }
#endif
};

// MsGFEMCoarseSpace (coarse_spaces.hh:663-831): GenEO's quotient with right-hand side D A_neu D on the interior, restricted to the
// a-harmonic functions; Dirichlet DoFs left out (zero entries in the basis).  Device: ddm_msgfem_basis.
template <class Mat, class MaskVec1, class MaskVec2, class Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>>
class MsGFEMCoarseSpace : public CoarseSpaceBuilder<Vec> {
public:
#if DUNE_DDM_HAVE_TASKFLOW
  MsGFEMCoarseSpace(std::shared_ptr<const Mat> A, std::shared_ptr<const PartitionOfUnity> pou, const MaskVec1& dirichlet_mask, const MaskVec2& subdomain_boundary_mask,
                    const Dune::ParameterTree& ptree, tf::Taskflow& taskflow, const std::string& ptree_prefix = "msgfem")   // :680-689
  {
    Dune::ParameterTree eig_ptree = ptree.sub(ptree_prefix).sub("eigensolver");
    this->setup_task = taskflow.emplace([A, pou, &dirichlet_mask, &subdomain_boundary_mask, eig_ptree, this] { setup_msgfem_impl(A, A, pou, dirichlet_mask, subdomain_boundary_mask, eig_ptree); })
                           .name("MsGFEM coarse space setup");
  }
  MsGFEMCoarseSpace(std::shared_ptr<const Mat> A_neu, std::shared_ptr<const Mat> A_dir, std::shared_ptr<const PartitionOfUnity> pou, const MaskVec1& dirichlet_mask,
                    const MaskVec2& subdomain_boundary_mask, const Dune::ParameterTree& ptree, tf::Taskflow& taskflow, const std::string& ptree_prefix = "msgfem")   // :691-700
  {
    Dune::ParameterTree eig_ptree = ptree.sub(ptree_prefix).sub("eigensolver");
    this->setup_task =
        taskflow.emplace([A_neu, A_dir, pou, &dirichlet_mask, &subdomain_boundary_mask, eig_ptree, this] { setup_msgfem_impl(A_neu, A_dir, pou, dirichlet_mask, subdomain_boundary_mask, eig_ptree); })
            .name("MsGFEM coarse space setup");
  }
#endif
  MsGFEMCoarseSpace() = default;
  const std::vector<double>& eigenvalues() const { return eigenvalues_; }
  const ddm_geneo_info& info() const { return info_; }

  template <class MV1, class MV2>
  void setup_msgfem_impl(std::shared_ptr<const Mat> A_neu, std::shared_ptr<const Mat> A_dir, std::shared_ptr<const PartitionOfUnity> pou, const MV1& dirichlet_mask, const MV2& subdomain_boundary_mask,
                         const Dune::ParameterTree& eig_ptree)   // :709-826
  {
    if (A_dir->N() != A_neu->N()) DUNE_THROW(Dune::Exception, "The two matrices must have the same size");                              // :714
    if (dirichlet_mask.N() != A_dir->N()) DUNE_THROW(Dune::Exception, "The matrix and the Dirichlet mask must have the same size");    // :716
    if (pou->size() != A_dir->N()) DUNE_THROW(Dune::Exception, "The matrix and the partition of unity must have the same size");       // :718
    const std::size_t n = A_dir->N();
    std::vector<double> w(n);
    std::vector<std::uint8_t> dir(n), bnd(n);
    for (std::size_t i = 0; i < n; ++i) {
      w[i] = (*pou)[i];
      dir[i] = dirichlet_mask[i] > 0 ? 1 : 0;
      bnd[i] = subdomain_boundary_mask[i] ? 1 : 0;
    }
    this->basis_ = ddm_hip::run_eigensolver<Vec>("ddm_msgfem_basis", *A_neu, *A_dir, w, dir, &bnd, ddm_hip::eigensolver_params(eig_ptree), eigenvalues_, info_);
  }

private:
  std::vector<double> eigenvalues_;
  ddm_geneo_info info_{};
};

// GenEORingCoarseSpace (coarse_spaces.hh:502-648): GenEO on the ring (the matrix A lives on the ring's own numbering), energy-minimal
// extension of the eigenvectors into the rest of the subdomain from one layer inside the ring.
template <class Mat, class Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>>
class GenEORingCoarseSpace : public CoarseSpaceBuilder<Vec> {
public:
#if DUNE_DDM_HAVE_TASKFLOW
  GenEORingCoarseSpace(std::shared_ptr<const Mat> A_dir, std::shared_ptr<const Mat> A, std::shared_ptr<const PartitionOfUnity> pou, const std::vector<std::size_t>& ring_to_subdomain,
                       const Dune::ParameterTree& ptree, tf::Taskflow& taskflow, const std::string& ptree_prefix = "geneo_ring")
  {
    Dune::ParameterTree eig_ptree = ptree.sub(ptree_prefix).sub("eigensolver");
    this->setup_task = taskflow.emplace([A_dir, A, pou, ring_to_subdomain, eig_ptree, this] { setup(A_dir, A, pou, ring_to_subdomain, eig_ptree); }).name("GenEO ring coarse space setup");
  }
#endif
  GenEORingCoarseSpace() = default;
  const std::vector<double>& eigenvalues() const { return eigenvalues_; }

  void setup(std::shared_ptr<const Mat> A_dir, std::shared_ptr<const Mat> A, std::shared_ptr<const PartitionOfUnity> pou, const std::vector<std::size_t>& ring_to_subdomain,
             const Dune::ParameterTree& eig_ptree)
  {
    const std::size_t n = A_dir->N(), nr = ring_to_subdomain.size();
    constexpr std::size_t none = std::numeric_limits<std::size_t>::max();
    std::vector<std::size_t> subdomain_to_ring(n, none);
    for (std::size_t i = 0; i < nr; ++i) subdomain_to_ring[ring_to_subdomain[i]] = i;
    // modified partition of unity: zero outside the ring and on its inner boundary (:541-560)
    std::vector<double> mod_pou(n);
    std::vector<std::uint8_t> on_inner_boundary(n, 0);
    std::vector<std::size_t> interior_to_subdomain, inner_ring_boundary_to_subdomain;
    for (std::size_t i = 0; i < n; ++i) {
      mod_pou[i] = (*pou)[i];
      if (subdomain_to_ring[i] == none) {
        interior_to_subdomain.push_back(i);
        mod_pou[i] = 0;
      } else {
        bool outside = false;
        ddm_hip::for_each_neighbour(*A_dir, i, [&](std::size_t j) { outside = outside || subdomain_to_ring[j] == none; });
        if (outside) {
          on_inner_boundary[i] = 1;
          inner_ring_boundary_to_subdomain.push_back(i);
          mod_pou[i] = 0;
        }
      }
    }
    // ring eigenproblem A x = lambda (D_mod A D_mod) x (:567-571); unit rows of the Dirichlet elimination are left out as in GenEOCoarseSpace
    std::vector<double> w(nr);
    std::vector<std::uint8_t> dir(nr, 0);
    for (std::size_t i = 0; i < nr; ++i) {
      w[i] = mod_pou[ring_to_subdomain[i]];
      bool diag_one = false, off = false;
      for (auto c = (*A)[i].begin(); c != (*A)[i].end(); ++c) {
        if (c.index() == i) diag_one = ((*c)[0][0] == 1.0);
        else if ((*c)[0][0] != 0.0) off = true;
      }
      dir[i] = (diag_one && !off) ? 1 : 0;
    }
    ddm_geneo_params par = ddm_hip::eigensolver_params(eig_ptree);
    par.raw = 1;
    ddm_geneo_info info{};
    auto eigenvectors_ring = ddm_hip::run_eigensolver<Vec>("ddm_geneo_basis (ring)", *A, *A, w, dir, nullptr, par, eigenvalues_, info);
    // extension from one layer inside the ring into interior + inner ring boundary (:582-598)
    std::vector<std::size_t> inside_ring_boundary_to_subdomain, extended_interior_to_subdomain;
    for (auto i : ring_to_subdomain) {
      if (on_inner_boundary[i]) continue;
      bool touches = false;
      ddm_hip::for_each_neighbour(*A_dir, i, [&](std::size_t j) { touches = touches || on_inner_boundary[j]; });
      if (touches) inside_ring_boundary_to_subdomain.push_back(i);
    }
    extended_interior_to_subdomain = interior_to_subdomain;
    extended_interior_to_subdomain.insert(extended_interior_to_subdomain.end(), inner_ring_boundary_to_subdomain.begin(), inner_ring_boundary_to_subdomain.end());
    EnergyMinimalExtension<Mat, Vec> ext(*A_dir, extended_interior_to_subdomain, inside_ring_boundary_to_subdomain);
    std::vector<Vec> data(eigenvectors_ring.size(), Vec(inside_ring_boundary_to_subdomain.size()));
    for (std::size_t k = 0; k < eigenvectors_ring.size(); ++k)
      for (std::size_t i = 0; i < inside_ring_boundary_to_subdomain.size(); ++i) data[k][i] = eigenvectors_ring[k][subdomain_to_ring[inside_ring_boundary_to_subdomain[i]]];
    auto interior_vecs = ext.extend(data);
    Vec zero(n);
    zero = 0;
    std::vector<Vec> combined(eigenvectors_ring.size(), zero);   // :603-624
    for (std::size_t k = 0; k < eigenvectors_ring.size(); ++k) {
      for (std::size_t i = 0; i < nr; ++i) combined[k][ring_to_subdomain[i]] = eigenvectors_ring[k][i];
      for (std::size_t i = 0; i < extended_interior_to_subdomain.size(); ++i) combined[k][extended_interior_to_subdomain[i]] = interior_vecs[k][i];
    }
    this->basis_ = std::move(combined);
    ddm_hip::finalize_eigenvectors(this->basis_, *pou);   // :627
  }

private:
  std::vector<double> eigenvalues_;
};

// MsGFEMRingCoarseSpace (coarse_spaces.hh:913-1163): the a-harmonically constrained eigenproblem on the ring, then the extension.
template <class Mat, class MaskVec1, class MaskVec2, class Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>>
class MsGFEMRingCoarseSpace : public CoarseSpaceBuilder<Vec> {
public:
#if DUNE_DDM_HAVE_TASKFLOW
  MsGFEMRingCoarseSpace(std::shared_ptr<const Mat> A_dir, std::shared_ptr<const Mat> A, int overlap, std::shared_ptr<const PartitionOfUnity> pou, const MaskVec1& dirichlet_mask,
                        const MaskVec2& subdomain_boundary_mask, const std::vector<std::size_t>& ring_to_subdomain, const Dune::ParameterTree& ptree, tf::Taskflow& taskflow,
                        const std::string& ptree_prefix = "msgfem_ring")
  {
    Dune::ParameterTree eig_ptree = ptree.sub(ptree_prefix).sub("eigensolver");
    this->setup_task = taskflow
                           .emplace([A_dir, A, overlap, pou, &dirichlet_mask, &subdomain_boundary_mask, ring_to_subdomain, eig_ptree, this] {
                             setup(A_dir, A, overlap, pou, dirichlet_mask, subdomain_boundary_mask, ring_to_subdomain, eig_ptree);
                           })
                           .name("MsGFEM ring coarse space setup");
  }
#endif
  MsGFEMRingCoarseSpace() = default;
  const std::vector<double>& eigenvalues() const { return eigenvalues_; }

  template <class MV1, class MV2>
  void setup(std::shared_ptr<const Mat> A_dir, std::shared_ptr<const Mat> A, int overlap, std::shared_ptr<const PartitionOfUnity> pou, const MV1& dirichlet_mask, const MV2& subdomain_boundary_mask,
             const std::vector<std::size_t>& ring_to_subdomain, const Dune::ParameterTree& eig_ptree)
  {
    if (ring_to_subdomain.empty()) DUNE_THROW(Dune::Exception, "The ring to subdomain mapping is empty, cannot build MsGFEM ring coarse space");   // :972
    const std::size_t n = A_dir->N(), nr = ring_to_subdomain.size();
    // distance to the subdomain boundary by 2 overlap + 2 in-place sweeps (:950-962)
    std::vector<int> boundary_distance(n, std::numeric_limits<int>::max() - 1);
    for (std::size_t i = 0; i < n; ++i)
      if (subdomain_boundary_mask[i] > 0) boundary_distance[i] = 0;
    for (int round = 0; round < 2 * overlap + 2; ++round)
      for (std::size_t i = 0; i < n; ++i)
        ddm_hip::for_each_neighbour(*A_dir, i, [&](std::size_t j) { boundary_distance[i] = std::min(boundary_distance[i], boundary_distance[j] + 1); });
    const int shrink = pou->get_shrink();
    const int ring_width = 2 * overlap - 2 * shrink;   // :964
    std::vector<double> w(nr);
    std::vector<std::uint8_t> dir(nr), bnd(nr);
    for (std::size_t i = 0; i < nr; ++i) {
      const auto s = ring_to_subdomain[i];
      w[i] = boundary_distance[s] >= shrink + ring_width ? 0.0 : (*pou)[s];                                      // :974-976
      dir[i] = dirichlet_mask[s] > 0 ? 1 : 0;
      bnd[i] = (subdomain_boundary_mask[s] || boundary_distance[s] == 2 * overlap) ? 1 : 0;                      // :978-1000
    }
    ddm_geneo_params par = ddm_hip::eigensolver_params(eig_ptree);
    par.raw = 1;
    ddm_geneo_info info{};
    auto eigenvectors_ring = ddm_hip::run_eigensolver<Vec>("ddm_msgfem_basis (ring)", *A, *A, w, dir, &bnd, par, eigenvalues_, info);
    std::vector<std::size_t> extension_interior_to_subdomain, extension_boundary_to_subdomain;                   // :1090-1092
    for (std::size_t i = 0; i < n; ++i)
      if (boundary_distance[i] > shrink + ring_width - 1) extension_interior_to_subdomain.push_back(i);
      else if (boundary_distance[i] == shrink + ring_width - 1) extension_boundary_to_subdomain.push_back(i);
    EnergyMinimalExtension<Mat, Vec> ext(*A_dir, extension_interior_to_subdomain, extension_boundary_to_subdomain);
    constexpr std::size_t none = std::numeric_limits<std::size_t>::max();
    std::vector<std::size_t> subdomain_to_ring(n, none);
    for (std::size_t i = 0; i < nr; ++i) subdomain_to_ring[ring_to_subdomain[i]] = i;
    std::vector<Vec> data(eigenvectors_ring.size(), Vec(extension_boundary_to_subdomain.size()));
    for (std::size_t k = 0; k < eigenvectors_ring.size(); ++k)
      for (std::size_t i = 0; i < extension_boundary_to_subdomain.size(); ++i) data[k][i] = eigenvectors_ring[k][subdomain_to_ring[extension_boundary_to_subdomain[i]]];
    auto interior_vecs = ext.extend(data);
    Vec zero(n);
    zero = 0;
    std::vector<Vec> combined(eigenvectors_ring.size(), zero);   // :1113-1133
    for (std::size_t k = 0; k < eigenvectors_ring.size(); ++k) {
      for (std::size_t i = 0; i < nr; ++i) combined[k][ring_to_subdomain[i]] = eigenvectors_ring[k][i];
      for (std::size_t i = 0; i < extension_interior_to_subdomain.size(); ++i) combined[k][extension_interior_to_subdomain[i]] = interior_vecs[k][i];
    }
    this->basis_ = std::move(combined);
    ddm_hip::finalize_eigenvectors(this->basis_, *pou);
  }

private:
  std::vector<double> eigenvalues_;
};

// HarmonicExtensionCoarseSpace (coarse_spaces.hh:1232-1266): given boundary data, extended energy-minimally and finalised.
template <class Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>>
class HarmonicExtensionCoarseSpace : public CoarseSpaceBuilder<Vec> {
public:
#if DUNE_DDM_HAVE_TASKFLOW
  template <class Mat, class MaskVec>
  HarmonicExtensionCoarseSpace(std::shared_ptr<Mat> A_ovlp, std::shared_ptr<PartitionOfUnity> pou, std::shared_ptr<std::vector<Vec>> boundary_data, const MaskVec& subdomain_boundary_mask,
                               tf::Taskflow& taskflow)
  {
    this->setup_task = taskflow.emplace([&subdomain_boundary_mask, boundary_data, A_ovlp, pou, this]() { setup(*A_ovlp, *pou, *boundary_data, subdomain_boundary_mask); });
  }
#endif
  HarmonicExtensionCoarseSpace() = default;

  template <class Mat, class MaskVec>
  void setup(const Mat& A_ovlp, const PartitionOfUnity& pou, const std::vector<Vec>& boundary_data, const MaskVec& subdomain_boundary_mask)
  {
    std::vector<std::size_t> interior_to_subdomain, boundary_to_subdomain;
    for (std::size_t i = 0; i < subdomain_boundary_mask.size(); ++i)
      if (subdomain_boundary_mask[i]) boundary_to_subdomain.push_back(i);
      else interior_to_subdomain.push_back(i);
    EnergyMinimalExtension<Mat, Vec> ext(A_ovlp, interior_to_subdomain, boundary_to_subdomain);
    auto interior = ext.extend(boundary_data);
    this->basis_.assign(boundary_data.size(), Vec(A_ovlp.N()));
    for (std::size_t k = 0; k < boundary_data.size(); ++k) {
      for (std::size_t j = 0; j < boundary_to_subdomain.size(); ++j) this->basis_[k][boundary_to_subdomain[j]] = boundary_data[k][j];
      for (std::size_t j = 0; j < interior_to_subdomain.size(); ++j) this->basis_[k][interior_to_subdomain[j]] = interior[k][j];
    }
    ddm_hip::finalize_eigenvectors(this->basis_, pou);
  }
};

// SVDCoarseSpace (coarse_spaces.hh:1268-1407): the `n` leading left singular vectors of T = D A_ii^-1 A_{i,Gamma} (keys `<prefix>.n`,
// `<prefix>.mult_pou`).  Device: ddm_svd_basis (T is never formed).  The reference also writes the singular values to
// singular_values_<rank>.txt and logs them; here they are available from singular_values().
template <class Vec = Dune::BlockVector<Dune::FieldVector<double, 1>>>
class SVDCoarseSpace : public CoarseSpaceBuilder<Vec> {
public:
#if DUNE_DDM_HAVE_TASKFLOW
  template <class Mat, class MaskVec, class MaskVec2>
  SVDCoarseSpace(std::shared_ptr<Mat> A_ovlp, std::shared_ptr<PartitionOfUnity> pou, const MaskVec& subdomain_boundary_mask, const MaskVec2& dirichlet_boundary_mask, const Dune::ParameterTree& ptree,
                 tf::Taskflow& taskflow, const std::string& ptree_prefix = "svd_coarse_space")
  {
    const int n = ptree.sub(ptree_prefix).get("n", 10);
    const bool mult_pou = ptree.sub(ptree_prefix).get("mult_pou", false);
    // masks by value, as the reference captures them (:1278)
    this->setup_task = taskflow.emplace([subdomain_boundary_mask, dirichlet_boundary_mask, A_ovlp, pou, n, mult_pou, this]() { setup(*A_ovlp, *pou, subdomain_boundary_mask, dirichlet_boundary_mask, n, mult_pou); });
  }
#endif
  SVDCoarseSpace() = default;
  const std::vector<double>& singular_values() const { return singular_values_; }

  template <class Mat, class MaskVec, class MaskVec2>
  void setup(const Mat& A_ovlp, const PartitionOfUnity& pou, const MaskVec& subdomain_boundary_mask, const MaskVec2& dirichlet_boundary_mask, int n_vectors, bool mult_pou)
  {
    auto ctx = ddm_hip::Context::get();
    const std::size_t n = A_ovlp.N();
    std::vector<double> w(n);
    std::vector<std::uint8_t> dir(n), bnd(n);
    for (std::size_t i = 0; i < n; ++i) {
      w[i] = pou[i];
      dir[i] = dirichlet_boundary_mask[i] > 0 ? 1 : 0;
      bnd[i] = subdomain_boundary_mask[i] ? 1 : 0;
    }
    ddm_hip::DeviceCsr dA(ctx, A_ovlp);
    const int64_t sub_ptr[2] = {0, (int64_t)n};
    std::vector<double> basis((std::size_t)n_vectors * n);
    singular_values_.assign(n_vectors, 0.0);
    ddm_geneo_info info{};
    ddm_hip::check(ctx->handle(), ddm_svd_basis(ctx->handle(), dA.handle(), 1, sub_ptr, w.data(), dir.data(), bnd.data(), n_vectors, mult_pou ? 1 : 0, 0.0, 0, basis.data(), singular_values_.data(), &info),
                   "ddm_svd_basis");
    if (!info.converged) DUNE_THROW(Dune::Exception, "SVD coarse space: eigensolver did not converge in " << info.iterations << " block iterations");
    this->basis_.assign(n_vectors, Vec(n));
    for (int j = 0; j < n_vectors; ++j)
      for (std::size_t i = 0; i < n; ++i) this->basis_[j][i] = basis[(std::size_t)j * n + i];
  }

private:
  std::vector<double> singular_values_;
};
