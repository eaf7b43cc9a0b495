// MI355X drop-in for the coarse-space builders of dune/ddm/coarsespaces/coarse_spaces.hh that lie on the GenEO path:
//   CoarseSpaceBuilder<Vec>   (:219-256)  get_basis() / size() / get_setup_task()
//   GenEOCoarseSpace<Mat,Vec> (:286-331)  ctor (A, B, pou, ptree, taskflow, prefix = "geneo"); eigenproblem on the device
//   POUCoarseSpace<Vec>       (:1175-1231) pou / ||pou||_2, or POU-scaled template vectors
// Same names, constructor signatures, ParameterTree keys (`<prefix>.eigensolver.{nev, tolerance, shift, threshold, nev_max}`,
// dune/ddm/eigensolvers/eigensolver_params.hh:8-62) and exception texts.  The eigensolver is ddm_geneo_basis (C ABI): block
// method on the device instead of Spectra's single-vector Lanczos; `ncv`, `maxit`, `seed`, `blocksize` are parsed by the
// reference but have no meaning for it (the reference's driver hard-codes maxit = 100, spectra.hh:137).
// As in the reference the returned vectors are v <- D v / ||D v||_2 (finalize_eigenvectors, :52-61); the caller zeroes the
// Dirichlet entries (examples/poisson.cc:235-238).  Rows of A that the symmetric Dirichlet elimination turned into unit rows
// (examples/pdelab_helper.hh:33-46) are recognised from the matrix and their decoupled unit modes are not returned (csrc/geneo.hpp).
#pragma once

#include <cmath>
#include <cstdint>
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
    const auto type = eig_ptree.get("type", std::string("Spectra"));
    if (type != "Spectra") DUNE_THROW(Dune::NotImplemented, "Unknown eigensolver type '" + type + "'");   // eigensolver_params.hh:35
    auto ctx = ddm_hip::Context::get();
    ddm_geneo_params par;
    ddm_geneo_params_default(&par);
    par.nev = eig_ptree.get("nev", par.nev);
    par.nev_max = eig_ptree.hasKey("nev_max") ? par.nev : 2 * par.nev;   // sic: the key `nev_max` overwrites ncv in the reference (:23), nev_max stays 2 nev unless unset
    par.tolerance = eig_ptree.get("tolerance", par.tolerance);
    par.shift = eig_ptree.get("shift", par.shift);
    par.threshold = eig_ptree.get("threshold", par.threshold);
    par.verbose = eig_ptree.get("verbose", 0);
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
  // :1211-1231: POU-scaled template vectors (TwoLevelSchwarzSolver uses 1, x, y, xy: twolevel_schwarz.hh:68-107)
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
