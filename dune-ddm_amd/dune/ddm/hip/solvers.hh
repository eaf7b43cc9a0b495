// Device-resident solvers behind Dune::InverseOperator -- the two plugin points of the reference:
//
// (1) the subdomain / coarse solver selected by string through the dune-istl solver factory
//     (dune/ddm/schwarz.hh:85-92, galerkin_preconditioner.hh:338-346); in-tree precedent for registering a new one:
//     DUNE_REGISTER_DIRECT_SOLVER("strumpack", Dune::StrumpackCreator()) (dune/ddm/strumpack.hh:95-122).
//     Dune::HipSubdomainSolver<M> wraps the library's local factor solvers (ILU(0) in natural order; sparse Cholesky / L U with
//     host factorisation and device triangular solves) for a flattened BCRSMatrix; registered under "hip_ilu0", "hip_cholesky",
//     "hip_lu" when the dune-istl factory macros are visible.
// (2) the outer Krylov solver (examples/poisson.cc:311-319 obtains it from the same factory): Dune::HipCGSolver /
//     Dune::HipRestartedGMResSolver run the WHOLE loop on the device (ddm_cg_solve / ddm_gmres_solve: dune-istl's recurrences,
//     SURVEY.md 3.2) -- one upload of x and b, one download of x and the defect, instead of two PCIe copies of n_o doubles per
//     virtual apply() when dune-istl's own host solvers drive the adaptors.
#pragma once

#include <memory>
#include <string>

#include <dune/common/exceptions.hh>
#include <dune/common/parametertree.hh>
#include <dune/istl/operators.hh>
#include <dune/istl/preconditioner.hh>
#include <dune/istl/solver.hh>

#include "backend.hh"
#include "combined_preconditioner.hh"

namespace Dune {

template <class M, class X = BlockVector<FieldVector<double, 1>>>
class HipSubdomainSolver : public InverseOperator<X, X> {
public:
  // kind: "ilu0" | "cholesky" | "lu" | "direct" (Cholesky if the values are symmetric, else L U)
  explicit HipSubdomainSolver(const M& A, const std::string& kind = "ilu0") : ctx(ddm_hip::Context::get()), dA(ctx, A), n(A.N()), dd(ctx, A.N()), dx(ctx, A.N())
  {
    const int64_t bp[2] = {0, (int64_t)n};
    if (kind == "ilu0") ddm_hip::check(ctx->handle(), ddm_ilu0_create(ctx->handle(), dA.handle(), 1, bp, &F), "ddm_ilu0_create");
    else if (kind == "cholesky") ddm_hip::check(ctx->handle(), ddm_direct_create(ctx->handle(), dA.handle(), 1, bp, 0, 0.0, &F), "ddm_direct_create");
    else if (kind == "lu") ddm_hip::check(ctx->handle(), ddm_direct_create(ctx->handle(), dA.handle(), 1, bp, 1, 0.0, &F), "ddm_direct_create");
    else if (kind == "direct") {
      if (ddm_direct_create(ctx->handle(), dA.handle(), 1, bp, 0, 0.0, &F) != DDM_OK)
        ddm_hip::check(ctx->handle(), ddm_direct_create(ctx->handle(), dA.handle(), 1, bp, 1, 0.0, &F), "ddm_direct_create");
    } else DUNE_THROW(NotImplemented, "Unknown device subdomain solver '" + kind + "'");
  }
  ~HipSubdomainSolver() override { ddm_ilu0_destroy(F); }
  SolverCategory::Category category() const override { return SolverCategory::sequential; }
  void apply(X& x, X& b, InverseOperatorResult& res) override
  {
    dd.upload(b);
    ddm_hip::check(ctx->handle(), ddm_ilu0_solve(ctx->handle(), F, dd.data(), dx.data()), "ddm_ilu0_solve");
    dx.download(x);
    int st = 0;
    ddm_hip::check(ctx->handle(), ddm_ilu0_status(ctx->handle(), F, &st), "ddm_ilu0_status");
    res.iterations = 1;
    res.converged = st == 0;
  }
  void apply(X& x, X& b, [[maybe_unused]] double reduction, InverseOperatorResult& res) override { apply(x, b, res); }
  ddm_ilu0* handle() const { return F; }

private:
  std::shared_ptr<ddm_hip::Context> ctx;
  ddm_hip::DeviceCsr dA;
  std::size_t n;
  ddm_hip::DeviceVector dd, dx;
  ddm_ilu0* F = nullptr;
};

#ifdef DUNE_REGISTER_DIRECT_SOLVER
// factory registration, mirroring StrumpackCreator (dune/ddm/strumpack.hh:95-122)
template <int KIND>
struct HipSubdomainSolverCreator {
  template <typename TL, typename M>
  std::shared_ptr<Dune::InverseOperator<typename Dune::TypeListElement<1, TL>::type, typename Dune::TypeListElement<2, TL>::type>> operator()(
      TL /*tl*/, const M& mat, const Dune::ParameterTree& /*config*/, std::enable_if_t<std::is_same_v<typename M::field_type, double>, int> = 0) const
  {
    return std::make_shared<Dune::HipSubdomainSolver<M>>(mat, KIND == 0 ? "ilu0" : (KIND == 1 ? "cholesky" : "lu"));
  }
  template <typename TL, typename M>
  std::shared_ptr<Dune::InverseOperator<typename Dune::TypeListElement<1, TL>::type, typename Dune::TypeListElement<2, TL>::type>> operator()(
      TL /*tl*/, const M& /*mat*/, const Dune::ParameterTree& /*config*/, std::enable_if_t<!std::is_same_v<typename M::field_type, double>, int> = 0) const
  {
    DUNE_THROW(UnsupportedType, "Unsupported type in HipSubdomainSolver (double only)");
  }
};
DUNE_REGISTER_DIRECT_SOLVER("hip_ilu0", Dune::HipSubdomainSolverCreator<0>());
DUNE_REGISTER_DIRECT_SOLVER("hip_cholesky", Dune::HipSubdomainSolverCreator<1>());
DUNE_REGISTER_DIRECT_SOLVER("hip_lu", Dune::HipSubdomainSolverCreator<2>());
#endif

// Outer Krylov loops on the device.  op must be this directory's NonOverlappingOperator, prec its CombinedPreconditioner.
template <class X>
class HipKrylovSolverBase : public InverseOperator<X, X> {
public:
  HipKrylovSolverBase(std::shared_ptr<LinearOperator<X, X>> op_, std::shared_ptr<Preconditioner<X, X>> prec_, double reduction, int maxit, int verbose)
      : op(std::move(op_)), prec(std::move(prec_)), reduction_(reduction), maxit_(maxit), verbose_(verbose)
  {
    dop = dynamic_cast<ddm_hip::DeviceOperator*>(op.get());
    cprec = dynamic_cast<CombinedPreconditioner<X>*>(prec.get());
    if (!dop || !cprec) DUNE_THROW(NotImplemented, "the device Krylov solvers need the device NonOverlappingOperator and CombinedPreconditioner");
  }
  SolverCategory::Category category() const override { return op->category(); }
  void apply(X& x, X& b, InverseOperatorResult& res) override { apply(x, b, reduction_, res); }
  void apply(X& x, X& b, double reduction, InverseOperatorResult& res) override
  {
    auto ctx = cprec->context();
    const std::size_t n = b.N();
    if (!dx || dx->size() != n) {
      dx = std::make_unique<ddm_hip::DeviceVector>(ctx, n);
      db = std::make_unique<ddm_hip::DeviceVector>(ctx, n);
    }
    prec->pre(x, b);
    dx->upload(x);   // the only host -> device copies of the solve
    db->upload(b);
    ddm_solve_result r{};
    ddm_hip::check(ctx->handle(), solve(ctx->handle(), dop->op_handle(), cprec->handle(n), dx->data(), db->data(), reduction, &r), "device Krylov solve");
    dx->download(x);   // the only device -> host copies
    db->download(b);   // dune-istl leaves the defect in b
    prec->post(x);
    res.clear();
    res.iterations = r.iterations;
    res.converged = r.converged != 0;
    res.reduction = r.reduction;
    res.elapsed = r.elapsed_s;
    res.conv_rate = r.iterations > 0 ? std::pow(r.reduction, 1.0 / r.iterations) : 0.0;
    if (verbose_ > 0) std::printf("=== device Krylov solve: %d iterations, reduction %.3e, %.3f s\n", r.iterations, r.reduction, r.elapsed_s);
  }

protected:
  virtual int solve(ddm_ctx* ctx, ddm_op* o, ddm_combined* p, double* x, double* b, double reduction, ddm_solve_result* r) = 0;
  std::shared_ptr<LinearOperator<X, X>> op;
  std::shared_ptr<Preconditioner<X, X>> prec;
  ddm_hip::DeviceOperator* dop = nullptr;
  CombinedPreconditioner<X>* cprec = nullptr;
  double reduction_;
  int maxit_, verbose_;
  std::unique_ptr<ddm_hip::DeviceVector> dx, db;
};

// [solver] type = cgsolver (examples/poisson.ini:12-17): dune-istl CGSolver::apply
template <class X>
class HipCGSolver : public HipKrylovSolverBase<X> {
public:
  HipCGSolver(std::shared_ptr<LinearOperator<X, X>> op, std::shared_ptr<Preconditioner<X, X>> prec, double reduction, int maxit, int verbose = 0)
      : HipKrylovSolverBase<X>(std::move(op), std::move(prec), reduction, maxit, verbose) {}
  HipCGSolver(std::shared_ptr<LinearOperator<X, X>> op, std::shared_ptr<Preconditioner<X, X>> prec, const ParameterTree& cfg)
      : HipCGSolver(std::move(op), std::move(prec), cfg.get("reduction", 1e-8), cfg.get("maxit", 1000), cfg.get("verbose", 0)) {}

protected:
  int solve(ddm_ctx* ctx, ddm_op* o, ddm_combined* p, double* x, double* b, double reduction, ddm_solve_result* r) override
  {
    return ddm_cg_solve(ctx, o, p, x, b, reduction, this->maxit_, 0, nullptr, r);
  }
};

// [solver] type = restartedgmressolver (default of TwoLevelSchwarzSolver, dune/ddm/twolevel_schwarz.hh:121-130)
template <class X>
class HipRestartedGMResSolver : public HipKrylovSolverBase<X> {
public:
  HipRestartedGMResSolver(std::shared_ptr<LinearOperator<X, X>> op, std::shared_ptr<Preconditioner<X, X>> prec, double reduction, int restart, int maxit, int verbose = 0)
      : HipKrylovSolverBase<X>(std::move(op), std::move(prec), reduction, maxit, verbose), restart_(restart) {}
  HipRestartedGMResSolver(std::shared_ptr<LinearOperator<X, X>> op, std::shared_ptr<Preconditioner<X, X>> prec, const ParameterTree& cfg)
      : HipRestartedGMResSolver(std::move(op), std::move(prec), cfg.get("reduction", 1e-8), cfg.get("restart", 30), cfg.get("maxit", 1000), cfg.get("verbose", 0)) {}

protected:
  int solve(ddm_ctx* ctx, ddm_op* o, ddm_combined* p, double* x, double* b, double reduction, ddm_solve_result* r) override
  {
    return ddm_gmres_solve(ctx, o, p, x, b, reduction, this->maxit_, restart_, nullptr, r);
  }
  int restart_;
};

// [solver] type = bicgstabsolver: dune-istl BiCGSTABSolver::apply
template <class X>
class HipBiCGSTABSolver : public HipKrylovSolverBase<X> {
public:
  HipBiCGSTABSolver(std::shared_ptr<LinearOperator<X, X>> op, std::shared_ptr<Preconditioner<X, X>> prec, double reduction, int maxit, int verbose = 0)
      : HipKrylovSolverBase<X>(std::move(op), std::move(prec), reduction, maxit, verbose) {}
  HipBiCGSTABSolver(std::shared_ptr<LinearOperator<X, X>> op, std::shared_ptr<Preconditioner<X, X>> prec, const ParameterTree& cfg)
      : HipBiCGSTABSolver(std::move(op), std::move(prec), cfg.get("reduction", 1e-8), cfg.get("maxit", 1000), cfg.get("verbose", 0)) {}

protected:
  int solve(ddm_ctx* ctx, ddm_op* o, ddm_combined* p, double* x, double* b, double reduction, ddm_solve_result* r) override
  {
    return ddm_bicgstab_solve(ctx, o, p, x, b, reduction, this->maxit_, nullptr, nullptr, r);
  }
};

// getSolverFromFactory(op, solver_subtree, prec) for the device solvers (examples/poisson.cc:311-316)
template <class X>
std::shared_ptr<InverseOperator<X, X>> getHipSolver(std::shared_ptr<LinearOperator<X, X>> op, const ParameterTree& cfg, std::shared_ptr<Preconditioner<X, X>> prec)
{
  const auto type = cfg.get("type", std::string("cgsolver"));
  if (type == "cgsolver") return std::make_shared<HipCGSolver<X>>(std::move(op), std::move(prec), cfg);
  if (type == "restartedgmressolver") return std::make_shared<HipRestartedGMResSolver<X>>(std::move(op), std::move(prec), cfg);
  if (type == "bicgstabsolver") return std::make_shared<HipBiCGSTABSolver<X>>(std::move(op), std::move(prec), cfg);
  DUNE_THROW(NotImplemented, "solver type '" + type + "' has no device implementation (cgsolver, restartedgmressolver, bicgstabsolver)");
}

}  // namespace Dune
