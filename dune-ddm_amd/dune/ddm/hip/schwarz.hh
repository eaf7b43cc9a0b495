// MI355X drop-in for dune/ddm/schwarz.hh (SchwarzPreconditioner), see nonoverlapping_operator.hh.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include <dune/common/parallel/interface.hh>
#include <dune/common/parametertree.hh>
#include <dune/istl/preconditioner.hh>
#include <dune/istl/solver.hh>
#include <dune/istl/solvercategory.hh>

#include <dune/ddm/pou.hh>

#if DUNE_DDM_HAVE_TASKFLOW
#include <taskflow/taskflow.hpp>
#endif

#include "backend.hh"

enum class SchwarzType : std::uint8_t { Standard, Restricted };

// PartitionOfUnity is the reference's own class (dune/ddm/pou.hh, CPU setup code that is used
// unchanged); only size() and operator[] are needed here.
template <class Mat, class Vec, class Communication>
class SchwarzPreconditioner : public Dune::Preconditioner<Vec, Vec>, public ddm_hip::DeviceLevel {
  using Solver = Dune::InverseOperator<Vec, Vec>;   // schwarz.hh:57

  // What getSolver() hands out (schwarz.hh:155): the local solver `A_dir^-1` on the OVERLAPPING index set as a
  // Dune::InverseOperator.  It is a view of the factor inside the ddm_schwarz object (ddm_schwarz_local_solver), not a second
  // factorisation; apply() = one upload, ddm_ilu0_solve, one download.
  class LocalSolverView : public Solver {
  public:
    explicit LocalSolverView(SchwarzPreconditioner& owner) : P(owner) {}
    Dune::SolverCategory::Category category() const override { return Dune::SolverCategory::sequential; }
    void apply(Vec& x, Vec& b, Dune::InverseOperatorResult& res) override
    {
      const std::size_t n = P.Aovlp->N();
      if (b.N() != n || x.N() != n) DUNE_THROW(Dune::InvalidStateException, "local solver: vectors must live on the overlapping index set (size " << n << ")");
      if (!P.S) P.create(P.novlp_comm ? P.novlp_comm->indexSet().size() : n);
      if (!bd) {
        bd = std::make_unique<ddm_hip::DeviceVector>(P.ctx, n);
        xd = std::make_unique<ddm_hip::DeviceVector>(P.ctx, n);
      }
      bd->upload(b);
      ddm_ilu0* F = ddm_schwarz_local_solver(P.S);
      ddm_hip::check(P.ctx->handle(), ddm_ilu0_solve(P.ctx->handle(), F, bd->data(), xd->data()), "ddm_ilu0_solve");
      xd->download(x);
      int st = 0;
      ddm_hip::check(P.ctx->handle(), ddm_ilu0_status(P.ctx->handle(), F, &st), "ddm_ilu0_status");
      res.clear();
      res.iterations = 1;
      res.converged = st == 0;
    }
    void apply(Vec& x, Vec& b, [[maybe_unused]] double reduction, Dune::InverseOperatorResult& res) override { apply(x, b, res); }

  private:
    SchwarzPreconditioner& P;
    std::unique_ptr<ddm_hip::DeviceVector> bd, xd;
  };

public:
  // reference ctor: schwarz.hh:73-94
  SchwarzPreconditioner(std::shared_ptr<Mat> Aovlp, std::shared_ptr<Communication> comm, std::shared_ptr<PartitionOfUnity> pou,
                           const Dune::ParameterTree& ptree, const std::string& subtree_name = "schwarz",
                           const std::string& solver_subtree_name = "subdomain_solver")
      : Aovlp(std::move(Aovlp)), comm(std::move(comm)), pou(std::move(pou)), ctx(ddm_hip::Context::get())
  {
    const auto& subtree = ptree.sub(subtree_name);
    auto type_string = subtree.get("type", std::string("restricted"));
    if (type_string == "restricted") type = SchwarzType::Restricted;
    else if (type_string == "standard") type = SchwarzType::Standard;
    else DUNE_THROW(Dune::NotImplemented, "Unknown Schwarz type '" + type_string + "'");   // :83
    const auto& solver_subtree = subtree.sub(solver_subtree_name);
    if (not solver_subtree.hasKey("type"))
      DUNE_THROW(Dune::Exception, "You must specify the solver in the subtree " << subtree_name << "." << solver_subtree_name << " using the key 'type'");   // :89-91
    // the factory key of the reference (schwarz.hh:85-92; shipped: cholmod / umfpack, examples/poisson.ini:23): ILU(0) or the
    // library's sparse direct solvers (host factorisation, device triangular solves) -- ddm_schwarz_create_ex validates the name
    solver = solver_subtree.get("type", std::string(""));
    if (solver == "hip_ilu0") solver = "ilu0";
    ctx->require(this->comm->communicator());
    // size checks of init() (:186-193)
    if (this->comm->indexSet().size() != this->Aovlp->N())
      DUNE_THROW(Dune::InvalidStateException, "Remote indices size (" << this->comm->indexSet().size() << ") does not match overlapping matrix size (" << this->Aovlp->N() << ").");
    if (this->pou && this->pou->size() != this->Aovlp->N())
      DUNE_THROW(Dune::InvalidStateException, "Partition of unity size (" << this->pou->size() << ") does not match overlapping matrix size (" << this->Aovlp->N() << ").");
    dA = std::make_unique<ddm_hip::DeviceCsr>(ctx, *this->Aovlp);
    typename Communication::OwnerSet owner;
    typename Communication::AllSet all;
    typename Communication::OwnerCopySet oc;
    Dune::Interface copy_if, add_if;
    copy_if.build(this->comm->remoteIndices(), owner, all);   // copyOwnerToAll
    add_if.build(this->comm->remoteIndices(), oc, oc);        // addOwnerCopyToOwnerCopy
    h_copy = std::make_unique<ddm_hip::Halo>(ctx, 2, 0, copy_if);
    h_add = std::make_unique<ddm_hip::Halo>(ctx, 3, 1, add_if);
  }
  ~SchwarzPreconditioner() override { ddm_schwarz_destroy(S); }

  Dune::SolverCategory::Category category() const override { return Dune::SolverCategory::nonoverlapping; }
  void pre(Vec&, Vec&) override {}
  // apply() has no error return (the reference discards the local solver's InverseOperatorResult, schwarz.hh:131): a local solve
  // that gave up (single-launch engine, GPU shared with another process) makes the NEXT apply throw (ddm_schwarz_apply looks at
  // the status word in pinned host memory on entry, no synchronisation) and is reported here at the latest
  void post(Vec&) override
  {
    if (S) ddm_hip::check(ctx->handle(), ddm_schwarz_status(ctx->handle(), S), "SchwarzPreconditioner::post");
  }

  void apply(Vec& x, const Vec& d) override   // :115-149
  {
    if (!S) create(d.N());
    dd->upload(d);
    ddm_hip::check(ctx->handle(), ddm_schwarz_apply(ctx->handle(), S, dx->data(), dd->data()), "ddm_schwarz_apply");
    dx->download(x);
  }
  ddm_schwarz* schwarz_handle(std::size_t n_novlp) override { return handle(n_novlp); }
  ddm_schwarz* handle(std::size_t n_novlp)
  {
    if (!S) create(n_novlp);
    return S;
  }
  ddm_hip::Halo& copyHalo() { return *h_copy; }
  ddm_hip::Halo& addHalo() { return *h_add; }

  // reference schwarz.hh:155: reference to the local subdomain solver
  Solver& getSolver()
  {
    if (!solver_view) solver_view = std::make_unique<LocalSolverView>(*this);
    return *solver_view;
  }
#if DUNE_DDM_HAVE_TASKFLOW
  // reference schwarz.hh:162 (a default-constructed tf::Task there, too: the constructor does the whole setup)
  tf::Task& get_setup_task() { return setup_task; }
#endif
  // reference schwarz.hh:165: public data member, set by TwoLevelSchwarzSolver (twolevel_schwarz.hh:109); here it also tells the
  // device object the non-overlapping size before the first defect vector arrives
  std::shared_ptr<Communication> novlp_comm;

private:
  // The non-overlapping size is only known from the first defect vector ("extend" is a prefix copy,
  // SURVEY.md A.1), so the device object (incl. the ILU(0) factorisation) is built on first use.
  void create(std::size_t n_novlp)
  {
    const std::size_t n = Aovlp->N();
    std::vector<int32_t> ext(n);
    for (std::size_t i = 0; i < n; ++i) ext[i] = i < n_novlp ? (int32_t)i : -1;
    std::vector<double> w;
    if (pou) {
      w.resize(n);
      for (std::size_t i = 0; i < n; ++i) w[i] = (*pou)[i];
    }
    const int64_t bp[2] = {0, (int64_t)n};
    ddm_hip::check(ctx->handle(),
                   ddm_schwarz_create_ex(ctx->handle(), dA->handle(), 1, bp, (int64_t)n_novlp, ext.data(), pou ? w.data() : nullptr,
                                         type == SchwarzType::Restricted ? 1 : 0, solver.c_str(), h_copy->handle(), h_add->handle(), &S),
                   "ddm_schwarz_create");
    dd = std::make_unique<ddm_hip::DeviceVector>(ctx, n_novlp);
    dx = std::make_unique<ddm_hip::DeviceVector>(ctx, n_novlp);
  }

  std::shared_ptr<Mat> Aovlp;
  std::shared_ptr<Communication> comm;
  std::shared_ptr<PartitionOfUnity> pou;
  std::shared_ptr<ddm_hip::Context> ctx;
  SchwarzType type;
  std::string solver;
  std::unique_ptr<ddm_hip::DeviceCsr> dA;
  std::unique_ptr<ddm_hip::Halo> h_copy, h_add;
  std::unique_ptr<ddm_hip::DeviceVector> dd, dx;
  std::unique_ptr<LocalSolverView> solver_view;
#if DUNE_DDM_HAVE_TASKFLOW
  tf::Task setup_task;
#endif
  ddm_schwarz* S = nullptr;
};
