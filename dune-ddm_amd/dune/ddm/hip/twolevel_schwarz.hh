// MI355X adaptor for the PDELab linear-solver backend of the reference (class TwoLevelSchwarzSolver, dune/ddm/twolevel_schwarz.hh:27-174;
// handed to StationaryLinearProblemSolver / Newton by examples/convectiondiffusiondg.cc:75-78 and nonlinearpoisson.cc:151-154).
//
// Layout of this file
//   ddm_hip::krylov_settings        the "solver" sub-tree of one solve, with the backend's defaults (reference :119-131)
//   ddm_hip::TwoLevelSchwarzCore    what happens on EVERY apply(): the two levels are rebuilt around the current matrix, the Krylov
//                                   loop runs on the device, one upload and one download per solve.  Native dune-istl types only, so it
//                                   compiles and is GPU-tested without PDELab (tests/cpp/twolevel_adaptor.cc).
//   ddm_hip::OverlapObjects         (HAVE_DUNE_PDELAB) the overlapping matrix / communication / partition of unity of a rank, produced
//                                   by the REFERENCE's own host setup headers (overlap_extension.hh, datahandles.hh, pou.hh): that layer
//                                   sits before the hot path and is used as it is in a DUNE build -- INTEGRATION.md, "PDELab backend".
//   TwoLevelSchwarzSolver           (HAVE_DUNE_PDELAB) the class PDELab sees: constructor, apply, norm and the result storage of the
//                                   reference (:38-56, :58, :149); owns an OverlapObjects and a TwoLevelSchwarzCore and forwards to them.
#pragma once

#include <array>
#include <cstddef>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include <dune/common/parametertree.hh>
#include <dune/istl/solver.hh>

#include <dune/ddm/pou.hh>

#include "coarse_spaces.hh"
#include "combined_preconditioner.hh"
#include "galerkin_preconditioner.hh"
#include "nonoverlapping_operator.hh"
#include "schwarz.hh"
#include "solvers.hh"

namespace ddm_hip {

// Settings of the outer Krylov solver for one apply(): the "solver" sub-tree when the user wrote one, otherwise restarted GMRES(30)
// with at most 1000 iterations; only the root rank may print; the reduction PDELab asks for is passed down as a key as well.
inline Dune::ParameterTree krylov_settings(const Dune::ParameterTree& backend_cfg, double reduction, bool root_rank)
{
  Dune::ParameterTree s;
  if (backend_cfg.hasSub("solver")) s = backend_cfg.sub("solver");
  else {
    s["type"] = "restartedgmressolver";
    s["restart"] = "30";
    s["maxit"] = "1000";
  }
  const std::string wanted = s.get("verbose", std::string("0"));
  s["verbose"] = root_rank ? wanted : std::string("0");
  s["reduction"] = std::to_string(reduction);
  return s;
}

template <class NativeMat, class NativeVec, class Communication>
class TwoLevelSchwarzCore {
public:
  using FineLevel = SchwarzPreconditioner<NativeMat, NativeVec, Communication>;
  using CoarseLevel = GalerkinPreconditioner<NativeVec, Communication>;
  using Operator = NonOverlappingOperator<NativeMat, NativeVec, NativeVec, Communication>;
  using Levels = CombinedPreconditioner<NativeVec>;

  TwoLevelSchwarzCore(std::shared_ptr<Communication> nonoverlapping, const Dune::ParameterTree& backend_cfg) : novlp(std::move(nonoverlapping)), cfg(backend_cfg) {}

  // The rank's overlapping objects; they outlive the solves (the matrix VALUES may be refreshed in place between two calls).
  void set_overlapping(std::shared_ptr<NativeMat> matrix, std::shared_ptr<Communication> communication, std::shared_ptr<PartitionOfUnity> partition,
                       std::vector<NativeVec> templates_on_overlap)
  {
    for (const auto& t : templates_on_overlap)
      if (t.N() != matrix->N()) DUNE_THROW(Dune::Exception, "Template vectors must match size of matrix");
    ovlp = Overlapping{std::move(matrix), std::move(communication), std::move(partition), std::move(templates_on_overlap)};
  }
  bool has_overlapping() const { return (bool)ovlp.comm; }
  const std::shared_ptr<NativeMat>& overlapping_matrix() const { return ovlp.matrix; }
  const std::shared_ptr<Communication>& overlapping_communication() const { return ovlp.comm; }

  // One linear solve A z = d to the given reduction.  A is the additive non-overlapping matrix; d is made consistent in place
  // (every holder of a shared DoF ends with the sum) because operator, scalar product and both levels work on consistent vectors,
  // and the solver then leaves the final defect in it.
  Dune::InverseOperatorResult solve(std::shared_ptr<NativeMat> A, NativeVec& z, NativeVec& d, double reduction)
  {
    if (!has_overlapping()) DUNE_THROW(Dune::InvalidStateException, "TwoLevelSchwarzCore::solve before set_overlapping");
    auto op = std::make_shared<Operator>(std::move(A), novlp);
    auto levels = build_levels(op);
    const bool root = ovlp.comm->communicator().rank() == 0;
    auto krylov = Dune::getHipSolver<NativeVec>(op, krylov_settings(cfg, reduction, root), levels);
    novlp->addOwnerCopyToAll(d, d);
    Dune::InverseOperatorResult outcome;
    krylov->apply(z, d, reduction, outcome);
    return outcome;
  }

  // Global 2-norm of an ADDITIVE vector: summed over the holders first, then counted once per owner.
  double norm(const NativeVec& additive) const
  {
    NativeVec consistent = additive;
    novlp->addOwnerCopyToOwnerCopy(consistent, consistent);
    return novlp->norm(consistent);
  }

  // the levels of the last solve (inspection / tests)
  std::shared_ptr<FineLevel> fine;
  std::shared_ptr<CoarseLevel> coarse;

private:
  struct Overlapping {
    std::shared_ptr<NativeMat> matrix;
    std::shared_ptr<Communication> comm;
    std::shared_ptr<PartitionOfUnity> pou;
    std::vector<NativeVec> templates;
  };

  // "fine" = Schwarz level on the overlapping matrix (it also needs the non-overlapping communication for its restriction),
  // "coarse" = Galerkin level on the partition-of-unity-weighted template vectors; how the two are combined is the "mode" key of the
  // backend's own sub-tree, which is why the combination is configured with an empty sub-tree name.
  std::shared_ptr<Levels> build_levels(const std::shared_ptr<Operator>& op)
  {
    fine = std::make_shared<FineLevel>(ovlp.matrix, ovlp.comm, ovlp.pou, cfg, "fine");
    fine->novlp_comm = novlp;
    const POUCoarseSpace<NativeVec> space(ovlp.templates, *ovlp.pou);
    coarse = std::make_shared<CoarseLevel>(*ovlp.matrix, space.get_basis(), ovlp.comm, cfg, "coarse");
    auto both = std::make_shared<Levels>(cfg, "");
    both->set_op(op);
    both->add(fine);
    both->add(coarse);
    return both;
  }

  std::shared_ptr<Communication> novlp;
  Dune::ParameterTree cfg;
  Overlapping ovlp;
};

}  // namespace ddm_hip

#if HAVE_DUNE_PDELAB
#include <dune/common/ftraits.hh>
#include <dune/common/parallel/interface.hh>
#include <dune/common/parallel/variablesizecommunicator.hh>
#include <dune/istl/owneroverlapcopy.hh>
#include <dune/pdelab/backend/interface.hh>
#include <dune/pdelab/backend/solver.hh>
#include <dune/pdelab/constraints/common/constraints.hh>
#include <dune/pdelab/gridfunctionspace/interpolate.hh>

// the reference's host setup layer (SURVEY 8 f-1 restates it in dune-ddm_amd/setup_dist.py for the tests; a DUNE build uses the originals)
#include <dune/ddm/datahandles.hh>
#include <dune/ddm/overlap_extension.hh>
#include <dune/ddm/pdelab_helper.hh>

namespace ddm_hip {

// Overlapping objects of one rank, made by the reference's setup functions.  create() is the expensive first-call path (index-set
// extension by `overlap` layers, sparsity pattern + values of the overlapping matrix, partition of unity); refresh() re-sends the
// values only, for the later Newton steps, on the pattern and the communication plan create() left behind.
template <class NativeMat, class NativeVec, class Communication>
class OverlapObjects {
public:
  std::shared_ptr<NativeMat> matrix;
  std::shared_ptr<Communication> comm;
  std::shared_ptr<PartitionOfUnity> pou;

  bool ready() const { return (bool)comm; }

  void create(const Communication& nonoverlapping, const NativeMat& A, const Dune::ParameterTree& backend_cfg)
  {
    const int layers = backend_cfg.get("overlap", 1);
    comm = make_overlapping_communication(nonoverlapping, A, layers).first;
    const typename Communication::AllSet everyone;
    plan.build(comm->remoteIndices(), everyone, everyone);
    exchange = std::make_unique<Dune::VariableSizeCommunicator<>>(plan);
    CreateMatrixDataHandle pattern(A, comm->indexSet());
    exchange->forward(pattern);
    matrix = std::make_shared<NativeMat>(pattern.getOverlappingMatrix());
    send_values(A);
    pou = std::make_shared<PartitionOfUnity>(*matrix, *comm, backend_cfg.sub("pou"), layers);
  }

  void refresh(const NativeMat& A)
  {
    *matrix = 0;
    send_values(A);
  }

  // a vector on the rank's own DoFs continued onto the overlap: own part copied, the rest filled with the owners' values
  NativeVec extended(const NativeVec& own) const
  {
    NativeVec e(matrix->N());
    e = 0;
    for (std::size_t k = 0; k < own.N(); ++k) e[k] = own[k];
    comm->copyOwnerToAll(e, e);
    return e;
  }

private:
  void send_values(const NativeMat& A)
  {
    AddMatrixDataHandle values(A, *matrix, comm->indexSet());
    exchange->forward(values);
  }
  Dune::Interface plan;
  std::unique_ptr<Dune::VariableSizeCommunicator<>> exchange;
};

}  // namespace ddm_hip

template <class Mat, class Vec>
class TwoLevelSchwarzSolver : public Dune::PDELab::LinearResultStorage {
  using NativeMat = Dune::PDELab::Backend::Native<Mat>;
  using NativeVec = Dune::PDELab::Backend::Native<Vec>;
  using Communication = Dune::OwnerOverlapCopyCommunication<std::size_t, int>;
  using Real = typename Dune::template FieldTraits<typename Vec::ElementType>::real_type;

public:
  // Same arguments as the reference's constructor.  The coarse space is spanned by the bilinear monomials 1, x, y, xy of the
  // function space, with constrained DoFs set to zero; they are kept as native vectors until the first apply() knows the overlap.
  template <class GFS, class CC>
  explicit TwoLevelSchwarzSolver(const GFS& gfs, const CC& cc, const Dune::ParameterTree& ptree, const std::string& subtree_name = "twolevelschwarz",
                                 bool matrix_is_additive = true)
      : novlp_comm(make_communication(gfs)), cfg(ptree.sub(subtree_name)), additive_input(matrix_is_additive), core(novlp_comm, cfg)
  {
    const std::array<std::array<int, 2>, 4> exponents = {{{0, 0}, {1, 0}, {0, 1}, {1, 1}}};
    for (const auto& e : exponents) {
      Vec monomial(gfs);
      Dune::PDELab::interpolate([e](const auto& x) { return (e[0] ? x[0] : 1.0) * (e[1] ? x[1] : 1.0); }, gfs, monomial);
      Dune::PDELab::set_constrained_dofs(cc, 0., monomial);
      templates.push_back(Dune::PDELab::Backend::native(monomial));
    }
  }

  void apply(Mat& A, Vec& z, Vec& r, Real reduction)
  {
    using Dune::PDELab::Backend::native;
    if (!additive_input) ::make_additive(A, *novlp_comm);
    if (overlap.ready()) overlap.refresh(native(A));
    else {
      overlap.create(*novlp_comm, native(A), cfg);
      std::vector<NativeVec> on_overlap;
      for (const auto& t : templates) on_overlap.push_back(overlap.extended(t));
      core.set_overlapping(overlap.matrix, overlap.comm, overlap.pou, std::move(on_overlap));
    }
    const Dune::InverseOperatorResult outcome = core.solve(A.storage(), native(z), native(r), reduction);
    res.converged = outcome.converged;
    res.iterations = outcome.iterations;
    res.elapsed = outcome.elapsed;
    res.reduction = outcome.reduction;
    res.conv_rate = outcome.conv_rate;
  }

  typename Vec::ElementType norm(const Vec& v) const { return core.norm(Dune::PDELab::Backend::native(v)); }

private:
  using Core = ddm_hip::TwoLevelSchwarzCore<NativeMat, NativeVec, Communication>;
  std::shared_ptr<Communication> novlp_comm;
  Dune::ParameterTree cfg;
  bool additive_input;
  Core core;
  ddm_hip::OverlapObjects<NativeMat, NativeVec, Communication> overlap;
  std::vector<NativeVec> templates;
};
#endif   // HAVE_DUNE_PDELAB
