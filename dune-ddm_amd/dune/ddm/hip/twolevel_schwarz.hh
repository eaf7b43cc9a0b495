// MI355X drop-in for dune/ddm/twolevel_schwarz.hh (TwoLevelSchwarzSolver, the PDELab linear-solver backend handed to
// StationaryLinearProblemSolver / Newton by examples/convectiondiffusiondg.cc:75-78 and nonlinearpoisson.cc:151-154).
//
// The reference's apply() has two halves:
//   (1) first call only (twolevel_schwarz.hh:93-128): overlap extension, overlapping matrix, partition of unity, template vectors
//       extended to the overlapping index set -- host code of the layer BEFORE the hot path (overlap_extension.hh, datahandles.hh,
//       pou.hh), used unchanged in a DUNE build;
//   (2) every call (:131-168): POUCoarseSpace -> SchwarzPreconditioner ("fine") + GalerkinPreconditioner ("coarse") in a
//       CombinedPreconditioner whose mode key sits in the sub-tree itself, NonOverlappingOperator, solver from the "solver"
//       sub-tree (default restarted GMRES(30), maxit 1000), right-hand side made consistent, solve, result stored.
// Half (2) is ddm_hip::TwoLevelSchwarzCore below, written on native dune-istl types only, so that it compiles (and is tested,
// tests/cpp/twolevel_adaptor.cc) without PDELab; the outer Krylov loop runs on the device (Dune::getHipSolver in place of
// Dune::getSolverFromFactory: one upload and one download per solve).  TwoLevelSchwarzSolver at the end of the file is the
// PDELab-facing class with the reference's constructor / apply / norm signatures; it needs dune-pdelab and the reference's own
// setup headers and is therefore only compiled in a DUNE build (HAVE_DUNE_PDELAB).
#pragma once

#include <cmath>
#include <cstddef>
#include <memory>
#include <string>
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

template <class NativeMat, class NativeVec, class Communication>
class TwoLevelSchwarzCore {
public:
  using FineLevel = SchwarzPreconditioner<NativeMat, NativeVec, Communication>;     // twolevel_schwarz.hh:34
  using CoarseLevel = GalerkinPreconditioner<NativeVec, Communication>;              // :35
  using Op = NonOverlappingOperator<NativeMat, NativeVec, NativeVec, Communication>;   // :88

  TwoLevelSchwarzCore(std::shared_ptr<Communication> novlp_comm_, const Dune::ParameterTree& subtree_) : novlp_comm(std::move(novlp_comm_)), subtree(subtree_) {}

  // the objects the first apply() of the reference builds once (:93-128)
  void set_overlapping(std::shared_ptr<NativeMat> A_ovlp_, std::shared_ptr<Communication> ovlp_comm_, std::shared_ptr<PartitionOfUnity> pou_,
                       std::vector<NativeVec> extended_template_vecs)
  {
    A_ovlp = std::move(A_ovlp_);
    ovlp_comm = std::move(ovlp_comm_);
    pou = std::move(pou_);
    native_template_vecs = std::move(extended_template_vecs);
    for (const auto& v : native_template_vecs)
      if (v.N() != A_ovlp->N()) DUNE_THROW(Dune::Exception, "Template vectors must match size of matrix");
  }
  bool has_overlapping() const { return (bool)ovlp_comm; }
  const std::shared_ptr<NativeMat>& overlapping_matrix() const { return A_ovlp; }
  const std::shared_ptr<Communication>& overlapping_communication() const { return ovlp_comm; }

  // twolevel_schwarz.hh:131-168.  A: the (additive) non-overlapping matrix; d: right-hand side, overwritten by the defect.
  Dune::InverseOperatorResult solve(std::shared_ptr<NativeMat> A, NativeVec& z, NativeVec& d, double reduction)
  {
    if (!has_overlapping()) DUNE_THROW(Dune::InvalidStateException, "TwoLevelSchwarzCore::solve before set_overlapping");
    // Set up the preconditioner (:105-116)
    POUCoarseSpace<NativeVec> coarse_space(native_template_vecs, *pou);
    fine = std::make_shared<FineLevel>(A_ovlp, ovlp_comm, pou, subtree, "fine");
    fine->novlp_comm = novlp_comm;   // :109
    coarse = std::make_shared<CoarseLevel>(*A_ovlp, coarse_space.get_basis(), ovlp_comm, subtree, "coarse");
    auto prec = std::make_shared<CombinedPreconditioner<NativeVec>>(subtree, "");
    auto op = std::make_shared<Op>(std::move(A), novlp_comm);
    prec->set_op(op);
    prec->add(fine);
    prec->add(coarse);

    // Set up the solver (:119-131)
    const int rank = ovlp_comm->communicator().rank();
    Dune::ParameterTree solver_subtree;
    if (subtree.hasSub("solver")) solver_subtree = subtree.sub("solver");
    else {
      solver_subtree["type"] = "restartedgmressolver";
      solver_subtree["restart"] = "30";
      solver_subtree["maxit"] = "1000";
      solver_subtree["verbose"] = "0";
    }
    solver_subtree["verbose"] = rank == 0 ? solver_subtree.get("verbose", std::string("0")) : std::string("0");   // verbosity on the root rank only
    solver_subtree["reduction"] = std::to_string(reduction);
    auto solver = Dune::getHipSolver<NativeVec>(op, solver_subtree, prec);   // getSolverFromFactory(op, solver_subtree, prec) (:133)

    // Make the rhs consistent (this is how the preconditioner, nonoverlapping operator and scalar product expect it) (:136-137)
    novlp_comm->addOwnerCopyToAll(d, d);

    // Solve the linear system (:140-141)
    Dune::InverseOperatorResult stat;
    solver->apply(z, d, reduction, stat);
    return stat;
  }

  // TwoLevelSchwarzSolver::norm (:149-158): consistent copy, then the owner-masked norm
  double norm(const NativeVec& v) const
  {
    auto x = v;
    novlp_comm->addOwnerCopyToOwnerCopy(x, x);
    return novlp_comm->norm(x);
  }

  // the levels of the last solve (inspection / tests)
  std::shared_ptr<FineLevel> fine;
  std::shared_ptr<CoarseLevel> coarse;

private:
  std::shared_ptr<Communication> novlp_comm, ovlp_comm;
  std::shared_ptr<NativeMat> A_ovlp;
  std::shared_ptr<PartitionOfUnity> pou;
  std::vector<NativeVec> native_template_vecs;
  Dune::ParameterTree subtree;
};

}  // namespace ddm_hip

#if HAVE_DUNE_PDELAB
// ---- PDELab-facing class: same constructor, apply and norm as the reference (twolevel_schwarz.hh:27-174) -----------------------
// Needs dune-pdelab and the reference's host setup headers (make_communication, make_overlapping_communication, the matrix data
// handles, PartitionOfUnity, make_additive): compiled in a DUNE build only.
#include <dune/ddm/datahandles.hh>
#include <dune/ddm/overlap_extension.hh>
#include <dune/ddm/pdelab_helper.hh>

#include <dune/common/parallel/variablesizecommunicator.hh>
#include <dune/istl/owneroverlapcopy.hh>
#include <dune/pdelab/backend/interface.hh>
#include <dune/pdelab/backend/solver.hh>
#include <dune/pdelab/constraints/common/constraints.hh>
#include <dune/pdelab/gridfunctionspace/interpolate.hh>

template <class Mat, class Vec>
class TwoLevelSchwarzSolver : public Dune::PDELab::LinearResultStorage {
  using NativeMat = Dune::PDELab::Backend::Native<Mat>;
  using NativeVec = Dune::PDELab::Backend::Native<Vec>;
  using Communication = Dune::OwnerOverlapCopyCommunication<std::size_t, int>;
  using Core = ddm_hip::TwoLevelSchwarzCore<NativeMat, NativeVec, Communication>;

public:
  template <class GFS, class CC>
  explicit TwoLevelSchwarzSolver(const GFS& gfs, const CC& cc, const Dune::ParameterTree& ptree, const std::string& subtree_name = "twolevelschwarz",
                                 bool matrix_is_additive = true)
      : novlp_comm(make_communication(gfs)), subtree(ptree.sub(subtree_name)), matrix_is_additive(matrix_is_additive), core(novlp_comm, subtree)
  {
    using Dune::PDELab::Backend::native;
    // the template vectors 1, x, y, xy with the constrained DoFs zeroed (:68-81)
    std::vector<Vec> template_vecs(4, gfs);
    Dune::PDELab::interpolate([](auto&&) { return 1; }, gfs, template_vecs[0]);
    Dune::PDELab::interpolate([](auto&& x) { return x[0]; }, gfs, template_vecs[1]);
    Dune::PDELab::interpolate([](auto&& x) { return x[1]; }, gfs, template_vecs[2]);
    Dune::PDELab::interpolate([](auto&& x) { return x[0] * x[1]; }, gfs, template_vecs[3]);
    for (auto& v : template_vecs) Dune::PDELab::set_constrained_dofs(cc, 0., v);
    for (auto& v : template_vecs) native_template_vecs.push_back(native(v));
  }

  void apply(Mat& A, Vec& z, Vec& r, typename Dune::template FieldTraits<typename Vec::ElementType>::real_type reduction)
  {
    using Dune::PDELab::Backend::native;
    if (!matrix_is_additive) ::make_additive(A, *novlp_comm);   // :90
    if (!core.has_overlapping()) {                               // :93-128
      const int overlap = subtree.get("overlap", 1);
      auto ovlp_comm = make_overlapping_communication(*novlp_comm, native(A), overlap).first;
      typename Communication::AllSet allset;
      interface_ext.build(ovlp_comm->remoteIndices(), allset, allset);
      varcomm = std::make_unique<Dune::VariableSizeCommunicator<>>(interface_ext);
      CreateMatrixDataHandle cmdh(native(A), ovlp_comm->indexSet());
      varcomm->forward(cmdh);
      auto A_ovlp = std::make_shared<NativeMat>(cmdh.getOverlappingMatrix());
      AddMatrixDataHandle amdh(native(A), *A_ovlp, ovlp_comm->indexSet());
      varcomm->forward(amdh);
      auto pou = std::make_shared<PartitionOfUnity>(*A_ovlp, *ovlp_comm, subtree.sub("pou"), overlap);
      std::vector<NativeVec> extended(native_template_vecs.size(), NativeVec(A_ovlp->N()));
      for (std::size_t i = 0; i < native_template_vecs.size(); ++i) {
        extended[i] = 0;
        for (std::size_t j = 0; j < native_template_vecs[i].N(); ++j) extended[i][j] = native_template_vecs[i][j];
        ovlp_comm->copyOwnerToAll(extended[i], extended[i]);
      }
      core.set_overlapping(A_ovlp, ovlp_comm, pou, std::move(extended));
    }
    else {   // update the overlapping matrix for subsequent calls (:122-127)
      *core.overlapping_matrix() = 0;
      AddMatrixDataHandle amdh(native(A), *core.overlapping_matrix(), core.overlapping_communication()->indexSet());
      varcomm->forward(amdh);
    }
    const auto stat = core.solve(A.storage(), native(z), native(r), reduction);   // :131-141
    res.converged = stat.converged;
    res.iterations = stat.iterations;
    res.elapsed = stat.elapsed;
    res.reduction = stat.reduction;
    res.conv_rate = stat.conv_rate;
  }

  typename Vec::ElementType norm(const Vec& v) const { return core.norm(Dune::PDELab::Backend::native(v)); }

private:
  std::shared_ptr<Communication> novlp_comm;
  Dune::ParameterTree subtree;
  bool matrix_is_additive;
  Core core;
  Dune::Interface interface_ext;
  std::unique_ptr<Dune::VariableSizeCommunicator<>> varcomm;
  std::vector<NativeVec> native_template_vecs;
};
#endif   // HAVE_DUNE_PDELAB
