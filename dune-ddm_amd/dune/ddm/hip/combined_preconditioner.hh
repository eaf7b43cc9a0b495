// MI355X variant of CombinedPreconditioner (reference: dune/ddm/combined_preconditioner.hh).
// Same public interface (ctor from the ParameterTree, add, set_op, pre/post/apply/category), but the
// levels are fused on the device: one H2D copy of the defect, ddm_combined_apply (Schwarz + coarse
// level + their sum, or the multiplicative residual update), one D2H copy of the result -- instead of
// one host round trip per level.  It therefore only accepts the device-backed levels of this
// directory.  The reference's own header is pure host logic on abstract Dune::Preconditioner objects
// and keeps working unchanged on top of the adaptors when host and device levels must be mixed.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include <dune/common/exceptions.hh>
#include <dune/common/parametertree.hh>
#include <dune/istl/operators.hh>
#include <dune/istl/preconditioner.hh>

#include "backend.hh"

template <class X, class Y = X>
class CombinedPreconditioner : public Dune::Preconditioner<X, Y> {
public:
  explicit CombinedPreconditioner(const Dune::ParameterTree& ptree, const std::string& subtree_name = "combined_preconditioner")
      : ctx(ddm_hip::Context::get())
  {
    const auto& subtree = subtree_name.size() == 0 ? ptree : ptree.sub(subtree_name);
    const auto m = subtree.get("mode", std::string("additive"));
    if (m == "additive") mode = 0;
    else if (m == "multiplicative") mode = 1;
    else DUNE_THROW(Dune::NotImplemented, "Unknown apply mode in CombinedPreconditioner, use either additive or multiplicative");
  }
  ~CombinedPreconditioner() override { ddm_combined_destroy(C); }

  Dune::SolverCategory::Category category() const override
  {
    if (levels.empty()) DUNE_THROW(Dune::Exception, "ERROR: No preconditioners added yet, add them using the `add` method");
    return levels[0]->category();
  }
  void add(std::shared_ptr<Dune::Preconditioner<X, Y>> prec)
  {
    if (!levels.empty() && prec->category() != levels[0]->category())
      DUNE_THROW(Dune::Exception, "ERROR: Categories of the new preconditioner does not match");
    if (!dynamic_cast<ddm_hip::DeviceLevel*>(prec.get()))
      DUNE_THROW(Dune::NotImplemented, "the fused device CombinedPreconditioner only takes device-backed levels");
    if (levels.size() == 2) DUNE_THROW(Dune::NotImplemented, "at most a fine (Schwarz) and a coarse (Galerkin) level");
    levels.push_back(std::move(prec));
  }
  void set_op(std::shared_ptr<Dune::LinearOperator<X, Y>> A) { op = std::move(A); }
  void pre(X& x, Y& y) override
  {
    for (auto& l : levels) l->pre(x, y);
  }
  void post(X& x) override
  {
    for (auto& l : levels) l->post(x);
  }
  void apply(X& x, const Y& d) override
  {
    if (!C) fuse(d.N());
    dd->upload(d);
    ddm_hip::check(ctx->handle(), ddm_combined_apply(ctx->handle(), C, dx->data(), dd->data()), "ddm_combined_apply");
    dx->download(x);
  }
  ddm_combined* handle(std::size_t n_novlp)
  {
    if (!C) fuse(n_novlp);
    return C;
  }
  std::shared_ptr<ddm_hip::Context> context() const { return ctx; }

private:
  void fuse(std::size_t n_novlp)
  {
    if (levels.empty()) DUNE_THROW(Dune::Exception, "ERROR: No preconditioners added yet, add them using the `add` method");
    ddm_schwarz* s = dynamic_cast<ddm_hip::DeviceLevel*>(levels[0].get())->schwarz_handle(n_novlp);
    if (!s) DUNE_THROW(Dune::NotImplemented, "the first level must be the SchwarzPreconditioner");
    ddm_galerkin* g = levels.size() > 1 ? dynamic_cast<ddm_hip::DeviceLevel*>(levels[1].get())->galerkin_handle(n_novlp) : nullptr;
    if (levels.size() > 1 && !g) DUNE_THROW(Dune::NotImplemented, "the second level must be the GalerkinPreconditioner");
    ddm_op* o = nullptr;
    if (op) {
      auto* dop = dynamic_cast<ddm_hip::DeviceOperator*>(op.get());
      if (dop) o = dop->op_handle();
    }
    if (mode == 1 && g && !o)
      DUNE_THROW(Dune::Exception, "ERROR: ApplyMode is multiplicative but operator A is not provided. Set with `set_op`");
    ddm_hip::check(ctx->handle(), ddm_combined_create(ctx->handle(), mode, o, s, g, &C), "ddm_combined_create");
    dd = std::make_unique<ddm_hip::DeviceVector>(ctx, n_novlp);
    dx = std::make_unique<ddm_hip::DeviceVector>(ctx, n_novlp);
  }

  std::shared_ptr<ddm_hip::Context> ctx;
  int mode = 0;
  std::vector<std::shared_ptr<Dune::Preconditioner<X, Y>>> levels;
  std::shared_ptr<Dune::LinearOperator<X, Y>> op;
  std::unique_ptr<ddm_hip::DeviceVector> dd, dx;
  ddm_combined* C = nullptr;
};
